// Stub standing in for <pcl/point_types.h> (PCL is not installed in this image).
// Only what include/ikd-Tree/ikd_Tree.{h,cpp} needs to instantiate its templates:
// three POD point types with float x,y,z members.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstring>   // the real PCL header pulls this in transitively (memset/memcpy used by ikd_Tree.cpp)
#include <cmath>
namespace pcl {
struct PointXYZ { float x = 0, y = 0, z = 0; };
struct PointXYZI { float x = 0, y = 0, z = 0, intensity = 0; };
struct PointXYZINormal { float x = 0, y = 0, z = 0, intensity = 0, normal_x = 0, normal_y = 0, normal_z = 0, curvature = 0; };
}
