// ORACLE / TEST INFRASTRUCTURE ONLY -- stands in for <pcl/point_types.h> (PCL is not in this image) when include/ikd-Tree/ikd_Tree.{h,cpp} and the
// mesher's bodies are compiled together (oracle/Makefile: _ref/libref_globalmap.so): the point PODs + pcl::PointCloud of ref_voxelmap's stand-in.
#pragma once
#include <cstring>
#include <cmath>
#include <pcl/common/io.h>      /* ref_voxelmap/stubs: PointXYZI, PointXYZINormal, PointCloud */
namespace pcl { struct PointXYZ { float x = 0, y = 0, z = 0; }; }
