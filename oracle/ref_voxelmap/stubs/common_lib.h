// ORACLE / TEST INFRASTRUCTURE ONLY -- stands in for include/common_lib.h when the reference's voxel_loc.hpp / voxel_loc.cpp and excerpts of
// voxel_mapping.cpp are compiled from where they lie (oracle/Makefile: _ref/libref_voxelmap.so): only the macros / typedefs those files use.
#pragma once
#include <Eigen/Core>
#include <pcl/common/io.h>
#define HASH_P 116101          /* include/common_lib.h:52 */
#define MAX_N 10000000000      /* include/common_lib.h:53 */
typedef pcl::PointXYZINormal PointType;   /* include/common_lib.h:58 */
typedef Eigen::Vector3d V3D;              /* :60-70 */
typedef Eigen::Matrix3d M3D;
typedef Eigen::Vector3f V3F;
