// ORACLE / TEST INFRASTRUCTURE ONLY -- stands in for src/tools/tools_eigen.hpp (Eigen3 is not in this image): the typedefs the mesher's bodies use,
// on the Eigen-shaped stub of ref_voxelmap.
#pragma once
#include "ref_voxelmap/stubs/mini_eigen.hpp"   /* (-I oracle/) */
typedef Eigen::Matrix< double, 3, 1 > vec_3;   /* src/tools/tools_eigen.hpp */
typedef Eigen::Matrix< double, 2, 1 > vec_2;
typedef Eigen::Matrix< double, 4, 1 > vec_4;
typedef Eigen::Matrix< float, 2, 1 > vec_2f;
typedef Eigen::Matrix< double, 3, 3 > mat_3_3;
