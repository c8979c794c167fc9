// Stub standing in for src/tools/tools_eigen.hpp (needs Eigen3, not installed).
// ikd_Tree.h only uses Eigen::aligned_allocator for its PointVector typedef.
#pragma once
#include <memory>
#include <vector>
namespace Eigen { template <class T> using aligned_allocator = std::allocator<T>; }
