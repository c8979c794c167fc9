// ORACLE -- TEST INFRASTRUCTURE ONLY.
// Minimal stand-in for src/meshing/r3live/pointcloud_rgbd.hpp so that the reference's OWN Triangle_manager (src/meshing/r3live/triangle.hpp /
// triangle.cpp + src/tools/tools_kd_hash.hpp) compiles from where it lies without Eigen / PCL / OpenCV: what those two files use of it is
// vec_3 (constructor, +, / scalar, operator()(i)), vec_2f (a member array type), RGB_pts::get_pos() and Global_map::m_rgb_pts_vec.
// The recipe (oracle/Makefile, target ref) symlinks the reference files next to this header under oracle/_ref/rt_src/.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <unordered_map>
#include <vector>

struct vec_3 {   // Eigen::Matrix<double, 3, 1>
    double v[3];
    vec_3() : v{0, 0, 0} {}
    vec_3(double x, double y, double z) : v{x, y, z} {}
    double operator()(int i) const { return v[i]; }
    vec_3 operator+(const vec_3& o) const { return vec_3(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    vec_3 operator/(double s) const { return vec_3(v[0] / s, v[1] / s, v[2] / s); }
};
struct vec_2f { float v[2] = {0, 0}; };

class RGB_pts {   // pointcloud_rgbd.hpp:77-140: the vertex position is all Triangle_manager reads
  public:
    double m_pos[3] = {0, 0, 0};
    vec_3 get_pos() { return vec_3(m_pos[0], m_pos[1], m_pos[2]); }
};
class Global_map {   // pointcloud_rgbd.hpp:234-298
  public:
    std::vector<std::shared_ptr<RGB_pts>> m_rgb_pts_vec;
};
