// ORACLE / TEST INFRASTRUCTURE ONLY -- stands in for src/meshing/r3live/pointcloud_rgbd.hpp (and, through it, for the Eigen / CGAL / tools headers
// mesh_rec_geometry.hpp pulls in) so that the reference's own triangle.hpp and excerpts of mesh_rec_geometry.cpp compile from where they lie
// (oracle/Makefile: _ref/libref_meshgeom.so).  The CGAL-shaped classes forward to the oracle's restatement of the triangulation: what is pinned is
// the reference's code AROUND the CGAL call, not CGAL.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <mutex>
#include <numeric>
#include <set>
#include <thread>
#include <unordered_map>
#include <vector>
#include "ref_voxelmap/stubs/mini_eigen.hpp"   /* (-I oracle/) */
#include "ref_meshgeom/stub_cgal.hpp"

typedef Eigen::Matrix<double, 3, 1> vec_3;   // src/tools/tools_eigen.hpp
typedef Eigen::Matrix<double, 2, 1> vec_2;
typedef Eigen::Matrix<float, 2, 1> vec_2f;

class RGB_pts {   // pointcloud_rgbd.hpp:77-163 (positions + index: all the compiled excerpts read)
  public:
    double m_pos[3] = {0, 0, 0}, m_pos_aft_smooth[3] = {0, 0, 0};
    int m_pt_index = 0;
    vec_3 get_pos(bool get_smooth = false) { return get_smooth ? vec_3(m_pos_aft_smooth[0], m_pos_aft_smooth[1], m_pos_aft_smooth[2]) : vec_3(m_pos[0], m_pos[1], m_pos[2]); }   // pointcloud_rgbd.cpp:77-84
};
typedef std::shared_ptr<RGB_pts> RGB_pt_ptr;
class Global_map { public: std::vector<RGB_pt_ptr> m_rgb_pts_vec; };
