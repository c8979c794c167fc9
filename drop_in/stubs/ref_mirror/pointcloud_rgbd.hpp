// drop_in/stubs/ref_mirror -- stands in for src/meshing/r3live/pointcloud_rgbd.hpp when the asynchronous drop-in is compiled against the REFERENCE'S OWN
// Triangle_manager (triangle.hpp / triangle.cpp + tools_kd_hash.hpp, symlinked next to this header by drop_in/Makefile: nothing is copied): the host
// mirror bench.py's "through the drop-in" leg then pays for is the real one -- per-vertex adjacency sets, region buckets (triangle.cpp:35-70), a mutex per
// operation -- not the hash-map stand-in of immesh_ref_shapes.hpp.  What triangle.hpp / triangle.cpp use of this header: vec_3 (constructor, +, / scalar,
// operator()(i)), vec_2f (a member array type), RGB_pts::get_pos(), Global_map::m_rgb_pts_vec; what the shim uses: set_pos / set_smooth_pos / m_pt_index.
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
    vec_3(double x = 0, double y = 0, double z = 0) : v{x, y, z} {}
    double operator()(int i) const { return v[i]; }
    vec_3 operator+(const vec_3& o) const { return vec_3(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    vec_3 operator/(double s) const { return vec_3(v[0] / s, v[1] / s, v[2] / s); }
};
struct vec_2f { float v[2] = {0, 0}; };
class RGB_pts {   // pointcloud_rgbd.hpp:77-163, pointcloud_rgbd.cpp:59-87
  public:
    double m_pos[3] = {0, 0, 0}, m_pos_aft_smooth[3] = {0, 0, 0};
    int m_pt_index = 0;
    bool m_smoothed = false;
    void set_pos(const vec_3& p) { for (int i = 0; i < 3; i++) { m_pos[i] = p(i); m_pos_aft_smooth[i] = p(i); } }
    void set_smooth_pos(const vec_3& p) { for (int i = 0; i < 3; i++) m_pos_aft_smooth[i] = p(i); m_smoothed = true; }
    vec_3 get_pos(bool get_smooth = false) { return get_smooth ? vec_3(m_pos_aft_smooth[0], m_pos_aft_smooth[1], m_pos_aft_smooth[2]) : vec_3(m_pos[0], m_pos[1], m_pos[2]); }
};
using RGB_pt_ptr = std::shared_ptr<RGB_pts>;
class Global_map {   // pointcloud_rgbd.hpp:234-298
  public:
    std::vector<RGB_pt_ptr> m_rgb_pts_vec;
    vec_3 smooth_pts(RGB_pt_ptr& rgb_pt, double smooth_factor, double knn = 20, double maximum_smooth_dis = 0);   // :287, pointcloud_rgbd.cpp:932-958 -- body replaced by the shim
};
