// drop_in/stubs -- the SHAPES of the reference types the five replaced bodies touch, so that drop_in/immesh_shim.cpp is compiled (and run, on a
// GPU box) without Eigen / PCL / ROS: every declaration below mirrors one of the reference (file:line cited), reduced to the members the shim
// uses.  In the reference build the shim includes the real headers instead (-DIMMESH_SHIM_REAL_HEADERS: "voxel_mapping.hpp", which pulls in
// Eigen, PCL, common_lib.h, triangle.hpp, pointcloud_rgbd.hpp).  Nothing here is linked into libimmesh_hip.so.
#pragma once
#include <algorithm>
#include <array>
#include <unordered_map>
#include <unordered_set>
#include <cstdint>
#include <map>
#include <memory>
#include <new>
#include <set>
#include <string>
#include <vector>

namespace Eigen {   // element access only: the shim marshals through operator() and never relies on the storage order
// (constructors, Identity / Zero, the float twins, Matrix< T, 3, 1 > and the two products below: what the member initialisers and the inline member templates of
//  the reference's class Voxel_mapping use, for the `realclass` compile check -- stubs/real_class)
template <typename T, int R, int C> struct Matrix;
template <typename T> struct Matrix<T, 3, 1> {
    T v[3] = {0, 0, 0};
    Matrix() {}
    Matrix(T x, T y, T z) : v{x, y, z} {}
    T& operator()(int i) { return v[i]; } T operator()(int i) const { return v[i]; }
    T& operator[](int i) { return v[i]; } T operator[](int i) const { return v[i]; }
    static Matrix Zero() { return Matrix(); }
    Matrix operator+(const Matrix& o) const { return Matrix(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
};
template <typename T> struct Matrix<T, 3, 3> {
    T m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    T& operator()(int r, int c) { return m[c * 3 + r]; } T operator()(int r, int c) const { return m[c * 3 + r]; }
    static Matrix Identity() { return Matrix(); }
    Matrix<T, 3, 1> operator*(const Matrix<T, 3, 1>& x) const { Matrix<T, 3, 1> o; for (int r = 0; r < 3; r++) o.v[r] = (*this)(r, 0) * x.v[0] + (*this)(r, 1) * x.v[1] + (*this)(r, 2) * x.v[2]; return o; }
};
typedef Matrix<double, 3, 1> Vector3d; typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<float, 3, 3> Matrix3f;
struct Quaterniond { double w = 1, x = 0, y = 0, z = 0; };
template <int N> struct MatrixNd { std::vector<double> m = std::vector<double>(N * N, 0.0); double& operator()(int r, int c) { return m[c * N + r]; } double operator()(int r, int c) const { return m[c * N + r]; } };
}  // namespace Eigen
typedef Eigen::Vector3d V3D;   // include/common_lib.h:60-70
typedef Eigen::Matrix3d M3D;

namespace pcl {
struct alignas(16) PointXYZINormal { float x = 0, y = 0, z = 0, _w = 1; float normal_x = 0, normal_y = 0, normal_z = 0, _n = 0; float intensity = 0, curvature = 0, _p[2] = {0, 0}; };   // 48 B, EIGEN_ALIGN16
struct alignas(16) PointXYZI { float x = 0, y = 0, z = 0, _w = 1; float intensity = 0, _p[3] = {0, 0, 0}; };                                                                       // 32 B
template <typename T> struct PointCloud {
    typedef std::shared_ptr<PointCloud<T>> Ptr;
    std::vector<T> points;
    size_t size() const { return points.size(); }
    void resize(size_t n) { points.resize(n); }
    void clear() { points.clear(); }
    PointCloud() {}
    PointCloud(unsigned w, unsigned h) : points((size_t)w * h) {}
    Ptr makeShared() const { return std::make_shared<PointCloud<T>>(*this); }
};
}  // namespace pcl
typedef pcl::PointXYZINormal PointType;               // include/common_lib.h:58
typedef pcl::PointCloud<PointType> PointCloudXYZI;

struct StatesGroup {   // include/common_lib.h:199-288
    M3D rot_end; V3D pos_end, vel_end, bias_g, bias_a, gravity;
    Eigen::MatrixNd<18> cov;
};

#ifdef IMMESH_SHIM_REF_MIRROR
// the host mirrors are THE REFERENCE'S OWN: triangle.hpp (Triangle, Triangle_manager; with triangle.cpp linked in) over ref_mirror/pointcloud_rgbd.hpp
// (drop_in/Makefile: _ref/libimmesh_dropin_async_refmirror.so, built where /root/reference exists)
#include "triangle.hpp"
inline int64_t immesh_mirror_live_count(Triangle_manager& m) { return (int64_t)m.get_triangle_list_size(); }
template <typename F> inline void immesh_mirror_for_each_live(Triangle_manager& m, F f) { for (auto* s : m.m_triangle_set_vector) for (auto& t : *s->get_triangle_set_ptr()) f(t); }
inline void immesh_mirror_reset(Triangle_manager& m, Global_map* map, double region_size) { m.~Triangle_manager(); new (&m) Triangle_manager(); m.m_pointcloud_map = map; m.m_region_size = region_size; }   // ImMesh_node.cpp:268-271
#else
// ---- src/meshing/r3live/pointcloud_rgbd.hpp:77-140, 234-298 and triangle.hpp:9-34, 115-395 (the host mirrors the renderer / PLY export read) --------
struct vec_3 { double v[3]; vec_3(double x = 0, double y = 0, double z = 0) : v{x, y, z} {} double operator()(int i) const { return v[i]; } };
class RGB_pts {
  public:
    double m_pos[3] = {0, 0, 0}, m_pos_aft_smooth[3] = {0, 0, 0};
    int m_pt_index = 0;
    bool m_smoothed = false;
    void set_pos(const vec_3& p) { for (int i = 0; i < 3; i++) m_pos[i] = p(i); }
    void set_smooth_pos(const vec_3& p) { for (int i = 0; i < 3; i++) m_pos_aft_smooth[i] = p(i); m_smoothed = true; }
    vec_3 get_pos(bool get_smooth = false) { return (get_smooth && m_smoothed) ? vec_3(m_pos_aft_smooth[0], m_pos_aft_smooth[1], m_pos_aft_smooth[2]) : vec_3(m_pos[0], m_pos[1], m_pos[2]); }   // pointcloud_rgbd.cpp:70-80
};
using RGB_pt_ptr = std::shared_ptr<RGB_pts>;
class Global_map {   // pointcloud_rgbd.hpp:234-298
  public:
    std::vector<RGB_pt_ptr> m_rgb_pts_vec;
    vec_3 smooth_pts(RGB_pt_ptr& rgb_pt, double smooth_factor, double knn = 20, double maximum_smooth_dis = 0);   // :287, pointcloud_rgbd.cpp:932-958 -- body replaced by the shim
};
class Triangle { public: int m_tri_pts_id[3] = {0, 0, 0}; int m_index_flip = 0; Triangle(int a, int b, int c) : m_tri_pts_id{a, b, c} { std::sort(m_tri_pts_id, m_tri_pts_id + 3); } };
using Triangle_ptr = std::shared_ptr<Triangle>;
using Triangle_set = std::set<Triangle_ptr>;
struct Triplet_hash { size_t operator()(const std::array<int, 3>& k) const { unsigned long long x = ((unsigned long long)(unsigned)k[0] * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)(unsigned)k[1] << 21) ^ ((unsigned long long)(unsigned)k[2] << 42); x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; return (size_t)(x ^ (x >> 32)); } };
class Triangle_manager {   // same three entry points and semantics as triangle.hpp:212 (remove_triangle_list), :311 (find_triangle), :330 (insert_triangle); hashed like m_triangle_hash
  public:
    std::unordered_map<std::array<int, 3>, Triangle_ptr, Triplet_hash> m_triangle_hash;   // entries persist after erase, like m_triangle_hash
    std::unordered_set<Triangle_ptr> m_live;
    Triangle_ptr find_triangle(int a, int b, int c) { std::array<int, 3> k = {a, b, c}; std::sort(k.begin(), k.end()); auto it = m_triangle_hash.find(k); return it == m_triangle_hash.end() ? nullptr : it->second; }
    void remove_triangle_list(const Triangle_set& s, const int = 0) { for (auto& t : s) if (t) m_live.erase(t); }
    Triangle_ptr insert_triangle(int a, int b, int c, int = 0, const int& = 0) {
        Triangle_ptr t = find_triangle(a, b, c);
        if (!t) { t = std::make_shared<Triangle>(a, b, c); m_triangle_hash[{t->m_tri_pts_id[0], t->m_tri_pts_id[1], t->m_tri_pts_id[2]}] = t; }
        m_live.insert(t);
        return t;
    }
};
inline int64_t immesh_mirror_live_count(Triangle_manager& m) { return (int64_t)m.m_live.size(); }
template <typename F> inline void immesh_mirror_for_each_live(Triangle_manager& m, F f) { for (auto& t : m.m_live) f(t); }
inline void immesh_mirror_reset(Triangle_manager& m, Global_map*, double) { m = Triangle_manager(); }
#endif
extern Global_map g_map_rgb_pts_mesh;           // src/ImMesh_mesh_reconstruction.cpp:40
extern Triangle_manager g_triangles_manager;    // :41

struct immesh_ctx;
#ifdef IMMESH_SHIM_REF_CLASS
// `make -C drop_in realclass`: class Voxel_mapping IS THE REFERENCE'S -- src/voxel_mapping.hpp:132-414 cut out by line range at build time (vm_class_body.inc,
// removed after the compile; nothing is copied) -- closed below with the members INTEGRATION.md ("One context per scan thread") has the maintainer add
#include "real_class/ref_class_prelude.hpp"
#include "vm_class_body.inc"
    immesh_ctx* m_hip = nullptr;
    std::vector<float> m_immesh_xyz, m_immesh_xyzi;
    bool m_immesh_scan_queued = false;
    void immesh_shim_init();
    void immesh_fetch_effect_features();
};
#else
struct Preprocess_shape { int calib_laser = 0; };   // src/preprocess.h:170

class Voxel_mapping {   // src/voxel_mapping.hpp:132-420 -- only what the replaced bodies read or write
  public:
    V3D m_extT; M3D m_extR;                                   // :149-150
    int NUM_MAX_ITERATIONS = 4;                               // :153
    int m_effct_feat_num = 0;                                 // :157
    double m_res_mean_last = 0.05;                            // :159
    double m_max_voxel_size = 0.5, m_min_eigen_value = 0.01;  // :176
    double m_beam_err = 0.05, m_dept_err = 0.02;              // :177
    int m_max_points_size = 100, m_max_layer = 2;             // :190-191
    std::vector<int> m_layer_init_size = {5, 5, 5, 5, 5};     // :192
    PointCloudXYZI::Ptr m_feats_undistort = std::make_shared<PointCloudXYZI>();   // :226
    PointCloudXYZI::Ptr m_feats_down_body = std::make_shared<PointCloudXYZI>();   // :227
    PointCloudXYZI::Ptr m_laserCloudOri = std::make_shared<PointCloudXYZI>();     // :230
    PointCloudXYZI::Ptr m_corr_normvect = std::make_shared<PointCloudXYZI>();     // :231
    StatesGroup state;                                        // :264
    double m_meshing_distance_scale = 1.0, m_meshing_points_minimum_scale = 0.1, m_meshing_voxel_resolution = 0.4, m_meshing_region_size = 10.0;   // :279-282
    int m_meshing_number_of_pts_append_to_map = 10000;        // :286
    std::shared_ptr<Preprocess_shape> m_p_pre = std::make_shared<Preprocess_shape>();
    immesh_ctx* m_hip = nullptr;                              // what the drop-in adds: the context, ...
    std::vector<float> m_immesh_xyz, m_immesh_xyzi;           // ... packed staging of the pcl clouds (asynchronous shim: reused from scan to scan) ...
    bool m_immesh_scan_queued = false;                        // ... and "lio_state_estimation queued this scan's map growth + mesh job"
    void immesh_shim_init();                                  // end of init_ros_node() (src/voxel_mapping.cpp:1654)
    void immesh_fetch_effect_features();                      // asynchronous shim: m_laserCloudOri / m_corr_normvect on demand (publish_effect_world)
    void map_incremental_grow();                              // :365  (src/ImMesh_mesh_reconstruction.cpp:377)
    bool voxel_map_init();                                    // :409  (src/voxel_mapping.cpp:1243)
    void lio_state_estimation(StatesGroup& state_propagat);   // :412  (src/voxel_mapping.cpp:1284)
};
#endif
struct Rec_mesh_data_package {   // src/ImMesh_mesh_reconstruction.cpp:63-76
    pcl::PointCloud<pcl::PointXYZI>::Ptr m_frame_pts;
    Eigen::Quaterniond m_pose_q;
    Eigen::Vector3d m_pose_t;
    int m_frame_idx;
    Rec_mesh_data_package(pcl::PointCloud<pcl::PointXYZI>::Ptr frame_pts, Eigen::Quaterniond pose_q, Eigen::Vector3d pose_t, int frame_idx)
        : m_frame_pts(frame_pts), m_pose_q(pose_q), m_pose_t(pose_t), m_frame_idx(frame_idx) {}
};
void service_reconstruct_mesh();                                                                                                                       // src/ImMesh_mesh_reconstruction.cpp:272
void incremental_mesh_reconstruction(pcl::PointCloud<pcl::PointXYZI>::Ptr frame_pts, Eigen::Quaterniond pose_q, Eigen::Vector3d pose_t, int frame_idx);   // src/ImMesh_mesh_reconstruction.cpp:92
void reconstruct_mesh_from_pointcloud(pcl::PointCloud<pcl::PointXYZI>::Ptr frame_pts, double minimum_pts_distance = 0.01);                               // :328, src/voxel_mapping.hpp:131
void save_to_ply_file(std::string ply_file, double smooth_factor = 0.1, double knn = 20);                                                               // src/meshing/mesh_rec_geometry.hpp:40
