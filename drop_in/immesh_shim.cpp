// The drop-in: the bodies of the reference functions that make up the per-scan hot path, replaced by calls into libimmesh_hip.so
// (include/immesh_c_api.h).  Signatures, classes and the globals the GUI / PLY export read are the reference's own; ImMesh_node.cpp and
// service_LiDAR_update compile unchanged.  In the reference tree: add this file to add_executable(ImMesh_mapping ...) in place of the five
// bodies and build with -DIMMESH_SHIM_REAL_HEADERS.  Here it is compiled against drop_in/stubs (the shapes of those types)
// and driven on a GPU box by drop_in/shim_main.cpp + tests/test_gpu_dropin.py, which checks it against the CPU oracle.
#ifdef IMMESH_SHIM_REAL_HEADERS
#include "voxel_mapping.hpp"
#else
#include "immesh_ref_shapes.hpp"
#endif
#include <cstdio>
#include <cstdlib>
#include "immesh_c_api.h"

// the mesher's context is the scan thread's (one context per scan thread; the reference serialises the two with g_mutex_reconstruct_mesh)
static immesh_ctx* g_immesh_ctx = nullptr;
static int g_immesh_frame_idx = 0;

// StatesGroup (include/common_lib.h:199-288) <-> 348 doubles, row-major
static void to_c(const StatesGroup& s, double* o) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o[r * 3 + c] = s.rot_end(r, c);
    for (int i = 0; i < 3; i++) { o[9 + i] = s.pos_end(i); o[12 + i] = s.vel_end(i); o[15 + i] = s.bias_g(i); o[18 + i] = s.bias_a(i); o[21 + i] = s.gravity(i); }
    for (int r = 0; r < 18; r++) for (int c = 0; c < 18; c++) o[24 + r * 18 + c] = s.cov(r, c);
}
static void from_c(const double* o, StatesGroup& s) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) s.rot_end(r, c) = o[r * 3 + c];
    for (int i = 0; i < 3; i++) { s.pos_end(i) = o[9 + i]; s.vel_end(i) = o[12 + i]; s.bias_g(i) = o[15 + i]; s.bias_a(i) = o[18 + i]; s.gravity(i) = o[21 + i]; }
    for (int r = 0; r < 18; r++) for (int c = 0; c < 18; c++) s.cov(r, c) = o[24 + r * 18 + c];
}
static std::vector<float> pack_xyz(const PointCloudXYZI& cl) {
    std::vector<float> v(cl.points.size() * 3);
    for (size_t i = 0; i < cl.points.size(); i++) { v[3 * i] = cl.points[i].x; v[3 * i + 1] = cl.points[i].y; v[3 * i + 2] = cl.points[i].z; }
    return v;
}
static void fail(immesh_ctx* c, const char* what, int rc) {   // the reference prints and continues on failures; a dead accelerator is fatal
    std::fprintf(stderr, "[immesh shim] %s failed (%d): %s\n", what, rc, c ? immesh_last_error(c) : immesh_create_error());
    if (rc == IMMESH_E_HIP || rc == IMMESH_E_NODEV || !c) std::exit(1);
}

// ---- end of Voxel_mapping::init_ros_node() (src/voxel_mapping.cpp:1654), after read_ros_parameters() ----------------------------------
void Voxel_mapping::immesh_shim_init() {
    immesh_config cfg;
    immesh_default_config(&cfg);
    cfg.voxel_size = m_max_voxel_size; cfg.max_layer = m_max_layer; cfg.max_points_size = m_max_points_size;
    for (int i = 0; i < 5; i++) cfg.layer_init[i] = m_layer_init_size[i];
    cfg.planer_threshold = m_min_eigen_value; cfg.dept_err = m_dept_err; cfg.beam_err = m_beam_err;
    cfg.calib_laser = m_p_pre->calib_laser;
    cfg.max_iter = NUM_MAX_ITERATIONS; cfg.sigma_num = 3.0;   // src/voxel_mapping.cpp:1365
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) cfg.extR[r * 3 + c] = m_extR(r, c); cfg.extT[r] = m_extT(r); }
    cfg.mesh_min_spacing = m_meshing_points_minimum_scale * m_meshing_distance_scale;   // ImMesh_node.cpp:254-270
    cfg.mesh_voxel = m_meshing_voxel_resolution * m_meshing_distance_scale;
    cfg.mesh_region = m_meshing_region_size * m_meshing_distance_scale;
    cfg.mesh_append_budget = m_meshing_number_of_pts_append_to_map;
    m_hip = immesh_create(&cfg);
    if (!m_hip) fail(nullptr, "immesh_create", IMMESH_E_NODEV);   // no CPU fallback
    g_immesh_ctx = m_hip;
}

// ---- bool Voxel_mapping::voxel_map_init()   src/voxel_mapping.cpp:1243 -------------------------------------------------------------------
bool Voxel_mapping::voxel_map_init() {
    const std::vector<float> xyz = pack_xyz(*m_feats_undistort);
    double st[IMMESH_STATE_DOUBLES];
    to_c(state, st);
    const int rc = immesh_map_build(m_hip, xyz.data(), (int64_t)m_feats_undistort->size(), st);
    if (rc) { fail(m_hip, "immesh_map_build", rc); return false; }
    return true;
}

// ---- void Voxel_mapping::lio_state_estimation(StatesGroup&)   src/voxel_mapping.cpp:1284 -------------------------------------------------
void Voxel_mapping::lio_state_estimation(StatesGroup& state_propagat) {
    const std::vector<float> down = pack_xyz(*m_feats_down_body);
    const int n = (int)m_feats_down_body->size();
    double prior[IMMESH_STATE_DOUBLES], st[IMMESH_STATE_DOUBLES];
    to_c(state_propagat, prior); to_c(state, st);
    std::vector<float> eff_pts((size_t)n * 3), eff_nd((size_t)n * 4);
    int iters = 0;
    const int rc = immesh_register(m_hip, down.data(), n, prior, st, &iters, &m_effct_feat_num, &m_res_mean_last, eff_pts.data(), eff_nd.data());
    if (rc) { fail(m_hip, "immesh_register", rc); return; }
    from_c(st, state);
    // m_laserCloudOri / m_corr_normvect of the last iteration: the publishers read them (src/voxel_mapping.cpp:1372-1392)
    m_laserCloudOri->resize(m_effct_feat_num); m_corr_normvect->resize(m_effct_feat_num);
    for (int i = 0; i < m_effct_feat_num; i++) {
        PointType& p = m_laserCloudOri->points[i]; PointType& q = m_corr_normvect->points[i];
        p.x = eff_pts[3 * i]; p.y = eff_pts[3 * i + 1]; p.z = eff_pts[3 * i + 2];
        q.x = eff_nd[4 * i]; q.y = eff_nd[4 * i + 1]; q.z = eff_nd[4 * i + 2]; q.intensity = eff_nd[4 * i + 3];
    }
}

// ---- void Voxel_mapping::map_incremental_grow()   src/ImMesh_mesh_reconstruction.cpp:377-424 --------------------------------------------
// voxel-map half: immesh_map_update; then, as the reference does (:413-417), the full scan goes to the world frame and to the mesher.  The
// reference queues it for service_reconstruct_mesh; here the (synchronous) incremental_mesh_reconstruction below is called directly -- the
// asynchronous variant is immesh_process_scan(.., IMMESH_MESH_ASYNC, ..) + immesh_mesh_wait.
void Voxel_mapping::map_incremental_grow() {
    const std::vector<float> down = pack_xyz(*m_feats_down_body);
    double st[IMMESH_STATE_DOUBLES];
    to_c(state, st);
    const int rc = immesh_map_update(m_hip, down.data(), (int)m_feats_down_body->size(), st);
    if (rc) { fail(m_hip, "immesh_map_update", rc); return; }
    // transformLidar(state.rot_end, state.pos_end, m_feats_undistort, world_lidar_full)   src/voxel_mapping_common.cpp:709-726: f64 compute, f32 store
    pcl::PointCloud<pcl::PointXYZI>::Ptr world(new pcl::PointCloud<pcl::PointXYZI>);
    world->resize(m_feats_undistort->size());
    for (size_t i = 0; i < m_feats_undistort->size(); i++) {
        const PointType& s = m_feats_undistort->points[i];
        const double p[3] = {s.x, s.y, s.z};
        double b[3], w[3];
        for (int r = 0; r < 3; r++) b[r] = m_extR(r, 0) * p[0] + m_extR(r, 1) * p[1] + m_extR(r, 2) * p[2] + m_extT(r);
        for (int r = 0; r < 3; r++) w[r] = state.rot_end(r, 0) * b[0] + state.rot_end(r, 1) * b[1] + state.rot_end(r, 2) * b[2] + state.pos_end(r);
        pcl::PointXYZI& d = world->points[i];
        d.x = (float)w[0]; d.y = (float)w[1]; d.z = (float)w[2]; d.intensity = s.intensity;
    }
    incremental_mesh_reconstruction(world, Eigen::Quaterniond(), state.pos_end, g_immesh_frame_idx++);
}

// ---- void incremental_mesh_reconstruction(cloud, q, t, frame_idx)   src/ImMesh_mesh_reconstruction.cpp:92-267 ------------------------------
void incremental_mesh_reconstruction(pcl::PointCloud<pcl::PointXYZI>::Ptr frame_pts, Eigen::Quaterniond, Eigen::Vector3d pose_t, int frame_idx) {
    immesh_ctx* c = g_immesh_ctx;
    const int n = (int)frame_pts->size();
    std::vector<float> xyzi((size_t)n * 4);   // pcl::PointXYZI is padded to 32 B; the ABI takes packed xyzI
    for (int i = 0; i < n; i++) { const pcl::PointXYZI& p = frame_pts->points[i]; xyzi[4 * i] = p.x; xyzi[4 * i + 1] = p.y; xyzi[4 * i + 2] = p.z; xyzi[4 * i + 3] = p.intensity; }
    const double cam[3] = {pose_t(0), pose_t(1), pose_t(2)};
    int rc = immesh_mesh_scan(c, xyzi.data(), n, cam, frame_idx);
    if (rc) { fail(c, "immesh_mesh_scan", rc); return; }
    // ---- host mirrors: Global_map::m_rgb_pts_vec (index == vertex id, pointcloud_rgbd.cpp:518-527) and the Triangle_manager
    immesh_mesh_sizes_t z;
    immesh_mesh_sizes(c, &z);
    std::vector<float> vtx((size_t)3 * z.n_new_vtx);
    std::vector<int32_t> add((size_t)3 * z.n_add), rem((size_t)3 * z.n_rem), upd((size_t)3 * z.n_upd), sid(z.n_smooth);
    std::vector<uint8_t> fadd(z.n_add), fupd(z.n_upd);
    std::vector<double> sxyz((size_t)3 * z.n_smooth);
    rc = immesh_mesh_fetch(c, vtx.data(), add.data(), fadd.data(), rem.data(), upd.data(), fupd.data(), sid.data(), sxyz.data());
    if (rc) { fail(c, "immesh_mesh_fetch", rc); return; }
    for (int i = 0; i < z.n_new_vtx; i++) {
        auto pt = std::make_shared<RGB_pts>();
        pt->set_pos(vec_3(vtx[3 * i], vtx[3 * i + 1], vtx[3 * i + 2]));
        pt->m_pt_index = (int)g_map_rgb_pts_mesh.m_rgb_pts_vec.size();
        g_map_rgb_pts_mesh.m_rgb_pts_vec.push_back(pt);
    }
    for (int i = 0; i < z.n_smooth; i++) g_map_rgb_pts_mesh.m_rgb_pts_vec[sid[i]]->set_smooth_pos(vec_3(sxyz[3 * i], sxyz[3 * i + 1], sxyz[3 * i + 2]));
    Triangle_set to_rem;   // all removes, then all adds (ImMesh_mesh_reconstruction.cpp:228-244)
    for (int i = 0; i < z.n_rem; i++) to_rem.insert(g_triangles_manager.find_triangle(rem[3 * i], rem[3 * i + 1], rem[3 * i + 2]));
    g_triangles_manager.remove_triangle_list(to_rem, frame_idx);
    for (int i = 0; i < z.n_add; i++) g_triangles_manager.insert_triangle(add[3 * i], add[3 * i + 1], add[3 * i + 2], 1, frame_idx)->m_index_flip = fadd[i];
    for (int i = 0; i < z.n_upd; i++) { Triangle_ptr t = g_triangles_manager.find_triangle(upd[3 * i], upd[3 * i + 1], upd[3 * i + 2]); if (t) t->m_index_flip = fupd[i]; }
}

// ---- void reconstruct_mesh_from_pointcloud(cloud, double)   src/ImMesh_mesh_reconstruction.cpp:328-345 -------------------------------------
void reconstruct_mesh_from_pointcloud(pcl::PointCloud<pcl::PointXYZI>::Ptr frame_pts, double minimum_pts_distance) {
    const int n = (int)frame_pts->size();
    std::vector<float> xyzi((size_t)n * 4);
    for (int i = 0; i < n; i++) { const pcl::PointXYZI& p = frame_pts->points[i]; xyzi[4 * i] = p.x; xyzi[4 * i + 1] = p.y; xyzi[4 * i + 2] = p.z; xyzi[4 * i + 3] = p.intensity; }
    const int rc = immesh_reconstruct_mesh_from_pointcloud(g_immesh_ctx, xyzi.data(), n, minimum_pts_distance);
    if (rc) fail(g_immesh_ctx, "immesh_reconstruct_mesh_from_pointcloud", rc);
}

// ---- vec_3 Global_map::smooth_pts( RGB_pt_ptr&, double smooth_factor, double knn, double maximum_smooth_dis )   src/meshing/r3live/pointcloud_rgbd.cpp:932-958 ----
// The renderer calls it for every triangle vertex the mesher has not smoothed (unparse_triangle_set_to_vector, src/meshing/mesh_rec_display.cpp:86-90).  The
// reference's body searches the HOST ikd-Tree (m_kdtree), which a drop-in never feeds -- zero neighbours, 0/0, a NaN vertex in the GL buffer; here the
// search runs on the device's map.  The value is stored in the point like the reference does (set_smooth_pos -> m_smoothed: asked once per vertex).
vec_3 Global_map::smooth_pts(RGB_pt_ptr& rgb_pt, double smooth_factor, double knn, double maximum_smooth_dis) {
    const int32_t id = rgb_pt->m_pt_index;
    double o[3] = {0, 0, 0};
    const int rc = immesh_smooth_pts(g_immesh_ctx, &id, 1, smooth_factor, (int32_t)knn, maximum_smooth_dis, o);
    if (rc) { fail(g_immesh_ctx, "immesh_smooth_pts", rc); return rgb_pt->get_pos(); }
    const vec_3 v(o[0], o[1], o[2]);
    rgb_pt->set_smooth_pos(v);
    return v;
}

// ---- void save_to_ply_file(std::string, double smooth_factor, double knn)   src/meshing/mesh_rec_geometry.cpp:71-131 -----------------------
void save_to_ply_file(std::string ply_file, double smooth_factor, double knn) {
    const int rc = immesh_save_ply(g_immesh_ctx, ply_file.c_str(), smooth_factor, (int32_t)knn);
    if (rc) fail(g_immesh_ctx, "immesh_save_ply", rc);
}
