// Test / bench driver of the ASYNCHRONOUS drop-in (immesh_shim_async.cpp), built with it into drop_in/libimmesh_dropin_async.so and loaded by
// tests/test_gpu_dropin.py and bench.py through ctypes.  It plays the two threads of the reference around the replaced bodies: the scan thread's per-scan
// section of service_LiDAR_update (src/voxel_mapping.cpp:1896-1904 first frame, :1959-1973 every other frame) on the caller's thread, and
// service_reconstruct_mesh (src/ImMesh_mesh_reconstruction.cpp:272-310) on a thread of its own, as ImMesh_node.cpp:274-281 starts it.
#include "immesh_ref_shapes.hpp"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include "immesh_c_api.h"

Global_map g_map_rgb_pts_mesh;            // ImMesh_node.cpp:108
Triangle_manager g_triangles_manager;     // :109
extern std::atomic<bool> g_immesh_service_stop;
extern std::atomic<long> g_immesh_frames_meshed;
extern void (*g_immesh_after_frame)(int frame_idx);
extern int g_frame_idx;
extern std::atomic<long long> g_immesh_shim_ns[5];

namespace {
struct FrameStat { int32_t nv, nl; unsigned long long hash; };
struct Driver {
    Voxel_mapping vm;
    std::thread service;
    bool record_hash = false;
    std::mutex mu;
    std::vector<FrameStat> stats;
};
Driver* g_drv = nullptr;
void load_state(const double* o, StatesGroup& s) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) s.rot_end(r, c) = o[r * 3 + c];
    for (int i = 0; i < 3; i++) { s.pos_end(i) = o[9 + i]; s.vel_end(i) = o[12 + i]; s.bias_g(i) = o[15 + i]; s.bias_a(i) = o[18 + i]; s.gravity(i) = o[21 + i]; }
    for (int r = 0; r < 18; r++) for (int c = 0; c < 18; c++) s.cov(r, c) = o[24 + r * 18 + c];
}
void store_state(const StatesGroup& s, double* o) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o[r * 3 + c] = s.rot_end(r, c);
    for (int i = 0; i < 3; i++) { o[9 + i] = s.pos_end(i); o[12 + i] = s.vel_end(i); o[15 + i] = s.bias_g(i); o[18 + i] = s.bias_a(i); o[21 + i] = s.gravity(i); }
    for (int r = 0; r < 18; r++) for (int c = 0; c < 18; c++) o[24 + r * 18 + c] = s.cov(r, c);
}
void after_frame(int frame_idx) {   // service thread, mirrors of `frame_idx` applied
    Driver* d = g_drv;
    if (!d) return;
    FrameStat st{(int32_t)g_map_rgb_pts_mesh.m_rgb_pts_vec.size(), (int32_t)immesh_mirror_live_count(g_triangles_manager), 0ull};
    if (d->record_hash) {
        unsigned long long h = 0;
        immesh_mirror_for_each_live(g_triangles_manager, [&](const Triangle_ptr& t) {
            unsigned long long x = ((unsigned long long)(unsigned)t->m_tri_pts_id[0] * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)(unsigned)t->m_tri_pts_id[1] << 21) ^ ((unsigned long long)(unsigned)t->m_tri_pts_id[2] << 42) ^ (unsigned long long)(t->m_index_flip & 1);
            x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
            h += x;
        });
        st.hash = h;
    }
    std::lock_guard<std::mutex> lk(d->mu);
    if ((int)d->stats.size() <= frame_idx) d->stats.resize(frame_idx + 1, FrameStat{0, 0, 0});
    d->stats[frame_idx] = st;
}
PointCloudXYZI::Ptr make_cloud(const float* p, int n, int stride) {
    PointCloudXYZI::Ptr c = std::make_shared<PointCloudXYZI>();
    c->resize(n);
    for (int i = 0; i < n; i++) { PointType& q = c->points[i]; q.x = p[(size_t)i * stride]; q.y = p[(size_t)i * stride + 1]; q.z = p[(size_t)i * stride + 2]; q.intensity = stride == 4 ? p[(size_t)i * 4 + 3] : 0.f; }
    return c;
}
}  // namespace

extern "C" {
// adopt != NULL: drive a context the caller created (it already holds a map); the driver then neither creates nor destroys one
void* dropin_create(immesh_ctx* adopt, const double* extT3, int record_hash) {
    if (g_drv) return nullptr;   // the reference's globals exist once per process
    Driver* d = new Driver();
    d->record_hash = record_hash != 0;
    for (int i = 0; i < 3; i++) d->vm.m_extT(i) = extT3[i];
    d->vm.m_hip = adopt;
    g_immesh_service_stop = false; g_immesh_frames_meshed = 0; g_frame_idx = 0;
    g_map_rgb_pts_mesh.m_rgb_pts_vec.clear();
    immesh_mirror_reset(g_triangles_manager, &g_map_rgb_pts_mesh, d->vm.m_meshing_region_size * d->vm.m_meshing_distance_scale);   // (ImMesh_node.cpp:268-271 in the node)
    d->vm.immesh_shim_init();
    g_drv = d;
    g_immesh_after_frame = after_frame;
    d->service = std::thread(service_reconstruct_mesh);   // ImMesh_node.cpp:277
    return d;
}
void dropin_destroy(void* p, int destroy_ctx) {
    Driver* d = (Driver*)p;
    if (!d) return;
    g_immesh_service_stop = true;
    if (d->service.joinable()) d->service.join();
    g_immesh_after_frame = nullptr;
    g_drv = nullptr;
    if (d->vm.m_hip) { (void)immesh_mesh_collect_enable(d->vm.m_hip, 0); if (destroy_ctx) immesh_destroy(d->vm.m_hip); }
    delete d;
}
// first frame (src/voxel_mapping.cpp:1896-1904): state = state0, the map is initialised from ALL raw points, the frame ends
int dropin_first_scan(void* p, const float* raw_xyzi, int n_raw, const double* state348) {
    Driver* d = (Driver*)p;
    d->vm.m_feats_undistort = make_cloud(raw_xyzi, n_raw, 4);
    load_state(state348, d->vm.state);
    return d->vm.voxel_map_init() ? 0 : -1;
}
// every other frame (:1959-1973); prior348 = state_propagat (the caller's Forward_without_imu / UndistortPcl), state = prior on entry as in the reference
int dropin_scan(void* p, const float* raw_xyzi, int n_raw, const float* down_xyz, int n_ds, const double* prior348, double* state_out348, int32_t* eff_out) {
    Driver* d = (Driver*)p;
    d->vm.m_feats_undistort = make_cloud(raw_xyzi, n_raw, 4);
    d->vm.m_feats_down_body = make_cloud(down_xyz, n_ds, 3);
    StatesGroup propagat;
    load_state(prior348, propagat);
    d->vm.state = propagat;
    d->vm.lio_state_estimation(propagat);
    d->vm.map_incremental_grow();
    store_state(d->vm.state, state_out348);
    if (eff_out) *eff_out = d->vm.m_effct_feat_num;
    return 0;
}
// A whole stream, timed the way an unchanged ImMesh_node.cpp would see it: the clouds exist as pcl clouds on the HOST when lio_state_estimation is called
// (they are built before the clock starts -- that is the upstream stages' work), the prior is Forward_without_imu of the previous posterior, every scan's
// result lists are fetched and applied to the Global_map / Triangle_manager mirrors by the service thread.  state_io: in = posterior of the scan before
// the first one, out = posterior of the last.  ms_out[0] = wall time until the last pose is back, ms_out[1] = until the mirrors hold the last frame.
int dropin_run_stream(void* p, int n_scans, const float* const* raw_xyzi, const int32_t* n_raw, const float* const* down_xyz, const int32_t* n_ds, double* state_io348,
                      double dt, double cov_gyr, double cov_acc, double* states_out, int32_t* eff_out, double* ms_out) {
    Driver* d = (Driver*)p;
    std::vector<PointCloudXYZI::Ptr> raws(n_scans), downs(n_scans);
    for (int k = 0; k < n_scans; k++) { raws[k] = make_cloud(raw_xyzi[k], n_raw[k], 4); downs[k] = make_cloud(down_xyz[k], n_ds[k], 3); }
    const long meshed0 = g_immesh_frames_meshed.load();
    double st[IMMESH_STATE_DOUBLES], prior[IMMESH_STATE_DOUBLES];
    std::memcpy(st, state_io348, sizeof(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n_scans; k++) {
        if (immesh_forward_without_imu(st, dt, cov_gyr, cov_acc, prior)) return -1;   // ImuProcess::Forward_without_imu, src/IMU_Processing.cpp:486-553
        d->vm.m_feats_undistort = raws[k];
        d->vm.m_feats_down_body = downs[k];
        StatesGroup propagat;
        load_state(prior, propagat);
        d->vm.state = propagat;
        d->vm.lio_state_estimation(propagat);
        d->vm.map_incremental_grow();
        store_state(d->vm.state, st);
        if (states_out) std::memcpy(states_out + (size_t)k * IMMESH_STATE_DOUBLES, st, sizeof(st));
        if (eff_out) eff_out[k] = d->vm.m_effct_feat_num;
    }
    const auto t1 = std::chrono::steady_clock::now();
    while (g_immesh_frames_meshed.load() < meshed0 + n_scans) {
        if (std::chrono::steady_clock::now() - t1 > std::chrono::seconds(60)) return -2;
        std::this_thread::yield();
    }
    const auto t2 = std::chrono::steady_clock::now();
    std::memcpy(state_io348, st, sizeof(st));
    if (ms_out) { ms_out[0] = std::chrono::duration<double, std::milli>(t1 - t0).count(); ms_out[1] = std::chrono::duration<double, std::milli>(t2 - t0).count(); }
    return 0;
}
// Harness only: bring the host mirrors up to a mesh map that was built on the device BEFORE the shim took over (bench.py pre-seeds the mesh map from a
// dense survey through immesh_mesh_scan without fetching 1 M triangles scan by scan).  In a deployment the mirrors follow the device from the first frame.
int dropin_seed_mirror(void* p) {
    Driver* d = (Driver*)p;
    int64_t nv = 0, nf = 0;
    int rc = immesh_mesh_export(d->vm.m_hip, 0.0, 20, &nv, &nf);   // raw positions, every live triangle with its winding
    if (rc) return rc;
    std::vector<float> v((size_t)nv * 3);
    std::vector<int32_t> f((size_t)nf * 3);
    if ((rc = immesh_mesh_export_fetch(d->vm.m_hip, v.data(), f.data()))) return rc;
    g_map_rgb_pts_mesh.m_rgb_pts_vec.clear();
    immesh_mirror_reset(g_triangles_manager, &g_map_rgb_pts_mesh, d->vm.m_meshing_region_size * d->vm.m_meshing_distance_scale);
    for (int64_t i = 0; i < nv; i++) {
        auto pt = std::make_shared<RGB_pts>();
        pt->set_pos(vec_3(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
        pt->m_pt_index = (int)i;
        g_map_rgb_pts_mesh.m_rgb_pts_vec.push_back(pt);
    }
    for (int64_t i = 0; i < nf; i++)   // faces are (v0, v1, v2) when m_index_flip != 0 else (v0, v2, v1), v0 < v1 < v2 (immesh_mesh_export)
        g_triangles_manager.insert_triangle(f[3 * i], f[3 * i + 1], f[3 * i + 2], 1, 0)->m_index_flip = f[3 * i + 1] < f[3 * i + 2] ? 1 : 0;
    return 0;
}
// host time per stage since the last call, milliseconds: pack, process_scan (scan thread); job wait, fetch, mirror update (service thread)
int dropin_stage_ms(void* p, double* ms5) {
    (void)p;
    for (int i = 0; i < 5; i++) ms5[i] = (double)g_immesh_shim_ns[i].exchange(0) * 1e-6;
    return 0;
}
int dropin_mirror_sizes(void* p, int64_t* nv, int64_t* nl) {
    (void)p;
    *nv = (int64_t)g_map_rgb_pts_mesh.m_rgb_pts_vec.size(); *nl = immesh_mirror_live_count(g_triangles_manager);
    return 0;
}
// What the renderer does with a region's Triangle_set (unparse_triangle_set_to_vector, src/meshing/mesh_rec_display.cpp:78-103), here over EVERY live triangle
// of the mirror: smooth what is still unsmoothed (Global_map::smooth_pts -- the shim's replaced body, one vertex per call, as the unchanged renderer would),
// then three float positions (get_pos(1)) per triangle.  The same buffer is asked of the library in ONE call (immesh_mesh_display_vertices).
// out[0] triangles, out[1] vertices smoothed by this pass, out[2] NaN coordinates in the buffer, out[3] coordinates that differ from the batched call's
int dropin_render_pass(void* p, double smooth_factor, double knn, double max_dis, int64_t* out4, float* buffer, int64_t cap_floats) {
    Driver* d = (Driver*)p;
    std::vector<int32_t> ids;
    immesh_mirror_for_each_live(g_triangles_manager, [&](const Triangle_ptr& t) { for (int k = 0; k < 3; k++) ids.push_back(t->m_tri_pts_id[k]); });
    std::vector<float> batched(ids.size() * 3, 0.f);
    // (the batched entry first: it reads the device's own smoothed positions, the per-vertex pass below then writes the mirror's)
    if (!ids.empty() && immesh_mesh_display_vertices(d->vm.m_hip, ids.data(), (int32_t)ids.size(), smooth_factor, (int32_t)knn, max_dis, batched.data()) != 0) return -1;
    int64_t n_now = 0, n_nan = 0, n_diff = 0;
    std::vector<float> buf(ids.size() * 3);
    for (size_t i = 0; i < ids.size(); i++) {
        RGB_pt_ptr& pt = g_map_rgb_pts_mesh.m_rgb_pts_vec[ids[i]];
        if (pt->m_smoothed == false) { g_map_rgb_pts_mesh.smooth_pts(pt, smooth_factor, knn, max_dis); n_now++; }
        const vec_3 v = pt->get_pos(1);
        for (int a = 0; a < 3; a++) {
            const float f = (float)v(a);
            buf[3 * i + a] = f;
            if (f != f) n_nan++;
            else if (f != batched[3 * i + a]) n_diff++;
        }
    }
    out4[0] = (int64_t)ids.size() / 3; out4[1] = n_now; out4[2] = n_nan; out4[3] = n_diff;
    if (buffer) std::memcpy(buffer, buf.data(), std::min<size_t>(buf.size(), (size_t)cap_floats) * 4);
    return 0;
}
int dropin_wait_meshed(void* p, long n_frames, int timeout_ms) {
    (void)p;
    const auto t0 = std::chrono::steady_clock::now();
    while (g_immesh_frames_meshed.load() < n_frames) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) return -1;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return 0;
}
int dropin_frame_stats(void* p, int frame, int32_t* nv, int32_t* nl, unsigned long long* hash) {
    Driver* d = (Driver*)p;
    std::lock_guard<std::mutex> lk(d->mu);
    if (frame < 0 || frame >= (int)d->stats.size()) return -1;
    *nv = d->stats[frame].nv; *nl = d->stats[frame].nl; *hash = d->stats[frame].hash;
    return 0;
}
int dropin_effect_features(void* p, float* eff_pts, float* eff_nd, int cap) {   // publish_effect_world's inputs, on demand
    Driver* d = (Driver*)p;
    d->vm.immesh_fetch_effect_features();
    const int n = (int)d->vm.m_laserCloudOri->size();
    for (int i = 0; i < n && i < cap; i++) {
        const PointType& a = d->vm.m_laserCloudOri->points[i]; const PointType& b = d->vm.m_corr_normvect->points[i];
        eff_pts[3 * i] = a.x; eff_pts[3 * i + 1] = a.y; eff_pts[3 * i + 2] = a.z;
        eff_nd[4 * i] = b.x; eff_nd[4 * i + 1] = b.y; eff_nd[4 * i + 2] = b.z; eff_nd[4 * i + 3] = b.intensity;
    }
    return n;
}
}  // extern "C"
