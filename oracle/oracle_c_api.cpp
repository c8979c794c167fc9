// ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
// C entry points mirroring include/immesh_c_api.h one-to-one (prefix orc_ instead of immesh_), so parity tests
// drive the CPU restatement and the HIP library with the same calls.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load liboracle.so.
#include "../include/immesh_c_api.h"
#include "orc_mesher.hpp"
#include "orc_imu.hpp"
#include "orc_ikdmap.hpp"
#include <string>
#include <chrono>
#include <climits>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

struct OrcCtx {
    std::vector<float> exp_vtx;
    std::vector<int> exp_faces;
    Config cfg;
    VoxelMap vm;
    Registration reg;
    Mesher mesher;
    MeshScanOut mout;
    orc::IkdMap ikd;
    std::vector<float> last_eff_pts, last_eff_nd;   // m_laserCloudOri / m_corr_normvect of the last registration (orc_last_matches)
    std::vector<float> last_world;   // the world-frame scan the newest mesh job was handed (orc_mesh_world_scan)
    OrcCtx() : reg(&vm) {}
};

static void load_state(const double* s, State& st) {
    std::memcpy(st.R, s, 9 * 8); std::memcpy(st.t, s + 9, 24); std::memcpy(st.vel, s + 12, 24); std::memcpy(st.bg, s + 15, 24);
    std::memcpy(st.ba, s + 18, 24); std::memcpy(st.g, s + 21, 24); std::memcpy(st.cov, s + 24, 324 * 8);
}
static void store_state(const State& st, double* s) {
    std::memcpy(s, st.R, 9 * 8); std::memcpy(s + 9, st.t, 24); std::memcpy(s + 12, st.vel, 24); std::memcpy(s + 15, st.bg, 24);
    std::memcpy(s + 18, st.ba, 24); std::memcpy(s + 21, st.g, 24); std::memcpy(s + 24, st.cov, 324 * 8);
}

extern "C" {

void* orc_create(const immesh_config* c) {
    OrcCtx* o = new OrcCtx();
    Config& g = o->cfg;
    g.voxel_size = c->voxel_size; g.max_layer = c->max_layer;
    for (int i = 0; i < 5; i++) g.layer_init[i] = c->layer_init[i];
    g.max_points_size = c->max_points_size; g.planer_threshold = c->planer_threshold;
    g.dept_err = c->dept_err; g.beam_err = c->beam_err; g.calib_laser = c->calib_laser; g.sigma_num = c->sigma_num; g.max_iter = c->max_iter;
    std::memcpy(g.extR, c->extR, sizeof(g.extR)); std::memcpy(g.extT, c->extT, sizeof(g.extT));
    g.mesh_min_spacing = c->mesh_min_spacing; g.mesh_voxel = c->mesh_voxel; g.mesh_region = c->mesh_region; g.mesh_append_budget = c->mesh_append_budget;
    o->vm.cfg = g;
    o->mesher.cfg = g;
    o->mesher.cnt = &o->vm.cnt;
    return o;
}
void orc_destroy(void* p) { delete (OrcCtx*)p; }

int orc_map_build(void* p, const float* pts, int64_t n, const double* state) {
    OrcCtx* o = (OrcCtx*)p;
    State s; load_state(state, s);
    o->reg.map_init(pts, (int)n, s);
    return 0;
}

int orc_register(void* p, const float* pts, int32_t n_ds, const double* state_prior, double* state_inout, int32_t* n_iter_out,
                 int32_t* n_match_out, double* res_mean_out, float* eff_pts_body, float* eff_norm_dis) {
    OrcCtx* o = (OrcCtx*)p;
    State prior, st;
    load_state(state_prior, prior); load_state(state_inout, st);
    RegDebug dbg;
    const int it = o->reg.run(pts, n_ds, prior, st, &dbg);
    store_state(st, state_inout);
    if (n_iter_out) *n_iter_out = it;
    const int M = dbg.n_match.empty() ? 0 : dbg.n_match.back();
    if (n_match_out) *n_match_out = M;
    if (res_mean_out) *res_mean_out = dbg.res_mean_last;
    o->last_eff_pts.resize((size_t)M * 3); o->last_eff_nd.resize((size_t)M * 4);
    for (int i = 0; i < M; i++) {
        const int j = dbg.match_idx_last[i];
        for (int k = 0; k < 3; k++) o->last_eff_pts[i * 3 + k] = pts[j * 3 + k];
        for (int k = 0; k < 3; k++) o->last_eff_nd[i * 4 + k] = (float)dbg.normals_last[i * 3 + k];
        o->last_eff_nd[i * 4 + 3] = dbg.dis_last[i];
    }
    if (eff_pts_body && M) std::memcpy(eff_pts_body, o->last_eff_pts.data(), (size_t)M * 12);
    if (eff_norm_dis && M) std::memcpy(eff_norm_dis, o->last_eff_nd.data(), (size_t)M * 16);
    return 0;
}
// orc_register with what every iteration held when it ended (RegDebug): HTH 36, HTz 6, solution 18, G 324, state 24, covariance 324, matches,
// mean residual, {converged, stopped} per iteration, at most `cap_iters` of them; the test tap of tests/test_ref_lio.py
int orc_register_trace(void* p, const float* pts, int32_t n_ds, const double* state_prior, double* state_inout, int32_t cap_iters, double* HTH, double* HTz,
                       double* sol, double* G, double* state24, double* cov, int32_t* n_match, double* res_mean, int32_t* flags2, float* eff_pts_body,
                       float* eff_norm_dis, double* rinv_last) {
    OrcCtx* o = (OrcCtx*)p;
    State prior, st;
    load_state(state_prior, prior); load_state(state_inout, st);
    RegDebug dbg;
    const int it = o->reg.run(pts, n_ds, prior, st, &dbg);
    store_state(st, state_inout);
    for (int k = 0; k < it && k < cap_iters; k++) {
        if (HTH) std::memcpy(HTH + k * 36, &dbg.HTH[(size_t)k * 36], 36 * 8);
        if (HTz) std::memcpy(HTz + k * 6, &dbg.HTz[(size_t)k * 6], 6 * 8);
        if (sol) std::memcpy(sol + k * 18, &dbg.sol[(size_t)k * 18], 18 * 8);
        if (G) std::memcpy(G + k * 324, &dbg.G[(size_t)k * 324], 324 * 8);
        if (state24) std::memcpy(state24 + k * 24, &dbg.state24[(size_t)k * 24], 24 * 8);
        if (cov) std::memcpy(cov + k * 324, &dbg.cov[(size_t)k * 324], 324 * 8);
        if (n_match) n_match[k] = dbg.n_match[k];
        if (res_mean) res_mean[k] = dbg.res_mean[k];
        if (flags2) { flags2[k * 2] = dbg.converged[k]; flags2[k * 2 + 1] = dbg.stopped[k]; }
    }
    const int M = dbg.n_match.empty() ? 0 : dbg.n_match.back();
    for (int i = 0; i < M; i++) {
        const int j = dbg.match_idx_last[i];
        if (eff_pts_body) for (int k = 0; k < 3; k++) eff_pts_body[i * 3 + k] = pts[j * 3 + k];
        if (eff_norm_dis) { for (int k = 0; k < 3; k++) eff_norm_dis[i * 4 + k] = (float)dbg.normals_last[i * 3 + k]; eff_norm_dis[i * 4 + 3] = dbg.dis_last[i]; }
        if (rinv_last) rinv_last[i] = dbg.rinv_last[i];
    }
    return it;
}
int orc_last_matches(void* p, float* eff_pts_body, float* eff_norm_dis, int32_t cap, int32_t* n_out) {
    OrcCtx* o = (OrcCtx*)p;
    const int32_t M = (int32_t)(o->last_eff_pts.size() / 3);
    *n_out = M;
    if (!eff_pts_body && !eff_norm_dis) return 0;
    if (M > cap) return -4;
    if (eff_pts_body && M) std::memcpy(eff_pts_body, o->last_eff_pts.data(), (size_t)M * 12);
    if (eff_norm_dis && M) std::memcpy(eff_norm_dis, o->last_eff_nd.data(), (size_t)M * 16);
    return 0;
}

// one matcher + H-build pass at a fixed state: run() with max_iter forced to 1 on a scratch copy of the state
int orc_residuals(void* p, const float* pts, int32_t n_ds, const double* state, double* HTH36, double* HTz6, int32_t* n_match,
                  int32_t* match_idx, double* normals, float* dis, double* r_inv) {
    OrcCtx* o = (OrcCtx*)p;
    State st; load_state(state, st);
    State prior = st;
    const int keep = o->vm.cfg.max_iter;
    o->vm.cfg.max_iter = 1;
    RegDebug dbg;
    o->reg.run(pts, n_ds, prior, st, &dbg);
    o->vm.cfg.max_iter = keep;
    std::memcpy(HTH36, dbg.HTH.data(), 36 * 8);
    std::memcpy(HTz6, dbg.HTz.data(), 6 * 8);
    const int M = dbg.n_match[0];
    if (n_match) *n_match = M;
    for (int i = 0; i < M; i++) {
        if (match_idx) match_idx[i] = dbg.match_idx_last[i];
        if (normals) for (int k = 0; k < 3; k++) normals[i * 3 + k] = dbg.normals_last[i * 3 + k];
        if (dis) dis[i] = dbg.dis_last[i];
        if (r_inv) r_inv[i] = dbg.rinv_last[i];
    }
    return 0;
}

int orc_map_update(void* p, const float* pts, int32_t n_ds, const double* state) {
    OrcCtx* o = (OrcCtx*)p;
    State s; load_state(state, s);
    o->reg.prepare(pts, n_ds);
    o->reg.map_grow(pts, n_ds, s);
    return 0;
}

int orc_mesh_scan(void* p, const float* pts_world_xyzi, int32_t n_raw, const double* sensor_pos, int32_t frame_idx) {
    (void)frame_idx;
    OrcCtx* o = (OrcCtx*)p;
    o->last_world.assign(pts_world_xyzi, pts_world_xyzi + (size_t)n_raw * 4);
    o->mesher.mesh_scan(pts_world_xyzi, n_raw, sensor_pos, o->mout);
    return 0;
}
int orc_mesh_world_scan(void* p, float* out_xyzi, int32_t cap_pts, int32_t* n_out) {
    OrcCtx* o = (OrcCtx*)p;
    const int32_t n = (int32_t)(o->last_world.size() / 4);
    *n_out = n;
    if (out_xyzi) { if (n > cap_pts) return -4; std::memcpy(out_xyzi, o->last_world.data(), o->last_world.size() * 4); }
    return 0;
}
int orc_forward_without_imu(const double*, double, double, double, double*) { return -1; }  // harness-side prior lives in synth.py for the checker
// pcl::VoxelGrid stand-in (SURVEY A.15 spec, the harness's voxel_grid_downsample): float32 arithmetic, stable order, sequential centroid sums
int orc_downsample(void* p, const float* pts, int32_t n, int32_t stride, double leaf, float* out_xyz, int32_t cap_out, int32_t* n_out) {
    (void)p;
    const float inv = (float)(1.0 / leaf);
    long mn[3] = {LONG_MAX, LONG_MAX, LONG_MAX}, mx[3] = {LONG_MIN, LONG_MIN, LONG_MIN};
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { const long c = (long)std::floor(pts[(size_t)i * stride + a] * inv); mn[a] = std::min(mn[a], c); mx[a] = std::max(mx[a], c); }
    const long dx = mx[0] - mn[0] + 1, dy = mx[1] - mn[1] + 1;
    std::vector<std::pair<unsigned long long, int>> kv(n);
    for (int i = 0; i < n; i++) {
        const long ix = (long)std::floor(pts[(size_t)i * stride + 0] * inv) - mn[0], iy = (long)std::floor(pts[(size_t)i * stride + 1] * inv) - mn[1],
                   iz = (long)std::floor(pts[(size_t)i * stride + 2] * inv) - mn[2];
        kv[i] = {(unsigned long long)(ix + iy * dx + iz * dx * dy), i};
    }
    std::stable_sort(kv.begin(), kv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    int cnt = 0;
    for (int i = 0; i < n;) {
        float sx = 0.f, sy = 0.f, sz = 0.f; int c = 0, j = i;
        for (; j < n && kv[j].first == kv[i].first; j++) { const int q = kv[j].second; sx += pts[(size_t)q * stride]; sy += pts[(size_t)q * stride + 1]; sz += pts[(size_t)q * stride + 2]; c++; }
        if (out_xyz && cnt < cap_out) { out_xyz[cnt * 3] = sx / (float)c; out_xyz[cnt * 3 + 1] = sy / (float)c; out_xyz[cnt * 3 + 2] = sz / (float)c; }
        cnt++; i = j;
    }
    *n_out = cnt;
    return (out_xyz && cnt > cap_out) ? IMMESH_E_CAPACITY : 0;
}
const float* orc_downsample_result(void*) { return nullptr; }
// the asynchronous pair of the product (immesh_downsample_begin / _end) is the same function run ahead of time: the checker runs it on the spot
static std::vector<float> g_ds_async;
static int32_t g_ds_async_n = 0;
int orc_downsample_begin(void* p, const float* pts, int32_t n, int32_t stride, double leaf) {
    g_ds_async.assign((size_t)n * 3, 0.f);
    return orc_downsample(p, pts, n, stride, leaf, g_ds_async.data(), n, &g_ds_async_n);
}
int orc_downsample_end(void*, int32_t* n_out, const float** xyz) { *n_out = g_ds_async_n; if (xyz) *xyz = g_ds_async.data(); return 0; }
int orc_mesh_collect_enable(void*, int32_t) { return 0; }   // (the checker is synchronous: the job of a call has finished when the call returns)
int orc_mesh_collect_begin(void*, int32_t, int64_t* ord) { if (ord) *ord = 0; return 0; }
int orc_mesh_collect_end(void*) { return 0; }
int orc_registration_fallbacks(void*, int64_t* n) { if (n) *n = 0; return 0; }
int orc_inputs_consumed(void*) { return 0; }   // (the checker is synchronous: a call has consumed its inputs when it returns)
int orc_mesh_scan(void* p, const float* pts_world_xyzi, int32_t n_raw, const double* sensor_pos, int32_t frame_idx);
int orc_reconstruct_mesh_from_pointcloud(void* p, const float* pts_xyzi, int32_t n, double leaf) {   // ImMesh_mesh_reconstruction.cpp:328-345
    std::vector<float> ds((size_t)n * 3);
    int32_t n_ds = 0;
    orc_downsample(p, pts_xyzi, n, 4, leaf, ds.data(), n, &n_ds);
    std::vector<float> w((size_t)n_ds * 4, 0.f);
    for (int i = 0; i < n_ds; i++) for (int a = 0; a < 3; a++) w[(size_t)i * 4 + a] = ds[(size_t)i * 3 + a];
    const double origin[3] = {0, 0, 0};
    return orc_mesh_scan(p, w.data(), n_ds, origin, 0);
}
int orc_set_allreduce(void*, immesh_allreduce_fn, void*) { return 0; }
int orc_stub_collectives(void*) { return 0; }
int orc_device_bytes(void*, int64_t* b) { if (b) *b = 0; return 0; }   // the checker is single-process: nothing to reduce
// ---- legacy registration path (a27)
int orc_ikd_build(void* p, const float* xyz, int32_t n, double ds) { OrcCtx* o = (OrcCtx*)p; o->ikd.ds = (float)ds; o->ikd.build(xyz, n); return 0; }
int orc_ikd_add_points(void* p, const float* xyz, int32_t n) { ((OrcCtx*)p)->ikd.add_points(xyz, n); return 0; }
int orc_ikd_delete_boxes(void* p, const float* boxes, int32_t nb, int32_t* n_deleted) {
    const int r = ((OrcCtx*)p)->ikd.delete_boxes(boxes, nb);
    if (n_deleted) *n_deleted = r;
    return 0;
}
int orc_ikd_fov_segment(void* p, const double* pos_lid, double cube_len, double detection_range, int32_t* n_deleted) {
    const int r = ((OrcCtx*)p)->ikd.fov_segment(pos_lid, cube_len, (float)detection_range);
    if (n_deleted) *n_deleted = r;
    return 0;
}
int orc_ikd_size(void* p, int64_t* n) { *n = (int64_t)((OrcCtx*)p)->ikd.count; return 0; }
int orc_ikd_dump(void* p, float* xyz, int64_t cap, int64_t* n_out) {
    std::vector<float> all;
    ((OrcCtx*)p)->ikd.dump(all);
    *n_out = (int64_t)all.size() / 3;
    if (xyz) std::memcpy(xyz, all.data(), (size_t)std::min<int64_t>(cap, *n_out) * 12);
    return 0;
}
int orc_ikd_knn(void* p, const float* q, int32_t nq, float* nn_xyz, float* d2, int32_t* n_found) {
    OrcCtx* o = (OrcCtx*)p;
    for (int i = 0; i < nq; i++) {
        orc::IkdPt pt[5]; float dd[5];
        const int k = o->ikd.knn(q[i * 3], q[i * 3 + 1], q[i * 3 + 2], 5, pt, dd);
        for (int j = 0; j < 5; j++) {
            if (nn_xyz) { nn_xyz[(i * 5 + j) * 3] = j < k ? pt[j].x : 0; nn_xyz[(i * 5 + j) * 3 + 1] = j < k ? pt[j].y : 0; nn_xyz[(i * 5 + j) * 3 + 2] = j < k ? pt[j].z : 0; }
            if (d2) d2[i * 5 + j] = j < k ? dd[j] : 0;
        }
        if (n_found) n_found[i] = k;
    }
    return 0;
}
int orc_ikd_register(void* p, const float* body, int32_t n, const double* state_prior, double* state_inout, double laser_point_cov, int32_t* n_iter,
                     int32_t* n_match, double* res_mean, int32_t* match_idx, float* normals_pd2) {
    OrcCtx* o = (OrcCtx*)p;
    orc::State prior, st;
    load_state(state_prior, prior); load_state(state_inout, st);
    orc::IkdRegResult r;
    orc::ikd_register(o->ikd, o->vm.cfg, body, n, prior, st, laser_point_cov, r);
    store_state(st, state_inout);
    if (n_iter) *n_iter = r.n_iter;
    if (n_match) *n_match = r.n_match;
    if (res_mean) *res_mean = r.res_mean;
    if (match_idx) std::memcpy(match_idx, r.match_idx.data(), r.match_idx.size() * sizeof(int));
    if (normals_pd2) std::memcpy(normals_pd2, r.normals_pd2.data(), r.normals_pd2.size() * sizeof(float));
    return 0;
}
int orc_decode_livox(void*, const uint8_t* wire, int32_t n, int32_t n_scans, int32_t point_filter_num, double blind, float* out, int32_t* n_out) {
    std::vector<float> o;
    const int k = orc::decode_livox(wire, n, n_scans, point_filter_num, blind, o);
    if (out) std::memcpy(out, o.data(), o.size() * sizeof(float));
    if (n_out) *n_out = k;
    return 0;
}
int orc_decode_velodyne(void*, const uint8_t* data, int32_t n, int32_t step, int32_t ox, int32_t oy, int32_t oz, int32_t oi, int32_t n_scans, float* out, int32_t* n_out) {
    std::vector<float> o;
    const int k = orc::decode_velodyne(data, n, step, ox, oy, oz, oi, n_scans, o);
    if (out) std::memcpy(out, o.data(), o.size() * sizeof(float));
    if (n_out) *n_out = k;
    return 0;
}
const float* orc_decode_result(void*) { return nullptr; }
int orc_undistort(void* p, const float* pts_xyzit, int32_t n, const immesh_imu_sample* imu, int32_t n_imu, double lidar_beg_time, double* last_update_time,
                  immesh_imu_ctx* ic, double* state_inout, float* out_xyzi) {   // ImuProcess::UndistortPcl, IMU_Processing.cpp:755-958
    (void)p;
    std::vector<float> out;
    orc::undistort_pcl(pts_xyzit, n, imu, n_imu, lidar_beg_time, last_update_time, ic, state_inout, out);
    if (out_xyzi) std::memcpy(out_xyzi, out.data(), out.size() * sizeof(float));
    return 0;
}
const float* orc_undistort_result(void*) { return nullptr; }
// checker only: worker threads of the voxel-parallel mesher part / the matcher loop (the reference: 12-thread TBB pool, MP_PROC_NUM = 4 OpenMP threads)
int orc_set_threads(void* p, int32_t mesher_threads, int32_t matcher_threads) {
    OrcCtx* o = (OrcCtx*)p;
    o->mesher.threads = mesher_threads > 0 ? mesher_threads : 1;
    o->vm.threads = matcher_threads > 0 ? matcher_threads : 1;
    return 0;
}
int orc_set_allgather(void*, immesh_allgather_fn, void*) { return 0; }
int orc_broadcast_scan(void*, const float* pts, int32_t n, int32_t, int32_t, const float** out, int32_t* n_out) { if (out) *out = pts; if (n_out) *n_out = n; return 0; }   // the checker is single-process: the scan is where it is
int orc_shard_traffic(void*, int64_t* bytes, int64_t* calls) { if (bytes) *bytes = 0; if (calls) *calls = 0; return 0; }
int orc_shard_owner(const immesh_config*, const int64_t*) { return 0; }
int orc_mesh_wait(void* p) { (void)p; return 0; }
int orc_mesh_export(void* p, double smooth_factor, int32_t knn, int64_t* n_vtx, int64_t* n_faces) {
    OrcCtx* o = (OrcCtx*)p;
    o->mesher.export_mesh(smooth_factor, knn, o->exp_vtx, o->exp_faces);
    if (n_vtx) *n_vtx = (int64_t)o->exp_vtx.size() / 3;
    if (n_faces) *n_faces = (int64_t)o->exp_faces.size() / 3;
    return 0;
}
int orc_mesh_export_fetch(void* p, float* vtx, int32_t* faces) {
    OrcCtx* o = (OrcCtx*)p;
    if (vtx) std::memcpy(vtx, o->exp_vtx.data(), o->exp_vtx.size() * 4);
    if (faces) std::memcpy(faces, o->exp_faces.data(), o->exp_faces.size() * 4);
    return 0;
}
int orc_smooth_pts(void* p, const int32_t* ids, int32_t n, double smooth_factor, int32_t knn, double max_dis, double* out_xyz) {
    OrcCtx* o = (OrcCtx*)p;
    for (int i = 0; i < n; i++) {
        if (ids[i] < 0 || ids[i] >= (int)o->mesher.verts.size()) return IMMESH_E_INVAL;
        o->mesher.smooth_pts(ids[i], smooth_factor, knn, max_dis, out_xyz + (size_t)i * 3);
    }
    return 0;
}
int orc_mesh_display_vertices(void* p, const int32_t* ids, int32_t n, double smooth_factor, int32_t knn, double max_dis, float* out_xyz) {
    OrcCtx* o = (OrcCtx*)p;
    for (int i = 0; i < n; i++) {
        if (ids[i] < 0 || ids[i] >= (int)o->mesher.verts.size()) return IMMESH_E_INVAL;
        o->mesher.display_vertex(ids[i], smooth_factor, knn, max_dis, out_xyz + (size_t)i * 3);
    }
    return 0;
}
int orc_save_ply(void*, const char*, double, int32_t) { return -1; }   // file output is a product feature; the checker compares the arrays  // the checker is synchronous
int orc_mesh_sizes(void* p, immesh_mesh_sizes_t* s) {
    OrcCtx* o = (OrcCtx*)p;
    const MeshScanOut& m = o->mout;
    s->vtx_base = m.vtx_base; s->n_new_vtx = (int)m.new_vtx.size() / 3; s->n_add = (int)m.tri_add.size() / 3; s->n_rem = (int)m.tri_rem.size() / 3;
    s->n_upd = (int)m.tri_upd.size() / 3; s->n_smooth = (int)m.smooth_ids.size(); s->n_voxels_meshed = m.v_act; s->reserved = 0;
    return 0;
}
int orc_mesh_neighbourhood_sizes(void* p, int32_t* out, int32_t cap, int32_t* n_out) {
    OrcCtx* o = (OrcCtx*)p;
    const MeshScanOut& m = o->mout;
    *n_out = (int32_t)m.n_u_list.size();
    if (out) { if ((int32_t)m.n_u_list.size() > cap) return -4; std::memcpy(out, m.n_u_list.data(), m.n_u_list.size() * 4); }
    return 0;
}
int orc_mesh_fetch(void* p, float* new_vtx_xyz, int32_t* tri_add, uint8_t* flip_add, int32_t* tri_rem, int32_t* tri_upd, uint8_t* flip_upd,
                   int32_t* smooth_ids, double* smooth_xyz) {
    OrcCtx* o = (OrcCtx*)p;
    const MeshScanOut& m = o->mout;
    if (new_vtx_xyz) std::memcpy(new_vtx_xyz, m.new_vtx.data(), m.new_vtx.size() * 4);
    if (tri_add) std::memcpy(tri_add, m.tri_add.data(), m.tri_add.size() * 4);
    if (flip_add) std::memcpy(flip_add, m.flip_add.data(), m.flip_add.size());
    if (tri_rem) std::memcpy(tri_rem, m.tri_rem.data(), m.tri_rem.size() * 4);
    if (tri_upd) std::memcpy(tri_upd, m.tri_upd.data(), m.tri_upd.size() * 4);
    if (flip_upd) std::memcpy(flip_upd, m.flip_upd.data(), m.flip_upd.size());
    if (smooth_ids) std::memcpy(smooth_ids, m.smooth_ids.data(), m.smooth_ids.size() * 4);
    if (smooth_xyz) std::memcpy(smooth_xyz, m.smooth_xyz.data(), m.smooth_xyz.size() * 8);
    return 0;
}

// body->world of the full scan (transformLidar, voxel_mapping_common.cpp:709-726): f64 compute, f32 store, intensity kept
static void transform_full(const Config& c, const State& s, const float* in_xyzi, int n, std::vector<float>& out) {
    out.resize((size_t)n * 4);
    for (int i = 0; i < n; i++) {
        const double p[3] = {in_xyzi[i * 4 + 0], in_xyzi[i * 4 + 1], in_xyzi[i * 4 + 2]};
        double pw[3];
        body_to_world_d(c, s.R, s.t, p, pw);
        out[i * 4 + 0] = (float)pw[0]; out[i * 4 + 1] = (float)pw[1]; out[i * 4 + 2] = (float)pw[2]; out[i * 4 + 3] = in_xyzi[i * 4 + 3];
    }
}

static thread_local float g_timing[4] = {0, 0, 0, 0};
int orc_process_scan(void* p, const float* pts_down, int32_t n_ds, const float* pts_raw_xyzi, int32_t n_raw, const double* state_prior,
                     double* state_inout, int32_t frame_idx, int32_t do_mesh, int32_t* n_iter_out, int32_t* n_match_out) {
    OrcCtx* o = (OrcCtx*)p;
    auto t0 = std::chrono::steady_clock::now();
    int rc = orc_register(p, pts_down, n_ds, state_prior, state_inout, n_iter_out, n_match_out, nullptr, nullptr, nullptr);
    if (rc) return rc;
    auto t1 = std::chrono::steady_clock::now();
    State s; load_state(state_inout, s);
    o->reg.map_grow(pts_down, n_ds, s);
    auto t2 = std::chrono::steady_clock::now();
    if (do_mesh & 3) {   // bit 4 (IMMESH_SCAN_NOWAIT) has no meaning for the synchronous checker
        std::vector<float>& world = o->last_world;
        transform_full(o->cfg, s, pts_raw_xyzi, n_raw, world);
        o->mesher.mesh_scan(world.data(), n_raw, s.t, o->mout);
    }
    auto t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return (float)std::chrono::duration<double, std::milli>(b - a).count(); };
    g_timing[0] = ms(t0, t3); g_timing[1] = ms(t0, t1); g_timing[2] = ms(t1, t2); g_timing[3] = ms(t2, t3);
    (void)frame_idx;
    return 0;
}
// the strided form of the same call (pcl-shaped clouds consumed in place): the checker unpacks and calls the packed form
int orc_process_scan_strided(void* p, const void* pts_down, int32_t n_ds, int32_t down_stride, const void* pts_raw, int32_t n_raw, int32_t raw_stride, int32_t raw_int_off,
                             const double* state_prior, double* state_inout, int32_t frame_idx, int32_t do_mesh, int32_t* n_iter_out, int32_t* n_match_out) {
    std::vector<float> down((size_t)n_ds * 3), raw(do_mesh ? (size_t)n_raw * 4 : 0);
    for (int i = 0; i < n_ds; i++) std::memcpy(&down[(size_t)i * 3], (const char*)pts_down + (size_t)i * down_stride, 12);
    if (do_mesh) for (int i = 0; i < n_raw; i++) { std::memcpy(&raw[(size_t)i * 4], (const char*)pts_raw + (size_t)i * raw_stride, 12); std::memcpy(&raw[(size_t)i * 4 + 3], (const char*)pts_raw + (size_t)i * raw_stride + raw_int_off, 4); }
    return orc_process_scan(p, down.data(), n_ds, do_mesh ? raw.data() : nullptr, n_raw, state_prior, state_inout, frame_idx, do_mesh, n_iter_out, n_match_out);
}
int orc_last_timing(void* p, float ms[4]) { (void)p; for (int i = 0; i < 4; i++) ms[i] = g_timing[i]; return 0; }

static void dump_node(const Key& k, const OctoTree* n, int path, int depth, immesh_plane_rec* out, int64_t cap, int64_t& cnt) {
    if (n->init_octo) {
        if (cnt < cap && out) {
            immesh_plane_rec& r = out[cnt];
            r.key[0] = k.x; r.key[1] = k.y; r.key[2] = k.z;
            r.layer = n->layer; r.path = path; r.is_plane = n->plane.is_plane ? 1 : 0; r.n_points = (int)n->temp_points.size();
            r.update_enable = n->update_enable ? 1 : 0; r.new_points = n->new_points;
            r.radius = n->plane.radius; r.min_eig = n->plane.min_eig; r.d = n->plane.d; r.pad = 0;
            for (int i = 0; i < 3; i++) { r.center[i] = n->plane.center[i]; r.normal[i] = n->plane.normal[i]; }
            std::memcpy(r.plane_var, n->plane.plane_var, sizeof(r.plane_var));
        }
        cnt++;
    }
    for (int l = 0; l < 8; l++)
        if (n->leaves[l]) dump_node(k, n->leaves[l], path | (l << (3 * depth)), depth + 1, out, cap, cnt);
}
int orc_dump_planes(void* p, immesh_plane_rec* out, int64_t cap, int64_t* n_out) {
    OrcCtx* o = (OrcCtx*)p;
    int64_t cnt = 0;
    for (const auto& kv : o->vm.map) dump_node(kv.first, kv.second, 0, 0, out, cap, cnt);
    *n_out = cnt;
    return 0;
}

int orc_counters(void* p, immesh_counters_t* c, int32_t reset) {
    OrcCtx* o = (OrcCtx*)p;
    const Counters& k = o->vm.cnt;
    c->n_ds = k.n_ds; c->n_iter = k.n_iter; c->n_match = k.n_match; c->n_plane_tests = k.n_plane_tests; c->n_extra_probe = k.n_extra_probe;
    c->n_refits = k.n_refits; c->n_refit_pts = k.n_refit_pts; c->n_app = k.n_app; c->n_new = k.n_new; c->v_act = k.v_act; c->n_v = k.n_v;
    c->n_u = k.n_u; c->t_v = k.t_v; c->t_add = k.t_add; c->t_rem = k.t_rem; c->c1 = k.c1; c->c20 = k.c20; c->n_degenerate_skips = k.n_degenerate_skips;
    c->n_root_voxels = (int64_t)o->vm.map.size(); c->n_nodes = 0; c->n_vertices = (int64_t)o->mesher.verts.size();
    c->n_triangles_live = (int64_t)o->mesher.live_triangle_count();
    if (reset) o->vm.cnt = Counters();
    return 0;
}

// test tap: the Point_with_var lists of the last map_build (which = 0: point = world), map_update / process_scan (1: point = world, scan order) and
// register / residuals (2: body point, f32-rounded world point) -- tests/test_ref_voxelmap.py feeds them to the reference's own functions
void orc_debug_tap(void* p, int32_t on) { ((OrcCtx*)p)->reg.tap_on = on != 0; }
int64_t orc_debug_pv(void* p, int32_t which, double* pt, double* pt_world, double* var9, int64_t cap) {
    OrcCtx* o = (OrcCtx*)p;
    if (which < 0 || which > 2) return -1;
    const std::vector<PointWithVar>& v = o->reg.tap[which];
    for (int64_t i = 0; i < (int64_t)v.size() && i < cap; i++) {
        for (int k = 0; k < 3; k++) { if (pt) pt[i * 3 + k] = v[i].p[k]; if (pt_world) pt_world[i * 3 + k] = v[i].pw[k]; }
        if (var9) std::memcpy(var9 + i * 9, v[i].var, 72);
    }
    return (int64_t)v.size();
}

// ---- fine-grained hooks used only by the oracle's own unit tests ------------------------------------------------
void orc_calc_body_var(const double* pb, float range_inc, float degree_inc, double* var9) {
    double p[3] = {pb[0], pb[1], pb[2]};
    calc_body_var(p, range_inc, degree_inc, var9);
}
void orc_sym3_eigen(const double* A9, double* evals3, double* V9) { sym3_eigen_jacobi(A9, evals3, V9); }
int orc_inv(const double* A, double* Ainv, int n) { return inv_gauss_jordan(A, Ainv, n) ? 0 : -1; }
void orc_key(const double* p3, double voxel_size, int64_t* key3) {
    const double q[3] = {p3[0] / voxel_size, p3[1] / voxel_size, p3[2] / voxel_size};
    Key k = key_from_quotient(q);
    key3[0] = k.x; key3[1] = k.y; key3[2] = k.z;
}
// 2-D Delaunay of n points -> triangles (local indices); returns triangle count, writes up to cap triangles
static bool g_dt_link_free = false;
void orc_delaunay_force_link_free(int on) { g_dt_link_free = on != 0; }   // tests: see Delaunay2D::force_link_free
int orc_delaunay2d(const double* xy, int n, int32_t* tris, int cap) {
    Delaunay2D dt;
    dt.force_link_free = g_dt_link_free;
    std::vector<int> f;
    dt.run(xy, n, f);
    const int nt = (int)f.size() / 3;
    for (int i = 0; i < nt && i < cap; i++) { tris[i * 3] = f[i * 3]; tris[i * 3 + 1] = f[i * 3 + 1]; tris[i * 3 + 2] = f[i * 3 + 2]; }
    return nt;
}
// the mesher's per-voxel triangulation (delaunay_triangulation restated, orc_mesher.hpp) on a free-standing vertex set: pos n x 3 (ids = 0..n-1);
// short_axis_out = the axis it derived; returns the number of ints written (3 per accepted face, local ids in emission order)
int orc_voxel_delaunay(const double* pos, int n, double* short_axis_out, int32_t* tris_out, int cap) {
    Mesher m;
    m.verts.resize(n);
    std::vector<int> ids(n);
    for (int i = 0; i < n; i++) { ids[i] = i; for (int k = 0; k < 3; k++) { m.verts[i].pos[k] = pos[i * 3 + k]; m.verts[i].smooth[k] = pos[i * 3 + k]; } }
    std::vector<int> t;
    double sa[3] = {0, 0, 0};
    m.delaunay_triangulation(ids, sa, t);
    for (int k = 0; k < 3; k++) short_axis_out[k] = sa[k];
    for (size_t i = 0; i < t.size() && (int)i < cap; i++) tris_out[i] = t[i];
    return (int)t.size();
}
// correct_triangle_index restated (Mesher::flip_of) on three smoothed positions
int orc_flip_of(const double* a, const double* b, const double* c, const double* cam, const double* short_axis) {
    Mesher m;
    m.verts.resize(3);
    const double* src[3] = {a, b, c};
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { m.verts[i].pos[k] = src[i][k]; m.verts[i].smooth[k] = src[i][k]; }
    return m.flip_of(Tri{0, 1, 2}, cam, short_axis);
}
// exact kNN over the mesher's current vertex set (for validation against oracle/_ref's real ikd-Tree)
int orc_mesh_knn(void* p, const float* q, int k, double r_max, int32_t* ids, float* d2) {
    OrcCtx* o = (OrcCtx*)p;
    std::vector<Mesher::NN> nn;
    o->mesher.knn(q, k, r_max, nn);
    for (size_t i = 0; i < nn.size(); i++) { ids[i] = nn[i].id; d2[i] = nn[i].d2; }
    return (int)nn.size();
}
int orc_mesh_live_triangles(void* p, int32_t* out, int64_t cap) {
    OrcCtx* o = (OrcCtx*)p;
    std::vector<int> t;
    o->mesher.live_triangles(t);
    const int64_t n = (int64_t)t.size() / 3;
    for (int64_t i = 0; i < n && i < cap; i++) for (int k = 0; k < 3; k++) out[i * 3 + k] = t[i * 3 + k];
    return (int)n;
}
int orc_mesh_vertices(void* p, double* pos, double* smooth, int64_t cap) {
    OrcCtx* o = (OrcCtx*)p;
    const int64_t n = (int64_t)o->mesher.verts.size();
    for (int64_t i = 0; i < n && i < cap; i++) for (int k = 0; k < 3; k++) { if (pos) pos[i * 3 + k] = o->mesher.verts[i].pos[k]; if (smooth) smooth[i * 3 + k] = o->mesher.verts[i].smooth[k]; }
    return (int)n;
}

}  // extern "C"
