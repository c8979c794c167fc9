// Host orchestration of the legacy registration path (`voxel_map_en = false`): device point map + "Old map ICP" iterated
// update (src/voxel_mapping.cpp:1400-1480 matcher, :1487-1650 H rows + EKF with R_inv = 1 / LASER_POINT_COV).  The pools are allocated on the
// first immesh_ikd_build, so contexts that never use the legacy path pay nothing.
#include "host_ctx.hpp"
#include <algorithm>
#include <cmath>

static int64_t ikd_np2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }

static int ikd_alloc(immesh_ctx* c) {
    IkdHost& h = c->ikd;
    if (h.ready) return 0;
    const int64_t cap_cells = c->cfg.cap_root_voxels > 0 ? c->cfg.cap_root_voxels : (1 << 20);
    const int64_t slots = ikd_np2(cap_cells * 2);
    int rc;
#define A(ptr, n) if ((rc = c->dalloc(&(ptr), (size_t)(n)))) return rc
    A(h.m.keys, slots); A(h.m.count, slots); A(h.m.pts, slots * IKD_CELL_PTS); A(h.m.best, slots); A(h.m.stamp, slots);
    A(h.m.touched, c->cap_scan); A(h.m.counters, 8);
    A(h.d_near, c->cap_scan * IKD_KNN * 3); A(h.d_near_n, c->cap_scan); A(h.d_sel, c->cap_scan); A(h.d_norm, c->cap_scan * 4);
    A(h.d_part, ((c->cap_scan + 63) / 64) * 32); A(h.d_out, 48); A(h.d_q, c->cap_scan * IKD_KNN);
#undef A
    h.m.mask = (uint64_t)slots - 1; h.m.cap_cells = (int32_t)cap_cells; h.m.seq = 0; h.m.ds = 0.5f;
    h.cap_pts = c->cap_scan;
    h.ready = true;
    return 0;
}
static int ikd_reset(immesh_ctx* c) {
    IkdHost& h = c->ikd;
    const size_t slots = (size_t)h.m.mask + 1;
    HIPCHK(c, hipMemsetAsync(h.m.keys, 0xFF, slots * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(h.m.count, 0, slots * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(h.m.stamp, 0, slots * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(h.m.counters, 0, 8 * 4, c->stream));
    h.m.seq = 0;
    h.localmap_initialized = false;
    return 0;
}
static int ikd_check(immesh_ctx* c) {   // after a stream sync
    int32_t cnt[8];
    HIPCHK(c, hipMemcpy(cnt, c->ikd.m.counters, sizeof(cnt), hipMemcpyDeviceToHost));
    if (cnt[2] == 1) { c->err = "legacy point map: cell hash full (cap_root_voxels)"; return IMMESH_E_CAPACITY; }
    if (cnt[2] == 2) { c->err = "legacy point map: more than 8 points of the first scan in one downsample box"; return IMMESH_E_CAPACITY; }
    return 0;
}

extern "C" {

// m_ikdtree.set_downsample_param(m_filter_size_map_min); m_ikdtree.Build(m_feats_down_world->points)   src/voxel_mapping.cpp:1906-1914
int immesh_ikd_build(immesh_ctx* c, const float* pts_world_xyz, int32_t n, double downsample_size) {
    if (!c || !pts_world_xyz || n <= 0 || n > c->cap_scan || !(downsample_size > 0)) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    int rc = ikd_alloc(c);
    if (rc) return rc;
    if ((rc = ikd_reset(c))) return rc;
    c->ikd.m.ds = (float)downsample_size;
    const void* d;
    if ((rc = resolve_input(c, pts_world_xyz, (size_t)n * 12, c->d_pts_down, &d))) return rc;
    launch_ikd_build(c->stream, c->ikd.m, (const float*)d, n);
    const int32_t next_id = n;
    HIPCHK(c, hipMemcpyAsync(c->ikd.m.counters + 3, &next_id, 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ikd_check(c);
}

// m_ikdtree.Add_Points(m_feats_down_world->points, true)   src/ImMesh_mesh_reconstruction.cpp:426-443
int immesh_ikd_add_points(immesh_ctx* c, const float* pts_world_xyz, int32_t n) {
    if (!c || !pts_world_xyz || n <= 0 || n > c->cap_scan) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    if (!c->ikd.ready) { c->err = "immesh_ikd_build first"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    const void* d;
    int rc = resolve_input(c, pts_world_xyz, (size_t)n * 12, c->d_pts_down, &d);
    if (rc) return rc;
    c->ikd.m.seq++;
    HIPCHK(c, hipMemsetAsync(c->ikd.m.counters + 1, 0, 4, c->stream));
    launch_ikd_add(c->stream, c->ikd.m, (const float*)d, n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ikd_check(c);
}

static int ikd_delete(immesh_ctx* c, const IkdBoxes& bx, int32_t* n_deleted) {
    IkdHost& h = c->ikd;
    int32_t* d_cnt = h.m.counters + 4;
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 4, c->stream));
    launch_ikd_delete_boxes(c->stream, h.m, bx, d_cnt);
    int32_t v = 0;
    HIPCHK(c, hipMemcpyAsync(&v, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_deleted) *n_deleted = v;
    return 0;
}
// m_ikdtree.Delete_Point_Boxes(boxes)
int immesh_ikd_delete_boxes(immesh_ctx* c, const float* boxes, int32_t nb, int32_t* n_deleted) {
    if (!c || !boxes || nb < 0 || nb > 3 || !c->ikd.ready) { if (c) c->err = "bad arguments (at most 3 boxes per call)"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    IkdBoxes bx;
    bx.n = nb;
    std::memcpy(bx.b, boxes, (size_t)nb * 24);
    if (n_deleted) *n_deleted = 0;
    return nb ? ikd_delete(c, bx, n_deleted) : 0;
}
// Voxel_mapping::laser_map_fov_segment   src/voxel_mapping_common.cpp:214-288
int immesh_ikd_fov_segment(immesh_ctx* c, const double* pos, double cube_len, double detection_range_d, int32_t* n_deleted) {
    if (!c || !pos || !(cube_len > 0) || !c->ikd.ready) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    IkdHost& h = c->ikd;
    const float MOV_THRESHOLD = 1.5f, det = (float)detection_range_d;   // src/voxel_mapping.hpp:136-137 (float members)
    if (n_deleted) *n_deleted = 0;
    if (!h.localmap_initialized) {
        for (int i = 0; i < 3; i++) { h.lm_min[i] = (float)(pos[i] - cube_len / 2.0); h.lm_max[i] = (float)(pos[i] + cube_len / 2.0); }
        h.localmap_initialized = true;
        return 0;
    }
    float edge[3][2];
    bool move = false;
    for (int i = 0; i < 3; i++) {
        edge[i][0] = (float)std::fabs(pos[i] - (double)h.lm_min[i]);
        edge[i][1] = (float)std::fabs(pos[i] - (double)h.lm_max[i]);
        move = move || edge[i][0] <= MOV_THRESHOLD * det || edge[i][1] <= MOV_THRESHOLD * det;
    }
    if (!move) return 0;
    const float shift = (float)std::max((cube_len - 2.0 * MOV_THRESHOLD * det) * 0.5 * 0.9, double(det * (MOV_THRESHOLD - 1)));
    float nmin[3] = {h.lm_min[0], h.lm_min[1], h.lm_min[2]}, nmax[3] = {h.lm_max[0], h.lm_max[1], h.lm_max[2]};
    IkdBoxes bx;
    bx.n = 0;
    for (int i = 0; i < 3; i++) {
        float* b = bx.b[bx.n];
        for (int a = 0; a < 3; a++) { b[a] = h.lm_min[a]; b[3 + a] = h.lm_max[a]; }
        if (edge[i][0] <= MOV_THRESHOLD * det) { nmax[i] -= shift; nmin[i] -= shift; b[i] = h.lm_max[i] - shift; bx.n++; }
        else if (edge[i][1] <= MOV_THRESHOLD * det) { nmax[i] += shift; nmin[i] += shift; b[3 + i] = h.lm_min[i] + shift; bx.n++; }
    }
    std::memcpy(h.lm_min, nmin, sizeof(nmin)); std::memcpy(h.lm_max, nmax, sizeof(nmax));
    return bx.n ? ikd_delete(c, bx, n_deleted) : 0;
}

int immesh_ikd_size(immesh_ctx* c, int64_t* n) {
    if (!c || !n) return IMMESH_E_INVAL;
    *n = 0;
    if (!c->ikd.ready) return 0;
    (void)hipSetDevice(c->cfg.device);
    int32_t v = 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(&v, c->ikd.m.counters, 4, hipMemcpyDeviceToHost));
    *n = v;
    return 0;
}

int immesh_ikd_dump(immesh_ctx* c, float* xyz, int64_t cap, int64_t* n_out) {
    if (!c || !n_out || !c->ikd.ready) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    float* d_out = nullptr;
    if (xyz && cap > 0) HIPCHK(c, hipMalloc((void**)&d_out, (size_t)cap * 12));
    HIPCHK(c, hipMemsetAsync(c->d_dump_count, 0, 8, c->stream));
    launch_ikd_dump(c->stream, c->ikd.m, d_out, d_out ? cap : 0, c->d_dump_count);
    unsigned long long cnt = 0;
    hipError_t e = hipMemcpyAsync(&cnt, c->d_dump_count, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && d_out) e = hipMemcpy(xyz, d_out, (size_t)std::min<int64_t>(cap, (int64_t)cnt) * 12, hipMemcpyDeviceToHost);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) { c->err = std::string("ikd_dump: ") + hipGetErrorString(e); return IMMESH_E_HIP; }
    *n_out = (int64_t)cnt;
    return 0;
}

// KD_TREE::Nearest_Search(point, 5, ..)   include/ikd-Tree/ikd_Tree.cpp:440-476 -- parity hook for the matcher's search
int immesh_ikd_knn(immesh_ctx* c, const float* q_xyz, int32_t nq, float* nn_xyz, float* d2, int32_t* n_found) {
    if (!c || !q_xyz || nq <= 0 || nq > c->cap_scan || !c->ikd.ready) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    IkdHost& h = c->ikd;
    const void* d;
    int rc = resolve_input(c, q_xyz, (size_t)nq * 12, c->d_pts_down, &d);
    if (rc) return rc;
    IkdMatchParams mp;
    std::memset(&mp, 0, sizeof(mp));
    HIPCHK(c, hipMemsetAsync(h.d_q, 0, (size_t)nq * IKD_KNN * 4, c->stream));
    launch_ikd_match(c->stream, h.m, mp, (const float*)d, nq, 0, h.d_near, h.d_near_n, h.d_sel, h.d_norm, h.d_q);
    if (nn_xyz) HIPCHK(c, hipMemcpyAsync(nn_xyz, h.d_near, (size_t)nq * IKD_KNN * 12, hipMemcpyDeviceToHost, c->stream));
    if (d2) HIPCHK(c, hipMemcpyAsync(d2, h.d_q, (size_t)nq * IKD_KNN * 4, hipMemcpyDeviceToHost, c->stream));
    if (n_found) HIPCHK(c, hipMemcpyAsync(n_found, h.d_near_n, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// Voxel_mapping::lio_state_estimation with m_use_new_map == false
int immesh_ikd_register(immesh_ctx* c, const float* pts_down_body_xyz, int32_t n_ds, const double* state_prior, double* state_inout, double laser_point_cov,
                        int32_t* n_iter_out, int32_t* n_match_out, double* res_mean_out, int32_t* match_idx, float* normals_pd2) {
    if (!c || !pts_down_body_xyz || n_ds <= 0 || n_ds > c->cap_scan || !state_prior || !state_inout || !(laser_point_cov > 0) || !c->ikd.ready) {
        if (c) c->err = "bad arguments (immesh_ikd_build first)";
        return IMMESH_E_INVAL;
    }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    IkdHost& h = c->ikd;
    hipStream_t s = c->stream;
    const void* d;
    int rc = resolve_input(c, pts_down_body_xyz, (size_t)n_ds * 12, c->d_pts_down, &d);
    if (rc) return rc;
    imh::State prior, st;
    imh::load_state(state_prior, prior); imh::load_state(state_inout, st);
    imh::EkfLoop ekf;
    bool nearest_search_en = true;
    double out48[48];
    int iters = 0;
    HIPCHK(c, hipMemsetAsync(h.d_sel, 1, (size_t)n_ds, s));   // m_point_selected_surf.resize(n, true)
    for (int it = 0; it < c->cfg.max_iter; it++) {
        iters++;
        IkdMatchParams mp;
        std::memcpy(mp.R, st.R, 72); std::memcpy(mp.t, st.t, 24); std::memcpy(mp.extR, c->cfg.extR, 72); std::memcpy(mp.extT, c->cfg.extT, 24);
        mp.r_inv = 1.0 / laser_point_cov;
        launch_ikd_match(s, h.m, mp, (const float*)d, n_ds, nearest_search_en ? 1 : 2, h.d_near, h.d_near_n, h.d_sel, h.d_norm, nullptr);
        launch_ikd_reduce(s, mp, (const float*)d, n_ds, h.d_sel, h.d_norm, h.d_part, h.d_out);
        HIPCHK(c, hipMemcpyAsync(out48, h.d_out, sizeof(out48), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (n_match_out) *n_match_out = (int)out48[42];
        if (res_mean_out) *res_mean_out = out48[42] > 0 ? out48[43] / out48[42] : 0.0;
        const int rematch_before = ekf.rematch_num;
        const bool stop = ekf.step(out48, out48 + 36, prior, st, it, c->cfg.max_iter);
        nearest_search_en = ekf.rematch_num != rematch_before;   // "Rematch Judgement", voxel_mapping.cpp:1626-1632
        if (stop) break;
    }
    imh::store_state(st, state_inout);
    if (n_iter_out) *n_iter_out = iters;
    if (match_idx || normals_pd2) {   // m_laserCloudOri / m_corr_normvect of the last iteration, ascending scan index
        std::vector<int8_t> sel(n_ds);
        std::vector<float> nv((size_t)n_ds * 4);
        HIPCHK(c, hipMemcpy(sel.data(), h.d_sel, (size_t)n_ds, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(nv.data(), h.d_norm, (size_t)n_ds * 16, hipMemcpyDeviceToHost));
        int k = 0;
        for (int i = 0; i < n_ds; i++)
            if (sel[i] && std::fabs((double)nv[(size_t)i * 4 + 3]) <= 2.0) {
                if (match_idx) match_idx[k] = i;
                if (normals_pd2) for (int a = 0; a < 4; a++) normals_pd2[(size_t)k * 4 + a] = nv[(size_t)i * 4 + a];
                k++;
            }
    }
    return 0;
}

}  // extern "C"
