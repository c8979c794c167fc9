// Host orchestration of the mesher kernels (mesh_kernels.hip): incremental_mesh_reconstruction (ImMesh_mesh_reconstruction.cpp:92-267)
// as a sequence of launches on the context stream.  No per-point / per-voxel / per-triangle work happens on the host: it only
// sizes launches from a handful of device counters and moves the result lists on immesh_mesh_fetch.
#include "host_ctx.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>

static int64_t np2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }

// worker-thread flavour of HIPCHK: the error text goes to MeshHost::err (the caller thread owns immesh_ctx::err)
#define MHIPCHK(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            (ctx)->mesh_host.err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return IMMESH_E_HIP;                                                                            \
        }                                                                                                   \
    } while (0)

static void mesh_worker_main(immesh_ctx* c);

int mesh_alloc(immesh_ctx* c) {
    const immesh_config& g = c->cfg;
    MeshDev& m = c->mesh;
    MeshHost& h = c->mesh_host;
    std::memset(&m, 0, sizeof(m));
    std::memset(h.cum, 0, sizeof(h.cum));
    if (g.mesh_min_spacing <= 0 || g.mesh_voxel <= 0 || g.mesh_append_budget <= 0) { c->err = "invalid meshing parameters"; return IMMESH_E_INVAL; }
    {   // every dedupe cell holds at most one vertex, so a mesh voxel holds at most (cells per axis)^3
        const int per_axis = (int)(g.mesh_voxel / g.mesh_min_spacing) + 1;
        if ((int64_t)per_axis * per_axis * per_axis > MV_VOX_CAP) { c->err = "mesh_voxel / mesh_min_spacing ratio above 4 is not supported (voxel point list stride)"; return IMMESH_E_INVAL; }
    }
    const int64_t cap_verts = g.cap_vertices > 0 ? g.cap_vertices : (1 << 22);
    const int64_t cap_voxels = cap_verts;
    const int64_t cap_tris = g.cap_triangles > 0 ? g.cap_triangles : cap_verts * 4;
    const int64_t cap_adj = cap_verts + cap_tris / 3;
    const int64_t cap_cand = c->cap_scan;
    const int64_t cap_active = std::min<int64_t>(cap_cand, 1 << 17);
    const int64_t cap_list = 1 << 22;
    if (cap_verts > 0x3fffffff || cap_tris > 0x3fffffff) { c->err = "mesh capacity too large for 31-bit indices"; return IMMESH_E_INVAL; }
    int rc;
#define A(ptr, n) if ((rc = c->dalloc(&(ptr), (size_t)(n)))) return rc
    A(m.v_pos, cap_verts * 3); A(m.v_smooth, cap_verts * 3); A(m.v_smooth_new, cap_verts * 3); A(m.v_voxel, cap_verts);
    const int64_t gcap = np2(cap_verts * 2), xcap = np2(cap_voxels * 2), tcap = np2(cap_tris * 2), ccap = np2(cap_cand * 4);
    A(m.g_ent, gcap); m.g_mask = (uint64_t)gcap - 1;
    A(m.x_ent, xcap); m.x_mask = (uint64_t)xcap - 1;
    A(m.vx_key, cap_voxels); A(m.vx_npts, cap_voxels); A(m.vx_pts, cap_voxels * MV_VOX_CAP); A(m.vx_meshing_times, cap_voxels);
    A(m.vx_new_added, cap_voxels); A(m.vx_stamp, cap_voxels); A(m.vx_rank, cap_voxels); A(m.vx_rank_seq, cap_voxels); A(m.vx_short_axis, cap_voxels * 3);
    A(m.t_v, cap_tris * 3); A(m.t_word, cap_tris); A(m.t_live, cap_tris); A(m.t_rem_seq, cap_tris); A(m.t_flip, cap_tris);
    A(m.th_slots, tcap); m.th_mask = (uint64_t)tcap - 1;
    A(m.a_head, cap_verts); A(m.a_chunks, cap_adj * MV_ADJ_STRIDE);
    A(m.sc, SC_COUNT); A(m.pc, PC_COUNT);
    A(m.cand_status, cap_cand); A(m.cand_vox, cap_cand); A(m.cand_cell, cap_cand); A(m.cand_next, cap_cand); A(m.cand_rank, cap_cand); A(m.cand_pt, cap_cand + 32768); A(m.cand_flags, cap_cand); A(m.bin_cnt, 2 * (1024 + 1));
    A(m.ch_keys, ccap); A(m.ch_head, ccap);
    A(m.recent, cap_cand);
    int64_t cap_active_p2 = 1; while (cap_active_p2 < cap_active) cap_active_p2 <<= 1;   // mesh_append_finish_kernel's ordering network pads to a power of two
    A(m.act_key, cap_active_p2); A(m.act_vox, cap_active_p2); A(m.act_key_s, cap_active); A(m.act_vox_s, cap_active);
    A(m.rel_ids, cap_active * MV_REL_CAP); A(m.rel_n, cap_active); A(m.rel_nq, cap_active);
    A(m.vox_tris, cap_active * 2 * MV_REL_CAP); A(m.vox_ntris, cap_active);
    A(m.tri_fh, cap_active * 256); A(m.tri_nf, cap_active); A(m.tri_axis, cap_active * 3);
    A(m.list_add, cap_list); A(m.list_rem, cap_list); A(m.list_upd, cap_list); A(m.list_smooth, cap_list);
    const bool shard_mesh = g.shard_world > 1 && g.shard_mesh != 0;
    if (shard_mesh && g.shard_brick_log2 > 0 && g.shard_brick_log2 < 2) { c->err = "sharded mesher: bricks below 4 voxels per axis are not supported (the boundary band reaches 2 voxels)"; return IMMESH_E_INVAL; }
    if (shard_mesh) A(m.list_smooth_rx, cap_list);
    for (int k = 0; k < MESH_NPAR; k++) {
        MeshOutSet& o = h.outs[k];
        A(o.tri_add, cap_list * 3); A(o.flip_add, cap_list); A(o.tri_rem, cap_list * 3); A(o.tri_upd, cap_list * 3); A(o.flip_upd, cap_list);
        A(o.smooth_ids, cap_list); A(o.smooth_xyz, cap_list * 3);
        if (shard_mesh) { A(o.own_add, cap_list); A(o.own_rem, cap_list); A(o.own_upd, cap_list); } else o.own_add = o.own_rem = o.own_upd = nullptr;
    }
    for (int k = 0; k < MESH_WORLD_BUFS; k++) A(h.d_world[k], cap_cand * 4);
    A(m.tick0, 16 * MESH_NPAR);   // (16 64-bit words per set: [0] the job's start, [1..] the phase marks)
    HIPCHK(c, hipMemsetAsync(m.tick0, 0, 128 * MESH_NPAR, c->stream));
    A(m.dv_scratch, (size_t)32 * 128 * 1024);   // (MV_GEN_BLOCKS x MV_GEN_SCRATCH of mesh_kernels.hip)
    A(h.p_a, cap_list);
    h.sort_temp_bytes = exclusive_sum_temp_bytes((int)cap_cand) + 256;
    { char* t; A(t, h.sort_temp_bytes); h.d_sort_temp = t; }
    { unsigned long long* t; A(t, (size_t)(4 * cap_list + cap_active + 5 * 1024) * 2); h.d_sort_recs = t; }
    { unsigned long long* t; A(t, (size_t)(cap_active + 5 * 1024) * 2); h.d_sort_recs_a = t; }
    // further sets of everything a job's phases hand to each other while other jobs are in flight (set 0 = the arrays above)
    MeshDev mx[MESH_NPAR];
    for (int p = 1; p < MESH_NPAR; p++) {
        MeshDev& m1 = mx[p];
        std::memset(&m1, 0, sizeof(m1));
        A(m1.v_smooth_new, cap_verts * 3); A(m1.vx_rank, cap_voxels); A(m1.vx_rank_seq, cap_voxels);
        A(m1.sc, SC_COUNT);
        A(m1.act_key, cap_active_p2); A(m1.act_vox, cap_active_p2); A(m1.act_key_s, cap_active); A(m1.act_vox_s, cap_active);
        A(m1.rel_ids, cap_active * MV_REL_CAP); A(m1.rel_n, cap_active); A(m1.rel_nq, cap_active);
        A(m1.tri_fh, cap_active * 256); A(m1.tri_nf, cap_active); A(m1.tri_axis, cap_active * 3);
    }
#undef A
    m.cap_verts = (int32_t)cap_verts; m.cap_voxels = (int32_t)cap_voxels; m.cap_tris = (int32_t)cap_tris; m.cap_adj_chunks = (int32_t)cap_adj;
    m.cap_cand = (int32_t)cap_cand; m.cap_active = (int32_t)cap_active; m.cap_list = (int32_t)cap_list;
    m.min_spacing = g.mesh_min_spacing; m.voxel = g.mesh_voxel; m.accept = g.mesh_voxel * 1.25;
    m.shard_rank = shard_mesh ? g.shard_rank : 0; m.shard_world = shard_mesh ? g.shard_world : 1; m.shard_brick_log2 = g.shard_brick_log2 > 0 ? g.shard_brick_log2 : 5; m.shard_scheme = g.shard_scheme == 1 ? 1 : 0;
    m.dbg = nullptr;
    if (getenv("IMMESH_DEBUG")) { unsigned long long* t; if ((rc = c->dalloc(&t, 64))) return rc; m.dbg = t; (void)hipMemset(t, 0, 512); }
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemsetAsync(m.g_ent, 0xFF, (size_t)gcap * sizeof(MeshGridEnt), s));   // (key == ~0: empty)
    HIPCHK(c, hipMemsetAsync(m.x_ent, 0xFF, (size_t)xcap * sizeof(MeshVoxEnt), s));   // (key == ~0: empty, val == -1)
    HIPCHK(c, hipMemsetAsync(m.th_slots, 0xFF, (size_t)tcap * 4, s));
    HIPCHK(c, hipMemsetAsync(m.a_head, 0xFF, (size_t)cap_verts * 4, s));
    HIPCHK(c, hipMemsetAsync(m.sc, 0, SC_COUNT * 4, s));
    HIPCHK(c, hipMemsetAsync(m.bin_cnt, 0, 2 * (1024 + 1) * 4, s));
    HIPCHK(c, hipMemsetAsync(m.pc, 0, PC_COUNT * 4, s));
    if (m.shard_world > 1) {   // exchange staging of the sharded mesher: this rank's block, and every rank's blocks side by side (mesh_exchange)
        h.xcap_bytes = 16 + (size_t)cap_list * sizeof(MeshSmRec);                               // everything a rank can have to send
        h.xall_bytes = (size_t)std::min(m.shard_world, 64) * (16 + (size_t)8192 * sizeof(MeshSmRec));   // the first blocks of all ranks (mesh_exchange: XCAP_* x record size)
        { char* t; if ((rc = c->dalloc(&t, h.xcap_bytes))) return rc; h.d_xsend = t; }
        { char* t; if ((rc = c->dalloc(&t, h.xall_bytes))) return rc; h.d_xall = t; }
    }
    HIPCHK(c, hipMemsetAsync(m.vx_rank_seq, 0, (size_t)cap_voxels * 4, s));
    for (int p = 1; p < MESH_NPAR; p++) {
        HIPCHK(c, hipMemsetAsync(mx[p].sc, 0, SC_COUNT * 4, s));
        HIPCHK(c, hipMemsetAsync(mx[p].vx_rank_seq, 0, (size_t)cap_voxels * 4, s));
    }
    for (int k = 0; k < MESH_NPAR; k++) {
        MeshDyn* t; if ((rc = c->dalloc(&t, 1))) return rc; h.d_dyn[k] = t;
        HIPCHK(c, hipHostMalloc((void**)&h.h_dyn[k], sizeof(MeshDyn), hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer((void**)&h.h_dyn_dev[k], h.h_dyn[k], 0));
        std::memset(h.h_dyn[k], 0, sizeof(MeshDyn));
        HIPCHK(c, hipHostMalloc((void**)&h.h_sc2[k], MESH_PUB_WORDS * 4, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer((void**)&h.h_sc2_dev[k], h.h_sc2[k], 0));
        std::memset(h.h_sc2[k], 0, MESH_PUB_WORDS * 4);
    }
    h.h_sc = h.h_sc2[0];
    m.dyn = h.d_dyn[0];
    static_assert(MESH_NPAR == 3, "the views below name the two other sets' stamps");
    m.vx_rank_seq_alt = mx[1].vx_rank_seq; m.vx_rank_seq_alt2 = mx[2].vx_rank_seq;
    {   // the views, one per set
        h.mpar[0] = m;
        for (int p = 1; p < MESH_NPAR; p++) {
            MeshDev& v = h.mpar[p];
            const MeshDev& m1 = mx[p];
            v = m;
            v.v_smooth_new = m1.v_smooth_new; v.vx_rank = m1.vx_rank; v.vx_rank_seq = m1.vx_rank_seq;
            v.vx_rank_seq_alt = p == 1 ? m.vx_rank_seq : mx[1].vx_rank_seq; v.vx_rank_seq_alt2 = p == 2 ? m.vx_rank_seq : mx[2].vx_rank_seq;
            v.sc = m1.sc; v.act_key = m1.act_key; v.act_vox = m1.act_vox; v.act_key_s = m1.act_key_s; v.act_vox_s = m1.act_vox_s;
            v.rel_ids = m1.rel_ids; v.rel_n = m1.rel_n; v.rel_nq = m1.rel_nq;
            v.tri_fh = m1.tri_fh; v.tri_nf = m1.tri_nf; v.tri_axis = m1.tri_axis;
            v.dyn = h.d_dyn[p];
            v.tick0 = m.tick0 + 16 * p;
        }
        for (int k = 0; k < MESH_NPAR; k++) {
            const MeshOutSet& o = h.outs[k];
            MeshDev& w = h.mpar[k];
            w.out_tri_add = o.tri_add; w.out_flip_add = o.flip_add; w.out_tri_rem = o.tri_rem; w.out_tri_upd = o.tri_upd; w.out_flip_upd = o.flip_upd;
            w.out_smooth_ids = o.smooth_ids; w.out_smooth_xyz = o.smooth_xyz;
            w.out_own_add = o.own_add; w.out_own_rem = o.own_rem; w.out_own_upd = o.own_upd;
        }
    }
    h.use_graph = getenv("IMMESH_NO_GRAPH") == nullptr;
    h.split_mode = getenv("IMMESH_NO_SPLIT") ? 0 : 2;
    if (const char* e = getenv("IMMESH_SPLIT")) h.split_mode = atoi(e) == 0 ? 0 : 1;
    h.room = 0;   // 0: the worker decides (two jobs in flight; three, with the triangulations on the third stream, while the mesher is behind)
    if (const char* e = getenv("IMMESH_MESH_ROOM")) h.room = std::min(std::max(atoi(e), 1), MESH_NPAR);   // (measurement knob: a fixed number of jobs in flight; 2 = rounds 1-5)
    h.pipeline = getenv("IMMESH_NO_PIPELINE") == nullptr;
    HIPCHK(c, hipHostMalloc((void**)&h.h_pc, PC_COUNT * 4));
    std::memset(h.h_pc, 0, PC_COUNT * 4);
    HIPCHK(c, hipStreamSynchronize(s));
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (getenv("IMMESH_NO_PRIORITY")) prio_least = 0;
    {
        // The mesher's two streams are plain non-blocking streams of the lowest priority.  IMMESH_MESH_CUS=n confines them to n CUs instead
        // (hipExtStreamCreateWithCUMask).  Round 1 ran with 160 of 256 CUs by default (+16 % when the mesher's lone wavefronts slowed the pose
        // chain's kernels); with the round-2 kernels the gain is within noise (3280 vs 3220 scans/s), and CU-masked streams are BLOCKING streams
        // on which event-timed launches failed intermittently (garbage counters / memory faults / hangs with the in-library profiler on,
        // tools/debug_profiler.sh) -- so the mask is opt-in.
        int ncu = 0;
        if (const char* e = getenv("IMMESH_MESH_CUS")) ncu = atoi(e);
        if (ncu >= 8 && ncu < 1024) {
            uint32_t mask[32];
            std::memset(mask, 0, sizeof(mask));
            for (int i = 0; i < ncu; i++) mask[i >> 5] |= 1u << (i & 31);
            HIPCHK(c, hipExtStreamCreateWithCUMask(&h.stream, 32, mask));
            HIPCHK(c, hipExtStreamCreateWithCUMask(&h.stream_b, 32, mask));
        } else {
            HIPCHK(c, hipStreamCreateWithPriority(&h.stream, hipStreamNonBlocking, prio_least));
            HIPCHK(c, hipStreamCreateWithPriority(&h.stream_b, hipStreamNonBlocking, prio_least));
        }
        HIPCHK(c, hipStreamCreateWithPriority(&h.stream_fetch, hipStreamNonBlocking, prio_least));
        h.stream_q = h.stream_fetch;   // (no HSA queue of its own: an extra queue in the process costs every stream -- immesh_create; queries and fetches are both rare and short)
    }
    for (int k = 0; k < MESH_WORLD_BUFS; k++) HIPCHK(c, hipEventCreateWithFlags(&h.ev_ready[k], hipEventDisableTiming));
    for (int k = 0; k < MESH_NPAR; k++) {
        HIPCHK(c, hipEventCreateWithFlags(&h.ev_a[k], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&h.ev_c[k], hipEventDisableTiming));
        HIPCHK(c, hipEventCreate(&h.ev_b[k]));   // (doubles as the end time of the job: every record is a barrier packet on the phase-B chain)
    }
    for (int k = 0; k < MESH_NPAR; k++) std::memset(&h.res[k].sizes, 0, sizeof(immesh_mesh_sizes_t));
    h.stop = false;
    h.worker = std::thread(mesh_worker_main, c);
    h.ready = true;
    return 0;
}

static void mesh_print_marks(const MeshHost& h) {
    // the last 64 jobs (minus the newest 4: the wind-down): mean kernel-entry times relative to the job's start, the job period and the scan thread's wait
    if (h.mark_n < 24 || h.wait_calls < 24) return;
    const long long hi = h.mark_n - 4, lo = std::max<long long>(hi - 56, 1);
    double rel[MESH_N_MARKS] = {}, period = 0, bgap = 0;
    long long n = 0;
    for (long long j = lo; j < hi; j++, n++) {
        const unsigned long long* mk = h.mark_ring[j & 63];
        const unsigned long long* pv = h.mark_ring[(j - 1) & 63];
        for (int k = 1; k < MESH_N_MARKS; k++) rel[k] += (mk[k] >= mk[0] && mk[k] - mk[0] < 100000000ull) ? 0.01 * (double)(long long)(mk[k] - mk[0]) : -1e9;   // (a mark of another arrangement's kernel: stale)
        period += 0.01 * (double)(long long)(mk[MESH_N_MARKS - 1] - pv[MESH_N_MARKS - 1]);
        bgap += 0.01 * (double)(long long)((mk[12] > mk[0] ? mk[12] : mk[5]) - pv[MESH_N_MARKS - 1]);
    }
    double w = 0; long long nw = 0;
    for (long long j = std::max<long long>(h.wait_calls - 60, 0); j < h.wait_calls - 4; j++, nw++) w += 1e-3 * (double)h.wait_ring[j & 63];
    static const char* nm[MESH_N_MARKS] = {"begin", "prepare", "resolve", "finish", "knn", "tri64", "general", "finalize", "chunk_sort", "merge_emit", "commit_add", "publish", "diff64", "end"};
    fprintf(stderr, "[mesh marks] last %lld jobs, us after the job's start:", n);
    for (int k = 1; k < MESH_N_MARKS; k++) if (rel[k] >= 0) fprintf(stderr, " %s %.1f", nm[k], rel[k] / (double)n);
    fprintf(stderr, " | end-to-end %.1f us, previous job's end -> phase B's first kernel %.1f us | scan thread's wait for a world buffer %.1f us (last %lld scans)\n", period / (double)n, bgap / (double)n, nw ? w / (double)nw : 0.0, nw);
}
void mesh_free(immesh_ctx* c) {
    if (getenv("IMMESH_DEBUG_WAITS")) mesh_print_marks(c->mesh_host);
    if (getenv("IMMESH_DEBUG_WAITS") && c->mesh_host.wait_calls) fprintf(stderr, "[mesh] scan thread waited for a world buffer: %.1f us per scan over %lld scans\n", 1e-3 * (double)c->mesh_host.wait_ns / (double)c->mesh_host.wait_calls, c->mesh_host.wait_calls);
    if (getenv("IMMESH_DEBUG_WAITS") && c->mesh_host.job_ms_n) fprintf(stderr, "[mesh] device time per job (begin_scan's poll answered -> publish): %.1f us over %lld jobs\n", 1e3 * c->mesh_host.job_ms_sum / (double)c->mesh_host.job_ms_n, c->mesh_host.job_ms_n);
    MeshHost& h = c->mesh_host;
    if (h.worker.joinable()) {
        { std::lock_guard<std::mutex> lk(h.mu); h.stop = true; }
        h.cv_job.notify_all();
        h.worker.join();
    }
    if (h.stream) { (void)hipStreamSynchronize(h.stream); (void)hipStreamDestroy(h.stream); h.stream = nullptr; }
    if (h.stream_b) { (void)hipStreamSynchronize(h.stream_b); (void)hipStreamDestroy(h.stream_b); h.stream_b = nullptr; }
    if (h.stream_fetch) { (void)hipStreamSynchronize(h.stream_fetch); (void)hipStreamDestroy(h.stream_fetch); h.stream_fetch = nullptr; }
    if (h.h_fetch) { (void)hipHostFree(h.h_fetch); h.h_fetch = nullptr; h.h_fetch_bytes = 0; }
    h.stream_q = nullptr;
    if (h.q_host) { (void)hipHostFree(h.q_host); h.q_host = nullptr; h.q_host_bytes = 0; }
    if (h.q_dev) { (void)hipFree(h.q_dev); h.q_dev = nullptr; h.q_dev_bytes = 0; }
    if (h.q_exp) { (void)hipFree(h.q_exp); h.q_exp = nullptr; h.q_exp_bytes = 0; }
    if (h.exp_vtx) (void)hipFree(h.exp_vtx);
    if (h.exp_work) (void)hipFree(h.exp_work);
    if (h.exp_tmp) (void)hipFree(h.exp_tmp);
    h.d_xall = nullptr; h.xall_bytes = 0;   // (from the context's pool)
    if (h.d_xbig) (void)hipFree(h.d_xbig);
    h.d_xbig = nullptr; h.xbig_bytes = 0;
    h.exp_vtx = h.exp_work = h.exp_tmp = nullptr;
    for (int k = 0; k < MESH_WORLD_BUFS; k++) if (h.ev_ready[k]) (void)hipEventDestroy(h.ev_ready[k]);
    for (int k = 0; k < MESH_NPAR; k++) {
        if (h.ev_a[k]) (void)hipEventDestroy(h.ev_a[k]);
        if (h.ev_c[k]) (void)hipEventDestroy(h.ev_c[k]);
        if (h.ev_b[k]) (void)hipEventDestroy(h.ev_b[k]);
        if (h.graph_exec[k]) { (void)hipGraphExecDestroy(h.graph_exec[k]); h.graph_exec[k] = nullptr; }
        for (int q = 0; q < 2; q++) if (h.graph_exec_b[k][q]) { (void)hipGraphExecDestroy(h.graph_exec_b[k][q]); h.graph_exec_b[k][q] = nullptr; }
        if (h.h_dyn[k]) (void)hipHostFree(h.h_dyn[k]);
        if (h.h_sc2[k]) (void)hipHostFree(h.h_sc2[k]);
        h.h_dyn[k] = nullptr; h.h_sc2[k] = nullptr;
    }
    if (h.h_pc) (void)hipHostFree(h.h_pc);
    h.h_sc = h.h_pc = nullptr;
}

static int mesh_overflow(immesh_ctx* c) {
    const int f = c->mesh_host.h_sc[SC_OVERFLOW];
    if (!f) return 0;
    static const char* why[] = {"", "mesh-voxel hash full", "mesh-voxel pool exhausted (cap_vertices)", "candidate-cell table full", "vertex pool exhausted (cap_vertices)",
                                "mesh voxel lookup failed", "dedupe grid hash full", "mesh voxel holds more than 128 vertices", "more active voxels than cap (131072 per scan)",
                                "voxel neighbourhood above 1024 vertices", "triangle pool exhausted (cap_triangles)", "triangle hash full", "Delaunay cavity / triangle buffer overflow",
                                "adjacency chunk pool exhausted", "per-scan result list above 4194304 entries"};
    c->mesh_host.err = std::string("mesh map capacity: ") + why[(f > 0 && f < 15) ? f : 0];
    return IMMESH_E_CAPACITY;
}

// The launch sequence of one scan, in two phases (fixed grids; everything scan-specific is read from MeshDev::dyn on the device).
// Every launch has a fixed grid and takes its work-list length from device counters, so a whole scan is enqueued without a host round
// trip; the host reads the counters once, at the end of phase B.
// Phase A: a17-a19 (vertex admission, active voxels in rank order, neighbourhood search).
static int mesh_enqueue_a(immesh_ctx* c, const MeshDev& m, int par, hipStream_t s, const float* d_pts, int n_cand, int64_t ccap, bool resolve) {
    MeshHost& h = c->mesh_host;
    if (resolve) {
        launch_mesh_begin_scan(s, m, h.h_dyn_dev[par], (unsigned long long)ccap);
        launch_mesh_append_prepare(s, m, n_cand, d_pts);
        // every block of the launch is resident (<= 256 blocks), so the lowest undecided candidate can always decide: the loop terminates;
        // the iteration bound only guards against a hung device and is checked by the caller
        launch_mesh_append_resolve(s, m, n_cand, d_pts, 1 << 16);
    }
    if (n_cand <= 16384) {
        // per-scan sized candidate sets: ids, commit, voxel selection and the active voxels' order in ONE single-workgroup launch
        launch_mesh_append_finish(s, m, d_pts);
    } else {
        launch_mesh_append_flags(s, m, n_cand);
        exclusive_sum_i32(s, h.d_sort_temp, h.sort_temp_bytes, m.cand_rank, m.cand_rank, n_cand);
        launch_mesh_append_commit(s, m, n_cand, d_pts);
        launch_mesh_select_active(s, m, n_cand);
        // ascending (x,y,z) voxel order defines "earlier / later voxel" for the order-dependent parts (smoothed positions seen by
        // correct_triangle_index, which voxel's flip wins): the deterministic sequential order of the CPU checker
        launch_mesh_sort_emit(s, m, 0, h.d_sort_recs_a, nullptr);   // sorted active list + ranks
    }
    launch_mesh_knn(s, m);                                      // a18-a19
    return 0;
}
// Phase B: a20-a24 (triangulation, diff against the live set, commit: all removes, then all adds -- ImMesh_mesh_reconstruction.cpp:228-244;
// result lists sorted by triplet), then the counters go to the host.
static int mesh_enqueue_b(immesh_ctx* c, const MeshDev& m, int par, hipStream_t s, int part = 0, bool split = false) {   // part 1: triangulation only, 2: the rest
    MeshHost& h = c->mesh_host;                                // split: the triangulations ran on the third stream (mesh_tri64_kernel): only the diff is left
    if (part != 2) { if (split) launch_mesh_diff64(s, m); else launch_mesh_delaunay(s, m); }   // a20-a23
    if (part == 1) return 0;
    launch_mesh_finalize(s, m);                               // (+ the removals: Triangle_manager::remove_triangle_list)
    launch_mesh_sort_emit(s, m, 1, h.d_sort_recs, h.p_a);
    launch_mesh_commit_add(s, m, h.p_a);
    launch_mesh_publish(s, m, h.h_sc2_dev[par]);
    return 0;
}

// Sharded mesher: one exchange = ONE all-gather in the common case.  Every rank packs its records behind a 16-byte header {records, aux, -, -} in
// d_xsend; the first `cap_small` of them travel with the header in a fixed-size block.  The host reads the gathered headers (one strided copy): the
// sum of `aux` is the admission's "anybody undecided?", and when a rank had more records than the block holds (a scan over fresh ground, tiny bricks) a
// second gather carries blocks padded to the largest count.  `unpack(gathered, block bytes, records per block)` -- one launch for all ranks' blocks, the
// counts read on the device -- is enqueued behind it.
//   RCCL:      ncclAllGather on the device buffers, in-stream.     callbacks (tests on a gloo group): the blocks are staged through the host.
static constexpr size_t XHDR = 16;
static constexpr int XCAP_CAND = 8192;    // admission records (8 B) per rank and round in the first block: a 10 000-candidate scan sends a few hundred
static constexpr int XCAP_BAND = 4096;    // smoothed positions (32 B) / triangle marks (24 B) per rank and scan in the first block
static int mesh_exchange(immesh_ctx* c, hipStream_t s, size_t rec, int cap_small, const std::function<void(const void*, size_t, int)>& unpack, int64_t* aux_sum = nullptr) {
    MeshHost& h = c->mesh_host;
    const int world = c->cfg.shard_world;
    const size_t capb = XHDR + (size_t)cap_small * rec;
    if (aux_sum) *aux_sum = 0;
    if (world > 64 || (size_t)world * capb > h.xall_bytes) { h.err = "sharded mesher: exchange block above the staging buffers"; return IMMESH_E_CAPACITY; }
    int32_t hdr[64 * 4];
    int rc;
    auto big_buffer = [&](size_t need) -> int {
        if (need <= h.xbig_bytes) return 0;
        if (h.d_xbig) (void)hipFree(h.d_xbig);
        h.d_xbig = nullptr; h.xbig_bytes = 0;
        if (hipMalloc(&h.d_xbig, need + need / 4) != hipSuccess) { h.err = "hipMalloc(exchange buffer)"; return IMMESH_E_NOMEM; }
        h.xbig_bytes = need + need / 4;
        return 0;
    };
    if (c->rccl_comm) {
        if ((rc = rccl_allgather_bytes(c, h.d_xsend, h.d_xall, capb, s, &h.err))) return rc;
        h.xcalls++;
        MHIPCHK(c, hipMemcpy2DAsync(hdr, XHDR, h.d_xall, capb, XHDR, (size_t)world, hipMemcpyDeviceToHost, s));
        MHIPCHK(c, hipStreamSynchronize(s));
    } else {
        if (!h.allgather) { h.err = "sharded mesher: no collective registered (immesh_rccl_init or immesh_set_allgather)"; return IMMESH_E_INVAL; }
        int32_t mine[4] = {0, 0, 0, 0};
        MHIPCHK(c, hipMemcpyAsync(mine, h.d_xsend, XHDR, hipMemcpyDeviceToHost, s));
        MHIPCHK(c, hipStreamSynchronize(s));
        const size_t used = (size_t)std::min(std::max(mine[0], 0), cap_small) * rec;
        h.h_xsend.assign(capb, 0);
        std::memcpy(h.h_xsend.data(), mine, XHDR);
        if (used) MHIPCHK(c, hipMemcpy(h.h_xsend.data() + XHDR, (const char*)h.d_xsend + XHDR, used, hipMemcpyDeviceToHost));
        h.h_xrecv.resize((size_t)world * capb);
        if (h.allgather(h.h_xsend.data(), (int64_t)capb, h.h_xrecv.data(), h.allgather_user)) { h.err = "all-gather callback failed"; return IMMESH_E_INVAL; }
        h.xcalls++;
        for (int r = 0; r < world; r++) std::memcpy(hdr + 4 * r, h.h_xrecv.data() + (size_t)r * capb, XHDR);
    }
    int64_t maxc = 0;
    for (int r = 0; r < world; r++) { if (hdr[4 * r] < 0) { h.err = "sharded mesher: corrupt exchange header"; return IMMESH_E_HIP; } maxc = std::max<int64_t>(maxc, hdr[4 * r]); if (aux_sum) *aux_sum += hdr[4 * r + 1]; }
    if ((size_t)maxc * rec + XHDR > h.xcap_bytes) { h.err = "sharded mesher: exchange buffer overflow"; return IMMESH_E_CAPACITY; }
    if (maxc <= cap_small) {
        if (!c->rccl_comm) {   // (callbacks: the gathered blocks go up; RCCL left them in d_xall)
            for (int r = 0; r < world; r++)
                MHIPCHK(c, hipMemcpyAsync((char*)h.d_xall + (size_t)r * capb, h.h_xrecv.data() + (size_t)r * capb, XHDR + (size_t)hdr[4 * r] * rec, hipMemcpyHostToDevice, s));
            MHIPCHK(c, hipStreamSynchronize(s));   // (the host vector is reused by the next exchange)
        }
        unpack(h.d_xall, capb, cap_small);
        return 0;
    }
    // ---- a rank had more records than the first block holds: blocks padded to the largest count
    const size_t bigb = XHDR + (size_t)maxc * rec;
    if ((rc = big_buffer((size_t)world * bigb))) return rc;
    if (c->rccl_comm) {
        if ((rc = rccl_allgather_bytes(c, h.d_xsend, h.d_xbig, bigb, s, &h.err))) return rc;
        h.xcalls++;
    } else {
        h.h_xsend.assign(bigb, 0);
        MHIPCHK(c, hipMemcpy(h.h_xsend.data(), h.d_xsend, XHDR + (size_t)hdr[4 * c->cfg.shard_rank] * rec, hipMemcpyDeviceToHost));
        h.h_xrecv.resize((size_t)world * bigb);
        if (h.allgather(h.h_xsend.data(), (int64_t)bigb, h.h_xrecv.data(), h.allgather_user)) { h.err = "all-gather callback failed"; return IMMESH_E_INVAL; }
        h.xcalls++;
        for (int r = 0; r < world; r++)
            MHIPCHK(c, hipMemcpyAsync((char*)h.d_xbig + (size_t)r * bigb, h.h_xrecv.data() + (size_t)r * bigb, XHDR + (size_t)hdr[4 * r] * rec, hipMemcpyHostToDevice, s));
        MHIPCHK(c, hipStreamSynchronize(s));
    }
    unpack(h.d_xbig, bigb, (int)maxc);
    return 0;
}

static int mesh_graph_run(immesh_ctx* c, hipGraphExec_t& exec, hipStream_t s, const std::function<int()>& enqueue, bool launch = true) {
    if (exec == nullptr) {
        hipGraph_t g = nullptr;
        MHIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int erc = enqueue();
        const hipError_t ce = hipStreamEndCapture(s, &g);
        if (erc || ce != hipSuccess || g == nullptr) { if (g) (void)hipGraphDestroy(g); c->mesh_host.err = "hipGraph capture of the mesher failed"; return IMMESH_E_HIP; }
        const hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) { exec = nullptr; c->mesh_host.err = std::string("hipGraphInstantiate: ") + hipGetErrorString(ie); return IMMESH_E_HIP; }
    }
    if (launch) MHIPCHK(c, hipGraphLaunch(exec, s));   // (launch == false: capture + instantiate only -- the other variant of phase B, ahead of its first use)
    return 0;
}

// Worker thread: enqueue one scan (phase A on h.stream, phase B on h.stream_b behind it).  Returns without waiting unless the scan is
// offline-sized.  `synced` tells the caller that both streams are already drained.
static int mesh_scan_launch(immesh_ctx* c, const MeshJob& job, bool& synced, bool deep) {
    MeshHost& h = c->mesh_host;
    const int par = (int)(job.id % MESH_NPAR);
    const MeshDev& m = h.mpar[par];
    hipStream_t sa = h.stream, sb = h.stream_b;
    const float* d_pts = job.d_pts;
    const int n_raw = job.n_raw;
    synced = false;
    h.seq++;
    MeshScanParams sp;
    sp.cam[0] = job.cam[0]; sp.cam[1] = job.cam[1]; sp.cam[2] = job.cam[2];
    sp.n_raw = n_raw;
    sp.step = std::max(1, (int)std::round((double)(n_raw / c->cfg.mesh_append_budget)));  // integer division first (ImMesh_mesh_reconstruction.cpp:111)
    sp.n_cand = (n_raw + sp.step - 1) / sp.step;
    sp.vtx_base = 0;   // filled in on the device (mesh_begin_scan_kernel)
    if (sp.n_cand > m.cap_cand) { h.err = "scan larger than cap_scan_points"; return IMMESH_E_CAPACITY; }
    // Everything the HOST sizes by the candidate count -- grids, the candidate hash's capacity, which admission tail runs, the graphs' key -- takes the count
    // rounded up to 2048 (the kernels take the real count from MeshDyn and bound themselves by it; a larger grid or hash costs idle workgroups / a few more
    // cleared entries).  A real sensor's scans differ in size from one to the next (rays without a return): keyed by the exact count, the phase graphs of a
    // job set were captured and instantiated again for EVERY job -- milliseconds of host time each -- on anything but a constant-size synthetic stream
    // (configs[3]: 1 450 scans/s with three graphs per set, 2 200 with round 5's two).  16 384 and 65 536, where the paths change, are multiples of 2048.
    const int n_key = (int)std::min<int64_t>(((int64_t)sp.n_cand + 2047) / 2048 * 2048, (int64_t)m.cap_cand);
    const int64_t ccap = np2((int64_t)n_key * 4);
    h.h_dyn[par]->sp = sp; h.h_dyn[par]->seq = h.seq; h.h_dyn[par]->ch_mask = (uint64_t)ccap - 1;
    h.h_dyn[par]->wait_flag = job.wait_flag; h.h_dyn[par]->wait_seq = job.wait_seq;
    h.h_dyn[par]->pts = d_pts;
    bool is_world = false;
    for (int k = 0; k < MESH_WORLD_BUFS; k++) is_world = is_world || d_pts == h.d_world[k];
    // the scan (transform / host copy) was produced on the registration stream: an event, or -- immesh_process_scan's fused path -- a flag the first kernel polls
    if (!job.wait_flag) MHIPCHK(c, hipStreamWaitEvent(sa, job.ready, 0));
    int rc = 0;
    if (m.shard_world > 1) {
        // ---- sharded mesher.  Admission: the owner of a candidate's mesh-voxel brick tests it against the map and decides it; the ranks
        // exchange only the boundary band (band survivors, then decisions: mesh_cand_pack_kernel) in rounds until nobody has an undecided candidate
        // left -- two exchanges when no dependency chain crosses a brick face twice.  Then every rank commits the same new vertices (ids = the serial
        // ones), searches / triangulates its own voxels, and the band of smoothed positions and triangle marks travels (mesh_pack_*_kernel).
        launch_mesh_begin_scan(sa, m, h.h_dyn_dev[par], (unsigned long long)ccap);
        launch_mesh_append_prepare(sa, m, sp.n_cand, d_pts);
        char* const xs = (char*)h.d_xsend;
        for (int round = 0;; round++) {
            if (round > 4096) { h.err = "sharded admission did not converge"; return IMMESH_E_HIP; }
            MHIPCHK(c, hipMemsetAsync(xs, 0, XHDR, sa));
            launch_mesh_cand_pack(sa, m, (MeshCdRec*)(xs + XHDR), (int32_t*)xs, (int)((h.xcap_bytes - XHDR) / sizeof(MeshCdRec)));
            int64_t undecided = 0;
            if ((rc = mesh_exchange(c, sa, sizeof(MeshCdRec), XCAP_CAND, [&](const void* d, size_t capb, int cr) { launch_mesh_cand_unpack(sa, m, d, capb, cr); }, &undecided))) return rc;
            h.x_rounds++;
            if (undecided == 0) break;
            MHIPCHK(c, hipMemsetAsync(m.sc + SC_UNDECIDED, 0, 4, sa));
            launch_mesh_append_resolve(sa, m, sp.n_cand, d_pts, 4096);
        }
        if ((rc = mesh_enqueue_a(c, m, par, sa, d_pts, sp.n_cand, ccap, false))) return rc;
        // exchange 1: this scan's smoothed positions (what correct_triangle_index reads across voxels); triangulate own voxels; exchange 2: triangle marks
        MHIPCHK(c, hipMemsetAsync(xs, 0, XHDR, sa));
        launch_mesh_pack_smooth(sa, m, (MeshSmRec*)(xs + XHDR), (int32_t*)xs, (int)((h.xcap_bytes - XHDR) / sizeof(MeshSmRec)));
        if ((rc = mesh_exchange(c, sa, sizeof(MeshSmRec), XCAP_BAND, [&](const void* d, size_t capb, int cr) { launch_mesh_unpack_smooth(sa, m, d, capb, cr); }))) return rc;
        if ((rc = mesh_enqueue_b(c, m, par, sa, 1))) return rc;
        MHIPCHK(c, hipMemsetAsync(xs, 0, XHDR, sa));
        launch_mesh_pack_marks(sa, m, (MeshMkRec*)(xs + XHDR), (int32_t*)xs, (int)((h.xcap_bytes - XHDR) / sizeof(MeshMkRec)));
        if ((rc = mesh_exchange(c, sa, sizeof(MeshMkRec), XCAP_BAND, [&](const void* d, size_t capb, int cr) { launch_mesh_unpack_marks(sa, m, d, capb, cr); }))) return rc;
        if ((rc = mesh_enqueue_b(c, m, par, sa, 2))) return rc;
        MHIPCHK(c, hipEventRecord(h.ev_b[par], sa));
        return 0;
    }
    if (n_key > 65536) {
        // offline-sized clouds: the admission kernel's blocks are no longer all resident -> bounded rounds with a host check in between
        launch_mesh_begin_scan(sa, m, h.h_dyn_dev[par], (unsigned long long)ccap);
        launch_mesh_append_prepare(sa, m, sp.n_cand, d_pts);
        for (int round = 0; round < 100000; round++) {
            if (round > 0) MHIPCHK(c, hipMemsetAsync(m.sc + SC_UNDECIDED, 0, 4, sa));
            launch_mesh_append_resolve(sa, m, sp.n_cand, d_pts, 64);
            MHIPCHK(c, hipMemcpyAsync(h.h_sc2[par], m.sc, SC_COUNT * 4, hipMemcpyDeviceToHost, sa));
            MHIPCHK(c, hipStreamSynchronize(sa));
            if (h.h_sc2[par][SC_UNDECIDED] == 0) break;
        }
        if ((rc = mesh_enqueue_a(c, m, par, sa, d_pts, sp.n_cand, ccap, false))) return rc;
    } else if (h.use_graph && !h.prof.on && is_world) {
        // steady state: the launches of a phase are captured once per (parity, candidate count) and replayed as one hipGraph
        if (h.graph_ncand[par] != n_key) {
            for (hipGraphExec_t* e : {&h.graph_exec[par], &h.graph_exec_b[par][0], &h.graph_exec_b[par][1]}) if (*e) { (void)hipGraphExecDestroy(*e); *e = nullptr; }
            h.graph_ncand[par] = n_key;
        }
        // (captured with a null scan pointer: the kernels take it from MeshDyn, so one graph serves every world buffer)
        if ((rc = mesh_graph_run(c, h.graph_exec[par], sa, [&] { return mesh_enqueue_a(c, m, par, sa, nullptr, n_key, ccap, true); }))) return rc;
    } else {
        if ((rc = mesh_enqueue_a(c, m, par, sa, d_pts, n_key, ccap, true))) return rc;
    }
    // (an event between the phases -- a poll at the head of phase B would hold LDS phase A's single-workgroup launch needs: measured deadlock --
    //  but none behind phase B: mesh_publish_kernel's ticket in pinned memory / the worker's poll)
    MHIPCHK(c, hipEventRecord(h.ev_a[par], sa));
    // Round 6: the triangulations of the job (80 % of what used to be phase B's first launch) need phase A's results and nothing of the previous job's
    // phase B: they run on a third stream (the fetch / query stream -- no HSA queue of its own, see mesh_alloc) and only the diff against the live
    // triangle set waits for the previous commit.  Phase B + its queue gap WAS the pipeline's period (profiles/r06_marks_*.txt).
    // WHEN: the third stream (and a third job in flight) buy throughput -- phase B's head shrinks from ~100 to ~20 us -- at the price of company for the pose
    // chain (~3 us per scan) and ~15 us of job latency (two cross-stream events).  That pays while the mesher is what the pipeline waits for and costs
    // 1.7 % where it is not (the driver's 20-scan run on a young map: the pose chain is slower than even the one-launch phase B).  The worker therefore
    // switches: `deep` = a job found another one queued behind it within the last 64 jobs (mesh_worker_main).  IMMESH_SPLIT = 1 / 0: always / never.
    const bool split = n_key <= 65536 && (h.split_mode == 1 || (h.split_mode == 2 && deep));
    if (split) {
        static const int which = [] { const char* e = getenv("IMMESH_TRI_STREAM"); return e ? atoi(e) : 1; }();   // (measurement knob: 0 fetch stream, 1 pre-processing stream (default), 2 null stream)
        hipStream_t st = which == 1 ? c->stream_pre : (which == 2 ? (hipStream_t)nullptr : h.stream_fetch);
        MHIPCHK(c, hipStreamWaitEvent(st, h.ev_a[par], 0));
        launch_mesh_tri64(st, m);
        MHIPCHK(c, hipEventRecord(h.ev_c[par], st));
        MHIPCHK(c, hipStreamWaitEvent(sb, h.ev_c[par], 0));
    } else MHIPCHK(c, hipStreamWaitEvent(sb, h.ev_a[par], 0));
    if (h.use_graph && !h.prof.on && n_key <= 65536 && is_world) {
        // (both variants of phase B are captured when the first of them is needed: instantiating a graph takes a millisecond, and the switch to the other
        //  arrangement must not pay it in the middle of a stream)
        if (h.graph_exec_b[par][split ? 0 : 1] == nullptr && h.split_mode == 2 &&
            (rc = mesh_graph_run(c, h.graph_exec_b[par][split ? 0 : 1], sb, [&] { return mesh_enqueue_b(c, m, par, sb, 0, !split); }, false))) return rc;
        if ((rc = mesh_graph_run(c, h.graph_exec_b[par][split ? 1 : 0], sb, [&] { return mesh_enqueue_b(c, m, par, sb, 0, split); }))) return rc;
    } else {
        if ((rc = mesh_enqueue_b(c, m, par, sb, 0, split))) return rc;
    }
    return 0;
}

// Worker thread, after ev_b of the job's parity has completed: sizes, cumulative counters, capacity / hang checks.
static int mesh_scan_finish(immesh_ctx* c, const MeshJob& job, immesh_mesh_sizes_t& sizes) {
    MeshHost& h = c->mesh_host;
    const int par = (int)(job.id % MESH_NPAR);
    const MeshDev& m = h.mpar[par];
    h.h_sc = h.h_sc2[par];
    int rc = mesh_overflow(c);
    if (rc) return rc;
    if (h.h_sc[SC_UNDECIDED] != 0) { h.err = "vertex admission did not converge (device hang guard)"; return IMMESH_E_HIP; }
    const int n_cand = h.h_dyn[par]->sp.n_cand;
    const int n_new = h.h_sc[SC_ACCEPTED], n_active = std::min(h.h_sc[SC_ACTIVE], (int)m.cap_active);
    const int n_add = h.h_sc[SC_ADD], n_rem = h.h_sc[SC_REM], n_upd = h.h_sc[SC_UPD], n_smooth = h.h_sc[SC_SMOOTH];
    sizes.vtx_base = h.h_sc[SC_VTXBASE]; sizes.n_new_vtx = n_new; sizes.n_voxels_meshed = n_active;
    sizes.n_add = n_add; sizes.n_rem = n_rem; sizes.n_upd = n_upd; sizes.n_smooth = n_smooth; sizes.reserved = 0;
    if (m.shard_world > 1) {   // the lists this rank REPORTS (triangles whose smallest vertex lies in its bricks); what it committed also covers its halo
        sizes.n_add = h.h_sc[SC_ADD_OWN]; sizes.n_rem = h.h_sc[SC_REM_OWN]; sizes.n_upd = h.h_sc[SC_UPD_OWN];
    }
    h.fin_state[0] = n_add; h.fin_state[1] = n_rem; h.fin_state[2] = n_upd;
    if (m.shard_world > 1) h.xbytes_sent += h.h_sc[SC_XBYTES];
    h.n_vertices = sizes.vtx_base + n_new;
    h.cum[SC_ACCEPTED] += n_new; h.cum[SC_ACTIVE] += n_active; h.cum[SC_C1] += h.h_sc[SC_C1];
    h.cum[SC_RECENT] += n_cand;  // n_app: candidates offered
    h.cum[SC_ADD] += sizes.n_add; h.cum[SC_REM] += sizes.n_rem; h.cum[SC_C20] += h.h_sc[SC_C20]; h.cum[SC_NV] += h.h_sc[SC_NV];
    h.cum[SC_NU] += h.h_sc[SC_NU]; h.cum[SC_TV] += h.h_sc[SC_TV]; h.cum[SC_DEGEN] += h.h_sc[SC_DEGEN];
    h.n_live += sizes.n_add - sizes.n_rem;   // (sharded: the triangles this rank reports -- the ranks' counts add up to the serial one)
    h.cum[SC_MAXNU] = std::max<int64_t>(h.cum[SC_MAXNU], h.h_sc[SC_MAXNU]); h.cum[SC_PASS2] += h.h_sc[SC_PASS2];
    if (m.dbg) {
        unsigned long long t[64];
        (void)hipMemcpy(t, m.dbg, 512, hipMemcpyDeviceToHost); (void)hipMemset(m.dbg, 0, 512);
        {
            const unsigned long long nw = std::max(1ull, t[39]);
            fprintf(stderr, "[append_prepare cycles/wavefront] until candidate %llu | index %llu voxel %llu own+leader %llu 1-NN batches %llu chain %llu | slowest workgroup %llu | wavefront rounds %llu + %llu overflow\n",
                    t[37] / std::max(1ull, t[40] + t[41]), t[32] / nw, t[33] / nw, t[34] / nw, t[35] / nw, t[36] / nw, t[38], t[40], t[41]);
        }
        fprintf(stderr, "[delaunay64 cycles/voxel] cavity %llu edges %llu extras %llu inplace %llu filter+emit %llu | sums %llu jacobi %llu proj %llu | points %llu bails %llu\n", t[16] / std::max(1, n_active),
                t[17] / std::max(1, n_active), t[18] / std::max(1, n_active), t[19] / std::max(1, n_active), t[20] / std::max(1, n_active), t[24] / std::max(1, n_active), t[25] / std::max(1, n_active),
                t[26] / std::max(1, n_active), t[15], t[7]);
        fprintf(stderr, "[slowest voxel] knn %llu cycles (nq %llu)  delaunay %llu cycles (n_u %llu)\n", t[13] >> 16, t[13] & 0xFFFF, t[14] >> 16, t[14] & 0xFFFF);
        fprintf(stderr, "[knn cycles/voxel] stage0 %llu query0 %llu stage1 %llu query1 %llu final %llu\n", t[8] / std::max(1, n_active), t[9] / std::max(1, n_active), t[10] / std::max(1, n_active),
                t[11] / std::max(1, n_active), t[12] / std::max(1, n_active));
        fprintf(stderr, "[delaunay cycles/voxel] load %llu pca+proj %llu sort %llu insert %llu filter %llu oldset %llu adds %llu\n", t[0] / std::max(1, n_active), t[1] / std::max(1, n_active),
                t[2] / std::max(1, n_active), t[3] / std::max(1, n_active), t[4] / std::max(1, n_active), t[5] / std::max(1, n_active), t[6] / std::max(1, n_active));
    }
    if (getenv("IMMESH_DEBUG")) fprintf(stderr, "[mesh] cand %d new %d active %d maxnu %d pass2 %d add %d rem %d\n", n_cand, n_new, n_active, h.h_sc[SC_MAXNU], h.h_sc[SC_PASS2], n_add, n_rem);
    return 0;
}

// The worker keeps up to two scans in flight on the device: as soon as the next job is queued it is enqueued behind the running one (phase A
// of scan k+1 overlaps phase B of scan k), and jobs are finished -- counters read, results published -- strictly in order.
static void mesh_worker_main(immesh_ctx* c) {
    MeshHost& h = c->mesh_host;
    (void)hipSetDevice(c->cfg.device);
    g_kprof = &h.prof;
    struct Flight { MeshJob job; MeshResult r; bool launched_ok; int seq; bool by_ticket; unsigned polls; };
    std::deque<Flight> fl;
    long deep_until = 0;   // (job id) the deep arrangement stays on for jobs below it
    bool deep = false;
    int backlog_streak = 0;
    for (;;) {
        // ---- take a new job when one is queued and the pipeline has room
        bool have = false;
        MeshJob job;
        {
            std::unique_lock<std::mutex> lk(h.mu);
            if (fl.empty()) h.cv_job.wait(lk, [&] { return h.stop || !h.q.empty(); });
            const bool deep_now = h.split_mode == 1 || (h.split_mode == 2 && !h.q.empty() && h.q.front().id < deep_until);
            const size_t room = (h.pipeline && !h.prof.on && c->mesh.shard_world <= 1) ? (h.room ? (size_t)h.room : (deep_now ? (size_t)MESH_NPAR : (size_t)2)) : 1;
            if (!h.q.empty() && fl.size() < room) {
                job = h.q.front(); h.q.pop_front(); have = true;
                // the mesher is behind when a job leaves the queue and the next one is already waiting: go deep (three jobs in flight, triangulations on
                // the third stream) for the next 64 jobs -- long enough not to flutter, short enough to fall back when the stream slows down
                backlog_streak = h.q.empty() ? 0 : backlog_streak + 1;
                if (backlog_streak >= 3) deep_until = job.id + 64;   // (three jobs in a row: a lag, not the hiccup of a graph capture or a first touch)
                deep = h.split_mode == 1 || (h.split_mode == 2 && job.id < deep_until);
            }
            else if (fl.empty() && h.q.empty()) break;   // stop requested and nothing left to do
        }
        if (have) {
            Flight f;
            f.job = job; f.r.id = job.id; f.r.d_pts = job.d_pts; f.r.n_raw = job.n_raw; f.launched_ok = false;
            std::memset(&f.r.sizes, 0, sizeof(f.r.sizes));
            h.err.clear();
            bool synced = false;
            { std::lock_guard<std::mutex> lq(h.launch_mu); f.r.rc = mesh_scan_launch(c, job, synced, deep); }   // (a smooth_pts query runs between two jobs, never beside one)
            f.seq = h.seq; f.polls = 0;
            f.by_ticket = c->mesh.shard_world <= 1;   // (the sharded mesher runs on one stream and keeps its event)
            if (f.r.rc) f.r.err = h.err; else f.launched_ok = true;
            fl.push_back(f);
            continue;
        }
        // ---- finish the oldest scan in flight once its phase B is done; meanwhile keep an eye on the queue
        Flight& f = fl.front();
        const int par = (int)(f.job.id % MESH_NPAR);
        if (f.launched_ok) {
            hipError_t q;
            if (f.by_ticket) {
                // mesh_publish_kernel's ticket in pinned memory; the stream is consulted now and then so that a faulted launch cannot leave the worker polling
                q = (*(volatile int32_t*)(h.h_sc2[par] + MESH_PUB_SEQ) == f.seq) ? hipSuccess : hipErrorNotReady;
                if (q == hipErrorNotReady && (++f.polls & 0xFFF) == 0) {
                    const hipError_t qs = hipStreamQuery(h.stream_b);
                    if (qs != hipErrorNotReady && *(volatile int32_t*)(h.h_sc2[par] + MESH_PUB_SEQ) != f.seq) q = (qs == hipSuccess) ? hipErrorUnknown : qs;
                }
                if (q == hipSuccess) std::atomic_thread_fence(std::memory_order_acquire);
            } else q = hipEventQuery(h.ev_b[par]);
            if (q == hipErrorNotReady) {
                bool more;
                { std::lock_guard<std::mutex> lk(h.mu); more = !h.q.empty() && fl.size() < ((h.pipeline && !h.prof.on && c->mesh.shard_world <= 1) ? (h.room ? (size_t)h.room : ((h.split_mode == 1 || h.q.front().id < deep_until) ? (size_t)MESH_NPAR : (size_t)2)) : (size_t)1); }
                if (!more) std::this_thread::yield();
                continue;
            }
            h.err.clear();
            if (q != hipSuccess) { f.r.rc = IMMESH_E_HIP; f.r.err = std::string("mesh job: ") + hipGetErrorString(q); }
            else {
                f.r.ms = (float)((double)*(volatile unsigned long long*)(h.h_sc2[par] + MESH_PUB_TICKS) * 1e-5);   // (100 MHz ticks of the device's real-time counter)
                h.job_ms_sum += f.r.ms; h.job_ms_n++;
                for (int k = 0; k < MESH_N_MARKS; k++) h.mark_ring[h.mark_n & 63][k] = *(volatile unsigned long long*)(h.h_sc2[par] + MESH_PUB_MARKS + 2 * k);
                h.mark_n++;
                f.r.rc = mesh_scan_finish(c, f.job, f.r.sizes);
                f.r.st_add = h.fin_state[0]; f.r.st_rem = h.fin_state[1]; f.r.st_upd = h.fin_state[2];
                if (f.r.rc) f.r.err = h.err;
            }
        } else {
            (void)hipStreamSynchronize(h.stream); (void)hipStreamSynchronize(h.stream_b);
        }
        if (h.prof.on) h.prof.flush();
        if (f.r.rc) {
            // a failed job may have stopped between the admission's binning and the kernel that hands the bucket counters back zeroed (mesh_knn_kernel):
            // the next scan must not find them filled
            (void)hipStreamSynchronize(h.stream); (void)hipStreamSynchronize(h.stream_b);
            (void)hipMemsetAsync(c->mesh.bin_cnt, 0, 2 * (1024 + 1) * sizeof(int32_t), h.stream);
        }
        {
            std::lock_guard<std::mutex> lk(h.mu);
            h.res[f.job.id % MESH_NPAR] = f.r;
            h.completed = f.job.id;
        }
        h.cv_done.notify_all();
        fl.pop_front();
    }
    g_kprof = nullptr;
}

// Called on the scan thread.  d_pts = world-frame xyzI already (being) produced on c->stream; returns the job id.
// The worker keeps two jobs in flight; their scans and those of the jobs queued behind them live in the MESH_WORLD_BUFS world buffers.
hipEvent_t mesh_record_ready(immesh_ctx* c) {
    MeshHost& h = c->mesh_host;
    long next;
    { std::unique_lock<std::mutex> lk(h.mu); next = h.submitted + 1; }
    (void)hipEventRecord(h.ev_ready[next % MESH_WORLD_BUFS], c->stream);
    return h.ev_ready[next % MESH_WORLD_BUFS];
}
long mesh_submit(immesh_ctx* c, const float* d_pts, int n_raw, const double* sensor_pos, int frame_idx, bool ready_recorded, const unsigned long long* wait_flag, unsigned long long wait_seq) {
    MeshHost& h = c->mesh_host;
    MeshJob job;
    {
        std::unique_lock<std::mutex> lk(h.mu);
        job.id = h.submitted + 1;
    }
    job.d_pts = d_pts; job.n_raw = n_raw; job.frame_idx = frame_idx;
    job.cam[0] = sensor_pos[0]; job.cam[1] = sensor_pos[1]; job.cam[2] = sensor_pos[2];
    job.ready = h.ev_ready[job.id % MESH_WORLD_BUFS];
    job.wait_flag = wait_flag; job.wait_seq = wait_seq;
    if (!ready_recorded && !wait_flag) (void)hipEventRecord(job.ready, c->stream);
    {
        std::lock_guard<std::mutex> lk(h.mu);
        h.q.push_back(job);
        h.submitted = job.id;
    }
    h.cv_job.notify_one();
    return job.id;
}
// world buffer the NEXT job will use; blocks until the job that last used it (next id - MESH_WORLD_BUFS) has finished
float* mesh_next_world_buffer(immesh_ctx* c) {
    MeshHost& h = c->mesh_host;
    std::unique_lock<std::mutex> lk(h.mu);
    const long next = h.submitted + 1;
    // IMMESH_DEBUG_WAITS: how long the scan thread stands here = how far the mesher is behind the pose chain (printed by mesh_free)
    static const bool dbg_waits = getenv("IMMESH_DEBUG_WAITS") != nullptr;
    const auto t0 = dbg_waits ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    h.cv_done.wait(lk, [&] { return h.completed >= next - MESH_WORLD_BUFS && (!h.collect_on || h.collected >= next - 2); });   // (result lists: two sets, job parity)
    if (dbg_waits) { const long long w = (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); h.wait_ns += w; h.wait_ring[h.wait_calls & 63] = w; h.wait_calls++; }
    return h.d_world[next % MESH_WORLD_BUFS];
}
// wait for job `id` (0 = the newest submitted) and make it the one immesh_mesh_sizes / fetch / last_timing report
int mesh_wait(immesh_ctx* c, long id) {
    MeshHost& h = c->mesh_host;
    std::unique_lock<std::mutex> lk(h.mu);
    if (id <= 0) id = h.submitted;
    if (id <= 0) return 0;
    if (id < h.submitted - 1) { c->err = "mesh job results already overwritten (only the two newest jobs are kept)"; return IMMESH_E_INVAL; }
    h.cv_done.wait(lk, [&] { return h.completed >= id; });
    h.current = id;
    const MeshResult& r = h.res[id % MESH_NPAR];
    c->timing[3] = r.ms;
    if (r.rc) c->err = r.err;
    return r.rc;
}
void mesh_wait_all(immesh_ctx* c) {
    MeshHost& h = c->mesh_host;
    std::unique_lock<std::mutex> lk(h.mu);
    h.cv_done.wait(lk, [&] { return h.completed >= h.submitted; });
}

int mesh_transform_full(immesh_ctx* c, const float* d_raw, float* d_world, int n_raw, const imh::State& st) {
    launch_mesh_transform(c->stream, d_raw, d_world, n_raw, st.R, st.t, c->cfg.extR, c->cfg.extT);
    return 0;
}

void mesh_counters(immesh_ctx* c, immesh_counters_t* out) {
    mesh_wait_all(c);
    const MeshHost& h = c->mesh_host;
    out->n_app = h.cum[SC_RECENT]; out->n_new = h.cum[SC_ACCEPTED]; out->v_act = h.cum[SC_ACTIVE]; out->n_v = h.cum[SC_NV]; out->n_u = h.cum[SC_NU];
    out->t_v = h.cum[SC_TV]; out->t_add = h.cum[SC_ADD]; out->t_rem = h.cum[SC_REM]; out->c1 = h.cum[SC_C1]; out->c20 = h.cum[SC_C20]; out->n_degenerate_skips = h.cum[SC_DEGEN];
    out->n_vertices = h.n_vertices; out->n_triangles_live = h.n_live;
}
void mesh_counters_reset(immesh_ctx* c) { mesh_wait_all(c); std::memset(c->mesh_host.cum, 0, sizeof(c->mesh_host.cum)); }

extern "C" {

int immesh_mesh_scan(immesh_ctx* c, const float* pts_world_xyzi, int32_t n_raw, const double* sensor_pos, int32_t frame_idx) {
    if (!c || !pts_world_xyzi || n_raw <= 0 || n_raw > c->cap_scan || !sensor_pos) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    float* world = mesh_next_world_buffer(c);
    const void* d_pts;
    int rc = resolve_input(c, pts_world_xyzi, (size_t)n_raw * 16, world, &d_pts);
    if (rc) return rc;
    const long id = mesh_submit(c, (const float*)d_pts, n_raw, sensor_pos, frame_idx);
    return mesh_wait(c, id);   // synchronous entry point: results are current when it returns
}

int immesh_set_allgather(immesh_ctx* c, immesh_allgather_fn cb, void* user) {
    if (!c) return IMMESH_E_INVAL;
    mesh_wait_all(c);
    c->mesh_host.allgather = cb; c->mesh_host.allgather_user = user;
    return 0;
}
int immesh_shard_traffic(immesh_ctx* c, int64_t* bytes, int64_t* calls) {
    if (!c) return IMMESH_E_INVAL;
    mesh_wait_all(c);
    if (bytes) *bytes = c->mesh_host.xbytes_sent;
    if (calls) *calls = c->mesh_host.xcalls;
    return 0;
}

int immesh_mesh_wait(immesh_ctx* c) {
    if (!c) return IMMESH_E_INVAL;
    return mesh_wait(c, 0);
}

// ---- the service-thread side of asynchronous meshing (service_reconstruct_mesh, ImMesh_mesh_reconstruction.cpp:272-310) ------------------------------
int immesh_mesh_collect_enable(immesh_ctx* c, int32_t on) {
    if (!c) return IMMESH_E_INVAL;
    MeshHost& h = c->mesh_host;
    {
        std::lock_guard<std::mutex> lk(h.mu);
        h.collect_on = on != 0;
        h.collected = h.completed;   // what has finished before this call is not handed out
    }
    h.cv_done.notify_all();
    return 0;
}
int immesh_mesh_collect_begin(immesh_ctx* c, int32_t timeout_ms, int64_t* job_ordinal) {
    if (!c) return IMMESH_E_INVAL;
    MeshHost& h = c->mesh_host;
    std::unique_lock<std::mutex> lk(h.mu);
    if (!h.collect_on) { return IMMESH_E_INVAL; }
    const long id = h.collected + 1;
    if (!h.cv_done.wait_for(lk, std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms), [&] { return h.completed >= id; })) return IMMESH_NOT_READY;
    h.current = id;
    if (job_ordinal) *job_ordinal = id;
    return h.res[id % MESH_NPAR].rc;
}
int immesh_mesh_collect_end(immesh_ctx* c) {
    if (!c) return IMMESH_E_INVAL;
    MeshHost& h = c->mesh_host;
    {
        std::lock_guard<std::mutex> lk(h.mu);
        if (!h.collect_on || h.current != h.collected + 1) return IMMESH_E_INVAL;
        h.collected = h.current;
    }
    h.cv_done.notify_all();
    return 0;
}

int immesh_mesh_sizes(immesh_ctx* c, immesh_mesh_sizes_t* sizes) {
    if (!c || !sizes) return IMMESH_E_INVAL;
    MeshHost& h = c->mesh_host;
    std::lock_guard<std::mutex> lk(h.mu);
    if (h.current <= 0) { std::memset(sizes, 0, sizeof(*sizes)); return 0; }
    *sizes = h.res[h.current % MESH_NPAR].sizes;
    return 0;
}

// diagnostics (bench.py's density leg): the 20-NN neighbourhood size n_u of every voxel the newest finished job triangulated, in active order
int immesh_mesh_neighbourhood_sizes(immesh_ctx* c, int32_t* out, int32_t cap, int32_t* n_out) {
    if (!c || !n_out) return IMMESH_E_INVAL;
    (void)hipSetDevice(c->cfg.device);
    MeshHost& h = c->mesh_host;
    int par, n;
    {
        std::lock_guard<std::mutex> lk(h.mu);
        if (h.current <= 0) { *n_out = 0; return 0; }
        if (h.current < h.submitted) { c->err = "a newer mesh job is in flight (immesh_mesh_wait first)"; return IMMESH_E_INVAL; }
        par = (int)(h.current % MESH_NPAR);
        n = h.res[par].sizes.n_voxels_meshed;
    }
    *n_out = n;
    if (out && n > 0) {
        if (n > cap) { c->err = "output buffer too small"; return IMMESH_E_CAPACITY; }
        HIPCHK(c, hipMemcpy(out, h.mpar[par].rel_n, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    return 0;
}

// diagnostics (parity tests): the world-frame scan of the newest finished job, straight from its world buffer
int immesh_mesh_world_scan(immesh_ctx* c, float* out_xyzi, int32_t cap_pts, int32_t* n_out) {
    if (!c || !n_out) return IMMESH_E_INVAL;
    (void)hipSetDevice(c->cfg.device);
    MeshHost& h = c->mesh_host;
    const float* d_pts; int n;
    {
        std::lock_guard<std::mutex> lk(h.mu);
        if (h.current <= 0) { *n_out = 0; return 0; }
        if (h.current < h.submitted - (MESH_WORLD_BUFS - 2)) { c->err = "the job's world buffer has been handed to a newer scan"; return IMMESH_E_INVAL; }
        d_pts = h.res[h.current % MESH_NPAR].d_pts; n = h.res[h.current % MESH_NPAR].n_raw;
    }
    *n_out = n;
    if (out_xyzi && n > 0) {
        if (n > cap_pts) { c->err = "output buffer too small"; return IMMESH_E_CAPACITY; }
        HIPCHK(c, hipMemcpy(out_xyzi, d_pts, (size_t)n * 16, hipMemcpyDeviceToHost));
    }
    return 0;
}

int immesh_mesh_fetch(immesh_ctx* c, float* new_vtx_xyz, int32_t* tri_add, uint8_t* flip_add, int32_t* tri_rem, int32_t* tri_upd, uint8_t* flip_upd,
                      int32_t* smooth_ids, double* smooth_xyz) {
    if (!c) return IMMESH_E_INVAL;
    (void)hipSetDevice(c->cfg.device);
    MeshHost& h = c->mesh_host;
    immesh_mesh_sizes_t z;
    MeshOutSet o;
    {
        std::lock_guard<std::mutex> lk(h.mu);
        if (h.current <= 0) return 0;
        if (h.current < h.submitted - 1) { c->err = "mesh job results already overwritten (only the two newest jobs are kept)"; return IMMESH_E_INVAL; }
        z = h.res[h.current % MESH_NPAR].sizes;
        o = h.outs[h.current % MESH_NPAR];
    }
    const MeshDev& m = c->mesh;
    hipStream_t s = h.stream_fetch;   // (the job has finished: nothing to order against; the scan thread may be enqueueing on its own streams meanwhile)
    if (m.shard_world > 1) {
        // sharded mesher: the device lists are what this rank COMMITTED (own bricks + halo), flagged entry by entry; the caller gets the entries
        // this rank reports -- the union of the ranks' lists is the serial list
        int st[3];
        { std::lock_guard<std::mutex> lk(h.mu); const MeshResult& r = h.res[h.current % MESH_NPAR]; st[0] = r.st_add; st[1] = r.st_rem; st[2] = r.st_upd; }
        const int32_t* d_tri[3] = {o.tri_add, o.tri_rem, o.tri_upd};
        const uint8_t* d_flip[3] = {o.flip_add, nullptr, o.flip_upd};
        const uint8_t* d_own[3] = {o.own_add, o.own_rem, o.own_upd};
        int32_t* out_tri[3] = {tri_add, tri_rem, tri_upd};
        uint8_t* out_flip[3] = {flip_add, nullptr, flip_upd};
        const int want[3] = {z.n_add, z.n_rem, z.n_upd};
        if (new_vtx_xyz && z.n_new_vtx) HIPCHK(c, hipMemcpyAsync(new_vtx_xyz, m.v_pos + (size_t)z.vtx_base * 3, (size_t)z.n_new_vtx * 12, hipMemcpyDeviceToHost, s));
        if (smooth_ids && z.n_smooth) HIPCHK(c, hipMemcpyAsync(smooth_ids, o.smooth_ids, (size_t)z.n_smooth * 4, hipMemcpyDeviceToHost, s));
        if (smooth_xyz && z.n_smooth) HIPCHK(c, hipMemcpyAsync(smooth_xyz, o.smooth_xyz, (size_t)z.n_smooth * 24, hipMemcpyDeviceToHost, s));
        std::vector<int32_t> tri; std::vector<uint8_t> flip, own;
        for (int j = 0; j < 3; j++) {
            if ((!out_tri[j] && !out_flip[j]) || st[j] <= 0) continue;
            tri.resize((size_t)st[j] * 3); own.resize((size_t)st[j]); flip.resize((size_t)st[j]);
            HIPCHK(c, hipMemcpyAsync(tri.data(), d_tri[j], (size_t)st[j] * 12, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(own.data(), d_own[j], (size_t)st[j], hipMemcpyDeviceToHost, s));
            if (d_flip[j]) HIPCHK(c, hipMemcpyAsync(flip.data(), d_flip[j], (size_t)st[j], hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            int k = 0;
            for (int e = 0; e < st[j]; e++) {
                if (!own[(size_t)e]) continue;
                if (k >= want[j]) { c->err = "sharded mesh fetch: more reported entries than counted"; return IMMESH_E_HIP; }
                if (out_tri[j]) std::memcpy(out_tri[j] + (size_t)k * 3, &tri[(size_t)e * 3], 12);
                if (out_flip[j]) out_flip[j][k] = flip[(size_t)e];
                k++;
            }
            if (k != want[j]) { c->err = "sharded mesh fetch: fewer reported entries than counted"; return IMMESH_E_HIP; }
        }
        HIPCHK(c, hipStreamSynchronize(s));
        return 0;
    }
    // Eight lists, each a few KB to a few hundred KB, into the caller's (pageable) buffers: a device-to-pageable copy is staged and waited for by the runtime one
    // at a time (0.3-0.65 ms for the eight: the drop-in's service thread spent its frame on it).  They go to ONE pinned staging block instead -- eight DMA
    // copies in flight together, one wait -- and from there by memcpy (round 5).
    struct Part { void* dst; const void* src; size_t bytes; };
    const Part parts[8] = {{new_vtx_xyz, m.v_pos + (size_t)z.vtx_base * 3, (size_t)z.n_new_vtx * 12}, {tri_add, o.tri_add, (size_t)z.n_add * 12}, {flip_add, o.flip_add, (size_t)z.n_add},
                           {tri_rem, o.tri_rem, (size_t)z.n_rem * 12}, {tri_upd, o.tri_upd, (size_t)z.n_upd * 12}, {flip_upd, o.flip_upd, (size_t)z.n_upd},
                           {smooth_ids, o.smooth_ids, (size_t)z.n_smooth * 4}, {smooth_xyz, o.smooth_xyz, (size_t)z.n_smooth * 24}};
    size_t total = 0;
    for (const Part& q : parts) if (q.dst && q.bytes) total += (q.bytes + 63) & ~(size_t)63;
    if (total == 0) return 0;
    // The staging block is shared: fetch_mu is held from its (re)allocation to the last memcpy out of it -- two fetches on one context (a scan thread and a
    // service thread) take turns instead of overwriting each other's lists or freeing the block the other copies from.  Not h.mu: the
    // mesher's worker needs that one, and a hipHostFree under it would stall the pipeline.
    std::lock_guard<std::mutex> lf(h.fetch_mu);
    if (total > h.h_fetch_bytes) {
        if (h.h_fetch) (void)hipHostFree(h.h_fetch);
        h.h_fetch = nullptr; h.h_fetch_bytes = 0;
        const size_t want = total + total / 2 + (1 << 16);
        if (hipHostMalloc((void**)&h.h_fetch, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->err = "hipHostMalloc(fetch staging)"; return IMMESH_E_NOMEM; }
        h.h_fetch_bytes = want;
    }
    size_t off = 0;
    for (const Part& q : parts)
        if (q.dst && q.bytes) { HIPCHK(c, hipMemcpyAsync(h.h_fetch + off, q.src, q.bytes, hipMemcpyDeviceToHost, s)); off += (q.bytes + 63) & ~(size_t)63; }
    HIPCHK(c, hipStreamSynchronize(s));
    off = 0;
    for (const Part& q : parts)
        if (q.dst && q.bytes) { std::memcpy(q.dst, h.h_fetch + off, q.bytes); off += (q.bytes + 63) & ~(size_t)63; }
    return 0;
}

// ---- mesh export (save_to_ply_file, mesh_rec_geometry.cpp:71-131): the consumer after the path -----------------------------------------
static int grow(immesh_ctx* c, void** p, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *have = 0;
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) { c->err = std::string("hipMalloc(export): ") + hipGetErrorString(e); return IMMESH_E_NOMEM; }
    *have = need;
    return 0;
}

// Global_map::smooth_pts (pointcloud_rgbd.cpp:932-958) for a batch of vertex ids, and the vertex positions the renderer puts into its GL buffer
// (unparse_triangle_set_to_vector, mesh_rec_display.cpp:78-103: get_pos(1) after smoothing what is still unsmoothed).  Both may be called from a thread of
// their own while scans are being meshed: the query holds launch_mu from "both mesher streams idle" until its results are on the host, so it reads the map
// between two jobs.  The map is not modified (the reference's smooth_pts also stores its result in the point: the host mirror's business).
static int mesh_query_smooth(immesh_ctx* c, const int32_t* ids, int32_t n, double smooth_factor, int32_t knn, double max_dis, int display, double* out_d, float* out_f) {
    if (!c || n < 0 || (n > 0 && (!ids || (!out_d && !out_f)))) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    if (knn != MV_KNN) { c->err = "smooth_pts: only knn = 20 (the reference's g_ply_smooth_k) is supported"; return IMMESH_E_INVAL; }
    MeshHost& h = c->mesh_host;
    const MeshDev& m = c->mesh;
    if (max_dis <= 0) max_dis = m.voxel * 0.8;   // (pointcloud_rgbd.cpp:940-943)
    if (max_dis > m.accept * 2.0) { c->err = "smooth_pts: maximum_smooth_dis above 2.5 x the mesh voxel (the search radius of the device's 20-NN pull)"; return IMMESH_E_INVAL; }
    if (n == 0) return 0;
    (void)hipSetDevice(c->cfg.device);
    std::lock_guard<std::mutex> lq(h.launch_mu);
    MHIPCHK(c, hipStreamSynchronize(h.stream));
    MHIPCHK(c, hipStreamSynchronize(h.stream_b));
    hipStream_t s = h.stream_q;
    // device: ids | voxel of each id | voxel list;  host (pinned): the same + the results
    const size_t bytes_dev = (size_t)n * 12 + (size_t)n * 24 + 64, bytes_host = bytes_dev;
    if (h.q_dev_bytes < bytes_dev) {
        if (h.q_dev) (void)hipFree(h.q_dev);
        h.q_dev = nullptr; h.q_dev_bytes = 0;
        if (hipMalloc(&h.q_dev, bytes_dev * 2) != hipSuccess) { c->err = "hipMalloc(smooth_pts)"; return IMMESH_E_NOMEM; }
        h.q_dev_bytes = bytes_dev * 2;
    }
    if (h.q_host_bytes < bytes_host) {
        if (h.q_host) (void)hipHostFree(h.q_host);
        h.q_host = nullptr; h.q_host_bytes = 0;
        if (hipHostMalloc((void**)&h.q_host, bytes_host * 2, hipHostMallocDefault) != hipSuccess) { c->err = "hipHostMalloc(smooth_pts)"; return IMMESH_E_NOMEM; }
        h.q_host_bytes = bytes_host * 2;
    }
    int32_t pc[PC_COUNT];
    MHIPCHK(c, hipMemcpyAsync(pc, m.pc, sizeof(pc), hipMemcpyDeviceToHost, s));
    int32_t* d_ids = (int32_t*)h.q_dev; int32_t* d_vox = d_ids + n; int32_t* d_list = d_vox + n;
    double* d_out = (double*)(((uintptr_t)(d_list + n) + 15) & ~(uintptr_t)15);
    int32_t* h_ids = (int32_t*)h.q_host; int32_t* h_vox = h_ids + n; int32_t* h_list = h_vox + n;
    double* h_out = (double*)(((uintptr_t)(h_list + n) + 15) & ~(uintptr_t)15);
    std::memcpy(h_ids, ids, (size_t)n * 4);
    MHIPCHK(c, hipMemcpyAsync(d_ids, h_ids, (size_t)n * 4, hipMemcpyHostToDevice, s));
    MHIPCHK(c, hipStreamSynchronize(s));
    const int nv = pc[PC_VERTS];
    launch_mesh_query_voxels(s, m, d_ids, n, nv, display, d_vox);
    MHIPCHK(c, hipMemcpyAsync(h_vox, d_vox, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    MHIPCHK(c, hipStreamSynchronize(s));
    int n_list = 0;
    for (int i = 0; i < n; i++) {
        if (h_vox[i] == -1) { c->err = "smooth_pts: vertex id out of range"; return IMMESH_E_INVAL; }
        if (h_vox[i] >= 0) h_list[n_list++] = h_vox[i];
    }
    std::sort(h_list, h_list + n_list);
    n_list = (int)(std::unique(h_list, h_list + n_list) - h_list);
    if (n_list > 0) {
        const size_t need = (size_t)std::max(nv, 1) * 24;
        if (h.q_exp_bytes < need) {
            if (h.q_exp) (void)hipFree(h.q_exp);
            h.q_exp = nullptr; h.q_exp_bytes = 0;
            if (hipMalloc(&h.q_exp, need + need / 4) != hipSuccess) { c->err = "hipMalloc(smooth_pts)"; return IMMESH_E_NOMEM; }
            h.q_exp_bytes = need + need / 4;
        }
        MHIPCHK(c, hipMemcpyAsync(d_list, h_list, (size_t)n_list * 4, hipMemcpyHostToDevice, s));
        launch_mesh_query_smooth(s, m, d_list, n_list, smooth_factor, max_dis, (double*)h.q_exp);
    }
    launch_mesh_query_gather(s, m, d_ids, d_vox, n, (const double*)h.q_exp, display, d_out, (float*)d_out);
    MHIPCHK(c, hipMemcpyAsync(h_out, d_out, (size_t)n * (display ? 12 : 24), hipMemcpyDeviceToHost, s));
    MHIPCHK(c, hipStreamSynchronize(s));
    if (display) std::memcpy(out_f, h_out, (size_t)n * 12); else std::memcpy(out_d, h_out, (size_t)n * 24);
    return 0;
}
int immesh_smooth_pts(immesh_ctx* c, const int32_t* ids, int32_t n, double smooth_factor, int32_t knn, double maximum_smooth_dis, double* out_xyz) {
    return mesh_query_smooth(c, ids, n, smooth_factor, knn, maximum_smooth_dis, 0, out_xyz, nullptr);
}
int immesh_mesh_display_vertices(immesh_ctx* c, const int32_t* ids, int32_t n, double smooth_factor, int32_t knn, double maximum_smooth_dis, float* out_xyz) {
    return mesh_query_smooth(c, ids, n, smooth_factor, knn, maximum_smooth_dis, 1, nullptr, out_xyz);
}

int immesh_mesh_export(immesh_ctx* c, double smooth_factor, int32_t knn, int64_t* n_vtx_out, int64_t* n_faces_out) {
    if (!c) return IMMESH_E_INVAL;
    if (smooth_factor != 0.0 && knn != MV_KNN) { c->err = "mesh export: only knn = 20 (the reference's g_ply_smooth_k) is supported"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    mesh_wait_all(c);
    ProfBind _pb(c);
    MeshHost& h = c->mesh_host;
    const MeshDev& m = c->mesh;
    hipStream_t s = c->stream;
    const int64_t nv = h.n_vertices, nf = h.n_live;
    int rc;
    if ((rc = grow(c, &h.exp_vtx, &h.exp_vtx_bytes, (size_t)std::max<int64_t>(nv, 1) * 12))) return rc;
    const size_t per = (size_t)std::max<int64_t>(nf, 1);
    if ((rc = grow(c, &h.exp_work, &h.exp_work_bytes, per * (4 + 4 + 4 + 4 + 8 + 8 + 12) + 64))) return rc;
    int32_t* idx_a = (int32_t*)h.exp_work; int32_t* idx_b = idx_a + per; int32_t* idx_c = idx_b + per;
    uint32_t* k32a = (uint32_t*)(idx_c + per);
    unsigned long long* k64a = (unsigned long long*)(((uintptr_t)(k32a + per) + 15) & ~(uintptr_t)15);
    unsigned long long* k64b = k64a + per;
    int32_t* faces = (int32_t*)(k64b + per);
    const size_t tmp_need = std::max(sort_pairs_u64_temp_bytes((int)per), sort_pairs_u32_temp_bytes((int)per)) + 256;
    if ((rc = grow(c, &h.exp_tmp, &h.exp_tmp_bytes, tmp_need))) return rc;
    if (nv > 0) {
        if (smooth_factor != 0.0) launch_mesh_export_vertices(s, m, (float*)h.exp_vtx, smooth_factor);
        else HIPCHK(c, hipMemcpyAsync(h.exp_vtx, m.v_pos, (size_t)nv * 12, hipMemcpyDeviceToDevice, s));
    }
    if (nf > 0) {
        int32_t* cnt = m.sc + SC_COUNT - 1;   // spare per-scan counter slot (the mesher is idle)
        HIPCHK(c, hipMemsetAsync(cnt, 0, 4, s));
        launch_mesh_export_faces(s, m, idx_a, cnt);
        // deterministic face order: lexicographic by (v0, v1, v2) -- two stable radix passes
        launch_mesh_export_keys(s, m, idx_a, (int)nf, 0, k32a, nullptr);
        sort_pairs_u32(s, h.exp_tmp, h.exp_tmp_bytes, k32a, (uint32_t*)k64b, idx_a, idx_b, (int)nf, 32);
        launch_mesh_export_keys(s, m, idx_b, (int)nf, 1, nullptr, k64a);
        sort_pairs_u64(s, h.exp_tmp, h.exp_tmp_bytes, k64a, k64b, idx_b, idx_c, (int)nf);
        launch_mesh_export_wind(s, m, idx_c, (int)nf, faces);
    }
    HIPCHK(c, hipStreamSynchronize(s));
    h.exp_faces = faces; h.exp_nv = nv; h.exp_nf = nf;
    if (n_vtx_out) *n_vtx_out = nv;
    if (n_faces_out) *n_faces_out = nf;
    return 0;
}

int immesh_mesh_export_fetch(immesh_ctx* c, float* vtx_xyz, int32_t* faces) {
    if (!c) return IMMESH_E_INVAL;
    (void)hipSetDevice(c->cfg.device);
    MeshHost& h = c->mesh_host;
    if (vtx_xyz && h.exp_nv) HIPCHK(c, hipMemcpy(vtx_xyz, h.exp_vtx, (size_t)h.exp_nv * 12, hipMemcpyDeviceToHost));
    if (faces && h.exp_nf) HIPCHK(c, hipMemcpy(faces, h.exp_faces, (size_t)h.exp_nf * 12, hipMemcpyDeviceToHost));
    return 0;
}

// binary little-endian PLY with the element / property layout pcl::io::savePLYFileBinary writes for a PolygonMesh of PointXYZ
int immesh_save_ply(immesh_ctx* c, const char* path, double smooth_factor, int32_t knn) {
    if (!c || !path) return IMMESH_E_INVAL;
    int64_t nv = 0, nf = 0;
    int rc = immesh_mesh_export(c, smooth_factor, knn, &nv, &nf);
    if (rc) return rc;
    std::vector<float> v((size_t)nv * 3);
    std::vector<int32_t> f((size_t)nf * 3);
    if ((rc = immesh_mesh_export_fetch(c, v.data(), f.data()))) return rc;
    FILE* fp = std::fopen(path, "wb");
    if (!fp) { c->err = std::string("cannot open ") + path; return IMMESH_E_INVAL; }
    std::fprintf(fp, "ply\nformat binary_little_endian 1.0\ncomment immesh-mi355x\nelement vertex %lld\nproperty float x\nproperty float y\nproperty float z\n"
                     "element face %lld\nproperty list uchar int vertex_indices\nend_header\n", (long long)nv, (long long)nf);
    std::fwrite(v.data(), 4, v.size(), fp);
    std::vector<unsigned char> rec((size_t)nf * 13);
    for (int64_t i = 0; i < nf; i++) { rec[(size_t)i * 13] = 3; std::memcpy(&rec[(size_t)i * 13 + 1], &f[(size_t)i * 3], 12); }
    std::fwrite(rec.data(), 1, rec.size(), fp);
    const bool ok = std::fclose(fp) == 0;
    if (!ok) { c->err = std::string("write failed: ") + path; return IMMESH_E_INVAL; }
    return 0;
}

}  // extern "C"
