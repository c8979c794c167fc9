// C ABI of libimmesh_hip.so (include/immesh_c_api.h): host orchestration of the HIP kernels.
// There is NO CPU compute path here: every per-point / per-voxel operation is a kernel in reg_kernels.hip / mesh_kernels.hip;
// the host keeps only the 18x18 EKF algebra (as the reference does) and stream plumbing.
#include "host_ctx.hpp"
#include "imu_host.hpp"
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <new>
#include <thread>

#define RP_ABORTED 1   /* internal: register_collect_fused found the resident-grid registration aborted (bounded gather) */
static thread_local std::string g_create_error;
thread_local KProf* g_kprof = nullptr;


static int64_t next_pow2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }

extern "C" {

void immesh_default_config(immesh_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->voxel_size = 0.5; c->max_layer = 2;
    for (int i = 0; i < 5; i++) c->layer_init[i] = 5;
    c->max_points_size = 100; c->planer_threshold = 0.01; c->dept_err = 0.02; c->beam_err = 0.05; c->calib_laser = 0;
    c->sigma_num = 3.0; c->max_iter = 4;
    c->extR[0] = c->extR[4] = c->extR[8] = 1.0;
    c->extT[0] = 0.04165; c->extT[1] = 0.02326; c->extT[2] = -0.0284;
    c->mesh_min_spacing = 0.1; c->mesh_voxel = 0.4; c->mesh_region = 10.0; c->mesh_append_budget = 10000;
}

const char* immesh_create_error(void) { return g_create_error.c_str(); }
const char* immesh_last_error(immesh_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

static int alloc_all(immesh_ctx* c) {
    const immesh_config& g = c->cfg;
    RegMapDev& m = c->map;
    const int64_t cap_roots = g.cap_root_voxels > 0 ? g.cap_root_voxels : (1 << 20);
    const int64_t cap_nodes = g.cap_nodes > 0 ? g.cap_nodes : cap_roots + cap_roots / 2;
    const int64_t cap_chunks = g.cap_point_chunks > 0 ? g.cap_point_chunks : cap_nodes * 2;
    const int64_t cap_ext = g.max_points_size > IM_INLINE_CHUNKS * IM_CHUNK_PTS ? std::max<int64_t>(1024, cap_nodes / 4) : std::max<int64_t>(1024, cap_nodes / 64);
    const int64_t hcap = next_pow2(cap_roots * 2);
    if (cap_nodes > 0x7fffffff || cap_chunks > 0x7fffffff || hcap > 0xffffffffLL) { c->err = "capacity too large for 32-bit indices"; return IMMESH_E_INVAL; }
    int rc;
#define A(ptr, n) if ((rc = c->dalloc(&(ptr), (size_t)(n)))) return rc
    A(m.htab, hcap); A(m.slot_head, hcap);
    m.hmask = (uint64_t)hcap - 1;
    A(m.nodes, cap_nodes);
    const int64_t cap_leaf = std::max<int64_t>(4096, cap_nodes / 2);
    A(m.leaf_chunks, cap_leaf * 16); m.cap_leaf_chunks = (int32_t)cap_leaf;
    A(m.chunk_data, cap_chunks * IM_CHUNK_PTS * IM_PT_DOUBLES); A(m.ext_tables, cap_ext * IM_EXT_CHUNKS);
    A(m.counters, 16); A(m.free_ready, cap_chunks); A(m.free_pending, cap_chunks);
    m.cap_nodes = (int32_t)cap_nodes; m.cap_chunks = (int32_t)cap_chunks; m.cap_ext = (int32_t)cap_ext;
    m.max_layer = g.max_layer; m.max_points_size = g.max_points_size;
    for (int i = 0; i < 5; i++) m.init_size[i] = g.layer_init[i];
    m.planer_threshold = (float)g.planer_threshold;
    m.voxel_size_f = (float)g.voxel_size;
    m.voxel_size_d = g.voxel_size;
    m.shard_rank = g.shard_world > 1 ? g.shard_rank : 0; m.shard_world = g.shard_world > 1 ? g.shard_world : 1;
    m.shard_brick_log2 = g.shard_brick_log2 > 0 ? g.shard_brick_log2 : 5;
    m.shard_scheme = g.shard_scheme == 1 ? 1 : 0;
    HIPCHK(c, hipMemsetAsync(m.counters, 0, 16 * sizeof(int32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(m.htab, 0xFF, hcap * sizeof(HashEnt), c->stream));   // key = IM_KEY_EMPTY, root = -1
    HIPCHK(c, hipMemsetAsync(m.slot_head, 0, hcap * sizeof(unsigned long long), c->stream));
    m.upd_seq = 0;
    A(c->d_stats, STATS_WORDS);
    if (getenv("IMMESH_DEBUG")) { A(c->reg_dbg, REG_DBG_WORDS); HIPCHK(c, hipMemsetAsync(c->reg_dbg, 0, (size_t)REG_DBG_WORDS * 8, c->stream)); }
    HIPCHK(c, hipMemsetAsync(c->d_stats, 0, STATS_WORDS * sizeof(int64_t), c->stream));

    const int64_t ns = g.cap_scan_points > 0 ? g.cap_scan_points : 600000;
    c->cap_scan = ns;
    A(c->d_pts_down, ns * 3); A(c->d_pts_raw, ns * 4);
    A(c->d_partials, ((ns + 63) / 64) * RES_NR_HOST); A(c->d_out48, RES_NV_HOST); A(c->d_done, 4);
    HIPCHK(c, hipMemsetAsync(c->d_done, 0, 16, c->stream));
    A(c->d_match, ns); A(c->d_mnode, ns); A(c->d_dis, ns); A(c->d_rinv, ns); A(c->d_normal, ns * 3);
    A(c->d_ptdata, ns * IM_PT_DOUBLES);
    A(c->d_key_a, ns); A(c->d_key_b, ns); A(c->d_idx_a, ns); A(c->d_idx_b, ns); A(c->d_idx_c, ns);
    A(c->d_slot, ns); A(c->d_slot_s, ns); A(c->d_seg_start, ns); A(c->d_nseg, 16); A(c->d_ds_out, ns * 3);
    c->sort_temp_bytes = std::max({sort_pairs_u64_temp_bytes((int)ns), sort_pairs_u32_temp_bytes((int)ns), exclusive_sum_temp_bytes((int)ns)}) + 256;
    { char* t; A(t, c->sort_temp_bytes); c->d_sort_temp = t; }
    A(c->p_key_a, ns); A(c->p_key_b, ns); A(c->p_idx_a, ns); A(c->p_idx_b, ns); A(c->p_idx_c, ns); A(c->p_seg, ns); A(c->p_nseg, 16); A(c->p_slot, ns); A(c->p_slot_s, ns); A(c->p_pool4, ns * 4);
    { char* t; A(t, c->sort_temp_bytes); c->p_sort_temp = t; }
    {   // the VoxelGrid's leaf table: >= 2 entries per point of the largest cloud, all empty (0xFF: key == ~0, chain head == -1)
        unsigned long long cap = 1024; while (cap < 2ull * (unsigned long long)ns) cap <<= 1;
        char* t; A(t, cap * 16); c->p_htab = t; c->p_htab_cap = cap;
        launch_ds_table_reset(c->stream, t, cap);
        HIPCHK(c, hipHostMalloc((void**)&c->h_ds_dyn, sizeof(DsDyn), hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_ds_dyn, c->h_ds_dyn, 0));
        std::memset(c->h_ds_dyn, 0, sizeof(DsDyn));
        HIPCHK(c, hipHostMalloc((void**)&c->h_ds_info, 16 * sizeof(int32_t), hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_ds_info, c->h_ds_info, 0));
        std::memset(c->h_ds_info, 0, 16 * sizeof(int32_t));
        HIPCHK(c, hipMemsetAsync(c->p_nseg, 0, 16 * sizeof(int32_t), c->stream));   // (the VoxelGrid's device counters start out zero and are handed back zeroed)
    }
    {   // deep octrees: subtree work items of the map update (regmap.hpp); off for the two-layer avia map, whose general voxels are new or just cut
        static const char* e = getenv("IMMESH_SPLIT_GENERAL");
        m.split_general = e ? atoi(e) : (g.max_layer >= 3 ? 1 : 0);
        A(m.sub_order, ns); A(m.sub_items, 2 * ns);
        HIPCHK(c, hipMemsetAsync(m.sub_items, 0, (size_t)2 * ns * sizeof(unsigned long long), c->stream));   // (a zero item = child 0, no points: harmless if ever read unwritten)
    }
    A(c->d_dump_count, 2);
    A(c->d_touched, 2 * ns + 16);
    A(c->d_regstate, 1);
    A(c->d_epi, 8);
    HIPCHK(c, hipMemsetAsync(c->d_epi, 0, 32, c->stream));
    for (int q = 0; q < 2; q++) { A(c->d_rp_slots[q], RP_SLOT_DOUBLES); launch_fill_u64(c->stream, (unsigned long long*)c->d_rp_slots[q], RP_SLOT_SENTINEL, RP_SLOT_DOUBLES); }
    HIPCHK(c, hipMemsetAsync(c->d_regstate, 0, sizeof(RegState), c->stream));
    A(c->d_und_in, ns * 5); A(c->d_und_out, ns * 4); A(c->d_und_tab, 64 * 23 + 24);
#undef A
    HIPCHK(c, hipHostMalloc((void**)&c->h_out48, RES_NV_HOST * sizeof(double), hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_out48_host, c->h_out48, 0));
    HIPCHK(c, hipHostMalloc((void**)&c->h_reg_out, REG_OUT_DOUBLES * sizeof(double), hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_reg_out_host, c->h_reg_out, 0));
    std::memset(c->h_reg_out, 0, REG_OUT_DOUBLES * sizeof(double));
    HIPCHK(c, hipHostMalloc((void**)&c->h_epi_flag, 64, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_epi_flag_host, c->h_epi_flag, 0));
    std::memset(c->h_epi_flag, 0, 64);
    HIPCHK(c, hipHostMalloc((void**)&c->h_counters, 16 * sizeof(int32_t), hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_counters_host, c->h_counters, 0));
    return 0;
}


immesh_ctx* immesh_create(const immesh_config* cfg) {
    g_create_error.clear();
    if (!cfg) { g_create_error = "null config"; return nullptr; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no usable HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") + "); the hot path has no CPU fallback";
        return nullptr;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "device ordinal out of range"; return nullptr; }
    if (cfg->max_layer < 0 || cfg->max_layer > 4 || cfg->voxel_size <= 0 || cfg->max_iter < 1) { g_create_error = "invalid config"; return nullptr; }
    if (cfg->shard_world > 1 && (cfg->shard_rank < 0 || cfg->shard_rank >= cfg->shard_world || cfg->shard_brick_log2 < 0 || cfg->shard_brick_log2 > 16)) { g_create_error = "invalid shard configuration"; return nullptr; }
    immesh_ctx* c = new (std::nothrow) immesh_ctx();
    if (!c) { g_create_error = "out of host memory"; return nullptr; }
    c->cfg = *cfg;
    std::memset(&c->cnt, 0, sizeof(c->cnt));
    // the registration stream is the latency-critical chain (pose out per scan): highest priority; the mesher's streams take the lowest
    int prio_least = 0, prio_greatest = 0;
    if (hipSetDevice(cfg->device) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (getenv("IMMESH_NO_PRIORITY")) prio_greatest = prio_least = 0;
    const hipError_t se = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest);
    if (se != hipSuccess) {
        g_create_error = "hipSetDevice/hipStreamCreate failed"; delete c; return nullptr;
    }
    for (auto& ev : c->ev) (void)hipEventCreate(&ev);
    // IMMESH_MESH_CUS=n (measurement knob, see mesh_alloc): the mesher's streams -- and this one, which carries the mesher's triangulations -- are confined to n CUs
    hipError_t pre_rc = hipErrorUnknown;
    {
        int ncu = 0;
        if (const char* e = getenv("IMMESH_MESH_CUS")) ncu = atoi(e);
        if (ncu >= 8 && ncu < 1024) {
            uint32_t mask[32];
            std::memset(mask, 0, sizeof(mask));
            for (int i = 0; i < ncu; i++) mask[i >> 5] |= 1u << (i & 31);
            pre_rc = hipExtStreamCreateWithCUMask(&c->stream_pre, 32, mask);
        } else pre_rc = hipStreamCreateWithFlags(&c->stream_pre, hipStreamNonBlocking);
    }
    if (pre_rc != hipSuccess || hipEventCreateWithFlags(&c->ev_inputs_free, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(c->ev_inputs_free, c->stream) != hipSuccess || !(c->ev_inputs_cur = c->ev_inputs_free)) {
        g_create_error = "hipStreamCreate/hipEventCreate failed"; immesh_destroy(c); return nullptr;
    }
    // per-config constants of calcBodyVar: pow(sin(DEG2RAD(deg)),2) with PCL's DEG2RAD(x) = x*0.017453293 and float `degree_inc`
    { const double s = std::sin((double)(float)cfg->beam_err * 0.017453293); c->dvar_beam = s * s; }
    { const double s = std::sin((double)(float)0.01 * 0.017453293); c->dvar_calib = s * s; }
    {   // resident-grid registration: never more than HALF of the workgroups the device holds at once (two contexts cannot wait for each other's CUs)
        const int resident = residual_persistent_resident_blocks(cfg->device);
        c->rp_max_blocks = std::max(1, resident / 2 - 1);
        if (const char* e = getenv("IMMESH_RP_BLOCKS")) c->rp_max_blocks = std::max(1, atoi(e));
        c->rp_force_abort = getenv("IMMESH_RP_FORCE_ABORT") != nullptr;
    }
    int rc = alloc_all(c);
    if (!rc) rc = mesh_alloc(c);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) { rc = IMMESH_E_HIP; c->err = "initialisation kernels failed"; }
    if (rc) { g_create_error = c->err; immesh_destroy(c); return nullptr; }
    return c;
}

void immesh_destroy(immesh_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream_pre) { (void)hipStreamSynchronize(c->stream_pre); (void)hipStreamDestroy(c->stream_pre); }
    if (c->ev_inputs_free) (void)hipEventDestroy(c->ev_inputs_free);
    if (c->dsa.ev) (void)hipEventDestroy(c->dsa.ev);
    if (c->dsa.h_info) (void)hipHostFree(c->dsa.h_info);
    if (c->ds_graph) (void)hipGraphExecDestroy(c->ds_graph);
    if (c->h_ds_dyn) (void)hipHostFree(c->h_ds_dyn);
    if (c->h_ds_info) (void)hipHostFree(c->h_ds_info);
    mesh_free(c);
    rccl_release(c);
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->h_out48) (void)hipHostFree(c->h_out48);
    if (c->h_reg_out) (void)hipHostFree(c->h_reg_out);
    if (c->h_epi_flag) (void)hipHostFree(c->h_epi_flag);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->h_pack) (void)hipHostFree(c->h_pack);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ---------------------------------------------------------------------------------------------------------------------
static void make_scan_params(const immesh_ctx* c, const imh::State& st, const double* prior_cov, ScanParams& sp) {
    const immesh_config& g = c->cfg;
    std::memcpy(sp.R, st.R, 72); std::memcpy(sp.t, st.t, 24);
    std::memcpy(sp.extR, g.extR, 72); std::memcpy(sp.extT, g.extT, 24);
    imh::mat3_mul(st.R, g.extR, sp.RextR);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { sp.rot_var[i * 3 + j] = prior_cov[i * 18 + j]; sp.t_var[i * 3 + j] = prior_cov[(3 + i) * 18 + (3 + j)]; }
    sp.dvar_beam = c->dvar_beam; sp.dvar_calib = c->dvar_calib; sp.sigma_num = g.sigma_num;
    sp.dept_err = (float)g.dept_err; sp.calib_laser = g.calib_laser;
    sp.dbg = c->reg_dbg;
}

static int check_overflow(immesh_ctx* c) {  // after a stream sync
    const int f = c->h_counters[5];
    if (f) {
        static const char* why[] = {"", "point-chunk pool exhausted (cap_point_chunks)", "node exceeds 32896 retained points", "extension-table pool exhausted",
                                    "node pool exhausted (cap_nodes)", "root-voxel hash full (cap_root_voxels)",
                                    "(unused)",
                                    "leaf-list pool exhausted (cap_nodes)", "a root's leaf-list lock was not released (device hang guard)"};
        c->err = std::string("registration map capacity: ") + why[f < 9 ? f : 0];
        return IMMESH_E_CAPACITY;
    }
    return 0;
}

// An asynchronous immesh_process_scan leaves its map update (and the copy of the capacity flags) running; whoever needs the stream idle, the
// stage timings or the flags settles it first.  `synced`: the caller knows the stream has already passed that work.
static int settle(immesh_ctx* c, bool synced = false) {
    if (!c->pending) return 0;
    if (!synced) {
        if (c->tail_deferred) { launch_map_update_tail(c->stream, c->map, c->d_counters_host); c->tail_deferred = false; }   // nobody registers next: run the update's tail now
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->pending = false;
    hipEvent_t* e = c->ev + 4 * c->ev_par;
    c->timing[1] = c->timing[2] = 0.f;
    if (c->timing_valid[c->ev_par]) {
        (void)hipEventElapsedTime(&c->timing[1], e[0], e[1]);
        (void)hipEventElapsedTime(&c->timing[2], e[1], e[2]);
    }
    c->timing[0] = c->timing[1] + c->timing[2] + c->timing[3];
    return check_overflow(c);
}

static int run_residual_pass(immesh_ctx* c, const float* d_pts, int n, const imh::State& st, const double* prior_cov) {
    ScanParams sp;
    make_scan_params(c, st, prior_cov, sp);
    // the last block of the launch writes the 48 sums straight into pinned host memory: one launch + one stream sync per EKF iteration
    // and a completion ticket after them; the host polls the ticket (a few microseconds) instead of paying a stream synchronisation
    const double ticket = (double)(++c->res_ticket);
    RegIterArgs& a = c->reg_args;
    a.mode = REG_MODE_HOST; a.it = 0; a.max_iter = c->cfg.max_iter; a.sp = sp;
    launch_residual(c->stream, c->map, a, c->d_regstate, d_pts, n, c->d_partials, c->d_done, c->d_out48_host, c->d_reg_out_host, ticket, c->d_match, c->d_mnode, c->d_dis,
                    c->d_rinv, c->d_normal);
    {
        volatile double* flag = c->h_out48 + (RES_NV_HOST - 1);
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (*flag != ticket) {
            if ((++spins & 0x3FF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;   // never spin unbounded: fall back to the stream
        }
        if (*flag != ticket || c->prof.on) HIPCHK(c, hipStreamSynchronize(c->stream));
        if (*flag != ticket) { c->err = "residual kernel did not complete"; return IMMESH_E_HIP; }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (c->allreduce) {   // sharded map: sum the partial normal equations (36 + 6) and the 4 counters over the ranks -- RCCL / gloo behind the callback
        const int rc = c->allreduce(c->h_out48, RES_NV_HOST - 2, c->allreduce_user);
        if (rc) { c->err = "all-reduce callback failed"; return IMMESH_E_INVAL; }
    }
    c->cnt.n_match += (int64_t)c->h_out48[42];
    c->cnt.n_plane_tests += (int64_t)c->h_out48[44];
    c->cnt.n_extra_probe += (int64_t)c->h_out48[45];
    return 0;
}

// compact per-point match outputs (ascending scan index == the reference's ptpl_list order)
static int fetch_matches(immesh_ctx* c, int n, std::vector<int8_t>& mt) {
    mt.resize(n);
    HIPCHK(c, hipMemcpyAsync(mt.data(), c->d_match, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// The same iterated update as a chain of launches per pass: residual_kernel (sums to device memory) -> [ncclAllReduce of the device-resident sums] ->
// the 18-state update as its own one-wavefront launch (ekf_step_kernel); all passes enqueued up front, no host involvement, no co-residency needed.
// Serves the sharded map (the all-reduce sits between pass and update) and is the fallback of a resident-grid registration that gave up.
// c->reg_args (.sp, .st, .prior, .max_iter) and c->reg_ticket are set by the caller.
static int register_enqueue_chain(immesh_ctx* c, const float* d_pts, int n_ds, const imh::State& st) {
    RegIterArgs& a = c->reg_args;
    const int max_iter = c->cfg.max_iter;
    for (int it = 0; it < max_iter; it++) {
        a.it = it; a.mode = REG_MODE_SUMS;
        if (it == 0) {
            double p11[36];
            for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) p11[r * 6 + q] = st.cov[r * 18 + q];
            if (!imh::invert(p11, a.mat, 6)) { c->err = "singular prior covariance"; return IMMESH_E_INVAL; }
            for (int i = 0; i < 12; i++)
                for (int q = 0; q < 6; q++) { double sacc = 0; for (int k = 0; k < 6; k++) sacc += st.cov[(6 + i) * 18 + k] * a.mat[k * 6 + q]; a.mat[36 + i * 6 + q] = sacc; }
        }
        else if (it == 1) std::memcpy(a.mat, st.cov, sizeof(a.mat));
        launch_residual(c->stream, c->map, a, c->d_regstate, d_pts, n_ds, c->d_partials, c->d_done, c->d_out48, c->d_reg_out_host, c->reg_ticket, c->d_match, c->d_mnode,
                        c->d_dis, c->d_rinv, c->d_normal);
        if (c->rccl_comm) {
            const int rc = rccl_allreduce_f64(c, c->d_out48, RES_NV_HOST - 2, c->stream);
            if (rc) return rc;
        }
        launch_ekf_step(c->stream, a, c->d_regstate, c->d_out48, c->d_reg_out_host, c->reg_ticket);
    }
    return 0;
}
// The iterated update with the 18-state step on the device (reg_kernels.hip: ekf_step_wave in the last block of every residual pass): all
// passes of the scan are enqueued up front, a pass that finds the loop already stopped returns at once.  The posterior stays on the device
// (RegState::sp) for the map update / full-scan transform queued behind it; the host only collects it.
static int register_enqueue_fused(immesh_ctx* c, const float* d_pts, int n_ds, const imh::State& prior, const imh::State& st, const RpEpilogue* ep = nullptr) {
    const int max_iter = c->cfg.max_iter;
    RegIterArgs& a = c->reg_args;
    make_scan_params(c, st, st.cov, a.sp);
    a.max_iter = max_iter; a.pad = 0;
    if (c->tail_deferred && !c->rccl_comm) { a.pad = 1; c->tail_deferred = false; }   // the previous scan's map update left its tail to this launch
    else if (c->tail_deferred) { launch_map_update_tail(c->stream, c->map, c->d_counters_host); c->tail_deferred = false; }
    std::memcpy(a.st, st.R, 72); std::memcpy(a.st + 9, st.t, 24); std::memcpy(a.st + 12, st.vel, 24); std::memcpy(a.st + 15, st.bg, 24); std::memcpy(a.st + 18, st.ba, 24); std::memcpy(a.st + 21, st.g, 24);
    std::memcpy(a.prior, prior.R, 72); std::memcpy(a.prior + 9, prior.t, 24); std::memcpy(a.prior + 12, prior.vel, 24); std::memcpy(a.prior + 15, prior.bg, 24);
    std::memcpy(a.prior + 18, prior.ba, 24); std::memcpy(a.prior + 21, prior.g, 24);
    c->reg_ticket = (double)(++c->res_ticket);
    if (!c->rccl_comm) {
        // ONE launch for the scan: a resident grid runs every pass and the 18-state update (residual_persistent_kernel); a.mat = the prior covariance
        a.mode = REG_MODE_FUSED; a.it = 0;
        if (c->rp_force_abort) a.pad |= 2;   // (IMMESH_RP_FORCE_ABORT: the test hook of the bounded gather)
        static const bool match_seq = getenv("IMMESH_MATCH_SEQ") != nullptr;   // A/B: the lane-by-lane leaf walk of rounds 1-5 instead of the wave-cooperative one
        if (match_seq) a.pad |= 4;
        std::memcpy(a.mat, st.cov, sizeof(a.mat));
        const int par = (c->rp_parity ^= 1);   // this scan's slot buffer; the launch re-arms the other one for the next scan
        RpEpilogue none{};
        launch_residual_persistent(c->stream, c->map, a, c->d_regstate, d_pts, n_ds, c->d_rp_slots[par], c->d_rp_slots[par ^ 1], c->d_counters_host, c->d_reg_out_host, c->reg_ticket, c->d_match, c->d_mnode,
                                   c->d_dis, c->d_rinv, c->d_normal, ep ? *ep : none, c->rp_max_blocks);
        return 0;
    }
    return register_enqueue_chain(c, d_pts, n_ds, st);
}
static int register_collect_fused(immesh_ctx* c, int n_ds, imh::State& st, int* n_iter, int* n_match, double* res_mean) {
    volatile double* flag = c->h_reg_out + (REG_OUT_DOUBLES - 1);
    const double ticket = c->reg_ticket;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (*flag != ticket) {
        if ((++spins & 0x3FF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) break;   // never spin unbounded: fall back to the stream
    }
    if (*flag != ticket || c->prof.on) HIPCHK(c, hipStreamSynchronize(c->stream));
    if (*flag != ticket) { c->err = "registration kernels did not complete"; return IMMESH_E_HIP; }
    std::atomic_thread_fence(std::memory_order_acquire);
    const double* o = c->h_reg_out;
    if (o[348] < 0) return RP_ABORTED;   // the resident grid gave up in a gather (it could not become resident): nothing was written, `st` is untouched
    imh::load_state(o, st);
    const int iters = (int)o[348];
    if (n_iter) *n_iter = iters;
    if (n_match) *n_match = (int)o[349];
    if (res_mean) *res_mean = o[349] > 0 ? o[350] / o[349] : 0.0;
    c->cnt.n_match += (int64_t)o[353];
    c->cnt.n_plane_tests += (int64_t)o[351];
    c->cnt.n_extra_probe += (int64_t)o[352];
    c->cnt.n_iter += iters;
    c->cnt.n_ds = n_ds;
    c->last_n_ds = n_ds;
    return 0;
}
// A resident-grid registration that gave up: register the scan with the per-pass chain instead.  The aborted launch (and whatever was queued behind it)
// has drained when this returns to the caller's collect.
static int register_fallback_chain(immesh_ctx* c, const float* d_pts, int n_ds, const imh::State& st) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->rp_fallbacks++;
    c->reg_args.pad = 0;
    c->reg_ticket = (double)(++c->res_ticket);
    return register_enqueue_chain(c, d_pts, n_ds, st);
}
static bool use_fused_ekf(const immesh_ctx* c) {
    static const bool host_ekf = getenv("IMMESH_HOST_EKF") != nullptr;   // debugging: the round-1 host loop (one round trip per pass)
    return !host_ekf && !c->allreduce && c->cfg.max_iter >= 2 && c->cfg.max_iter < 62 && (c->cfg.shard_world <= 1 || c->rccl_comm != nullptr);
}

// the iterated update on device-resident points; leaves per-point match outputs of the LAST iteration in the ctx
static int register_device(immesh_ctx* c, const float* d_pts, int n_ds, const imh::State& prior, imh::State& st, int* n_iter, int* n_match, double* res_mean) {
    c->last_reg_pts = d_pts;
    if (use_fused_ekf(c)) {
        int rc = register_enqueue_fused(c, d_pts, n_ds, prior, st);
        if (rc) return rc;
        rc = register_collect_fused(c, n_ds, st, n_iter, n_match, res_mean);
        if (rc != RP_ABORTED) return rc;
        if ((rc = register_fallback_chain(c, d_pts, n_ds, st))) return rc;
        return register_collect_fused(c, n_ds, st, n_iter, n_match, res_mean);
    }
    imh::EkfLoop ekf;
    const int max_iter = c->cfg.max_iter;
    int iters = 0;
    for (int it = 0; it < max_iter; it++) {
        iters++;
        int rc = run_residual_pass(c, d_pts, n_ds, st, st.cov);
        if (rc) return rc;
        const double* o = c->h_out48;
        if (n_match) *n_match = (int)o[42];
        if (res_mean) *res_mean = o[42] > 0 ? o[43] / o[42] : 0.0;
        if (ekf.step(o, o + 36, prior, st, it, max_iter)) break;
    }
    c->cnt.n_iter += iters;
    c->cnt.n_ds = n_ds;
    c->last_n_ds = n_ds;
    if (n_iter) *n_iter = iters;
    return 0;
}

int immesh_register(immesh_ctx* c, const float* pts, int32_t n_ds, const double* state_prior, double* state_inout, int32_t* n_iter_out,
                    int32_t* n_match_out, double* res_mean_out, float* eff_pts_body, float* eff_norm_dis) {
    if (!c || !pts || n_ds <= 0 || n_ds > c->cap_scan || !state_prior || !state_inout) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    c->ds_gate_ok = false;
    if (const int s_rc = settle(c)) return s_rc;
    const void* d_pts;
    int rc = resolve_input(c, pts, (size_t)n_ds * 12, c->d_pts_down, &d_pts);
    if (rc) return rc;
    imh::State prior, st;
    imh::load_state(state_prior, prior); imh::load_state(state_inout, st);
    int n_iter = 0, n_match = 0; double res = 0;
    rc = register_device(c, (const float*)d_pts, n_ds, prior, st, &n_iter, &n_match, &res);
    if (rc) return rc;
    imh::store_state(st, state_inout);
    if (n_iter_out) *n_iter_out = n_iter;
    if (n_match_out) *n_match_out = n_match;
    if (res_mean_out) *res_mean_out = res;
    if (eff_pts_body || eff_norm_dis) {
        int32_t k = 0;
        if ((rc = immesh_last_matches(c, eff_pts_body, eff_norm_dis, n_ds, &k))) return rc;
    }
    return 0;
}

int immesh_last_matches(immesh_ctx* c, float* eff_pts_body, float* eff_norm_dis, int32_t cap, int32_t* n_out) {
    if (!c || !n_out) return IMMESH_E_INVAL;
    (void)hipSetDevice(c->cfg.device);
    const int n_ds = c->last_n_ds;
    const float* d_pts = c->last_reg_pts;
    *n_out = 0;
    if (n_ds <= 0 || !d_pts) return 0;
    int rc;
    std::vector<int8_t> mt;
    if ((rc = fetch_matches(c, n_ds, mt))) return rc;
    int m = 0;
    for (int i = 0; i < n_ds; i++) m += mt[i] != 0;
    *n_out = m;
    if (!eff_pts_body && !eff_norm_dis) return 0;
    if (m > cap) { c->err = "output buffer too small"; return IMMESH_E_CAPACITY; }
    std::vector<float> hp((size_t)n_ds * 3), hd(n_ds);
    std::vector<double> hn((size_t)n_ds * 3);
    HIPCHK(c, hipMemcpy(hp.data(), d_pts, (size_t)n_ds * 12, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(hd.data(), c->d_dis, (size_t)n_ds * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(hn.data(), c->d_normal, (size_t)n_ds * 24, hipMemcpyDeviceToHost));
    int k = 0;
    for (int i = 0; i < n_ds; i++)
        if (mt[i]) {
            if (eff_pts_body) for (int a = 0; a < 3; a++) eff_pts_body[k * 3 + a] = hp[(size_t)i * 3 + a];
            if (eff_norm_dis) { for (int a = 0; a < 3; a++) eff_norm_dis[k * 4 + a] = (float)hn[(size_t)i * 3 + a]; eff_norm_dis[k * 4 + 3] = hd[i]; }
            k++;
        }
    return 0;
}

int immesh_residuals(immesh_ctx* c, const float* pts, int32_t n_ds, const double* state, double* HTH36, double* HTz6, int32_t* n_match,
                     int32_t* match_idx, double* normals, float* dis, double* r_inv) {
    if (!c || !pts || n_ds <= 0 || n_ds > c->cap_scan || !state || !HTH36 || !HTz6) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    c->ds_gate_ok = false;
    if (const int s_rc = settle(c)) return s_rc;
    const void* d_pts;
    int rc = resolve_input(c, pts, (size_t)n_ds * 12, c->d_pts_down, &d_pts);
    if (rc) return rc;
    imh::State st;
    imh::load_state(state, st);
    if ((rc = run_residual_pass(c, (const float*)d_pts, n_ds, st, st.cov))) return rc;
    std::memcpy(HTH36, c->h_out48, 36 * 8);
    std::memcpy(HTz6, c->h_out48 + 36, 6 * 8);
    if (n_match) *n_match = (int)c->h_out48[42];
    if (match_idx || normals || dis || r_inv) {
        std::vector<int8_t> mt;
        if ((rc = fetch_matches(c, n_ds, mt))) return rc;
        std::vector<float> hd(n_ds);
        std::vector<double> hn((size_t)n_ds * 3), hr(n_ds);
        HIPCHK(c, hipMemcpy(hd.data(), c->d_dis, (size_t)n_ds * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(hn.data(), c->d_normal, (size_t)n_ds * 24, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(hr.data(), c->d_rinv, (size_t)n_ds * 8, hipMemcpyDeviceToHost));
        int k = 0;
        for (int i = 0; i < n_ds; i++)
            if (mt[i]) {
                if (match_idx) match_idx[k] = i;
                if (normals) for (int a = 0; a < 3; a++) normals[k * 3 + a] = hn[(size_t)i * 3 + a];
                if (dis) dis[k] = hd[i];
                if (r_inv) r_inv[k] = hr[i];
                k++;
            }
    }
    return 0;
}

// shared by map_build / map_update: per-point var + root slots, sort, per-voxel replay
// spd != nullptr: pose + covariance blocks come from device memory (the posterior the in-kernel EKF left in RegState::sp); `st` then only
// supplies the per-configuration constants
static int map_ingest_device(immesh_ctx* c, const float* d_pts, int64_t n, int stride, const imh::State& st, int mode, hipEvent_t after_point_var = nullptr,
                             const ScanParams* spd = nullptr, const float* d_raw = nullptr, float* world = nullptr, int n_raw = 0, bool defer_tail = false, bool prep_done = false) {
    ScanParams sp;
    make_scan_params(c, st, st.cov, sp);
    hipStream_t s = c->stream;
    if (mode == 0) {
        // map_incremental_grow: no global sort -- points are chained per root voxel and each voxel's wavefront orders its own points
        // (ascending covariance norm, ties by scan index = std::sort(pv_list, var_contrast) restricted to that voxel) before replaying them
        if (!prep_done) {   // (prep_done: the registration launch ran this part as its epilogue -- immesh_process_scan's fused path)
            if (c->tail_deferred) { launch_map_update_tail(s, c->map, c->d_counters_host); c->tail_deferred = false; }   // (safety: a deferred tail precedes the next update; immesh_process_scan runs it in the residual kernel, every other entry settles first)
            c->map.upd_seq++;
            c->map.touched = c->d_touched;
            launch_point_var(s, c->map, sp, spd, d_pts, (int)n, stride, mode, c->d_ptdata, c->d_key_a, c->d_slot, c->d_idx_a, d_raw, world, n_raw);
            // the scan's input clouds are consumed here (the replay works on its own copies) and the mesher's scan is in its world buffer: ONE event
            // record serves both -- every record is a barrier packet in the queue, ~6 us of bubble on the pose chain (rocprofv3 timeline, round 2)
            if (world) { c->ev_inputs_cur = mesh_record_ready(c); c->inputs_seq = 0; }
            else if (after_point_var) { HIPCHK(c, hipEventRecord(after_point_var, s)); c->ev_inputs_cur = after_point_var; c->inputs_seq = 0; }
        }
        launch_replay_lists(s, c->map, c->d_idx_a, c->d_key_a, c->d_ptdata, (int)n, c->d_stats, c->d_counters_host, c->d_idx_b, c->d_idx_c, c->d_slot_s, c->reg_dbg, !defer_tail,
                            prep_done ? (unsigned long long*)(c->d_epi + 2) : nullptr, prep_done ? c->d_epi_flag_host : nullptr, c->epi_seq);
        c->tail_deferred = defer_tail;
        return 0;   // (the tail kernel has already put the counters into pinned host memory)
    } else {
        // buildVoxelMap: bucket all points per voxel in scan order (stable sort by slot), then initialise every voxel
        launch_point_var(s, c->map, sp, nullptr, d_pts, (int)n, stride, mode, c->d_ptdata, c->d_key_a, c->d_slot, nullptr);
        launch_iota(s, c->d_idx_a, (int)n);
        sort_pairs_u32(s, c->d_sort_temp, c->sort_temp_bytes, c->d_slot, c->d_slot_s, c->d_idx_a, c->d_idx_c, (int)n, 32);  // 0xFFFFFFFF "no slot" sorts last
        launch_segment_heads(s, c->d_slot_s, (int)n, c->d_seg_start, c->d_nseg);
        launch_replay(s, c->map, c->d_slot_s, c->d_idx_c, c->d_ptdata, (int)n, c->d_seg_start, c->d_nseg, (int)n, mode, c->d_stats);
    }
    HIPCHK(c, hipMemcpyAsync(c->h_counters, c->map.counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    return 0;
}

int immesh_map_build(immesh_ctx* c, const float* pts, int64_t n, const double* state) {
    if (!c || !pts || n <= 0 || n > c->cap_scan || !state) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    if (const int s_rc = settle(c)) return s_rc;
    const void* d_pts;
    int rc = resolve_input(c, pts, (size_t)n * 12, c->d_pts_down, &d_pts);
    if (rc) return rc;
    imh::State st;
    imh::load_state(state, st);
    if ((rc = map_ingest_device(c, (const float*)d_pts, n, 3, st, 1))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_overflow(c);
}

int immesh_map_update(immesh_ctx* c, const float* pts, int32_t n_ds, const double* state) {
    if (!c || !pts || n_ds <= 0 || n_ds > c->cap_scan || !state) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    c->ds_gate_ok = false;
    if (const int s_rc = settle(c)) return s_rc;
    const void* d_pts;
    int rc = resolve_input(c, pts, (size_t)n_ds * 12, c->d_pts_down, &d_pts);
    if (rc) return rc;
    imh::State st;
    imh::load_state(state, st);
    if ((rc = map_ingest_device(c, (const float*)d_pts, n_ds, 3, st, 0))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_overflow(c);
}


int immesh_process_scan(immesh_ctx* c, const float* pts_down, int32_t n_ds, const float* pts_raw, int32_t n_raw, const double* state_prior,
                        double* state_inout, int32_t frame_idx, int32_t do_mesh, int32_t* n_iter_out, int32_t* n_match_out) {
    if (!c || !pts_down || n_ds <= 0 || n_ds > c->cap_scan || !state_prior || !state_inout || (do_mesh && (!pts_raw || n_raw <= 0 || n_raw > c->cap_scan))) {
        if (c) c->err = "bad arguments";
        return IMMESH_E_INVAL;
    }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    c->ds_gate_ok = false;
    const void *d_down, *d_raw = nullptr;
    int rc = resolve_input(c, pts_down, (size_t)n_ds * 12, c->d_pts_down, &d_down);
    if (rc) return rc;
    if (do_mesh && (rc = resolve_input(c, pts_raw, (size_t)n_raw * 16, c->d_pts_raw, &d_raw))) return rc;
    imh::State prior, st;
    imh::load_state(state_prior, prior); imh::load_state(state_inout, st);
    int mesh_mode = do_mesh & 3;
    if (mesh_mode == IMMESH_MESH_ASYNC && c->mesh.shard_world > 1) mesh_mode = IMMESH_MESH_SYNC;   // sharded mesher: its collectives must not interleave with the next scan's all-reduces
    // (sharded mesher: never return before mesh_wait -- the worker's all-gathers must not interleave with the next scan's all-reduces)
    const bool nowait = mesh_mode == IMMESH_MESH_ASYNC || ((do_mesh & IMMESH_SCAN_NOWAIT) && !(mesh_mode && c->mesh.shard_world > 1));
    const int par = c->ev_par ^ 1;
    hipEvent_t* ev = c->ev + 4 * par;
    // stage timings (immesh_last_timing [1], [2]) are taken for synchronous calls only: an asynchronous call keeps event records -- barrier packets,
    // each a few microseconds of bubble between two kernels of the pose chain -- out of the queue and reports zeros
    const bool fused = use_fused_ekf(c);
    const bool timed = !(fused && nowait) || c->prof.on;
    if (timed) (void)hipEventRecord(ev[0], c->stream);
    int n_iter = 0, n_match = 0;
    if (fused) {
        // Everything of the scan is enqueued before the host looks at a single result: the residual passes with the in-kernel 18-state update,
        // the full-scan transform and the map update (both read the posterior from RegState::sp on the device).  The host then collects the
        // pose -- by then the device is already growing the map -- and hands the scan to the mesher.
        float* world = nullptr;
        if (mesh_mode) world = mesh_next_world_buffer(c);
        // One launch registers the scan AND prepares its map update (point covariances, root voxels, per-voxel lists) AND moves the full scan into the
        // mesher's world buffer: the resident grid holds the posterior when its loop stops (RpEpilogue).  Sharded map (in-stream all-reduce between
        // the passes): the per-pass launches, then point_var_kernel as before.
        RpEpilogue ep{};
        const bool epi = !c->rccl_comm;
        if (epi) {
            c->map.upd_seq++;
            c->map.touched = c->d_touched;
            ep.enabled = 1; ep.n_raw = world ? n_raw : 0;
            ep.pt_data = c->d_ptdata; ep.sort_key = c->d_key_a; ep.slot_out = c->d_slot; ep.pt_next = c->d_idx_a;
            ep.raw = world ? (const float*)d_raw : nullptr; ep.world = world;
            ++c->epi_seq;   // (stored to the flags by the launch queued behind the registration: launch_replay_lists below)
        }
        c->last_reg_pts = (const float*)d_down;
        if ((rc = register_enqueue_fused(c, (const float*)d_down, n_ds, prior, st, epi ? &ep : nullptr))) return rc;
        if (epi) c->inputs_seq = c->epi_seq;
        if (timed) (void)hipEventRecord(ev[1], c->stream);
        if ((rc = map_ingest_device(c, (const float*)d_down, n_ds, 3, st, 0, c->ev_inputs_free, &c->d_regstate->sp, world ? (const float*)d_raw : nullptr, world, n_raw,
                                    /*defer_tail=*/nowait && !c->rccl_comm, /*prep_done=*/epi))) return rc;
        if (timed) (void)hipEventRecord(ev[2], c->stream);
        c->timing_valid[par] = timed;
        rc = register_collect_fused(c, n_ds, st, &n_iter, &n_match, nullptr);
        bool fell_back = false;
        if (rc == RP_ABORTED) {
            // the resident grid gave up (bounded gather): neither the registration nor its epilogue wrote anything, the replay launches queued behind it
            // found no touched voxel.  Register with the per-pass chain, then prepare the map update with point_var_kernel (+ the transform) as the
            // synchronous entry points do
            fell_back = true;
            if ((rc = register_fallback_chain(c, (const float*)d_down, n_ds, st))) return rc;
            if ((rc = map_ingest_device(c, (const float*)d_down, n_ds, 3, st, 0, c->ev_inputs_free, &c->d_regstate->sp, world ? (const float*)d_raw : nullptr, world, n_raw,
                                        /*defer_tail=*/false, /*prep_done=*/false))) return rc;
            if (timed) (void)hipEventRecord(ev[2], c->stream);
            rc = register_collect_fused(c, n_ds, st, &n_iter, &n_match, nullptr);
            if (rc == RP_ABORTED) { c->err = "registration did not complete"; rc = IMMESH_E_HIP; }
        }
        imh::store_state(st, state_inout);
        if (n_iter_out) *n_iter_out = n_iter;
        if (n_match_out) *n_match_out = n_match;
        // this scan's passes ran behind the previous scan's map update on the same stream: that update is complete now.  On either error return
        // THIS scan's map update is still in flight: it is left pending (under its own event parity) so that the next call settles it and reads its
        // capacity flags
        const int rc_prev = rc ? 0 : settle(c, true);   // (deferred capacity error of the previous update; this scan's pose has been handed back)
        if (rc || rc_prev) { c->ev_par = par; c->pending = true; return rc ? rc : rc_prev; }
        c->ev_par = par;
        long job = 0;
        // IMMESH_SERIAL_SAFE (counter collection: rocprofv3 --pmc runs one kernel at a time, and a mesher kernel polling a flag that a kernel queued
        // BEHIND it on another stream will store never sees it): the mesher waits for an event behind the map update's launches instead -- same
        // kernels, same data, a more conservative order
        static const bool serial_safe = getenv("IMMESH_SERIAL_SAFE") != nullptr;
        if (mesh_mode && serial_safe && epi && !fell_back) { (void)mesh_record_ready(c); job = mesh_submit(c, world, n_raw, st.t, frame_idx, true); }
        else if (mesh_mode) job = (epi && !fell_back) ? mesh_submit(c, world, n_raw, st.t, frame_idx, true, (const unsigned long long*)(c->d_epi + 2), c->epi_seq) : mesh_submit(c, world, n_raw, st.t, frame_idx, true);
        c->timing[3] = 0.f;
        c->pending = true;
        c->ds_gate_ok = nowait && epi && !fell_back;
        if (nowait) return 0;
        if ((rc = settle(c))) return rc;
        if (mesh_mode == IMMESH_MESH_SYNC && (rc = mesh_wait(c, job))) return rc;
        c->timing[0] = c->timing[1] + c->timing[2] + c->timing[3];
        return 0;
    }
    if ((rc = register_device(c, (const float*)d_down, n_ds, prior, st, &n_iter, &n_match, nullptr))) return rc;
    // the residual passes of this scan ran behind the previous scan's map update on the same stream: that update is complete now
    if ((rc = settle(c, true))) {   // deferred capacity error of the PREVIOUS scan's map update: this scan's pose is still valid and is handed back
        imh::store_state(st, state_inout);
        if (n_iter_out) *n_iter_out = n_iter;
        if (n_match_out) *n_match_out = n_match;
        return rc;
    }
    c->ev_par = par;
    c->timing_valid[par] = true;
    (void)hipEventRecord(ev[1], c->stream);
    long job = 0;
    // IMMESH_SERIAL_ORDER: map growth first, then the hand-over to the mesher -- the order of the reference's map_incremental_grow; the default
    // hands the scan over first (the pose is final), so the mesher starts a map update earlier and the input clouds are free for the next
    // scan's pre-processing as soon as point_var has run
    static const bool serial_order = getenv("IMMESH_SERIAL_ORDER") != nullptr;
    if (serial_order) {
        if ((rc = map_ingest_device(c, (const float*)d_down, n_ds, 3, st, 0, nullptr))) return rc;
        (void)hipEventRecord(ev[2], c->stream);
    }
    if (mesh_mode) {
        // transformLidar of the full scan on this stream, then hand the scan to the mesher (its own streams + worker thread), as
        // map_incremental_grow hands it to service_reconstruct_mesh (ImMesh_mesh_reconstruction.cpp:413-417)
        float* world = mesh_next_world_buffer(c);
        if ((rc = mesh_transform_full(c, (const float*)d_raw, world, n_raw, st))) return rc;
        job = mesh_submit(c, world, n_raw, st.t, frame_idx);
    }
    if (serial_order) { (void)hipEventRecord(c->ev_inputs_free, c->stream); c->ev_inputs_cur = c->ev_inputs_free; c->inputs_seq = 0; }
    else {
        if ((rc = map_ingest_device(c, (const float*)d_down, n_ds, 3, st, 0, c->ev_inputs_free))) return rc;
        (void)hipEventRecord(ev[2], c->stream);
    }
    (void)hipEventRecord(ev[3], c->stream);
    c->timing[3] = 0.f;   // (immesh_mesh_wait fills in the mesher's time)
    imh::store_state(st, state_inout);
    if (n_iter_out) *n_iter_out = n_iter;
    if (n_match_out) *n_match_out = n_match;
    c->pending = true;
    if (nowait) return 0;   // the pose is final; map growth (and meshing) finish in the background, ordered before the next call's work
    if ((rc = settle(c))) return rc;
    if (mesh_mode == IMMESH_MESH_SYNC && (rc = mesh_wait(c, job))) return rc;   // synchronous mode: results are current on return
    c->timing[0] = c->timing[1] + c->timing[2] + c->timing[3];
    return 0;
}

// One cloud of immesh_process_scan_strided into the library's packed staging buffer on the registration stream.  Device memory: a gather kernel.  Host
// memory: packed by this thread straight into PINNED staging (one pass over the cloud), then one asynchronous copy -- the pageable path costs a pass by
// the caller (pcl -> packed floats) plus the runtime's own staging pass.
static int stage_strided(immesh_ctx* c, const void* p, int n, int stride, int int_off, float* d_dst, size_t pack_off) {
    hipPointerAttribute_t attr;
    bool is_dev = false;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess) is_dev = (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
    else (void)hipGetLastError();
    if (is_dev) { launch_unpack_strided(c->stream, p, n, stride, int_off, d_dst); return 0; }
    const int nf = int_off >= 0 ? 4 : 3;
    float* dst = (float*)(c->h_pack + pack_off);
    const unsigned char* src = (const unsigned char*)p;
    if (nf == 4) for (int i = 0; i < n; i++) { const float* q = (const float*)(src + (size_t)i * stride); dst[4 * i] = q[0]; dst[4 * i + 1] = q[1]; dst[4 * i + 2] = q[2]; dst[4 * i + 3] = *(const float*)(src + (size_t)i * stride + int_off); }
    else for (int i = 0; i < n; i++) { const float* q = (const float*)(src + (size_t)i * stride); dst[3 * i] = q[0]; dst[3 * i + 1] = q[1]; dst[3 * i + 2] = q[2]; }
    HIPCHK(c, hipMemcpyAsync(d_dst, dst, (size_t)n * nf * 4, hipMemcpyHostToDevice, c->stream));
    return 0;
}
int immesh_process_scan_strided(immesh_ctx* c, const void* pts_down, int32_t n_ds, int32_t down_stride_bytes, const void* pts_raw, int32_t n_raw, int32_t raw_stride_bytes,
                                int32_t raw_intensity_offset_bytes, const double* state_prior, double* state_inout, int32_t frame_idx, int32_t do_mesh, int32_t* n_iter_out, int32_t* n_match_out) {
    if (!c || !pts_down || n_ds <= 0 || n_ds > c->cap_scan || down_stride_bytes < 12 || (down_stride_bytes & 3) || !state_prior || !state_inout ||
        (do_mesh && (!pts_raw || n_raw <= 0 || n_raw > c->cap_scan || raw_stride_bytes < 16 || (raw_stride_bytes & 3) || raw_intensity_offset_bytes < 12 || (raw_intensity_offset_bytes & 3) ||
                     raw_intensity_offset_bytes + 4 > raw_stride_bytes))) {
        if (c) c->err = "bad arguments";
        return IMMESH_E_INVAL;
    }
    (void)hipSetDevice(c->cfg.device);
    const size_t need = (size_t)n_ds * 12 + 64 + (do_mesh ? (size_t)n_raw * 16 : 0);
    if (c->h_pack_bytes < need) {
        // (the previous call's copies out of the old block have completed: a call returns with the pose, which the registration launch behind the copies produced)
        if (c->h_pack) (void)hipHostFree(c->h_pack);
        c->h_pack = nullptr; c->h_pack_bytes = 0;
        const size_t want = std::max(need, (size_t)c->cap_scan * 28 + 64);
        if (hipHostMalloc((void**)&c->h_pack, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->err = "hipHostMalloc(strided staging)"; return IMMESH_E_NOMEM; }
        c->h_pack_bytes = want;
    }
    int rc;
    if ((rc = stage_strided(c, pts_down, n_ds, down_stride_bytes, -1, c->d_pts_down, 0))) return rc;
    if (do_mesh && (rc = stage_strided(c, pts_raw, n_raw, raw_stride_bytes, raw_intensity_offset_bytes, c->d_pts_raw, ((size_t)n_ds * 12 + 63) & ~(size_t)63))) return rc;
    // the caller's clouds are consumed here (host: packed; device: the gather is queued ahead of everything that reads the staging copy): the scan proper
    // works on the library's own buffers
    return immesh_process_scan(c, c->d_pts_down, n_ds, do_mesh ? c->d_pts_raw : nullptr, n_raw, state_prior, state_inout, frame_idx, do_mesh, n_iter_out, n_match_out);
}

// The stages before the path run on their own stream (they do not read the map) with their own scratch, so they overlap the previous scan's
// map update when immesh_process_scan was asynchronous.  What they share with that scan are its INPUT clouds: the result buffers below are the
// down-sampled / raw clouds an asynchronous immesh_process_scan may still be reading (point_var, transform) -- writers wait for ev_inputs_free.
// "the last asynchronous scan has consumed its input clouds": an event on the registration stream, or -- when the registration launch consumed them in
// its epilogue -- that launch's flag in pinned memory (it arrives a few microseconds behind the pose the caller already holds)
// wait (bounded) until the pinned "input clouds consumed" flag of the registration launch's epilogue has reached inputs_seq; the flag trails the pose the
// caller already holds by a few microseconds, so the wait spins with a pause instead of sleeping, and hands over to the stream when it takes longer
static int wait_inputs_flag(immesh_ctx* c) {
    volatile unsigned long long* f = c->h_epi_flag;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (*f < c->inputs_seq) {
        __builtin_ia32_pause();
        if ((++spins & 0x3FF) == 0) {
            const auto dt = std::chrono::steady_clock::now() - t0;
            if (dt > std::chrono::milliseconds(500)) {
                HIPCHK(c, hipStreamSynchronize(c->stream));   // (a faulted stream returns its error here)
                if (*f < c->inputs_seq) { c->err = "the scan's input clouds were not released (registration stream idle, flag not stored)"; return IMMESH_E_HIP; }
                break;
            }
            if (dt > std::chrono::microseconds(200)) std::this_thread::yield();
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}
static int pre_inputs_fence(immesh_ctx* c) {
    if (c->inputs_seq) return wait_inputs_flag(c);
    HIPCHK(c, hipStreamWaitEvent(c->stream_pre, c->ev_inputs_cur, 0));
    return 0;
}
int immesh_inputs_consumed(immesh_ctx* c) {
    if (!c) return IMMESH_E_INVAL;
    (void)hipSetDevice(c->cfg.device);
    if (c->inputs_seq) return wait_inputs_flag(c);   // the registration launch consumed them in its epilogue: the flag the launch behind it stores, in pinned memory
    HIPCHK(c, hipEventSynchronize(c->ev_inputs_cur));
    return 0;
}
static int pre_resolve(immesh_ctx* c, const void* p, size_t bytes, void* staging, const void** dev_out) {
    hipPointerAttribute_t attr;
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    bool is_dev = false;
    if (e == hipSuccess) is_dev = (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
    else (void)hipGetLastError();
    if (is_dev) { *dev_out = p; return 0; }
    if (const int frc = pre_inputs_fence(c)) return frc;   // the staging buffers double as immesh_process_scan's own staging
    HIPCHK(c, hipMemcpyAsync(staging, p, bytes, hipMemcpyHostToDevice, c->stream_pre));
    *dev_out = staging;
    return 0;
}
#define PRE_OUTPUT_FENCE(c) do { if (const int _frc = pre_inputs_fence(c)) return _frc; } while (0)

// ---- sensor decode: flag -> exclusive scan -> compact, in arrival order
static int decode_finish(immesh_ctx* c, int n, float* out_xyzit, int32_t* n_out) {
    int32_t cnt = 0;
    HIPCHK(c, hipMemcpyAsync(&cnt, c->p_nseg, 4, hipMemcpyDeviceToHost, c->stream_pre));
    HIPCHK(c, hipStreamSynchronize(c->stream_pre));
    if (out_xyzit && cnt > 0) HIPCHK(c, hipMemcpy(out_xyzit, c->d_und_in, (size_t)cnt * 20, hipMemcpyDefault));
    if (n_out) *n_out = cnt;
    (void)n;
    return 0;
}
// Preprocess::avia_handler, feature_enabled == false   src/preprocess.cpp:139-232
int immesh_decode_livox(immesh_ctx* c, const uint8_t* wire, int32_t n, int32_t n_scans, int32_t point_filter_num, double blind, float* out_xyzit, int32_t* n_out) {
    if (!c || !wire || n <= 0 || n > c->cap_scan || n_scans <= 0 || point_filter_num <= 0) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    hipStream_t s = c->stream_pre;
    const void* d_in;
    int rc;
    if (!c->d_raw_stage && (rc = c->dalloc(&c->d_raw_stage, (size_t)c->cap_scan * 64))) return rc;
    if ((rc = pre_resolve(c, wire, (size_t)n * 19, c->d_raw_stage, &d_in))) return rc;
    int32_t* flag = c->p_idx_a; int32_t* scan = c->p_idx_b; int32_t* keep = c->p_idx_c; int32_t* pos = c->p_seg;
    launch_decode_livox_count(s, (const uint8_t*)d_in, n, n_scans, flag);
    exclusive_sum_i32(s, c->p_sort_temp, c->sort_temp_bytes, flag, scan, n);
    launch_decode_livox_keep(s, (const uint8_t*)d_in, n, n_scans, point_filter_num, blind * blind, scan, keep);
    exclusive_sum_i32(s, c->p_sort_temp, c->sort_temp_bytes, keep, pos, n);
    PRE_OUTPUT_FENCE(c);
    launch_decode_livox_emit(s, (const uint8_t*)d_in, n, keep, pos, c->d_und_in, c->p_nseg);
    return decode_finish(c, n, out_xyzit, n_out);
}
// Preprocess::velodyne_handler   src/preprocess.cpp:497-526
int immesh_decode_velodyne(immesh_ctx* c, const uint8_t* data, int32_t n, int32_t point_step, int32_t off_x, int32_t off_y, int32_t off_z, int32_t off_intensity,
                           int32_t n_scans, float* out_xyzit, int32_t* n_out) {
    const int32_t mx = std::max(std::max(off_x, off_y), std::max(off_z, off_intensity));
    if (!c || !data || n <= 0 || n > c->cap_scan || point_step < 16 || point_step > 64 || std::min(std::min(off_x, off_y), std::min(off_z, off_intensity)) < 0 || mx + 4 > point_step) {
        if (c) c->err = "bad arguments (point_step 16..64, float32 fields inside the point)";
        return IMMESH_E_INVAL;
    }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    hipStream_t s = c->stream_pre;
    const void* d_in;
    int rc;
    if (!c->d_raw_stage && (rc = c->dalloc(&c->d_raw_stage, (size_t)c->cap_scan * 64))) return rc;
    if ((rc = pre_resolve(c, data, (size_t)n * point_step, c->d_raw_stage, &d_in))) return rc;
    int32_t* keep = c->p_idx_c; int32_t* pos = c->p_seg;
    launch_decode_velodyne_keep(s, (const uint8_t*)d_in, n, point_step, off_x, off_y, off_z, n_scans, keep);
    exclusive_sum_i32(s, c->p_sort_temp, c->sort_temp_bytes, keep, pos, n);
    PRE_OUTPUT_FENCE(c);
    launch_decode_velodyne_emit(s, (const uint8_t*)d_in, n, point_step, off_x, off_y, off_z, off_intensity, keep, pos, c->d_und_in, c->p_nseg);
    return decode_finish(c, n, out_xyzit, n_out);
}
const float* immesh_decode_result(immesh_ctx* c) { return c ? c->d_und_in : nullptr; }

// ImuProcess::UndistortPcl (src/IMU_Processing.cpp:755-958): IMU forward propagation on the host, per-point compensation on the device
int immesh_undistort(immesh_ctx* c, const float* pts, int32_t n, const immesh_imu_sample* imu, int32_t n_imu, double lidar_beg_time,
                     double* last_update_time, immesh_imu_ctx* ic, double* state_inout, float* out_xyzi) {
    if (!c || !pts || n <= 0 || n > c->cap_scan || n_imu < 0 || (n_imu > 0 && !imu) || n_imu > 62 || !last_update_time || !ic || !state_inout) {
        if (c) c->err = "bad arguments (at most 62 IMU samples per package)";
        return IMMESH_E_INVAL;
    }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    hipStream_t s = c->stream_pre;
    const void* d_in;
    int rc = pre_resolve(c, pts, (size_t)n * 20, c->d_und_in, &d_in);
    if (rc) return rc;
    float end_curv = 0.f;   // curvature of the package's last point in arrival order (pcl_end_time, :785)
    HIPCHK(c, hipMemcpyAsync(&end_curv, (const float*)d_in + (size_t)(n - 1) * 5 + 4, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    imh::State st;
    imh::load_state(state_inout, st);
    std::vector<imh::ImuPose> poses;
    imh::imu_forward(imu, n_imu, lidar_beg_time, last_update_time, end_curv, *ic, st, poses);
    imh::store_state(st, state_inout);
    std::vector<double> tab(poses.size() * 23 + 24);
    for (size_t k = 0; k < poses.size(); k++) std::memcpy(&tab[k * 23], &poses[k], 23 * sizeof(double));
    double* fe = &tab[poses.size() * 23];
    std::memcpy(fe, st.R, 72); std::memcpy(fe + 9, st.t, 24); std::memcpy(fe + 12, ic->lid_rot_to_imu, 72); std::memcpy(fe + 21, ic->lid_offset_to_imu, 24);
    HIPCHK(c, hipMemcpyAsync(c->d_und_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, s));
    launch_undistort_keys(s, (const float*)d_in, n, c->p_slot, c->p_idx_a);
    sort_pairs_u32(s, c->p_sort_temp, c->sort_temp_bytes, c->p_slot, c->p_slot_s, c->p_idx_a, c->p_idx_c, n, 32);   // stable: equal stamps keep arrival order
    PRE_OUTPUT_FENCE(c);
    launch_undistort(s, (const float*)d_in, c->p_idx_c, n, c->d_und_tab, (int)poses.size(), c->d_und_tab + poses.size() * 23, c->d_und_out);
    if (out_xyzi) {
        HIPCHK(c, hipMemcpyAsync(out_xyzi, c->d_und_out, (size_t)n * 16, hipMemcpyDefault, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));   // (also keeps `tab` alive until the upload has run)
    return 0;
}
const float* immesh_undistort_result(immesh_ctx* c) { return c ? c->d_und_out : nullptr; }

int immesh_last_timing(immesh_ctx* c, float ms[4]) {
    if (!c || !ms) return IMMESH_E_INVAL;
    const int rc = settle(c);
    if (rc) return rc;
    for (int i = 0; i < 4; i++) ms[i] = c->timing[i];
    return 0;
}

int immesh_dump_planes(immesh_ctx* c, immesh_plane_rec* out, int64_t cap, int64_t* n_out) {
    if (!c || !n_out) return IMMESH_E_INVAL;
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    if (const int s_rc = settle(c)) return s_rc;
    static_assert(sizeof(PlaneRecDev) == sizeof(immesh_plane_rec), "plane record layout");
    PlaneRecDev* d_out = nullptr;
    if (out && cap > 0) HIPCHK(c, hipMalloc((void**)&d_out, (size_t)cap * sizeof(PlaneRecDev)));
    launch_dump_planes(c->stream, c->map, d_out, d_out ? cap : 0, c->d_dump_count);
    unsigned long long cnt = 0;
    hipError_t e = hipMemcpyAsync(&cnt, c->d_dump_count, sizeof(cnt), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && d_out) e = hipMemcpy(out, d_out, (size_t)std::min<int64_t>(cap, (int64_t)cnt) * sizeof(PlaneRecDev), hipMemcpyDeviceToHost);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) { c->err = std::string("dump_planes: ") + hipGetErrorString(e); return IMMESH_E_HIP; }
    *n_out = (int64_t)cnt;
    return 0;
}


int immesh_counters(immesh_ctx* c, immesh_counters_t* out, int32_t reset) {
    if (!c || !out) return IMMESH_E_INVAL;
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    if (const int s_rc = settle(c)) return s_rc;
    int64_t stats[STATS_WORDS];
    // (copies on the context's stream, never through the legacy stream: this is called in the middle of a scan loop -- bench.py resets the counters behind
    //  its warm-up -- and a legacy-stream operation fails while the mesher's worker thread has a graph capture open: "operation would make the legacy
    //  stream depend on a capturing blocking stream".  Seen in the GPU tier once the mesher captured three graphs per job set instead of two.)
    HIPCHK(c, hipMemcpyAsync(stats, c->d_stats, sizeof(stats), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_counters, c->map.counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 64; k++) { stats[0] += stats[16 + k * 16]; stats[1] += stats[16 + k * 16 + 1]; }   // the fused replay kernel's shards
    *out = c->cnt;
    out->n_refits = stats[0]; out->n_refit_pts = stats[1];
    out->n_root_voxels = c->h_counters[6]; out->n_nodes = c->h_counters[0];
    mesh_counters(c, out);
    if (c->reg_dbg) {   // IMMESH_DEBUG: in-kernel phase timers since the last call (cycles of the shader clock; see the FDBG / RDBG markers in reg_kernels.hip)
        unsigned long long t[64];
        HIPCHK(c, hipMemcpy(t, c->reg_dbg, sizeof(t), hipMemcpyDeviceToHost));
        if (const char* tf = getenv("IMMESH_TRACE_FILE")) {   // the per-wavefront trace records of the LAST launches (tools/trace_report.py reads them)
            std::vector<unsigned long long> all(REG_DBG_WORDS);
            HIPCHK(c, hipMemcpy(all.data(), c->reg_dbg, all.size() * 8, hipMemcpyDeviceToHost));
            if (FILE* f = fopen(tf, "wb")) { fwrite(all.data(), 8, all.size(), f); fclose(f); }
        }
        HIPCHK(c, hipMemset(c->reg_dbg, 0, sizeof(t)));
        fprintf(stderr, "[replay_list] slowest fast-path voxel %llu cycles (%llu pts), slowest general voxel %llu cycles (%llu pts); mean cycles fast %llu (%llu voxels) general %llu (%llu voxels); list gather + sort %llu per voxel\n",
                t[8] >> 16, t[8] & 0xFFFF, t[9] >> 16, t[9] & 0xFFFF, t[10] / std::max(1ull, t[12]), t[12], t[11] / std::max(1ull, t[13]), t[13], t[14] / std::max(1ull, t[12] + t[13]));
        const unsigned long long nwv = std::max(1ull, t[6]), npass = std::max(1ull, t[7]);
        fprintf(stderr, "[residual cycles of wavefront 0, %llu blocks x passes in %llu passes] prep %llu match %llu retry %llu hbuild %llu reduce %llu | last-block tail %llu per pass\n", t[6], t[7], t[0] / nwv, t[1] / nwv,
                t[2] / nwv, t[3] / nwv, t[4] / nwv, t[5] / npass);
    }
    if (reset) {
        mesh_counters_reset(c);
        std::memset(&c->cnt, 0, sizeof(c->cnt));
        HIPCHK(c, hipMemsetAsync(c->d_stats, 0, sizeof(stats), c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

// pcl::VoxelGrid stand-in on the device (the stage before lio_state_estimation, src/voxel_mapping.cpp:1888-1891)
int immesh_downsample(immesh_ctx* c, const float* pts, int32_t n, int32_t stride, double leaf, float* out_xyz, int32_t cap_out, int32_t* n_out) {
    if (!c || !pts || n <= 0 || n > c->cap_scan || (stride != 3 && stride != 4) || leaf <= 0 || !n_out) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    // the synchronous call and an asynchronous job share the pinned parameter block, the leaf table, the counters and the ticket word: between
    // immesh_downsample_begin and immesh_downsample_end the job's kernels still read them
    if (c->dsa.active) { c->err = "immesh_downsample: an asynchronous job is in flight (collect it with immesh_downsample_end first)"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    hipStream_t s = c->stream_pre;
    const void* d_pts;
    int rc = pre_resolve(c, pts, (size_t)n * stride * 4, stride == 4 ? (void*)c->d_pts_raw : (void*)c->d_pts_down, &d_pts);
    if (rc) return rc;
    const float inv = (float)(1.0 / leaf);   // np.float32(1.0 / leaf)
    static const bool radix_only = getenv("IMMESH_DS_RADIX") != nullptr;
    if (!radix_only && !c->ds_skip_hash) {
        // the hashed form (ds_kernels.hip: leaf table, leaf sort, point scatter + output positions, per-leaf ordered sums, publish); the radix pipeline below only when that gives up
        PRE_OUTPUT_FENCE(c);
        *c->h_ds_dyn = DsDyn{(const float*)d_pts, c->d_ds_out, n, stride, inv, 0, nullptr, 0, 0};
        launch_ds_hash_pipeline(s, c->d_ds_dyn, c->p_htab, c->p_htab_cap, c->p_idx_a, c->p_idx_b, c->p_key_b, c->p_idx_c, c->p_pool4, (int32_t*)c->p_slot, c->p_nseg + 12, c->p_nseg + 8);
        launch_ds_publish(s, c->p_nseg + 12, c->d_ds_info, c->d_ds_dyn);
        HIPCHK(c, hipStreamSynchronize(s));
        const int32_t info[2] = {c->h_ds_info[0], c->h_ds_info[1]};
        if (!info[1]) {
            const int32_t cnt = info[0];
            *n_out = cnt;
            if (out_xyz) {
                if (cnt > cap_out) { c->err = "output buffer too small"; return IMMESH_E_CAPACITY; }
                hipPointerAttribute_t attr;
                const bool dev_out = hipPointerGetAttributes(&attr, out_xyz) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
                if (!dev_out) (void)hipGetLastError();
                HIPCHK(c, hipMemcpyAsync(out_xyz, c->d_ds_out, (size_t)cnt * 12, dev_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipStreamSynchronize(s));
            }
            return 0;
        }
        launch_ds_table_reset(s, c->p_htab, c->p_htab_cap);   // (a cell outside the key's range, a full table or a leaf above 2048 points: the general path)
    }
    int32_t* mm = c->p_nseg;                 // 6 ints of scratch: floor(min), floor(max)
    const int init[6] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    HIPCHK(c, hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, s));
    launch_ds_minmax(s, (const float*)d_pts, n, stride, inv, mm);
    launch_ds_index(s, (const float*)d_pts, n, stride, inv, mm, c->p_key_a, c->p_idx_a);
    // the leaf index is below dx * dy * dz: sorting only its significant bits (typically ~24 of 64) cuts the radix passes from 8 to 3 -- worth the
    // small read-back of the grid extents
    int h_mm[6];
    HIPCHK(c, hipMemcpyAsync(h_mm, mm, sizeof(h_mm), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    int key_bits = 1;
    {
        const unsigned long long total = (unsigned long long)((long long)h_mm[3] - h_mm[0] + 1) * (unsigned long long)((long long)h_mm[4] - h_mm[1] + 1) *
                                         (unsigned long long)((long long)h_mm[5] - h_mm[2] + 1);
        while (key_bits < 64 && (total >> key_bits) != 0) key_bits++;
    }
    sort_pairs_u64(s, c->p_sort_temp, c->sort_temp_bytes, c->p_key_a, c->p_key_b, c->p_idx_a, c->p_idx_b, n, key_bits);   // stable: ties keep scan order
    launch_ds_heads(s, c->p_key_b, n, c->p_idx_c);
    exclusive_sum_i32(s, c->p_sort_temp, c->sort_temp_bytes, c->p_idx_c, c->p_idx_c, n);
    float* d_out = c->d_ds_out;
    PRE_OUTPUT_FENCE(c);
    launch_ds_centroid(s, (const float*)d_pts, n, stride, c->p_key_b, c->p_idx_b, c->p_idx_c, d_out, c->p_nseg + 8);
    int32_t cnt = 0;
    HIPCHK(c, hipMemcpyAsync(&cnt, c->p_nseg + 8, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    *n_out = cnt;
    if (out_xyz) {
        if (cnt > cap_out) { c->err = "output buffer too small"; return IMMESH_E_CAPACITY; }
        hipPointerAttribute_t attr;
        const bool dev_out = hipPointerGetAttributes(&attr, out_xyz) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
        if (!dev_out) (void)hipGetLastError();
        HIPCHK(c, hipMemcpyAsync(out_xyz, d_out, (size_t)cnt * 12, dev_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    return 0;
}
const float* immesh_downsample_result(immesh_ctx* c) { return c ? c->d_ds_out : nullptr; }

// ---- the asynchronous pair: the VoxelGrid of scan k+1 enqueued on the pre-processing stream beside scan k's registration, collected later.  The
// hashed form needs nothing from the host in between; when it gives up (flag in the job's pinned info) the job is redone synchronously.
// the asynchronous VoxelGrid's launch sequence (parameters are in the pinned block already)
static int ds_job_launch(immesh_ctx* c) {
    hipStream_t s = c->stream_pre;
    auto enqueue = [&]() -> int {
        launch_ds_gate(s, c->d_ds_dyn);
        launch_ds_hash_pipeline(s, c->d_ds_dyn, c->p_htab, c->p_htab_cap, c->p_idx_a, c->p_idx_b, c->p_key_b, c->p_idx_c, c->p_pool4, (int32_t*)c->p_slot, c->p_nseg + 12, c->p_nseg + 8);
        launch_ds_publish(s, c->p_nseg + 12, c->d_ds_info, c->d_ds_dyn);   // [0] leaves, [1] fall-back wanted, [2] the job's ticket -> pinned memory; device counters back to zero
        return 0;
    };
    static const bool no_graph = getenv("IMMESH_NO_GRAPH") != nullptr;
    int rc;
    if (no_graph || c->prof.on) { if ((rc = enqueue())) return rc; }
    else {
        // the launches never change (everything cloud-specific is in the pinned block): captured once, replayed with ONE call
        if (!c->ds_graph) {
            hipGraph_t g = nullptr;
            HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const int erc = enqueue();
            const hipError_t ce = hipStreamEndCapture(s, &g);
            if (erc || ce != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); c->err = "hipGraph capture of the VoxelGrid failed"; return IMMESH_E_HIP; }
            const hipError_t ie = hipGraphInstantiate(&c->ds_graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ie != hipSuccess) { c->ds_graph = nullptr; c->err = std::string("hipGraphInstantiate(VoxelGrid): ") + hipGetErrorString(ie); return IMMESH_E_HIP; }
        }
        HIPCHK(c, hipGraphLaunch(c->ds_graph, s));
    }
    return 0;
}
int immesh_downsample_begin(immesh_ctx* c, const float* pts, int32_t n, int32_t stride, double leaf) {
    if (!c || !pts || n <= 0 || n > c->cap_scan || (stride != 3 && stride != 4) || leaf <= 0) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    if (c->dsa.active) { c->err = "immesh_downsample_begin: the previous job has not been collected (immesh_downsample_end)"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    immesh_ctx::DsAsync& a = c->dsa;
    int rc;
    if (!a.ready) {   // first use: all or nothing (a failed allocation leaves the job state untouched; what was allocated stays in the context's pool)
        int32_t* info = nullptr; float* o0 = nullptr; float* o1 = nullptr; float* stg = nullptr;
        if ((rc = c->dalloc(&o0, (size_t)c->cap_scan * 3)) || (rc = c->dalloc(&o1, (size_t)c->cap_scan * 3)) || (rc = c->dalloc(&stg, (size_t)c->cap_scan * 4))) return rc;
        if (hipHostMalloc((void**)&info, 16 * sizeof(int32_t)) != hipSuccess) { c->err = "hipHostMalloc(downsample job)"; return IMMESH_E_NOMEM; }
        const int init[6] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000};
        std::memcpy(info + 8, init, sizeof(init));   // (pinned: the source of the asynchronous initialisation below)
        a.h_info = info; a.out[0] = o0; a.out[1] = o1; a.stage = stg; a.ready = true;   // (completion is the ticket ds_publish_kernel stores to pinned memory: no event)
    }
    hipStream_t s = c->stream_pre;
    const void* d_pts;
    {
        // a HOST cloud is staged into the job's OWN buffer: the context's staging buffers (d_pts_raw / d_pts_down) are written by the next
        // immesh_process_scan / immesh_register with host inputs on the registration stream, which is not ordered against this stream
        hipPointerAttribute_t attr;
        const hipError_t pe = hipPointerGetAttributes(&attr, pts);
        const bool is_dev = pe == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
        if (pe != hipSuccess) (void)hipGetLastError();
        if (is_dev) d_pts = pts;
        else { HIPCHK(c, hipMemcpyAsync(a.stage, pts, (size_t)n * stride * 4, hipMemcpyHostToDevice, s)); d_pts = a.stage; }
    }
    a.par ^= 1; a.n = n; a.stride = stride; a.leaf = leaf; a.d_in = d_pts;
    const float inv = (float)(1.0 / leaf);
    // six launches (a gate in front of the five that do the work), nothing the host has to look at in between (the radix pipeline needed the grid extents for its sort width): leaf table, leaf sort,
    // point scatter + output positions, per-leaf ordered sums, publish (ds_kernels.hip)
    PRE_OUTPUT_FENCE(c);   // (the buffer being written was the input of the scan before the one in flight: its point_var has to be through)
    if (++a.ticket <= 0) a.ticket = 1;
    // a scan loop (the last scan was an asynchronous immesh_process_scan): the sequence is held at its gate until the NEXT registration launch is running
    static const bool no_gate = getenv("IMMESH_DS_NO_GATE") != nullptr || getenv("IMMESH_SERIAL_SAFE") != nullptr;
    const bool gate = c->ds_gate_ok && !no_gate && !c->prof.on;
    *c->h_ds_dyn = DsDyn{(const float*)d_pts, a.out[a.par], n, stride, inv, a.ticket, gate ? &c->d_regstate->started : nullptr, gate ? (int32_t)(long long)(c->res_ticket + 1) : 0, 0};   // (pinned: read by thread 0 of every workgroup; the previous job has been collected)
    if ((rc = ds_job_launch(c))) return rc;
    a.active = true;
    return 0;
}
int immesh_downsample_end(immesh_ctx* c, int32_t* n_out, const float** dev_xyz) {
    if (!c || !n_out) return IMMESH_E_INVAL;
    immesh_ctx::DsAsync& a = c->dsa;
    if (!a.active) { c->err = "immesh_downsample_end: no job in flight"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    {
        // the publishing kernel's ticket in pinned memory (bounded spin, then the stream)
        volatile int32_t* flag = c->h_ds_info + 2;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (*flag != a.ticket) {
            if ((++spins & 0x3FF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
        if (*flag != a.ticket || c->prof.on) HIPCHK(c, hipStreamSynchronize(c->stream_pre));
        if (*flag != a.ticket) { a.active = false; c->err = "immesh_downsample_end: the VoxelGrid launches did not complete"; return IMMESH_E_HIP; }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    a.active = false;
    a.h_info[0] = c->h_ds_info[0]; a.h_info[1] = c->h_ds_info[1];
    if (a.h_info[1]) {
        // the hashed form gave up (a cell outside the key's range, a leaf above 2048 points): the general path, synchronously (it resets the table)
        int32_t cnt = 0;
        launch_ds_table_reset(c->stream_pre, c->p_htab, c->p_htab_cap);
        c->ds_skip_hash = true;
        const int rc = immesh_downsample(c, (const float*)a.d_in, a.n, a.stride, a.leaf, nullptr, 0, &cnt);
        c->ds_skip_hash = false;
        if (rc) return rc;
        HIPCHK(c, hipMemcpy(a.out[a.par], c->d_ds_out, (size_t)cnt * 12, hipMemcpyDeviceToDevice));
        a.h_info[0] = cnt;
    }
    *n_out = a.h_info[0];
    if (dev_xyz) *dev_xyz = a.out[a.par];
    return 0;
}

// void reconstruct_mesh_from_pointcloud(pcl::PointCloud<pcl::PointXYZI>::Ptr, double)   src/ImMesh_mesh_reconstruction.cpp:328-345:
// VoxelGrid(leaf) -> one package with the identity pose, frame 0 -> incremental_mesh_reconstruction
int immesh_reconstruct_mesh_from_pointcloud(immesh_ctx* c, const float* pts_xyzi, int32_t n, double leaf) {
    if (!c || !pts_xyzi || n <= 0 || n > c->cap_scan || leaf <= 0) { if (c) c->err = "bad arguments"; return IMMESH_E_INVAL; }
    int32_t n_ds = 0;
    int rc = immesh_downsample(c, pts_xyzi, n, 4, leaf, nullptr, 0, &n_ds);
    if (rc) return rc;
    (void)hipSetDevice(c->cfg.device);
    ProfBind _pb(c);
    float* world = mesh_next_world_buffer(c);
    launch_ds_expand_xyzi(c->stream, c->d_ds_out, n_ds, world);
    const double origin[3] = {0.0, 0.0, 0.0};   // pose_t of the package: Eigen::Vector3d::Zero()
    const long id = mesh_submit(c, world, n_ds, origin, 0);
    return mesh_wait(c, id);
}

int immesh_forward_without_imu(const double* state_in, double dt, double cov_gyr, double cov_acc, double* state_out) {
    if (!state_in || !state_out) return IMMESH_E_INVAL;
    imh::State a, b;
    imh::load_state(state_in, a);
    imh::forward_without_imu(a, dt, cov_gyr, cov_acc, b);
    imh::store_state(b, state_out);
    return 0;
}

int immesh_set_allreduce(immesh_ctx* c, immesh_allreduce_fn fn, void* user) {
    if (!c) return IMMESH_E_INVAL;
    c->allreduce = fn; c->allreduce_user = user;
    return 0;
}

// host mirror of the device ownership function (regmap.hpp shard_owner): lets callers / tests partition keys exactly as the kernels do
static uint64_t h_hash64(uint64_t k) { k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull; k ^= k >> 27; k *= 0x94d049bb133111ebull; k ^= k >> 31; return k; }
static uint64_t h_pack(int64_t x, int64_t y, int64_t z) {
    const uint64_t B = 1 << 20, M = (1ull << 21) - 1;
    return ((uint64_t)(x + B) & M) | (((uint64_t)(y + B) & M) << 21) | (((uint64_t)(z + B) & M) << 42);
}
int immesh_shard_owner(const immesh_config* cfg, const int64_t* key3) {
    if (!cfg || !key3) return IMMESH_E_INVAL;
    if (cfg->shard_world <= 1) return 0;
    const int b = cfg->shard_brick_log2 > 0 ? cfg->shard_brick_log2 : 5;
    if (cfg->shard_scheme == 1) return (int)(h_hash64(h_pack(key3[0] >> b, key3[1] >> b, key3[2] >> b)) % (uint64_t)cfg->shard_world);
    const int64_t col = ((key3[0] >> b) + 3 * (key3[1] >> b) + 5 * (key3[2] >> b)) % (int64_t)cfg->shard_world;   // lattice colouring (immesh_config::shard_scheme)
    return (int)(col < 0 ? col + cfg->shard_world : col);
}

int immesh_registration_fallbacks(immesh_ctx* c, int64_t* n) {
    if (!c || !n) return IMMESH_E_INVAL;
    *n = c->rp_fallbacks;
    return 0;
}

int immesh_profile_enable(immesh_ctx* c, int32_t on) {
    if (!c) return IMMESH_E_INVAL;
    mesh_wait_all(c);
    c->prof.on = on != 0;
    c->mesh_host.prof.on = on != 0;
    return 0;
}

int immesh_profile_read(immesh_ctx* c, immesh_kernel_stat* out, int32_t cap, int32_t* n_out, int32_t reset) {
    if (!c || !n_out) return IMMESH_E_INVAL;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) { HIPCHK(c, hipStreamSynchronize(c->stream)); c->prof.flush(); }
    mesh_wait_all(c);   // the mesher's worker thread keeps its own table (flushed by the worker after every job)
    std::vector<std::string> names = c->prof.names;
    std::vector<double> ms = c->prof.ms;
    std::vector<long long> cnt = c->prof.cnt;
    for (size_t k = 0; k < c->mesh_host.prof.names.size(); k++) { names.push_back(c->mesh_host.prof.names[k]); ms.push_back(c->mesh_host.prof.ms[k]); cnt.push_back(c->mesh_host.prof.cnt[k]); }
    const int n = (int)names.size();
    for (int i = 0; i < n && i < cap && out; i++) {
        std::memset(&out[i], 0, sizeof(out[i]));
        std::strncpy(out[i].name, names[i].c_str(), sizeof(out[i].name) - 1);
        out[i].launches = cnt[i];
        out[i].total_ms = ms[i];
    }
    *n_out = n;
    if (reset) { c->prof.reset(); c->mesh_host.prof.reset(); }
    return 0;
}

}  // extern "C"
