// Host-side context of libimmesh_hip.so: owns the HIP stream, every HBM-resident pool and the scratch buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <string>
#include <vector>
#include <cstdio>
#include "../../include/immesh_c_api.h"
#include "regmap.hpp"
#include "kernels.hpp"
#include "mesh_kernels.hpp"
#include "ikd_map.hpp"
#include "ekf_host.hpp"
#include "prof.hpp"

#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                                 \
            return IMMESH_E_HIP;                                                                            \
        }                                                                                                   \
    } while (0)

struct DevBuf {  // grow-only device buffer
    void* p = nullptr;
    size_t bytes = 0;
};

struct immesh_ctx {
    immesh_config cfg;
    std::string err;
    hipStream_t stream = nullptr;
    hipStream_t stream_pre = nullptr;    // the stages before the path (decode / undistort / down-sample): they do not touch the map, so they run beside the previous scan's map update
    hipEvent_t ev_inputs_free = nullptr; // recorded on `stream` when the last asynchronous scan has consumed its input clouds (after point_var + transform)
    hipEvent_t ev_inputs_cur = nullptr;  // the event that currently carries that meaning (the mesher's "scan is in its world buffer" event when one was recorded)
    bool timing_valid[2] = {true, true}; // per event parity: that immesh_process_scan recorded its stage events (synchronous calls only)
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // two sets of four (scan parity)
    int ev_par = 0;
    bool pending = false;            // the last immesh_process_scan returned without waiting for its map update (IMMESH_SCAN_NOWAIT)
    bool tail_deferred = false;      // ... and left the update's tail (free-list merge, counters to the host) to the next scan's residual_persistent_kernel
    float timing[4] = {0, 0, 0, 0};
    std::vector<void*> allocs;
    size_t bytes_allocated = 0;

    // ---- registration map
    RegMapDev map;
    IkdHost ikd;                     // legacy point-map path, allocated on first use
    int64_t* d_stats = nullptr;      // [8] refits, refit pts, ...
    double dvar_beam = 0, dvar_calib = 0;

    // ---- per-scan scratch (sized for cap_scan_points)
    int64_t cap_scan = 0;
    float* d_pts_down = nullptr;     // staging for host inputs (n x 3)
    float* d_pts_raw = nullptr;      // staging (n x 4)
    float* d_ds_out = nullptr;       // immesh_downsample result (n x 3)
    char* h_pack = nullptr; size_t h_pack_bytes = 0;   // immesh_process_scan_strided: pinned staging the host-side strided clouds are packed into (one pass: no second copy by the runtime)
    // immesh_downsample_begin / _end: two result buffers, the grid extents + leaf count of the running job in pinned memory, its parameters for the fallback
    struct DsAsync { bool ready = false; float* stage = nullptr; bool active = false; int par = 0, n = 0, stride = 0, used_bits = 64, pred_bits = 64; double leaf = 0; const void* d_in = nullptr; hipEvent_t ev = nullptr; int32_t ticket = 0; int32_t* h_info = nullptr; int32_t* h_info_dev = nullptr; float* out[2] = {nullptr, nullptr}; } dsa;
    double* d_partials = nullptr;    // residual block partials
    int rp_parity = 0;
    int rp_max_blocks = 127;         // grid cap of residual_persistent_kernel: half of the device's resident workgroups - 1 (occupancy query at create)
    bool rp_force_abort = false;     // IMMESH_RP_FORCE_ABORT (tests): every resident-grid registration gives up in its first gather
    int64_t rp_fallbacks = 0;        // scans registered by the per-pass chain because the resident grid gave up
    // epilogue of the registration launch (map update preparation + full-scan transform): the "scan is in its world buffer / input clouds consumed" flag,
    // 64-bit at d_epi + 2 on the device (the mesher's first kernel polls it) and in pinned memory for the host; stored by the launch queued behind the
    // registration.  epi_seq = last sequence number handed out; inputs_seq != 0: the host-side "input clouds consumed" fence is the pinned flag reaching
    // inputs_seq (instead of ev_inputs_cur)
    unsigned int* d_epi = nullptr;
    unsigned long long* h_epi_flag = nullptr;
    unsigned long long* d_epi_flag_host = nullptr;
    unsigned long long epi_seq = 0, inputs_seq = 0;
    double* d_rp_slots[2] = {nullptr, nullptr};   // residual_persistent_kernel: block-partial slots (per pass x block), one buffer per scan parity
    double* d_out48 = nullptr;
    double* h_out48 = nullptr;       // pinned, device-mapped
    double* d_out48_host = nullptr;  // device view of h_out48
    unsigned int* d_done = nullptr;  // residual_kernel's "blocks finished" counter
    RegState* d_regstate = nullptr;  // device-resident iterate of the scan being registered (in-kernel EKF)
    RegIterArgs reg_args;            // argument block of the residual passes (host staging)
    double* h_reg_out = nullptr;     // pinned, device-mapped: posterior state + counters + ticket written by the pass that stops the loop
    double* d_reg_out_host = nullptr;
    double reg_ticket = 0;
    immesh_allreduce_fn allreduce = nullptr;   // sharded map: sums the per-rank partial normal equations
    void* allreduce_user = nullptr;
    unsigned long long* reg_dbg = nullptr;  // phase timers of residual_kernel (IMMESH_DEBUG)
    unsigned long long res_ticket = 0;  // completion ticket of the last residual launch (polled in h_out48[47])
    int8_t* d_match = nullptr;
    int32_t* d_mnode = nullptr;
    float* d_dis = nullptr;
    double* d_rinv = nullptr;
    double* d_normal = nullptr;
    double* d_ptdata = nullptr;      // n x 9
    unsigned long long *d_key_a = nullptr, *d_key_b = nullptr;
    int32_t *d_idx_a = nullptr, *d_idx_b = nullptr, *d_idx_c = nullptr;
    uint32_t *d_slot = nullptr, *d_slot_s = nullptr;
    int32_t* d_seg_start = nullptr;
    uint32_t* d_touched = nullptr;   // map update: (hash slot, root node) per touched root voxel
    int32_t* d_nseg = nullptr;
    void* d_sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    // scratch of the pre-processing stream (its own copies: the map update uses d_key_a / d_idx_a / d_slot / d_sort_temp concurrently)
    unsigned long long *p_key_a = nullptr, *p_key_b = nullptr;
    int32_t *p_idx_a = nullptr, *p_idx_b = nullptr, *p_idx_c = nullptr, *p_seg = nullptr, *p_nseg = nullptr;
    uint32_t *p_slot = nullptr, *p_slot_s = nullptr;
    float* p_pool4 = nullptr;   // the VoxelGrid's point pool: (x, y, z, scan index) of every point, grouped by leaf
    void* p_sort_temp = nullptr;
    bool ds_skip_hash = false;       // immesh_downsample_end's fall-back: straight to the radix pipeline
    bool ds_gate_ok = false;   // the last scan went through immesh_process_scan's asynchronous path on the resident registration grid: a scan loop -- the next VoxelGrid job is scheduled beside the next registration launch (ds_gate_kernel)
    int32_t* h_ds_info = nullptr; int32_t* d_ds_info = nullptr;   // the VoxelGrid's result words ([0] leaves, [1] fall-back wanted): pinned + device-side address
    DsDyn* h_ds_dyn = nullptr; DsDyn* d_ds_dyn = nullptr;   // the VoxelGrid's per-cloud parameters: pinned host memory + its device-side address
    hipGraphExec_t ds_graph = nullptr;                      // immesh_downsample_begin's launch sequence, captured once
    void* p_htab = nullptr; unsigned long long p_htab_cap = 0;   // VoxelGrid leaf table (16-byte entries, all empty between calls)
    char* d_raw_stage = nullptr;     // sensor decode: staging for wire-format clouds handed over as host memory (cap_scan x 64 B, first use)
    float *d_und_in = nullptr, *d_und_out = nullptr; double* d_und_tab = nullptr;   // immesh_undistort staging: n x 5 in, n x 4 out, pose table
    int32_t* d_counters_host = nullptr;   // device view of h_counters
    int32_t* h_counters = nullptr;   // pinned copy of map.counters (8 ints)
    unsigned long long* d_dump_count = nullptr;

    KProf prof;
    void* rccl_comm = nullptr;       // ncclComm_t of a sharded context after immesh_rccl_init (comm_rccl.cpp); null: host callbacks
    float* d_bcast[4] = {nullptr, nullptr, nullptr, nullptr};   // immesh_broadcast_scan: the scan as every rank holds it ([2 parity + 0] stride 3, [2 parity + 1] stride 4; cap_scan points each, first use; two parities used in turn)
    int bcast_parity = 0;
    int32_t* d_bcast_hdr = nullptr;           // ... and its headers: this rank's 16 bytes {points, stride, cap_scan, -}, then every rank's (all-gather)
    std::atomic<int64_t> rccl_calls{0};   // (counted from the scan thread and from the mesher's worker thread)

    // cumulative counters (host side)
    immesh_counters_t cnt;
    int last_n_ds = 0;
    const float* last_reg_pts = nullptr;   // device cloud of the last registration (immesh_last_matches)

    // ---- mesher
    MeshDev mesh;
    MeshHost mesh_host;

    template <typename T>
    int dalloc(T** out, size_t count) {
        void* p = nullptr;
        const size_t bytes = count * sizeof(T);
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) { err = "hipMalloc(" + std::to_string(bytes) + " B): " + hipGetErrorString(e); return IMMESH_E_NOMEM; }
        allocs.push_back(p);
        bytes_allocated += bytes;
        *out = (T*)p;
        return 0;
    }
};

// resolve an input pointer that may be host or device memory; host data is staged into `staging` on the ctx stream
inline int resolve_input(immesh_ctx* c, const void* p, size_t bytes, void* staging, const void** dev_out) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    bool is_dev = false;
    if (e == hipSuccess) is_dev = (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
    else (void)hipGetLastError();  // plain host pointer: clear the sticky error
    if (is_dev) { *dev_out = p; return 0; }
    HIPCHK(c, hipMemcpyAsync(staging, p, bytes, hipMemcpyHostToDevice, c->stream));
    *dev_out = staging;
    return 0;
}

struct ProfBind {  // binds the ctx profiler to the calling thread for the duration of one C-ABI call
    immesh_ctx* c;
    explicit ProfBind(immesh_ctx* ctx) : c(ctx) { g_kprof = &ctx->prof; }
    ~ProfBind() { if (c->prof.on && c->stream) { (void)hipStreamSynchronize(c->stream); if (c->stream_pre) (void)hipStreamSynchronize(c->stream_pre); c->prof.flush(); } g_kprof = nullptr; }
};

// RCCL inside the library (comm_rccl.cpp)
int rccl_allreduce_f64(immesh_ctx* c, double* dev_buf, size_t n, hipStream_t s);
int rccl_allgather_bytes(immesh_ctx* c, const void* dev_send, void* dev_recv, size_t bytes_per_rank, hipStream_t s, std::string* err);
void rccl_release(immesh_ctx* c);

// mesher host orchestration (mesh_host.cpp)
int mesh_alloc(immesh_ctx* c);
void mesh_free(immesh_ctx* c);
long mesh_submit(immesh_ctx* c, const float* d_pts_world_xyzi, int n_raw, const double* sensor_pos, int frame_idx, bool ready_recorded = false,
                 const unsigned long long* wait_flag = nullptr, unsigned long long wait_seq = 0);   // wait_flag: the scan's producer stores wait_seq there (no event)
hipEvent_t mesh_record_ready(immesh_ctx* c);   // record the NEXT job's "scan is in its world buffer" event on the registration stream now (before more work is queued behind it)
float* mesh_next_world_buffer(immesh_ctx* c);
int mesh_wait(immesh_ctx* c, long id);
void mesh_wait_all(immesh_ctx* c);
int mesh_transform_full(immesh_ctx* c, const float* d_pts_raw_xyzi, float* d_world, int n_raw, const imh::State& st);
void mesh_counters(immesh_ctx* c, immesh_counters_t* out);
void mesh_counters_reset(immesh_ctx* c);
