// RCCL inside the library: the data-path collectives of a sharded context are issued by the C++ layer itself on the
// context's own HIP streams, on device-resident buffers -- no host staging, no callback into the application:
//   * registration map sharded by root-voxel bricks: ncclAllReduce(sum) of the 46 doubles of a residual pass (H^T R^-1 H, H^T R^-1 z,
//     counters) between the pass and the in-kernel 18-state update (ekf_step_kernel); the whole scan is enqueued without a host round trip;
//   * sharded mesher: ncclAllGather of the exchange records (smoothed boundary-band vertices, triangle marks).
// librccl.so is opened at immesh_rccl_init: a single-GPU process never loads it.  xGMI note: every message here is <= a few hundred KB, so
// the collectives are latency-bound (one ring step per peer over point-to-point links), never link-bound.
#include <algorithm>
#include "host_ctx.hpp"
#include <dlfcn.h>
#include <cstdint>

namespace {
// the handful of RCCL entry points used (signatures of rccl/rccl.h; ncclDataType_t / ncclRedOp_t passed as their integer values)
typedef struct { char internal[128]; } rcclUniqueId;
typedef int (*fn_get_unique_id)(rcclUniqueId*);
typedef int (*fn_comm_init_rank)(void**, int, rcclUniqueId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);
struct RcclApi {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_broadcast broadcast = nullptr;
    fn_error_string error_string = nullptr;
    std::string err;
};
RcclApi g_rccl;
enum { RCCL_INT8 = 0, RCCL_INT32 = 2, RCCL_FLOAT64 = 8, RCCL_SUM = 0 };   // ncclInt8 / ncclInt32 / ncclFloat64, ncclSum

bool rccl_load() {
    if (g_rccl.lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) { g_rccl.err = std::string("dlopen(librccl.so): ") + dlerror(); return false; }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.all_gather = (fn_all_gather)dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.broadcast = (fn_broadcast)dlsym(g_rccl.lib, "ncclBroadcast");
    g_rccl.error_string = (fn_error_string)dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce || !g_rccl.all_gather || !g_rccl.broadcast) {
        g_rccl.err = "librccl.so lacks an expected entry point"; dlclose(g_rccl.lib); g_rccl.lib = nullptr; return false;
    }
    return true;
}
std::string rccl_why(int rc) { return g_rccl.error_string ? std::string(g_rccl.error_string(rc)) : std::to_string(rc); }
}  // namespace

// Stubbed collectives (immesh_stub_collectives): ONE rank of a sharded job runs alone -- the all-reduce leaves this rank's partial sums as they are, the
// all-gather delivers only this rank's own block (the others read as zero records).  What it is for: sizing and timing one rank's share of a map that
// needs a node (configs[4]) on a single GPU; the results are those of the sub-problem this rank sees, not of the job.
static void* const RCCL_STUB = (void*)(uintptr_t)1;
int rccl_allreduce_f64(immesh_ctx* c, double* dev_buf, size_t n, hipStream_t s) {
    if (c->rccl_comm == RCCL_STUB) { c->rccl_calls++; return 0; }
    const int rc = g_rccl.all_reduce(dev_buf, dev_buf, n, RCCL_FLOAT64, RCCL_SUM, c->rccl_comm, s);
    if (rc) { c->err = "ncclAllReduce: " + rccl_why(rc); return IMMESH_E_HIP; }
    c->rccl_calls++;
    return 0;
}
int rccl_allgather_bytes(immesh_ctx* c, const void* dev_send, void* dev_recv, size_t bytes_per_rank, hipStream_t s, std::string* err) {
    if (c->rccl_comm == RCCL_STUB) {
        const int world = c->cfg.shard_world > 1 ? c->cfg.shard_world : 1, me = c->cfg.shard_world > 1 ? c->cfg.shard_rank : 0;
        if (hipMemsetAsync(dev_recv, 0, bytes_per_rank * (size_t)world, s) != hipSuccess ||
            hipMemcpyAsync((char*)dev_recv + bytes_per_rank * (size_t)me, dev_send, bytes_per_rank, hipMemcpyDeviceToDevice, s) != hipSuccess) { *err = "stubbed all-gather: copy failed"; return IMMESH_E_HIP; }
        c->rccl_calls++;
        return 0;
    }
    const int rc = g_rccl.all_gather(dev_send, dev_recv, bytes_per_rank, RCCL_INT8, c->rccl_comm, s);
    if (rc) { *err = "ncclAllGather: " + rccl_why(rc); return IMMESH_E_HIP; }
    c->rccl_calls++;
    return 0;
}
void rccl_release(immesh_ctx* c) {
    if (c->rccl_comm && c->rccl_comm != RCCL_STUB && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->rccl_comm);
    c->rccl_comm = nullptr;
}

extern "C" {

int immesh_rccl_unique_id(uint8_t id_out[128]) {
    if (!id_out) return IMMESH_E_INVAL;
    if (!rccl_load()) return IMMESH_E_NODEV;
    rcclUniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc) { g_rccl.err = "ncclGetUniqueId: " + rccl_why(rc); return IMMESH_E_HIP; }
    std::memcpy(id_out, id.internal, 128);
    return 0;
}

int immesh_rccl_init(immesh_ctx* c, const uint8_t id_in[128]) {
    if (!c || !id_in) return IMMESH_E_INVAL;
    if (c->cfg.shard_world < 1) { c->err = "immesh_rccl_init: the context is not sharded (shard_world)"; return IMMESH_E_INVAL; }
    if (!rccl_load()) { c->err = g_rccl.err; return IMMESH_E_NODEV; }
    (void)hipSetDevice(c->cfg.device);
    mesh_wait_all(c);
    rccl_release(c);
    rcclUniqueId id;
    std::memcpy(id.internal, id_in, 128);
    const int world = c->cfg.shard_world > 1 ? c->cfg.shard_world : 1, rank = c->cfg.shard_world > 1 ? c->cfg.shard_rank : 0;
    if (world > 64) { c->err = "immesh_rccl_init: more than 64 ranks (the mesher's count exchange is sized for one node)"; return IMMESH_E_INVAL; }
    // the count buffer of the mesher's exchanges: allocated here, on the calling thread with the worker idle (the allocator is not shared with it)
    if (!c->mesh_host.d_xcounts) { const int arc = c->dalloc(&c->mesh_host.d_xcounts, 128); if (arc) return arc; }
    const int rc = g_rccl.comm_init_rank(&c->rccl_comm, world, id, rank);
    if (rc) { c->rccl_comm = nullptr; c->err = "ncclCommInitRank: " + rccl_why(rc); return IMMESH_E_HIP; }
    c->allreduce = nullptr; c->mesh_host.allgather = nullptr;   // the library's own collectives replace the host callbacks
    return 0;
}

// SURVEY 8(e), first row: "Scan (<= 0.5 M x 12 B = 6 MB) broadcast once per scan" -- the library distributes the scan, not the harness.  Collective over
// the ranks of the sharded job, on the context's registration stream (whatever is enqueued there next -- immesh_process_scan -- is ordered behind it):
//   RCCL     a 16-byte header {points, stride} (ncclBroadcast + one host read: the payload's size is a host-side argument of the collective), then
//            ncclBroadcast of the points, device to device;
//   callbacks (gloo tests)   the same two messages through the registered all-gather (the root's block carries the data, the others' are ignored).
int immesh_broadcast_scan(immesh_ctx* c, const float* pts, int32_t n, int32_t stride, int32_t root, const float** dev_out, int32_t* n_out) {
    if (!c || !dev_out || !n_out) return IMMESH_E_INVAL;
    const int world = c->cfg.shard_world > 1 ? c->cfg.shard_world : 1, me = c->cfg.shard_world > 1 ? c->cfg.shard_rank : 0;
    // (local argument errors every rank would see alike -- the same call on every rank -- may return before the collective)
    if (root < 0 || root >= world) { c->err = "immesh_broadcast_scan: root outside the job"; return IMMESH_E_INVAL; }
    const bool stub = c->rccl_comm == RCCL_STUB;
    // stubbed collectives (one process standing in for one rank of the job, bench.py's dry run): nobody sends, so EVERY rank must be handed the scan
    const bool am_root = me == root || stub;
    const bool use_rccl = c->rccl_comm && !stub;
    if (world > 1 && !use_rccl && !stub && !c->mesh_host.allgather) { c->err = "immesh_broadcast_scan: no collective registered (immesh_rccl_init or immesh_set_allgather)"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    hipStream_t s = c->stream;
    // two buffers per layout, used in turn: the buffer handed out by broadcast k is written again by broadcast k + 2 -- a mesh job that reads scan k
    // asynchronously (immesh_mesh_scan / an asynchronous immesh_process_scan on it) has one whole scan of slack; the header says so
    for (int k = 0; k < 4; k++)
        if (!c->d_bcast[k]) { const int arc = c->dalloc(&c->d_bcast[k], (size_t)c->cap_scan * ((k & 1) ? 4 : 3)); if (arc) return arc; }
    if (!c->d_bcast_hdr) { const int arc = c->dalloc(&c->d_bcast_hdr, 4 * 65); if (arc) return arc; }
    // ---- header: an ALL-GATHER every rank takes part in whatever it was handed (a root that returned on a bad scan before the collective left
    // the other ranks blocked in it; a rank too small for the scan returned while the root went on to the payload).  Each rank contributes
    // {points (root: n, or -1 for an unusable scan; others 0), floats per point, its cap_scan_points, 0}; all ranks then take the SAME decision.
    const bool root_ok = pts && n > 0 && n <= c->cap_scan && (stride == 3 || stride == 4);
    int32_t hdr[4] = {me == root || stub ? (root_ok ? n : -1) : 0, me == root || stub ? stride : 0, (int32_t)std::min<int64_t>(c->cap_scan, 0x7fffffff), 0};
    std::vector<int32_t> all((size_t)world * 4, 0);
    if (world == 1 || stub) { for (int r = 0; r < world; r++) std::memcpy(&all[(size_t)r * 4], hdr, 16); }
    else if (use_rccl) {
        HIPCHK(c, hipMemcpyAsync(c->d_bcast_hdr, hdr, 16, hipMemcpyHostToDevice, s));
        std::string err;
        const int rc = rccl_allgather_bytes(c, c->d_bcast_hdr, c->d_bcast_hdr + 4, 16, s, &err);
        if (rc) { c->err = err; return rc; }
        HIPCHK(c, hipMemcpyAsync(all.data(), c->d_bcast_hdr + 4, (size_t)world * 16, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    } else {
        if (c->mesh_host.allgather(hdr, 16, all.data(), c->mesh_host.allgather_user)) { c->err = "all-gather callback failed"; return IMMESH_E_INVAL; }
    }
    const int32_t np = all[(size_t)root * 4], st = all[(size_t)root * 4 + 1];
    if (np <= 0 || (st != 3 && st != 4)) { c->err = "immesh_broadcast_scan: the root's scan is empty, larger than its cap_scan_points or not 3 / 4 floats per point"; return IMMESH_E_INVAL; }
    for (int r = 0; r < world; r++)
        if (all[(size_t)r * 4 + 2] < np) { c->err = "immesh_broadcast_scan: the scan exceeds the cap_scan_points of rank " + std::to_string(r); return IMMESH_E_CAPACITY; }
    // ---- payload
    const int par = (c->bcast_parity ^= 1);
    float* buf = c->d_bcast[2 * par + (st == 4 ? 1 : 0)];
    const size_t bytes = (size_t)np * st * sizeof(float);
    if (am_root) {
        hipPointerAttribute_t attr;
        const bool dev_in = hipPointerGetAttributes(&attr, pts) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
        if (!dev_in) (void)hipGetLastError();
        if ((const void*)pts != (const void*)buf) HIPCHK(c, hipMemcpyAsync(buf, pts, bytes, dev_in ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    }
    if (use_rccl) {
        const int rc = g_rccl.broadcast(buf, buf, bytes, RCCL_INT8, root, c->rccl_comm, s);
        if (rc) { c->err = "ncclBroadcast: " + rccl_why(rc); return IMMESH_E_HIP; }
        c->rccl_calls++;
    } else if (world > 1 && !stub) {
        std::vector<char> send(bytes, 0), gathered((size_t)world * bytes);
        if (am_root) { HIPCHK(c, hipMemcpyAsync(send.data(), buf, bytes, hipMemcpyDeviceToHost, s)); HIPCHK(c, hipStreamSynchronize(s)); }
        if (c->mesh_host.allgather(send.data(), (int64_t)bytes, gathered.data(), c->mesh_host.allgather_user)) { c->err = "all-gather callback failed"; return IMMESH_E_INVAL; }
        if (!am_root) { HIPCHK(c, hipMemcpyAsync(buf, gathered.data() + (size_t)root * bytes, bytes, hipMemcpyHostToDevice, s)); HIPCHK(c, hipStreamSynchronize(s)); }
    }
    *dev_out = buf; *n_out = np;
    return 0;
}

int immesh_stub_collectives(immesh_ctx* c) {
    if (!c) return IMMESH_E_INVAL;
    if (c->cfg.shard_world < 1) { c->err = "immesh_stub_collectives: the context is not sharded (shard_world)"; return IMMESH_E_INVAL; }
    const int world = c->cfg.shard_world > 1 ? c->cfg.shard_world : 1;
    if (world > 64) { c->err = "immesh_stub_collectives: more than 64 ranks"; return IMMESH_E_INVAL; }
    (void)hipSetDevice(c->cfg.device);
    mesh_wait_all(c);
    rccl_release(c);
    if (!c->mesh_host.d_xcounts) { const int arc = c->dalloc(&c->mesh_host.d_xcounts, 128); if (arc) return arc; }
    c->rccl_comm = RCCL_STUB;
    c->allreduce = nullptr; c->mesh_host.allgather = nullptr;
    return 0;
}
int immesh_device_bytes(immesh_ctx* c, int64_t* bytes) {
    if (!c || !bytes) return IMMESH_E_INVAL;
    *bytes = (int64_t)c->bytes_allocated;
    return 0;
}

const char* immesh_rccl_error(void) { return g_rccl.err.c_str(); }

}  // extern "C"
