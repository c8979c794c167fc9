// pcl::VoxelGrid down-sampling on the device: the stage immediately before the hot path
// (src/voxel_mapping.cpp:1715, :1888-1891).  Spec = the harness's voxel_grid_downsample (immesh_amd/synth.py), which both the CPU checker and
// the GPU path consume: leaf index ijk = floor(p * inv_leaf) - floor(min * inv_leaf) in float32, linear index i + j*dx + k*dx*dy, points
// ordered by index (stable), one output point per occupied leaf = float32 centroid accumulated in point order, output ordered by index.
//   ds_minmax_kernel   per-axis floor(min) / floor(max) of the cloud (block reduction + integer atomics)
//   ds_index_kernel    linear leaf index per point
//   (stable radix sort of (index, point) pairs: rocPRIM)
//   ds_heads_kernel    run heads -> 0/1 flags        (exclusive scan: rocPRIM)
//   ds_centroid_kernel one thread per run head: sequential float32 sum of its run, in order -> bit-identical to the CPU spec
// Bound: HBM streaming (12-16 B per point per pass); the sort dominates.
#include <algorithm>
#include "kernels.hpp"
#include "prof.hpp"
#include "dev_math.hpp"
using namespace imd;

__global__ __launch_bounds__(256) void ds_minmax_kernel(const float* __restrict__ pts, int n, int stride, float inv, int* __restrict__ mm /*[6]: min xyz, max xyz*/) {
    __shared__ int smin[3], smax[3];
    if (threadIdx.x < 3) { smin[threadIdx.x] = 0x7FFFFFFF; smax[threadIdx.x] = (int)0x80000000; }
    __syncthreads();
    int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int c = (int)floorf(pts[(size_t)i * stride + a] * inv);   // floor is monotone: floor(min * inv) == min over points of floor(p * inv)
            lo[a] = min(lo[a], c); hi[a] = max(hi[a], c);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) { atomicMin(&smin[a], lo[a]); atomicMax(&smax[a], hi[a]); }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&mm[threadIdx.x], smin[threadIdx.x]); atomicMax(&mm[3 + threadIdx.x], smax[threadIdx.x]); }
}

__global__ __launch_bounds__(256) void ds_index_kernel(const float* __restrict__ pts, int n, int stride, float inv, const int* __restrict__ mm,
                                                        unsigned long long* __restrict__ keys, int32_t* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long dx = (long long)mm[3] - mm[0] + 1, dy = (long long)mm[4] - mm[1] + 1;
    const long long ix = (long long)floorf(pts[(size_t)i * stride + 0] * inv) - mm[0];
    const long long iy = (long long)floorf(pts[(size_t)i * stride + 1] * inv) - mm[1];
    const long long iz = (long long)floorf(pts[(size_t)i * stride + 2] * inv) - mm[2];
    keys[i] = (unsigned long long)(ix + iy * dx + iz * dx * dy);
    vals[i] = i;
}

__global__ __launch_bounds__(256) void ds_heads_kernel(const unsigned long long* __restrict__ keys_sorted, int n, int32_t* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys_sorted[i] != keys_sorted[i - 1]) ? 1 : 0;
}

// One wavefront per tile of 64 sorted points.  The wave owns the runs (= leaves) whose first point lies in its tile and follows the last of them
// past the tile end if it must.  A batch of 64 points is fetched by the 64 lanes at once; the float32 sums are then formed strictly in point
// order by a wave-uniform loop over the lanes (v_readlane), i.e. the same sequential `centroid += pt` as the CPU spec, with all memory latency
// taken once per 64 points instead of once per point.
__global__ __launch_bounds__(256) void ds_centroid_kernel(const float* __restrict__ pts, int n, int stride, const unsigned long long* __restrict__ keys_sorted,
                                                           const int32_t* __restrict__ idx_sorted, const int32_t* __restrict__ rank, float* __restrict__ out,
                                                           int32_t* __restrict__ n_out) {
    const int lane = threadIdx.x & 63;
    const int tile = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((long long)tile * 64 >= n) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int cnt = 0, orank = -1;
    bool open = false;
    // software pipeline: the next batch's gathers are in flight while this batch is summed
    unsigned long long k = ~0ull, kprev = ~0ull;
    float x = 0.f, y = 0.f, z = 0.f;
    int rk = 0;
    auto fetch = [&](int base, unsigned long long& fk, unsigned long long& fkp, float& fx, float& fy, float& fz, int& frk) {
        const int i = base + lane;
        fk = ~0ull; fkp = ~0ull; fx = fy = fz = 0.f; frk = 0;
        if (i < n) {
            fk = keys_sorted[i];
            fkp = i > 0 ? keys_sorted[i - 1] : ~0ull;
            const int p = idx_sorted[i];
            fx = pts[(size_t)p * stride + 0]; fy = pts[(size_t)p * stride + 1]; fz = pts[(size_t)p * stride + 2];
            frk = rank[i];
        }
    };
    fetch(tile * 64, k, kprev, x, y, z, rk);
    for (int b = 0;; b++) {
        const int base = tile * 64 + b * 64;
        if (base >= n) break;
        const int i = base + lane;
        unsigned long long nk, nkp; float nx, ny, nz; int nrk;
        fetch(base + 64, nk, nkp, nx, ny, nz, nrk);
        const unsigned long long valid = __ballot(i < n);
        const unsigned long long heads = __ballot(i < n && (i == 0 || k != kprev));
        bool stop = false;
        for (int l = 0; l < 64; l++) {
            if (!((valid >> l) & 1ull)) break;
            const bool hd = (heads >> l) & 1ull;
            if (hd && b > 0) { stop = true; break; }            // the next run starts in another wave's tile
            const float xl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
            const float yl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), l));
            const float zl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), l));
            if (hd) {
                if (open && lane == 0) { const float c = (float)cnt; out[(size_t)orank * 3 + 0] = sx / c; out[(size_t)orank * 3 + 1] = sy / c; out[(size_t)orank * 3 + 2] = sz / c; }
                open = true; orank = __builtin_amdgcn_readlane(rk, l);
                sx = 0.f + xl; sy = 0.f + yl; sz = 0.f + zl; cnt = 1;
            } else if (open) { sx += xl; sy += yl; sz += zl; cnt++; }
        }
        if (stop || !open) break;
        k = nk; kprev = nkp; x = nx; y = ny; z = nz; rk = nrk;
    }
    if (open && lane == 0) {
        const float c = (float)cnt;
        out[(size_t)orank * 3 + 0] = sx / c; out[(size_t)orank * 3 + 1] = sy / c; out[(size_t)orank * 3 + 2] = sz / c;
    }
    // the run containing the last point is the last leaf
    if (lane == 0 && (long long)tile * 64 <= (long long)n - 1 && (long long)tile * 64 + 64 > (long long)n - 1) {
        const int i = n - 1;
        const int last_head_rank = rank[i] + ((i == 0 || keys_sorted[i] != keys_sorted[i - 1]) ? 1 : 0);   // exclusive scan + own flag
        *n_out = last_head_rank;
    }
}

// =====================================================================================================================
// The same VoxelGrid in FIVE short launches (+ a one-thread gate and a one-thread publish) instead of fifteen (round 4; the radix pipeline above stays as the fall-back and for clouds whose leaves do
// not fit the single-key form).  The order of the output -- ascending linear leaf index i + j*dx + k*dx*dy -- is the lexicographic order of (k, j, i),
// and that needs neither the bounding box nor a sort of the POINTS:
//   ds_hash_kernel       one thread per point: leaf cell (floor(p * inv), the spec's float32 arithmetic) -> packed (k, j, i) key -> find-or-create in an
//                        open-addressing table; the leaf's point count grows by one atomic; the creator of an entry appends its slot to the leaf list.
//                        Nobody waits for anybody.  (count and segment offset share ONE 64-bit word of the entry)
//   ds_leaf_sort_kernel  the occupied leaves (~8 k for a 100 k-point scan, not 100 k points) sorted by key in chunks of 512 (bitonic network in LDS, one
//                        workgroup per chunk), and a segment of the point pool reserved for every leaf (scan of the counts inside the chunk, one
//                        atomic per chunk for its base: the pool's order does not matter)
//   ds_scatter_kernel    one thread per point: the point (x, y, z, scan index) into its leaf's segment (one atomic: arrival order, i.e. no order); then every leaf's output
//                        position (own position in its chunk + lower bounds in the other chunks, one lane per chunk)
//   ds_leaf_emit_kernel  one wavefront per leaf: its segment ordered by scan index, the float32 sums formed strictly in that order -- the spec's
//                        sequential `centroid += pt` -- and its table entry handed back empty: the table is never cleared as a whole.
// Bound: latency (one hash round trip per point in ds_hash_kernel, one returning atomic + one 16-byte write in ds_scatter_kernel); HBM traffic ~ 3 x 16 B
// per point + the leaves' lines.
// =====================================================================================================================
#define DSH_EMPTY 0xFFFFFFFFFFFFFFFFull
#define DSH_BIAS (1 << 20)
#define DSH_CHUNK 512
#define DSH_LEAF_CAP 2048          /* points of one leaf ordered in LDS; a leaf holding more sets the fall-back flag */
struct DsEnt { unsigned long long key; int cnt; int off; };   // 16 B; key == DSH_EMPTY: free (then cnt == 0).  (cnt, off) = one 64-bit word for ds_scatter_kernel's atomic
// What changes from cloud to cloud lives in PINNED host memory (DsDyn, kernels.hpp): thread 0 of every workgroup reads it (one PCIe round trip per
// workgroup) -- so the kernel arguments never change and the whole sequence (gate, five working launches, publish: the counters are handed back zeroed and
// the result goes to pinned memory by kernels, not by memset / copy nodes) is ONE hipGraph: one API call per cloud on the scan thread.
#define DS_DYN(dynp) __shared__ DsDyn s_dyn_; if (threadIdx.x == 0) s_dyn_ = *(dynp); __syncthreads(); const DsDyn& dyn = s_dyn_
// First launch of the asynchronous sequence: one thread that waits until the registration launch named by the job has started (DsDyn::gate_word) -- or
// for DS_GATE_TICKS, whichever comes first: it is a scheduling hint, nothing depends on it.  Measured (round 4, meshing off, period 145 us without the
// VoxelGrid): enqueued right behind a scan's pose the sequence ran beside that scan's MAP UPDATE and stretched its two kernels by 20 + 30 us (period
// 203 us); beside the next scan's registration launch it stretches that one by 30 us and the update by 10 (period 184 us).  Whatever runs beside these
// 100 k random accesses pays in memory latency (CU masks, stream priorities and smaller grids change nothing: profiles/README.md).
#define DS_GATE_TICKS 15000ull   /* 150 us of s_memrealtime (100 MHz) */
__global__ void ds_gate_kernel(const DsDyn* __restrict__ dynp) {
    const int32_t* const w = dynp->gate_word;
    if (!w) return;
    const int want = dynp->gate_val;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > DS_GATE_TICKS) break;
        __builtin_amdgcn_s_sleep(32);
    }
}
void launch_ds_gate(hipStream_t s, const DsDyn* dyn) { KLAUNCH(ds_gate_kernel, dim3(1), dim3(1), 0, s, dyn); }
// info: [0] leaves, [1] fall-back wanted (cell out of the key's range / table full / a leaf above DSH_LEAF_CAP points), [2] pool fill, [3] big leaves
__global__ __launch_bounds__(256) void ds_hash_kernel(const DsDyn* __restrict__ dynp, DsEnt* __restrict__ tab, unsigned long long mask,
                                                       int32_t* __restrict__ pt_slot, int32_t* __restrict__ leaf_slot, int32_t* __restrict__ info) {
    DS_DYN(dynp);
    const float* __restrict__ pts = dyn.pts;
    const int n = dyn.n, stride = dyn.stride;
    const float inv = dyn.inv;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = pts + (size_t)i * stride;
    const long long cx = (long long)floorf(p[0] * inv), cy = (long long)floorf(p[1] * inv), cz = (long long)floorf(p[2] * inv);
    if (cx < -DSH_BIAS || cx >= DSH_BIAS || cy < -DSH_BIAS || cy >= DSH_BIAS || cz < -DSH_BIAS || cz >= DSH_BIAS) { info[1] = 1; pt_slot[i] = -1; continue; }
    const unsigned long long key = ((unsigned long long)(cz + DSH_BIAS) << 42) | ((unsigned long long)(cy + DSH_BIAS) << 21) | (unsigned long long)(cx + DSH_BIAS);
    unsigned long long h = hash64(key) & mask;
    int slot = -1;
    for (int probe = 0; probe < 4096; probe++) {
        unsigned long long k = tab[h].key;
        if (k == DSH_EMPTY) {
            k = atomicCAS(&tab[h].key, (unsigned long long)DSH_EMPTY, key);
            if (k == DSH_EMPTY) { leaf_slot[atomicAdd(&info[0], 1)] = (int)h; k = key; }
        }
        if (k == key) { atomicAdd(&tab[h].cnt, 1); slot = (int)h; break; }
        h = (h + 1) & mask;
    }
    if (slot < 0) info[1] = 1;
    pt_slot[i] = slot;
    }
}
__global__ __launch_bounds__(256) void ds_leaf_sort_kernel(DsEnt* __restrict__ tab, const int32_t* __restrict__ leaf_slot, int32_t* __restrict__ info,
                                                            unsigned long long* __restrict__ keys_sorted, int32_t* __restrict__ slots_sorted, int32_t* __restrict__ big_list) {
    __shared__ unsigned long long sk[DSH_CHUNK];
    __shared__ int sv[DSH_CHUNK];
    __shared__ unsigned short sc[DSH_CHUNK];   // the leaf's point count, capped (only "above 64?" is needed behind the sort)
    __shared__ int s_scan[256];
    __shared__ int s_base;
    const int nleaf = info[0];
    const int tid = threadIdx.x;
    for (int first = blockIdx.x * DSH_CHUNK; first < nleaf; first += gridDim.x * DSH_CHUNK) {
    const int cnt = min(DSH_CHUNK, nleaf - first);
    int np2 = 1; while (np2 < cnt) np2 <<= 1;
    // a segment of the point pool for each of the chunk's leaves (two leaves per thread)
    int c0 = 0, c1 = 0, sl0 = -1, sl1 = -1;
    unsigned long long k0 = ~0ull, k1 = ~0ull;
    if (2 * tid < cnt) { sl0 = leaf_slot[first + 2 * tid]; const DsEnt e = tab[sl0]; k0 = e.key; c0 = e.cnt; }
    if (2 * tid + 1 < cnt) { sl1 = leaf_slot[first + 2 * tid + 1]; const DsEnt e = tab[sl1]; k1 = e.key; c1 = e.cnt; }
    if (c0 > DSH_LEAF_CAP || c1 > DSH_LEAF_CAP) info[1] = 1;
    s_scan[tid] = c0 + c1;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) { const int v = tid >= off ? s_scan[tid - off] : 0; __syncthreads(); s_scan[tid] += v; __syncthreads(); }
    if (tid == 255) s_base = atomicAdd(&info[2], s_scan[255]);
    __syncthreads();
    const int excl = s_base + s_scan[tid] - (c0 + c1);
    if (sl0 >= 0) tab[sl0].off = excl;
    if (sl1 >= 0) tab[sl1].off = excl + c0;
    for (int k = tid; k < np2; k += 256) { sk[k] = ~0ull; sv[k] = -1; sc[k] = 0; }
    __syncthreads();
    if (sl0 >= 0) { sk[2 * tid] = k0; sv[2 * tid] = sl0; sc[2 * tid] = (unsigned short)min(c0, 65535); }
    if (sl1 >= 0) { sk[2 * tid + 1] = k1; sv[2 * tid + 1] = sl1; sc[2 * tid + 1] = (unsigned short)min(c1, 65535); }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int lj = 31 - __clz(k >> 1); lj >= 0; lj--) {
            const int j = 1 << lj;
            for (int p = tid; p < (np2 >> 1); p += 256) {
                const int a = ((p >> lj) << (lj + 1)) | (p & (j - 1)), b = a + j;
                const bool up = ((a & k) == 0);
                const unsigned long long x = sk[a], y = sk[b];
                if ((x > y) == up) { sk[a] = y; sk[b] = x; const int t = sv[a]; sv[a] = sv[b]; sv[b] = t; const unsigned short u = sc[a]; sc[a] = sc[b]; sc[b] = u; }
            }
            __syncthreads();
        }
    for (int k = tid; k < cnt; k += 256) {
        const bool big = sc[k] > 64;   // leaves above one wavefront's worth of points: marked in the sign bit of their slot and listed for the long-leaf wavefronts
        keys_sorted[first + k] = sk[k]; slots_sorted[first + k] = big ? (int)((unsigned int)sv[k] | 0x80000000u) : sv[k];
        if (big) big_list[atomicAdd(&info[3], 1)] = first + k;
    }
    __syncthreads();
    }
}
__global__ __launch_bounds__(256) void ds_scatter_kernel(const DsDyn* __restrict__ dynp, DsEnt* __restrict__ tab, const int32_t* __restrict__ pt_slot, float4* __restrict__ pool4,
                                                          const unsigned long long* __restrict__ keys_sorted, const int32_t* __restrict__ info, int32_t* __restrict__ leaf_rank) {
    DS_DYN(dynp);
    // the sequence has given up already (a cell outside the key's range, a full table, a leaf above DSH_LEAF_CAP points -- flagged by the two launches in
    // front): nothing of this launch's work will be looked at, and the counts below are only sure to stay inside their fields when no leaf is above the cap
    if (info[1]) return;
    const float* __restrict__ pts = dyn.pts;
    const int n = dyn.n, stride = dyn.stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int slot = pt_slot[i];
    if (slot < 0) continue;
    const float* p = pts + (size_t)i * stride;
    const float x = p[0], y = p[1], z = p[2];
    // ONE returning atomic on the entry's (count, segment offset) word: low 16 bits of the count = points still to place (<= DSH_LEAF_CAP), high 16 =
    // points placed; the offset rides back in the upper half (a second random read per point otherwise).  The point itself goes into the pool, not its
    // index: ds_leaf_emit_kernel reads a leaf's segment as consecutive 16-byte records instead of gathering 100 k points one line each
    const unsigned long long old = atomicAdd((unsigned long long*)&tab[slot].cnt, 0xFFFFull);
    const int left = (int)(old & 0xFFFFull), off = (int)(old >> 32);
    if (left > 0 && left <= DSH_LEAF_CAP) pool4[off + left - 1] = make_float4(x, y, z, __int_as_float(i));
    }
    // ---- the output position of every leaf (its rank among all keys): own position in its chunk + the number of smaller keys in every other chunk
    // (keys are unique).  L lanes per leaf, one lane per chunk -- nine dependent reads that used to head every wavefront of ds_leaf_emit_kernel; here
    // they ride behind the scatter's single round trip
    const int nleaf = info[0];
    const int nchunks = (nleaf + DSH_CHUNK - 1) / DSH_CHUNK;
    int L = 1;
    while (L < nchunks && L < 64) L <<= 1;
    const int per_wave = 64 / L, lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const int sub = lane / L, sl = lane % L;
    for (int e0 = gw * per_wave; e0 < nleaf; e0 += nw * per_wave) {
        const int e = e0 + sub;
        const bool live = e < nleaf;
        const unsigned long long key = live ? keys_sorted[e] : 0ull;
        const int own = e / DSH_CHUNK;
        int lo = 0;
        for (int c = sl; c < nchunks; c += L) {
            int l = 0, hgh = (live && c != own) ? min(DSH_CHUNK, nleaf - c * DSH_CHUNK) : 0;
            while (l < hgh) { const int mid = (l + hgh) >> 1; if (keys_sorted[(size_t)c * DSH_CHUNK + mid] < key) l = mid + 1; else hgh = mid; }
            lo += l;
        }
        for (int off = L >> 1; off > 0; off >>= 1) lo += __shfl_xor(lo, off, 64);
        if (live && sl == 0) leaf_rank[e] = e % DSH_CHUNK + lo;
    }
}
// One launch for both kinds of leaves: wavefront 0 of the first DSH_BIG_BLOCKS workgroups takes the leaves of 65 .. DSH_LEAF_CAP points (the ground right in
// front of the sensor: a few dozen to a few hundred per scan, each a chain of tens of microseconds), every other wavefront the leaves of <= 64 points --
// as two launches the short leaves waited 50 us behind the long ones.
//   <= 64 points: no LDS; the segment's indices in the lanes, ranked by counting, the sums by v_readlane from the lane holding rank r;
//   65 .. 2048:   the segment ordered through a bitmap over the scan's point indices in LDS (set the members' bits, read the words back in order: no
//                 sorting network -- a lone wavefront needs ~130 us to sort 2048 keys that way) when the scan has at most DSH_BITMAP_PTS points, else
//                 by the network.
#define DSH_BITMAP_PTS 131072
#define DSH_BIG_BLOCKS 768
__global__ __launch_bounds__(256) void ds_leaf_emit_kernel(const DsDyn* __restrict__ dynp, DsEnt* __restrict__ tab, const float4* __restrict__ pool4,
                                                            const unsigned long long* __restrict__ keys_sorted, const int32_t* __restrict__ slots_sorted,
                                                            const int32_t* __restrict__ big_list, const int32_t* __restrict__ leaf_rank, int32_t* __restrict__ info,
                                                            int32_t* __restrict__ n_out) {
    DS_DYN(dynp);
    // one region, two lives: the bitmap over the scan's point indices while a long leaf is being ordered, then the leaf's coordinates by component
    __shared__ __attribute__((aligned(16))) unsigned int bm[3 * DSH_LEAF_CAP];
    static_assert(3 * DSH_LEAF_CAP >= DSH_BITMAP_PTS / 32, "the coordinate table covers the bitmap");
    float* const c3 = (float*)bm;
    __shared__ int idx[DSH_LEAF_CAP];
    const float* __restrict__ pts = dyn.pts;
    float* __restrict__ out = dyn.out;
    const int n = dyn.n, stride = dyn.stride;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nleaf = info[0], nbig = info[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = nleaf;
    const int n_big_blocks = min((int)gridDim.x, DSH_BIG_BLOCKS);
    auto lds_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };
    if ((int)blockIdx.x < n_big_blocks && wv == 0) {
        // ---- the long leaves
        const int nwords = (min(n, DSH_BITMAP_PTS) + 31) / 32;
        for (int b = blockIdx.x; b < nbig; b += n_big_blocks) {
            const int e = big_list[b];
            const int slot = slots_sorted[e] & 0x7FFFFFFF;
            const DsEnt ent = tab[slot];
            const int cnt = ent.cnt >> 16;
            const int rank = leaf_rank[e];
            if (lane == 0) { tab[slot].key = DSH_EMPTY; tab[slot].cnt = 0; }   // the entry goes back empty: the table is never cleared as a whole
            if (cnt > DSH_LEAF_CAP || cnt <= 64) { if (lane == 0) info[1] = 1; continue; }
            if (n <= DSH_BITMAP_PTS) {
                for (int w = lane; w < nwords; w += 64) bm[w] = 0u;
                lds_sync();
                for (int k = lane; k < cnt; k += 64) { const int p = __float_as_int(pool4[ent.off + k].w); atomicOr(&bm[p >> 5], 1u << (p & 31)); }
                lds_sync();
                int base = 0;
                for (int w0 = 0; w0 < nwords; w0 += 64) {
                    const int w = w0 + lane;
                    unsigned int bits = w < nwords ? bm[w] : 0u;
                    if (!__any(bits != 0u)) continue;
                    const int pc = __popc(bits);
                    int incl = pc;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
                    int o = base + incl - pc;
                    while (bits) { const int bit = __ffs(bits) - 1; bits &= bits - 1; idx[o++] = (w << 5) | bit; }
                    base += __shfl(incl, 63, 64);
                }
                lds_sync();
            } else {
                int np2 = 128; while (np2 < cnt) np2 <<= 1;
                for (int k = lane; k < np2; k += 64) idx[k] = k < cnt ? __float_as_int(pool4[ent.off + k].w) : 0x7FFFFFFF;
                lds_sync();
                for (int k = 2; k <= np2; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int p = lane; p < np2; p += 64) {
                            const int q = p ^ j;
                            if (q > p) { const bool up = ((p & k) == 0); const int a_ = idx[p], b_ = idx[q]; if ((a_ > b_) == up) { idx[p] = b_; idx[q] = a_; } }
                        }
                        lds_sync();
                    }
            }
            // the coordinates in index order, one table per component (all gathers in flight at once), then the spec's sequential `centroid += pt` with
            // lane c adding component c: one 16-byte LDS read per four terms and nothing but dependent v_add_f32 between them (as v_readlane from the
            // lanes holding a batch it was ~40 cycles a term: the long pole of the launch for a leaf of 2 000 points)
            for (int k = lane; k < cnt; k += 64) {
                const float* q = pts + (size_t)idx[k] * stride;
                c3[k] = q[0]; c3[DSH_LEAF_CAP + k] = q[1]; c3[2 * DSH_LEAF_CAP + k] = q[2];
            }
            lds_sync();
            {
                const float* const col = c3 + min(lane, 2) * DSH_LEAF_CAP;
                float sum = 0.f;
                int k = 0;
#pragma unroll 4
                for (; k + 4 <= cnt; k += 4) { const float4 v = *(const float4*)(col + k); sum += v.x; sum += v.y; sum += v.z; sum += v.w; }
                for (; k < cnt; k++) sum += col[k];
                if (lane < 3) out[(size_t)rank * 3 + lane] = sum / (float)cnt;
            }
            lds_sync();
        }
        return;
    }
    // ---- the short leaves: every other wavefront, compactly numbered
    const int wid = blockIdx.x * 4 + wv - min((int)blockIdx.x, n_big_blocks) - (((int)blockIdx.x < n_big_blocks) ? 1 : 0);
    const int n_small_waves = gridDim.x * 4 - n_big_blocks;
    for (int e = wid; e < nleaf; e += n_small_waves) {
        const int slot = slots_sorted[e];
        if (slot < 0) continue;   // a long leaf: its own wavefront's (its table entry may be handed back at any moment -- never looked at here)
        const DsEnt ent = tab[slot];
        const int cnt = ent.cnt >> 16;   // (complete: the scatter launch is behind us)
        // the segment holds the points themselves (one round trip behind the entry); the output position comes ready from ds_scatter_kernel
        int mine = 0x7FFFFFFF;
        float x = 0.f, y = 0.f, z = 0.f;
        if (lane < cnt) { const float4 v = pool4[ent.off + lane]; x = v.x; y = v.y; z = v.z; mine = __float_as_int(v.w); }
        const int rank = leaf_rank[e];
        if (lane == 0) { tab[slot].key = DSH_EMPTY; tab[slot].cnt = 0; }
        if (cnt <= 0) { if (lane == 0) info[1] = 1; continue; }
        // rank by counting (indices are distinct), then the sequential float32 sum in index order: v_readlane from the lane holding rank r
        int r = 0;
        for (int l = 0; l < cnt; l++) r += __builtin_amdgcn_readlane(mine, l) < mine ? 1 : 0;
        int holder = 0;   // holder (in lane q) = the lane whose point has rank q
        for (int l = 0; l < cnt; l++) { const int rl = __builtin_amdgcn_readlane(r, l); if (rl == lane) holder = l; }
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int q = 0; q < cnt; q++) {
            const int hl = __builtin_amdgcn_readlane(holder, q);
            sx += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), hl));   // (0.f + first, then += : the spec's order)
            sy += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), hl));
            sz += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), hl));
        }
        if (lane == 0) { const float c = (float)cnt; out[(size_t)rank * 3 + 0] = sx / c; out[(size_t)rank * 3 + 1] = sy / c; out[(size_t)rank * 3 + 2] = sz / c; }
    }
}
// tab: table of `cap` (power of two, >= 2 n_max) entries, all empty on entry and on exit.  info: 4 ints, zero on entry (ds_publish_kernel, the last launch of every sequence, hands them back zeroed).  pool4: n_max float4.  dyn: the cloud's
// parameters in pinned, device-mapped memory.  Fixed grids (the kernels stride): the sequence can be captured once and replayed.
void launch_ds_hash_pipeline(hipStream_t s, const DsDyn* dyn, void* tab, unsigned long long cap, int32_t* pt_slot, int32_t* leaf_slot,
                             unsigned long long* keys_sorted, int32_t* slots_sorted, float* pool4, int32_t* big_list, int32_t* info, int32_t* n_out) {
    KLAUNCH(ds_hash_kernel, dim3(512), dim3(256), 0, s, dyn, (DsEnt*)tab, cap - 1, pt_slot, leaf_slot, info);
    // (big_list: sorted positions of the leaves above 64 points -- at most n / 65 of them)
    KLAUNCH(ds_leaf_sort_kernel, dim3(256), dim3(256), 0, s, (DsEnt*)tab, leaf_slot, info, keys_sorted, slots_sorted, big_list);
    // (leaf_slot has done its job once the leaves are sorted: the scatter launch hands it on holding every sorted leaf's output position)
    KLAUNCH(ds_scatter_kernel, dim3(512), dim3(256), 0, s, dyn, (DsEnt*)tab, pt_slot, (float4*)pool4, keys_sorted, info, leaf_slot);
    KLAUNCH(ds_leaf_emit_kernel, dim3(2304), dim3(256), 0, s, dyn, (DsEnt*)tab, (const float4*)pool4, keys_sorted, slots_sorted, big_list, leaf_slot, info, n_out);
}
// last launch of the asynchronous form: the leaf count and the fall-back flag go to pinned host memory (a plain kernel, not a copy node: the whole
// sequence is one hipGraph of kernels only)
__global__ void ds_publish_kernel(int32_t* __restrict__ info, int32_t* __restrict__ host_info, const DsDyn* __restrict__ dynp) {
    if (threadIdx.x == 0) {
        const int ticket = dynp->pad;   // (the job number the host waits for: the host polls host_info[2] -- no event packet behind the sequence)
        __hip_atomic_store(&host_info[0], info[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_info[1], info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_info[2], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x < 4) info[threadIdx.x] = 0;   // the counters go back zeroed: the launch sequence holds kernels only (no memset node, no copy node)
}
void launch_ds_publish(hipStream_t s, int32_t* info, int32_t* host_info, const DsDyn* dyn) { KLAUNCH(ds_publish_kernel, dim3(1), dim3(64), 0, s, info, host_info, dyn); }
// a fall-back left part of the table occupied: back to all-empty
__global__ void ds_table_reset_kernel(DsEnt* tab, unsigned long long cap) {
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < cap; k += (size_t)gridDim.x * blockDim.x) { tab[k].key = DSH_EMPTY; tab[k].cnt = 0; tab[k].off = 0; }
}
void launch_ds_table_reset(hipStream_t s, void* tab, unsigned long long cap) { KLAUNCH(ds_table_reset_kernel, dim3(1024), dim3(256), 0, s, (DsEnt*)tab, cap); }

// xyz (3 floats) -> xyzI (4 floats, intensity 0): the mesher consumes pcl::PointXYZI-shaped clouds
__global__ void ds_expand_xyzi_kernel(const float* __restrict__ xyz, int n, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float4(xyz[(size_t)i * 3], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2], 0.f);
}
void launch_ds_expand_xyzi(hipStream_t s, const float* xyz, int n, float* out_xyzi) {
    KLAUNCH(ds_expand_xyzi_kernel, dim3((n + 255) / 256), dim3(256), 0, s, xyz, n, (float4*)out_xyzi);
}

// profiler prelude (prof.hpp): keep the stream busy for ~25 us (wall_clock64 ticks at 100 MHz)
__global__ void kprof_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void kprof_spin(hipStream_t s) { hipLaunchKernelGGL(kprof_spin_kernel, dim3(1), dim3(1), 0, s, 2500LL); }

// =====================================================================================================================
// ImuProcess::UndistortPcl, per-point part (src/IMU_Processing.cpp:914-957): sort key + compensation into the scan-end frame
// =====================================================================================================================
__global__ void undistort_keys_kernel(const float* __restrict__ pts5, int n, uint32_t* __restrict__ key, int32_t* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = __float_as_uint(pts5[(size_t)i * 5 + 4]);
    key[i] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // order-preserving map of float to uint (ascending offset time)
    idx[i] = i;
}
IMD void und_exp_rate(const double* w, double dt, double* R) {   // Exp(ang_vel, dt), include/so3_math.h:30-50
    const double nrm = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    if (nrm > 0.0000001) {
        const double r[3] = {w[0] / nrm, w[1] / nrm, w[2] / nrm};
        const double K[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
        double KK[9];
        m3_mul(K, K, KK);
        const double ang = nrm * dt, s = sin(ang), c1 = 1.0 - cos(ang);
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = (R[i] + s * K[i]) + c1 * KK[i];
    }
}
// poses: n_poses x 23 doubles {offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]}; fe: {R_end[9], p_end[3], Lr[9], Lo[3]}
__global__ __launch_bounds__(256) void undistort_kernel(const float* __restrict__ pts5, const int32_t* __restrict__ order, int n,
                                                        const double* __restrict__ poses, int n_poses, const double* __restrict__ fe,
                                                        float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pts5 + (size_t)order[i] * 5;
    float x = p[0], y = p[1], z = p[2];
    const double t = (double)p[4] / double(1000);
    // interval: the last pose whose offset time lies strictly before the point (the reference walks the intervals backwards and takes every
    // point with curvature/1000 > head->offset_time that a later interval has not taken)
    int h = -1;
    for (int k = n_poses - 2; k >= 0; k--) if (t > poses[(size_t)k * 23]) { h = k; break; }
    const int h_last = (i == 0) ? 0 : h;   // the earliest point is compensated again by every earlier interval (the loop re-enters with it_pcl == begin)
    for (int k = h; k >= h_last && k >= 0; k--) {
        const double* q = poses + (size_t)k * 23;
        if (!(t > q[0])) break;
        const double dt = t - q[0];
        double E[9], R_i[9];
        und_exp_rate(q + 4, dt, E);
        m3_mul(q + 13, E, R_i);
        const double T_ei[3] = {((q[10] + q[7] * dt) + 0.5 * q[1] * dt * dt) - fe[9], ((q[11] + q[8] * dt) + 0.5 * q[2] * dt * dt) - fe[10],
                                ((q[12] + q[9] * dt) + 0.5 * q[3] * dt * dt) - fe[11]};
        const double P_i[3] = {(double)x, (double)y, (double)z};
        double a1[3], a2[3], a3[3], a4[3];
        m3_vec(fe + 12, P_i, a1);
        a1[0] += fe[21]; a1[1] += fe[22]; a1[2] += fe[23];
        m3_vec(R_i, a1, a2);
        a2[0] += T_ei[0]; a2[1] += T_ei[1]; a2[2] += T_ei[2];
        m3t_vec(fe, a2, a3);
        a3[0] -= fe[21]; a3[1] -= fe[22]; a3[2] -= fe[23];
        m3t_vec(fe + 12, a3, a4);
        x = (float)a4[0]; y = (float)a4[1]; z = (float)a4[2];
    }
    out[i] = make_float4(x, y, z, p[3]);
}
void launch_undistort_keys(hipStream_t s, const float* pts5, int n, uint32_t* key, int32_t* idx) {
    KLAUNCH(undistort_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts5, n, key, idx);
}
void launch_undistort(hipStream_t s, const float* pts5, const int32_t* order, int n, const double* poses, int n_poses, const double* fe, float* out_xyzi) {
    KLAUNCH(undistort_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts5, order, n, poses, n_poses, fe, (float4*)out_xyzi);
}

// =====================================================================================================================
// sensor decode: Preprocess::avia_handler / velodyne_handler as flag -> scan -> compact
// =====================================================================================================================
IMD float rd_f32(const uint8_t* p) { uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); return __uint_as_float(v); }
IMD uint32_t rd_u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
// pass 0: counted[i] = 1 when the point takes part in the valid_num count (i >= 1 && line < N_SCANS)
__global__ void decode_livox_count_kernel(const uint8_t* __restrict__ w, int n, int n_scans, int32_t* __restrict__ counted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    counted[i] = (i >= 1 && (int)w[(size_t)i * 19 + 18] < n_scans) ? 1 : 0;
}
// pass 1 (after an exclusive scan of counted): keep[i]
__global__ void decode_livox_keep_kernel(const uint8_t* __restrict__ w, int n, int n_scans, int filter, double blind_sqr, const int32_t* __restrict__ valid_excl,
                                         int32_t* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int k = 0;
    const uint8_t* p = w + (size_t)i * 19;
    if (i >= 1 && (int)p[18] < n_scans) {
        const int valid_num = valid_excl[i] + 1;   // valid_num++ happens before the test
        if (valid_num % filter == 0) {
            const float x = rd_f32(p + 4), y = rd_f32(p + 8), z = rd_f32(p + 12), inten = (float)p[16];
            if (inten > 4 && (double)(x * x + y * y + z * z) > blind_sqr) k = 1;
        }
    }
    keep[i] = k;
}
__global__ void decode_livox_emit_kernel(const uint8_t* __restrict__ w, int n, const int32_t* __restrict__ keep_flag, const int32_t* __restrict__ pos, float* __restrict__ out,
                                         int32_t* __restrict__ n_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (keep_flag[i]) {
        const uint8_t* p = w + (size_t)i * 19;
        float* o = out + (size_t)pos[i] * 5;
        o[0] = rd_f32(p + 4); o[1] = rd_f32(p + 8); o[2] = rd_f32(p + 12); o[3] = (float)p[16];
        o[4] = (float)rd_u32(p) / float(1000000);   // offset_time / float(1000000): curvature = time of the point in ms
    }
    if (i == n - 1) *n_out = pos[i] + keep_flag[i];
}
__global__ void decode_velodyne_keep_kernel(const uint8_t* __restrict__ d, int n, int step, int ox, int oy, int oz, int n_scans, int32_t* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = d + (size_t)i * step;
    const float x = rd_f32(p + ox), y = rd_f32(p + oy), z = rd_f32(p + oz);
    const float angle = (float)((double)(atanf(z / sqrtf(x * x + y * y)) * 180) / 3.14159265358979323846);
    int scan_id;
    if ((double)angle >= -8.83) scan_id = (int)((2 - (double)angle) * 3.0 + 0.5);
    else scan_id = n_scans / 2 + (int)((-8.83 - (double)angle) * 2.0 + 0.5);
    keep[i] = ((double)angle > 2 || (double)angle < -24.33 || scan_id > 50 || scan_id < 0) ? 0 : 1;
}
__global__ void decode_velodyne_emit_kernel(const uint8_t* __restrict__ d, int n, int step, int ox, int oy, int oz, int oi, const int32_t* __restrict__ keep_flag,
                                            const int32_t* __restrict__ pos, float* __restrict__ out, int32_t* __restrict__ n_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (keep_flag[i]) {
        const uint8_t* p = d + (size_t)i * step;
        float* o = out + (size_t)pos[i] * 5;
        o[0] = rd_f32(p + ox); o[1] = rd_f32(p + oy); o[2] = rd_f32(p + oz); o[3] = rd_f32(p + oi); o[4] = 0.f;
    }
    if (i == n - 1) *n_out = pos[i] + keep_flag[i];
}
void launch_decode_livox_count(hipStream_t s, const uint8_t* w, int n, int n_scans, int32_t* counted) {
    KLAUNCH(decode_livox_count_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, n, n_scans, counted);
}
void launch_decode_livox_keep(hipStream_t s, const uint8_t* w, int n, int n_scans, int filter, double blind_sqr, const int32_t* valid_excl, int32_t* keep) {
    KLAUNCH(decode_livox_keep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, n, n_scans, filter, blind_sqr, valid_excl, keep);
}
void launch_decode_livox_emit(hipStream_t s, const uint8_t* w, int n, const int32_t* keep, const int32_t* pos, float* out, int32_t* n_out) {
    KLAUNCH(decode_livox_emit_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, n, keep, pos, out, n_out);
}
void launch_decode_velodyne_keep(hipStream_t s, const uint8_t* d, int n, int step, int ox, int oy, int oz, int n_scans, int32_t* keep) {
    KLAUNCH(decode_velodyne_keep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d, n, step, ox, oy, oz, n_scans, keep);
}
void launch_decode_velodyne_emit(hipStream_t s, const uint8_t* d, int n, int step, int ox, int oy, int oz, int oi, const int32_t* keep, const int32_t* pos, float* out,
                                 int32_t* n_out) {
    KLAUNCH(decode_velodyne_emit_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d, n, step, ox, oy, oz, oi, keep, pos, out, n_out);
}

// ---- launchers ------------------------------------------------------------------------------------------------------
void launch_ds_minmax(hipStream_t s, const float* pts, int n, int stride, float inv, int* mm) {
    KLAUNCH(ds_minmax_kernel, dim3(min(1024, (n + 255) / 256)), dim3(256), 0, s, pts, n, stride, inv, mm);
}
void launch_ds_index(hipStream_t s, const float* pts, int n, int stride, float inv, const int* mm, unsigned long long* keys, int32_t* vals) {
    KLAUNCH(ds_index_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, stride, inv, mm, keys, vals);
}
void launch_ds_heads(hipStream_t s, const unsigned long long* keys_sorted, int n, int32_t* flags) {
    KLAUNCH(ds_heads_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keys_sorted, n, flags);
}
void launch_ds_centroid(hipStream_t s, const float* pts, int n, int stride, const unsigned long long* keys_sorted, const int32_t* idx_sorted, const int32_t* rank,
                        float* out, int32_t* n_out) {
    KLAUNCH(ds_centroid_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, stride, keys_sorted, idx_sorted, rank, out, n_out);
}
