// HIP kernels of the mesher half of the hot path, gfx950.
//   mesh_transform_kernel        transformLidar of the full scan (voxel_mapping_common.cpp:709-726)
//   mesh_append_*_kernel         Global_map::append_points_to_global_map (pointcloud_rgbd.cpp:411-552): the sequential
//                                "accept a point iff no earlier accepted vertex within min_spacing" rule as a parallel fixed
//                                point over the 27-cell neighbourhood of the dedupe grid (lowest scan index wins)
//   mesh_knn_kernel              retrieve_neighbor_pts_kdtree (mesh_rec_geometry.cpp:336-377): one workgroup per active mesh
//                                voxel, the surrounding voxel block staged through LDS, exact 20-NN per vertex (float
//                                distances as KD_TREE::calc_dist, ikd_Tree.cpp:1722), smoothing, neighbourhood union
//   mesh_delaunay64_kernel       delaunay_triangulation + triangle_compare + correct_triangle_index
//                                (mesh_rec_geometry.cpp:174-295, 137-172, 399-433): one wavefront per voxel; PCA, 2-D
//                                projection, wave-parallel Bowyer-Watson with the plain-double Simple_cartesian predicates,
//                                skinny-face filter, diff against the live triangles found through the min-vertex lists
//   mesh_finalize/commit/emit    cross-voxel resolution ("all removes, then all adds", later voxel wins a flip) and the
//                                Triangle_manager update (triangle.hpp:164-246, 330-395)
// Bound: latency / LDS (integer, pointer and predicate work); nothing here is GEMM-shaped, MFMA is not used.
#include "../../include/immesh_c_api.h"
#include "mesh_kernels.hpp"
#include "dev_math.hpp"
#include "prof.hpp"

using namespace imd;

#define MKEY_BIAS (1 << 20)
#define MKEY_MASK ((1ull << 21) - 1)
#define MKEY_EMPTY 0xFFFFFFFFFFFFFFFFull
#define ST_UNDECIDED 0
#define ST_ACCEPT 1
#define ST_REJECT 2
#define ST_WAIT 3            /* sharded admission: undecided, and what it waits for is another rank's decision (stalled until the next exchange) */
// per-candidate flags of the sharded admission (MeshDev::cand_flags; zero for every candidate the rank knows nothing about)
#define CF_OWN 1             /* this rank decides the candidate (its mesh voxel lies in one of the rank's bricks) */
#define CF_BAND 2            /* ... and a candidate of another rank may lie within min_spacing of it */
#define CF_SURV_SENT 4       /* announced to the other ranks as a survivor of the test against the map */
#define CF_DEC_SENT 8        /* its decision has been sent */
#define CF_KNOWN 16          /* another rank's band survivor, chained under its cell here */
#define TRI_ADD_BIT 0x80000000u

// x-major packing: ascending packed key == ascending (x,y,z), the order in which the CPU checker visits active voxels
IMD unsigned long long mkey(long x, long y, long z) {
    return (((unsigned long long)(x + MKEY_BIAS) & MKEY_MASK) << 42) | (((unsigned long long)(y + MKEY_BIAS) & MKEY_MASK) << 21) |
           ((unsigned long long)(z + MKEY_BIAS) & MKEY_MASK);
}
IMD long rnd_cell(float p, double cell) { return (long)(int)round((double)p / cell); }  // std::round of the f64 quotient, pointcloud_rgbd.cpp:467-472
// sharded mesher: mesh voxels are owned in bricks of 2^shard_brick_log2 voxels per axis; the owner function is the registration map's (regmap.hpp brick_owner)
IMD int mesh_owner_xyz(const MeshDev& m, long x, long y, long z) {
    const int b = m.shard_brick_log2;   // arithmetic shifts: bricks tile negative cells too
    if (m.shard_scheme == 1) return (int)(hash64(mkey(x >> b, y >> b, z >> b)) % (unsigned long long)m.shard_world);
    const long c = ((x >> b) + 3 * (y >> b) + 5 * (z >> b)) % (long)m.shard_world;
    return (int)(c < 0 ? c + m.shard_world : c);
}
IMD int mesh_owner(const MeshDev& m, unsigned long long vkey) {
    const long x = (long)((vkey >> 42) & MKEY_MASK) - MKEY_BIAS, y = (long)((vkey >> 21) & MKEY_MASK) - MKEY_BIAS, z = (long)(vkey & MKEY_MASK) - MKEY_BIAS;
    return mesh_owner_xyz(m, x, y, z);
}
// The boundary band.  A box of mesh voxels [lo, hi] no wider than a brick touches at most 2 x 2 x 2 bricks -- those of its corners.
//   box_foreign: does a brick of ANOTHER rank than `rank` touch the box?      box_has: does a brick of `rank` touch it?
IMD bool box_foreign(const MeshDev& m, const long* lo, const long* hi, int rank) {
    bool f = false;
#pragma unroll
    for (int q = 0; q < 8; q++) f = f || mesh_owner_xyz(m, (q & 4) ? hi[0] : lo[0], (q & 2) ? hi[1] : lo[1], (q & 1) ? hi[2] : lo[2]) != rank;
    return f;
}
IMD bool box_has(const MeshDev& m, const long* lo, const long* hi, int rank) {
    bool f = false;
#pragma unroll
    for (int q = 0; q < 8; q++) f = f || mesh_owner_xyz(m, (q & 4) ? hi[0] : lo[0], (q & 2) ? hi[1] : lo[1], (q & 1) ? hi[2] : lo[2]) == rank;
    return f;
}
// a mesh voxel's reach: the vertices of its neighbourhood union lie within accept = 1.25 voxel of its own vertices, i.e. at most 2 voxel indices away
// (mesh_rec_geometry.cpp:343) -- so what a voxel computes can matter to the voxels within +-2 of it, and only to them
#define MV_REACH 2
IMD void voxel_box(unsigned long long vkey, int h, long* lo, long* hi) {
    const long x = (long)((vkey >> 42) & MKEY_MASK) - MKEY_BIAS, y = (long)((vkey >> 21) & MKEY_MASK) - MKEY_BIAS, z = (long)(vkey & MKEY_MASK) - MKEY_BIAS;
    lo[0] = x - h; lo[1] = y - h; lo[2] = z - h; hi[0] = x + h; hi[1] = y + h; hi[2] = z + h;
}
// the mesh voxels a point within min_spacing of p can fall into (2 * min_spacing < voxel: two per axis at most)
IMD void cand_box(const MeshDev& m, float px, float py, float pz, long* lo, long* hi) {
    const float p[3] = {px, py, pz};
#pragma unroll
    for (int a = 0; a < 3; a++) { lo[a] = (long)round(((double)p[a] - m.min_spacing) / m.voxel); hi[a] = (long)round(((double)p[a] + m.min_spacing) / m.voxel); }
}
IMD int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
IMD void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

IMD long long h_find(const unsigned long long* keys, unsigned long long mask, unsigned long long key) {
    unsigned long long h = hash64(key) & mask;
    for (int probe = 0; probe < 8192; probe++) {
        const unsigned long long k = keys[h];
        if (k == key) return (long long)h;
        if (k == MKEY_EMPTY) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}
IMD long long h_find_or_insert(unsigned long long* keys, unsigned long long mask, unsigned long long key, bool* created) {
    unsigned long long h = hash64(key) & mask;
    *created = false;
    for (int probe = 0; probe < 8192; probe++) {
        const unsigned long long k = keys[h];
        if (k == key) return (long long)h;
        if (k == MKEY_EMPTY) {
            const unsigned long long prev = atomicCAS(&keys[h], (unsigned long long)MKEY_EMPTY, key);
            if (prev == MKEY_EMPTY) { *created = true; return (long long)h; }
            if (prev == key) return (long long)h;
        }
        h = (h + 1) & mask;
    }
    return -1;
}

// ---- dedupe grid (m_hashmap_3d_pts): bricked open addressing, see MeshDev::g_ent
IMD unsigned long long grid_home(const MeshDev& m, long gx, long gy, long gz) {
    const unsigned long long region = hash64(mkey(gx >> 2, gy >> 2, gz >> 2)) & (m.g_mask >> 6);   // arithmetic shifts: bricks tile negative cells too
    return (region << 6) | (unsigned long long)(((gx & 3) << 4) | ((gy & 3) << 2) | (gz & 3));
}
// continue the linear probe behind a home slot that holds another cell's entry (rare: a real call, not 27 inlined copies of the loop -- the admission
// kernel is latency-bound code that has to stay in the instruction cache)
__device__ __noinline__ long long grid_find_from(const MeshDev& m, unsigned long long slot, unsigned long long key) {
    for (int probe = 0; probe < 8192; probe++) {
        slot = (slot + 1) & m.g_mask;
        const unsigned long long k = m.g_ent[slot].key;
        if (k == key) return (long long)slot;
        if (k == MKEY_EMPTY) return -1;
    }
    return -1;
}
IMD long long grid_insert(const MeshDev& m, long gx, long gy, long gz, unsigned long long key) {
    unsigned long long h = grid_home(m, gx, gy, gz);
    for (int probe = 0; probe < 8192; probe++) {
        const unsigned long long k = m.g_ent[h].key;
        if (k == key) return (long long)h;
        if (k == MKEY_EMPTY) {
            const unsigned long long prev = atomicCAS(&m.g_ent[h].key, (unsigned long long)MKEY_EMPTY, key);
            if (prev == MKEY_EMPTY || prev == key) return (long long)h;
        }
        h = (h + 1) & m.g_mask;
    }
    return -1;
}
IMD void mkey_unpack(unsigned long long k, long& x, long& y, long& z) {
    x = (long)((k >> 42) & MKEY_MASK) - MKEY_BIAS; y = (long)((k >> 21) & MKEY_MASK) - MKEY_BIAS; z = (long)(k & MKEY_MASK) - MKEY_BIAS;
}

// ---- mesh-voxel hash (m_hashmap_voxels): 16-byte entries {key, voxel index}: one round trip per lookup
IMD int vox_find(const MeshDev& m, unsigned long long key) {   // voxel index, or -1 (no such voxel / being created in this launch)
    unsigned long long h = hash64(key) & m.x_mask;
    for (int probe = 0; probe < 8192; probe++) {
        const MeshVoxEnt e = m.x_ent[h];
        if (e.key == key) return e.val;
        if (e.key == MKEY_EMPTY) return -1;
        h = (h + 1) & m.x_mask;
    }
    return -1;
}
// slot of the key (inserted when absent: *created); *val = the index stored there (-1: not published yet); -1: table full
IMD long long vox_find_or_insert(const MeshDev& m, unsigned long long key, bool* created, int* val) {
    unsigned long long h = hash64(key) & m.x_mask;
    *created = false; *val = -1;
    for (int probe = 0; probe < 8192; probe++) {
        const MeshVoxEnt e = m.x_ent[h];
        if (e.key == key) { *val = e.val; return (long long)h; }
        if (e.key == MKEY_EMPTY) {
            const unsigned long long prev = atomicCAS(&m.x_ent[h].key, (unsigned long long)MKEY_EMPTY, key);
            if (prev == MKEY_EMPTY) { *created = true; return (long long)h; }
            if (prev == key) { *val = ld_agent(&m.x_ent[h].val); return (long long)h; }
        }
        h = (h + 1) & m.x_mask;
    }
    return -1;
}

// float squared distance exactly as KD_TREE::calc_dist (include/ikd-Tree/ikd_Tree.cpp:1722-1728)
IMD float dist2f(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz);
}

// Per-scan parameters live in device memory (MeshDev::dyn, refreshed by a copy at the head of every scan) so that the kernel arguments
// are identical from scan to scan and the whole launch sequence can be replayed as a hipGraph.
#ifndef MESH_B_PRIO
#define MESH_B_PRIO 2   /* wave priority of the mesher's phase B + triangulation kernels (s_setprio; the registration runs at 3, the map update at REG_UPD_PRIO) */
#endif
// phase marks (IMMESH_DEBUG_WAITS): thread 0 of block 0 leaves the device's real-time counter at kernel entry; mesh_publish_kernel hands the marks to the host
#define MESH_MARK(arg, k) do { if (blockIdx.x == 0 && threadIdx.x == 0) (arg).tick0[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MESH_DYN(arg)                                  \
    MeshDev m = (arg);                                 \
    const MeshScanParams sp = (arg).dyn->sp;           \
    m.seq = (arg).dyn->seq;                            \
    m.ch_mask = (arg).dyn->ch_mask;                    \
    const float* const dyn_pts = (arg).dyn->pts;       \
    (void)dyn_pts;                                     \
    (void)sp

IMD void list_push(const MeshDev& m, int32_t* list, int counter, int v) {
    const int pos = atomicAdd(&m.sc[counter], 1);
    if (pos < m.cap_list) list[pos] = v; else m.sc[SC_OVERFLOW] = 14;
}

// =====================================================================================================================
// transformLidar of the full scan
// =====================================================================================================================
struct XformParams { double R[9], t[3], extR[9], extT[3]; };
// rt_dev != nullptr: R (9 doubles) and t (3) of the pose are read from device memory -- the head of RegState::sp, left there by the in-kernel EKF
__global__ __launch_bounds__(256) void mesh_transform_kernel(const float4* __restrict__ in, float4* __restrict__ out, int n, XformParams xp, const double* __restrict__ rt_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (rt_dev) {
#pragma unroll
        for (int k = 0; k < 9; k++) xp.R[k] = rt_dev[k];
#pragma unroll
        for (int k = 0; k < 3; k++) xp.t[k] = rt_dev[9 + k];
    }
    const float4 v = in[i];
    const double p[3] = {(double)v.x, (double)v.y, (double)v.z};
    double pi[3], pw[3];
    m3_vec(xp.extR, p, pi);
    pi[0] += xp.extT[0]; pi[1] += xp.extT[1]; pi[2] += xp.extT[2];
    m3_vec(xp.R, pi, pw);
    out[i] = make_float4((float)(pw[0] + xp.t[0]), (float)(pw[1] + xp.t[1]), (float)(pw[2] + xp.t[2]), v.w);
}

// =====================================================================================================================
// append_points_to_global_map
// =====================================================================================================================
// first kernel of a scan: takes the scan's parameters from pinned host memory (no separate copy node), clears the per-scan counters and the
// candidate-cell table (no separate fill nodes: each of those ran as a ~5 us kernel of its own), and snapshots the vertex count -- the id of the
// scan's first new vertex is the device's own count, so the host can enqueue a scan before the previous one has reported its size
#define MV_FIN_CAND 16384
// Admission order.  A scan's directions are not spatially ordered: in scan order every one of the 27 probes of every candidate was a line miss of its own
// (32 MB of line traffic for 0.4 MB of entries).  mesh_begin_scan_kernel therefore drops every candidate into the bucket of the 8-cell (0.8 m) cube it
// falls into -- MV_BIN_BUCKETS buckets of MV_BIN_SLOTS slots, one returning atomic per candidate, no prefix sum, no second pass; what does not fit its
// bucket goes to an overflow list -- and mesh_append_prepare_kernel walks the slots: the lanes of a wavefront are the candidates of four cubes and
// probe the SAME lines of the dedupe grid / the mesh-voxel hash.  The order only decides who probes what when; "lowest scan index wins" is settled by the
// candidate indices themselves.  Offline-sized clouds (> MV_FIN_CAND candidates) keep scan order.
#define MV_BIN_BUCKETS 1024
#define MV_BIN_SLOTS 16
#define MV_BIN_NSLOT (MV_BIN_BUCKETS * MV_BIN_SLOTS)
__global__ __launch_bounds__(256) void mesh_begin_scan_kernel(MeshDev m, const MeshDyn* __restrict__ h_dyn, unsigned long long ccap) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i; k < ccap; k += stride) { m.ch_keys[k] = MKEY_EMPTY; m.ch_head[k] = -1; }
    // every candidate starts out rejected: the admission writes a candidate's records only when it survives the test against the map
    // (n_cand <= ccap / 4 by construction of ccap)
    for (size_t k = i; k < (size_t)min((unsigned long long)m.cap_cand, ccap >> 2); k += stride) { m.cand_status[k] = ST_REJECT; m.cand_flags[k] = 0; }
    __shared__ MeshDyn s_dyn;
    __shared__ int s_late;
    if (threadIdx.x == 0) {   // every workgroup takes the scan's parameters from pinned host memory itself and waits for the scan itself
        MeshDyn d = *h_dyn;
        bool late = false;
        if (d.wait_flag) {   // the producer was enqueued before this launch; bounded all the same (~1 s) -- reported through the hang guard of the admission
            unsigned int spins = 0;
            while (__hip_atomic_load(d.wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < d.wait_seq) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 20)) { late = true; break; }
            }
        }
        s_dyn = d; s_late = late;
        if (late) m.bin_cnt[MV_BIN_BUCKETS + 1] = 1;   // (ANY workgroup that gave up: the admission kernel turns it into the scan's hang-guard error)
        if (blockIdx.x == 0) {
            *m.tick0 = __builtin_amdgcn_s_memrealtime();   // (behind the poll: the job's device time does not include waiting for the registration stream)
            const int base = m.pc[PC_VERTS];
            d.sp.vtx_base = base;
            *m.dyn = d;
            for (int k = 0; k < SC_COUNT; k++) m.sc[k] = 0;
            m.sc[SC_VTXBASE] = base;
            if (late) m.sc[SC_UNDECIDED] = 1;
        }
    }
    __syncthreads();
    const int n = s_dyn.sp.n_cand, step = s_dyn.sp.step;
    const float* __restrict__ pts = s_dyn.pts;
    int* __restrict__ cnt = m.bin_cnt;   // (zero: cleared by the previous scan's mesh_knn_kernel)
    if (n > MV_FIN_CAND || pts == nullptr || s_late) return;
    const float inv_coarse = (float)(1.0 / (8.0 * m.min_spacing));   // (the bucket only decides who sits beside whom in a wavefront: no exact rounding needed)
    for (size_t c = i; c < (size_t)n; c += stride) {
        const float4 p = *(const float4*)(pts + 4 * c * step);
        const int qx = __float2int_rd(p.x * inv_coarse), qy = __float2int_rd(p.y * inv_coarse), qz = __float2int_rd(p.z * inv_coarse);
        const int bk = (int)(((unsigned int)qx * 73856093u ^ (unsigned int)qy * 19349663u ^ (unsigned int)qz * 83492791u) & (MV_BIN_BUCKETS - 1));
        const int r = atomicAdd(&cnt[bk], 1);
        const int pos = r < MV_BIN_SLOTS ? bk * MV_BIN_SLOTS + r : MV_BIN_NSLOT + atomicAdd(&cnt[MV_BIN_BUCKETS], 1);
        m.cand_pt[pos] = make_float4(p.x, p.y, p.z, __int_as_float((int)c));
    }
}
// last launch of a job: the per-scan counters, the job's device time and -- last -- its sequence number go to pinned host memory (the worker thread
// polls the sequence number: no device-to-host copy packet, no event record behind the last kernel of phase B)
__global__ void mesh_publish_kernel(MeshDev m_in, int32_t* __restrict__ host_sc) {
    const int k = threadIdx.x;
    if (k < SC_COUNT) __hip_atomic_store(&host_sc[k], m_in.sc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (k == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        const unsigned long long ticks = now - *m_in.tick0;
        __hip_atomic_store((unsigned long long*)&host_sc[MESH_PUB_TICKS], ticks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (k < MESH_N_MARKS) {   // phase marks: [0] = the job's first kernel past its poll, [1..10] kernel entries, [11] = now (absolute ticks: the host takes differences)
        const unsigned long long v = (k == MESH_N_MARKS - 1 || k == 11) ? __builtin_amdgcn_s_memrealtime() : m_in.tick0[k];   // ([11]: this kernel's own entry)
        __hip_atomic_store((unsigned long long*)&host_sc[MESH_PUB_MARKS + 2 * k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (k == 0) __hip_atomic_store(&host_sc[MESH_PUB_SEQ], m_in.dyn->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_mesh_publish(hipStream_t s, const MeshDev& m, int32_t* host_sc) { static_assert(SC_COUNT <= 64, "one wavefront publishes the counters"); KLAUNCH(mesh_publish_kernel, dim3(1), dim3(64), 0, s, m, host_sc); }
void launch_mesh_begin_scan(hipStream_t s, const MeshDev& m, const MeshDyn* h_dyn_dev, unsigned long long ccap) {
    KLAUNCH(mesh_begin_scan_kernel, dim3(64), dim3(256), 0, s, m, h_dyn_dev, ccap);
}

// One candidate: mesh voxel found / created + marked visited, test against the map (own dedupe cell, then the 1-NN test over the 26 cells around it),
// survivors chained under their cell.  Every lane of the wavefront calls it (dead lanes with live = false: the visited marks are agreed wave-wide).
// Returns the number of vertices the 1-NN test inspected; *visit_new = a voxel this lane is the first to visit this scan (-1: none).
// The code is kept SMALL on purpose (the dx loop is rolled, the displaced-entry probe is a call): every wavefront runs it exactly once, so what it costs
// is instruction fetch -- a fully unrolled version (all 26 keys in one round trip, 50 KB of code) took 60 us where this one takes a third of it.
#define ADBG(k) do { if (m.dbg) { const unsigned long long _t = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&m.dbg[32 + (k)], _t - tprev); tprev = _t; } } while (0)
IMD int mesh_admit_candidate(const MeshDev& m, bool live, int i, float px, float py, float pz, int lane, int* visit_new) {
    *visit_new = -1;
    unsigned long long tprev = m.dbg ? __builtin_readcyclecounter() : 0;
    const long gx = rnd_cell(px, m.min_spacing), gy = rnd_cell(py, m.min_spacing), gz = rnd_cell(pz, m.min_spacing);
    const long bx = rnd_cell(px, m.voxel), by = rnd_cell(py, m.voxel), bz = rnd_cell(pz, m.voxel);
    const unsigned long long vkey = mkey(bx, by, bz);
    const unsigned long long gkey = mkey(gx, gy, gz);
    const unsigned long long rmask = m.g_mask >> 6;
    // round trip 1: the candidate's own dedupe cell (occupied -> rejected before the 1-NN test, pointcloud_rgbd.cpp:480-500) + the first probe of the
    // mesh-voxel hash
    const unsigned long long own_slot = ((hash64(mkey(gx >> 2, gy >> 2, gz >> 2)) & rmask) << 6) | (unsigned long long)(((gx & 3) << 4) | ((gy & 3) << 2) | (gz & 3));
    const unsigned long long own_k = live ? m.g_ent[own_slot].key : gkey;
    ADBG(0);   // index arithmetic
    // mesh voxel: find or create, mark visited (m_voxels_recent_visited, pointcloud_rgbd.cpp:480-500)
    int vi = -1;
    bool stamp = false;
    if (live) {
        bool created;
        int found;
        const long long vs = vox_find_or_insert(m, vkey, &created, &found);
        if (vs < 0) m.sc[SC_OVERFLOW] = 1;
        else if (created) {
            vi = atomicAdd(&m.pc[PC_VOXELS], 1);
            if (vi >= m.cap_voxels) { m.sc[SC_OVERFLOW] = 2; vi = -1; }
            else {
                m.vx_key[vi] = vkey; m.vx_npts[vi] = 0; m.vx_meshing_times[vi] = 0; m.vx_new_added[vi] = 0; m.vx_rank_seq[vi] = 0; m.vx_rank_seq_alt[vi] = 0; m.vx_rank_seq_alt2[vi] = 0; m.vx_stamp[vi] = m.seq;
                m.vx_short_axis[(size_t)vi * 3 + 0] = 0; m.vx_short_axis[(size_t)vi * 3 + 1] = 0; m.vx_short_axis[(size_t)vi * 3 + 2] = 0;
                *visit_new = vi;
                __threadfence();
                st_agent(&m.x_ent[vs].val, vi);
            }
        } else {
            vi = found;  // -1 while the creating lane of this launch has not published it: the creator marks it visited
            stamp = vi >= 0;
        }
    }
    ADBG(1);   // voxel hash: probe + index
    // visited mark: the lanes of a wavefront share a handful of voxels -- one exchange per distinct voxel (not 64 on the same address), all in flight together
    bool leader = false;
    {
        unsigned long long todo = __ballot(stamp);
        while (todo) {
            const int l = (int)__builtin_ctzll(todo);
            const int lead_vi = __builtin_amdgcn_readlane(vi, l);
            if (lane == l) leader = true;
            todo &= ~__ballot(stamp && vi == lead_vi);
        }
    }
    bool occupied = own_k == gkey;
    if (!occupied && own_k != MKEY_EMPTY) occupied = grid_find_from(m, own_slot, gkey) >= 0;
    // (issued behind the call above -- a call waits for everything outstanding -- and looked at behind the 1-NN test: it shares the first batch's round trip)
    int old_stamp = m.seq;
    if (leader) old_stamp = atomicExch(&m.vx_stamp[vi], m.seq);
    ADBG(2);   // leader election + own cell
    // sharded admission: every rank marks the voxels of ALL candidates visited (the voxel bookkeeping is replicated: 4 bytes per candidate), but only the
    // owner of a candidate's mesh-voxel brick tests it against the map and decides it; everybody else hears of it only if it matters
    // (mesh_cand_pack_kernel / mesh_cand_unpack_kernel)
    const bool probe = live && !occupied && (m.shard_world <= 1 || mesh_owner(m, vkey) == m.shard_rank);
    int status = probe ? ST_UNDECIDED : ST_REJECT;
    int probes = 0;
    // 1-NN test (pointcloud_rgbd.cpp:503-516): a vertex closer than min_spacing lies in one of the 26 other cells around the candidate's cell and every
    // cell holds at most one vertex, stored with its position in the grid entry (key and record in ONE 32-byte entry; the four cells of a z-row share
    // a line).  Nine cells per round trip; a candidate is out as soon as one is too close.
    const long rby = (gy - 1) >> 2, rbz = (gz - 1) >> 2;
    for (int dx = -1; dx <= 1; dx++) {
        if (!__any(status == ST_UNDECIDED)) break;
        if (status != ST_UNDECIDED) continue;
        const long cx = gx + dx;
        unsigned int rg[4];   // the 3 x 3 cells of this x-layer lie in at most 2 x 2 bricks
#pragma unroll
        for (int q = 0; q < 4; q++) rg[q] = (unsigned int)(hash64(mkey(cx >> 2, rby + (q >> 1), rbz + (q & 1))) & rmask);
        unsigned long long k9[9];
        unsigned int s9[9];
        float4 r9[9];
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const long cy = gy + (q / 3 - 1), cz = gz + (q % 3 - 1);
            const int sel = (int)((((cy >> 2) - rby) << 1) | ((cz >> 2) - rbz));
            const unsigned int r = sel == 3 ? rg[3] : (sel == 2 ? rg[2] : (sel == 1 ? rg[1] : rg[0]));
            s9[q] = (r << 6) | (unsigned int)(((cx & 3) << 4) | ((cy & 3) << 2) | (cz & 3));
            k9[q] = m.g_ent[s9[q]].key;
            r9[q] = *(const float4*)&m.g_ent[s9[q]];
        }
#pragma unroll
        for (int q = 0; q < 9; q++) {
            if (dx == 0 && q == 4) continue;            // the candidate's own cell (found free above)
            if (k9[q] == MKEY_EMPTY) continue;
            const unsigned long long want = mkey(cx, gy + (q / 3 - 1), gz + (q % 3 - 1));
            float4 r = r9[q];
            if (k9[q] != want) {   // displaced entry: continue the linear probe
                const long long s2 = grid_find_from(m, s9[q], want);
                if (s2 < 0) continue;
                r = *(const float4*)&m.g_ent[s2];
            }
            probes++;
            if ((double)sqrtf(dist2f(px, py, pz, r.x, r.y, r.z)) < m.min_spacing) status = ST_REJECT;
        }
    }
    ADBG(3);   // the three batches of the 1-NN test
    if (leader && old_stamp != m.seq) *visit_new = vi;
    if (m.dbg && lane == 0) atomicAdd(&m.dbg[32 + 7], 1ull);
    if (status != ST_UNDECIDED) return probes;              // (cand_status was preset to ST_REJECT by mesh_begin_scan_kernel)
    // survivor: its records, and the chain under its cell for the in-scan conflict resolution
    if (m.shard_world > 1) {
        long lo[3], hi[3];
        cand_box(m, px, py, pz, lo, hi);
        m.cand_flags[i] = CF_OWN | (box_foreign(m, lo, hi, m.shard_rank) ? CF_BAND : 0);
    }
    m.cand_vox[i] = vi;
    m.cand_cell[i] = gkey;
    bool c2;
    const long long cs = h_find_or_insert(m.ch_keys, m.ch_mask, gkey, &c2);
    if (cs < 0) { m.sc[SC_OVERFLOW] = 3; m.cand_next[i] = -1; }
    else m.cand_next[i] = atomicExch(&m.ch_head[cs], i);
    st_agent(&m.cand_status[i], ST_UNDECIDED);
    ADBG(4);   // survivors: chain insert
    return probes;
}
__global__ __launch_bounds__(256) void mesh_append_prepare_kernel(MeshDev m_in, const float* __restrict__ pts_arg) {
    MESH_MARK(m_in, 1);
    // (everything a thread needs to find its candidate is requested before anything is waited for: the slot's fill count, the slot, the scan's parameters)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int* __restrict__ cnt = m_in.bin_cnt;
    const int my_cnt = t < MV_BIN_NSLOT ? cnt[t / MV_BIN_SLOTS] : cnt[MV_BIN_BUCKETS];
    const float4 slot_pt = m_in.cand_pt[t];   // (allocated for MV_BIN_NSLOT + cap_cand entries: always readable)
    MESH_DYN(m_in);
    const float* __restrict__ pts = pts_arg ? pts_arg : dyn_pts;
    __shared__ int s_c1, s_nvis, s_base;
    __shared__ int s_vis[1024];
    if (threadIdx.x == 0) { s_c1 = 0; s_nvis = 0; }
    if (t == 0 && cnt[MV_BIN_BUCKETS + 1]) m.sc[SC_UNDECIDED] = 1;   // a workgroup of mesh_begin_scan_kernel never saw the scan arrive: its candidates are missing
    __syncthreads();
    const unsigned long long tk0 = m.dbg ? __builtin_readcyclecounter() : 0;
    // admission order: the slots of the 8-cell cubes, then -- further workgroups of the same launch -- the overflow list (mesh_begin_scan_kernel);
    // offline-sized clouds: scan order
    bool live;
    int i = t;
    float4 pv = slot_pt;
    if (sp.n_cand <= MV_FIN_CAND) {
        live = t < MV_BIN_NSLOT ? (t & (MV_BIN_SLOTS - 1)) < my_cnt : (t - MV_BIN_NSLOT) < min(my_cnt, m.cap_cand);
        i = __float_as_int(pv.w);
    } else {
        live = t < sp.n_cand;
        if (live) pv = *(const float4*)(pts + 4 * (size_t)t * sp.step);
    }
    if (m.dbg && lane == 0) { atomicAdd(&m.dbg[32 + 5], __builtin_readcyclecounter() - tk0); atomicAdd(&m.dbg[32 + 8], 1ull); }
    int probes = 0;
    if (__any(live)) {
        int vnew;
        probes = mesh_admit_candidate(m, live, i, pv.x, pv.y, pv.z, lane, &vnew);
        if (vnew >= 0) { const int k = atomicAdd(&s_nvis, 1); if (k < 1024) s_vis[k] = vnew; else m.recent[atomicAdd(&m.sc[SC_RECENT], 1)] = vnew; }
    }
    // one atomic per workgroup on the scan-wide counters (hundreds of wavefronts adding to ONE address serialise at the memory side)
    if (probes) atomicAdd(&s_c1, probes);
    __syncthreads();
    const int nv = min(s_nvis, 1024);
    if (threadIdx.x == 0) {
        if (s_c1) atomicAdd(&m.sc[SC_C1], s_c1);
        s_base = nv ? atomicAdd(&m.sc[SC_RECENT], nv) : 0;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nv; k += 256) m.recent[s_base + k] = s_vis[k];
    if (m.dbg && threadIdx.x == 0) atomicMax(&m.dbg[32 + 6], __builtin_readcyclecounter() - tk0);   // the slowest workgroup, start to end
}

// Candidate i is accepted iff no ACCEPTED candidate j < i shares its cell or lies within min_spacing: exactly the sequential
// loop's outcome.  Each lane re-evaluates until every lower-index conflicting candidate is decided (bounded; relaunched by the host).
// Round 6: the FIRST evaluation of a candidate walks the 27 cells around it (three batches of nine lookups + the chains under them: tens of dependent
// round trips) and keeps the conflicting lower-index candidates that were not decided yet in a per-thread list in LDS; every later evaluation only polls
// those candidates' status words -- one round trip.  On fresh ground the "lowest scan index wins" chains are long (a survey lattice offered in row order:
// every accepted point decides its neighbour, which decides the next -- hundreds of rounds), and every round used to repeat the whole walk: 1.2 ms
// average over the bench's seeding packages, 4-6 ms at worst.  The candidate set under the cells does not change during a launch and a status only
// moves from UNDECIDED to a final value, so the cached list decides exactly what the walk would.
#define RC_MAX 14   /* cached undecided conflicts per candidate (more: the candidate keeps walking) */
__global__ __launch_bounds__(256) void mesh_append_resolve_kernel(MeshDev m_in, const float* __restrict__ pts_arg, int max_iter) {
    MESH_MARK(m_in, 2);
    MESH_DYN(m_in);
    __shared__ int s_conf[RC_MAX][256];
    const float* __restrict__ pts = pts_arg ? pts_arg : dyn_pts;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int tid = threadIdx.x;
    const bool sharded = m.shard_world > 1;
    // Lanes of one wavefront may depend on each other, so the decision store must happen INSIDE the loop body and the loop must be left
    // by the whole wavefront together (__all): with a per-lane `return` the compiler may sink the store to the loop exit, which the
    // waiting lanes of the same wavefront would then never see.
    int my = ST_REJECT;   // lanes with nothing to decide just ride along
    float px = 0, py = 0, pz = 0;
    long gx = 0, gy = 0, gz = 0;
    unsigned long long own = 0;
    // (sharded: a rank decides only its own candidates; the other ranks' band survivors sit in the chains as undecided until their owners report)
    if (i < sp.n_cand && ld_agent(&m.cand_status[i]) == ST_UNDECIDED && (!sharded || (m.cand_flags[i] & CF_OWN))) {
        my = ST_UNDECIDED;
        const float* p = pts + 4 * (size_t)i * sp.step;
        px = p[0]; py = p[1]; pz = p[2];
        gx = rnd_cell(px, m.min_spacing); gy = rnd_cell(py, m.min_spacing); gz = rnd_cell(pz, m.min_spacing);
        own = m.cand_cell[i];
    }
    int n_conf = -1;   // -1: not walked yet; >= 0: the cached list is complete; -2: more than RC_MAX undecided conflicts (keeps walking)
    for (int iter = 0; iter < max_iter; iter++) {
        if (my == ST_UNDECIDED) {
            bool rej = false, blocked = false, blocked_remote = false;
            if (n_conf >= 0) {
                // ---- later rounds: poll the cached conflicts (all loads in flight together)
                int sj[RC_MAX];
#pragma unroll
                for (int k = 0; k < RC_MAX; k++) sj[k] = k < n_conf ? ld_agent(&m.cand_status[s_conf[k][tid]]) : ST_REJECT;
#pragma unroll
                for (int k = 0; k < RC_MAX; k++) {
                    if (k < n_conf && sj[k] != ST_REJECT) {
                        if (sj[k] == ST_ACCEPT) rej = true;
                        else if (sj[k] == ST_WAIT || (sharded && !(m.cand_flags[s_conf[k][tid]] & CF_OWN))) blocked_remote = true;
                        else blocked = true;
                    }
                }
            } else {
                int nc = 0;
                bool over = false;
                for (int dx = -1; dx <= 1 && !rej; dx++) {
                    unsigned long long key9[9], k9[9], h9[9];
                    int hd9[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) {  // 9 independent cell lookups in flight
                        key9[q] = mkey(gx + dx, gy + (q / 3 - 1), gz + (q % 3 - 1));
                        h9[q] = hash64(key9[q]) & m.ch_mask;
                        k9[q] = m.ch_keys[h9[q]];
                        hd9[q] = m.ch_head[h9[q]];
                    }
#pragma unroll
                    for (int q = 0; q < 9; q++) {
                        int head = -1;
                        if (k9[q] == key9[q]) head = hd9[q];
                        else if (k9[q] != MKEY_EMPTY) { const long long s2 = h_find(m.ch_keys, m.ch_mask, key9[q]); if (s2 >= 0) head = m.ch_head[s2]; }
                        for (int j = head; j >= 0 && !rej;) {
                            const int nxt = m.cand_next[j];
                            const int sj = ld_agent(&m.cand_status[j]);
                            const float4 qv = *(const float4*)(pts + 4 * (size_t)j * sp.step);
                            if (j < i && sj != ST_REJECT) {
                                const bool conflict = (key9[q] == own) || ((double)sqrtf(dist2f(px, py, pz, qv.x, qv.y, qv.z)) < m.min_spacing);
                                if (conflict) {
                                    if (sj == ST_ACCEPT) rej = true;
                                    else {
                                        if (sj == ST_WAIT || (sharded && !(m.cand_flags[j] & CF_OWN))) blocked_remote = true;   // nothing in THIS launch will decide j
                                        else blocked = true;
                                        if (nc < RC_MAX) s_conf[nc][tid] = j; else over = true;
                                        nc++;
                                    }
                                }
                            }
                            j = nxt;
                        }
                    }
                }
                n_conf = over ? -2 : nc;   // (a walk cut short by a rejection leaves a partial list behind: the candidate is decided, nobody reads it)
            }
            if (rej) my = ST_REJECT; else if (!blocked && !blocked_remote) my = ST_ACCEPT; else if (!blocked) my = ST_WAIT;
            if (my != ST_UNDECIDED) st_agent(&m.cand_status[i], my);
        }
        if (__all(my != ST_UNDECIDED)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    if (my == ST_UNDECIDED) atomicAdd(&m.sc[SC_UNDECIDED], 1);
}

// ---- sharded admission: what the ranks tell each other (SURVEY 8(e) "Mesh append": the boundary band) -------------------------------------------------
// mesh_cand_pack_kernel    this rank's candidates that matter elsewhere and have not been sent yet: band survivors of the test against the map (as
//                          UNDECIDED: the neighbours chain them), then decisions -- every ACCEPT (the 16-byte vertex commit is replicated on all ranks, so
//                          ids stay the serial ones: "exclusive scan over ranks in scan-index order" is the prefix sum every rank runs over the same accept
//                          flags) and the REJECTs of band survivors (they unblock the neighbours).  Rejections against the map and of interior survivors
//                          never travel.  count[0] += records, count[1] += candidates this rank has still to decide.
// mesh_cand_unpack_kernel  another rank's records: a band survivor that lies within min_spacing of one of this rank's bricks is chained under its cell;
//                          decisions overwrite the status.
__global__ __launch_bounds__(256) void mesh_cand_pack_kernel(MeshDev m_in, MeshCdRec* __restrict__ out, int32_t* __restrict__ count, int cap_rec) {
    MESH_DYN(m_in);
    const int lane = threadIdx.x & 63;
    for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < sp.n_cand; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + lane;
        int send = -1, und = 0;
        if (i < sp.n_cand) {
            int f = m.cand_flags[i];
            if (f & CF_OWN) {
                int st = m.cand_status[i];
                if (st == ST_WAIT) { st = ST_UNDECIDED; m.cand_status[i] = ST_UNDECIDED; }
                if (st == ST_UNDECIDED) {
                    und = 1;
                    if ((f & CF_BAND) && !(f & CF_SURV_SENT)) { send = ST_UNDECIDED; f |= CF_SURV_SENT; }
                } else if (!(f & CF_DEC_SENT) && (st == ST_ACCEPT || (f & CF_BAND))) { send = st; f |= CF_DEC_SENT | CF_SURV_SENT; }
                if (send >= 0) m.cand_flags[i] = f;
            }
        }
        const unsigned long long mask = __ballot(send >= 0), umask = __ballot(und != 0);
        int base = 0;
        if (lane == 0) {
            if (mask) base = atomicAdd(&count[0], (int)__popcll(mask));
            if (umask) atomicAdd(&count[1], (int)__popcll(umask));
        }
        base = __shfl(base, 0, 64);
        if (send >= 0) {
            const int pos = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
            if (pos < cap_rec) { MeshCdRec r; r.i = i; r.status = send; out[pos] = r; }   // (beyond the block's capacity: the count in the header says so to everybody)
        }
    }
}
// every rank's block of the gathered buffer: 16-byte header {records, aux, -, -}, then the records.  One launch unpacks all the other ranks' blocks with
// the counts read on the device (no host look at the gather unless the host has a decision to take)
IMD int xblock_count(const MeshDev& m, const char* gathered, size_t capb, int r, int cap_rec) {
    const int n = *(const int32_t*)(gathered + (size_t)r * capb);
    if (n > cap_rec || n < 0) { m.sc[SC_OVERFLOW] = 14; return 0; }   // a band above the exchange block: every rank sees it and fails the scan
    return n;
}
__global__ __launch_bounds__(256) void mesh_cand_unpack_kernel(MeshDev m_in, const char* __restrict__ gathered, size_t capb, int cap_rec) {
    MESH_DYN(m_in);
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gstride = gridDim.x * blockDim.x;
    if (gtid == 0) atomicAdd(&m.sc[SC_XBYTES], xblock_count(m, gathered, capb, m.shard_rank, cap_rec) * (int)sizeof(MeshCdRec));
    for (int r = 0; r < m.shard_world; r++) {
        if (r == m.shard_rank) continue;
        const int n = xblock_count(m, gathered, capb, r, cap_rec);
        const MeshCdRec* in = (const MeshCdRec*)(gathered + (size_t)r * capb + 16);
        for (int k = gtid; k < n; k += gstride) {
            const MeshCdRec rec = in[k];
            const int j = rec.i;
            if (j < 0 || j >= sp.n_cand) continue;
            const float4 p = *(const float4*)(dyn_pts + 4 * (size_t)j * sp.step);
            const unsigned long long gkey = mkey(rnd_cell(p.x, m.min_spacing), rnd_cell(p.y, m.min_spacing), rnd_cell(p.z, m.min_spacing));
            const int f = m.cand_flags[j];
            if (rec.status == ST_UNDECIDED) {
                if (f & CF_KNOWN) continue;
                long lo[3], hi[3];
                cand_box(m, p.x, p.y, p.z, lo, hi);
                if (!box_has(m, lo, hi, m.shard_rank)) continue;   // not within min_spacing of anything this rank decides
                m.cand_flags[j] = CF_KNOWN;
                m.cand_cell[j] = gkey; m.cand_vox[j] = -1;
                bool c2;
                const long long cs = h_find_or_insert(m.ch_keys, m.ch_mask, gkey, &c2);
                if (cs < 0) { m.sc[SC_OVERFLOW] = 3; m.cand_next[j] = -1; }
                else m.cand_next[j] = atomicExch(&m.ch_head[cs], j);
                st_agent(&m.cand_status[j], ST_UNDECIDED);
            } else {
                if (rec.status == ST_ACCEPT && !(f & CF_KNOWN)) { m.cand_cell[j] = gkey; m.cand_vox[j] = -1; }   // (what the replicated commit reads)
                st_agent(&m.cand_status[j], rec.status);
            }
        }
    }
}
void launch_mesh_cand_pack(hipStream_t s, const MeshDev& m, MeshCdRec* out, int32_t* count, int cap_rec) { KLAUNCH(mesh_cand_pack_kernel, dim3(64), dim3(256), 0, s, m, out, count, cap_rec); }
void launch_mesh_cand_unpack(hipStream_t s, const MeshDev& m, const void* gathered, size_t capb, int cap_rec) {
    KLAUNCH(mesh_cand_unpack_kernel, dim3(32), dim3(256), 0, s, m, (const char*)gathered, capb, cap_rec);
}

__global__ void mesh_append_flags_kernel(MeshDev m_in) {
    MESH_DYN(m_in);
    const int n = sp.n_cand;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) m.cand_rank[i] = (m.cand_status[i] == ST_ACCEPT) ? 1 : 0;
}

// new vertex id = vtx_base + (number of accepted candidates with a lower scan index): ids grow in scan order as in the reference
__global__ __launch_bounds__(256) void mesh_append_commit_kernel(MeshDev m_in, const float* __restrict__ pts_arg) {
    MESH_DYN(m_in);
    const float* __restrict__ pts = pts_arg ? pts_arg : dyn_pts;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sp.n_cand) return;
    const bool acc = m.cand_status[i] == ST_ACCEPT;
    if (i == sp.n_cand - 1) {
        const int total = m.cand_rank[i] + (acc ? 1 : 0);
        m.sc[SC_ACCEPTED] = total;
        m.pc[PC_VERTS] = sp.vtx_base + total;
    }
    if (!acc) return;
    const int id = sp.vtx_base + m.cand_rank[i];
    if (id >= m.cap_verts) { m.sc[SC_OVERFLOW] = 4; return; }
    const float* p = pts + 4 * (size_t)i * sp.step;
    const float px = p[0], py = p[1], pz = p[2];
    m.v_pos[(size_t)id * 3 + 0] = px; m.v_pos[(size_t)id * 3 + 1] = py; m.v_pos[(size_t)id * 3 + 2] = pz;
    m.v_smooth[(size_t)id * 3 + 0] = (double)px; m.v_smooth[(size_t)id * 3 + 1] = (double)py; m.v_smooth[(size_t)id * 3 + 2] = (double)pz;
    int vi = m.cand_vox[i];
    if (vi < 0) {
        vi = vox_find(m, mkey(rnd_cell(px, m.voxel), rnd_cell(py, m.voxel), rnd_cell(pz, m.voxel)));
    }
    if (vi < 0) { m.sc[SC_OVERFLOW] = 5; return; }
    m.v_voxel[id] = vi;
    long cgx, cgy, cgz;
    mkey_unpack(m.cand_cell[i], cgx, cgy, cgz);
    const long long gs = grid_insert(m, cgx, cgy, cgz, m.cand_cell[i]);
    if (gs < 0) { m.sc[SC_OVERFLOW] = 6; return; }
    *(float4*)&m.g_ent[gs] = make_float4(px, py, pz, __int_as_float(id));
    const int pos = atomicAdd(&m.vx_npts[vi], 1);
    if (pos >= MV_VOX_CAP) { m.sc[SC_OVERFLOW] = 7; atomicSub(&m.vx_npts[vi], 1); return; }
    m.vx_pts[(size_t)vi * MV_VOX_CAP + pos] = id;
    atomicAdd(&m.vx_new_added[vi], 1);
    m.vx_meshing_times[vi] = 0;
}

// voxels to (re)mesh this scan: visited, m_meshing_times < 1, m_new_added_pts_count >= 0, >= 3 vertices
// (ImMesh_mesh_reconstruction.cpp:132-151)
__global__ void mesh_select_active_kernel(MeshDev m_in) {
    MESH_DYN(m_in);
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m.sc[SC_RECENT]) return;
    const int vi = m.recent[r];
    if (m.vx_meshing_times[vi] >= 1 || m.vx_new_added[vi] < 0) return;
    m.vx_meshing_times[vi] = m.vx_meshing_times[vi] + 1;
    m.vx_new_added[vi] = 0;
    if (m.vx_npts[vi] < 3) return;
    const int a = atomicAdd(&m.sc[SC_ACTIVE], 1);
    if (a >= m.cap_active) { m.sc[SC_OVERFLOW] = 8; return; }
    m.act_key[a] = m.vx_key[vi];
    m.act_vox[a] = vi;
}

// The tail of vertex admission for per-scan sized candidate sets (<= MV_FIN_CAND), ONE launch of one 1024-thread workgroup instead of seven
// (flags, 2 x device scan, commit, select, 2 x sort): new vertex ids by a block-wide prefix sum over the accept flags (ids grow in scan order as in
// the reference), the commit of mesh_append_commit_kernel, the voxel selection of mesh_select_active_kernel, and the ascending-(x, y, z) order of
// the active voxels (bitonic network in LDS) that defines "earlier / later voxel" for the order-dependent parts.  Every one of those launches
// was a few microseconds of work behind ~5 us of launch latency on the mesher's phase-A chain.
IMD int next_pow2_i(int v) { int p = 1; while (p < v) p <<= 1; return p; }
#define MV_FIN_ACT 8192        /* visited voxels whose selection + ordering run in LDS; more than that (sparse far-field scans) take the global arrays */
// (~100 KB of static LDS for one workgroup: the 160 KB of a gfx950 CU -- the only target of this library, see the Makefile -- is assumed)
__global__ __launch_bounds__(1024) void mesh_append_finish_kernel(MeshDev m_in, const float* __restrict__ pts_arg) {
    MESH_MARK(m_in, 3);
    MESH_DYN(m_in);
    const float* __restrict__ pts = pts_arg ? pts_arg : dyn_pts;
    __shared__ unsigned long long skey[MV_FIN_ACT];
    __shared__ int svox[MV_FIN_ACT];
    __shared__ int sscan[1024];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int n = sp.n_cand;
    const int per = (n + 1023) / 1024;            // <= 16 consecutive candidates per thread: ids stay in scan order
    const int i0 = tid * per, i1 = min(n, i0 + per);
    int local = 0;
    for (int i = i0; i < i1; i++) local += (m.cand_status[i] == ST_ACCEPT) ? 1 : 0;
    sscan[tid] = local;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        const int v = tid >= off ? sscan[tid - off] : 0;
        __syncthreads();
        sscan[tid] += v;
        __syncthreads();
    }
    const int total = sscan[1023];
    int rank = sscan[tid] - local;
    if (tid == 0) { m.sc[SC_ACCEPTED] = total; m.pc[PC_VERTS] = sp.vtx_base + total; }
    // the hang guard's "a workgroup never saw the scan arrive" survives the sharded admission's rounds (which zero SC_UNDECIDED before every resolve,
    // ADVICE r04): candidates are missing, the job has to fail
    if (tid == 0 && m.bin_cnt[MV_BIN_BUCKETS + 1]) m.sc[SC_UNDECIDED] = 1;
    // the accepted candidates, compacted in scan order (accepted candidates cluster: a thread's own 11 may hold several, and each commit is a chain
    // of dependent global round trips -- so the commits are dealt out one per thread from the compacted list)
    int* clist = (int*)skey;   // MV_FIN_CAND ints == the key array's bytes; the keys are not in use yet
    for (int i = i0; i < i1; i++)
        if (m.cand_status[i] == ST_ACCEPT) clist[rank++] = i;
    __syncthreads();
    for (int e = tid; e < total; e += 1024) {
        const int i = clist[e];
        const int id = sp.vtx_base + e;
        if (id >= m.cap_verts) { m.sc[SC_OVERFLOW] = 4; continue; }
        const float* p = pts + 4 * (size_t)i * sp.step;
        const float px = p[0], py = p[1], pz = p[2];
        m.v_pos[(size_t)id * 3 + 0] = px; m.v_pos[(size_t)id * 3 + 1] = py; m.v_pos[(size_t)id * 3 + 2] = pz;
        m.v_smooth[(size_t)id * 3 + 0] = (double)px; m.v_smooth[(size_t)id * 3 + 1] = (double)py; m.v_smooth[(size_t)id * 3 + 2] = (double)pz;
        int vi = m.cand_vox[i];
        if (vi < 0) {
            vi = vox_find(m, mkey(rnd_cell(px, m.voxel), rnd_cell(py, m.voxel), rnd_cell(pz, m.voxel)));
        }
        if (vi < 0) { m.sc[SC_OVERFLOW] = 5; continue; }
        m.v_voxel[id] = vi;
        long cgx, cgy, cgz;
        mkey_unpack(m.cand_cell[i], cgx, cgy, cgz);
        const long long gs = grid_insert(m, cgx, cgy, cgz, m.cand_cell[i]);
        if (gs < 0) { m.sc[SC_OVERFLOW] = 6; continue; }
        *(float4*)&m.g_ent[gs] = make_float4(px, py, pz, __int_as_float(id));
        const int pos = atomicAdd(&m.vx_npts[vi], 1);
        if (pos >= MV_VOX_CAP) { m.sc[SC_OVERFLOW] = 7; atomicSub(&m.vx_npts[vi], 1); continue; }
        m.vx_pts[(size_t)vi * MV_VOX_CAP + pos] = id;
        atomicAdd(&m.vx_new_added[vi], 1);
        st_agent(&m.vx_meshing_times[vi], 0);
    }
    __threadfence();
    __syncthreads();
    // ---- voxels to (re)mesh this scan: visited, m_meshing_times < 1, m_new_added_pts_count >= 0, >= 3 vertices (ImMesh_mesh_reconstruction.cpp:132-151),
    //      then their ascending-(x, y, z) order.  A voxel goes active with ONE new vertex beside two older ones, so the active set is bounded by the
    //      visited voxels, not by a third of the candidates: the LDS arrays serve up to MV_FIN_ACT visited voxels (any dense scan), a sparse far-field
    //      scan that visits more selects and orders in the global act_key / act_vox arrays -- same network, global round trips, still one launch.
    const int n_recent = m.sc[SC_RECENT];
    auto select_and_order = [&](auto* K, auto* V, const int cap) {
        for (int r = tid; r < n_recent; r += 1024) {
            const int vi = m.recent[r];
            if (ld_agent(&m.vx_meshing_times[vi]) >= 1 || ld_agent(&m.vx_new_added[vi]) < 0) continue;
            st_agent(&m.vx_meshing_times[vi], ld_agent(&m.vx_meshing_times[vi]) + 1);
            st_agent(&m.vx_new_added[vi], 0);
            if (ld_agent(&m.vx_npts[vi]) < 3) continue;
            const int a = atomicAdd(&s_n, 1);
            if (a >= cap) { m.sc[SC_OVERFLOW] = 8; continue; }
            K[a] = m.vx_key[vi];
            V[a] = vi;
        }
        __syncthreads();
        const int na = min(s_n, cap);
        const int np2 = next_pow2_i(max(na, 1));
        for (int k = na + tid; k < np2; k += 1024) { K[k] = ~0ull; V[k] = -1; }
        __syncthreads();
        for (int k = 2; k <= np2; k <<= 1)            // ascending packed key == ascending (x, y, z); keys are unique (one entry per voxel)
            for (int lj = 31 - __clz(k >> 1); lj >= 0; lj--) {
                const int j = 1 << lj;
                for (int p = tid; p < (np2 >> 1); p += 1024) {
                    const int i = ((p >> lj) << (lj + 1)) | (p & (j - 1)), ixj = i + j;
                    const bool up = ((i & k) == 0);
                    const unsigned long long x = K[i], y = K[ixj];
                    if ((x > y) == up) { K[i] = y; K[ixj] = x; const int t = V[i]; V[i] = V[ixj]; V[ixj] = t; }
                }
                __syncthreads();
            }
        for (int r = tid; r < na; r += 1024) {
            const int vi = V[r];
            m.act_vox_s[r] = vi;
            m.vx_rank[vi] = r;
            m.vx_rank_seq[vi] = m.seq;
        }
        if (tid == 0) m.sc[SC_ACTIVE] = na;
    };
    if (n_recent <= MV_FIN_ACT) select_and_order(skey, svox, MV_FIN_ACT);
    else select_and_order(m.act_key, m.act_vox, m.cap_active);   // (the arrays are allocated up to the next power of two: the network pads)
}
void launch_mesh_append_finish(hipStream_t s, const MeshDev& m, const float* pts) { KLAUNCH(mesh_append_finish_kernel, dim3(1), dim3(1024), 0, s, m, pts); }

// =====================================================================================================================
// LDS helpers
// =====================================================================================================================
template <typename T, int NT>
IMD void lds_bitonic_sort(T* a, int N, int tid) {  // N power of two; ascending; every thread of the workgroup participates
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < N; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = ((i & k) == 0);
                    const T x = a[i], y = a[ixj];
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
}
IMD int lds_bsearch_i32(const int* a, int n, int key) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) { const int mid = (lo + hi) >> 1; const int v = a[mid]; if (v == key) return mid; if (v < key) lo = mid + 1; else hi = mid - 1; }
    return -1;
}
IMD int lds_bsearch_u32(const unsigned int* a, int n, unsigned int key) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) { const int mid = (lo + hi) >> 1; const unsigned int v = a[mid]; if (v == key) return mid; if (v < key) lo = mid + 1; else hi = mid - 1; }
    return -1;
}
IMD float readlane_f(float x, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), k)); }
IMD unsigned long long wave_min_u64(unsigned long long x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long y = __shfl_xor(x, off, 64); x = y < x ? y : x; }
    return x;
}
IMD float wave_min_f(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fminf(x, __shfl_xor(x, off, 64));
    return x;
}
IMD float wave_max_f(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}
IMD double wave_min_d(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmin(x, __shfl_xor(x, off, 64));
    return x;
}
IMD double wave_max_d(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off, 64));
    return x;
}

// =====================================================================================================================
// 20-NN neighbourhood pull + smoothing: one workgroup (4 wavefronts) per active voxel
// =====================================================================================================================
#define KC 500                  /* candidates staged per batch (a 20-NN pull of the shipped configurations inspects ~200 per query); with the per-wave work
                                   lists this sets the kernel's LDS footprint: 56 KB -> two workgroups per CU still leave room for the other chains' kernels */
#define WL (KC + MV_KNN + 4)    /* per-wave work list */
#define HSET 4096

// EXPORT = false: the per-scan neighbourhood pull.  EXPORT = true: Global_map::smooth_pts over EVERY vertex for save_to_ply_file
// (mesh_rec_geometry.cpp:71-131, pointcloud_rgbd.cpp:932-958): the same 20-NN machinery run over all mesh voxels, output = the exported
// vertex position pt*(1-f) + mean(neighbours 1..19 closer than accept)*f  (the nearest neighbour, the vertex itself, is skipped; no neighbour
// -> 0/0 = NaN, as in the reference); nothing of the map is modified.
// QUERY (round 6; EXPORT with a voxel list): Global_map::smooth_pts for the vertices of the listed mesh voxels only -- what the renderer asks for
// (mesh_rec_display.cpp:78-103) -- as doubles into export_d (3 per vertex id), neighbours closer than `max_dis`; export_vtx is not written.
template <bool EXPORT>
__global__ __launch_bounds__(256) void mesh_knn_kernel(MeshDev m_in, float* __restrict__ export_vtx, double smooth_factor, const int32_t* __restrict__ vox_list, int n_list,
                                                        double* __restrict__ export_d, double max_dis) {
    if (!EXPORT) MESH_MARK(m_in, 4);
    MESH_DYN(m_in);
    __shared__ float cx[KC], cy[KC], cz[KC];
    __shared__ int cid[KC];
    __shared__ unsigned long long best[MV_VOX_CAP][MV_KNN];
    __shared__ unsigned long long wl[4][WL];
    __shared__ unsigned long long sel[4][64];
    __shared__ int nbest[MV_VOX_CAP];
    __shared__ float qx[MV_VOX_CAP], qy[MV_VOX_CAP], qz[MV_VOX_CAP];
    __shared__ int qid[MV_VOX_CAP];
    __shared__ int rel_l[MV_REL_CAP];
    __shared__ int wsum[4];
    __shared__ int s_incl[256], s_v2[256];
    __shared__ int s_misc[8];   // 0 ncand, 1 vend, 2 nrel, 3 any-needs-pass-2
    __shared__ long s_box[6];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!EXPORT && blockIdx.x == 0)   // the admission's bucket fill counts: back to zero for the next scan (mesh_begin_scan_kernel counts, mesh_append_prepare_kernel reads)
        for (int k = tid; k <= MV_BIN_BUCKETS + 1; k += 256) m.bin_cnt[k] = 0;
    const int n_active = EXPORT ? (vox_list ? n_list : m.pc[PC_VOXELS]) : min(m.sc[SC_ACTIVE], m.cap_active);   // launch size is fixed; the work list length lives on the device
    for (int r = blockIdx.x; r < n_active; r += gridDim.x) {
    unsigned long long tprev = m.dbg ? __builtin_readcyclecounter() : 0;
    const unsigned long long tvox0 = tprev;
#define KDBG(k) do { if (m.dbg) { const unsigned long long _t = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&m.dbg[k], _t - tprev); tprev = _t; } } while (0)
    const int vi = EXPORT ? (vox_list ? vox_list[r] : r) : m.act_vox_s[r];
    const int nq = min(m.vx_npts[vi], MV_VOX_CAP);
    if (EXPORT && nq == 0) continue;
    if (!EXPORT && m.shard_world > 1 && mesh_owner(m, m.vx_key[vi]) != m.shard_rank) {
        if (tid == 0) { m.rel_n[r] = 0; m.rel_nq[r] = nq; }   // another rank searches and triangulates this voxel; its results arrive by all-gather
        continue;
    }
    if (tid < nq) {
        const int id = m.vx_pts[(size_t)vi * MV_VOX_CAP + tid];
        qid[tid] = id;
        qx[tid] = m.v_pos[(size_t)id * 3 + 0]; qy[tid] = m.v_pos[(size_t)id * 3 + 1]; qz[tid] = m.v_pos[(size_t)id * 3 + 2];
        nbest[tid] = 0;
    }
    if (tid == 0) s_misc[3] = 0;
    __syncthreads();
    long long inspected = 0;
    const double r_max = m.accept * 2.0;  // retrieve_neighbor_pts_kdtree only uses neighbours closer than 2 x accept
    for (int pass = 0; pass < 2; pass++) {
        // pass 0: radius r_max/2; when it already holds >= 20 neighbours they are the global 20 nearest.  pass 1 (rare): radius r_max.
        const double rr = (pass == 0 ? r_max * 0.5 : r_max) * 1.001 + 1e-6;
        if (pass == 1) {
            if (tid < nq) { const bool need = nbest[tid] < MV_KNN; if (need) { nbest[tid] = 0; s_misc[3] = 1; } else nbest[tid] = nbest[tid] | 0x10000; }
            __syncthreads();
            if (!s_misc[3]) break;
            if (tid == 0) atomicAdd(&m.sc[SC_PASS2], 1);
        }
        if (wv == 0) {  // voxel-index box covering every query's ball (index of x is round(x/voxel): monotone)
            float mnx = 3e38f, mny = 3e38f, mnz = 3e38f, mxx = -3e38f, mxy = -3e38f, mxz = -3e38f;
            for (int q = lane; q < nq; q += 64) {
                mnx = fminf(mnx, qx[q]); mny = fminf(mny, qy[q]); mnz = fminf(mnz, qz[q]);
                mxx = fmaxf(mxx, qx[q]); mxy = fmaxf(mxy, qy[q]); mxz = fmaxf(mxz, qz[q]);
            }
            mnx = wave_min_f(mnx); mny = wave_min_f(mny); mnz = wave_min_f(mnz);
            mxx = wave_max_f(mxx); mxy = wave_max_f(mxy); mxz = wave_max_f(mxz);
            if (lane == 0) {
                s_box[0] = (long)round(((double)mnx - rr) / m.voxel); s_box[1] = (long)round(((double)mxx + rr) / m.voxel);
                s_box[2] = (long)round(((double)mny - rr) / m.voxel); s_box[3] = (long)round(((double)mxy + rr) / m.voxel);
                s_box[4] = (long)round(((double)mnz - rr) / m.voxel); s_box[5] = (long)round(((double)mxz + rr) / m.voxel);
            }
        }
        __syncthreads();
        const long bx0 = s_box[0], by0 = s_box[2], bz0 = s_box[4];
        const int ex = (int)(s_box[1] - bx0 + 1), ey = (int)(s_box[3] - by0 + 1), ez = (int)(s_box[5] - bz0 + 1);
        const int nvox = ex * ey * ez;
        int vstart = 0;
        while (vstart < nvox) {
            // ---- stage the next batch of voxels (<= 256 voxels, <= KC vertices) into LDS
            int v2 = -1, n2 = 0;
            const int vidx = vstart + tid;
            if (vidx < nvox) {
                const int iz = vidx % ez, iy = (vidx / ez) % ey, ix = vidx / (ez * ey);
                v2 = vox_find(m, mkey(bx0 + ix, by0 + iy, bz0 + iz));
                if (v2 >= 0) n2 = min(m.vx_npts[v2], MV_VOX_CAP);
            }
            int incl = n2;  // inclusive scan over the 256 threads
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
            if (lane == 63) wsum[wv] = incl;
            if (tid == 0) { s_misc[0] = 0; s_misc[1] = min(vstart + 256, nvox); }
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wv; w++) woff += wsum[w];
            incl += woff;
            const int excl = incl - n2;
            const bool fits = incl <= KC;
            if (!fits && excl <= KC && vidx < nvox) s_misc[1] = vidx;  // first voxel that does not fit starts the next batch (unique writer)
            s_incl[tid] = fits ? incl : 0x7FFFFFFF;
            s_v2[tid] = v2;
            if (fits && n2 > 0) atomicMax(&s_misc[0], incl);
            __syncthreads();
            {   // copy at vertex granularity: thread c stages candidate c (voxel found by binary search in the prefix sums),
                // so the two dependent gathers (vertex id, position) happen once per thread instead of once per vertex of a voxel
                const int ntot = s_misc[0];
                for (int cpos = tid; cpos < ntot; cpos += 256) {
                    int lo = 0, hi = 255;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_incl[mid] > cpos) hi = mid; else lo = mid + 1; }
                    const int vv = s_v2[lo];
                    const int first = lo > 0 ? s_incl[lo - 1] : 0;  // == exclusive prefix of voxel lo (empty voxels repeat the previous value)
                    const int id = m.vx_pts[(size_t)vv * MV_VOX_CAP + (cpos - first)];
                    cx[cpos] = m.v_pos[(size_t)id * 3 + 0]; cy[cpos] = m.v_pos[(size_t)id * 3 + 1]; cz[cpos] = m.v_pos[(size_t)id * 3 + 2];
                    cid[cpos] = id;
                }
            }
            __syncthreads();
            KDBG(8 + 2 * pass);
            const int ncand = s_misc[0];
            vstart = s_misc[1];
            // ---- every query against the staged candidates: one wavefront per query
            for (int q = wv; q < nq; q += 4) {
                if (nbest[q] & 0x10000) continue;  // finished in pass 0
                const float ax = qx[q], ay = qy[q], az = qz[q];
                int nl = nbest[q];
                for (int k = lane; k < nl; k += 64) wl[wv][k] = best[q][k];
                for (int base = 0; base < ncand; base += 64) {
                    const int c = base + lane;
                    bool ok = false;
                    unsigned long long key = 0;
                    if (c < ncand) {
                        const float d2 = dist2f(ax, ay, az, cx[c], cy[c], cz[c]);
                        ok = (double)sqrtf(d2) < rr;
                        key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)cid[c];
                    }
                    const unsigned long long mask = __ballot(ok);
                    if (ok) wl[wv][nl + __popcll(mask & ((1ull << lane) - 1ull))] = key;
                    nl += __popcll(mask);
                }
                inspected += (lane == 0) ? ncand : 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                // the MV_KNN smallest (d2, id) keys, ascending.  Keys are distinct (ids are), so a key's output slot is the number of
                // smaller keys.  More than 64 keys: first a bisection on the d2 bit pattern (ballot counts, no cross-lane reduction)
                // finds the 20th smallest d2; the (normally exactly 20) keys at or below it are compacted and ranked the same way.
                const int cnt = min(MV_KNN, nl);
                const unsigned long long* src = wl[wv];
                int nsel = nl;
                bool ranked = true;
                if (nl > 64) {
                    unsigned int hreg[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int e = u * 64 + lane; hreg[u] = e < nl ? (unsigned int)(wl[wv][e] >> 32) : 0xFFFFFFFFu; }
                    auto count_le = [&](unsigned int mid) -> int {
                        int c = 0;
#pragma unroll
                        for (int u = 0; u < 4; u++) c += __popcll(__ballot(hreg[u] <= mid));
                        for (int e0 = 256; e0 < nl; e0 += 64) { const int e = e0 + lane; c += __popcll(__ballot(e < nl && (unsigned int)(wl[wv][e] >> 32) <= mid)); }
                        return c;
                    };
                    unsigned int lo = 0, hi = 0x7F7FFFFFu;
                    while (lo < hi) { const unsigned int mid = lo + ((hi - lo) >> 1); if (count_le(mid) >= cnt) hi = mid; else lo = mid + 1; }
                    nsel = count_le(lo);
                    if (nsel <= 64) {
                        int base = 0;
                        for (int e0 = 0; e0 < nl; e0 += 64) {
                            const int e = e0 + lane;
                            const unsigned long long key = e < nl ? wl[wv][e] : ~0ull;
                            const bool in = e < nl && (unsigned int)(key >> 32) <= lo;
                            const unsigned long long mask = __ballot(in);
                            if (in) sel[wv][base + __popcll(mask & ((1ull << lane) - 1ull))] = key;
                            base += __popcll(mask);
                        }
                        src = sel[wv];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    } else ranked = false;   // > 44 exact distance ties at the 20th neighbour: the slow, general extraction below
                }
                if (ranked) {
                    const unsigned long long mykey = lane < nsel ? src[lane] : ~0ull;
                    const unsigned int klo = (unsigned int)mykey, khi = (unsigned int)(mykey >> 32);
                    int rank = 0;
                    for (int j = 0; j < nsel; j++) {
                        const unsigned long long other = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)khi, j) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)klo, j);
                        rank += other < mykey ? 1 : 0;
                    }
                    if (lane < nsel && rank < MV_KNN) best[q][rank] = mykey;
                } else {
                    unsigned long long prev = 0;
                    for (int k = 0; k < cnt; k++) {
                        unsigned long long mine = ~0ull;
                        for (int e = lane; e < nl; e += 64) {
                            const unsigned long long v = wl[wv][e];
                            if ((k == 0 || v > prev) && v < mine) mine = v;
                        }
                        prev = wave_min_u64(mine);
                        if (lane == 0) best[q][k] = prev;
                    }
                }
                if (lane == 0) nbest[q] = cnt;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            __syncthreads();
            KDBG(9 + 2 * pass);
        }
    }
    __syncthreads();
    if (EXPORT) {
        for (int q = wv; q < nq; q += 4) {
            const int nb = nbest[q] & 0xFFFF;
            float nxp = 0, nyp = 0, nzp = 0;
            bool use_l = false;
            if (lane < nb) {
                const unsigned long long key = best[q][lane];
                const int id = (int)(unsigned int)(key & 0xFFFFFFFFull);
                use_l = lane >= 1 && (double)sqrtf(__uint_as_float((unsigned int)(key >> 32))) < (export_d ? max_dis : m.accept);   // k = 1 .. size-1, sqrt(dis) < maximum_smooth_dis
                nxp = m.v_pos[(size_t)id * 3 + 0]; nyp = m.v_pos[(size_t)id * 3 + 1]; nzp = m.v_pos[(size_t)id * 3 + 2];
            }
            double sx = 0, sy = 0, sz = 0, valid = 0.0;
            for (int k = 1; k < nb; k++) {
                const int use = __builtin_amdgcn_readlane((int)use_l, k);
                const float x = readlane_f(nxp, k), y = readlane_f(nyp, k), z = readlane_f(nzp, k);
                if (use) { valid += 1.0; sx += (double)x; sy += (double)y; sz += (double)z; }
            }
            if (lane == 0) {
                const int id = qid[q];
                const double p0 = (double)qx[q], p1 = (double)qy[q], p2 = (double)qz[q];
                const double o0 = p0 * (1.0 - smooth_factor) + sx * smooth_factor / valid, o1 = p1 * (1.0 - smooth_factor) + sy * smooth_factor / valid,
                             o2 = p2 * (1.0 - smooth_factor) + sz * smooth_factor / valid;
                if (export_d) { export_d[(size_t)id * 3 + 0] = o0; export_d[(size_t)id * 3 + 1] = o1; export_d[(size_t)id * 3 + 2] = o2; }
                else { export_vtx[(size_t)id * 3 + 0] = (float)o0; export_vtx[(size_t)id * 3 + 1] = (float)o1; export_vtx[(size_t)id * 3 + 2] = (float)o2; }
            }
        }
        __syncthreads();
        continue;
    }
    // ---- smoothing (mean of the neighbours closer than 2 x accept, smooth_factor 1.0) and the neighbourhood union (closer than accept)
    int* hset = (int*)&wl[0][0];
    for (int k = tid; k < HSET; k += 256) hset[k] = -1;
    if (tid == 0) s_misc[2] = 0;
    __syncthreads();
    for (int q = wv; q < nq; q += 4) {
        const int nb = nbest[q] & 0xFFFF;
        unsigned long long key = 0;
        float nxp = 0, nyp = 0, nzp = 0;
        bool in_acc = false, in_sm = false;
        if (lane < nb) {
            key = best[q][lane];
            const int id = (int)(unsigned int)(key & 0xFFFFFFFFull);
            const float d = sqrtf(__uint_as_float((unsigned int)(key >> 32)));
            in_acc = (double)d < m.accept;
            in_sm = (double)d < m.accept * 2.0;
            nxp = m.v_pos[(size_t)id * 3 + 0]; nyp = m.v_pos[(size_t)id * 3 + 1]; nzp = m.v_pos[(size_t)id * 3 + 2];
            if (in_acc) {
                unsigned int h = ((unsigned int)id * 2654435761u) & (HSET - 1);
                for (int probe = 0; probe < HSET; probe++) {
                    const int old = atomicCAS(&hset[h], -1, id);
                    if (old == -1 || old == id) break;
                    h = (h + 1) & (HSET - 1);
                }
            }
        }
        double sx = 0, sy = 0, sz = 0;
        int sc = 0;
        for (int k = 0; k < nb; k++) {  // neighbour order = ascending (d2, id), as returned by the tree search
            const int use = __builtin_amdgcn_readlane((int)in_sm, k);
            const float x = readlane_f(nxp, k), y = readlane_f(nyp, k), z = readlane_f(nzp, k);
            if (use) { sc++; sx += (double)x; sy += (double)y; sz += (double)z; }
        }
        if (lane == 0 && sc > 0) {
            const int id = qid[q];
            m.v_smooth_new[(size_t)id * 3 + 0] = sx / (double)sc;
            m.v_smooth_new[(size_t)id * 3 + 1] = sy / (double)sc;
            m.v_smooth_new[(size_t)id * 3 + 2] = sz / (double)sc;
        }
    }
    __syncthreads();
    for (int k = tid; k < HSET; k += 256) {
        const int v = hset[k];
        if (v >= 0) { const int pos = atomicAdd(&s_misc[2], 1); if (pos < MV_REL_CAP) rel_l[pos] = v; }
    }
    __syncthreads();
    int nrel = s_misc[2];
    if (nrel > MV_REL_CAP) { if (tid == 0) m.sc[SC_OVERFLOW] = 9; nrel = 0; }
    const int np2 = next_pow2_i(max(nrel, 1));
    for (int k = nrel + tid; k < np2; k += 256) rel_l[k] = 0x7FFFFFFF;
    __syncthreads();
    lds_bitonic_sort<int, 256>(rel_l, np2, tid);
    for (int k = tid; k < nrel; k += 256) m.rel_ids[(size_t)r * MV_REL_CAP + k] = rel_l[k];
    if (tid == 0) { m.rel_n[r] = nrel; m.rel_nq[r] = nq; atomicAdd(&m.sc[SC_NV], nq); atomicAdd(&m.sc[SC_NU], nrel); atomicMax(&m.sc[SC_MAXNU], nrel); }
    if (lane == 0 && inspected) atomicAdd(&m.sc[SC_C20], (int)inspected);
    __syncthreads();
    KDBG(12);
    if (m.dbg && tid == 0) atomicMax(&m.dbg[13], ((__builtin_readcyclecounter() - tvox0) << 16) | (unsigned long long)nq);
    }
}

// =====================================================================================================================
// per-voxel PCA -> 2-D Delaunay -> filter -> diff: one wavefront per active voxel
// =====================================================================================================================
#define DT_INF 0xFFFFu
#define DT_CAV_CAP 256

// CGAL Simple_cartesian<double> predicates as plain double expressions
IMD double orient2d(const double* p, const double* q, const double* r) { return (q[0] - p[0]) * (r[1] - p[1]) - (r[0] - p[0]) * (q[1] - p[1]); }
IMD double incircle2d(const double* p, const double* q, const double* r, const double* t) {
    const double qpx = q[0] - p[0], qpy = q[1] - p[1], rpx = r[0] - p[0], rpy = r[1] - p[1], tpx = t[0] - p[0], tpy = t[1] - p[1];
    return (qpx * tpy - qpy * tpx) * (rpx * (r[0] - q[0]) + rpy * (r[1] - q[1])) - (tpx * (t[0] - q[0]) + tpy * (t[1] - q[1])) * (qpx * rpy - qpy * rpx);
}
// does the circumdisk of the (possibly infinite) triangle contain p ?
IMD bool dt_in_disk(const double* xy, unsigned v0, unsigned v1, unsigned v2, const double* p) {
    int gi = -1;
    if (v0 == DT_INF) gi = 0; else if (v1 == DT_INF) gi = 1; else if (v2 == DT_INF) gi = 2;
    if (gi >= 0) {  // ghost: hull edge a->b seen from outside -> half-plane test; collinear: strictly between a and b
        const unsigned a = gi == 0 ? v1 : (gi == 1 ? v2 : v0), b = gi == 0 ? v2 : (gi == 1 ? v0 : v1);
        const double* A = xy + 2 * a; const double* B = xy + 2 * b;
        const double o = orient2d(A, B, p);
        if (o > 0) return true;
        if (o < 0) return false;
        const double d = (p[0] - A[0]) * (B[0] - A[0]) + (p[1] - A[1]) * (B[1] - A[1]);
        const double l = (B[0] - A[0]) * (B[0] - A[0]) + (B[1] - A[1]) * (B[1] - A[1]);
        return d > 0 && d < l;
    }
    return incircle2d(xy + 2 * v0, xy + 2 * v1, xy + 2 * v2, p) > 0;
}

// vertex position "after smoothing" as the sequential per-voxel loop would see it while meshing the voxel of rank my_rank:
// vertices of voxels already meshed this scan (rank <= my_rank) carry this scan's value, the rest last scan's
IMD void smooth_seen(const MeshDev& m, int vtx, int my_rank, double* out) {
    const int w = m.v_voxel[vtx];
    const bool fresh = (m.vx_rank_seq[w] == m.seq) && (m.vx_rank[w] <= my_rank);
    const double* src = (fresh ? m.v_smooth_new : m.v_smooth) + (size_t)vtx * 3;
    out[0] = src[0]; out[1] = src[1]; out[2] = src[2];
}
// correct_triangle_index (mesh_rec_geometry.cpp:399-433): m_index_flip.  A, B, C = smoothed positions of the (id-sorted) vertices
IMD int flip_of(const double* A, const double* B, const double* C, const double* cam, const double* short_axis) {
    const double ab[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, ac[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
    const double tc[3] = {cam[0] - A[0], cam[1] - A[1], cam[2] - A[2]};
    double nrm[3];
    cross3(ab, ac, nrm);
    const double nn = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    if (nn != 0) { nrm[0] /= nn; nrm[1] /= nn; nrm[2] /= nn; }
    else { nrm[0] = 0; nrm[1] = 0; nrm[2] = 1; }
    double sa[3] = {short_axis[0], short_axis[1], short_axis[2]};
    if (sa[0] * tc[0] + sa[1] * tc[1] + sa[2] * tc[2] < 0) { sa[0] *= -1; sa[1] *= -1; sa[2] *= -1; }
    return (sa[0] * nrm[0] + sa[1] * nrm[1] + sa[2] * nrm[2] < 0) ? 0 : 1;
}

IMD unsigned long long tri_hash3(int a, int b, int c) {
    return hash64(((unsigned long long)(unsigned int)a * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)(unsigned int)b << 21) ^ ((unsigned long long)(unsigned int)c << 42) ^ (unsigned long long)(unsigned int)c);
}
// m_triangle_hash: sorted triplet -> triangle index (created on first use; entries persist after erase)
IMD int tri_find_or_insert(const MeshDev& m, int a, int b, int c, int* spare) {
    unsigned long long h = tri_hash3(a, b, c) & m.th_mask;
    for (int probe = 0; probe < 16384; probe++) {
        int s = ld_agent(&m.th_slots[h]);
        if (s < 0) {
            if (*spare < 0) {
                const int t = atomicAdd(&m.pc[PC_TRIS], 1);
                if (t >= m.cap_tris) { m.sc[SC_OVERFLOW] = 10; return -1; }
                *spare = t;
            }
            const int t = *spare;
            st_agent(&m.t_v[(size_t)t * 3 + 0], a); st_agent(&m.t_v[(size_t)t * 3 + 1], b); st_agent(&m.t_v[(size_t)t * 3 + 2], c);
            m.t_word[t] = 0; m.t_live[t] = 0; m.t_rem_seq[t] = 0; m.t_flip[t] = 0;
            __threadfence();
            const int prev = atomicCAS(&m.th_slots[h], -1, t);
            if (prev == -1) { *spare = -1; return t; }
            s = prev;
        }
        if (ld_agent(&m.t_v[(size_t)s * 3 + 0]) == a && ld_agent(&m.t_v[(size_t)s * 3 + 1]) == b && ld_agent(&m.t_v[(size_t)s * 3 + 2]) == c) return s;
        h = (h + 1) & m.th_mask;
    }
    m.sc[SC_OVERFLOW] = 11;
    return -1;
}

#define DBG_T(k) do { if (m.dbg) { const unsigned long long _t = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&m.dbg[k], _t - tprev); tprev = _t; } } while (0)
// The general per-voxel triangulation (any neighbourhood size up to CAP): triangle table in LDS.  One wavefront; r = rank of the active voxel.
// GLOBAL = false: the per-voxel tables live in LDS (~31 KB at CAP = 256).  GLOBAL = true: in this block's slice of MeshDev::dv_scratch (the tables of
// CAP = MV_REL_CAP are ~120 KB: as static LDS they made every block of the launch claim a whole CU, and the launch -- which has nothing to do in
// the shipped configurations -- waited tens of microseconds for empty CUs whenever the registration chain's kernels were resident).
#define MV_GEN_BLOCKS 32
#define MV_GEN_SCRATCH (128 * 1024)
template <int CAP, bool GLOBAL>
__device__ __forceinline__ void mesh_delaunay_voxel(const MeshDev& m, const MeshScanParams& sp, const int r, unsigned char* gs) {
    constexpr int TCAP = 2 * CAP + 8;
    constexpr int LC = GLOBAL ? 1 : CAP, LT = GLOBAL ? 1 : TCAP;
    __shared__ int ids_l[LC];
    __shared__ float pf_l[LC * 3];
    __shared__ double xy_l[LC * 2];
    __shared__ unsigned long long keys_l[LC];
    __shared__ __attribute__((aligned(16))) unsigned short tri_l[LT * 4];
    __shared__ unsigned short cav[DT_CAV_CAP];
    __shared__ unsigned short ea[DT_CAV_CAP * 3], eb[DT_CAV_CAP * 3];
    __shared__ unsigned char bdf[DT_CAV_CAP * 3];     // boundary flag of every directed edge of the cavity
    __shared__ unsigned char cmem[DT_CAV_CAP];        // degenerate input: membership in the connected part of the cavity
    __shared__ unsigned int fresh_l[LT];
    __shared__ unsigned char fhit_l[LT];
    __shared__ double sm_l[LC * 3];   // smoothed positions as this voxel's turn in the sequential loop would see them
    __shared__ int s_cnt[2];
    __shared__ int old_t_l[2 * LC], old_v1_l[2 * LC], old_v2_l[2 * LC];   // live triangles whose smallest vertex is in the neighbourhood
    __shared__ unsigned short old_i_l[2 * LC];
    constexpr size_t O_XY = 0, O_SM = O_XY + 16ull * CAP, O_KEYS = O_SM + 24ull * CAP, O_TRI = O_KEYS + 8ull * CAP, O_IDS = O_TRI + 8ull * TCAP, O_PF = O_IDS + 4ull * CAP,
                     O_FRESH = O_PF + 12ull * CAP, O_OT = O_FRESH + 4ull * TCAP, O_OV1 = O_OT + 8ull * CAP, O_OV2 = O_OV1 + 8ull * CAP, O_OI = O_OV2 + 8ull * CAP,
                     O_FHIT = O_OI + 4ull * CAP, O_END = O_FHIT + TCAP;
    static_assert(!GLOBAL || O_END <= MV_GEN_SCRATCH, "scratch slice too small");
    int* const ids = GLOBAL ? (int*)(gs + O_IDS) : ids_l;
    float* const pf = GLOBAL ? (float*)(gs + O_PF) : pf_l;
    double* const xy = GLOBAL ? (double*)(gs + O_XY) : xy_l;
    unsigned long long* const keys = GLOBAL ? (unsigned long long*)(gs + O_KEYS) : keys_l;
    unsigned short* const tri = GLOBAL ? (unsigned short*)(gs + O_TRI) : tri_l;
    unsigned int* const fresh = GLOBAL ? (unsigned int*)(gs + O_FRESH) : fresh_l;
    unsigned char* const fhit = GLOBAL ? (unsigned char*)(gs + O_FHIT) : fhit_l;
    double* const sm = GLOBAL ? (double*)(gs + O_SM) : sm_l;
    int* const old_t = GLOBAL ? (int*)(gs + O_OT) : old_t_l;
    int* const old_v1 = GLOBAL ? (int*)(gs + O_OV1) : old_v1_l;
    int* const old_v2 = GLOBAL ? (int*)(gs + O_OV2) : old_v2_l;
    unsigned short* const old_i = GLOBAL ? (unsigned short*)(gs + O_OI) : old_i_l;

    const int lane = threadIdx.x;
    const int n = m.rel_n[r];
    unsigned long long tprev = m.dbg ? __builtin_readcyclecounter() : 0;
    const unsigned long long tvox0 = tprev;
    const int vi = m.act_vox_s[r];
    for (int i = lane; i < n; i += 64) {
        const int id = m.rel_ids[(size_t)r * MV_REL_CAP + i];
        ids[i] = id;
        pf[i * 3 + 0] = m.v_pos[(size_t)id * 3 + 0]; pf[i * 3 + 1] = m.v_pos[(size_t)id * 3 + 1]; pf[i * 3 + 2] = m.v_pos[(size_t)id * 3 + 2];
        smooth_seen(m, id, r, &sm[i * 3]);
    }
    if (lane == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    DBG_T(0);
    int nf = 0;
    if (n >= 3) {
        // ---- centre and covariance: sequential sums in vertex order (bit-identical to the CPU path), one lane per component
        double acc = 0;
        if (lane < 3) { for (int i = 0; i < n; i++) acc += (double)pf[i * 3 + lane]; acc /= (double)n; }
        const double c[3] = {__shfl(acc, 0, 64), __shfl(acc, 1, 64), __shfl(acc, 2, 64)};
        double cv = 0;
        if (lane < 6) {
            const int a = lane < 3 ? 0 : (lane < 5 ? 1 : 2), b = lane < 3 ? lane : (lane < 5 ? lane - 2 : 2);
            for (int i = 0; i < n; i++) cv += ((double)pf[i * 3 + a] - c[a]) * ((double)pf[i * 3 + b] - c[b]);
            cv /= (double)n;
        }
        const double c00 = __shfl(cv, 0, 64), c01 = __shfl(cv, 1, 64), c02 = __shfl(cv, 2, 64), c11 = __shfl(cv, 3, 64), c12 = __shfl(cv, 4, 64), c22 = __shfl(cv, 5, 64);
        const double cov[9] = {c00, c01, c02, c01, c11, c12, c02, c12, c22};
        double ev[3], U[9];
        sym3_eigen_jacobi(cov, ev, U);  // SelfAdjointEigenSolver::compute, mesh_rec_geometry.cpp:199-200 (eigenvalues ascending)
        int o0 = 0, o1 = 1, o2 = 2;     // stable ascending order of three
        if (ev[o1] < ev[o0]) { const int t = o0; o0 = o1; o1 = t; }
        if (ev[o2] < ev[o1]) { const int t = o1; o1 = o2; o2 = t; if (ev[o1] < ev[o0]) { const int t2 = o0; o0 = o1; o1 = t2; } }
        double sh[3] = {U[0 * 3 + o0], U[1 * 3 + o0], U[2 * 3 + o0]};
        double mid[3] = {U[0 * 3 + o1], U[1 * 3 + o1], U[2 * 3 + o1]};
        {
            const double X0[3] = {(double)pf[0] - c[0], (double)pf[1] - c[1], (double)pf[2] - c[2]};
            const double X1[3] = {(double)pf[3] - c[0], (double)pf[4] - c[1], (double)pf[5] - c[2]};
            if (X0[0] * sh[0] + X0[1] * sh[1] + X0[2] * sh[2] < 0) { sh[0] *= -1; sh[1] *= -1; sh[2] *= -1; }          // :204-207
            if (X1[0] * mid[0] + X1[1] * mid[1] + X1[2] * mid[2] < 0) { mid[0] *= -1; mid[1] *= -1; mid[2] *= -1; }    // :208-211
        }
        double lg[3];
        cross3(sh, mid, lg);
        if (lane == 0) { m.vx_short_axis[(size_t)vi * 3 + 0] = sh[0]; m.vx_short_axis[(size_t)vi * 3 + 1] = sh[1]; m.vx_short_axis[(size_t)vi * 3 + 2] = sh[2]; }
        // ---- 2-D projection
        double mn0 = 1e300, mn1 = 1e300, mx0 = -1e300, mx1 = -1e300;
        for (int i = lane; i < n; i += 64) {
            const double X[3] = {(double)pf[i * 3 + 0] - c[0], (double)pf[i * 3 + 1] - c[1], (double)pf[i * 3 + 2] - c[2]};
            const double u = X[0] * lg[0] + X[1] * lg[1] + X[2] * lg[2];
            const double v = X[0] * mid[0] + X[1] * mid[1] + X[2] * mid[2];
            xy[2 * i] = u; xy[2 * i + 1] = v;
            mn0 = fmin(mn0, u); mx0 = fmax(mx0, u); mn1 = fmin(mn1, v); mx1 = fmax(mx1, v);
        }
        mn0 = wave_min_d(mn0); mn1 = wave_min_d(mn1); mx0 = wave_max_d(mx0); mx1 = wave_max_d(mx1);
        const double ext = fmax(mx0 - mn0, mx1 - mn1);
        // ---- insertion order: Morton code over the bounding box, ties by index
        const int np2 = next_pow2_i(n);
        for (int i = lane; i < np2; i += 64) {
            unsigned long long key = ~0ull;
            if (i < n) {
                const double f0 = ext > 0 ? (xy[2 * i] - mn0) / ext : 0, f1 = ext > 0 ? (xy[2 * i + 1] - mn1) / ext : 0;
                const unsigned int q0 = (unsigned int)fmin(65535.0, fmax(0.0, f0 * 65535.0)), q1 = (unsigned int)fmin(65535.0, fmax(0.0, f1 * 65535.0));
                unsigned int code = 0;
#pragma unroll
                for (int b = 0; b < 16; b++) code |= (((q0 >> b) & 1u) << (2 * b)) | (((q1 >> b) & 1u) << (2 * b + 1));
                key = ((unsigned long long)code << 16) | (unsigned long long)i;
            }
            keys[i] = key;
        }
        __syncthreads();
        DBG_T(1);
        lds_bitonic_sort<unsigned long long, 64>(keys, np2, lane);
        DBG_T(2);
        // ---- first non-degenerate triangle
        const int i0 = (int)(keys[0] & 0xFFFF);
        int i1 = -1, i2 = -1;
        for (int k = 1; k < n && i1 < 0; k++) { const int cc = (int)(keys[k] & 0xFFFF); if (xy[2 * cc] != xy[2 * i0] || xy[2 * cc + 1] != xy[2 * i0 + 1]) i1 = cc; }
        if (i1 >= 0)
            for (int k = 1; k < n && i2 < 0; k++) { const int cc = (int)(keys[k] & 0xFFFF); if (cc != i1 && orient2d(xy + 2 * i0, xy + 2 * i1, xy + 2 * cc) != 0) i2 = cc; }
        int nt = 0;
        if (i2 >= 0) {
            if (orient2d(xy + 2 * i0, xy + 2 * i1, xy + 2 * i2) < 0) { const int t = i1; i1 = i2; i2 = t; }
            if (lane == 0) {
                const unsigned short init[16] = {(unsigned short)i0, (unsigned short)i1, (unsigned short)i2, 0, (unsigned short)i2, (unsigned short)i1, DT_INF, 0,
                                                 (unsigned short)i0, (unsigned short)i2, DT_INF, 0, (unsigned short)i1, (unsigned short)i0, DT_INF, 0};
                for (int k = 0; k < 16; k++) tri[k] = init[k];
            }
            nt = 4;
            __syncthreads();
            bool fail = false;
            int n_skip = 0;

            const unsigned long long* tri64 = (const unsigned long long*)tri;
            int pin = (int)(keys[0] & 0xFFFF);
            double pnx = xy[2 * pin], pny = xy[2 * pin + 1];
            for (int oi = 0; oi < n && !fail; oi++) {
                const int pi = pin;
                const double p[2] = {pnx, pny};
                if (oi + 1 < n) { pin = (int)(keys[oi + 1] & 0xFFFF); pnx = xy[2 * pin]; pny = xy[2 * pin + 1]; }   // the next point's fetch overlaps this step
                if (pi == i0 || pi == i1 || pi == i2) continue;
                // cavity: every triangle whose circumdisk contains p; each member immediately publishes its three directed edges
                int ncav = 0;
                for (int base = 0; base < nt; base += 64) {
                    const int t = base + lane;
                    bool in = false;
                    unsigned v0 = 0, v1 = 0, v2 = 0;
                    if (t < nt) {
                        const unsigned long long tv = tri64[t];
                        v0 = (unsigned)(tv & 0xFFFF); v1 = (unsigned)((tv >> 16) & 0xFFFF); v2 = (unsigned)((tv >> 32) & 0xFFFF);
                        in = dt_in_disk(xy, v0, v1, v2, p);
                    }
                    const unsigned long long mask = __ballot(in);
                    if (in) {
                        const int pos = ncav + __popcll(mask & ((1ull << lane) - 1ull));
                        if (pos < DT_CAV_CAP) {
                            cav[pos] = (unsigned short)t;
                            ea[pos * 3 + 0] = (unsigned short)v1; eb[pos * 3 + 0] = (unsigned short)v2;   // edge k: v[(k+1)%3] -> v[(k+2)%3]
                            ea[pos * 3 + 1] = (unsigned short)v2; eb[pos * 3 + 1] = (unsigned short)v0;
                            ea[pos * 3 + 2] = (unsigned short)v0; eb[pos * 3 + 2] = (unsigned short)v1;
                        }
                    }
                    ncav += __popcll(mask);
                }
                if (ncav == 0) { n_skip++; continue; }  // duplicate / on every circle: not inserted
                if (ncav > DT_CAV_CAP) { fail = true; break; }
                __syncthreads();
                int ne = 3 * ncav;
                // boundary edges (twin not in the cavity): flags first, the fan afterwards
                auto mark_boundary = [&]() -> int {
                    int nbq = 0;
                    for (int base = 0; base < ne; base += 64) {
                        const int e = base + lane;
                        bool bd = false;
                        unsigned short a = 0, b = 0;
                        if (e < ne) { a = ea[e]; b = eb[e]; bd = true; }
                        if (ne <= 64) {  // one edge per lane: the twin search runs on registers (v_readlane), no LDS round trips
                            const unsigned int key = ((unsigned int)a << 16) | (unsigned int)b, twin = ((unsigned int)b << 16) | (unsigned int)a;
                            for (int f = 0; f < ne; f++) if ((unsigned int)__builtin_amdgcn_readlane((int)key, f) == twin) bd = false;
                        } else if (e < ne) {
                            for (int f = 0; f < ne; f++) if (ea[f] == b && eb[f] == a) { bd = false; break; }
                        }
                        if (e < ne) bdf[e] = bd ? 1 : 0;
                        nbq += __popcll(__ballot(bd));
                    }
                    return nbq;
                };
                int nb = mark_boundary();
                if (nb != ncav + 2) {
                    // DEGENERATE INPUT (cocircular / collinear points: the in-circle determinants are rounding noise around zero).  The triangles that
                    // tested in-disk are not ONE disk (Euler: a disk of c triangles has c + 2 boundary edges, ghosts included) -- a far-away triangle
                    // tested positive, or one next to p negative.  The checker's rule (oracle/orc_delaunay.hpp, header), literally: the cavity is the
                    // part of the set that is edge-connected to the START SET = its finite triangles that contain p (all three orientations >= 0);
                    // none (p outside the hull) -> its ghost that sees p best (largest orient2d(a, b, p), ties: smallest sorted vertex triple); no
                    // ghost either -> its triangle with the smallest sorted vertex triple.  Never reached on points in general position.
                    __syncthreads();
                    bool any = false;
                    for (int base = 0; base < ncav; base += 64) {   // start set: the finite triangles of the set that contain p (closed)
                        const int cidx = base + lane;
                        bool cont = false;
                        if (cidx < ncav) {
                            const unsigned v0 = ea[cidx * 3 + 2], v1 = ea[cidx * 3 + 0], v2 = ea[cidx * 3 + 1];
                            cont = v0 != DT_INF && v1 != DT_INF && v2 != DT_INF &&
                                   orient2d(xy + 2 * v0, xy + 2 * v1, p) >= 0 && orient2d(xy + 2 * v1, xy + 2 * v2, p) >= 0 && orient2d(xy + 2 * v2, xy + 2 * v0, p) >= 0;
                            cmem[cidx] = cont ? 1 : 0;
                        }
                        any = any || (__ballot(cont) != 0ull);
                    }
                    if (!any) {   // p outside the hull: the ghost that sees it best (largest orient2d(a, b, p); ties: smallest sorted vertex triple); no ghost: smallest triple
                        double best_o = -1.0;                 // (a ghost of the set has o >= 0)
                        unsigned long long best_k = ~0ull;    // sorted triple << 16 | cavity index
                        bool have_ghost = false;
                        for (int pass = 0; pass < 2 && !have_ghost; pass++) {   // pass 0: ghosts by (o desc, key asc); pass 1 (no ghost at all): every triangle by key
                            for (int cidx = lane; cidx < ncav; cidx += 64) {
                                unsigned v0 = ea[cidx * 3 + 2], v1 = ea[cidx * 3 + 0], v2 = ea[cidx * 3 + 1];
                                const int gi = v0 == DT_INF ? 0 : (v1 == DT_INF ? 1 : (v2 == DT_INF ? 2 : -1));
                                if (pass == 0 && gi < 0) continue;
                                double o = 0.0;
                                if (pass == 0) { const unsigned a = gi == 0 ? v1 : (gi == 1 ? v2 : v0), b = gi == 0 ? v2 : (gi == 1 ? v0 : v1); o = orient2d(xy + 2 * a, xy + 2 * b, p); }
                                if (v0 > v1) { const unsigned x = v0; v0 = v1; v1 = x; }
                                if (v1 > v2) { const unsigned x = v1; v1 = v2; v2 = x; }
                                if (v0 > v1) { const unsigned x = v0; v0 = v1; v1 = x; }
                                const unsigned long long key = ((((unsigned long long)v0 << 32) | ((unsigned long long)v1 << 16) | (unsigned long long)v2) << 16) | (unsigned long long)cidx;
                                if (o > best_o || (o == best_o && key < best_k)) { best_o = o; best_k = key; }
                            }
                            for (int off = 32; off > 0; off >>= 1) {
                                const double oo = __shfl_xor(best_o, off, 64); const unsigned long long ok = __shfl_xor(best_k, off, 64);
                                if (oo > best_o || (oo == best_o && ok < best_k)) { best_o = oo; best_k = ok; }
                            }
                            have_ghost = best_k != ~0ull;
                            if (!have_ghost) best_o = -1.0;
                        }
                        if (lane == 0) cmem[(int)(best_k & 0xFFFFull)] = 1;
                    }
                    __syncthreads();
                    for (bool changed = true; changed;) {
                        bool ch = false;
                        for (int base = 0; base < ncav; base += 64) {
                            const int cidx = base + lane;
                            bool join = false;
                            if (cidx < ncav && !cmem[cidx])
                                for (int f = 0; f < ncav && !join; f++) {
                                    if (!cmem[f]) continue;
                                    for (int i = 0; i < 3 && !join; i++)
                                        for (int j = 0; j < 3; j++)
                                            if (ea[cidx * 3 + i] == eb[f * 3 + j] && eb[cidx * 3 + i] == ea[f * 3 + j]) { join = true; break; }
                                }
                            ch = ch || (__ballot(join) != 0ull);
                            __syncthreads();
                            if (join) cmem[cidx] = 1;
                            __syncthreads();
                        }
                        changed = ch;
                    }
                    if (lane == 0) {   // compact the member triangles to the front (cavity order kept)
                        int j = 0;
                        for (int cidx = 0; cidx < ncav; cidx++)
                            if (cmem[cidx]) {
                                cav[j] = cav[cidx];
                                for (int k = 0; k < 3; k++) { ea[j * 3 + k] = ea[cidx * 3 + k]; eb[j * 3 + k] = eb[cidx * 3 + k]; }
                                j++;
                            }
                        s_cnt[1] = j;
                    }
                    __syncthreads();
                    ncav = s_cnt[1];
                    __syncthreads();
                    if (lane == 0) s_cnt[1] = 0;
                    ne = 3 * ncav;
                    nb = mark_boundary();
                }
                __syncthreads();
                // the fan of new triangles (a, b, p) over the boundary edges, reusing the cavity slots first
                {
                    int nbw = 0;
                    for (int base = 0; base < ne; base += 64) {
                        const int e = base + lane;
                        const bool bd = e < ne && bdf[e];
                        const unsigned long long mask = __ballot(bd);
                        if (bd) {
                            const int j = nbw + __popcll(mask & ((1ull << lane) - 1ull));
                            const int slot = j < ncav ? (int)cav[j] : nt + (j - ncav);
                            if (slot < TCAP) { tri[slot * 4 + 0] = ea[e]; tri[slot * 4 + 1] = eb[e]; tri[slot * 4 + 2] = (unsigned short)pi; }
                        }
                        nbw += __popcll(mask);
                    }
                }
                if (nb >= ncav) { nt += nb - ncav; if (nt > TCAP) { fail = true; break; } }
                else {  // inconsistent predicates left fewer new triangles than holes: close the holes from the back
                    __syncthreads();
                    if (lane == 0)
                        for (int j = ncav - 1; j >= nb; j--) {
                            const int slot = cav[j], last = nt - 1;
                            if (slot != last) { tri[slot * 4 + 0] = tri[last * 4 + 0]; tri[slot * 4 + 1] = tri[last * 4 + 1]; tri[slot * 4 + 2] = tri[last * 4 + 2]; }
                            nt--;
                        }
                    nt = __shfl(nt, 0, 64);
                }
                __syncthreads();
            }
            if (fail) { if (lane == 0) m.sc[SC_OVERFLOW] = 12; nt = 0; }
            else if (n_skip && threadIdx.x == 0) atomicAdd(&m.sc[SC_DEGEN], n_skip);
        }
        DBG_T(3);
        // ---- finite faces that pass the skinny-face filter (is_face_is_ok: every interior angle * 57.3 <= 150) as sorted local triplets
        for (int base = 0; base < nt; base += 64) {
            const int t = base + lane;
            bool ok = false;
            unsigned int packed = 0;
            if (t < nt) {
                const unsigned a = tri[t * 4 + 0], b = tri[t * 4 + 1], cc = tri[t * 4 + 2];
                if (a != DT_INF && b != DT_INF && cc != DT_INF) {
                    const double* A = xy + 2 * a; const double* B = xy + 2 * b; const double* C = xy + 2 * cc;
                    auto angle = [](const double* P, const double* Q, const double* R) {  // compute_angle at P
                        const double abx = Q[0] - P[0], aby = Q[1] - P[1], acx = R[0] - P[0], acy = R[1] - P[1];
                        return acos((abx * acx + aby * acy) / (sqrt(abx * abx + aby * aby) * sqrt(acx * acx + acy * acy))) * 57.3;
                    };
                    ok = !(angle(A, B, C) > 150) && !(angle(B, A, C) > 150) && !(angle(C, A, B) > 150);
                    unsigned l0 = a, l1 = b, l2 = cc;  // local index order == vertex id order (ids ascending)
                    if (l0 > l1) { const unsigned x = l0; l0 = l1; l1 = x; }
                    if (l1 > l2) { const unsigned x = l1; l1 = l2; l2 = x; }
                    if (l0 > l1) { const unsigned x = l0; l0 = l1; l1 = x; }
                    packed = (l0 << 20) | (l1 << 10) | l2;
                }
            }
            const unsigned long long mask = __ballot(ok);
            if (ok) fresh[nf + __popcll(mask & ((1ull << lane) - 1ull))] = packed;
            nf += __popcll(mask);
        }
        const int nfp = next_pow2_i(max(nf, 1));
        for (int k = nf + lane; k < nfp; k += 64) fresh[k] = 0xFFFFFFFFu;
        for (int k = lane; k < nf; k += 64) fhit[k] = 0;
        __syncthreads();
        lds_bitonic_sort<unsigned int, 64>(fresh, nfp, lane);
        {   // the faces are a SET (triangle_compare keys them by sorted triplet, mesh_rec_geometry.cpp:137-172): on degenerate input the triangulation can
            // hold the same face twice -- keep one (in place: a chunk is read into registers before anything of it is overwritten, and the write
            // positions never run ahead of the read positions)
            int nu = 0;
            for (int base = 0; base < nf; base += 64) {
                const int k = base + lane;
                __syncthreads();
                const unsigned int v = k < nf ? fresh[k] : 0u;
                const bool keep = k < nf && (k == 0 || fresh[k - 1] != v);
                const unsigned long long mask = __ballot(keep);
                __syncthreads();
                if (keep) fresh[nu + __popcll(mask & ((1ull << lane) - 1ull))] = v;
                nu += __popcll(mask);
            }
            __syncthreads();
            for (int k = nu + lane; k < nf; k += 64) fresh[k] = 0xFFFFFFFFu;
            nf = nu;
        }
        DBG_T(4);
    }
    __syncthreads();
    // ---- old = live triangles with all three vertices in the neighbourhood (find_relative_triangulation_combination), via the
    //      min-vertex lists; old \ fresh -> remove, old & fresh -> existing (flip rewritten), fresh \ old -> add
    const double* sa = m.vx_short_axis + (size_t)vi * 3;
    const double short_axis[3] = {sa[0], sa[1], sa[2]};
    int* touched = m.vox_tris + (size_t)r * (2 * MV_REL_CAP);
    const unsigned long long wbase = ((unsigned long long)(unsigned int)m.seq << 32) | ((unsigned long long)(unsigned int)r << 1);
    // (1) gather: every lane walks the min-vertex list of its vertex and appends the live triangles to one LDS list ...
    for (int i = lane; i < n; i += 64) {
        const int id = ids[i];
        for (int ch = m.a_head[id]; ch >= 0;) {
            const int* cp = m.a_chunks + (size_t)ch * MV_ADJ_STRIDE;
            int e[MV_ADJ_STRIDE];
#pragma unroll
            for (int k = 0; k < MV_ADJ_STRIDE; k++) e[k] = cp[k];   // one 64-byte chunk: 5 x (triangle, v1, v2) + next
#pragma unroll
            for (int sl = 0; sl < MV_ADJ_SLOTS; sl++) {
                if (e[sl * 3] < 0) continue;
                const int pos = atomicAdd(&s_cnt[1], 1);
                if (pos < 2 * CAP) { old_t[pos] = e[sl * 3]; old_v1[pos] = e[sl * 3 + 1]; old_v2[pos] = e[sl * 3 + 2]; old_i[pos] = (unsigned short)i; }
            }
            ch = e[MV_ADJ_STRIDE - 1];
        }
    }
    __syncthreads();
    // (2) ... then the list is classified with all lanes busy and uniform control flow
    const int nold = s_cnt[1];
    if (nold > 2 * CAP) {
        // more live triangles around this voxel than the LDS list holds (space-filling clouds pile up triangles of many projection planes):
        // classify them straight from the vertex lists -- same decisions, divergent lanes
        for (int i = lane; i < n; i += 64) {
            const int id = ids[i];
            for (int ch = m.a_head[id]; ch >= 0; ch = m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + MV_ADJ_STRIDE - 1])
                for (int sl = 0; sl < MV_ADJ_SLOTS; sl++) {
                    const int t = m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + sl * 3];
                    if (t < 0) continue;
                    const int l1 = lds_bsearch_i32(ids, n, m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + sl * 3 + 1]);
                    const int l2 = lds_bsearch_i32(ids, n, m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + sl * 3 + 2]);
                    if (l1 < 0 || l2 < 0) continue;
                    const int pos = lds_bsearch_u32(fresh, nf, ((unsigned int)i << 20) | ((unsigned int)l1 << 10) | (unsigned int)l2);
                    if (pos >= 0) {
                        fhit[pos] = 1;
                        const int fl = flip_of(&sm[i * 3], &sm[l1 * 3], &sm[l2 * 3], sp.cam, short_axis);
                        atomicMax(&m.t_word[t], wbase | (unsigned long long)fl);
                        touched[atomicAdd(&s_cnt[0], 1)] = t;
                    } else if (atomicExch(&m.t_rem_seq[t], m.seq) != m.seq) {
                        list_push(m, m.list_rem, SC_REM, t);
                    }
                }
        }
    }
    for (int k = lane; k < (nold > 2 * CAP ? 0 : nold); k += 64) {
        const int t = old_t[k], i = old_i[k];
        const int l1 = lds_bsearch_i32(ids, n, old_v1[k]), l2 = lds_bsearch_i32(ids, n, old_v2[k]);
        if (l1 < 0 || l2 < 0) continue;   // not entirely inside this neighbourhood
        const int pos = lds_bsearch_u32(fresh, nf, ((unsigned int)i << 20) | ((unsigned int)l1 << 10) | (unsigned int)l2);
        if (pos >= 0) {
            fhit[pos] = 1;
            const int fl = flip_of(&sm[i * 3], &sm[l1 * 3], &sm[l2 * 3], sp.cam, short_axis);
            atomicMax(&m.t_word[t], wbase | (unsigned long long)fl);
            touched[atomicAdd(&s_cnt[0], 1)] = t;
        } else if (atomicExch(&m.t_rem_seq[t], m.seq) != m.seq) {
            list_push(m, m.list_rem, SC_REM, t);
        }
    }
    __syncthreads();
    DBG_T(5);
    int spare = -1;
    for (int k = lane; k < nf; k += 64) {
        if (fhit[k]) continue;
        const unsigned int pk = fresh[k];
        const int la = (int)(pk >> 20), lb = (int)((pk >> 10) & 1023), lc = (int)(pk & 1023);
        const int a = ids[la], b = ids[lb], c = ids[lc];
        const int t = tri_find_or_insert(m, a, b, c, &spare);
        if (t < 0) continue;
        const int fl = flip_of(&sm[la * 3], &sm[lb * 3], &sm[lc * 3], sp.cam, short_axis);
        atomicMax(&m.t_word[t], wbase | (unsigned long long)fl);
        touched[atomicAdd(&s_cnt[0], 1)] = (int)((unsigned int)t | TRI_ADD_BIT);
    }
    __syncthreads();
    if (lane == 0) { m.vox_ntris[r] = s_cnt[0]; atomicAdd(&m.sc[SC_TV], nf); }
    __syncthreads();
    DBG_T(6);
    if (m.dbg && lane == 0) atomicMax(&m.dbg[14], ((__builtin_readcyclecounter() - tvox0) << 16) | (unsigned long long)n);
    __syncthreads();
}
// What the register fast path (mesh_delaunay64_kernel, launched before) does not take, ONE small launch (with the shipped configurations it has
// nothing to do, so what counts is how quickly its blocks are placed and gone): neighbourhoods of 65..256 vertices and the voxels the fast path
// handed over from LDS tables, neighbourhoods above 256 vertices (space-filling clouds) from tables in global scratch.
__global__ __launch_bounds__(64) void mesh_delaunay_general_kernel(MeshDev m_in) {
    MESH_MARK(m_in, 6);
    MESH_DYN(m_in);
    const int n_active = min(m.sc[SC_ACTIVE], m.cap_active);
    for (int r = (int)blockIdx.x; r < n_active; r += (int)gridDim.x) {
        const int n = m.rel_n[r];
        if (n > 256) mesh_delaunay_voxel<MV_REL_CAP, true>(m, sp, r, m.dv_scratch + (size_t)blockIdx.x * MV_GEN_SCRATCH);
        else if (n > 64 || (n > 0 && m.vox_ntris[r] < 0)) mesh_delaunay_voxel<256, false>(m, sp, r, nullptr);
    }
}

#include "mesh_delaunay64.inc"

// cross-voxel resolution: the voxel with the highest rank that touched a triangle owns its flip (later voxel wins, as the
// sequential loop); it also queues the triangle for insertion / reports a changed flip.  Then this scan's smoothed positions commit.
// Triangle_manager::remove_triangle_list (triangle.hpp:212-221): drop from the live set and from its smallest vertex's list.  Runs inside the
// finalize launch (the two touch disjoint data: live flags + adjacency here, flip words + result lists there).
IMD void mesh_commit_rem_slice(const MeshDev& m, const int32_t* __restrict__ tris, int first, int stride) {
    const int n = min(m.sc[SC_REM], m.cap_list);
    for (int i = first; i < n; i += stride) {
        const int t = tris[i];
        m.t_live[t] = 0;
        const int v0 = m.t_v[(size_t)t * 3 + 0];
        bool done = false;
        for (int ch = m.a_head[v0]; ch >= 0 && !done; ch = m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + MV_ADJ_STRIDE - 1])
            for (int s = 0; s < MV_ADJ_SLOTS; s++)
                if (m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + s * 3] == t) { m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + s * 3] = -1; done = true; break; }
    }
}
__global__ __launch_bounds__(64) void mesh_finalize_kernel(MeshDev m_in) {
    MESH_MARK(m_in, 7);
    __builtin_amdgcn_s_setprio(MESH_B_PRIO);   // phase B is the mesher's longest chain: issue ahead of the map update's waves (1), behind the registration's (3)
    MESH_DYN(m_in);
    mesh_commit_rem_slice(m, m.list_rem, blockIdx.x * 64 + threadIdx.x, gridDim.x * 64);   // Triangle_manager::remove_triangle_list rides along (independent data)
    const int lane = threadIdx.x;
    const int n_active = min(m.sc[SC_ACTIVE], m.cap_active);
    for (int r = blockIdx.x; r < n_active; r += gridDim.x) {
    const int vi = m.act_vox_s[r];
    const int nt = m.vox_ntris[r];
    const int* touched = m.vox_tris + (size_t)r * (2 * MV_REL_CAP);
    for (int k = lane; k < nt; k += 64) {
        const unsigned int e = (unsigned int)touched[k];
        const int t = (int)(e & 0x7FFFFFFFu);
        const unsigned long long w = m.t_word[t];
        if ((unsigned int)(w >> 32) != (unsigned int)m.seq || (int)((w >> 1) & 0x7FFFFFFFull) != r) continue;
        const int8_t fl = (int8_t)(w & 1ull);
        if (e & TRI_ADD_BIT) { m.t_flip[t] = fl; list_push(m, m.list_add, SC_ADD, t); }
        else if (m.t_flip[t] != fl) { m.t_flip[t] = fl; list_push(m, m.list_upd, SC_UPD, t); }
    }
    if (m.shard_world > 1 && mesh_owner(m, m.vx_key[vi]) != m.shard_rank) continue;   // another rank's voxel: what this rank needs of it arrived by all-gather (below)
    const int np = m.rel_nq[r];   // the vertices the voxel held when phase A searched it (phase A of the next scan may be appending already)
    for (int k = lane; k < np; k += 64) {
        const int id = m.vx_pts[(size_t)vi * MV_VOX_CAP + k];
        m.v_smooth[(size_t)id * 3 + 0] = m.v_smooth_new[(size_t)id * 3 + 0];
        m.v_smooth[(size_t)id * 3 + 1] = m.v_smooth_new[(size_t)id * 3 + 1];
        m.v_smooth[(size_t)id * 3 + 2] = m.v_smooth_new[(size_t)id * 3 + 2];
        list_push(m, m.list_smooth, SC_SMOOTH, id);
    }
    }
    if (m.shard_world > 1) {   // smoothed positions received from the other ranks (the band this rank's voxels can reach): committed, not reported
        const int nrx = min(m.sc[SC_SMOOTH_RX], m.cap_list);
        for (int k = blockIdx.x * 64 + lane; k < nrx; k += gridDim.x * 64) {
            const int id = m.list_smooth_rx[k];
            m.v_smooth[(size_t)id * 3 + 0] = m.v_smooth_new[(size_t)id * 3 + 0];
            m.v_smooth[(size_t)id * 3 + 1] = m.v_smooth_new[(size_t)id * 3 + 1];
            m.v_smooth[(size_t)id * 3 + 2] = m.v_smooth_new[(size_t)id * 3 + 2];
        }
    }
}

// =====================================================================================================================
// sharded mesher: exchange records (see immesh_set_allgather)
// =====================================================================================================================
// Only the BOUNDARY BAND travels: what a voxel computes can matter to the voxels within MV_REACH of it, so a rank sends
//   * the smoothed positions of a voxel it searched only when a brick of another rank lies within MV_REACH of the voxel,
//   * a triangle mark only when a brick of another rank lies within MV_REACH of the voxel of one of the triangle's vertices
// and a receiver keeps only what lies within MV_REACH of one of ITS bricks -- its triangle store and smoothed positions cover its bricks + halo.
IMD bool vertex_near_foreign(const MeshDev& m, int vtx) {
    long lo[3], hi[3];
    voxel_box(m.vx_key[m.v_voxel[vtx]], MV_REACH, lo, hi);
    return box_foreign(m, lo, hi, m.shard_rank);
}
IMD bool vertex_near_me(const MeshDev& m, int vtx) {
    long lo[3], hi[3];
    voxel_box(m.vx_key[m.v_voxel[vtx]], MV_REACH, lo, hi);
    return box_has(m, lo, hi, m.shard_rank);
}
__global__ __launch_bounds__(64) void mesh_pack_smooth_kernel(MeshDev m_in, MeshSmRec* __restrict__ out, int32_t* __restrict__ count, int cap_rec) {
    MESH_DYN(m_in);
    const int lane = threadIdx.x;
    const int n_active = min(m.sc[SC_ACTIVE], m.cap_active);
    for (int r = blockIdx.x; r < n_active; r += gridDim.x) {
        const int vi = m.act_vox_s[r];
        const unsigned long long vkey = m.vx_key[vi];
        if (mesh_owner(m, vkey) != m.shard_rank) continue;
        long lo[3], hi[3];
        voxel_box(vkey, MV_REACH, lo, hi);
        if (!box_foreign(m, lo, hi, m.shard_rank)) continue;   // interior voxel: nobody else reads its vertices' smoothed positions
        const int nq = m.rel_nq[r];
        int base = 0;
        if (lane == 0) base = atomicAdd(count, nq);
        base = __shfl(base, 0, 64);
        for (int k = lane; k < nq; k += 64) {
            if (base + k >= cap_rec) break;   // (the header's count tells everybody)
            const int id = m.vx_pts[(size_t)vi * MV_VOX_CAP + k];
            MeshSmRec rec;
            rec.id = id; rec.pad = 0;
            rec.x = m.v_smooth_new[(size_t)id * 3 + 0]; rec.y = m.v_smooth_new[(size_t)id * 3 + 1]; rec.z = m.v_smooth_new[(size_t)id * 3 + 2];
            out[base + k] = rec;
        }
    }
}
__global__ __launch_bounds__(256) void mesh_unpack_smooth_kernel(MeshDev m, const char* __restrict__ gathered, size_t capb, int cap_rec) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gstride = gridDim.x * blockDim.x;
    if (gtid == 0) atomicAdd(&m.sc[SC_XBYTES], xblock_count(m, gathered, capb, m.shard_rank, cap_rec) * (int)sizeof(MeshSmRec));
    for (int r = 0; r < m.shard_world; r++) {
        if (r == m.shard_rank) continue;
        const int n = xblock_count(m, gathered, capb, r, cap_rec);
        const MeshSmRec* in = (const MeshSmRec*)(gathered + (size_t)r * capb + 16);
        for (int i = gtid; i < n; i += gstride) {
            const MeshSmRec rec = in[i];
            if (!vertex_near_me(m, rec.id)) continue;
            m.v_smooth_new[(size_t)rec.id * 3 + 0] = rec.x; m.v_smooth_new[(size_t)rec.id * 3 + 1] = rec.y; m.v_smooth_new[(size_t)rec.id * 3 + 2] = rec.z;
            list_push(m, m.list_smooth_rx, SC_SMOOTH_RX, rec.id);
        }
    }
}
// blocks [0, n_active): the touched lists of the voxels this rank triangulated; the blocks after them: its removal marks
__global__ __launch_bounds__(64) void mesh_pack_marks_kernel(MeshDev m_in, MeshMkRec* __restrict__ out, int32_t* __restrict__ count, int cap_rec) {
    MESH_DYN(m_in);
    const int lane = threadIdx.x;
    const int n_active = min(m.sc[SC_ACTIVE], m.cap_active);
    const int n_rem = min(m.sc[SC_REM], m.cap_list);
    const int n_rem_blocks = (n_rem + 63) / 64;
    for (int blk = blockIdx.x; blk < n_active + n_rem_blocks; blk += gridDim.x) {
        int r = -1, nt = 0;
        const int* touched = nullptr;
        if (blk < n_active) {
            r = blk;
            const int vi = m.act_vox_s[r];
            if (mesh_owner(m, m.vx_key[vi]) != m.shard_rank) continue;
            nt = m.vox_ntris[r];
            touched = m.vox_tris + (size_t)r * (2 * MV_REL_CAP);
        } else nt = min(64, n_rem - (blk - n_active) * 64);
        for (int k0 = 0; k0 < nt; k0 += 64) {
            const int k = k0 + lane;
            bool send = false;
            MeshMkRec rec;
            rec.a = rec.b = rec.c = 0; rec.rk = -1; rec.word = 0;
            if (k < nt) {
                int t;
                if (r >= 0) {
                    const unsigned int e = (unsigned int)touched[k];
                    t = (int)(e & 0x7FFFFFFFu);
                    rec.rk = (r << 1) | ((e & TRI_ADD_BIT) ? 1 : 0);
                    rec.word = m.t_word[t];   // this rank's maximum so far; the receivers max it with their own
                } else t = m.list_rem[(blk - n_active) * 64 + k];
                rec.a = m.t_v[(size_t)t * 3 + 0]; rec.b = m.t_v[(size_t)t * 3 + 1]; rec.c = m.t_v[(size_t)t * 3 + 2];
                send = vertex_near_foreign(m, rec.a) || vertex_near_foreign(m, rec.b) || vertex_near_foreign(m, rec.c);
            }
            const unsigned long long mask = __ballot(send);
            int base = 0;
            if (lane == 0 && mask) base = atomicAdd(count, (int)__popcll(mask));
            base = __shfl(base, 0, 64);
            if (send) {
                const int pos = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
                if (pos < cap_rec) out[pos] = rec;   // (the header's count tells everybody)
            }
        }
    }
}
__global__ __launch_bounds__(256) void mesh_unpack_marks_kernel(MeshDev m_in, const char* __restrict__ gathered, size_t capb, int cap_rec) {
    MESH_DYN(m_in);
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gstride = gridDim.x * blockDim.x;
    if (gtid == 0) atomicAdd(&m.sc[SC_XBYTES], xblock_count(m, gathered, capb, m.shard_rank, cap_rec) * (int)sizeof(MeshMkRec));
    for (int rk = 0; rk < m.shard_world; rk++) {
        if (rk == m.shard_rank) continue;
        const int n = xblock_count(m, gathered, capb, rk, cap_rec);
        const MeshMkRec* in = (const MeshMkRec*)(gathered + (size_t)rk * capb + 16);
        for (int i = gtid; i < n; i += gstride) {
            const MeshMkRec rec = in[i];
            if (!(vertex_near_me(m, rec.a) || vertex_near_me(m, rec.b) || vertex_near_me(m, rec.c))) continue;   // out of this rank's bricks + halo: never queried here
            int spare = -1;
            const int t = tri_find_or_insert(m, rec.a, rec.b, rec.c, &spare);
            if (t < 0) continue;
            if (rec.rk < 0) {   // removal mark of another rank's voxel: same once-per-scan rule as the local ones
                if (atomicExch(&m.t_rem_seq[t], m.seq) != m.seq) list_push(m, m.list_rem, SC_REM, t);
                continue;
            }
            atomicMax(&m.t_word[t], rec.word);
            const int r = rec.rk >> 1;
            const int pos = atomicAdd(&m.vox_ntris[r], 1);
            if (pos >= 2 * MV_REL_CAP) { m.sc[SC_OVERFLOW] = 12; continue; }
            m.vox_tris[(size_t)r * (2 * MV_REL_CAP) + pos] = (int)((unsigned int)t | ((rec.rk & 1) ? TRI_ADD_BIT : 0u));
        }
    }
}
void launch_mesh_pack_smooth(hipStream_t s, const MeshDev& m, MeshSmRec* out, int32_t* count, int cap_rec) { KLAUNCH(mesh_pack_smooth_kernel, dim3(1024), dim3(64), 0, s, m, out, count, cap_rec); }
void launch_mesh_unpack_smooth(hipStream_t s, const MeshDev& m, const void* gathered, size_t capb, int cap_rec) {
    KLAUNCH(mesh_unpack_smooth_kernel, dim3(32), dim3(256), 0, s, m, (const char*)gathered, capb, cap_rec);
}
void launch_mesh_pack_marks(hipStream_t s, const MeshDev& m, MeshMkRec* out, int32_t* count, int cap_rec) { KLAUNCH(mesh_pack_marks_kernel, dim3(2048), dim3(64), 0, s, m, out, count, cap_rec); }
void launch_mesh_unpack_marks(hipStream_t s, const MeshDev& m, const void* gathered, size_t capb, int cap_rec) {
    KLAUNCH(mesh_unpack_marks_kernel, dim3(32), dim3(256), 0, s, m, (const char*)gathered, capb, cap_rec);
}

// =====================================================================================================================
// small sorts: one workgroup, 16-byte records in LDS (<= 8192 records = 128 KB), bitonic network with a lexicographic (k0, k1)
// comparator.  Replaces ~10 multi-kernel device-wide radix sorts per scan whose cost at these sizes is pure launch latency.
// =====================================================================================================================
struct SortRec { unsigned long long k0, k1; };
IMD bool rec_gt(const SortRec& a, const SortRec& b) { return a.k0 > b.k0 || (a.k0 == b.k0 && a.k1 > b.k1); }
template <int NT>
IMD void lds_sort_recs(SortRec* a, int np2, int tid) {
    for (int k = 2; k <= np2; k <<= 1)
        for (int lj = 31 - __clz(k >> 1); lj >= 0; lj--) {   // j = 2^lj: shifts, not the integer division a runtime j costs in every stage
            const int j = 1 << lj;
            for (int p = tid; p < (np2 >> 1); p += NT) {
                const int i = ((p >> lj) << (lj + 1)) | (p & (j - 1)), ixj = i + j;
                const bool up = ((i & k) == 0);
                const SortRec x = a[i], y = a[ixj];
                if (rec_gt(x, y) == up) { a[i] = y; a[ixj] = x; }
            }
            __syncthreads();
        }
}
// Sorting the per-scan result lists (remove / add / flip-update triangle lists by (v0, v1, v2); smoothed vertex ids; active voxels by
// key) in two launches for all lists at once, any length:
//   mesh_chunk_sort_kernel   one workgroup per 1024-record chunk: build the records, bitonic sort in 16 KB of LDS, store the sorted chunk
//   mesh_merge_emit_kernel   one thread per record: final position = position in its own chunk + number of smaller records in every
//                            other chunk (binary searches, 4 chunks in flight); the output entry is written directly at that position
#define LS_CHUNK 1024
// Per-scan lists of a stream are a few hundred entries: the LDS bitonic sort of a chunk is a chain of log^2 barrier-separated stages (55 at 1024
// records, 27 us on the mesher's longest chain), so short lists are cut into chunks of 256 (36 stages of half the width; the merge pays two or three more
// binary searches per record); long lists (offline clouds) keep 1024 -- the merge is linear in the number of chunks.
#define LS_CHUNK_SMALL 256
#define LS_SMALL_LIST 4096
// element `j` of a small plan array BY VALUE (a chain of selects over compile-time indices): `a[j]` with a run-time j sends the whole LSortPlan to scratch
// memory -- 112 bytes per lane, written by EVERY lane of the launch: mesh_merge_emit_kernel's 6.2 MB and mesh_chunk_sort_kernel's 1.8 MB of WRITE_SIZE per
// launch in traffic_r04.json (for ~0.1 MB of records) were exactly that, plus a scratch round trip on the mesher's longest chain (round 5)
template <int N> IMD int pick(const int (&a)[N], int j) {
    int v = a[0];
#pragma unroll
    for (int k = 1; k < N; k++) { const int ak = a[k]; v = (j == k) ? ak : v; }   // (by value: `c ? a[k] : v` on lvalues selects ADDRESSES and pins the array to memory)
    return v;
}
template <int N> IMD void lsort_locate(const int (&base)[N], int blk, int& job, int& local) {
    job = 0;
    int b = base[0];
#pragma unroll
    for (int j = 1; j < LS_JOBS; j++) { const int bj = base[j]; const bool ge = blk >= bj; job = ge ? j : job; b = ge ? bj : b; }
    local = blk - b;
}
// which 0: the active-voxel list; which 1: the four result lists.  Lengths are read from the device counters, so the launch needs no host sync.
IMD void lsort_plan_dev(const MeshDev& m, int which, LSortPlan& pl) {
    int n[LS_JOBS] = {0, 0, 0, 0, 0};
    if (which == 0) n[4] = min(m.sc[SC_ACTIVE], m.cap_active);
    else { n[0] = min(m.sc[SC_REM], m.cap_list); n[1] = min(m.sc[SC_ADD], m.cap_list); n[2] = min(m.sc[SC_UPD], m.cap_list); n[3] = min(m.sc[SC_SMOOTH], m.cap_list); }
    int blk = 0, eblk = 0, off = 0;
#pragma unroll
    for (int j = 0; j < LS_JOBS; j++) {
        pl.n[j] = n[j]; pl.blk_base[j] = blk; pl.eblk_base[j] = eblk; pl.rec_off[j] = off;
        const int cs = n[j] <= LS_SMALL_LIST ? LS_CHUNK_SMALL : LS_CHUNK;
        pl.cs[j] = cs;
        const int chunks = (n[j] + cs - 1) / cs;
        blk += chunks; eblk += (n[j] + 255) / 256; off += chunks * cs;
    }
    pl.blk_base[LS_JOBS] = blk; pl.eblk_base[LS_JOBS] = eblk;
}
__global__ __launch_bounds__(256) void mesh_chunk_sort_kernel(MeshDev m_in, int which, SortRec* __restrict__ recs_out) {
    if (which == 1) MESH_MARK(m_in, 8);
    __builtin_amdgcn_s_setprio(MESH_B_PRIO);   // phase B is the mesher's longest chain: issue ahead of the map update's waves (1), behind the registration's (3)
    MESH_DYN(m_in);
    __shared__ SortRec recs[LS_CHUNK];
    LSortPlan pl;
    lsort_plan_dev(m, which, pl);
    const int tid = threadIdx.x;
    for (int blk = blockIdx.x; blk < pl.blk_base[LS_JOBS]; blk += gridDim.x) {
        int job, chunk;
        lsort_locate(pl.blk_base, blk, job, chunk);
        const int n = pick(pl.n, job);
        const int cs = pick(pl.cs, job);
        const int rec_off = pick(pl.rec_off, job);
        const int first = chunk * cs, cnt = min(cs, n - first);
        const int32_t* list = job == 0 ? m.list_rem : (job == 1 ? m.list_add : (job == 2 ? m.list_upd : (job == 3 ? m.list_smooth : nullptr)));
        const int np2 = next_pow2_i(cnt);
        for (int i = tid; i < np2; i += 256) {
            SortRec r; r.k0 = ~0ull; r.k1 = ~0ull;
            if (i < cnt) {
                if (job < 3) {
                    const int t = list[first + i];
                    r.k0 = ((unsigned long long)(unsigned int)m.t_v[(size_t)t * 3 + 0] << 32) | (unsigned long long)(unsigned int)m.t_v[(size_t)t * 3 + 1];
                    r.k1 = ((unsigned long long)(unsigned int)m.t_v[(size_t)t * 3 + 2] << 32) | (unsigned long long)(unsigned int)t;
                } else if (job == 3) { r.k0 = (unsigned long long)(unsigned int)list[first + i]; r.k1 = 0; }
                else { r.k0 = m.act_key[first + i]; r.k1 = (unsigned long long)(unsigned int)m.act_vox[first + i]; }
            }
            recs[i] = r;
        }
        __syncthreads();
        lds_sort_recs<256>(recs, np2, tid);
        for (int i = tid; i < cnt; i += 256) recs_out[(size_t)rec_off + first + i] = recs[i];
        __syncthreads();
    }
}
IMD int lsort_lower_bound(const SortRec* __restrict__ a, int n, const SortRec& key) {  // number of records < key
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; const SortRec v = a[mid]; if (rec_gt(key, v)) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void mesh_merge_emit_kernel(MeshDev m_in, int which, const SortRec* __restrict__ recs, int32_t* __restrict__ add_sorted) {
    if (which == 1) MESH_MARK(m_in, 9);
    __builtin_amdgcn_s_setprio(MESH_B_PRIO);
    MESH_DYN(m_in);
    LSortPlan pl;
    lsort_plan_dev(m, which, pl);
    for (int eblk = blockIdx.x; eblk < pl.eblk_base[LS_JOBS]; eblk += gridDim.x) {
    int job, lb;
    lsort_locate(pl.eblk_base, eblk, job, lb);
    const int i = lb * 256 + threadIdx.x;
    const int n = pick(pl.n, job);
    if (i >= n) continue;
    const SortRec* base = recs + (size_t)pick(pl.rec_off, job);
    const SortRec r = base[i];
    const int cs = pick(pl.cs, job);
    const int own = i / cs, nchunks = (n + cs - 1) / cs;
    int rank = i - own * cs;
    for (int c0 = 0; c0 < nchunks; c0 += 4) {  // four independent binary searches in flight
        int lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int cc = c0 + u; lo[u] = 0; hi[u] = (cc < nchunks && cc != own) ? min(cs, n - cc * cs) : 0; }
        for (int step = 0; step < 11; step++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (lo[u] < hi[u]) {
                    const int mid = (lo[u] + hi[u]) >> 1;
                    const SortRec v = base[(size_t)(c0 + u) * cs + mid];
                    if (rec_gt(r, v)) lo[u] = mid + 1; else hi[u] = mid;
                }
        }
        rank += lo[0] + lo[1] + lo[2] + lo[3];
    }
    if (job < 3) {
        int32_t* out_tri = job == 0 ? m.out_tri_rem : (job == 1 ? m.out_tri_add : m.out_tri_upd);
        uint8_t* out_flip = job == 0 ? nullptr : (job == 1 ? m.out_flip_add : m.out_flip_upd);
        const int t = (int)(unsigned int)(r.k1 & 0xFFFFFFFFull);
        out_tri[(size_t)rank * 3 + 0] = (int)(r.k0 >> 32); out_tri[(size_t)rank * 3 + 1] = (int)(unsigned int)(r.k0 & 0xFFFFFFFFull); out_tri[(size_t)rank * 3 + 2] = (int)(r.k1 >> 32);
        if (out_flip) out_flip[rank] = (uint8_t)m.t_flip[t];
        if (job == 1) add_sorted[rank] = t;
        if (m.shard_world > 1) {
            // sharded mesher: every rank that holds the triangle commits the change; the rank owning the voxel of its SMALLEST vertex reports it --
            // the union of the ranks' result lists is the serial list, every entry exactly once
            const bool mine = mesh_owner(m, m.vx_key[m.v_voxel[(int)(r.k0 >> 32)]]) == m.shard_rank;
            (job == 0 ? m.out_own_rem : (job == 1 ? m.out_own_add : m.out_own_upd))[rank] = mine ? 1 : 0;
            if (mine) atomicAdd(&m.sc[job == 0 ? SC_REM_OWN : (job == 1 ? SC_ADD_OWN : SC_UPD_OWN)], 1);
        }
    } else if (job == 3) {
        const int id = (int)(unsigned int)r.k0;
        m.out_smooth_ids[rank] = id;
        m.out_smooth_xyz[(size_t)rank * 3 + 0] = m.v_smooth[(size_t)id * 3 + 0];
        m.out_smooth_xyz[(size_t)rank * 3 + 1] = m.v_smooth[(size_t)id * 3 + 1];
        m.out_smooth_xyz[(size_t)rank * 3 + 2] = m.v_smooth[(size_t)id * 3 + 2];
    } else {
        const int vi = (int)(unsigned int)r.k1;
        m.act_vox_s[rank] = vi;
        m.vx_rank[vi] = rank;
        m.vx_rank_seq[vi] = m.seq;
    }
    }
}

// Triangle_manager::remove_triangle_list (triangle.hpp:212-221): drop from the live set and from its smallest vertex's list
__global__ void mesh_commit_rem_kernel(MeshDev m_in, const int32_t* __restrict__ tris) {
    MESH_DYN(m_in);
    const int n = min(m.sc[SC_REM], m.cap_list);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int t = tris[i];
        m.t_live[t] = 0;
        const int v0 = m.t_v[(size_t)t * 3 + 0];
        bool done = false;
        for (int ch = m.a_head[v0]; ch >= 0 && !done; ch = m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + MV_ADJ_STRIDE - 1])
            for (int s = 0; s < MV_ADJ_SLOTS; s++)
                if (m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + s * 3] == t) { m.a_chunks[(size_t)ch * MV_ADJ_STRIDE + s * 3] = -1; done = true; break; }
    }
}
// Triangle_manager::insert_triangle (triangle.hpp:330-395).  The list is sorted by triplet, so triangles sharing their smallest
// vertex are contiguous: the lane at the head of such a run inserts the whole run -- no two lanes touch the same vertex list.
__global__ void mesh_commit_add_kernel(MeshDev m_in, const int32_t* __restrict__ tris) {
    MESH_MARK(m_in, 10);
    __builtin_amdgcn_s_setprio(MESH_B_PRIO);
    MESH_DYN(m_in);
    const int n = min(m.sc[SC_ADD], m.cap_list);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int v0 = m.t_v[(size_t)tris[i] * 3 + 0];
    if (i > 0 && m.t_v[(size_t)tris[i - 1] * 3 + 0] == v0) continue;
    int ch = m.a_head[v0], s = 0;
    for (int j = i; j < n; j++) {
        const int t = tris[j];
        if (m.t_v[(size_t)t * 3 + 0] != v0) break;
        const int v1 = m.t_v[(size_t)t * 3 + 1], v2 = m.t_v[(size_t)t * 3 + 2];
        m.t_live[t] = 1;
        bool placed = false;
        while (ch >= 0 && !placed) {
            int* cp = m.a_chunks + (size_t)ch * MV_ADJ_STRIDE;
            for (; s < MV_ADJ_SLOTS; s++)
                if (cp[s * 3] < 0) { cp[s * 3 + 1] = v1; cp[s * 3 + 2] = v2; cp[s * 3] = t; placed = true; s++; break; }
            if (!placed) { ch = cp[MV_ADJ_STRIDE - 1]; s = 0; }
        }
        if (!placed) {  // every chunk of the chain is full: push a new chunk at the front
            const int nc = atomicAdd(&m.pc[PC_ADJ_CHUNKS], 1);
            if (nc >= m.cap_adj_chunks) { m.sc[SC_OVERFLOW] = 13; break; }
            int* cp = m.a_chunks + (size_t)nc * MV_ADJ_STRIDE;
            cp[0] = t; cp[1] = v1; cp[2] = v2;
            for (int k = 1; k < MV_ADJ_SLOTS; k++) cp[k * 3] = -1;
            cp[MV_ADJ_STRIDE - 1] = m.a_head[v0];
            m.a_head[v0] = nc;
            ch = nc; s = 1;
        }
    }
    }
}
// ---- launchers ------------------------------------------------------------------------------------------------------
static inline dim3 g1(int n, int b = 256) { return dim3((unsigned)((n + b - 1) / b)); }

void launch_mesh_transform(hipStream_t s, const float* raw_xyzi, float* world_xyzi, int n, const double* R, const double* t, const double* extR,
                           const double* extT, const double* rt_dev) {
    XformParams xp;
    for (int i = 0; i < 9; i++) { xp.R[i] = R ? R[i] : 0.0; xp.extR[i] = extR[i]; }
    for (int i = 0; i < 3; i++) { xp.t[i] = t ? t[i] : 0.0; xp.extT[i] = extT[i]; }
    KLAUNCH(mesh_transform_kernel, g1(n), dim3(256), 0, s, (const float4*)raw_xyzi, (float4*)world_xyzi, n, xp, rt_dev);
}
// n_cand only sizes the grids here; the kernels take every per-scan value from MeshDev::dyn
void launch_mesh_append_prepare(hipStream_t s, const MeshDev& m, int n_cand, const float* pts) {
    KLAUNCH(mesh_append_prepare_kernel, g1(n_cand <= MV_FIN_CAND ? MV_BIN_NSLOT + n_cand : n_cand), dim3(256), 0, s, m, pts);
}
void launch_mesh_append_resolve(hipStream_t s, const MeshDev& m, int n_cand, const float* pts, int max_iter) {
    KLAUNCH(mesh_append_resolve_kernel, g1(n_cand), dim3(256), 0, s, m, pts, max_iter);
}
void launch_mesh_append_commit(hipStream_t s, const MeshDev& m, int n_cand, const float* pts) {
    KLAUNCH(mesh_append_commit_kernel, g1(n_cand), dim3(256), 0, s, m, pts);
}
void launch_mesh_append_flags(hipStream_t s, const MeshDev& m, int n_cand) { KLAUNCH(mesh_append_flags_kernel, g1(n_cand), dim3(256), 0, s, m); }
void launch_mesh_select_active(hipStream_t s, const MeshDev& m, int n_cand) { KLAUNCH(mesh_select_active_kernel, g1(n_cand), dim3(256), 0, s, m); }
// IMMESH_MESH_GRID_DIV (experiments): divides the grids of the two big per-voxel kernels -- fewer resident mesher wavefronts per SIMD leave register
// room for the registration chain's kernels
static int mesh_grid_div() { static const int v = getenv("IMMESH_MESH_GRID_DIV") ? std::max(1, atoi(getenv("IMMESH_MESH_GRID_DIV"))) : 1; return v; }
void launch_mesh_knn(hipStream_t s, const MeshDev& m) { KLAUNCH(mesh_knn_kernel<false>, dim3(512 / mesh_grid_div()), dim3(256), 0, s, m, (float*)nullptr, 1.0, (const int32_t*)nullptr, 0, (double*)nullptr, 0.0); }
void launch_mesh_export_vertices(hipStream_t s, const MeshDev& m, float* export_vtx, double smooth_factor) {
    KLAUNCH(mesh_knn_kernel<true>, dim3(2048), dim3(256), 0, s, m, export_vtx, smooth_factor, (const int32_t*)nullptr, 0, (double*)nullptr, 0.0);
}
// ---- Global_map::smooth_pts on demand (the renderer's consumer of the map, mesh_rec_display.cpp:78-103) ---------------------------------------------
// voxel of each requested vertex (-1: id out of range; display mode: also -1 for a vertex the mesher has smoothed -- its m_pos_aft_smooth is served as is)
__global__ void mesh_query_voxels_kernel(MeshDev m, const int32_t* __restrict__ ids, int n, int n_vertices, int display, int32_t* __restrict__ vox_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int id = ids[i];
    int v = -1;
    if (id >= 0 && id < n_vertices) {
        v = m.v_voxel[id];
        if (display) {
            const bool smoothed = m.v_smooth[(size_t)id * 3 + 0] != (double)m.v_pos[(size_t)id * 3 + 0] || m.v_smooth[(size_t)id * 3 + 1] != (double)m.v_pos[(size_t)id * 3 + 1] ||
                                  m.v_smooth[(size_t)id * 3 + 2] != (double)m.v_pos[(size_t)id * 3 + 2];
            if (smoothed) v = -2;
        }
    }
    vox_out[i] = v;
}
// results in request order.  display == 0: smooth_pts' return value (doubles); display != 0: RGB_pts::get_pos(1) after the renderer's on-demand smoothing
// (floats: m_pos_aft_smooth where the mesher has smoothed the vertex, smooth_pts' value elsewhere)
__global__ void mesh_query_gather_kernel(MeshDev m, const int32_t* __restrict__ ids, const int32_t* __restrict__ vox, int n, const double* __restrict__ export_d, int display,
                                         double* __restrict__ out_d, float* __restrict__ out_f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int id = ids[i];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double v = vox[i] == -2 ? m.v_smooth[(size_t)id * 3 + a] : (vox[i] >= 0 ? export_d[(size_t)id * 3 + a] : __longlong_as_double(0x7FF8000000000000ll));
        if (display) out_f[(size_t)i * 3 + a] = (float)v; else out_d[(size_t)i * 3 + a] = v;
    }
}
void launch_mesh_query_voxels(hipStream_t s, const MeshDev& m, const int32_t* ids, int n, int n_vertices, int display, int32_t* vox_out) {
    KLAUNCH(mesh_query_voxels_kernel, dim3((n + 255) / 256), dim3(256), 0, s, m, ids, n, n_vertices, display, vox_out);
}
void launch_mesh_query_smooth(hipStream_t s, const MeshDev& m, const int32_t* vox_list, int n_list, double smooth_factor, double max_dis, double* export_d) {
    KLAUNCH(mesh_knn_kernel<true>, dim3(std::min(std::max(n_list, 1), 2048)), dim3(256), 0, s, m, (float*)nullptr, smooth_factor, vox_list, n_list, export_d, max_dis);
}
void launch_mesh_query_gather(hipStream_t s, const MeshDev& m, const int32_t* ids, const int32_t* vox, int n, const double* export_d, int display, double* out_d, float* out_f) {
    KLAUNCH(mesh_query_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, m, ids, vox, n, export_d, display, out_d, out_f);
}
// live triangles with the winding save_to_ply_file writes: m_index_flip != 0 -> (v0, v1, v2), else (v0, v2, v1)
__global__ void mesh_export_faces_kernel(MeshDev m, int32_t* __restrict__ tri_idx, int32_t* __restrict__ count) {
    const int nt = m.pc[PC_TRIS];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += gridDim.x * blockDim.x)
        if (m.t_live[t] && (m.shard_world <= 1 || mesh_owner(m, m.vx_key[m.v_voxel[m.t_v[(size_t)t * 3]]]) == m.shard_rank)) tri_idx[atomicAdd(count, 1)] = t;
}
__global__ void mesh_export_wind_kernel(MeshDev m, const int32_t* __restrict__ tri_sorted, int n, int32_t* __restrict__ faces) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = tri_sorted[i];
    const int a = m.t_v[(size_t)t * 3 + 0], b = m.t_v[(size_t)t * 3 + 1], c = m.t_v[(size_t)t * 3 + 2];
    const bool keep = m.t_flip[t] != 0;
    faces[(size_t)i * 3 + 0] = a; faces[(size_t)i * 3 + 1] = keep ? b : c; faces[(size_t)i * 3 + 2] = keep ? c : b;
}
__global__ void mesh_export_keys_kernel(MeshDev m, const int32_t* __restrict__ tris, int n, int which, uint32_t* __restrict__ k32, unsigned long long* __restrict__ k64) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = tris[i];
    if (which == 0) k32[i] = (uint32_t)m.t_v[(size_t)t * 3 + 2];
    else k64[i] = ((unsigned long long)(unsigned int)m.t_v[(size_t)t * 3 + 0] << 32) | (unsigned long long)(unsigned int)m.t_v[(size_t)t * 3 + 1];
}
void launch_mesh_export_faces(hipStream_t s, const MeshDev& m, int32_t* tri_idx, int32_t* count) { KLAUNCH(mesh_export_faces_kernel, dim3(1024), dim3(256), 0, s, m, tri_idx, count); }
void launch_mesh_export_keys(hipStream_t s, const MeshDev& m, const int32_t* tris, int n, int which, uint32_t* k32, unsigned long long* k64) {
    KLAUNCH(mesh_export_keys_kernel, g1(n), dim3(256), 0, s, m, tris, n, which, k32, k64);
}
void launch_mesh_export_wind(hipStream_t s, const MeshDev& m, const int32_t* tri_sorted, int n, int32_t* faces) { KLAUNCH(mesh_export_wind_kernel, g1(n), dim3(256), 0, s, m, tri_sorted, n, faces); }
void launch_mesh_delaunay(hipStream_t s, const MeshDev& m) {
    // (grids: the dispatcher places ~130 workgroups per us, so a launch of 2048 workgroups lasts >= 16 us however little they do; the kernels stride)
    KLAUNCH(mesh_delaunay64_kernel<0>, dim3(768 / mesh_grid_div()), dim3(64), 0, s, m);           // n_u <= 64: register fast path
    KLAUNCH(mesh_delaunay_general_kernel, dim3(MV_GEN_BLOCKS), dim3(64), 0, s, m);     // 64 < n_u and what the fast path handed over
}
// the same in two launches on two streams (mesh_delaunay64.inc): the triangulations behind phase A, the diff against the live set at the head of phase B
void launch_mesh_tri64(hipStream_t s, const MeshDev& m) { KLAUNCH(mesh_tri64_kernel, dim3(768 / mesh_grid_div()), dim3(64), 0, s, m); }
void launch_mesh_diff64(hipStream_t s, const MeshDev& m) {
    KLAUNCH(mesh_diff64_kernel, dim3(768 / mesh_grid_div()), dim3(64), 0, s, m);
    KLAUNCH(mesh_delaunay_general_kernel, dim3(MV_GEN_BLOCKS), dim3(64), 0, s, m);
}
void launch_mesh_finalize(hipStream_t s, const MeshDev& m) { KLAUNCH(mesh_finalize_kernel, dim3(128), dim3(64), 0, s, m); }
void launch_mesh_commit_rem(hipStream_t s, const MeshDev& m, const int32_t* tris) { KLAUNCH(mesh_commit_rem_kernel, dim3(128), dim3(256), 0, s, m, tris); }
void launch_mesh_commit_add(hipStream_t s, const MeshDev& m, const int32_t* tris_sorted) { KLAUNCH(mesh_commit_add_kernel, dim3(128), dim3(256), 0, s, m, tris_sorted); }
// which 0: active-voxel list (-> act_vox_s + ranks); which 1: remove / add / flip-update / smooth lists (-> sorted outputs)
void launch_mesh_sort_emit(hipStream_t s, const MeshDev& m, int which, void* recs, int32_t* add_sorted) {
    KLAUNCH(mesh_chunk_sort_kernel, dim3(which == 0 ? 32 : 64), dim3(256), 0, s, m, which, (SortRec*)recs);
    KLAUNCH(mesh_merge_emit_kernel, dim3(which == 0 ? 64 : 256), dim3(256), 0, s, m, which, (const SortRec*)recs, add_sorted);
}
