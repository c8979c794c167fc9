// HBM-resident registration map: what the reference keeps as std::unordered_map<VOXEL_LOC, OctoTree*> + heap OctoTree /
// Plane objects (src/voxel_loc.hpp:89-177) laid out for the GPU:
//   * open-addressing hash  packed 3x21-bit key -> {key, root node id}                (one 16-byte probe per lookup)
//   * flat pool of 384-byte octree-node records (children, flags, plane scalars, centres, normal, 21-entry symmetric plane_var):
//     a plane test is one dependent gather of adjacent cache lines, not a pointer chase through hash -> OctoTree -> Plane
//   * retained points (OctoTree::m_temp_points_) in 16-point chunks of a shared pool (xyz + symmetric 3x3 covariance = 9 doubles/pt)
// All of it lives in one context and never leaves the device between scans.
#pragma once
#include "dev_math.hpp"

#define IM_CHUNK_PTS 16
#define IM_PT_DOUBLES 9           /* x y z + var(00 01 02 11 12 22) */
#define IM_INLINE_CHUNKS 8        /* 128 points inline per node; beyond that an extension table */
#define IM_EXT_CHUNKS 2048        /* + 32768 points: KITTI max_points_size / g_max_points 1000, and buildVoxelMap roots that hold a whole 3 m voxel of the first scan */
#define IM_LEAF_SLOTS 15          /* node ids per leaf-list chunk (+ next pointer = one 64-byte line) */
#define IM_G_MAX_POINTS 1000      /* g_max_points, src/voxel_loc.cpp:45 */

// node flag bits
#define NF_INIT 1          /* m_init_octo_ */
#define NF_PLANE 2         /* m_plane_ptr_->m_is_plane */
#define NF_UPDATE_EN 4     /* m_update_enable_ */

// One octree node = one 384-byte record (6 cache lines): a point-to-plane test reads the topology / gate scalars (line 0), the
// centres + normal (lines 1-2) and the symmetric plane covariance (lines 2-4) with ONE dependent gather instead of a dozen.
struct alignas(64) NodeRec {
    int32_t child[8];                      // m_leaves_ (-1 = none)
    int32_t flags, layer, npts, newpts;    // NF_* bits, m_layer_, m_temp_points_.size(), m_new_points_
    float quarter, d, radius, min_eig;     // m_quater_length_; Plane::m_d, m_radius, m_min_eigen_value
    double center[3];                      // m_voxel_center_
    double p_center[3];                    // Plane::m_center
    double p_normal[3];                    // Plane::m_normal
    double p_var[21];                      // upper triangle of the 6x6 Plane::m_plane_var, row-major
    int32_t chunks[IM_INLINE_CHUNKS];      // chunk ids of the retained points (-1 = none)
    int32_t ext, path;                     // extension table id or -1; child path from the root (3 bits per level)
    unsigned long long key;                // packed root key (for dumps)
    int32_t leaf_head;                     // root nodes: chain of the planar descendants (what build_single_residual's recursion would reach)
    int32_t lock;                          // root nodes: guards leaf_head's chain when several wavefronts update ONE root (replay_sub_kernel); 0 = free
    unsigned long long pad[3];
};
static_assert(sizeof(NodeRec) == 384, "NodeRec layout");

struct HashEnt { unsigned long long key; int32_t root; int32_t pad; };  // one 16-byte probe yields key and root node

struct RegMapDev {
    // hash: open addressing, packed 3x21-bit VOXEL_LOC key -> root node id
    HashEnt* htab;              // [hcap]
    uint64_t hmask;
    // node pool
    NodeRec* nodes;             // [cap_nodes]
    // pools
    double* chunk_data;         // [cap_chunks * IM_CHUNK_PTS * IM_PT_DOUBLES]
    int32_t* ext_tables;        // [cap_ext * IM_EXT_CHUNKS]
    // counters: [0] nodes used [1] chunk bump [2] ready-free top [3] pending-free top [4] ext used [5] overflow flag [6] root voxels [7] touched slots of the current update [8] leaf-list chunks used [9] bump allocator of the long-list scratch (replay_list_kernel) [10] length of the work list the fused replay kernel hands to replay_list_kernel [11] spare
    int32_t* counters;
    int32_t* free_ready;        // chunk ids available for allocation
    int32_t* free_pending;      // chunk ids freed by the running kernel (merged into ready afterwards)
    // per-root flat lists of planar descendants: chunks of IM_LEAF_SLOTS node ids + next pointer (counters[8] = chunks used)
    int32_t* leaf_chunks;
    int32_t cap_leaf_chunks;
    // per-update root-voxel point lists (map_incremental_grow): head word per hash slot = (update seq << 32 | last point index), stamped so it never needs clearing
    unsigned long long* slot_head;  // [hcap]
    uint32_t* touched;          // (hash slot, root node) of the voxels touched by the current update (counters[7] pairs)
    // subtree work items (deep octrees: max_layer >= 3): a subdivided root's points of this scan, split by first-level octant -- the octants are independent
    // state machines -- and replayed by one wavefront each (replay_sub_kernel).  counters[11] = sub_order fill, counters[12] = items
    int32_t split_general;      // 1: replay_list_kernel hands the octants of subdivided roots to replay_sub_kernel
    int32_t* sub_order;         // point indices of the items, each item's in the voxel's replay order
    unsigned long long* sub_items;   // 2 words per item: child node | root << 32, first index in sub_order | count << 32
    int32_t upd_seq;
    // multi-GPU sharding of the registration map: root voxels are owned in bricks of 2^shard_brick_log2 voxels per axis,
    // owner = brick_owner() below; a rank also keeps the 1-voxel halo around its bricks (the near-voxel retry looks one voxel over)
    int32_t shard_rank, shard_world, shard_brick_log2, shard_scheme;
    int32_t cap_nodes, cap_chunks, cap_ext;
    // parameters
    int32_t max_layer, max_points_size, init_size[5];
    float planer_threshold;
    float voxel_size_f;         // the float voxel_size of build/update (voxel_mapping.cpp:110,320)
    double voxel_size_d;        // the double of the matcher (:153)
};

namespace imd {

IMD int sym21_index(int r, int c) {  // r <= c, 6x6 upper triangle row-major
    return r * 6 - (r * (r - 1)) / 2 + (c - r);
}

// ---- sharding ---------------------------------------------------------------------------------------------------
// owner of brick (bx, by, bz) -- ONE function for the registration map, the mesher and the host mirror (immesh_shard_owner): scheme 0 = lattice colouring
// (bx + 3 by + 5 bz) mod P, scheme 1 = hash(brick) mod P (immesh_config::shard_scheme).  The colouring deals every axis out over ALL ranks only while 1, 3
// and 5 are units mod P: for a world divisible by 3 or 5 one axis would drop out (P = 3: every brick along y the same owner), so such worlds use the hash
// whatever the scheme says.
IMD int brick_owner(int scheme, int world, int64_t bx, int64_t by, int64_t bz, uint64_t packed) {
    if (scheme == 1 || world % 3 == 0 || world % 5 == 0) return (int)(hash64(packed) % (uint64_t)world);
    const int64_t c = (bx + 3 * by + 5 * bz) % (int64_t)world;
    return (int)(c < 0 ? c + world : c);
}
IMD int shard_owner(const RegMapDev& m, int64_t kx, int64_t ky, int64_t kz) {
    const int b = m.shard_brick_log2;   // arithmetic shift: bricks tile negative keys too
    return brick_owner(m.shard_scheme, m.shard_world, kx >> b, ky >> b, kz >> b, pack_key(kx >> b, ky >> b, kz >> b));
}
// does this rank keep voxel k ?  (owned, or within one voxel of an owned brick)
IMD bool shard_keeps(const RegMapDev& m, int64_t kx, int64_t ky, int64_t kz) {
    if (shard_owner(m, kx, ky, kz) == m.shard_rank) return true;
    const int64_t msk = ((int64_t)1 << m.shard_brick_log2) - 1;
    const int64_t lx = kx & msk, ly = ky & msk, lz = kz & msk;
    if (lx != 0 && lx != msk && ly != 0 && ly != msk && lz != 0 && lz != msk) return false;   // interior of a foreign brick
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dz = -1; dz <= 1; dz++)
                if ((dx | dy | dz) != 0 && shard_owner(m, kx + dx, ky + dy, kz + dz) == m.shard_rank) return true;
    return false;
}

// ---- hash ---------------------------------------------------------------------------------------------------
IMD int64_t hash_find(const RegMapDev& m, uint64_t key) {  // returns slot or -1
    uint64_t h = hash64(key) & m.hmask;
    for (int probe = 0; probe < 4096; probe++) {
        const unsigned long long k = m.htab[h].key;
        if (k == key) return (int64_t)h;
        if (k == IM_KEY_EMPTY) return -1;
        h = (h + 1) & m.hmask;
    }
    return -1;
}
// find or claim a slot for `key`; *created = true if this call inserted it.  Values are filled by the caller.
IMD int64_t hash_find_or_insert(const RegMapDev& m, uint64_t key, bool* created) {
    uint64_t h = hash64(key) & m.hmask;
    *created = false;
    for (int probe = 0; probe < 4096; probe++) {
        unsigned long long k = m.htab[h].key;
        if (k == key) return (int64_t)h;
        if (k == IM_KEY_EMPTY) {
            const unsigned long long prev = atomicCAS(&m.htab[h].key, (unsigned long long)IM_KEY_EMPTY, (unsigned long long)key);
            if (prev == IM_KEY_EMPTY) { *created = true; return (int64_t)h; }
            if (prev == key) return (int64_t)h;
        }
        h = (h + 1) & m.hmask;
    }
    return -1;
}

// ---- chunk pool --------------------------------------------------------------------------------------------------
IMD int alloc_chunk(const RegMapDev& m) {
    // the ready stack is empty most of the time on a settled map: look before popping (a pop that fails costs two more contended atomics)
    if (__hip_atomic_load(&m.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {
        int i = atomicSub(&m.counters[2], 1) - 1;
        if (i >= 0) return m.free_ready[i];
        atomicAdd(&m.counters[2], 1);  // undo
    }
    const int c = atomicAdd(&m.counters[1], 1);
    if (c >= m.cap_chunks) { m.counters[5] = 1; return -1; }
    return c;
}
IMD void free_chunk(const RegMapDev& m, int c) {
    if (c < 0) return;
    const int i = atomicAdd(&m.counters[3], 1);
    m.free_pending[i] = c;
}
IMD int node_chunk_id(const RegMapDev& m, int node, int ci) {
    if (ci < IM_INLINE_CHUNKS) return m.nodes[node].chunks[ci];
    const int e = m.nodes[node].ext;
    return m.ext_tables[(size_t)e * IM_EXT_CHUNKS + (ci - IM_INLINE_CHUNKS)];
}
IMD double* node_point_ptr(const RegMapDev& m, int node, int i) {
    const int c = node_chunk_id(m, node, i / IM_CHUNK_PTS);
    return m.chunk_data + ((size_t)c * IM_CHUNK_PTS + (i % IM_CHUNK_PTS)) * IM_PT_DOUBLES;
}
// make sure chunk slot `ci` of `node` exists (called by one lane)
IMD bool node_ensure_chunk(const RegMapDev& m, int node, int ci) {
    if (ci < IM_INLINE_CHUNKS) {
        int* slot = &m.nodes[node].chunks[ci];
        if (*slot < 0) { const int c = alloc_chunk(m); if (c < 0) return false; *slot = c; }
        return true;
    }
    if (ci - IM_INLINE_CHUNKS >= IM_EXT_CHUNKS) { m.counters[5] = 2; return false; }
    if (m.nodes[node].ext < 0) {
        const int e = atomicAdd(&m.counters[4], 1);
        if (e >= m.cap_ext) { m.counters[5] = 3; return false; }
        for (int k = 0; k < IM_EXT_CHUNKS; k++) m.ext_tables[(size_t)e * IM_EXT_CHUNKS + k] = -1;
        m.nodes[node].ext = e;
    }
    int* slot = &m.ext_tables[(size_t)m.nodes[node].ext * IM_EXT_CHUNKS + (ci - IM_INLINE_CHUNKS)];
    if (*slot < 0) { const int c = alloc_chunk(m); if (c < 0) return false; *slot = c; }
    return true;
}
// release all retained points of a node (std::vector<Point_with_var>().swap(m_temp_points_)); one lane
IMD void node_free_points(const RegMapDev& m, int node) {
    const int n = m.nodes[node].npts;
    const int nch = (n + IM_CHUNK_PTS - 1) / IM_CHUNK_PTS;
    for (int ci = 0; ci < nch; ci++) {
        if (ci < IM_INLINE_CHUNKS) {
            int* slot = &m.nodes[node].chunks[ci];
            free_chunk(m, *slot); *slot = -1;
        } else {
            int* slot = &m.ext_tables[(size_t)m.nodes[node].ext * IM_EXT_CHUNKS + (ci - IM_INLINE_CHUNKS)];
            free_chunk(m, *slot); *slot = -1;
        }
    }
    m.nodes[node].npts = 0;
}

// allocate + initialise a node (one lane).  centre/quarter/layer as OctoTree ctor + caller-provided geometry.
IMD int node_alloc(const RegMapDev& m, int layer, const double* center, float quarter, unsigned long long key, int path) {
    const int id = atomicAdd(&m.counters[0], 1);
    if (id >= m.cap_nodes) { m.counters[5] = 4; return -1; }
    NodeRec& nd = m.nodes[id];
#pragma unroll
    for (int k = 0; k < 8; k++) nd.child[k] = -1;
#pragma unroll
    for (int k = 0; k < IM_INLINE_CHUNKS; k++) nd.chunks[k] = -1;
    nd.center[0] = center[0]; nd.center[1] = center[1]; nd.center[2] = center[2];
    nd.quarter = quarter;
    nd.flags = NF_UPDATE_EN;
    nd.layer = layer;
    nd.npts = 0; nd.newpts = 0; nd.ext = -1;
    nd.key = key; nd.path = path; nd.leaf_head = -1; nd.lock = 0;
    nd.d = 0.f; nd.radius = 0.f; nd.min_eig = 1.f;
    return id;
}

// ---- per-root list of planar descendants (maintained by the voxel's own wavefront during map build / update; one lane) -----------------
// The matcher's recursion "test every plane reachable through non-planar ancestors" visits exactly the nodes with NF_PLANE set below a
// non-planar root (a planar node never has children), so a flat list turns a pointer chase over hundreds of internal nodes into a scan.
IMD void leaf_add(const RegMapDev& m, int root, int nd) {
    for (int ch = m.nodes[root].leaf_head; ch >= 0; ch = m.leaf_chunks[(size_t)ch * 16 + 15])
        for (int s = 0; s < IM_LEAF_SLOTS; s++)
            if (m.leaf_chunks[(size_t)ch * 16 + s] < 0) { m.leaf_chunks[(size_t)ch * 16 + s] = nd; return; }
    const int nc = atomicAdd(&m.counters[8], 1);
    if (nc >= m.cap_leaf_chunks) { m.counters[5] = 7; return; }
    m.leaf_chunks[(size_t)nc * 16 + 0] = nd;
    for (int s = 1; s < IM_LEAF_SLOTS; s++) m.leaf_chunks[(size_t)nc * 16 + s] = -1;
    m.leaf_chunks[(size_t)nc * 16 + 15] = m.nodes[root].leaf_head;
    m.nodes[root].leaf_head = nc;
}
IMD void leaf_remove(const RegMapDev& m, int root, int nd) {
    for (int ch = m.nodes[root].leaf_head; ch >= 0; ch = m.leaf_chunks[(size_t)ch * 16 + 15])
        for (int s = 0; s < IM_LEAF_SLOTS; s++)
            if (m.leaf_chunks[(size_t)ch * 16 + s] == nd) { m.leaf_chunks[(size_t)ch * 16 + s] = -1; return; }
}
// write a node's flags; keeps the root's leaf list in step with the plane bit (one lane).  shared: other wavefronts update other subtrees of the SAME
// root at the same time (replay_sub_kernel) -- the chain is then edited under the root's lock, with device-scope acquire / release around it
IMD void node_set_flags(const RegMapDev& m, int root, int nd, int old_flags, int new_flags, bool shared = false) {
    m.nodes[nd].flags = new_flags;
    if (nd != root && ((old_flags ^ new_flags) & NF_PLANE)) {
        if (shared) {
            // bounded like every other poll of the library: the lock is held for a list edit of a few hundred cycles by wavefronts
            // that are RUNNING (replay_sub_kernel's grid is resident), so 2^20 tries (~0.1 s) mean something is wrong -- the update fails as a whole
            // (flag 8) instead of hanging the device
            int tries = 0;
            while (atomicCAS(&m.nodes[root].lock, 0, 1) != 0) {
                if (++tries > (1 << 20)) { m.counters[5] = 8; return; }
                __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (new_flags & NF_PLANE) leaf_add(m, root, nd); else leaf_remove(m, root, nd);
        if (shared) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicExch(&m.nodes[root].lock, 0);
        }
    }
}
// position of a node in the depth-first order of the reference's recursion (child indices, first level most significant)
IMD unsigned int dfs_key(const RegMapDev& m, int nd) {
    const int path = m.nodes[nd].path, layer = m.nodes[nd].layer;
    unsigned int k = 0;
    for (int l = 1; l <= layer; l++) k |= (unsigned int)((path >> (3 * (l - 1))) & 7) << (3 * (4 - l));
    return k;
}

}  // namespace imd
