// HBM-resident mesh map + launcher prototypes of the mesher half of the hot path.
// What the reference keeps as Global_map (vector<shared_ptr<RGB_pts>>, two Hash_map_3d, ikd-Tree of vertices;
// src/meshing/r3live/pointcloud_rgbd.hpp:234-298) and Triangle_manager (triangle hash + vertex->triangle adjacency;
// src/meshing/r3live/triangle.hpp:115-395) becomes:
//   * vertex store, SoA                      v_pos f32 xyz (the reference's f64 m_pos holds exactly these float values),
//                                            v_smooth f64 xyz (m_pos_aft_smooth), v_voxel (owner mesh voxel)
//   * dedupe grid hash (m_hashmap_3d_pts)    packed 3x21-bit cell key -> vertex id         (<= 1 vertex per min-spacing cell)
//   * mesh-voxel hash (m_hashmap_voxels)     packed key -> voxel index; voxel SoA with a fixed-stride point list
//     -- the grid doubles as the spatial index that replaces the ikd-Tree: the 1-NN insert test probes the 27 cells around
//        a candidate, the 20-NN neighbourhood pull stages the voxel block around a mesh voxel through LDS.
//   * triangle pool + triangle hash           sorted-triplet -> triangle index (m_triangle_hash; entries persist after erase),
//     live flag, flip word, and "triangles by smallest vertex" chunk lists (the role of m_map_pt_triangle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include <string>
#include <deque>
#include <thread>
#include <mutex>
#include <condition_variable>
#include "../../include/immesh_c_api.h"
#include "prof.hpp"

#define MV_VOX_CAP 128      /* vertices per mesh voxel: ((int)(voxel/min_spacing)+1)^3 = 125 for every shipped config */
#define MV_KNN 20           /* neighbours pulled per vertex, mesh_rec_geometry.cpp:350 */
#define MV_REL_CAP 1024     /* vertices in one voxel's neighbourhood union (n_u) */
#define MV_ADJ_SLOTS 5      /* (triangle, v1, v2) entries per adjacency chunk */
#define MV_ADJ_STRIDE 16     /* ints per chunk: 5 x 3 + next pointer in the last int = one 64-byte line */

// per-scan device counters (MeshDev::sc)
enum {
    SC_UNDECIDED = 0, SC_ACCEPTED, SC_RECENT, SC_ACTIVE, SC_ADD, SC_REM, SC_UPD, SC_SMOOTH, SC_OVERFLOW, SC_C1, SC_C20, SC_NV, SC_NU, SC_TV, SC_MAXNU, SC_PASS2,
    SC_VTXBASE,      /* vertex count when the scan started (the id of its first new vertex) */
    SC_ADD_OWN, SC_REM_OWN, SC_UPD_OWN,   /* sharded mesher: entries of the result lists this rank REPORTS (the lists it commits also hold its halo) */
    SC_SMOOTH_RX,    /* sharded mesher: smoothed positions received from other ranks this scan */
    SC_XBYTES,       /* sharded mesher: payload bytes this rank contributed to the scan's exchanges */
    SC_DEGEN,        /* neighbourhood points the triangulations of this scan did not insert (no circumdisk contains them: immesh_counters_t::n_degenerate_skips) */
    SC_COUNT = 24
};
// persistent device counters (MeshDev::pc)
enum { PC_VERTS = 0, PC_VOXELS, PC_TRIS, PC_ADJ_CHUNKS, PC_LIVE, PC_COUNT = 8 };

#define LS_JOBS 5   /* sorted per scan: remove / add / flip-update triangle lists, smoothed vertex ids, active voxels */
struct LSortPlan { int n[LS_JOBS]; int blk_base[LS_JOBS + 1]; int eblk_base[LS_JOBS + 1]; int rec_off[LS_JOBS]; int cs[LS_JOBS]; };   // cs: chunk size of the job

struct MeshVoxEnt { unsigned long long key; int32_t val; int32_t pad; };   // key == ~0: empty; val == -1: being created
struct MeshGridEnt { float x, y, z; int32_t id; unsigned long long key; unsigned long long pad; };   // 32 B; key == ~0: empty
struct MeshScanParams {
    double cam[3];
    int32_t n_raw, step, n_cand, vtx_base;
};
// everything that changes from scan to scan, in device memory (the kernels' arguments stay constant -> graph replay)
struct MeshDyn {
    MeshScanParams sp;
    int32_t seq, pad;
    unsigned long long ch_mask;
    // != nullptr: the scan reaches its world buffer in the epilogue of the registration launch, which then stores wait_seq there -- the first
    // kernel of the scan waits for it (no event record -- a barrier packet -- on the pose chain)
    const unsigned long long* wait_flag; unsigned long long wait_seq;
    const float* pts;   // the scan (world-frame xyzI): the admission kernels take it from here when their argument is null (graph replay: any world buffer)
};

struct MeshDev {
    // vertices
    float* v_pos; double* v_smooth; double* v_smooth_new; int32_t* v_voxel;
    // dedupe grid hash: 32-byte entries {vertex xyz + id, cell key}; the home slot of a cell is (hash(brick of 4x4x4 cells) << 6) | (cell within the
    // brick, z fastest), linear probing from there -- the four cells of a z-row share one 128-byte line, so the 27-cell probe of a candidate touches
    // ~13 lines instead of 54 (key and record of every cell in lines of their own) and candidates of one neighbourhood touch the SAME lines
    MeshGridEnt* g_ent; uint64_t g_mask;
    // mesh voxels
    MeshVoxEnt* x_ent; uint64_t x_mask;                        // 16-byte entries {key, voxel index}: one round trip per lookup
    unsigned long long* vx_key; int32_t* vx_npts; int32_t* vx_pts; int32_t* vx_meshing_times; int32_t* vx_new_added; int32_t* vx_stamp;
    int32_t* vx_rank; int32_t* vx_rank_seq; int32_t* vx_rank_seq_alt; int32_t* vx_rank_seq_alt2; double* vx_short_axis;   // (rank, stamp): per job parity; _alt = the other parity's stamps
    // triangles
    int32_t* t_v; unsigned long long* t_word; int32_t* t_live; int32_t* t_rem_seq; int8_t* t_flip;
    int32_t* th_slots; uint64_t th_mask;
    int32_t* a_head; int32_t* a_chunks;
    // counters
    int32_t* sc; int32_t* pc;
    // per-scan scratch
    int32_t* cand_status; int32_t* cand_vox; unsigned long long* cand_cell; int32_t* cand_next; int32_t* cand_rank;
    int32_t* cand_flags;                                                   // sharded admission (CF_* of mesh_kernels.hip)
    int32_t* bin_cnt;                                                      // admission order: 2 parities x (bucket fill counts + overflow count)
    float4* cand_pt;                                                       // the candidates (xyz, scan index in w) in admission order: bucketed by 8-cell cube (mesh_begin_scan_kernel)
    unsigned long long* ch_keys; int32_t* ch_head; uint64_t ch_mask;       // candidate-cell chains
    int32_t* recent;                                                       // voxel indices visited this scan
    unsigned long long* act_key; int32_t* act_vox; unsigned long long* act_key_s; int32_t* act_vox_s;
    int32_t* rel_ids; int32_t* rel_n; int32_t* rel_nq;                     // [n_active][MV_REL_CAP], [n_active], [n_active] (vertices the voxel held when it was searched)
    unsigned int* tri_fh; int32_t* tri_nf; double* tri_axis;                // per job parity: mesh_tri64_kernel -> mesh_diff64_kernel ([n_active][256] fresh-face hash set, face count / -1, short axis)
    int32_t* vox_tris; int32_t* vox_ntris;                                 // [n_active][2*MV_REL_CAP] triangle ids touched (bit 31 = add)
    int32_t* list_add; int32_t* list_rem; int32_t* list_upd; int32_t* list_smooth;   // unsorted unique lists
    int32_t* list_smooth_rx;                                               // sharded mesher: vertices whose smoothed positions arrived from other ranks
    uint8_t* out_own_add; uint8_t* out_own_rem; uint8_t* out_own_upd;      // sharded mesher: 1 = this rank reports the entry of the sorted list
    // sorted outputs
    int32_t* out_tri_add; uint8_t* out_flip_add; int32_t* out_tri_rem; int32_t* out_tri_upd; uint8_t* out_flip_upd; int32_t* out_smooth_ids;
    double* out_smooth_xyz;
    // capacities
    int32_t cap_verts, cap_voxels, cap_tris, cap_adj_chunks, cap_cand, cap_active, cap_list;
    // parameters
    double min_spacing, voxel, accept;   // accept = voxel * 1.25 (g_kd_tree_accept_pt_dis, mesh_rec_geometry.cpp:343)
    unsigned long long* tick0;   // s_memrealtime of the job's first kernel (per parity); mesh_publish_kernel turns it into the job's device time
    unsigned char* dv_scratch;   // mesh_delaunay_general_kernel: per-block tables of the neighbourhoods above 256 vertices (MV_GEN_BLOCKS x MV_GEN_SCRATCH bytes)
    int32_t shard_rank, shard_world, shard_brick_log2, shard_scheme;   // sharded mesher: owner-computes per mesh-voxel brick (shard_world <= 1: off)
    int32_t seq;                         // scan sequence number (>= 1); kernels take it (and ch_mask) from *dyn
    MeshDyn* dyn;                        // per-scan parameters (device memory)
    unsigned long long* dbg;             // optional phase timers (IMMESH_DEBUG): [16] sums of s_memtime deltas, nullptr = off
};

// Two views of the mesh map, one per job parity.  A scan is meshed in two phases: A = vertex admission + neighbourhood search (a17-a19),
// B = triangulation + diff + commit (a20-a24).  A(k+1) depends only on A(k); B(k) on A(k) and B(k-1) -- so A(k+1) runs on its own stream
// while B(k) is still busy (the reference, too, lets frame k+1 append while frame k triangulates, ImMesh_mesh_reconstruction.cpp:60-61,
// there without a defined order).  Everything A(k+1) writes and B(k) reads is double-buffered by parity: per-scan counters, dyn, the active
// list + ranks, the neighbourhood lists, this scan's smoothed positions, the result lists.
// One queued incremental_mesh_reconstruction call.  The reference runs the mesher on its own service thread + pool
// (service_reconstruct_mesh, ImMesh_mesh_reconstruction.cpp:272-310) so scan k's meshing overlaps scan k+1's registration; here a
// worker thread drives a second HIP stream, strictly in submission order (the sequential-deterministic frame order of the checker).
// exchange records of the sharded mesher: this scan's smoothed positions of the vertices of the voxels a rank meshed, and the
// triangle marks its triangulations produced (rk = (voxel rank << 1) | add, word = the rank's current flip word; rk = -1: removal mark)
struct MeshCdRec { int32_t i, status; };          // sharded admission: candidate (scan index) + ST_UNDECIDED (band survivor) / ST_ACCEPT / ST_REJECT
struct MeshSmRec { int32_t id, pad; double x, y, z; };
struct MeshMkRec { int32_t a, b, c, rk; unsigned long long word; };
#define MESH_WORLD_BUFS 4
#define MESH_NPAR 3   /* jobs in flight = sets of everything a job's phases hand to each other (round 6: three -- phase A of job k+2 beside the triangulations of k+1 and phase B of k) */
#define MESH_PUB_SEQ (SC_COUNT + 0)     /* job sequence number: the completion ticket the worker polls */
#define MESH_PUB_TICKS (SC_COUNT + 2)   /* 64-bit: device time of the job, 100 MHz ticks */
#define MESH_N_MARKS 14
#define MESH_PUB_MARKS (SC_COUNT + 8)  /* MESH_N_MARKS x 64-bit: absolute real-time-counter values at the entries of the job's kernels (MESH_MARK) */
#define MESH_PUB_WORDS (SC_COUNT + 8 + 2 * MESH_N_MARKS)
struct MeshJob { const float* d_pts; int n_raw; double cam[3]; int frame_idx; long id; hipEvent_t ready; const unsigned long long* wait_flag = nullptr; unsigned long long wait_seq = 0; };
struct MeshOutSet { int32_t* tri_add; uint8_t* flip_add; int32_t* tri_rem; int32_t* tri_upd; uint8_t* flip_upd; int32_t* smooth_ids; double* smooth_xyz;
                    uint8_t* own_add; uint8_t* own_rem; uint8_t* own_upd; };
struct MeshResult { immesh_mesh_sizes_t sizes; int st_add = 0, st_rem = 0, st_upd = 0;   /* sharded: lengths of the committed lists (sizes = the reported parts) */
                    int rc = 0; std::string err; float ms = 0.f; long id = 0; const float* d_pts = nullptr; int n_raw = 0; };

struct MeshHost {
    int32_t seq = 0;
    int64_t cum[SC_COUNT];
    int32_t* p_a = nullptr;    // add list as sorted triangle indices (input of the adjacency commit)
    int32_t* h_sc = nullptr;   // pinned copy of the per-scan counters of the job being finished (points into h_sc2)
    int32_t* h_sc2[MESH_NPAR] = {};      // pinned + mapped: [0, SC_COUNT) the job's counters, then MESH_PUB_* words written by mesh_publish_kernel
    int32_t* h_sc2_dev[MESH_NPAR] = {};
    int32_t* h_pc = nullptr;
    void* d_sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    void* d_sort_recs = nullptr;   // 16-byte sort records of the chunk sort (phase B: result lists)
    void* d_sort_recs_a = nullptr; // same, phase A (active-voxel list)
    MeshDev mpar[MESH_NPAR];               // per-parity views
    int32_t n_vertices = 0;
    int64_t n_live = 0;
    bool ready = false;
    // asynchronous execution
    hipStream_t stream = nullptr;            // the mesher's own streams: phase A ...
    hipStream_t stream_b = nullptr;          // ... and phase B
    hipEvent_t ev_ready[MESH_WORLD_BUFS] = {};   // (a job's device time comes from mesh_publish_kernel, not from events)
    hipEvent_t ev_c[MESH_NPAR] = {};   // the job's triangulations (third stream) finished
    hipEvent_t ev_a[MESH_NPAR] = {}, ev_b[MESH_NPAR] = {};   // phase A / B of the job of that parity finished
    // world-frame full scans.  The mesher pipelines two jobs (phase A of one over phase B of the other) and takes ~2.4 scan periods per job, so the scan
    // thread writes up to MESH_WORLD_BUFS scans ahead of the oldest running job (buffer = job id mod MESH_WORLD_BUFS) before it has to wait
    float* d_world[MESH_WORLD_BUFS] = {};
    MeshOutSet outs[MESH_NPAR];                      // result lists, double-buffered (job id parity)
    MeshResult res[MESH_NPAR];
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<MeshJob> q;
    long submitted = 0, completed = 0, current = 0;   // job ids start at 1; `current` = job whose results sizes/fetch return
    // service-thread collection (immesh_mesh_collect_*): jobs are handed to the collector strictly in order; with it enabled a submission that would
    // overwrite the result buffers of a job not yet collected (id - 2) waits -- nothing is dropped
    bool collect_on = false;
    double job_ms_sum = 0; long long job_ms_n = 0;
    unsigned long long mark_ring[64][MESH_N_MARKS] = {}; long long mark_n = 0;   // IMMESH_DEBUG_WAITS: phase marks of the last 64 jobs
    long long wait_ring[64] = {}; 
    long long wait_ns = 0, wait_calls = 0;   // IMMESH_DEBUG_WAITS (mesh_next_world_buffer)
    long collected = 0;
    std::mutex fetch_mu;                     // immesh_mesh_fetch's staging block: one fetch at a time, from allocation to the last memcpy
    char* h_fetch = nullptr; size_t h_fetch_bytes = 0;   // immesh_mesh_fetch: pinned staging (the lists land here by DMA, all copies in flight together, then one memcpy each into the caller's pageable buffers)
    hipStream_t stream_fetch = nullptr;      // immesh_mesh_fetch's copies: a stream of their own, so that a service thread can fetch while the scan thread enqueues
    bool stop = false;
    // per-scan parameters + graph replay
    MeshDyn* d_dyn[MESH_NPAR] = {};  // device copies read by the kernels (job parity)
    MeshDyn* h_dyn[MESH_NPAR] = {};  // pinned, device-mapped host copies (read by the first kernel of a scan)
    MeshDyn* h_dyn_dev[MESH_NPAR] = {};   // their device-side addresses
    hipGraphExec_t graph_exec[MESH_NPAR] = {};   // phase A, one per job parity (world buffer / result set pointers differ)
    hipGraphExec_t graph_exec_b[MESH_NPAR][2] = {}; // phase B ([1]: without the triangulations)
    int graph_ncand[MESH_NPAR] = {-1, -1, -1};           // candidate count the graph was captured for (grid sizes, clear sizes)
    bool use_graph = true;
    int room = 0;            // jobs the worker keeps in flight: 0 = two, three while the mesher is behind (mesh_worker_main); IMMESH_MESH_ROOM fixes it
    int split_mode = 2;      // the triangulations on the third stream, the diff at the head of phase B: 0 never (IMMESH_NO_SPLIT / IMMESH_SPLIT=0), 1 always (IMMESH_SPLIT=1), 2 while the mesher is behind (with the third job in flight)
    bool pipeline = true;                    // phase A of scan k+1 may overlap phase B of scan k (IMMESH_NO_PIPELINE turns it off)
    // mesh export scratch (grow-only)
    void *exp_vtx = nullptr, *exp_work = nullptr, *exp_tmp = nullptr;
    size_t exp_vtx_bytes = 0, exp_work_bytes = 0, exp_tmp_bytes = 0;
    int32_t* exp_faces = nullptr;
    int64_t exp_nv = 0, exp_nf = 0;
    // Global_map::smooth_pts on demand (immesh_smooth_pts / immesh_mesh_display_vertices): callable from a third thread (the renderer's) while scans are
    // being meshed.  launch_mu is held by the worker while it ENQUEUES a job and by a query from "both mesher streams idle" to "results on the host": a
    // query sees the map between two jobs, never one half appended
    std::mutex launch_mu;
    hipStream_t stream_q = nullptr;
    void *q_dev = nullptr, *q_exp = nullptr; size_t q_dev_bytes = 0, q_exp_bytes = 0;
    char* q_host = nullptr; size_t q_host_bytes = 0;
    // sharded mesher: all-gather callback + staging
    immesh_allgather_fn allgather = nullptr;
    void* allgather_user = nullptr;
    void* d_xsend = nullptr; void* d_xrecv = nullptr; int32_t* d_xcount = nullptr;   // device staging (cap_list records each) + record counter
    int32_t* d_xcounts = nullptr; void* d_xall = nullptr; size_t xall_bytes = 0;     // RCCL path: every rank's record count / records (grow-only)
    size_t xcap_bytes = 0;
    void* d_xbig = nullptr; size_t xbig_bytes = 0;   // blocks padded to the largest count, when a rank had more records than the first block holds (grow-only)
    std::vector<char> h_xsend, h_xrecv;
    int64_t xbytes_sent = 0, xcalls = 0;     // cumulative exchange volume of this rank (payload bytes, collective calls)
    int64_t x_rounds = 0;                    // cumulative admission exchange rounds (>= 1 per scan; 2 when no dependency chain crosses a brick face twice)
    int fin_state[3] = {0, 0, 0};            // committed list lengths of the job being finished (sharded: sizes carry the reported parts)
    KProf prof;                              // kernels launched by the worker thread
    std::string err;                         // worker-side error text (moved into the MeshResult of the failing job)
};

void launch_mesh_transform(hipStream_t s, const float* raw_xyzi, float* world_xyzi, int n, const double* R, const double* t, const double* extR,
                           const double* extT, const double* rt_dev = nullptr);
void launch_mesh_begin_scan(hipStream_t s, const MeshDev& m, const MeshDyn* h_dyn_dev, unsigned long long ccap);
void launch_mesh_publish(hipStream_t s, const MeshDev& m, int32_t* host_sc);         // last launch of a job: counters, device time, ticket -> pinned memory
void launch_mesh_append_prepare(hipStream_t s, const MeshDev& m, int n_cand, const float* pts);
void launch_mesh_append_resolve(hipStream_t s, const MeshDev& m, int n_cand, const float* pts, int max_iter);
void launch_mesh_append_commit(hipStream_t s, const MeshDev& m, int n_cand, const float* pts);
void launch_mesh_append_flags(hipStream_t s, const MeshDev& m, int n_cand);
void launch_mesh_select_active(hipStream_t s, const MeshDev& m, int n_cand);
void launch_mesh_append_finish(hipStream_t s, const MeshDev& m, const float* pts);   // flags + scan + commit + select + active-voxel order in one launch (n_cand <= 16384)
void launch_mesh_knn(hipStream_t s, const MeshDev& m);
void launch_mesh_export_vertices(hipStream_t s, const MeshDev& m, float* export_vtx, double smooth_factor);
void launch_mesh_query_voxels(hipStream_t s, const MeshDev& m, const int32_t* ids, int n, int n_vertices, int display, int32_t* vox_out);
void launch_mesh_query_smooth(hipStream_t s, const MeshDev& m, const int32_t* vox_list, int n_list, double smooth_factor, double max_dis, double* export_d);
void launch_mesh_query_gather(hipStream_t s, const MeshDev& m, const int32_t* ids, const int32_t* vox, int n, const double* export_d, int display, double* out_d, float* out_f);
void launch_mesh_export_faces(hipStream_t s, const MeshDev& m, int32_t* tri_idx, int32_t* count);
void launch_mesh_export_keys(hipStream_t s, const MeshDev& m, const int32_t* tris, int n, int which, uint32_t* k32, unsigned long long* k64);
void launch_mesh_export_wind(hipStream_t s, const MeshDev& m, const int32_t* tri_sorted, int n, int32_t* faces);
void launch_mesh_delaunay(hipStream_t s, const MeshDev& m);
void launch_mesh_tri64(hipStream_t s, const MeshDev& m);
void launch_mesh_diff64(hipStream_t s, const MeshDev& m);
void launch_mesh_finalize(hipStream_t s, const MeshDev& m);
// exchange blocks of the sharded mesher: every rank contributes ONE fixed-size block per exchange -- 16-byte header {records, aux, -, -} + records -- so an
// exchange is a single all-gather, and the unpack kernels read the counts on the device (cap_rec = records a block holds)
void launch_mesh_cand_pack(hipStream_t s, const MeshDev& m, MeshCdRec* out, int32_t* count, int cap_rec);   // count: [0] records, [1] candidates still undecided here
void launch_mesh_cand_unpack(hipStream_t s, const MeshDev& m, const void* gathered, size_t capb, int cap_rec);
void launch_mesh_pack_smooth(hipStream_t s, const MeshDev& m, MeshSmRec* out, int32_t* count, int cap_rec);
void launch_mesh_unpack_smooth(hipStream_t s, const MeshDev& m, const void* gathered, size_t capb, int cap_rec);
void launch_mesh_pack_marks(hipStream_t s, const MeshDev& m, MeshMkRec* out, int32_t* count, int cap_rec);
void launch_mesh_unpack_marks(hipStream_t s, const MeshDev& m, const void* gathered, size_t capb, int cap_rec);
void launch_mesh_commit_rem(hipStream_t s, const MeshDev& m, const int32_t* tris);
void launch_mesh_commit_add(hipStream_t s, const MeshDev& m, const int32_t* tris_sorted);
void launch_mesh_sort_emit(hipStream_t s, const MeshDev& m, int which, void* recs, int32_t* add_sorted);

// device prefix sum (sort.hip)
size_t exclusive_sum_temp_bytes(int n);
void exclusive_sum_i32(hipStream_t s, void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, int n);
