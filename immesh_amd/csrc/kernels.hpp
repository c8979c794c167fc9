// Host <-> kernel interface structs and launcher prototypes (shared by the .hip translation units and the C++ host layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct RegMapDev;

struct ScanParams {          // everything a per-point kernel needs about the current state estimate
    double R[9], t[3];       // state.rot_end / pos_end
    double extR[9], extT[3]; // m_extR / m_extT
    double RextR[9];         // R * extR
    double rot_var[9], t_var[9];  // state.cov blocks (0:3,0:3) and (3:6,3:6)
    double dvar_beam;        // sin(DEG2RAD(beam_err))^2, host-computed
    double dvar_calib;       // same for CALIB_ANGLE_COV (include/common_lib.h:41)
    double sigma_num;
    float dept_err;
    int calib_laser;
    unsigned long long* dbg;  // optional phase timers (IMMESH_DEBUG), nullptr = off
};

struct PlaneRecDev {  // layout == immesh_plane_rec (include/immesh_c_api.h)
    long long key[3];
    int layer, path, is_plane, n_points, update_enable, new_points;
    float radius, min_eig, d, pad;
    double center[3], normal[3], plane_var[36];
};

#define REG_DBG_WORDS (64 + 16384 * 8 + 8 * 512 * 8)   /* IMMESH_DEBUG buffer (reg_kernels.hip DBG_*): counters + per-wavefront trace records */
#define STATS_WORDS (16 + 64 * 16)   /* refit counters: [0] refits [1] refit points, then 64 shards of the fused replay kernel (one 128-byte line each) */
#define RES_NV_HOST 48
#define RES_NR_HOST 32

// ---- device-resident iterated EKF (Voxel_mapping::lio_state_estimation, src/voxel_mapping.cpp:1585-1650) ---------------------
// The state of the scan being registered stays on the device between the residual passes: the last block of a pass runs the 18-state
// update in-kernel and leaves the next pass's parameters behind, so a whole scan is enqueued without a host round trip.
struct RegState {
    ScanParams sp;            // what the per-point kernels of the NEXT pass (and the map update / full-scan transform) read
    double st[24];            // current iterate: R[9] t[3] vel[3] bias_g[3] bias_a[3] gravity[3]
    double prior[24];         // state_propagat
    double p11inv[36];        // inverse of the pose block P11 of the prior covariance (state.cov is the prior for every iteration of a scan)
    double tmat[72];          // P21 P11^-1 (12 x 6): with X = (H^T R^-1 H + P11^-1)^-1 the gain's six columns are [X; tmat X]
    double tot[4];            // cumulative over the passes: n_plane_tests, n_extra_probe, passes run, n_match
    int rematch, done, started, pad1;   // started: low bits of the ticket of the last residual_persistent_kernel launch whose workgroups are running (a scheduling hint: ds_gate_kernel)
};
// argument block of one residual pass (kernel arguments are limited to 4 KB: RegMapDev + this + a dozen pointers stay below)
#define REG_MODE_HOST 0       /* legacy: parameters by value, the 48 sums + ticket go to pinned host memory, the host runs the EKF step */
#define REG_MODE_FUSED 1      /* residual_persistent_kernel: every pass of the scan + the 18-state update in one launch; mat = the prior covariance */
#define REG_MODE_SUMS 3       /* parameters from RegState (it > 0) or by value (it == 0); the 48 sums go to device memory for an in-stream all-reduce, ekf_step_kernel follows */
struct RegIterArgs {
    int mode, it, max_iter, pad;   // pad: residual_persistent_kernel -- bit 0 = run the previous map update's deferred tail in the prologue, bit 1 = test hook: abort in the first gather
    ScanParams sp;
    double st[24], prior[24];
    double mat[324];          // first pass: [0,36) P11^-1, [36,108) P21 P11^-1; later passes: the prior covariance
};
#define REG_OUT_DOUBLES 360   /* pinned result block: [0,348) posterior state, 348 passes run, 349 n_match (last pass), 350 sum|dis| (last pass), 351 plane tests, 352 extra probes, 353 n_match summed over the passes, 359 ticket */

void launch_residual(hipStream_t s, const RegMapDev& m, const RegIterArgs& a, RegState* rs, const float* pts, int n, double* partials, unsigned int* done_counter,
                     double* out48, double* reg_out, double ticket, int8_t* o_match, int32_t* o_node, float* o_dis, double* o_rinv, double* o_normal);
// all passes of a scan + the 18-state update as one resident grid (residual_persistent_kernel); a.mat = the prior covariance.
// slots / slots_next: the two parities of the block-partial buffer, RP_SLOT_DOUBLES each, filled with RP_SLOT_SENTINEL before first use
#define RP_SLOT_DOUBLES (64 * 128 * 32)
#define RP_SLOT_SENTINEL 0x7FF8DEADBEEF0001ull
// The map update's preparation (point_var: world points, covariances, root voxels, per-voxel lists) and the transform of the full scan for the mesher
// as the EPILOGUE of the registration launch: every block of the resident grid holds the posterior when the loop stops, so neither a kernel boundary
// nor a launch is needed in between (immesh_process_scan; the other entry points keep point_var_kernel).  `m` must already be the map of THIS
// update (upd_seq advanced).  The launch queued behind it (replay_fused_kernel, launch_replay_lists' flag arguments) stores a sequence number to a
// device flag (the mesher's first kernel waits for it instead of an event record -- a barrier packet -- on the pose chain) and to a pinned one (the
// host's "input clouds consumed" fence): behind the kernel boundary, so the epilogue itself needs no fence.
struct RpEpilogue {
    int enabled, n_raw;
    double* pt_data; unsigned long long* sort_key; uint32_t* slot_out; int32_t* pt_next;
    const float* raw; float* world;          // xyzI in / out (nullptr: no mesher)
};
#define RP_TAIL_WORD (63 * 128 * 32)   /* slot word the deferred-tail workgroup publishes (passes 62, 63 are never run: max_iter < 62) */
#define RP_ABORT_WORD (62 * 128 * 32)  /* slot word a block publishes when its gather runs out of patience (the grid did not become resident) */
void launch_residual_persistent(hipStream_t s, const RegMapDev& m, const RegIterArgs& a, RegState* rs, const float* pts, int n, double* slots, double* slots_next,
                                int32_t* host_counters, double* reg_out, double ticket, int8_t* o_match, int32_t* o_node, float* o_dis, double* o_rinv, double* o_normal,
                                const RpEpilogue& ep, int max_blocks);
int residual_persistent_resident_blocks(int device);
// the 18-state update as its own launch (sharded map with an in-stream all-reduce of the 48 sums between the residual pass and the update)
void launch_ekf_step(hipStream_t s, const RegIterArgs& a, RegState* rs, const double* sums48, double* reg_out, double ticket);
// spd != nullptr: the parameters are read from device memory (RegState::sp of the scan just registered) instead of `sp`
void launch_point_var(hipStream_t s, const RegMapDev& m, const ScanParams& sp, const ScanParams* spd, const float* pts, int n, int stride, int mode, double* pt_data,
                      unsigned long long* sort_key, uint32_t* slot, int32_t* pt_next, const float* raw_xyzi = nullptr, float* world_xyzi = nullptr, int n_raw = 0);
// (raw_xyzi != nullptr: the same launch also transforms the full xyzI scan into the world frame for the mesher)
void launch_replay_lists(hipStream_t s, const RegMapDev& m, const int32_t* pt_next, const unsigned long long* sort_key, const double* pt_data, int n,
                         int64_t* stats, int32_t* host_counters, int32_t* big_idx, int32_t* big_order, uint32_t* general_list, unsigned long long* dbg = nullptr, bool with_tail = true, unsigned long long* flag_dev = nullptr, unsigned long long* flag_host = nullptr, unsigned long long flag_seq = 0);
// the tail of a map update whose launch was deferred (launch_replay_lists with_tail = false; RegIterArgs::pad of the next residual_persistent_kernel)
void launch_map_update_tail(hipStream_t s, const RegMapDev& m, int32_t* host_counters);
void launch_segment_heads(hipStream_t s, const uint32_t* sorted_slot, int n, int32_t* seg_start, int32_t* nseg);
void launch_replay(hipStream_t s, const RegMapDev& m, const uint32_t* sorted_slot, const int32_t* sorted_idx, const double* pt_data, int n,
                   const int32_t* seg_start, const int32_t* nseg, int max_segments, int mode, int64_t* stats);
void launch_dump_planes(hipStream_t s, const RegMapDev& m, PlaneRecDev* out, long long cap, unsigned long long* count);
void launch_fill_u64(hipStream_t s, unsigned long long* p, unsigned long long v, size_t n);
void launch_iota(hipStream_t s, int32_t* p, int n);
void launch_unpack_strided(hipStream_t s, const void* src, int n, int stride_bytes, int intensity_off_bytes, float* out);   // intensity_off_bytes < 0: xyz only

// VoxelGrid down-sampling (ds_kernels.hip)
void launch_ds_minmax(hipStream_t s, const float* pts, int n, int stride, float inv, int* mm);
void launch_ds_index(hipStream_t s, const float* pts, int n, int stride, float inv, const int* mm, unsigned long long* keys, int32_t* vals);
void launch_ds_heads(hipStream_t s, const unsigned long long* keys_sorted, int n, int32_t* flags);
void launch_ds_centroid(hipStream_t s, const float* pts, int n, int stride, const unsigned long long* keys_sorted, const int32_t* idx_sorted, const int32_t* rank,
                        float* out, int32_t* n_out);
void launch_decode_livox_count(hipStream_t s, const uint8_t* w, int n, int n_scans, int32_t* counted);
void launch_decode_livox_keep(hipStream_t s, const uint8_t* w, int n, int n_scans, int filter, double blind_sqr, const int32_t* valid_excl, int32_t* keep);
void launch_decode_livox_emit(hipStream_t s, const uint8_t* w, int n, const int32_t* keep, const int32_t* pos, float* out, int32_t* n_out);
void launch_decode_velodyne_keep(hipStream_t s, const uint8_t* d, int n, int step, int ox, int oy, int oz, int n_scans, int32_t* keep);
void launch_decode_velodyne_emit(hipStream_t s, const uint8_t* d, int n, int step, int ox, int oy, int oz, int oi, const int32_t* keep, const int32_t* pos, float* out,
                                 int32_t* n_out);
void launch_undistort_keys(hipStream_t s, const float* pts5, int n, uint32_t* key, int32_t* idx);
void launch_undistort(hipStream_t s, const float* pts5, const int32_t* order, int n, const double* poses, int n_poses, const double* fe, float* out_xyzi);
// the VoxelGrid's per-cloud parameters (pinned, device-mapped host memory: the kernels' arguments stay the same from cloud to cloud -> one hipGraph)
struct DsDyn { const float* pts; float* out; int32_t n, stride; float inv; int32_t pad; const int32_t* gate_word; int32_t gate_val, gate_pad; };   // pad: the job's ticket (ds_publish_kernel hands it to the host); gate_word != nullptr: ds_gate_kernel holds the sequence back until *gate_word has reached gate_val
void launch_ds_hash_pipeline(hipStream_t s, const DsDyn* dyn, void* tab, unsigned long long cap, int32_t* pt_slot, int32_t* leaf_slot,
                             unsigned long long* keys_sorted, int32_t* slots_sorted, float* pool4, int32_t* big_list, int32_t* info, int32_t* n_out);
void launch_ds_publish(hipStream_t s, int32_t* info, int32_t* host_info, const DsDyn* dyn);
void launch_ds_gate(hipStream_t s, const DsDyn* dyn);
void launch_ds_table_reset(hipStream_t s, void* tab, unsigned long long cap);
void launch_ds_expand_xyzi(hipStream_t s, const float* xyz, int n, float* out_xyzi);
size_t exclusive_sum_temp_bytes(int n);
void exclusive_sum_i32(hipStream_t s, void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, int n);

// stable LSD radix sort of (key,value) pairs, device-resident (sort.hip)
size_t sort_pairs_u64_temp_bytes(int n);
size_t sort_pairs_u32_temp_bytes(int n);
void sort_pairs_u64(hipStream_t s, void* temp, size_t temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                    const int32_t* vals_in, int32_t* vals_out, int n, int end_bit = 64);
void sort_pairs_u32(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in,
                    int32_t* vals_out, int n, int end_bit);
