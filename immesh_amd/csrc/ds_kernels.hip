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
