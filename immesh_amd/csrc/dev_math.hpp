// Device-side small dense math for the ImMesh hot path (gfx950).  Compiled with -ffp-contract=off: the reference
// build has no FMA contraction (CMakeLists.txt:14) and float/double gates must round exactly as on the CPU.
// Everything is plain IEEE double/float arithmetic in registers; no libraries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IMD __device__ __forceinline__

namespace imd {

IMD void m3_mul(const double* A, const double* B, double* C) {  // C = A*B (C may not alias)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
IMD void m3_mul_bt(const double* A, const double* B, double* C) {  // C = A*B^T
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3 + 0] * B[j * 3 + 0] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}
IMD void m3_vec(const double* A, const double* v, double* o) {
    const double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    const double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    const double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
IMD void m3t_vec(const double* A, const double* v, double* o) {
    const double t0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    const double t1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    const double t2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
IMD void skew(const double* v, double* K) {  // SKEW_SYM_MATRX, include/so3_math.h:9
    K[0] = 0.0; K[1] = -v[2]; K[2] = v[1];
    K[3] = v[2]; K[4] = 0.0; K[5] = -v[0];
    K[6] = -v[1]; K[7] = v[0]; K[8] = 0.0;
}
IMD void normalize3(double* v) {
    const double z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (z > 0) { const double n = sqrt(z); v[0] /= n; v[1] /= n; v[2] /= n; }
}
IMD void cross3(const double* a, const double* b, double* o) {
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
IMD void m3_sandwich(const double* A, const double* V, double* S) {  // (A*V)*A^T
    double T[9];
    m3_mul(A, V, T);
    m3_mul_bt(T, A, S);
}

// calcBodyVar, src/voxel_mapping.cpp:1221-1241.  dvar = sin(DEG2RAD(degree_inc))^2 is a per-config constant computed
// on the host (same libm as the CPU path).  Mutates pb[2] (0 -> 1e-4) exactly like the reference.
IMD void calc_body_var(double* pb, const float range_inc, const double dvar, double* var) {
    if (pb[2] == 0) pb[2] = 0.0001;
    const float range = (float)sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
    const float range_var = range_inc * range_inc;
    double dir[3] = {pb[0], pb[1], pb[2]};
    normalize3(dir);
    double dhat[9];
    skew(dir, dhat);
    double b1[3] = {1.0, 1.0, -(dir[0] + dir[1]) / dir[2]};
    normalize3(b1);
    double b2[3];
    cross3(b1, dir, b2);
    normalize3(b2);
    double A[6];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double r0 = (double)range * dhat[i * 3 + 0], r1 = (double)range * dhat[i * 3 + 1], r2 = (double)range * dhat[i * 3 + 2];
        A[i * 2 + 0] = r0 * b1[0] + r1 * b1[1] + r2 * b1[2];
        A[i * 2 + 1] = r0 * b2[0] + r1 * b2[1] + r2 * b2[2];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double t1 = (dir[i] * (double)range_var) * dir[j];
            const double t2 = (A[i * 2 + 0] * dvar) * A[j * 2 + 0] + (A[i * 2 + 1] * dvar) * A[j * 2 + 1];
            var[i * 3 + j] = t1 + t2;
        }
}

// Root-voxel key quantisation (src/voxel_mapping.cpp:118-127): double quotient -> float, -1 if negative, truncate.
IMD int64_t key_axis(const double q) {
    float loc = (float)q;
    if (loc < 0) loc = (float)((double)loc - 1.0);
    return (int64_t)loc;
}
IMD float loc_axis(const double q) {  // the float `loc_xyz` itself (needed by the near-voxel quirk)
    float loc = (float)q;
    if (loc < 0) loc = (float)((double)loc - 1.0);
    return loc;
}
#define IM_KEY_BIAS (1 << 20)
#define IM_KEY_MASK ((1ull << 21) - 1)
#define IM_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
IMD uint64_t pack_key(int64_t x, int64_t y, int64_t z) {
    return ((uint64_t)(x + IM_KEY_BIAS) & IM_KEY_MASK) | (((uint64_t)(y + IM_KEY_BIAS) & IM_KEY_MASK) << 21) | (((uint64_t)(z + IM_KEY_BIAS) & IM_KEY_MASK) << 42);
}
IMD uint64_t hash64(uint64_t k) {  // splitmix64 finaliser
    k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
    k ^= k >> 27; k *= 0x94d049bb133111ebull;
    k ^= k >> 31;
    return k;
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi (same sweep order / thresholds as the CPU checker so that the two
// agree to the last bits).  a: row-major symmetric; evals in diagonal-position order; V columns = eigenvectors.
IMD void sym3_eigen_jacobi(const double* Ain, double* evals, double* V) {
    double a00 = Ain[0], a01 = Ain[1], a02 = Ain[2], a11 = Ain[4], a12 = Ain[5], a22 = Ain[8];
    double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 64; sweep++) {
        bool rotated = false;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            // (p,q,r): (0,1,2), (0,2,1), (1,2,0)
            double app, aqq, apq, arp, arq;
            if (e == 0) { app = a00; aqq = a11; apq = a01; arp = a02; arq = a12; }
            else if (e == 1) { app = a00; aqq = a22; apq = a02; arp = a01; arq = a12; }
            else { app = a11; aqq = a22; apq = a12; arp = a01; arq = a02; }
            if (apq == 0.0) continue;
            double napp, naqq, nrp, nrq;
            bool rot;
            if (fabs(apq) <= 1e-300 || (fabs(app) + fabs(apq) == fabs(app) && fabs(aqq) + fabs(apq) == fabs(aqq))) {
                napp = app; naqq = aqq; nrp = arp; nrq = arq; rot = false;
            } else {
                rot = true; rotated = true;
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                napp = app - t * apq; naqq = aqq + t * apq;
                nrp = c * arp - s * arq; nrq = s * arp + c * arq;
                const int p = (e == 2) ? 1 : 0, q = (e == 0) ? 1 : 2;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const double vip = v[i * 3 + p], viq = v[i * 3 + q];
                    v[i * 3 + p] = c * vip - s * viq;
                    v[i * 3 + q] = s * vip + c * viq;
                }
            }
            (void)rot;
            if (e == 0) { a00 = napp; a11 = naqq; a01 = 0.0; a02 = nrp; a12 = nrq; }
            else if (e == 1) { a00 = napp; a22 = naqq; a02 = 0.0; a01 = nrp; a12 = nrq; }
            else { a11 = napp; a22 = naqq; a12 = 0.0; a01 = nrp; a02 = nrq; }
        }
        if (!rotated) break;
    }
    evals[0] = a00; evals[1] = a11; evals[2] = a22;
#pragma unroll
    for (int i = 0; i < 9; i++) V[i] = v[i];
}

// ---- the same cyclic Jacobi for OctoTree::init_plane on the device (reg_kernels.hip: the plane fit is the unit of the map update) ----------
// Same sweep order, same rotation per step, same convergence tests as sym3_eigen_jacobi -- so the eigenvalue that ends up at each diagonal
// position and the sign of every eigenvector column are the checker's -- but the three IEEE divides and two IEEE square roots of a rotation
// (~430 cycles of a lone wavefront's dependent chain, 11 k cycles per decomposition) become one reciprocal and two reciprocal square roots by
// v_rcp_f64 / v_rsq_f64 + two Newton steps (no scaling / fix-up: the operands are covariances, far from the ends of the exponent range):
//   theta = (aqq - app) / (2 apq),  t = sgn(theta) / (|theta| + sqrt(theta^2 + 1))   ==   t = s |apq| / (|h| + sqrt(h^2 + apq^2)),  h = (aqq - app) / 2
// Differences to the IEEE version are a few ulp per rotation (parity bar: 1e-5); NOT for the mesher's PCA, whose axes feed bit-exact predicates.
IMD double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
IMD double fast_rsqrt(double x) {   // x > 0
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = __builtin_fma(__builtin_fma(-hx * y, y, 0.5), y, y);
    y = __builtin_fma(__builtin_fma(-hx * y, y, 0.5), y, y);
    return y;
}
IMD void sym3_eigen_jacobi_fast(const double* Ain, double* evals, double* V) {
    double a00 = Ain[0], a01 = Ain[1], a02 = Ain[2], a11 = Ain[4], a12 = Ain[5], a22 = Ain[8];
    double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 64; sweep++) {
        bool rotated = false;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            double app, aqq, apq, arp, arq;
            if (e == 0) { app = a00; aqq = a11; apq = a01; arp = a02; arq = a12; }
            else if (e == 1) { app = a00; aqq = a22; apq = a02; arp = a01; arq = a12; }
            else { app = a11; aqq = a22; apq = a12; arp = a01; arq = a02; }
            if (apq == 0.0) continue;
            double napp = app, naqq = aqq, nrp = arp, nrq = arq;
            if (!(fabs(apq) <= 1e-300 || (fabs(app) + fabs(apq) == fabs(app) && fabs(aqq) + fabs(apq) == fabs(aqq)))) {
                rotated = true;
                const double h = 0.5 * (aqq - app);
                const double ah = fabs(h), aq = fabs(apq);
                const double r2 = __builtin_fma(h, h, apq * apq);
                const double root = r2 * fast_rsqrt(r2);                      // sqrt(h^2 + apq^2) > 0
                double t = aq * fast_rcp(ah + root);
                if (h != 0.0 && ((h < 0.0) != (apq < 0.0))) t = -t;           // sign of theta; theta == +-0 counts as >= 0
                const double c = fast_rsqrt(__builtin_fma(t, t, 1.0));
                const double s = t * c;
                napp = app - t * apq; naqq = aqq + t * apq;
                nrp = c * arp - s * arq; nrq = s * arp + c * arq;
                const int p = (e == 2) ? 1 : 0, q = (e == 0) ? 1 : 2;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const double vip = v[i * 3 + p], viq = v[i * 3 + q];
                    v[i * 3 + p] = c * vip - s * viq;
                    v[i * 3 + q] = s * vip + c * viq;
                }
            }
            if (e == 0) { a00 = napp; a11 = naqq; a01 = 0.0; a02 = nrp; a12 = nrq; }
            else if (e == 1) { a00 = napp; a22 = naqq; a02 = 0.0; a01 = nrp; a12 = nrq; }
            else { a11 = napp; a22 = naqq; a12 = 0.0; a01 = nrp; a02 = nrq; }
        }
        if (!rotated) break;
    }
    evals[0] = a00; evals[1] = a11; evals[2] = a22;
#pragma unroll
    for (int i = 0; i < 9; i++) V[i] = v[i];
}

// wave64 all-reduce (sum) of a double via xor shuffles; every lane gets the total.  Fixed butterfly order => deterministic.
IMD double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
// The same all-reduce without the LDS crossbar: ds_bpermute steps are ~100-cycle round trips, and a plane fit chains 24 of these reductions
// (9 moments, 15 plane-covariance terms) -- two thirds of its time.  Within a row of 16 lanes the butterfly runs on DPP moves (quad_perm
// xor 1 / xor 2, row_half_mirror, row_mirror: VALU latency only); the four row sums are read with v_readlane and added in row order.
template <int CTRL>
IMD double dpp_d(const double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
IMD double wave_sum_fast(double x) {
    x += dpp_d<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_d<0x4E>(x);    // quad_perm [2,3,0,1]
    x += dpp_d<0x141>(x);   // row_half_mirror
    x += dpp_d<0x140>(x);   // row_mirror
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 0), __builtin_amdgcn_readlane(__double2loint(x), 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 16), __builtin_amdgcn_readlane(__double2loint(x), 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 32), __builtin_amdgcn_readlane(__double2loint(x), 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 48), __builtin_amdgcn_readlane(__double2loint(x), 48));
    return (r0 + r1) + (r2 + r3);
}
IMD int wave_sum_i(int x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

}  // namespace imd
