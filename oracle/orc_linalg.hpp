// ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
// Tiny dependency-free dense linear algebra used by the CPU restatement of ImMesh's hot path.
// Stands in for the Eigen3 calls the reference makes (Eigen is NOT in /root/reference and not in this
// image; version unpinned by CMakeLists.txt:56):
//   Eigen::EigenSolver<Matrix3d>            src/voxel_loc.cpp:62        -> sym3_eigen_jacobi (unsorted)
//   Eigen::SelfAdjointEigenSolver<Matrix3d> src/meshing/mesh_rec_geometry.cpp:199 -> sym3_eigen_jacobi + ascending sort
//   Matrix<double,18,18>::inverse()         src/voxel_mapping.cpp:1588  -> inv_gauss_jordan
// All arithmetic is plain IEEE double, compiled with -ffp-contract=off (the reference build has no FMA).
#pragma once
#include <cmath>
#include <cstring>
#include <cstdint>
#include <vector>
#include <algorithm>

namespace orc {

// ---- 3-vectors / 3x3 row-major -------------------------------------------------------------------
inline void m3_mul(const double* A, const double* B, double* C) {  // C = A*B, inner sum in k order
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
    std::memcpy(C, T, sizeof(T));
}
inline void m3_mul_bt(const double* A, const double* B, double* C) {  // C = A*B^T
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3 + 0] * B[j * 3 + 0] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
    std::memcpy(C, T, sizeof(T));
}
inline void m3_vec(const double* A, const double* v, double* o) {
    double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
inline void m3t_vec(const double* A, const double* v, double* o) {  // A^T v
    double t0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    double t1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    double t2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
inline void skew(const double* v, double* K) {  // SKEW_SYM_MATRX, include/so3_math.h:9
    K[0] = 0.0; K[1] = -v[2]; K[2] = v[1];
    K[3] = v[2]; K[4] = 0.0; K[5] = -v[0];
    K[6] = -v[1]; K[7] = v[0]; K[8] = 0.0;
}
inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross3(const double* a, const double* b, double* o) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
inline void normalize3(double* v) {  // Eigen normalize(): v /= sqrt(squaredNorm) when > 0
    double z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (z > 0) { double n = std::sqrt(z); v[0] /= n; v[1] /= n; v[2] /= n; }
}
// S = A * V * A^T  (3x3), evaluated (A*V)*A^T
inline void m3_sandwich(const double* A, const double* V, double* S) {
    double T[9];
    m3_mul(A, V, T);
    m3_mul_bt(T, A, S);
}

// ---- symmetric 3x3 eigen-decomposition: cyclic Jacobi ------------------------------------------
// evals[k] with eigenvector column k in V (row-major V[i*3+k]); NOT sorted (position order), V orthonormal.
inline void sym3_eigen_jacobi(const double* Ain, double* evals, double* V) {
    double a[9];
    std::memcpy(a, Ain, sizeof(a));
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 64; sweep++) {
        bool rotated = false;
        for (int e = 0; e < 3; e++) {
            const int p = PQ[e][0], q = PQ[e][1];
            const double apq = a[p * 3 + q];
            if (apq == 0.0) continue;
            const double app = a[p * 3 + p], aqq = a[q * 3 + q];
            // negligible off-diagonal: annihilate without rotation
            if (std::fabs(apq) <= 1e-300 || (std::fabs(app) + std::fabs(apq) == std::fabs(app) && std::fabs(aqq) + std::fabs(apq) == std::fabs(aqq))) {
                a[p * 3 + q] = 0.0; a[q * 3 + p] = 0.0;
                continue;
            }
            rotated = true;
            const double theta = (aqq - app) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0);
            const double s = t * c;
            const int r = 3 - p - q;
            const double arp = a[r * 3 + p], arq = a[r * 3 + q];
            a[p * 3 + p] = app - t * apq;
            a[q * 3 + q] = aqq + t * apq;
            a[p * 3 + q] = 0.0; a[q * 3 + p] = 0.0;
            const double nrp = c * arp - s * arq, nrq = s * arp + c * arq;
            a[r * 3 + p] = nrp; a[p * 3 + r] = nrp;
            a[r * 3 + q] = nrq; a[q * 3 + r] = nrq;
            for (int i = 0; i < 3; i++) {
                const double vip = V[i * 3 + p], viq = V[i * 3 + q];
                V[i * 3 + p] = c * vip - s * viq;
                V[i * 3 + q] = s * vip + c * viq;
            }
        }
        if (!rotated) break;
    }
    evals[0] = a[0]; evals[1] = a[4]; evals[2] = a[8];
}

// ---- generic N x N inverse (Gauss-Jordan, partial pivoting); returns false if singular ------------
inline bool inv_gauss_jordan(const double* A, double* Ainv, int n) {
    std::vector<double> M(A, A + n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ainv[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int col = 0; col < n; col++) {
        int piv = col;
        double best = std::fabs(M[col * n + col]);
        for (int r = col + 1; r < n; r++) {
            double v = std::fabs(M[r * n + col]);
            if (v > best) { best = v; piv = r; }
        }
        if (best == 0.0) return false;
        if (piv != col)
            for (int j = 0; j < n; j++) { std::swap(M[piv * n + j], M[col * n + j]); std::swap(Ainv[piv * n + j], Ainv[col * n + j]); }
        const double d = M[col * n + col];
        for (int j = 0; j < n; j++) { M[col * n + j] /= d; Ainv[col * n + j] /= d; }
        for (int r = 0; r < n; r++) {
            if (r == col) continue;
            const double f = M[r * n + col];
            if (f == 0.0) continue;
            for (int j = 0; j < n; j++) { M[r * n + j] -= f * M[col * n + j]; Ainv[r * n + j] -= f * Ainv[col * n + j]; }
        }
    }
    return true;
}

// ---- SO(3) exp / log, include/so3_math.h:71-98 ------------------------------------------------------
inline void so3_exp(double v1, double v2, double v3, double* R) {
    double norm = std::sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (norm > 0.00001) {
        double r[3] = {v1 / norm, v2 / norm, v3 / norm};
        double K[9], KK[9];
        skew(r, K);
        m3_mul(K, K, KK);
        const double s = std::sin(norm), c1 = 1.0 - std::cos(norm);
        for (int i = 0; i < 9; i++) R[i] = (R[i] + s * K[i]) + c1 * KK[i];
    }
}
inline void so3_log(const double* R, double* out) {
    const double tr = R[0] + R[4] + R[8];
    const double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
    const double K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    if (std::fabs(theta) < 0.001) { for (int i = 0; i < 3; i++) out[i] = 0.5 * K[i]; }
    else { const double f = 0.5 * theta / std::sin(theta); for (int i = 0; i < 3; i++) out[i] = f * K[i]; }
}

}  // namespace orc
