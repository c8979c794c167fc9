// Host side of the iterated EKF (src/voxel_mapping.cpp:1585-1646): the 18-state algebra that the reference
// keeps serial.  27 meaningful doubles come back from the device per iteration (H^T R^-1 H, H^T R^-1 z); everything here is
// O(18^3) and stays on the CPU next to the ROS node, as in the reference.
#pragma once
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

namespace imh {

struct State {  // StatesGroup, include/common_lib.h:199-288
    double R[9], t[3], vel[3], bg[3], ba[3], g[3], cov[324];
};
inline void load_state(const double* s, State& st) {
    std::memcpy(st.R, s, 72); std::memcpy(st.t, s + 9, 24); std::memcpy(st.vel, s + 12, 24); std::memcpy(st.bg, s + 15, 24);
    std::memcpy(st.ba, s + 18, 24); std::memcpy(st.g, s + 21, 24); std::memcpy(st.cov, s + 24, 324 * 8);
}
inline void store_state(const State& st, double* s) {
    std::memcpy(s, st.R, 72); std::memcpy(s + 9, st.t, 24); std::memcpy(s + 12, st.vel, 24); std::memcpy(s + 15, st.bg, 24);
    std::memcpy(s + 18, st.ba, 24); std::memcpy(s + 21, st.g, 24); std::memcpy(s + 24, st.cov, 324 * 8);
}

inline void mat3_mul(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
    std::memcpy(C, T, sizeof(T));
}

// dense inverse, Gauss-Jordan with partial pivoting (stands in for Eigen's Matrix<double,18,18>::inverse())
inline bool invert(const double* A, double* Ainv, int n) {
    double M[324];   // n <= 18
    std::memcpy(M, A, sizeof(double) * n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ainv[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int col = 0; col < n; col++) {
        int piv = col;
        double best = std::fabs(M[col * n + col]);
        for (int r = col + 1; r < n; r++) { const double v = std::fabs(M[r * n + col]); if (v > best) { best = v; piv = r; } }
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

inline void so3_exp(double v1, double v2, double v3, double* R) {  // include/so3_math.h:71-89
    const double norm = std::sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (norm > 0.00001) {
        const double r[3] = {v1 / norm, v2 / norm, v3 / norm};
        const double K[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
        double KK[9];
        mat3_mul(K, K, KK);
        const double s = std::sin(norm), c1 = 1.0 - std::cos(norm);
        for (int i = 0; i < 9; i++) R[i] = (R[i] + s * K[i]) + c1 * KK[i];
    }
}
inline void so3_log(const double* R, double* out) {  // include/so3_math.h:92-98
    const double tr = R[0] + R[4] + R[8];
    const double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
    const double K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    if (std::fabs(theta) < 0.001) for (int i = 0; i < 3; i++) out[i] = 0.5 * K[i];
    else { const double f = 0.5 * theta / std::sin(theta); for (int i = 0; i < 3; i++) out[i] = f * K[i]; }
}
inline void state_plus(State& s, const double* d) {  // StatesGroup::operator+=, common_lib.h:249-258
    double E[9], Rn[9];
    so3_exp(d[0], d[1], d[2], E);
    mat3_mul(s.R, E, Rn);
    std::memcpy(s.R, Rn, sizeof(Rn));
    for (int k = 0; k < 3; k++) { s.t[k] += d[3 + k]; s.vel[k] += d[6 + k]; s.bg[k] += d[9 + k]; s.ba[k] += d[12 + k]; s.g[k] += d[15 + k]; }
}
inline void state_minus(const State& a, const State& b, double* out) {  // a - b, common_lib.h:260-271
    double Rt[9], rotd[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rt[i * 3 + j] = b.R[j * 3 + i];
    mat3_mul(Rt, a.R, rotd);
    so3_log(rotd, out);
    for (int k = 0; k < 3; k++) { out[3 + k] = a.t[k] - b.t[k]; out[6 + k] = a.vel[k] - b.vel[k]; out[9 + k] = a.bg[k] - b.bg[k]; out[12 + k] = a.ba[k] - b.ba[k]; out[15 + k] = a.g[k] - b.g[k]; }
}

// One Kalman step from the reduced normal equations.  Returns true when the EKF must stop (covariance updated).
struct EkfLoop {
    int rematch_num = 0;
    double G[324];
    double covinv[324];      // state.cov is the prior covariance for every iteration of a scan (overwritten only at stop): invert it once
    bool have_covinv = false;
    EkfLoop() { std::memset(G, 0, sizeof(G)); }
    bool step(const double* HTH /*36*/, const double* HTz /*6*/, const State& prior, State& st, int it, int max_iter) {
        double HTH18[324];
        std::memset(HTH18, 0, sizeof(HTH18));
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) HTH18[r * 18 + c] = HTH[r * 6 + c];
        double S[324], K1[324];
        if (!have_covinv) { invert(st.cov, covinv, 18); have_covinv = true; }
        for (int k = 0; k < 324; k++) S[k] = HTH18[k] + covinv[k];
        invert(S, K1, 18);
        for (int r = 0; r < 18; r++)
            for (int c = 0; c < 6; c++) {
                double s = 0;
                for (int k = 0; k < 6; k++) s += K1[r * 18 + k] * HTH[k * 6 + c];
                G[r * 18 + c] = s;
            }
        double vec[18], sol[18];
        state_minus(prior, st, vec);
        for (int r = 0; r < 18; r++) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 6; k++) { s1 += K1[r * 18 + k] * HTz[k]; s2 += G[r * 18 + k] * vec[k]; }
            sol[r] = (s1 + vec[r]) - s2;
        }
        state_plus(st, sol);
        const double rn = std::sqrt(sol[0] * sol[0] + sol[1] * sol[1] + sol[2] * sol[2]);
        const double tn = std::sqrt(sol[3] * sol[3] + sol[4] * sol[4] + sol[5] * sol[5]);
        const bool converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
        if (converged || ((rematch_num == 0) && (it == (max_iter - 2)))) rematch_num++;
        if (rematch_num >= 2 || (it == max_iter - 1)) {
            double IG[324], nc[324];
            for (int r = 0; r < 18; r++)
                for (int c = 0; c < 18; c++) IG[r * 18 + c] = ((r == c) ? 1.0 : 0.0) - G[r * 18 + c];
            for (int r = 0; r < 18; r++)
                for (int c = 0; c < 18; c++) { double s = 0; for (int k = 0; k < 18; k++) s += IG[r * 18 + k] * st.cov[k * 18 + c]; nc[r * 18 + c] = s; }
            std::memcpy(st.cov, nc, sizeof(nc));
            return true;
        }
        return false;
    }
};

// ImuProcess::Forward_without_imu (src/IMU_Processing.cpp:486-553): constant-velocity prior between two scans; bias_g plays the role of
// the angular rate ("omega in constant model").  Host-side, 18x18 algebra only -- the step immediately before lio_state_estimation.
inline void forward_without_imu(const State& in, double dt, double cov_gyr, double cov_acc, State& out) {
    double F[324], W[324], FC[324];
    for (int i = 0; i < 324; i++) { F[i] = (i % 19 == 0) ? 1.0 : 0.0; W[i] = 0.0; }
    double E[9];
    so3_exp(-in.bg[0] * dt, -in.bg[1] * dt, -in.bg[2] * dt, E);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F[r * 18 + c] = E[r * 3 + c];
    for (int k = 0; k < 3; k++) { F[k * 18 + 9 + k] = dt; F[(3 + k) * 18 + 6 + k] = dt; W[(9 + k) * 18 + 9 + k] = cov_gyr * dt * dt; W[(6 + k) * 18 + 6 + k] = cov_acc * dt * dt; }
    for (int r = 0; r < 18; r++)
        for (int c = 0; c < 18; c++) { double s = 0; for (int k = 0; k < 18; k++) s += F[r * 18 + k] * in.cov[k * 18 + c]; FC[r * 18 + c] = s; }
    out = in;
    for (int r = 0; r < 18; r++)
        for (int c = 0; c < 18; c++) { double s = 0; for (int k = 0; k < 18; k++) s += FC[r * 18 + k] * F[c * 18 + k]; out.cov[r * 18 + c] = s + W[r * 18 + c]; }
    double Ep[9], Rn[9];
    so3_exp(in.bg[0] * dt, in.bg[1] * dt, in.bg[2] * dt, Ep);
    mat3_mul(in.R, Ep, Rn);
    std::memcpy(out.R, Rn, sizeof(Rn));
    for (int k = 0; k < 3; k++) out.t[k] = in.t[k] + in.vel[k] * dt;
}

}  // namespace imh
