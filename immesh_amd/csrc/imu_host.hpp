// Host side of ImuProcess::UndistortPcl (src/IMU_Processing.cpp:755-958): forward propagation of the 18-state + covariance over the IMU samples
// of one LiDAR package, producing the table of IMU-rate poses that the device kernel (undistort_kernel, ds_kernels.hip) compensates points with.
// O(n_imu * 18^3) on the host, next to the EKF algebra, as in the reference.
#pragma once
#include "ekf_host.hpp"
#include "../../include/immesh_c_api.h"
#include <algorithm>

namespace imh {

struct ImuPose { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; };   // Pose6D (include/common_lib.h), 23 doubles

inline void rot_exp_rate(const double w[3], double dt, double* R) {   // Exp(ang_vel, dt), include/so3_math.h:30-50
    const double nrm = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (nrm > 0.0000001) {
        const double r[3] = {w[0] / nrm, w[1] / nrm, w[2] / nrm};
        const double K[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
        double KK[9];
        mat3_mul(K, K, KK);
        const double ang = nrm * dt, s = std::sin(ang), c1 = 1.0 - std::cos(ang);
        for (int i = 0; i < 9; i++) R[i] = (R[i] + s * K[i]) + c1 * KK[i];
    }
}

// Propagates `st` to the scan end, updates the carried ImuProcess members, fills `poses` (IMUpose).  pcl_end_curv_ms = curvature of the package's
// last point in arrival order (:785).
inline void imu_forward(const immesh_imu_sample* imu, int n_imu, double lidar_beg_time, double* last_update_time, float pcl_end_curv_ms,
                        immesh_imu_ctx& ic, State& st, std::vector<ImuPose>& poses) {
    const double G = 9.81;   // G_m_s2, include/common_lib.h:35
    std::vector<immesh_imu_sample> v;
    v.reserve((size_t)n_imu + 1);
    v.push_back(ic.last_imu);
    v.insert(v.end(), imu, imu + n_imu);
    const double imu_end = v.back().t;
    const double pcl_beg = std::max(lidar_beg_time, *last_update_time);
    const double pcl_end = lidar_beg_time + (double)pcl_end_curv_ms / double(1000);
    *last_update_time = pcl_end;
    auto snapshot = [&](double t, const double* a, const double* g, const double* vel, const double* pos, const double* R) {
        ImuPose p;
        p.offset_time = t;
        std::memcpy(p.acc, a, 24); std::memcpy(p.gyr, g, 24); std::memcpy(p.vel, vel, 24); std::memcpy(p.pos, pos, 24); std::memcpy(p.rot, R, 72);
        poses.push_back(p);
    };
    poses.clear();
    snapshot(0.0, ic.acc_s_last, ic.angvel_last, st.vel, st.t, st.R);
    double acc_w[3], gyr[3], acc_b[3], vel[3], pos[3], R[9];
    std::memcpy(acc_w, ic.acc_s_last, 24); std::memcpy(gyr, ic.angvel_last, 24);
    std::memcpy(vel, st.vel, 24); std::memcpy(pos, st.t, 24); std::memcpy(R, st.R, 72);
    double dt = 0;
    std::vector<double> F(324), FC(324), NC(324);
    for (size_t k = 1; k < v.size(); k++) {
        const immesh_imu_sample& h = v[k - 1];
        const immesh_imu_sample& t = v[k];
        if (t.t < ic.last_lidar_end_time) continue;
        for (int a = 0; a < 3; a++) {
            gyr[a] = 0.5 * (h.gyr[a] + t.gyr[a]) - st.bg[a];
            acc_b[a] = 0.5 * (h.acc[a] + t.acc[a]) * G / ic.mean_acc_norm - st.ba[a];
        }
        dt = (h.t < ic.last_lidar_end_time) ? t.t - ic.last_lidar_end_time : t.t - h.t;
        double Ef[9], Eb[9], RA[9], RD[9], Q[9];
        rot_exp_rate(gyr, dt, Ef);
        rot_exp_rate(gyr, -dt, Eb);
        const double S[9] = {0.0, -acc_b[2], acc_b[1], acc_b[2], 0.0, -acc_b[0], -acc_b[1], acc_b[0], 0.0};
        mat3_mul(R, S, RA);
        std::fill(F.begin(), F.end(), 0.0);
        for (int i = 0; i < 18; i++) F[i * 18 + i] = 1.0;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                F[r * 18 + c] = Eb[r * 3 + c];
                F[(6 + r) * 18 + c] = -RA[r * 3 + c] * dt;
                F[(6 + r) * 18 + 12 + c] = -R[r * 3 + c] * dt;
                RD[r * 3 + c] = R[r * 3 + c] * ic.cov_acc[c];
            }
        for (int a = 0; a < 3; a++) { F[a * 18 + 9 + a] = -dt; F[(3 + a) * 18 + 6 + a] = dt; F[(6 + a) * 18 + 15 + a] = dt; }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Q[r * 3 + c] = RD[r * 3 + 0] * R[c * 3 + 0] + RD[r * 3 + 1] * R[c * 3 + 1] + RD[r * 3 + 2] * R[c * 3 + 2];
        for (int r = 0; r < 18; r++)
            for (int c = 0; c < 18; c++) { double s = 0; for (int q = 0; q < 18; q++) s += F[r * 18 + q] * st.cov[q * 18 + c]; FC[r * 18 + c] = s; }
        for (int r = 0; r < 18; r++)
            for (int c = 0; c < 18; c++) { double s = 0; for (int q = 0; q < 18; q++) s += FC[r * 18 + q] * F[c * 18 + q]; NC[r * 18 + c] = s; }
        for (int a = 0; a < 3; a++) {
            NC[a * 18 + a] += ic.cov_gyr[a] * dt * dt;
            NC[(9 + a) * 18 + 9 + a] += ic.cov_bias_gyr[a] * dt * dt;
            NC[(12 + a) * 18 + 12 + a] += ic.cov_bias_acc[a] * dt * dt;
            for (int c = 0; c < 3; c++) NC[(6 + a) * 18 + 6 + c] += Q[a * 3 + c] * dt * dt;
        }
        std::memcpy(st.cov, NC.data(), sizeof(st.cov));
        double Rn[9];
        mat3_mul(R, Ef, Rn);
        std::memcpy(R, Rn, sizeof(Rn));
        for (int a = 0; a < 3; a++) acc_w[a] = (R[a * 3 + 0] * acc_b[0] + R[a * 3 + 1] * acc_b[1] + R[a * 3 + 2] * acc_b[2]) + st.g[a];
        for (int a = 0; a < 3; a++) { pos[a] = (pos[a] + vel[a] * dt) + 0.5 * acc_w[a] * dt * dt; vel[a] = vel[a] + acc_w[a] * dt; }
        std::memcpy(ic.angvel_last, gyr, 24); std::memcpy(ic.acc_s_last, acc_w, 24);
        snapshot(t.t - pcl_beg, acc_w, gyr, vel, pos, R);
    }
    double sgn;
    if (imu_end > pcl_beg) { sgn = pcl_end > imu_end ? 1.0 : -1.0; dt = sgn * (pcl_end - imu_end); }
    else { sgn = pcl_end > pcl_beg ? 1.0 : -1.0; dt = sgn * (pcl_end - pcl_beg); }
    const double w[3] = {sgn * gyr[0], sgn * gyr[1], sgn * gyr[2]};
    double E[9], Rn[9];
    rot_exp_rate(w, dt, E);
    mat3_mul(R, E, Rn);
    std::memcpy(st.R, Rn, sizeof(Rn));
    for (int a = 0; a < 3; a++) {
        st.vel[a] = vel[a] + sgn * acc_w[a] * dt;
        st.t[a] = (pos[a] + sgn * vel[a] * dt) + sgn * 0.5 * acc_w[a] * dt * dt;
    }
    ic.last_imu = v.back();
    ic.last_lidar_end_time = pcl_end;
}

}  // namespace imh
