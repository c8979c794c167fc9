// ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
// CPU restatement of ImuProcess::UndistortPcl (src/IMU_Processing.cpp:755-958) for LiDAR-only packages (is_lidar_end == true):
// forward propagation of the state over the package's IMU samples, then backward compensation of every point into the scan-end frame.
// Determinism convention: std::sort(time_list) (:784) is unstable; equal offset times keep their arrival order here.
#pragma once
#include "orc_linalg.hpp"
#include "../include/immesh_c_api.h"

namespace orc {

struct Pose6D { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; };   // include/common_lib.h (set_pose6d :303-319)

// Exp(const Matrix<T,3,1>& ang_vel, const Ts& dt)   include/so3_math.h:30-50
inline void so3_exp_rate(const double* w, double dt, double* R) {
    const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n > 0.0000001) {
        const double r[3] = {w[0] / n, w[1] / n, w[2] / n};
        double K[9], KK[9];
        skew(r, K);
        m3_mul(K, K, KK);
        const double a = n * dt, s = std::sin(a), c1 = 1.0 - std::cos(a);
        for (int i = 0; i < 9; i++) R[i] = (R[i] + s * K[i]) + c1 * KK[i];
    }
}

// state layout: R[9] t[3] vel[3] bg[3] ba[3] g[3] cov[324]  (StatesGroup, include/common_lib.h:199-288)
inline void undistort_pcl(const float* pts, int n, const immesh_imu_sample* imu, int n_imu, double lidar_beg_time, double* last_update_time,
                          immesh_imu_ctx* ic, double* st, std::vector<float>& out_xyzi) {
    const double G_m_s2 = 9.81;   // include/common_lib.h:35
    double* R_end = st; double* p_end = st + 9; double* v_end = st + 12; const double* bg = st + 15; const double* ba = st + 18; const double* grav = st + 21;
    double* cov = st + 24;
    std::vector<immesh_imu_sample> v;                                   // :759-761
    v.push_back(ic->last_imu);
    for (int i = 0; i < n_imu; i++) v.push_back(imu[i]);
    const double imu_end_time = v.back().t;
    const double pcl_beg_time = std::max(lidar_beg_time, *last_update_time);   // :764
    // :783-787  (the END time comes from the last point in ARRIVAL order, before the sort)
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pts[(size_t)a * 5 + 4] < pts[(size_t)b * 5 + 4]; });
    const double pcl_end_time = n > 0 ? lidar_beg_time + (double)pts[(size_t)(n - 1) * 5 + 4] / double(1000) : lidar_beg_time;
    *last_update_time = pcl_end_time;
    std::vector<Pose6D> poses;                                          // :799-800
    auto push = [&](double t, const double* a, const double* g, const double* vv, const double* pp, const double* RR) {
        Pose6D q; q.offset_time = t;
        for (int i = 0; i < 3; i++) { q.acc[i] = a[i]; q.gyr[i] = g[i]; q.vel[i] = vv[i]; q.pos[i] = pp[i]; }
        for (int i = 0; i < 9; i++) q.rot[i] = RR[i];
        poses.push_back(q);
    };
    push(0.0, ic->acc_s_last, ic->angvel_last, v_end, p_end, R_end);
    double acc_imu[3] = {ic->acc_s_last[0], ic->acc_s_last[1], ic->acc_s_last[2]};
    double angvel_avr[3] = {ic->angvel_last[0], ic->angvel_last[1], ic->angvel_last[2]};
    double acc_avr[3], vel_imu[3] = {v_end[0], v_end[1], v_end[2]}, pos_imu[3] = {p_end[0], p_end[1], p_end[2]}, R_imu[9];
    std::memcpy(R_imu, R_end, sizeof(R_imu));
    double dt = 0;
    for (size_t k = 0; k + 1 < v.size(); k++) {                         // :808-877
        const immesh_imu_sample& head = v[k]; const immesh_imu_sample& tail = v[k + 1];
        if (tail.t < ic->last_lidar_end_time) continue;
        for (int a = 0; a < 3; a++) { angvel_avr[a] = 0.5 * (head.gyr[a] + tail.gyr[a]); acc_avr[a] = 0.5 * (head.acc[a] + tail.acc[a]); }
        for (int a = 0; a < 3; a++) { angvel_avr[a] -= bg[a]; acc_avr[a] = acc_avr[a] * G_m_s2 / ic->mean_acc_norm - ba[a]; }
        dt = head.t < ic->last_lidar_end_time ? tail.t - ic->last_lidar_end_time : tail.t - head.t;
        double Exp_f[9], Exp_b[9], askew[9];
        so3_exp_rate(angvel_avr, dt, Exp_f);
        so3_exp_rate(angvel_avr, -dt, Exp_b);
        skew(acc_avr, askew);
        std::vector<double> F(324, 0.0), W(324, 0.0), FC(324), NC(324);
        for (int i = 0; i < 18; i++) F[i * 18 + i] = 1.0;
        double RA[9];
        m3_mul(R_imu, askew, RA);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                F[r * 18 + c] = Exp_b[r * 3 + c];                        // (0,0)
                F[r * 18 + 9 + c] = (r == c) ? -dt : 0.0;                // (0,9) = -I dt   (note: -Eye3d * dt has -0.0 off the diagonal; sums are unaffected)
                F[(3 + r) * 18 + 6 + c] = (r == c) ? dt : 0.0;           // (3,6)
                F[(6 + r) * 18 + c] = -RA[r * 3 + c] * dt;               // (6,0) = -R_imu * acc_skew * dt
                F[(6 + r) * 18 + 12 + c] = -R_imu[r * 3 + c] * dt;       // (6,12)
                F[(6 + r) * 18 + 15 + c] = (r == c) ? dt : 0.0;          // (6,15)
            }
        double RD[9], RDR[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) RD[r * 3 + c] = R_imu[r * 3 + c] * ic->cov_acc[c];
        m3_mul_bt(RD, R_imu, RDR);
        for (int a = 0; a < 3; a++) {
            W[a * 18 + a] = ic->cov_gyr[a] * dt * dt;
            W[(9 + a) * 18 + 9 + a] = ic->cov_bias_gyr[a] * dt * dt;
            W[(12 + a) * 18 + 12 + a] = ic->cov_bias_acc[a] * dt * dt;
            for (int c = 0; c < 3; c++) W[(6 + a) * 18 + 6 + c] = RDR[a * 3 + c] * dt * dt;
        }
        for (int r = 0; r < 18; r++) for (int c = 0; c < 18; c++) { double s = 0; for (int q = 0; q < 18; q++) s += F[r * 18 + q] * cov[q * 18 + c]; FC[r * 18 + c] = s; }
        for (int r = 0; r < 18; r++) for (int c = 0; c < 18; c++) { double s = 0; for (int q = 0; q < 18; q++) s += FC[r * 18 + q] * F[c * 18 + q]; NC[r * 18 + c] = s + W[r * 18 + c]; }
        std::memcpy(cov, NC.data(), 324 * sizeof(double));
        double Rn[9];
        m3_mul(R_imu, Exp_f, Rn);
        std::memcpy(R_imu, Rn, sizeof(Rn));
        double Ra[3];
        m3_vec(R_imu, acc_avr, Ra);
        for (int a = 0; a < 3; a++) acc_imu[a] = Ra[a] + grav[a];
        for (int a = 0; a < 3; a++) pos_imu[a] = (pos_imu[a] + vel_imu[a] * dt) + 0.5 * acc_imu[a] * dt * dt;
        for (int a = 0; a < 3; a++) vel_imu[a] = vel_imu[a] + acc_imu[a] * dt;
        for (int a = 0; a < 3; a++) { ic->angvel_last[a] = angvel_avr[a]; ic->acc_s_last[a] = acc_imu[a]; }
        push(tail.t - pcl_beg_time, acc_imu, angvel_avr, vel_imu, pos_imu, R_imu);
    }
    {   // :879-895 prediction at the frame end
        double note;
        if (imu_end_time > pcl_beg_time) { note = pcl_end_time > imu_end_time ? 1.0 : -1.0; dt = note * (pcl_end_time - imu_end_time); }
        else { note = pcl_end_time > pcl_beg_time ? 1.0 : -1.0; dt = note * (pcl_end_time - pcl_beg_time); }
        const double w[3] = {note * angvel_avr[0], note * angvel_avr[1], note * angvel_avr[2]};
        double E[9], Rn[9];
        so3_exp_rate(w, dt, E);
        m3_mul(R_imu, E, Rn);
        for (int a = 0; a < 3; a++) {
            v_end[a] = vel_imu[a] + note * acc_imu[a] * dt;
            p_end[a] = (pos_imu[a] + note * vel_imu[a] * dt) + note * 0.5 * acc_imu[a] * dt * dt;
        }
        std::memcpy(R_end, Rn, sizeof(Rn));
    }
    ic->last_imu = v.back();                                            // :897-898
    ic->last_lidar_end_time = pcl_end_time;
    // sorted copy
    out_xyzi.resize((size_t)n * 4);
    std::vector<float> curv(n);
    for (int i = 0; i < n; i++) {
        const float* p = pts + (size_t)order[i] * 5;
        out_xyzi[(size_t)i * 4 + 0] = p[0]; out_xyzi[(size_t)i * 4 + 1] = p[1]; out_xyzi[(size_t)i * 4 + 2] = p[2]; out_xyzi[(size_t)i * 4 + 3] = p[3];
        curv[i] = p[4];
    }
    if (n < 1) return;                                                  // :911-912
    const double* Lr = ic->lid_rot_to_imu; const double* Lo = ic->lid_offset_to_imu;
    int it = n - 1;                                                     // :914-957 backward compensation
    for (int kp = (int)poses.size() - 1; kp >= 1; kp--) {
        const Pose6D& head = poses[kp - 1];
        for (; (double)curv[it] / double(1000) > head.offset_time; it--) {
            dt = (double)curv[it] / double(1000) - head.offset_time;
            double E[9], R_i[9], T_ei[3];
            so3_exp_rate(head.gyr, dt, E);
            m3_mul(head.rot, E, R_i);
            for (int a = 0; a < 3; a++) T_ei[a] = ((head.pos[a] + head.vel[a] * dt) + 0.5 * head.acc[a] * dt * dt) - p_end[a];
            float* q = &out_xyzi[(size_t)it * 4];
            const double P_i[3] = {(double)q[0], (double)q[1], (double)q[2]};
            double a1[3], a2[3], a3[3], a4[3];
            m3_vec(Lr, P_i, a1);
            for (int a = 0; a < 3; a++) a1[a] += Lo[a];
            m3_vec(R_i, a1, a2);
            for (int a = 0; a < 3; a++) a2[a] += T_ei[a];
            m3t_vec(R_end, a2, a3);
            for (int a = 0; a < 3; a++) a3[a] -= Lo[a];
            m3t_vec(Lr, a3, a4);
            q[0] = (float)a4[0]; q[1] = (float)a4[1]; q[2] = (float)a4[2];
            if (it == 0) break;   // the earliest point is compensated again by every earlier interval (the reference's loop re-enters with it_pcl == begin)
        }
    }
}

// Preprocess::avia_handler with feature_enabled == false (src/preprocess.cpp:139-232).  wire: n x 19 bytes {u32 offset_time; f32 x, y, z; u8 reflectivity, tag, line}
inline int decode_livox(const uint8_t* wire, int n, int n_scans, int point_filter_num, double blind, std::vector<float>& out) {
    out.clear();
    unsigned valid_num = 0;
    const double blind_sqr = blind * blind;
    for (int i = 1; i < n; i++) {
        const uint8_t* p = wire + (size_t)i * 19;
        if (!((int)p[18] < n_scans)) continue;
        valid_num++;
        if (valid_num % (unsigned)point_filter_num != 0) continue;
        float x, y, z; uint32_t ot;
        std::memcpy(&ot, p, 4); std::memcpy(&x, p + 4, 4); std::memcpy(&y, p + 8, 4); std::memcpy(&z, p + 12, 4);
        const float inten = (float)p[16];
        const float curv = (float)ot / float(1000000);
        if ((inten > 4) && ((double)(x * x + y * y + z * z) > blind_sqr)) { out.push_back(x); out.push_back(y); out.push_back(z); out.push_back(inten); out.push_back(curv); }
    }
    return (int)out.size() / 5;
}
// Preprocess::velodyne_handler (src/preprocess.cpp:497-526)
inline int decode_velodyne(const uint8_t* data, int n, int step, int ox, int oy, int oz, int oi, int n_scans, std::vector<float>& out) {
    out.clear();
    for (int i = 0; i < n; i++) {
        const uint8_t* p = data + (size_t)i * step;
        float x, y, z, inten;
        std::memcpy(&x, p + ox, 4); std::memcpy(&y, p + oy, 4); std::memcpy(&z, p + oz, 4); std::memcpy(&inten, p + oi, 4);
        const float angle = (float)((double)(std::atan(z / std::sqrt(x * x + y * y)) * 180) / M_PI);   // float atan / sqrt overloads
        int scan_id;
        if (angle >= -8.83) scan_id = int((2 - angle) * 3.0 + 0.5);
        else scan_id = n_scans / 2 + int((-8.83 - angle) * 2.0 + 0.5);
        if (angle > 2 || angle < -24.33 || scan_id > 50 || scan_id < 0) continue;
        out.push_back(x); out.push_back(y); out.push_back(z); out.push_back(inten); out.push_back(0.f);
    }
    return (int)out.size() / 5;
}

}  // namespace orc
