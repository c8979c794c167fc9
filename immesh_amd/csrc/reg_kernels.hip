// HIP kernels of the registration half of the hot path, gfx950.
//   residual_kernel      BuildResidualListOMP + build_single_residual (src/voxel_mapping.cpp:153-318) fused with the
//                        residual / Jacobian loops (:1372-1392, :1487-1575) and a deterministic block reduction of
//                        H^T R^-1 H (36) and H^T R^-1 z (6): one thread per scan point, map read-only.
//   point_var_kernel     world transform + covariance propagation of map_incremental_grow / voxel_map_init and
//                        find-or-insert of the root voxel (ImMesh_mesh_reconstruction.cpp:393-404, voxel_mapping.cpp:1243-1281, :320-354)
//   replay_kernel        updateVoxelMap / buildVoxelMap semantics: ONE WAVEFRONT PER ROOT VOXEL replays that voxel's points in
//                        the reference's order (OctoTree::UpdateOctoTree state machine, src/voxel_loc.cpp:219-308);
//                        plane fits (OctoTree::init_plane, :47-139) are wave-parallel: lanes stride the retained points,
//                        64-lane butterfly reductions for the moment sums and the 21-entry plane covariance.
// Bound: HBM/latency (dependent gathers through hash -> node -> plane); no GEMM-shaped work, MFMA is not used.
#include <algorithm>
#include <cstdlib>
#include "regmap.hpp"
#include "kernels.hpp"
#include "prof.hpp"

using namespace imd;

// =====================================================================================================================
// matcher + H build
// =====================================================================================================================
// what a point-to-plane test reads of a node record: kept in registers from the matcher to the H build (no second gather)
struct PlaneRegs { double n[3], c[3], pv[21]; float d, radius; };
struct BestMatch { int node; int layer; double prob; bool ok; PlaneRegs P; };

IMD double plane_sigma(const double* pv, const double* J) {  // J * plane_var * J^T, plane_var symmetric (21)
    double tmp[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double s = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) s += J[r] * pv[(r <= c) ? sym21_index(r, c) : sym21_index(c, r)];
        tmp[c] = s;
    }
    double sig = 0;
#pragma unroll
    for (int c = 0; c < 6; c++) sig += tmp[c] * J[c];
    return sig;
}
IMD void load_plane_geometry(const NodeRec& nr, PlaneRegs& P) {
#pragma unroll
    for (int k = 0; k < 3; k++) { P.n[k] = nr.p_normal[k]; P.c[k] = nr.p_center[k]; }
    P.d = nr.d; P.radius = nr.radius;
}
IMD void load_plane_var(const NodeRec& nr, PlaneRegs& P) {
#pragma unroll
    for (int k = 0; k < 21; k++) P.pv[k] = nr.p_var[k];
}

// build_single_residual on one plane node (voxel_mapping.cpp:252-290) with the plane record P already in registers: both gates are evaluated
// from there; computing sigma_l before knowing that the range gate passed has no side effect, so the accept set is the reference's.
// TIE: an exact tie of probabilities goes to the plane the reference's depth-first recursion reaches first (leaf lists are in insertion order).
template <bool TIE>
IMD void test_plane(const RegMapDev& m, const PlaneRegs& P, int node, int layer, const double* pw, const double* var, double sigma_num, float dis_to_plane, float range_dis,
                    BestMatch& best) {
    const double nx = P.n[0], ny = P.n[1], nz = P.n[2];
    const double J[6] = {pw[0] - P.c[0], pw[1] - P.c[1], pw[2] - P.c[2], -nx, -ny, -nz};
    double sigma_l = plane_sigma(P.pv, J);
    const double nrm[3] = {nx, ny, nz};
    double vn[3];
    m3t_vec(var, nrm, vn);
    sigma_l += vn[0] * nx + vn[1] * ny + vn[2] * nz;
    if ((double)range_dis <= 3.0 * (double)P.radius && (double)dis_to_plane < sigma_num * sqrt(sigma_l)) {
        best.ok = true;
        const double this_prob = 1.0 / (sqrt(sigma_l)) * exp(-0.5 * (double)dis_to_plane * (double)dis_to_plane / sigma_l);
        if (this_prob > best.prob || (TIE && this_prob == best.prob && best.node >= 0 && dfs_key(m, node) < dfs_key(m, best.node))) {
            best.prob = this_prob; best.node = node; best.layer = layer; best.P = P;
        }
    }
}
IMD void plane_gates(const PlaneRegs& P, const double* pw, float& dis_to_plane, float& range_dis) {
    dis_to_plane = (float)fabs(P.n[0] * pw[0] + P.n[1] * pw[1] + P.n[2] * pw[2] + (double)P.d);
    const float dis_to_center = (float)((P.c[0] - pw[0]) * (P.c[0] - pw[0]) + (P.c[1] - pw[1]) * (P.c[1] - pw[1]) + (P.c[2] - pw[2]) * (P.c[2] - pw[2]));
    range_dis = sqrtf(dis_to_center - dis_to_plane * dis_to_plane);  // NaN compares false in the gate
}

// build_single_residual's recursion over ALL existing children of non-plane nodes (voxel_mapping.cpp:299-312): the planes it reaches are the
// root's flat leaf list.  The root's own plane record is gathered together with its flags -- ONE dependent access for the common case of a
// planar root voxel (a subdivided root pays three wasted lines); leaves fetch their covariance only when they survive the range gate.
IMD void match_tree(const RegMapDev& m, int root, const double* pw, const double* var, double sigma_num, BestMatch& best, int& n_tests) {
    const NodeRec& nr = m.nodes[root];
    const int flags = nr.flags, leaf_head = nr.leaf_head;
    PlaneRegs P;
    load_plane_geometry(nr, P);
    load_plane_var(nr, P);
    if (flags & NF_PLANE) {
        n_tests++;
        float dp, rd;
        plane_gates(P, pw, dp, rd);
        test_plane<false>(m, P, root, 0, pw, var, sigma_num, dp, rd, best);
        return;
    }
    if (m.max_layer <= 0) return;
    for (int ch = leaf_head; ch >= 0;) {
        int e[16];
#pragma unroll
        for (int k = 0; k < 16; k++) e[k] = m.leaf_chunks[(size_t)ch * 16 + k];   // one 64-byte line: 15 node ids + next
#pragma unroll
        for (int s2 = 0; s2 < IM_LEAF_SLOTS; s2++)
            if (e[s2] >= 0) {
                n_tests++;
                const NodeRec& lr = m.nodes[e[s2]];
                load_plane_geometry(lr, P);
                float dp, rd;
                plane_gates(P, pw, dp, rd);
                if (!((double)rd <= 3.0 * (double)P.radius)) continue;
                load_plane_var(lr, P);
                test_plane<true>(m, P, e[s2], 1, pw, var, sigma_num, dp, rd, best);
            }
        ch = e[15];
    }
}



// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the leaf lists of a wavefront's points, matched by the WHOLE wavefront.  match_tree above walks a non-planar root's flat leaf list in the lane
// of its point -- one dependent gather per leaf, one leaf after the other -- while the lanes whose root voxel is a plane (97 % on the avia map) have long
// finished: a lane with twenty leaves held its wavefront, and through the pass's all-gather every wavefront of the grid, for twenty gathers (velodyne.yaml,
// every root subdivided: 45 k of a pass's 60 k cycles).  Here the (point, leaf) PAIRS of the wavefront are enumerated into an LDS work list -- each "tree
// lane" contributes one 15-id chunk line of its list per step -- and taken 64 at a time, one pair per lane: the leaf gathers of a step are in flight
// together.  A pair's lane evaluates build_single_residual's test with the owner's point (fetched by lane shuffles) in the SAME expressions as
// test_plane; the owner's best is the maximum probability (ties: the smaller depth-first key -- the plane the reference's recursion reaches first), found
// by LDS atomics on the probability's bit pattern, which is monotone for the positive values that can win (`this_prob > best.prob`, best.prob >= 0).
// The accept set, the winner and every value derived from it are the sequential walk's; only who computes what has changed.
// wl: COOP_WORDS words of LDS of this wavefront.  Node ids must fit 26 bits (the caller checks).
// ---------------------------------------------------------------------------------------------------------------------
#define COOP_LIST 960                                  /* 64 lanes x 15 leaves of one enumeration step */
#define COOP_STAGE (64 * 12 * 2)                       /* words: every lane's world point (3 doubles) + its covariance (9): what a pair's lane needs of the owner */
#define COOP_WORDS (COOP_STAGE + COOP_LIST + 64 * 2 + 64 * 3)   /* + per owner: best probability (64-bit), winner, any-accepted flag, finalists of the round.  13.4 KB:
                                                          lives in the wavefront's reduction transpose buffer (16.6 KB), which is idle while points are matched */
IMD void coop_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
IMD double shfl_d(const double x, const int src) {
    const int lo = __shfl(__double2loint(x), src, 64), hi = __shfl(__double2hiint(x), src, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void match_trees_coop(const RegMapDev& m, const bool is_tree, const int head, const double* pw, const double* var, const double sigma_num, BestMatch& best,
                                                 int& n_tests, const int lane, unsigned int* __restrict__ wl) {
    double* const stage = (double*)wl;                 // [64][12]: pw, var of every lane (read by the lanes that take its pairs)
    unsigned int* const list = wl + COOP_STAGE;
    unsigned long long* const bestp = (unsigned long long*)(list + COOP_LIST);
    int* const bestn = (int*)(list + COOP_LIST + 128);
    int* const okf = (int*)(list + COOP_LIST + 192);
    unsigned int* const nfin = list + COOP_LIST + 256;
    bestp[lane] = 0ull; bestn[lane] = -1; okf[lane] = 0;
    if (is_tree) {
#pragma unroll
        for (int k = 0; k < 3; k++) stage[lane * 12 + k] = pw[k];
#pragma unroll
        for (int k = 0; k < 9; k++) stage[lane * 12 + 3 + k] = var[k];
    }
    int ch = is_tree ? head : -1;
    while (__any(ch >= 0)) {
        // ---- enumeration step: every tree lane's next chunk line (15 node ids + next) goes onto the work list
        int e[16];
        int cnt = 0;
        if (ch >= 0) {
#pragma unroll
            for (int k = 0; k < 16; k++) e[k] = m.leaf_chunks[(size_t)ch * 16 + k];
#pragma unroll
            for (int k = 0; k < IM_LEAF_SLOTS; k++) cnt += e[k] >= 0 ? 1 : 0;
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
        const int total = __shfl(incl, 63, 64);
        if (ch >= 0) {
            int k2 = incl - cnt;
#pragma unroll
            for (int k = 0; k < IM_LEAF_SLOTS; k++) if (e[k] >= 0) list[k2++] = ((unsigned int)lane << 26) | (unsigned int)e[k];
            n_tests += cnt;
            ch = e[15];
        }
        coop_sync();
        // ---- the pairs, 64 at a time
        for (int base = 0; base < total; base += 64) {
            const int idx = base + lane;
            const bool live = idx < total;
            const unsigned int ent = live ? list[idx] : ((unsigned int)lane << 26);
            const int owner = (int)(ent >> 26), node = (int)(ent & 0x3FFFFFFu);
            bool cand = false;
            double prob = 0.0;
            if (live) {
                const NodeRec& lr = m.nodes[node];
                PlaneRegs P;
                load_plane_geometry(lr, P);
                const double opw[3] = {stage[owner * 12 + 0], stage[owner * 12 + 1], stage[owner * 12 + 2]};
                float dp, rd;
                plane_gates(P, opw, dp, rd);
                if ((double)rd <= 3.0 * (double)P.radius) {
                    load_plane_var(lr, P);
                    double ovar[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) ovar[k] = stage[owner * 12 + 3 + k];
                    // test_plane<true>, its comparison against the running best replaced by the reduction below
                    const double nx = P.n[0], ny = P.n[1], nz = P.n[2];
                    const double J[6] = {opw[0] - P.c[0], opw[1] - P.c[1], opw[2] - P.c[2], -nx, -ny, -nz};
                    double sigma_l = plane_sigma(P.pv, J);
                    const double nrm[3] = {nx, ny, nz};
                    double vn[3];
                    m3t_vec(ovar, nrm, vn);
                    sigma_l += vn[0] * nx + vn[1] * ny + vn[2] * nz;
                    if ((double)dp < sigma_num * sqrt(sigma_l)) {
                        atomicOr(&okf[owner], 1);
                        prob = 1.0 / (sqrt(sigma_l)) * exp(-0.5 * (double)dp * (double)dp / sigma_l);
                        cand = prob > 0.0;
                    }
                }
            }
            if (!__any(cand)) continue;   // (wave-uniform)
            // ---- the owners' maxima.  Exact ties (two planes, one probability to the last bit) go to the smaller depth-first key: the rare path below
            const unsigned long long bits = (unsigned long long)__double_as_longlong(prob);
            nfin[lane] = 0u;
            const unsigned long long before = cand ? bestp[owner] : 0ull;
            coop_sync();
            if (cand) atomicMax(&bestp[owner], bits);
            coop_sync();
            const unsigned long long after = cand ? bestp[owner] : 0ull;
            const bool fin = cand && bits == after;
            const int prevn = fin ? bestn[owner] : -1;
            if (fin) atomicAdd(&nfin[owner], 1u);
            coop_sync();
            const bool tie = fin && (nfin[owner] > 1u || (after == before && prevn >= 0));
            if (fin && !tie) bestn[owner] = node;
            if (__any(tie)) {
                // all of an owner's finalists (and the winner of an earlier step that holds the same probability) compete with their depth-first keys
                unsigned int* const bestk = nfin;   // (the counts have been read)
                coop_sync();
                if (tie) bestk[owner] = 0xFFFFFFFFu;
                coop_sync();
                const unsigned int dk = tie ? dfs_key(m, node) : 0xFFFFFFFFu;
                if (tie && after == before && prevn >= 0) atomicMin(&bestk[owner], dfs_key(m, prevn));
                if (tie) atomicMin(&bestk[owner], dk);
                coop_sync();
                if (tie && bestk[owner] == dk) bestn[owner] = node;
            }
            coop_sync();
        }
        coop_sync();   // (the work list is rewritten by the next step)
    }
    coop_sync();
    if (is_tree) {
        if (okf[lane]) best.ok = true;
        const int nd = bestn[lane];
        const double p = __longlong_as_double((long long)bestp[lane]);
        if (nd >= 0 && p > best.prob) {
            best.node = nd; best.layer = 1; best.prob = p;
            const NodeRec& lr = m.nodes[nd];
            load_plane_geometry(lr, best.P);
            load_plane_var(lr, best.P);
        }
    }
    coop_sync();
}

// =====================================================================================================================
// the 18-state iterated-EKF update on one wavefront (imh::EkfLoop::step, ekf_host.hpp -- the same operations in the same order:
// Gauss-Jordan with partial pivoting on [H^T R^-1 H + P^-1 | e_0..e_5], G = K1 H^T R^-1 H, solution, boxplus, stop rule, (I - G) P)
// =====================================================================================================================
IMD double rl_d(double x, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), k), hi = __builtin_amdgcn_readlane(__double2hiint(x), k);
    return __hiloint2double(hi, lo);
}
// write-through store / coherent load of a double other wavefronts of the same launch read (the L2 of another XCD may hold a stale line)
IMD void dev_publish(double* p, const double v) { __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
IMD double dev_observe(const double* p) { return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
IMD double uni_d(const double x) {   // a wave-uniform value into scalar registers
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
// The rotation increments of the iterated update are far below a radian, and the generic double-precision sin / cos / acos cost ~1.7 us EACH on
// the single wavefront that runs the update (4 k cycles for Exp, 4 k for Log, measured): below 0.5 rad (Exp) / 0.05 (Log) the functions are summed
// from their Taylor series to below 1e-17 relative -- at least as close to the true value as libm -- and libm serves the rest.
IMD void sin_1mcos(const double x, double* sn, double* c1) {
    if (fabs(x) < 0.5) {
        const double x2 = x * x;
        double a = 1.0 - x2 * (1.0 / 272.0);
        a = 1.0 - x2 * (1.0 / 210.0) * a; a = 1.0 - x2 * (1.0 / 156.0) * a; a = 1.0 - x2 * (1.0 / 110.0) * a; a = 1.0 - x2 * (1.0 / 72.0) * a;
        a = 1.0 - x2 * (1.0 / 42.0) * a; a = 1.0 - x2 * (1.0 / 20.0) * a; a = 1.0 - x2 * (1.0 / 6.0) * a;
        *sn = x * a;
        double b = 1.0 - x2 * (1.0 / 306.0);
        b = 1.0 - x2 * (1.0 / 240.0) * b; b = 1.0 - x2 * (1.0 / 182.0) * b; b = 1.0 - x2 * (1.0 / 132.0) * b; b = 1.0 - x2 * (1.0 / 90.0) * b;
        b = 1.0 - x2 * (1.0 / 56.0) * b; b = 1.0 - x2 * (1.0 / 30.0) * b; b = 1.0 - x2 * (1.0 / 12.0) * b;
        *c1 = 0.5 * x2 * b;
    } else {
        double cs;
        sincos(x, sn, &cs);
        *c1 = 1.0 - cs;
    }
}
IMD void dev_so3_exp(double v1, double v2, double v3, double* R) {  // include/so3_math.h:71-89
    const double norm = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (norm > 0.00001) {
        const double r[3] = {v1 / norm, v2 / norm, v3 / norm};
        const double K[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
        double KK[9];
        m3_mul(K, K, KK);
        double sn, c1;
        sin_1mcos(norm, &sn, &c1);
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = (R[i] + sn * K[i]) + c1 * KK[i];
    }
}
IMD void dev_so3_log(const double* R, double* out) {  // include/so3_math.h:92-98: theta = acos((tr - 1) / 2); theta < 1e-3 ? K / 2 : theta / (2 sin theta) K
    const double tr = R[0] + R[4] + R[8];
    const double ct = 0.5 * (tr - 1);
    const double K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double s2 = 0.25 * (K[0] * K[0] + K[1] * K[1] + K[2] * K[2]);   // sin^2 theta (K is twice the sine times the axis)
    double f = 0.5;
    if (!(tr > 3.0 - 1e-6)) {
        if (ct > 0 && s2 < 0.0025) {
            // theta / sin(theta) = asin(s) / s, s = sin(theta) < 0.05: the series of the arc sine (the reference's acos(1 - theta^2 / 2) only resolves theta to 1e-16 / theta^2 anyway)
            const double q = 1.0 + s2 * (1.0 / 6.0 + s2 * (3.0 / 40.0 + s2 * (15.0 / 336.0 + s2 * (105.0 / 3456.0 + s2 * (945.0 / 42240.0 + s2 * (10395.0 / 599040.0))))));
            if (!(s2 * q * q < 1e-6)) f = 0.5 * q;          // theta = s q against the 1e-3 cut
        } else {
            const double theta = acos(ct);
            if (!(fabs(theta) < 0.001)) f = 0.5 * theta / sqrt((1.0 - ct) * (1.0 + ct));
        }
    }
    out[0] = f * K[0]; out[1] = f * K[1]; out[2] = f * K[2];
}
// hth: 6x6 row-major, htz: 6 (LDS).  W: >= 260 doubles of LDS scratch.  cov: the prior covariance (only read when the loop stops).
// Returns true when the loop stops with this pass (rs->done set, posterior written to reg_out).
// The gain: K1 = (H^T R^-1 H + P^-1)^-1 restricted to its first six columns (H only has the six pose columns).  With P = [P11 P12; P21 P22]
// those columns are [X; P21 P11^-1 X], X = (H^T R^-1 H + P11^-1)^-1 (block inverse + Schur complement) -- a 6x6 inverse per pass instead of the
// 18x18 one of ekf_host.hpp / Eigen; P11^-1 and P21 P11^-1 are per-scan constants the host supplies.  Same result up to rounding (~1e-14 rel.).
// C (LDS): the staged inputs -- [0,36) P11^-1, [36,108) P21 P11^-1, [108,132) iterate, [132,156) prior, [156,160) cumulative counters (this pass
// included), [160] rematch count.
#define EKF_C_DOUBLES 164
__device__ __forceinline__ bool ekf_step_wave(RegState* __restrict__ rs, const double* C, const double* __restrict__ cov, const double* hth, const double* htz, const double n_match,
                                              const double res_sum, double* W, const int lane, const int it, const int max_iter, const double* __restrict__ extR,
                                              double* __restrict__ reg_out, const double ticket, unsigned long long* __restrict__ dbg = nullptr, double* __restrict__ hist = nullptr) {
    unsigned long long tk = dbg ? __builtin_readcyclecounter() : 0;
#define EDBG(k) do { if (dbg) { const unsigned long long _t = __builtin_readcyclecounter(); if (lane == 0) dbg[40 + (k)] += _t - tk; tk = _t; } } while (0)
    // ---- X = (H + P11^-1)^-1: lane j < 12 holds column j of [S | I] in registers; Gauss-Jordan without pivoting (S is symmetric positive definite)
    double a[6];
    {
        const int j = lane < 12 ? lane : 0;
#pragma unroll
        for (int r = 0; r < 6; r++) a[r] = j < 6 ? hth[r * 6 + j] + C[r * 6 + j] : ((r == j - 6) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int col = 0; col < 6; col++) {
        const double d = rl_d(a[col], col);
        double f[6];
#pragma unroll
        for (int r = 0; r < 6; r++) f[r] = rl_d(a[r], col);
        a[col] = a[col] / d;
#pragma unroll
        for (int r = 0; r < 6; r++) if (r != col) a[r] -= f[r] * a[col];
    }
    EDBG(0);
    // K1 rows 0..5 = X -> W[0,36); rows 6..17 = (P21 P11^-1) X -> W[36,108)
    if (lane >= 6 && lane < 12) {
#pragma unroll
        for (int r = 0; r < 6; r++) W[r * 6 + (lane - 6)] = a[r];
    }
    __syncthreads();
    for (int e = lane; e < 72; e += 64) {
        const int i = e / 6, k = e % 6;
        double sacc = 0;
#pragma unroll
        for (int mm = 0; mm < 6; mm++) sacc += C[36 + i * 6 + mm] * W[mm * 6 + k];
        W[36 + e] = sacc;
    }
    __syncthreads();
    // G = K1 * HTH (18 x 6) -> W[108,216)
    for (int e = lane; e < 108; e += 64) {
        const int r = e / 6, c = e % 6;
        double sacc = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) sacc += W[r * 6 + k] * hth[k * 6 + c];
        W[108 + e] = sacc;
    }
    EDBG(1);
    // vec = prior [-] state  (every lane computes the same 18 values)
    double st[24], vec[18];
#pragma unroll
    for (int k = 0; k < 24; k++) st[k] = C[108 + k];
    {
        double Rt[9], rotd[9], pR[9];
#pragma unroll
        for (int k = 0; k < 9; k++) pR[k] = C[132 + k];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j2 = 0; j2 < 3; j2++) Rt[i * 3 + j2] = st[j2 * 3 + i];
        m3_mul(Rt, pR, rotd);
        dev_so3_log(rotd, vec);
#pragma unroll
        for (int k = 0; k < 15; k++) vec[3 + k] = C[132 + 9 + k] - st[9 + k];
    }
    __syncthreads();
    // solution: (K1 HTz + vec) - G vec    (G and vec only act on the first 6 states)
    double my_sol = 0;
    if (lane < 18) {
        double s1 = 0, s2 = 0;
        double vr = 0;
#pragma unroll
        for (int k = 0; k < 18; k++) if (lane == k) vr = vec[k];
#pragma unroll
        for (int k = 0; k < 6; k++) { s1 += W[lane * 6 + k] * htz[k]; s2 += W[108 + lane * 6 + k] * vec[k]; }
        my_sol = (s1 + vr) - s2;
    }
    double sol[18];
#pragma unroll
    for (int k = 0; k < 18; k++) sol[k] = rl_d(my_sol, k);
    EDBG(2);
    // state += solution
    {
        double E[9], Rn[9];
        dev_so3_exp(sol[0], sol[1], sol[2], E);
        m3_mul(st, E, Rn);
#pragma unroll
        for (int k = 0; k < 9; k++) st[k] = Rn[k];
#pragma unroll
        for (int k = 0; k < 15; k++) st[9 + k] += sol[3 + k];
    }
    const double rn = sqrt(sol[0] * sol[0] + sol[1] * sol[1] + sol[2] * sol[2]);
    const double tn = sqrt(sol[3] * sol[3] + sol[4] * sol[4] + sol[5] * sol[5]);
    const bool converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
    int rematch = (int)C[160];
    if (converged || ((rematch == 0) && (it == (max_iter - 2)))) rematch++;
    const bool stop = rematch >= 2 || (it == max_iter - 1);
    EDBG(3);
    // next pass / map update parameters
    if (lane < 24) {
        double v = 0;
#pragma unroll
        for (int k = 0; k < 24; k++) if (lane == k) v = st[k];
        dev_publish(&rs->st[lane], v);
        if (hist) dev_publish(&hist[lane], v);
    }
    {
        double RextR[9];
        m3_mul(st, extR, RextR);
        if (lane < 9) { double v = 0, w = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) if (lane == k) { v = st[k]; w = RextR[k]; }
            rs->sp.R[lane] = v; rs->sp.RextR[lane] = w; }
        if (lane < 3) { double v = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) if (lane == k) v = st[9 + k];
            rs->sp.t[lane] = v; }
    }
    const double passes = C[158] + 1.0;
    if (lane == 0) {
        dev_publish(&rs->tot[0], C[156]); dev_publish(&rs->tot[1], C[157]); dev_publish(&rs->tot[2], passes); dev_publish(&rs->tot[3], C[159]);
        __hip_atomic_store(&rs->rematch, rematch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&rs->done, stop ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hist) { dev_publish(&hist[24], C[156]); dev_publish(&hist[25], C[157]); dev_publish(&hist[26], passes); dev_publish(&hist[27], C[159]); dev_publish(&hist[28], (double)rematch); dev_publish(&hist[29], stop ? 1.0 : 0.0); }
    }
    if (stop) {
        // cov = (I - G) * cov ; G is zero outside its first 6 columns:  cov[r][c] - sum_{k<6} G[r][k] cov[k][c]
        for (int e = lane; e < 324; e += 64) {
            const int r = e / 18, c = e % 18;
            double g6[6], c6[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { g6[k] = W[108 + r * 6 + k]; c6[k] = cov[k * 18 + c]; }
            double sub = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) sub += g6[k] * c6[k];
            const double sacc = cov[e] - sub;
            __hip_atomic_store((unsigned long long*)&reg_out[24 + e], (unsigned long long)__double_as_longlong(sacc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (r < 3 && c < 3) rs->sp.rot_var[r * 3 + c] = sacc;                       // the map update propagates the POSTERIOR covariance blocks
            if (r >= 3 && r < 6 && c >= 3 && c < 6) rs->sp.t_var[(r - 3) * 3 + (c - 3)] = sacc;
        }
        if (lane < 24) {
            double v = 0;
#pragma unroll
            for (int k = 0; k < 24; k++) if (lane == k) v = st[k];
            __hip_atomic_store((unsigned long long*)&reg_out[lane], (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (lane == 0) {
            const double o[6] = {passes, n_match, res_sum, C[156], C[157], C[159]};
            for (int k = 0; k < 6; k++) __hip_atomic_store((unsigned long long*)&reg_out[348 + k], (unsigned long long)__double_as_longlong(o[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (lane == 0) __hip_atomic_store(&reg_out[REG_OUT_DOUBLES - 1], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    EDBG(4);
    if (dbg && lane == 0) dbg[47] += 1;
#undef EDBG
    return stop;
}

#define RES_NV 48   // host layout: 36 HTH + 6 HTz + n_match + sum|dis| + n_plane_tests + n_extra_probe + 2 spare
#define RES_NR 32   // reduced per block: 21 (upper triangle of HTH) + 6 HTz + 4 counters + 1 spare


// One point of one residual pass: transformLidar + covariance propagation (lio_state_estimation :1302-1359), BuildResidualListOMP with the
// near-voxel retry (:171-222), residual (:1372-1392), H / R^-1 (:1493-1575); adds the point's terms of H^T R^-1 H / H^T R^-1 z and the counters
// to acc[RES_NR] and writes the per-point match outputs.  Shared by the single-pass kernel and the persistent one.
// What does not depend on the iterate is computed once per scan (residual_prep) and stays in registers over the passes of the resident grid.
#define RDBG(k) do { if (sp.dbg) { const unsigned long long _t = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&sp.dbg[k], _t - tprev); tprev = _t; } } while (0)
// The tail of a map update (all 256 threads of one workgroup): chunks freed by the replay kernels join the free list, the per-update counters reset,
// and the map counters (node / chunk usage, capacity flag) go to pinned host memory.  Its own one-workgroup launch for the synchronous entry points; an
// asynchronous immesh_process_scan leaves it to the prologue of the NEXT scan's residual_persistent_kernel (one launch less on the pose chain).
__device__ __forceinline__ void map_update_tail(const RegMapDev& m, int32_t* __restrict__ host_counters) {
    const int np = m.counters[3];
    const int base = m.counters[2];
    for (int i = threadIdx.x; i < np; i += 256) m.free_ready[base + i] = m.free_pending[i];
    __syncthreads();
    if (threadIdx.x == 0) { m.counters[2] = base + np; m.counters[3] = 0; m.counters[7] = 0; m.counters[9] = 0; m.counters[10] = 0; m.counters[11] = 0; m.counters[12] = 0; }
    __syncthreads();
    if (threadIdx.x < 16) __hip_atomic_store(&host_counters[threadIdx.x], m.counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct PointPrep {
    double pimu[3];     // the point in the IMU frame
    double bcov[9];     // calcBodyVar of the lidar-frame point (:1302-1316)
    double b[9];        // (-[p_imu]x) Srot (-[p_imu]x)^T with the z == 0 -> 1e-3 quirk
    double bv[9];       // calcBodyVar of the IMU-frame point (H build, :1509-1517)
    unsigned long long key; int root;   // root voxel the previous pass found the point in (PREP_NO_KEY: none yet)
    unsigned int slot;                  // ... and its hash slot (valid with root >= 0): the map update's preparation starts from it (RpEpilogue)
};
#ifndef REG_UPD_PRIO
#define REG_UPD_PRIO 1   /* wave priority of the map update's kernels */
#endif
#define PREP_NO_KEY 0xFFFFFFFFFFFFFFFEull
#ifndef IMMESH_EPI_HINT
#define IMMESH_EPI_HINT 1   /* (0: A/B builds without the epilogue's slot hint) */
#endif
IMD void residual_prep(const ScanParams& sp, const float* __restrict__ pts, const int i, PointPrep& q) {
    const double p[3] = {(double)pts[(size_t)i * 3 + 0], (double)pts[(size_t)i * 3 + 1], (double)pts[(size_t)i * 3 + 2]};
    double pz[3] = {p[0], p[1], p[2]};
    if (pz[2] == 0) pz[2] = 0.001;
    calc_body_var(pz, sp.dept_err, sp.dvar_beam, q.bcov);
    double pimu_z[3];
    m3_vec(sp.extR, pz, pimu_z);
    pimu_z[0] += sp.extT[0]; pimu_z[1] += sp.extT[1]; pimu_z[2] += sp.extT[2];
    m3_vec(sp.extR, p, q.pimu);
    q.pimu[0] += sp.extT[0]; q.pimu[1] += sp.extT[1]; q.pimu[2] += sp.extT[2];
    {
        double cm[9], nc[9], nct[9], tmp[9];
        skew(pimu_z, cm);
#pragma unroll
        for (int k = 0; k < 9; k++) nc[k] = -cm[k];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) nct[r * 3 + c] = -cm[c * 3 + r];
        m3_mul(nc, sp.rot_var, tmp);
        m3_mul(tmp, nct, q.b);
    }
    double pthis[3] = {q.pimu[0], q.pimu[1], q.pimu[2]};
    calc_body_var(pthis, sp.dept_err, sp.calib_laser ? sp.dvar_calib : sp.dvar_beam, q.bv);
    q.key = PREP_NO_KEY; q.root = -1; q.slot = 0;
}
// sp: the per-scan constants (kernel arguments: scalar loads); Rm / tv / RextR: the iterate of this pass (wave-uniform)
template <bool coop>
IMD void residual_pass(const RegMapDev& m, const ScanParams& sp, const double* Rm, const double* tv, const double* RextR, PointPrep& q, const int i, double* acc, unsigned long long& tprev,
                       int8_t* __restrict__ o_match, int32_t* __restrict__ o_node, float* __restrict__ o_dis, double* __restrict__ o_rinv, double* __restrict__ o_normal,
                       const bool active, unsigned int* __restrict__ wl) {
        // EVERY lane of the wavefront calls this (active = the lane holds a point): the leaf lists of non-planar roots are matched by the whole wavefront
        // (match_trees_coop).  coop == false (node ids beyond 26 bits, or IMMESH_MATCH_SEQ): the lane-by-lane walk of rounds 1-5 (match_tree).
        const int lane = (int)(threadIdx.x & 63);
        // --- transformLidar (:1344): f64 compute, f32 store
        double pwd[3];
        m3_vec(Rm, q.pimu, pwd);
        pwd[0] += tv[0]; pwd[1] += tv[1]; pwd[2] += tv[2];
        const double pw[3] = {(double)(float)pwd[0], (double)(float)pwd[1], (double)(float)pwd[2]};
        // --- covariance propagation (:1346-1359)
        double var[9];
        {
            double a[9];
            m3_sandwich(Rm, q.bcov, a);
#pragma unroll
            for (int k = 0; k < 9; k++) var[k] = (a[k] + q.b[k]) + sp.t_var[k];
        }
        RDBG(0);
        // --- BuildResidualListOMP (:171-222)
        float loc[3];
        int64_t kx[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { loc[j] = loc_axis(pw[j] / m.voxel_size_d); kx[j] = (int64_t)loc[j]; }
        // sharded map: a point is matched by the rank that owns its root voxel (every rank sees the whole scan; the 48 sums are all-reduced)
        const bool mine = active && (m.shard_world <= 1 || shard_owner(m, kx[0], kx[1], kx[2]) == m.shard_rank);
        const uint64_t key = pack_key(kx[0], kx[1], kx[2]);
        int root = -1;
        if (mine) {
            if (key == q.key) root = q.root;   // the point stayed in the voxel of the previous pass (the map does not change during a scan's passes)
            else {
                const int64_t slot = hash_find(m, key);
                root = slot >= 0 ? m.htab[slot].root : -2;   // -2: no such voxel
                q.key = key; q.root = root; q.slot = (unsigned int)slot;
            }
        }
        BestMatch best; best.node = -1; best.layer = 0; best.prob = 0; best.ok = false;
        int n_tests = 0, n_extra = 0;
        // two rounds: the point's own root voxel, then -- no plane accepted -- the neighbour of the near-voxel retry (the literal unit mismatch)
        int cur_root = (mine && root >= 0) ? root : -1;
        for (int round = 0; round < 2; round++) {
            bool is_tree = false;
            int tree_head = -1;
            if (cur_root >= 0) {
                if (!coop) match_tree(m, cur_root, pw, var, sp.sigma_num, best, n_tests);
                else {
                    // (the root's plane record is gathered together with its flags: one dependent access for the common case of a planar root voxel)
                    const NodeRec& nr = m.nodes[cur_root];
                    const int flags = nr.flags, leaf_head = nr.leaf_head;
                    PlaneRegs P;
                    load_plane_geometry(nr, P);
                    load_plane_var(nr, P);
                    if (flags & NF_PLANE) {
                        n_tests++;
                        float dp, rd;
                        plane_gates(P, pw, dp, rd);
                        test_plane<false>(m, P, cur_root, 0, pw, var, sp.sigma_num, dp, rd, best);
                    } else if (m.max_layer > 0) { is_tree = true; tree_head = leaf_head; }
                }
            }
            if (coop && __any(is_tree)) match_trees_coop(m, is_tree, tree_head, pw, var, sp.sigma_num, best, n_tests, lane, wl);
            if (round == 0) {
                RDBG(1);
                const bool retry = mine && root >= 0 && !best.ok;
                cur_root = -1;
                if (retry) {
                    int64_t nk[3] = {kx[0], kx[1], kx[2]};
                    const float ql = m.nodes[root].quarter;
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const double c = m.nodes[root].center[j];
                        if ((double)loc[j] > (c + (double)ql)) nk[j] = nk[j] + 1;
                        else if ((double)loc[j] < (c - (double)ql)) nk[j] = nk[j] - 1;
                    }
                    n_extra = 1;
                    const int64_t s2 = hash_find(m, pack_key(nk[0], nk[1], nk[2]));
                    if (s2 >= 0 && m.htab[s2].root >= 0) cur_root = m.htab[s2].root;
                }
                if (!__any(cur_root >= 0)) break;
            }
        }
        if (!active) return;
        RDBG(2);
        acc[29] += (double)n_tests; acc[30] += (double)n_extra;
        // is_success with prob still 0 (sigma_l = inf / NaN after a diverged covariance): the reference pushes an uninitialised ptpl there;
        // here it is no match instead of a read of nodes[-1]
        const bool matched = best.ok && best.node >= 0;
        o_match[i] = matched ? 1 : 0;
        o_node[i] = best.node;
        if (matched) {
            const PlaneRegs& P = best.P;
            // residual (:1372-1392): float normals, unrounded world point
            const float nxf = (float)P.n[0], nyf = (float)P.n[1], nzf = (float)P.n[2];
            const float dis = (float)(pwd[0] * (double)nxf + pwd[1] * (double)nyf + pwd[2] * (double)nzf + (double)P.d);
            o_dis[i] = dis;
            o_normal[(size_t)i * 3 + 0] = P.n[0]; o_normal[(size_t)i * 3 + 1] = P.n[1]; o_normal[(size_t)i * 3 + 2] = P.n[2];
            // H / R^-1 (:1493-1575)
            const double nv[3] = {(double)nxf, (double)nyf, (double)nzf};
            double cm[9];
            skew(q.pimu, cm);
            double varw[9];
            m3_sandwich(RextR, q.bv, varw);
            const double J[6] = {pwd[0] - P.c[0], pwd[1] - P.c[1], pwd[2] - P.c[2], -P.n[0], -P.n[1], -P.n[2]};
            const double sigma_l = plane_sigma(P.pv, J);
            double vn[3];
            m3t_vec(varw, nv, vn);
            const double nvn = vn[0] * nv[0] + vn[1] * nv[1] + vn[2] * nv[2];
            const double ri = 1.0 / (sigma_l + nvn);
            o_rinv[i] = ri;
            double T1[9], A[3];
            m3_mul_bt(cm, Rm, T1);  // crossmat * R^T
            m3_vec(T1, nv, A);
            const double H[6] = {A[0], A[1], A[2], nv[0], nv[1], nv[2]};
            const double meas = -(double)dis;
            int k = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const double hr = H[r] * ri;
#pragma unroll
                for (int c = r; c < 6; c++) acc[k++] += hr * H[c];   // (H^T R^-1) H is symmetric up to rounding of hr*H[c] vs hc*H[r]
                acc[21 + r] += hr * meas;
            }
            acc[27] += 1.0;
            acc[28] += fabs((double)dis);
        }
}

// One wavefront per block (n/64 blocks: a down-sampled scan is only ~8k points, so small blocks are what spreads it over the CUs).
// Block sums go through an LDS transpose (lane k adds column k in lane order: fixed order, deterministic); the last block to
// finish adds the per-block partials in block order and writes the 48-double result straight into pinned host memory.
template <bool COOP>   // COOP: the leaf lists of non-planar roots are matched by the whole wavefront (match_trees_coop); false: the lane-by-lane walk (node ids beyond 26 bits, IMMESH_MATCH_SEQ)
__global__ __launch_bounds__(64) void residual_kernel(RegMapDev m, RegIterArgs a, RegState* rs, const float* __restrict__ pts, int n,
                                                       double* __restrict__ partials, unsigned int* __restrict__ done_counter, double* __restrict__ out48,
                                                       double* __restrict__ reg_out, double ticket,
                                                       int8_t* __restrict__ o_match, int32_t* __restrict__ o_node,
                                                       float* __restrict__ o_dis, double* __restrict__ o_rinv, double* __restrict__ o_normal) {
    __builtin_amdgcn_s_setprio(3);   // the pose chain: issue ahead of the mesher's waves sharing the SIMD
    const bool from_dev = a.mode == REG_MODE_SUMS && a.it > 0;   // (modes: HOST = one pass, sums to the host; SUMS = sharded map, sums to device memory for the in-stream all-reduce)
    if (from_dev && rs->done) return;   // the iterated update stopped with an earlier pass (the passes of a scan are enqueued up front)
    ScanParams sp = a.sp;               // constants always by value; the iterate-dependent part from the device when this is a later pass
    if (from_dev) {
#pragma unroll
        for (int k = 0; k < 9; k++) { sp.R[k] = rs->sp.R[k]; sp.RextR[k] = rs->sp.RextR[k]; sp.rot_var[k] = rs->sp.rot_var[k]; sp.t_var[k] = rs->sp.t_var[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) sp.t[k] = rs->sp.t[k];
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long tprev = sp.dbg ? __builtin_readcyclecounter() : 0;
    if (sp.dbg && threadIdx.x == 0) { atomicAdd(&sp.dbg[6], 1ull); if (blockIdx.x == 0) atomicAdd(&sp.dbg[7], 1ull); }
    double acc[RES_NR];
#pragma unroll
    for (int k = 0; k < RES_NR; k++) acc[k] = 0;
    __shared__ double red[RES_NR][65];
    unsigned int* const coop_wl = (unsigned int*)&red[0][0];   // (the transpose buffer is idle until the points are done)
    {
        PointPrep q = {};
        q.key = PREP_NO_KEY; q.root = -1;
        if (i < n) residual_prep(sp, pts, i, q);
        residual_pass<COOP>(m, sp, sp.R, sp.t, sp.RextR, q, i, acc, tprev, o_match, o_node, o_dis, o_rinv, o_normal, i < n, coop_wl);
    }
    RDBG(3);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __shared__ int s_last;
    const int lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < RES_NR; k++) red[k][lane] = acc[k];
    __syncthreads();
    {
        const int k = lane & 31, half = lane >> 5;
        double ssum = 0;
        for (int j = 0; j < 32; j++) ssum += red[k][half * 32 + j];
        ssum += __shfl_xor(ssum, 32, 64);
        // write-through (sc1) stores + a drained store queue publish the partials; no agent-scope release fence (a ~3 us L2 write-back) needed
        if (lane < RES_NR) __hip_atomic_store((unsigned long long*)&partials[(size_t)blockIdx.x * RES_NR + lane], (unsigned long long)__double_as_longlong(ssum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) s_last = (atomicAdd(done_counter, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    RDBG(4);
    if (!s_last) return;
    // final sum over blocks: 64 lanes = 32 values x 2 interleaved halves of the block list, 32 independent (L2-served, ~900-cycle) loads in
    // flight per lane; each half is added in ascending block order and the halves are combined last -- a fixed order, so the result is deterministic
    {
        const int k = lane & 31;
        const unsigned int nb = gridDim.x;
        double tot = 0;
        for (unsigned int b0 = (unsigned int)(lane >> 5); b0 < nb; b0 += 64) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; u++) {
                const unsigned int b = b0 + 2u * u;
                v[u] = b < nb ? __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)&partials[(size_t)b * RES_NR + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 32; u++) tot += v[u];
        }
        tot += __shfl_xor(tot, 32, 64);
        red[0][lane] = tot;
    }
    __syncthreads();
    double v48 = 0;   // host layout of the sums: 36 HTH (row-major 6x6) + 6 HTz + n_match + sum|dis| + n_plane_tests + n_extra_probe
    if (lane < 36) { const int r = lane / 6, c = lane % 6; v48 = red[0][r <= c ? sym21_index(r, c) : sym21_index(c, r)]; }
    else if (lane < 42) v48 = red[0][21 + (lane - 36)];
    else if (lane < 46) v48 = red[0][27 + (lane - 42)];
    if (lane == 0) *done_counter = 0;
    // pinned, device-mapped host memory (HOST) / device memory ahead of an in-stream all-reduce (SUMS)
    if (lane < RES_NV - 1) __hip_atomic_store((unsigned long long*)&out48[lane], (unsigned long long)__double_as_longlong(v48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // slot 47 is the completion ticket the host polls: a release store, issued after the 47 value stores of this (single) wavefront drained
    RDBG(5);
    if (lane == 0 && a.mode == REG_MODE_HOST) __hip_atomic_store(&out48[RES_NV - 1], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole iterated update of a scan as ONE launch (Voxel_mapping::lio_state_estimation, src/voxel_mapping.cpp:1284-1652).
// Round 2 enqueued one residual_kernel per EKF iteration and let the LAST block of each launch add the block partials and run the 18-state
// update: per pass ~9 us of point work and ~15 us of serial tail (arrive atomic, three dependent batches of partial loads, the update, the
// publication of the new pose, the relaunch).  Here the grid stays resident for all passes and the tail is an ALL-GATHER instead of a hand-over:
//   * a block = 4 wavefronts = 256 points per pass (tiles strided over the grid); per-wavefront LDS transpose sums, combined in wave order;
//   * the block's 32 partial sums go out as write-through stores into this pass's slots of a buffer that holds a SENTINEL (a NaN bit pattern
//     no computation produces) everywhere else: the data is its own flag -- no arrive counter, no release fence, no ordering between words;
//   * wavefront 0 of EVERY block polls all slots of the pass (coherent loads, retried until no sentinel is left), adds them in block order and
//     runs the update itself.  Every block thereby holds the same iterate bit for bit (same values, same order, same instructions) and goes on
//     to the next pass without waiting for anybody's publication.  Block 0 also writes what leaves the kernel (posterior for the map update /
//     the host).
//   * slots are per pass and per scan parity; a launch re-arms the OTHER parity's slots for the next scan.
// No deadlock between two contexts: the grid is capped at HALF of what the device holds resident (immesh_ctx::rp_max_blocks, from the occupancy
// query at create) and waits for nothing queued behind it.  More contexts than that, or a device somebody else has filled, could still leave
// blocks undispatched while the resident ones wait for their slots: the gather is therefore BOUNDED (RP_SPIN_TICKS of the 100 MHz real-time
// counter).  A block that runs out of patience publishes the abort word; every block looks at it in every poll, nobody runs the epilogue, the
// result block says "aborted" (passes = -1) and the host registers the scan with the per-pass chain instead (residual_kernel -> ekf_step_kernel,
// which needs no co-residency).  A pass-0 abort is seen by every block before it can stop (the loop runs at least two passes).
// The update itself (rp_update) is the algebra of ekf_host.hpp / the reference regrouped for latency: with K1 = [X; T X] (X = (H^T R^-1 H +
// P11^-1)^-1, T = P21 P11^-1) the solution is K1 (H^T z - H^T H vec6) + vec -- G = K1 H^T H is never formed -- and the posterior covariance
// P - K1 (H^T H P[0:6,:]) is only evaluated by the pass that stops the loop.
// ---------------------------------------------------------------------------------------------------------------------
#define RP_MAX_BLOCKS 128
#define RP_SPIN_TICKS 100000000ull   /* 1 s */
#define RP_SENTINEL 0x7FF8DEADBEEF0001ull
static_assert(COOP_WORDS * 4 <= RES_NR * 65 * 8, "the cooperative matcher's scratch lives in a wavefront's reduction transpose buffer");
struct RpShared {
    double red[4][RES_NR][65];   // per-wavefront transposes
    double wsum[4][RES_NR];
    double pc[108];              // [0,36) P11^-1, [36,108) T = P21 P11^-1
    double st[24];               // the iterate (identical in every block)
    double prior[24];            // the prior state (the update's prior [-] iterate reads it per lane)
    double sums[48];             // this pass: 36 HTH (row-major) + 6 HTz + n_match + sum|dis| + n_plane_tests + n_extra_probe
    double X[36], y[6], sol[18];
    double M[108], XM[108];      // stop pass: H^T H P[0:6,:] and X times it
    double tot[4];               // cumulative over the passes: n_plane_tests, n_extra_probe, passes, n_match
    int rematch, stop;
};
IMD void lds_wave_sync() {   // make this wavefront's LDS writes visible to its own later reads (the other wavefronts of the block are parked at a barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// vec6 = (prior [-] iterate)[0:6] of the pass (rotation difference through Log, translation difference): every lane of the wavefront computes the
// same six values.  It only needs the iterate, so the caller evaluates it while the other blocks' partial sums are still in flight.
__device__ __forceinline__ void rp_prior_minus_state(const RpShared& S, double* __restrict__ R12, double* __restrict__ vec6) {
#pragma unroll
    for (int k = 0; k < 12; k++) R12[k] = S.st[k];
    double Rt[9], rotd[9], pR[9];
#pragma unroll
    for (int k = 0; k < 9; k++) pR[k] = S.prior[k];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j2 = 0; j2 < 3; j2++) Rt[i * 3 + j2] = R12[j2 * 3 + i];
    m3_mul(Rt, pR, rotd);
    dev_so3_log(rotd, vec6);
#pragma unroll
    for (int k = 0; k < 3; k++) vec6[3 + k] = S.prior[9 + k] - R12[9 + k];
}
// The 18-state update on one wavefront; S.sums / S.st / S.rematch / S.tot in, S.st / S.rematch / S.tot / S.stop / S.X out.  Laid out for the latency of a
// lone wavefront: nothing is handed over through LDS inside the update.  Lanes 6..11 own one column of X = (H^T R^-1 H + P11^-1)^-1 each (the
// Gauss-Jordan leaves it in their registers; X is symmetric, so the column serves as the row); w = H^T z - H^T H vec6 and y = X w travel between
// lanes as v_readlane broadcasts; lane l < 18 then owns component l of the solution K1 w + vec (rows 6..17: T y, T row l - 6 from LDS) and lanes
// 6..17 add theirs to their own state component; only the rotation / translation part (solution[0:6] = y + vec6, uniform) is computed by every lane.
__device__ __forceinline__ void rp_update(RpShared& S, const RegIterArgs& a, const int it, const int lane, const bool writer, const double* __restrict__ R12,
                                          const double* __restrict__ vec6, unsigned long long* dbg) {
    unsigned long long tk = dbg ? __builtin_readcyclecounter() : 0;
#define EDBG(k) do { if (dbg) { const unsigned long long _t = __builtin_readcyclecounter(); if (lane == 0 && writer) dbg[40 + (k)] += _t - tk; tk = _t; } } while (0)
    const int own = lane < 18 ? lane : 0;                 // the solution component of this lane
    const int row = (lane >= 6 && lane < 12) ? lane - 6 : 0;
    // ---- loads that depend on nothing computed here: this lane's row of H^T H / H^T z (w), of T (solution rows 6..17), its state / prior component
    double hrow[6], trow[6];
#pragma unroll
    for (int c = 0; c < 6; c++) { hrow[c] = S.sums[row * 6 + c]; trow[c] = S.pc[36 + (own >= 6 ? own - 6 : 0) * 6 + c]; }
    const double hz = S.sums[36 + row];
    const double st_own = S.st[own + 6], prior_own = S.prior[own + 6];     // (components 3..17 of vec are plain differences)
    // ---- w = H^T z - H^T H vec6 (lanes 6..11), broadcast
    double w[6];
    {
        double sacc = 0;
#pragma unroll
        for (int c = 0; c < 6; c++) sacc += hrow[c] * vec6[c];
        const double wl = hz - sacc;
#pragma unroll
        for (int c = 0; c < 6; c++) w[c] = rl_d(wl, 6 + c);
    }
    // ---- X: lane j < 12 holds column j of [S | I]; Gauss-Jordan without pivoting (symmetric positive definite)
    double c6[6];
    {
        const int j = lane < 12 ? lane : 0;
#pragma unroll
        for (int r = 0; r < 6; r++) c6[r] = j < 6 ? S.sums[r * 6 + j] + S.pc[r * 6 + j] : ((r == j - 6) ? 1.0 : 0.0);
#pragma unroll
        for (int col = 0; col < 6; col++) {
            const double d = rl_d(c6[col], col);
            double f[6];
#pragma unroll
            for (int r = 0; r < 6; r++) f[r] = rl_d(c6[r], col);
            c6[col] = c6[col] * fast_rcp(d);
#pragma unroll
            for (int r = 0; r < 6; r++) if (r != col) c6[r] -= f[r] * c6[col];
        }
        if (lane >= 6 && lane < 12) {
#pragma unroll
            for (int r = 0; r < 6; r++) S.X[r * 6 + (lane - 6)] = c6[r];   // (rp_finish reads it behind a barrier)
        }
    }
    EDBG(0);
    // ---- y = X w (lanes 6..11: column = row), broadcast; solution component of this lane
    double y[6];
    {
        double yl = 0;
#pragma unroll
        for (int c = 0; c < 6; c++) yl += c6[c] * w[c];
#pragma unroll
        for (int c = 0; c < 6; c++) y[c] = rl_d(yl, 6 + c);
    }
    double sol6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) sol6[k] = y[k] + vec6[k];
    double sol_own;
    {
        double sacc = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) sacc += trow[q] * y[q];
        sol_own = sacc + (prior_own - st_own);
    }
    EDBG(2);
    // ---- state += solution, stop rule (voxel_mapping.cpp:1600-1650)
    double Rn[9];
    {
        double E[9];
        dev_so3_exp(sol6[0], sol6[1], sol6[2], E);
        m3_mul(R12, E, Rn);
    }
    const double rn = sqrt(sol6[0] * sol6[0] + sol6[1] * sol6[1] + sol6[2] * sol6[2]);
    const double tn = sqrt(sol6[3] * sol6[3] + sol6[4] * sol6[4] + sol6[5] * sol6[5]);
    const bool converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
    int rematch = S.rematch;
    if (converged || ((rematch == 0) && (it == (a.max_iter - 2)))) rematch++;
    const bool stop = rematch >= 2 || (it == a.max_iter - 1);
    const double t_tests = S.tot[0] + S.sums[44], t_extra = S.tot[1] + S.sums[45], t_pass = S.tot[2] + 1.0, t_match = S.tot[3] + S.sums[42];
    if (lane >= 6 && lane < 18) S.st[lane + 6] = st_own + sol_own;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) S.st[k] = Rn[k];
#pragma unroll
        for (int k = 0; k < 3; k++) S.st[9 + k] = R12[9 + k] + sol6[3 + k];
        S.rematch = rematch; S.stop = stop ? 1 : 0; S.tot[0] = t_tests; S.tot[1] = t_extra; S.tot[2] = t_pass; S.tot[3] = t_match;
    }
    EDBG(3);
    if (dbg && lane == 0 && writer) dbg[47] += 1;
#undef EDBG
}

IMD float4 transform_value(const double* extR, const double* extT, const double* R, const double* t, const float4 v);
IMD void point_var_point(const RegMapDev& m, const ScanParams& sp, const float* __restrict__ pts, const int i, const int stride, const int mode,
                         double* __restrict__ pt_data, unsigned long long* __restrict__ sort_key, uint32_t* __restrict__ slot_out, int32_t* __restrict__ pt_next,
                         const unsigned long long hint_key = PREP_NO_KEY, const unsigned int hint_slot = 0, const int hint_root = -1);

// The pass that stops the loop, all 256 threads: the posterior covariance is P - K1 (H^T H P[0:6,:]); M = H^T H P6 and XM = X M (rows 0..5 of the
// correction; rows 6..17 = T XM) are left in LDS.  Every block does this when the map update's preparation runs as the epilogue (it propagates
// the posterior 3 x 3 rotation / translation blocks), block 0 always.
__device__ __forceinline__ void rp_posterior(RpShared& S, const RegIterArgs& a) {
    const int tid = threadIdx.x;
    const double* cov = a.mat;
    if (tid < 108) {
        const int r = tid / 18, c = tid % 18;
        double sacc = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) sacc += S.sums[r * 6 + k] * cov[k * 18 + c];
        S.M[tid] = sacc;
    }
    __syncthreads();
    if (tid < 108) {
        const int r = tid / 18, c = tid % 18;
        double sacc = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) sacc += S.X[r * 6 + k] * S.M[k * 18 + c];
        S.XM[tid] = sacc;
    }
    __syncthreads();
}
// block 0, behind rp_posterior: posterior state / covariance for the host (pinned memory), posterior pose for whatever is queued behind this
// launch (RegState::sp, read after the kernel boundary), ticket.
__device__ __forceinline__ void rp_finish(RpShared& S, const RegIterArgs& a, RegState* __restrict__ rs, double* __restrict__ reg_out, const double ticket) {
    const int tid = threadIdx.x;
    const double* cov = a.mat;
    for (int e = tid; e < 324; e += 256) {
        const int r = e / 18, c = e % 18;
        double upd;
        if (r < 6) upd = S.XM[r * 18 + c];
        else {
            upd = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) upd += S.pc[36 + (r - 6) * 6 + k] * S.XM[k * 18 + c];
        }
        const double sacc = cov[e] - upd;
        __hip_atomic_store((unsigned long long*)&reg_out[24 + e], (unsigned long long)__double_as_longlong(sacc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (r < 3 && c < 3) rs->sp.rot_var[r * 3 + c] = sacc;                       // the map update propagates the POSTERIOR covariance blocks
        if (r >= 3 && r < 6 && c >= 3 && c < 6) rs->sp.t_var[(r - 3) * 3 + (c - 3)] = sacc;
    }
    if (tid < 24) {
        const double v = S.st[tid];
        rs->st[tid] = v;
        __hip_atomic_store((unsigned long long*)&reg_out[tid], (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tid < 9) {
            rs->sp.R[tid] = v;
            const int r = tid / 3, c = tid % 3;
            rs->sp.RextR[tid] = S.st[r * 3 + 0] * a.sp.extR[0 * 3 + c] + S.st[r * 3 + 1] * a.sp.extR[1 * 3 + c] + S.st[r * 3 + 2] * a.sp.extR[2 * 3 + c];
        } else if (tid < 12) rs->sp.t[tid - 9] = v;
    }
    if (tid == 64) {
        rs->rematch = S.rematch; rs->done = 1;
        const double o[6] = {S.tot[2], S.sums[42], S.sums[43], S.tot[0], S.tot[1], S.tot[3]};
        for (int k = 0; k < 6; k++) __hip_atomic_store((unsigned long long*)&reg_out[348 + k], (unsigned long long)__double_as_longlong(o[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&reg_out[REG_OUT_DOUBLES - 1], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <bool COOP>   // (see residual_kernel)
__global__ __launch_bounds__(256) void residual_persistent_kernel(RegMapDev m, RegIterArgs a, RegState* rs, const float* __restrict__ pts, int n,
                                                                   double* __restrict__ slots, double* __restrict__ slots_next, int n_slots_next,
                                                                   int32_t* __restrict__ host_counters,
                                                                   double* __restrict__ reg_out, double ticket,
                                                                   int8_t* __restrict__ o_match, int32_t* __restrict__ o_node,
                                                                   float* __restrict__ o_dis, double* __restrict__ o_rinv, double* __restrict__ o_normal, RpEpilogue ep) {
    __shared__ RpShared S;
    __builtin_amdgcn_s_setprio(3);   // the pose chain: issue ahead of the mesher's waves sharing the SIMD
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ScanParams& sp = a.sp;     // per-scan constants stay kernel arguments (scalar loads); only the iterate changes between passes
    unsigned long long tprev = sp.dbg ? __builtin_readcyclecounter() : 0;
    const unsigned long long t_entry = sp.dbg ? __builtin_amdgcn_s_memrealtime() : 0;   // (trace: [4] of the pass-0 record = kernel entry, [5] = block 0 finished)
    // re-arm the other parity's slots for the next scan (fire-and-forget: the kernel boundary publishes them)
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n_slots_next; e += gridDim.x * 256) ((unsigned long long*)slots_next)[e] = RP_SENTINEL;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ((unsigned long long*)slots_next)[RP_TAIL_WORD] = RP_SENTINEL; ((unsigned long long*)slots_next)[RP_ABORT_WORD] = RP_SENTINEL;
        // "this scan's registration is running": what the next cloud's VoxelGrid waits for (ds_gate_kernel) so that it shares the chip with this launch,
        // which does not mind, and not with the map update in front of it, which does
        __hip_atomic_store(&rs->started, (int)(long long)ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the previous scan's map update left its tail to this launch (a.pad): nothing of it is read by the passes below.  It gets a workgroup of its
    // own (the launcher adds one): inside a working block it delayed that block's first partial sums, i.e. everybody's first gather
    const int G = (int)gridDim.x - 1;
    if ((int)blockIdx.x == G) {
        if (a.pad & 1) map_update_tail(m, host_counters);
        // the epilogue below works on the counters the tail resets: its completion is published like a block partial (release: fence, then the
        // word) and every block's first gather waits for it along with the partials
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) dev_publish(&slots[RP_TAIL_WORD], 1.0);
        return;
    }
    if (wv == 0) {
        // ---- per-scan constants of the gain: a.mat = the prior covariance P (18 x 18); lane j < 12 holds column j of [P11 | I].  Everything read
        //      from the argument block is requested up front (one round trip to wherever the runtime keeps kernel arguments, not three)
        double c6[6], trw[2][6];
        const int j = lane < 12 ? lane : 0;
#pragma unroll
        for (int r = 0; r < 6; r++) c6[r] = j < 6 ? a.mat[r * 18 + j] : ((r == j - 6) ? 1.0 : 0.0);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int e = lane + 64 * h, i = e < 72 ? e / 6 : 0;
#pragma unroll
            for (int k = 0; k < 6; k++) trw[h][k] = a.mat[(6 + i) * 18 + k];
        }
        const double st_l = a.st[lane < 24 ? lane : 0], prior_l = a.prior[lane < 24 ? lane : 0];
#pragma unroll
        for (int col = 0; col < 6; col++) {
            const double d = rl_d(c6[col], col);
            double f[6];
#pragma unroll
            for (int r = 0; r < 6; r++) f[r] = rl_d(c6[r], col);
            c6[col] = c6[col] * fast_rcp(d);
#pragma unroll
            for (int r = 0; r < 6; r++) if (r != col) c6[r] -= f[r] * c6[col];
        }
        if (lane >= 6 && lane < 12) {
#pragma unroll
            for (int r = 0; r < 6; r++) S.pc[r * 6 + (lane - 6)] = c6[r];
        }
        lds_wave_sync();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int e = lane + 64 * h;
            if (e < 72) {
                const int q = e % 6;
                double sacc = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) sacc += trw[h][k] * S.pc[k * 6 + q];
                S.pc[36 + e] = sacc;
            }
        }
        if (lane < 24) { S.st[lane] = st_l; S.prior[lane] = prior_l; }
        if (lane == 0) { S.rematch = 0; S.stop = 0; S.tot[0] = S.tot[1] = S.tot[2] = S.tot[3] = 0.0; }
    }
    __syncthreads();
    const int ntiles = (n + 63) / 64;
    const bool one_tile = ntiles <= G * 4;
    PointPrep prep = {};
    prep.key = PREP_NO_KEY; prep.root = -1;
    for (int it = 0; it < a.max_iter; it++) {
        // IMMESH_DEBUG trace (s_memrealtime, 100 MHz) per (pass, block): [0] pass start [1] block partials out [2] all partials in [3] update done
        unsigned long long* const tr = (sp.dbg && threadIdx.x == 0 && it < 8) ? sp.dbg + (64 + 16384 * 8) + ((size_t)it * 512 + blockIdx.x) * 8 : nullptr;
        if (tr) {
            tr[0] = __builtin_amdgcn_s_memrealtime();
            if (it == 0) tr[4] = t_entry;
            // (pass-1 record, spare words: WHERE the block runs -- HW_ID (CU / SE / wave slot) and XCC_ID -- for the "which workgroups are placed late" question)
            if (it == 1) { tr[4] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); tr[5] = (unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); }
        }
        if (sp.dbg && threadIdx.x == 0) { atomicAdd(&sp.dbg[6], 1ull); if (blockIdx.x == 0) atomicAdd(&sp.dbg[7], 1ull); }   // (the RDBG phase sums are wavefront 0's)
        // the iterate of this pass: wave-uniform, through readfirstlane into scalar registers
        double Rm[9], tv[3], RextR[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Rm[k] = uni_d(S.st[k]);
#pragma unroll
        for (int k = 0; k < 3; k++) tv[k] = uni_d(S.st[9 + k]);
        {
            double t9[9];
            m3_mul(Rm, sp.extR, t9);
#pragma unroll
            for (int k = 0; k < 9; k++) RextR[k] = uni_d(t9[k]);
        }
        if (sp.dbg) tprev = __builtin_readcyclecounter();
        double acc[RES_NR];
#pragma unroll
        for (int k = 0; k < RES_NR; k++) acc[k] = 0;
        for (int tile = blockIdx.x * 4 + wv; tile < ntiles; tile += G * 4) {
            const int i = tile * 64 + lane;
            if (i < n && !(one_tile && it > 0)) residual_prep(sp, pts, i, prep);   // one tile per wavefront (any down-sampled scan): the pass-independent part is computed once
            residual_pass<COOP>(m, sp, Rm, tv, RextR, prep, i, acc, tprev, o_match, o_node, o_dis, o_rinv, o_normal, i < n, (unsigned int*)&S.red[wv][0][0]);   // (every lane: the leaf lists are matched by the whole wavefront)
        }
        RDBG(3);
        // ---- per-wavefront sums (LDS transpose: lane k adds column k in lane order -- a fixed order), combined in wave order
#pragma unroll
        for (int k = 0; k < RES_NR; k++) S.red[wv][k][lane] = acc[k];
        lds_wave_sync();
        {
            const int k = lane & 31, half = lane >> 5;
            double ssum = 0;
            for (int j = 0; j < 32; j++) ssum += S.red[wv][k][half * 32 + j];
            ssum += __shfl_xor(ssum, 32, 64);
            if (lane < RES_NR) S.wsum[wv][lane] = ssum;
        }
        __syncthreads();
        double* const pass_slots = slots + (size_t)it * RP_MAX_BLOCKS * RES_NR;
        if (wv == 0) {
            if (lane < RES_NR) dev_publish(&pass_slots[(size_t)blockIdx.x * RES_NR + lane], ((S.wsum[0][lane] + S.wsum[1][lane]) + S.wsum[2][lane]) + S.wsum[3][lane]);
            RDBG(4);
            if (tr) tr[1] = __builtin_amdgcn_s_memrealtime();
            // ---- all-gather: poll every block's slots of this pass until no sentinel is left; blocks are added in ascending order, two
            // interleaved halves of the list combined last (fixed order: every block computes the same bits)
            const int k = lane & 31;
            double tot = 0;
            double R12[12], vec6[6];
            bool have_vec = false;
            __builtin_amdgcn_s_setprio(1);
            bool aborted = false;
            unsigned long long spin_t0 = 0;
            unsigned int spins = 0;
            for (int b0 = lane >> 5; b0 < G && !aborted; b0 += 64) {
                double v[32];
                for (;;) {
                    unsigned long long bits[32];
#pragma unroll
                    for (int u = 0; u < 32; u++) {
                        const int b = b0 + 2 * u;
                        bits[u] = 0;
                        if (b < G) bits[u] = __hip_atomic_load((const unsigned long long*)&pass_slots[(size_t)b * RES_NR + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    unsigned long long tailbits = 0;
                    if (it == 0) tailbits = __hip_atomic_load((const unsigned long long*)&slots[RP_TAIL_WORD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long abortbits = __hip_atomic_load((const unsigned long long*)&slots[RP_ABORT_WORD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // the part of the update that needs only the iterate runs while the first round of loads is in flight
                    if (!have_vec) { rp_prior_minus_state(S, R12, vec6); have_vec = true; }
                    bool ok = tailbits != RP_SENTINEL;
#pragma unroll
                    for (int u = 0; u < 32; u++) { ok = ok && bits[u] != RP_SENTINEL; v[u] = __longlong_as_double((long long)bits[u]); }
                    // (a.pad & 2: the test hook -- every block gives up in its first poll, as if the grid had not become resident)
                    if (__any(abortbits != RP_SENTINEL) || (a.pad & 2)) { aborted = true; break; }
                    if (__all(ok)) break;
                    // bounded: the slots of a block that is not resident never arrive (see the header comment)
                    if ((++spins & 63u) == 0) {
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                        if (spin_t0 == 0) spin_t0 = now;
                        else if (now - spin_t0 > RP_SPIN_TICKS) { aborted = true; break; }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int u = 0; u < 32; u++) tot += v[u];
            }
            __builtin_amdgcn_s_setprio(3);
            if (__any(aborted)) {   // (wave-uniform: the lanes have reconverged behind the loop)
                if (lane == 0) { dev_publish(&slots[RP_ABORT_WORD], 1.0); S.stop = 2; }
            } else {
            tot += __shfl_xor(tot, 32, 64);
            if (tr) tr[2] = __builtin_amdgcn_s_memrealtime();
            // host layout of the sums: 36 HTH (row-major 6x6) + 6 HTz + n_match + sum|dis| + n_plane_tests + n_extra_probe
            S.wsum[0][lane & 31] = tot;   // (both halves hold the same totals)
            lds_wave_sync();
            double v48 = 0;
            if (lane < 36) { const int r = lane / 6, c = lane % 6; v48 = S.wsum[0][r <= c ? sym21_index(r, c) : sym21_index(c, r)]; }
            else if (lane < 42) v48 = S.wsum[0][21 + (lane - 36)];
            else if (lane < 46) v48 = S.wsum[0][27 + (lane - 42)];
            if (lane < 46) S.sums[lane] = v48;
            lds_wave_sync();
            rp_update(S, a, it, lane, blockIdx.x == 0, R12, vec6, sp.dbg);
            if (tr) tr[3] = __builtin_amdgcn_s_memrealtime();
            RDBG(5);
            }
        }
        __syncthreads();
        if (S.stop == 2) {
            // aborted: nothing of the scan has been written (no epilogue); every aborting block stores the same words
            if (threadIdx.x == 0) {
                __hip_atomic_store((unsigned long long*)&reg_out[348], (unsigned long long)__double_as_longlong(-1.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&reg_out[REG_OUT_DOUBLES - 1], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            break;
        }
        if (S.stop) {
            if (blockIdx.x == 0 || ep.enabled) rp_posterior(S, a);
            if (blockIdx.x == 0) {
                rp_finish(S, a, rs, reg_out, ticket);
                if (sp.dbg && threadIdx.x == 0) sp.dbg[(64 + 16384 * 8) + 5] = __builtin_amdgcn_s_memrealtime();
            }
            if (ep.enabled) {
                // ---- epilogue: map_incremental_grow's per-point preparation with the posterior every block holds (same expressions as rp_finish
                //      leaves in RegState::sp for point_var_kernel), then the full scan into the world frame for the mesher
                unsigned long long* const tre = (sp.dbg && threadIdx.x == 0) ? sp.dbg + (64 + 16384 * 8) + (size_t)blockIdx.x * 8 : nullptr;   // (pass-0 record: [5] epilogue start [6] points prepared [7] end)
                if (tre && blockIdx.x != 0) tre[5] = __builtin_amdgcn_s_memrealtime();
                ScanParams q = a.sp;
#pragma unroll
                for (int k = 0; k < 9; k++) q.R[k] = S.st[k];
#pragma unroll
                for (int k = 0; k < 3; k++) q.t[k] = S.st[9 + k];
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        q.RextR[r * 3 + c] = S.st[r * 3 + 0] * a.sp.extR[0 * 3 + c] + S.st[r * 3 + 1] * a.sp.extR[1 * 3 + c] + S.st[r * 3 + 2] * a.sp.extR[2 * 3 + c];
                        q.rot_var[r * 3 + c] = a.mat[r * 18 + c] - S.XM[r * 18 + c];
                        q.t_var[r * 3 + c] = a.mat[(3 + r) * 18 + (3 + c)] - S.XM[(3 + r) * 18 + (3 + c)];
                    }
                // (the transform's loads are requested first and consumed last: ~12 points per thread, one memory latency instead of twelve)
                const float4* const raw4 = (const float4*)ep.raw;
                float4* const world4 = (float4*)ep.world;
                const int tstride = G * 256, t0i = blockIdx.x * 256 + threadIdx.x;
                constexpr int TB = 16;
                float4 rv[TB];
                if (raw4) {
#pragma unroll
                    for (int u = 0; u < TB; u++) { const int i = t0i + u * tstride; if (i < ep.n_raw) rv[u] = raw4[i]; }
                }
                // (one tile per wavefront: this thread's point of the passes is the first point of this loop, and `prep` still names its root voxel)
                for (int i = t0i; i < n; i += tstride) {
                    const bool own = IMMESH_EPI_HINT && one_tile && i == t0i && m.shard_world <= 1;
                    point_var_point(m, q, pts, i, 3, 0, ep.pt_data, ep.sort_key, ep.slot_out, ep.pt_next, own ? prep.key : PREP_NO_KEY, prep.slot, own ? prep.root : -1);
                }
                if (tre) tre[6] = __builtin_amdgcn_s_memrealtime();
                if (raw4) {
#pragma unroll
                    for (int u = 0; u < TB; u++) { const int i = t0i + u * tstride; if (i < ep.n_raw) world4[i] = transform_value(a.sp.extR, a.sp.extT, q.R, q.t, rv[u]); }
                    for (int i = t0i + TB * tstride; i < ep.n_raw; i += tstride) world4[i] = transform_value(a.sp.extR, a.sp.extT, q.R, q.t, raw4[i]);
                }
                if (tre) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tre[7] = __builtin_amdgcn_s_memrealtime(); }
                // No completion protocol here: what consumes these stores in THIS stream sits behind the kernel boundary, and the flags that tell the
                // mesher's stream / the host "the scan is in its world buffer, the input clouds are consumed" are stored by the first workgroup of the
                // next launch (replay_fused_kernel) -- behind the same boundary, so no fence (an L2 write-back + invalidate per block) is needed
            }
            break;
        }
    }
}

// the same update as its own launch: sums48 = the (all-reduced) 48 sums of the pass in device memory
__global__ __launch_bounds__(64) void ekf_step_kernel(RegIterArgs a, RegState* rs, const double* __restrict__ sums48, double* __restrict__ reg_out, double ticket) {
    __shared__ double L[800];
    const int lane = threadIdx.x;
    if (a.it > 0 && rs->done) return;
    if (lane < 46) L[64 + lane] = sums48[lane];
    for (int e = lane; e < EKF_C_DOUBLES; e += 64) {
        double v;
        if (a.it == 0) v = e < 108 ? a.mat[e] : (e < 132 ? a.st[e - 108] : (e < 156 ? a.prior[e - 132] : 0.0));
        else v = e < 36 ? rs->p11inv[e] : (e < 108 ? rs->tmat[e - 36] : (e < 132 ? rs->st[e - 108] : (e < 156 ? rs->prior[e - 132] : (e < 160 ? rs->tot[e - 156] : (double)rs->rematch))));
        L[400 + e] = v;
    }
    if (a.it == 0) {
        for (int e = lane; e < 108; e += 64) { if (e < 36) rs->p11inv[e] = a.mat[e]; else rs->tmat[e - 36] = a.mat[e]; }
        if (lane < 24) rs->prior[lane] = a.prior[lane];
        const unsigned long long* src = (const unsigned long long*)&a.sp;
        unsigned long long* dst = (unsigned long long*)&rs->sp;
        for (int e = lane; e < (int)(sizeof(ScanParams) / 8); e += 64) dst[e] = src[e];
    }
    __syncthreads();
    if (lane == 0) { L[400 + 156] += L[64 + 44]; L[400 + 157] += L[64 + 45]; L[400 + 159] += L[64 + 42]; }
    __syncthreads();
    ekf_step_wave(rs, L + 400, a.mat, L + 64, L + 64 + 36, L[64 + 42], L[64 + 43], L + 128, lane, a.it, a.max_iter, a.sp.extR, reg_out, ticket);
}

// =====================================================================================================================
// map update / build: per-point preparation
// =====================================================================================================================
// mode 0: map_incremental_grow  (var = (R extR) bcov (R extR)^T + (-[p_imu]x) Srot (-[p_imu]x)^T + St, p_imu with the z==0 -> 1e-3 quirk)
// mode 1: voxel_map_init        (var = R bcov R^T + (-[p_lidar]x) Srot (..)^T + St, p_lidar after calcBodyVar's z==0 -> 1e-4 quirk)
// transformLidar of one point of the FULL scan for the mesher (voxel_mapping_common.cpp:709-726): world = R (extR p + extT) + t, f64 compute, f32 store
IMD float4 transform_value(const double* extR, const double* extT, const double* R, const double* t, const float4 v) {
    const double p[3] = {(double)v.x, (double)v.y, (double)v.z};
    double pi[3], pw[3];
    m3_vec(extR, p, pi);
    pi[0] += extT[0]; pi[1] += extT[1]; pi[2] += extT[2];
    m3_vec(R, pi, pw);
    return make_float4((float)(pw[0] + t[0]), (float)(pw[1] + t[1]), (float)(pw[2] + t[2]), v.w);
}
IMD void transform_point(const double* extR, const double* extT, const double* R, const double* t, const float4* __restrict__ raw, float4* __restrict__ world, const int i) {
    world[i] = transform_value(extR, extT, R, t, raw[i]);
}
// one point of the map update's preparation (sp = the pose / covariance blocks to propagate with): world point, covariance, sort key, root voxel
// (found or created), push on the voxel's list of this update.  Shared by point_var_kernel and the epilogue of residual_persistent_kernel.
// hint_key / hint_slot / hint_root (residual_persistent_kernel's epilogue): the root voxel the LAST matcher pass found the point in.  The posterior moves a
// point by a fraction of a millimetre against that pass's iterate, so the key is almost always the same -- then the voxel exists and its slot and root
// node are known: no hash probe and no read of the entry in front of the list push (two dependent round trips less on the pose chain's tail).
IMD void point_var_point(const RegMapDev& m, const ScanParams& sp, const float* __restrict__ pts, const int i, const int stride, const int mode,
                         double* __restrict__ pt_data, unsigned long long* __restrict__ sort_key, uint32_t* __restrict__ slot_out, int32_t* __restrict__ pt_next,
                         const unsigned long long hint_key, const unsigned int hint_slot, const int hint_root) {
    const double p[3] = {(double)pts[(size_t)i * stride + 0], (double)pts[(size_t)i * stride + 1], (double)pts[(size_t)i * stride + 2]};
    double pimu[3], pwd[3];
    m3_vec(sp.extR, p, pimu);
    pimu[0] += sp.extT[0]; pimu[1] += sp.extT[1]; pimu[2] += sp.extT[2];
    m3_vec(sp.R, pimu, pwd);
    pwd[0] += sp.t[0]; pwd[1] += sp.t[1]; pwd[2] += sp.t[2];
    const double pw[3] = {(double)(float)pwd[0], (double)(float)pwd[1], (double)(float)pwd[2]};
    double bcov[9], cm[9], a[9];
    if (mode == 0) {
        double pz[3] = {p[0], p[1], p[2]};
        if (pz[2] == 0) pz[2] = 0.001;
        calc_body_var(pz, sp.dept_err, sp.dvar_beam, bcov);
        double pimu_z[3];
        m3_vec(sp.extR, pz, pimu_z);
        pimu_z[0] += sp.extT[0]; pimu_z[1] += sp.extT[1]; pimu_z[2] += sp.extT[2];
        skew(pimu_z, cm);
        m3_sandwich(sp.RextR, bcov, a);
    } else {
        double pl[3] = {p[0], p[1], p[2]};
        calc_body_var(pl, sp.dept_err, sp.dvar_beam, bcov);
        skew(pl, cm);
        m3_sandwich(sp.R, bcov, a);
    }
    double nc[9], tmp[9], b[9], var[9];
#pragma unroll
    for (int k = 0; k < 9; k++) nc[k] = -cm[k];
    m3_mul(nc, sp.rot_var, tmp);
    m3_mul_bt(tmp, nc, b);
#pragma unroll
    for (int k = 0; k < 9; k++) var[k] = (a[k] + b[k]) + sp.t_var[k];
    double* o = pt_data + (size_t)i * IM_PT_DOUBLES;
    o[0] = pw[0]; o[1] = pw[1]; o[2] = pw[2];
    o[3] = var[0]; o[4] = var[1]; o[5] = var[2]; o[6] = var[4]; o[7] = var[5]; o[8] = var[8];
    // var_contrast key (voxel_mapping.cpp:49): ||diag(var)||, non-negative double -> order-preserving as uint64
    const double key = sqrt(var[0] * var[0] + var[4] * var[4] + var[8] * var[8]);
    sort_key[i] = (unsigned long long)__double_as_longlong(key);
    // root voxel (voxel_mapping.cpp:328-351): key from the float-rounded world point / float voxel size
    int64_t kx[3];
#pragma unroll
    for (int j = 0; j < 3; j++) kx[j] = key_axis(pw[j] / (double)m.voxel_size_f);
    const uint64_t pk = pack_key(kx[0], kx[1], kx[2]);
    if (m.shard_world > 1 && !shard_keeps(m, kx[0], kx[1], kx[2])) { slot_out[i] = 0xFFFFFFFFu; return; }   // another rank's voxel (outside our halo)
    bool created = false;
    const bool hinted = hint_root >= 0 && pk == hint_key;
    const int64_t slot = hinted ? (int64_t)hint_slot : hash_find_or_insert(m, pk, &created);
    if (slot < 0) { m.counters[5] = 5; slot_out[i] = 0xFFFFFFFFu; return; }
    if (created) {
        const float vs = m.voxel_size_f;
        const double c[3] = {(0.5 + (double)kx[0]) * (double)vs, (0.5 + (double)kx[1]) * (double)vs, (0.5 + (double)kx[2]) * (double)vs};
        const int node = node_alloc(m, 0, c, vs / 4, pk, 0);
        m.htab[slot].root = node;
        atomicAdd(&m.counters[6], 1);
    }
    slot_out[i] = (uint32_t)slot;
    if (pt_next) {  // push the point on its root voxel's list for this update; the first point to arrive registers the voxel as touched
        const unsigned long long mine = ((unsigned long long)(unsigned int)m.upd_seq << 32) | (unsigned int)i;
        const unsigned long long old = atomicExch(&m.slot_head[slot], mine);
        const bool first = (unsigned int)(old >> 32) != (unsigned int)m.upd_seq;
        // the touched list grows by ONE atomic per wavefront (the first arrivers of a wavefront share it): ~4 900 returning atomics on one address were
        // the longest dependent step of this function
        const unsigned long long fm = __ballot(first);
        if (first) {
            pt_next[i] = -1;
            const int lane = (int)(threadIdx.x & 63), leader = __builtin_ctzll(fm);
            int base = 0;
            if (lane == leader) base = atomicAdd(&m.counters[7], (int)__popcll(fm));
            base = __shfl(base, leader, 64);
            const int k = base + (int)__popcll(fm & ((1ull << lane) - 1ull));
            // (slot, root node): the replay kernel starts from the node without a second trip through the hash; a root another lane of this launch
            // is still creating reads as -1 here and is looked up there
            m.touched[2 * (size_t)k] = (uint32_t)slot; m.touched[2 * (size_t)k + 1] = hinted ? (uint32_t)hint_root : (uint32_t)m.htab[slot].root;
        }
        else pt_next[i] = (int)(unsigned int)(old & 0xFFFFFFFFull);
    }
}

__global__ __launch_bounds__(256) void point_var_kernel(RegMapDev m, ScanParams sp, const ScanParams* __restrict__ spd, const float* __restrict__ pts, int n, int stride, int mode,
                                                         double* __restrict__ pt_data, unsigned long long* __restrict__ sort_key, uint32_t* __restrict__ slot_out,
                                                         int32_t* __restrict__ pt_next, const float4* __restrict__ raw, float4* __restrict__ world, int n_raw, int nb_pv) {
    if ((int)blockIdx.x >= nb_pv) {
        // the transform of the full scan rides in the same launch: it was a launch of its own on the pose chain
        const int i = ((int)blockIdx.x - nb_pv) * 256 + threadIdx.x;
        if (i >= n_raw) return;
        const ScanParams* q = spd ? spd : &sp;
        double R[9], t[3];
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = q->R[k];
#pragma unroll
        for (int k = 0; k < 3; k++) t[k] = q->t[k];
        transform_point(sp.extR, sp.extT, R, t, raw, world, i);
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (spd) {   // posterior of the scan just registered, left on the device by the in-kernel EKF update
#pragma unroll
        for (int k = 0; k < 9; k++) { sp.R[k] = spd->R[k]; sp.RextR[k] = spd->RextR[k]; sp.rot_var[k] = spd->rot_var[k]; sp.t_var[k] = spd->t_var[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) sp.t[k] = spd->t[k];
    }
    point_var_point(m, sp, pts, i, stride, mode, pt_data, sort_key, slot_out, pt_next);
}

// segment heads of the slot-sorted point list
__global__ void segment_heads_kernel(const uint32_t* __restrict__ sorted_slot, int n, int32_t* __restrict__ seg_start, int32_t* __restrict__ nseg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (sorted_slot[i] == 0xFFFFFFFFu) return;
    if (i == 0 || sorted_slot[i - 1] != sorted_slot[i]) seg_start[atomicAdd(nseg, 1)] = i;
}

// =====================================================================================================================
// wave-cooperative octree maintenance
// =====================================================================================================================
struct WaveCtx { int lane; int64_t* stats; int root; bool shared = false; };  // stats[0] refits, stats[1] refit points; shared: other wavefronts work on the same root (replay_sub_kernel)

// ---------------------------------------------------------------------------------------------------------------------
// OctoTree::init_plane (src/voxel_loc.cpp:47-139) on one wavefront.  The fit is the unit of the map update (one fit used to be 44 k cycles =
// 18 us of a lone wavefront, and the update was three of them in a row), so it is built for latency:
//   * the points are visited through a functor: register-resident (lane i holds points i and i + 64: the settled-root path and every node
//     of <= 128 points) or strided from the chunk pool (larger nodes);
//   * moments -> covariance exactly as the reference (sum / n - c c^T), eigen-decomposition by the checker's own cyclic Jacobi with cheap
//     reciprocals (sym3_eigen_jacobi_fast: same rotations, same eigenvalue positions and eigenvector signs, ~1/4 of the cycles);
//   * the plane covariance sum_i J_i V_i J_i^T (voxel_loc.cpp:80-105) in closed form.  With U = [u_0 u_1 u_2], m1, m2 the two indices other
//     than imin, alpha_k = 1 / (n (ev_min - ev_mk)), dp = p_i - c:  J_i = [A_i; I/n],  A_i = sum_k alpha_k u_mk g_k^T,
//     g_k = (dp . u_min) u_mk + (dp . u_mk) u_min   (this is evecs * F of the reference, F row m = dp^T (u_m u_min^T + u_min u_m^T) / denom_m).
//     Hence with h_k = V_i g_k:   top-left  = sum_kl alpha_k alpha_l (sum_i g_k . h_l) u_mk u_ml^T,
//                                 top-right = (1/n) sum_k alpha_k u_mk (sum_i h_k)^T,      bottom-right = (sum_i V_i) / n^2
//     -- 15 wave sums instead of 21, ~90 flops per point instead of ~350 and no divide inside the loop.  Same value up to rounding.
// ---------------------------------------------------------------------------------------------------------------------
struct PlaneFit { double c[3], ev[3], U[9]; int imin, imax; bool planar; };
// by-value select (a conditional expression on lvalues is a select of ADDRESSES: it would pin the struct to scratch memory)
IMD double sel3(const int i, const double a, const double b, const double c) { return i == 0 ? a : (i == 1 ? b : c); }

template <class FE>
IMD void fit_eigen(FE&& for_each_point, const int n, const float planer_threshold, PlaneFit& f) {
    double s[9];
#pragma unroll
    for (int k = 0; k < 9; k++) s[k] = 0;
    for_each_point([&](const double* q) __attribute__((always_inline)) {
        const double x = q[0], y = q[1], z = q[2];
        s[0] += x; s[1] += y; s[2] += z;
        s[3] += x * x; s[4] += x * y; s[5] += x * z; s[6] += y * y; s[7] += y * z; s[8] += z * z;
    });
#pragma unroll
    for (int k = 0; k < 9; k++) s[k] = wave_sum_fast(s[k]);
    const double dn = (double)n;
    f.c[0] = s[0] / dn; f.c[1] = s[1] / dn; f.c[2] = s[2] / dn;
    double cov[9];
    cov[0] = s[3] / dn - f.c[0] * f.c[0]; cov[1] = s[4] / dn - f.c[0] * f.c[1]; cov[2] = s[5] / dn - f.c[0] * f.c[2];
    cov[3] = cov[1];                      cov[4] = s[6] / dn - f.c[1] * f.c[1]; cov[5] = s[7] / dn - f.c[1] * f.c[2];
    cov[6] = cov[2];                      cov[7] = cov[5];                      cov[8] = s[8] / dn - f.c[2] * f.c[2];
    sym3_eigen_jacobi_fast(cov, f.ev, f.U);
    // minCoeff / maxCoeff: the first extremal index (voxel_loc.cpp:68-69); selects, not indexed reads -- the struct has to stay in registers
    const double e0 = f.ev[0], e1 = f.ev[1], e2 = f.ev[2];
    int imin = 0, imax = 0;
    double emin = e0, emax = e0;
    if (e1 < emin) { imin = 1; emin = e1; }
    if (e2 < emin) { imin = 2; emin = e2; }
    if (e1 > emax) { imax = 1; emax = e1; }
    if (e2 > emax) { imax = 2; emax = e2; }
    f.imin = imin; f.imax = imax;
    f.planar = emin < (double)planer_threshold;
}

template <class FE>
IMD void fit_plane_var(FE&& for_each_point, const int n, const PlaneFit& f, double* pv) {
#pragma clang fp contract(fast)
    const int imin = f.imin, m1 = imin == 0 ? 1 : 0, m2 = imin == 2 ? 1 : 2;
    double um1[3], um2[3], umin[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double u0 = f.U[r * 3 + 0], u1 = f.U[r * 3 + 1], u2 = f.U[r * 3 + 2];
        um1[r] = sel3(m1, u0, u1, u2);
        um2[r] = sel3(m2, u0, u1, u2);
        umin[r] = sel3(imin, u0, u1, u2);
    }
    double acc[15];
#pragma unroll
    for (int k = 0; k < 15; k++) acc[k] = 0;
    for_each_point([&](const double* q) __attribute__((always_inline)) {
#pragma clang fp contract(fast)
        const double dp[3] = {q[0] - f.c[0], q[1] - f.c[1], q[2] - f.c[2]};
        const double a1 = dp[0] * um1[0] + dp[1] * um1[1] + dp[2] * um1[2];
        const double a2 = dp[0] * um2[0] + dp[1] * um2[1] + dp[2] * um2[2];
        const double b = dp[0] * umin[0] + dp[1] * umin[1] + dp[2] * umin[2];
        double g1[3], g2[3], h1[3], h2[3];
#pragma unroll
        for (int r = 0; r < 3; r++) { g1[r] = b * um1[r] + a1 * umin[r]; g2[r] = b * um2[r] + a2 * umin[r]; }
        const double V[9] = {q[3], q[4], q[5], q[4], q[6], q[7], q[5], q[7], q[8]};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            h1[r] = V[r * 3 + 0] * g1[0] + V[r * 3 + 1] * g1[1] + V[r * 3 + 2] * g1[2];
            h2[r] = V[r * 3 + 0] * g2[0] + V[r * 3 + 1] * g2[1] + V[r * 3 + 2] * g2[2];
        }
        acc[0] += g1[0] * h1[0] + g1[1] * h1[1] + g1[2] * h1[2];
        acc[1] += g1[0] * h2[0] + g1[1] * h2[1] + g1[2] * h2[2];
        acc[2] += g2[0] * h2[0] + g2[1] * h2[1] + g2[2] * h2[2];
#pragma unroll
        for (int r = 0; r < 3; r++) { acc[3 + r] += h1[r]; acc[6 + r] += h2[r]; }
#pragma unroll
        for (int r = 0; r < 6; r++) acc[9 + r] += q[3 + r];
    });
#pragma unroll
    for (int k = 0; k < 15; k++) acc[k] = wave_sum_fast(acc[k]);
    const double dn = (double)n;
    const double e0 = f.ev[0], e1 = f.ev[1], e2 = f.ev[2];
    const double evmin = sel3(imin, e0, e1, e2);
    const double al1 = 1.0 / (dn * (evmin - sel3(m1, e0, e1, e2)));
    const double al2 = 1.0 / (dn * (evmin - sel3(m2, e0, e1, e2)));
    const double w11 = al1 * al1 * acc[0], w12 = al1 * al2 * acc[1], w22 = al2 * al2 * acc[2];
    const double inv_n = 1.0 / dn, inv_n2 = inv_n * inv_n;
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = r; c < 6; c++) {
            double v;
            if (c < 3) v = w11 * um1[r] * um1[c] + w12 * (um1[r] * um2[c] + um2[r] * um1[c]) + w22 * um2[r] * um2[c];
            else if (r < 3) v = inv_n * (al1 * um1[r] * acc[3 + (c - 3)] + al2 * um2[r] * acc[6 + (c - 3)]);
            else { const int a = r - 3, b2 = c - 3; v = inv_n2 * acc[9 + (a == 0 ? b2 : (a == 1 ? 2 + b2 : 5))]; }   // V upper triangle: 00 01 02 11 12 22
            pv[k++] = v;
        }
}

// write the fitted plane (or, for a non-planar node, what the reference leaves behind) into the node record; every lane holds the same values
IMD void fit_store(NodeRec& nr, const PlaneFit& f, const double* pv, const int lane) {
    if (f.planar) {
        const int imin = f.imin;
        const double Umin[3] = {sel3(imin, f.U[0], f.U[1], f.U[2]), sel3(imin, f.U[3], f.U[4], f.U[5]), sel3(imin, f.U[6], f.U[7], f.U[8])};
        if (lane < 21) {
            double v = 0;
#pragma unroll
            for (int k = 0; k < 21; k++) if (lane == k) v = pv[k];
            nr.p_var[lane] = v;
        }
        if (lane == 0) {
            const double evmin = sel3(imin, f.ev[0], f.ev[1], f.ev[2]);
            const double evmax = sel3(f.imax, f.ev[0], f.ev[1], f.ev[2]);
#pragma unroll
            for (int k = 0; k < 3; k++) { nr.p_center[k] = f.c[k]; nr.p_normal[k] = Umin[k]; }
            nr.min_eig = (float)evmin;
            nr.radius = (float)sqrt(evmax);
            nr.d = (float)(-(Umin[0] * f.c[0] + Umin[1] * f.c[1] + Umin[2] * f.c[2]));
        }
    } else if (lane == 0) {
        // reference zeroes centre/normal/plane_var before the test and leaves them zero when not planar; only the centre is observable (dump)
#pragma unroll
        for (int k = 0; k < 3; k++) { nr.p_center[k] = f.c[k]; nr.p_normal[k] = 0.0; }
        nr.radius = 0.f;
    }
}

// OctoTree::init_plane for node `nd` holding `n` points in the chunk pool; returns planar?  All lanes get the result.
__device__ __noinline__ bool wave_init_plane(const RegMapDev& m, int nd, int n, const WaveCtx& w) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (w.lane == 0 && w.stats) { atomicAdd((unsigned long long*)&w.stats[0], 1ull); atomicAdd((unsigned long long*)&w.stats[1], (unsigned long long)n); }
    PlaneFit f;
    double pv[21];
    const int lane = w.lane;
    if (n <= 128) {
        // one gather: every lane pulls its (up to) two points into registers; both passes of the fit run from there
        double P0[9], P1[9];
#pragma unroll
        for (int k = 0; k < 9; k++) { P0[k] = 0; P1[k] = 0; }
        if (lane < n) { const double* q = node_point_ptr(m, nd, lane);
#pragma unroll
            for (int k = 0; k < 9; k++) P0[k] = q[k]; }
        if (lane + 64 < n) { const double* q = node_point_ptr(m, nd, lane + 64);
#pragma unroll
            for (int k = 0; k < 9; k++) P1[k] = q[k]; }
        auto fe = [&](auto&& fn) __attribute__((always_inline)) { if (lane < n) fn(P0); if (lane + 64 < n) fn(P1); };
        fit_eigen(fe, n, m.planer_threshold, f);
        if (f.planar) fit_plane_var(fe, n, f, pv);
    } else {
        auto fe = [&](auto&& fn) __attribute__((always_inline)) {
            for (int i = lane; i < n; i += 64) {
                const double* q = node_point_ptr(m, nd, i);
                double P[9];
#pragma unroll
                for (int k = 0; k < 9; k++) P[k] = q[k];
                fn(P);
            }
        };
        fit_eigen(fe, n, m.planer_threshold, f);
        if (f.planar) fit_plane_var(fe, n, f, pv);
    }
    fit_store(m.nodes[nd], f, pv, lane);
    return f.planar;
}

// append one point (9 doubles at src) to node `nd` currently holding `n` points; lanes 0..8 copy.  Returns false on pool exhaustion.
__device__ bool wave_push_point(const RegMapDev& m, int nd, int n, const double* src, const WaveCtx& w) {
    int ok = 1;
    if (w.lane == 0 && (n % IM_CHUNK_PTS) == 0) ok = node_ensure_chunk(m, nd, n / IM_CHUNK_PTS) ? 1 : 0;
    ok = __shfl(ok, 0, 64);
    if (!ok) return false;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (w.lane < IM_PT_DOUBLES) {
        double* dst = node_point_ptr(m, nd, n);
        dst[w.lane] = src[w.lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    return true;
}

IMD int octant_of(const double* q, const double* center) {  // strict > against the centre (voxel_loc.cpp:169-181)
    return 4 * (q[0] > center[0] ? 1 : 0) + 2 * (q[1] > center[1] ? 1 : 0) + (q[2] > center[2] ? 1 : 0);
}
// create child `oct` of `nd` (one lane) -- voxel_loc.cpp:184-190
IMD int make_child(const RegMapDev& m, int nd, int oct) {
    const float ql = m.nodes[nd].quarter;
    const int xyz[3] = {(oct >> 2) & 1, (oct >> 1) & 1, oct & 1};
    double c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = m.nodes[nd].center[k] + (double)((float)(2 * xyz[k] - 1) * ql);
    const int layer = m.nodes[nd].layer + 1;
    const int child = node_alloc(m, layer, c, ql / 2, m.nodes[nd].key, m.nodes[nd].path | (oct << (3 * (layer - 1))));
    m.nodes[nd].child[oct] = child;
    return child;
}

// OctoTree::init_octo_tree + recursive cut_octo_tree (voxel_loc.cpp:141-217) for a node that just exceeded its init size.
// stack: per-wave LDS scratch (>= 40 ints).
__device__ void wave_init_octo_tree(const RegMapDev& m, int node, int* stack, const WaveCtx& w) {
    int sp = 0;
    if (w.lane == 0) stack[0] = node;
    sp = 1;
    while (sp > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int nd = __shfl(stack[sp - 1], 0, 64);  // LDS read by every lane is fine too; shfl keeps it uniform
        sp--;
        const int n = m.nodes[nd].npts;
        const bool planar = wave_init_plane(m, nd, n, w);
        const int layer = m.nodes[nd].layer;
        if (w.lane == 0) {
            const int f0 = m.nodes[nd].flags;
            int f = f0 | NF_INIT;
            f = planar ? (f | NF_PLANE) : (f & ~NF_PLANE);
            node_set_flags(m, w.root, nd, f0, f, w.shared);
            m.nodes[nd].newpts = 0;
        }
        if (planar || layer >= m.max_layer) continue;
        // ---- cut_octo_tree: distribute the retained points to the 8 octants, order-preserving
        const double ctr[3] = {m.nodes[nd].center[0], m.nodes[nd].center[1], m.nodes[nd].center[2]};
        int child_id[8], child_n[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { child_id[k] = m.nodes[nd].child[k]; child_n[k] = child_id[k] >= 0 ? m.nodes[child_id[k]].npts : 0; }
        for (int base = 0; base < n; base += 64) {
            const int i = base + w.lane;
            const double* q = (i < n) ? node_point_ptr(m, nd, i) : nullptr;
            const int oct = (i < n) ? octant_of(q, ctr) : -1;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const unsigned long long mask = __ballot(oct == k);
                if (mask == 0) continue;
                const int cnt = __popcll(mask);
                int cid = child_id[k];
                int ok = 1;
                if (w.lane == 0) {
                    if (cid < 0) cid = make_child(m, nd, k);
                    if (cid < 0) ok = 0;
                    else for (int pos = child_n[k]; pos < child_n[k] + cnt; pos++)
                        if ((pos % IM_CHUNK_PTS) == 0 || pos == child_n[k]) { if (!node_ensure_chunk(m, cid, pos / IM_CHUNK_PTS)) ok = 0; }
                }
                cid = __shfl(cid, 0, 64); ok = __shfl(ok, 0, 64);
                child_id[k] = cid;
                if (!ok) return;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (oct == k) {
                    const int pos = child_n[k] + __popcll(mask & ((1ull << w.lane) - 1ull));
                    double* dst = node_point_ptr(m, cid, pos);
#pragma unroll
                    for (int e = 0; e < IM_PT_DOUBLES; e++) dst[e] = q[e];
                }
                child_n[k] += cnt;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (w.lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (child_id[k] >= 0) { m.nodes[child_id[k]].npts = child_n[k]; m.nodes[child_id[k]].newpts = child_n[k]; }
            node_free_points(m, nd);  // the parent's buffer is never read again (UpdateOctoTree drops it on the next visit, voxel_loc.cpp:263-266)
        }
        // children that exceed their init size are fitted next (depth-first like the reference; order does not affect results)
#pragma unroll
        for (int k = 7; k >= 0; k--)
            if (child_id[k] >= 0 && child_n[k] > m.init_size[min(layer + 1, 4)]) {
                if (w.lane == 0) stack[sp] = child_id[k];
                sp++;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
}

// OctoTree::UpdateOctoTree (voxel_loc.cpp:219-308) for one point; wave-uniform control flow
__device__ void wave_update_point(const RegMapDev& m, int root, const double* src, int* stack, const WaveCtx& w) {
    int nd = root;
    for (int depth = 0; depth < 8; depth++) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int flags = m.nodes[nd].flags;
        const int layer = m.nodes[nd].layer;
        const int n = m.nodes[nd].npts;
        if (!(flags & NF_INIT)) {
            if (!wave_push_point(m, nd, n, src, w)) return;
            if (w.lane == 0) { m.nodes[nd].npts = n + 1; m.nodes[nd].newpts = m.nodes[nd].newpts + 1; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (n + 1 > m.init_size[min(layer, 4)]) wave_init_octo_tree(m, nd, stack, w);
            return;
        }
        if (flags & NF_PLANE) {
            if (!(flags & NF_UPDATE_EN)) return;
            if (!wave_push_point(m, nd, n, src, w)) return;
            int newp = m.nodes[nd].newpts + 1;
            if (w.lane == 0) m.nodes[nd].npts = n + 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (newp > 5) {  // m_update_size_threshold_
                const bool planar = wave_init_plane(m, nd, n + 1, w);
                if (w.lane == 0) node_set_flags(m, w.root, nd, flags, planar ? (flags | NF_PLANE) : (flags & ~NF_PLANE), w.shared);
                newp = 0;
            }
            if (w.lane == 0) {
                m.nodes[nd].newpts = newp;
                if (n + 1 >= m.max_points_size) {
                    m.nodes[nd].flags = m.nodes[nd].flags & ~NF_UPDATE_EN;
                    node_free_points(m, nd);
                    m.nodes[nd].newpts = 0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            return;
        }
        if (layer < m.max_layer) {
            int child;
            if (w.lane == 0) {
                if (n != 0) node_free_points(m, nd);
                const double ctr[3] = {m.nodes[nd].center[0], m.nodes[nd].center[1], m.nodes[nd].center[2]};
                const int oct = octant_of(src, ctr);
                child = m.nodes[nd].child[oct];
                if (child < 0) child = make_child(m, nd, oct);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            child = __shfl(child, 0, 64);
            if (child < 0) return;
            nd = child;
            continue;
        }
        // non-planar node at the last layer (voxel_loc.cpp:289-305)
        if (!(flags & NF_UPDATE_EN)) return;
        if (!wave_push_point(m, nd, n, src, w)) return;
        int newp = m.nodes[nd].newpts + 1;
        if (w.lane == 0) m.nodes[nd].npts = n + 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (newp > 5) {
            const bool planar = wave_init_plane(m, nd, n + 1, w);
            if (w.lane == 0) node_set_flags(m, w.root, nd, flags, planar ? (flags | NF_PLANE) : (flags & ~NF_PLANE), w.shared);
            newp = 0;
        }
        if (w.lane == 0) {
            m.nodes[nd].newpts = newp;
            if (n + 1 > IM_G_MAX_POINTS) { m.nodes[nd].flags = m.nodes[nd].flags & ~NF_UPDATE_EN; node_free_points(m, nd); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        return;
    }
}

// Fast path of the per-voxel replay for the common state of a settled map: the root is an initialised, planar, update-enabled node whose
// retained points fit the inline chunk table.  Same state machine as wave_update_point (OctoTree::UpdateOctoTree, voxel_loc.cpp:240-262), but
//   * the node header lives in registers for the whole batch (the general path re-reads flags / counts / chunk ids from HBM for every point:
//     three to four dependent round trips per point);
//   * a refit that falls due in the middle of the batch is DEFERRED when its outcome is certain: the smallest eigenvalue of the covariance is
//     at most its smallest diagonal entry, so "min_k var_k < min_eigen_value" (with a rounding margin) proves the node stays planar without the
//     eigen-decomposition; only the plane of the LAST refit is observable after the update (the matcher reads the map between scans), and
//     it is fitted once, at the end, over exactly the points the reference's last refit saw (the first n_at_refit retained points).
//     When the bound does not decide, the fit runs on the spot as in the general path.
// Returns the number of points consumed; the node header in memory is current again on return (the general path may continue from it).
__device__ int wave_replay_planar_root(const RegMapDev& m, const int root, const int* order, const int cnt, const double* __restrict__ pt_data, const WaveCtx& w) {
    NodeRec& nd = m.nodes[root];
    int flags = nd.flags;
    const int layer = nd.layer;
    int npts = nd.npts, newp = nd.newpts;
    const int want = NF_INIT | NF_PLANE | NF_UPDATE_EN;
    if (layer == 0 && (flags & want) == (NF_INIT | NF_PLANE)) return cnt;   // a full planar root (m_update_enable_ == false) drops every point
    if ((flags & want) != want || layer != 0 || npts + cnt > IM_INLINE_CHUNKS * IM_CHUNK_PTS) return 0;
    int chunks[IM_INLINE_CHUNKS];
#pragma unroll
    for (int k = 0; k < IM_INLINE_CHUNKS; k++) chunks[k] = nd.chunks[k];
    auto chunk_of = [&](int ci) { int c = chunks[0];
#pragma unroll
        for (int k = 1; k < IM_INLINE_CHUNKS; k++) { const int ck = chunks[k]; c = (ci == k) ? ck : c; }   // (by value: a conditional on lvalues selects ADDRESSES and pins the array to memory)
        return c; };
    // running per-axis sums of the retained points: only needed when a refit falls due inside this batch
    double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
    const bool will_refit = newp + cnt > 5;
    if (will_refit) {
        for (int i = w.lane; i < npts; i += 64) {
            const double* q = m.chunk_data + ((size_t)chunk_of(i / IM_CHUNK_PTS) * IM_CHUNK_PTS + (i % IM_CHUNK_PTS)) * IM_PT_DOUBLES;
#pragma unroll
            for (int a = 0; a < 3; a++) { const double v = q[a]; s1[a] += v; s2[a] += v * v; }
        }
#pragma unroll
        for (int a = 0; a < 3; a++) { s1[a] = wave_sum(s1[a]); s2[a] = wave_sum(s2[a]); }
    }
    WaveCtx wq = w; wq.stats = nullptr;   // refits are counted here (deferred ones included), not by wave_init_plane
    int last_refit_n = 0, n_ref = 0, j = 0;
    long long n_ref_pts = 0;
    bool header_dirty = false;
    auto write_header = [&]() {
        if (w.lane == 0) {
            nd.flags = flags; nd.npts = npts; nd.newpts = newp;
#pragma unroll
            for (int k = 0; k < IM_INLINE_CHUNKS; k++) nd.chunks[k] = chunks[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        header_dirty = false;
    };
    for (; j < cnt; j++) {
        const double* src = pt_data + (size_t)order[j] * IM_PT_DOUBLES;
        const int ci = npts / IM_CHUNK_PTS;
        if ((npts % IM_CHUNK_PTS) == 0 && chunk_of(ci) < 0) {
            int c = -1;
            if (w.lane == 0) c = alloc_chunk(m);
            c = __shfl(c, 0, 64);
            if (c < 0) { write_header(); return j; }   // pool exhausted (flag set by alloc_chunk): stop here, the caller's loop sees the flag path
#pragma unroll
            for (int k = 0; k < IM_INLINE_CHUNKS; k++) if (ci == k) chunks[k] = c;
        }
        const double pv = w.lane < IM_PT_DOUBLES ? src[w.lane] : 0.0;
        if (w.lane < IM_PT_DOUBLES) m.chunk_data[((size_t)chunk_of(ci) * IM_CHUNK_PTS + (npts % IM_CHUNK_PTS)) * IM_PT_DOUBLES + w.lane] = pv;
        if (will_refit) {
#pragma unroll
            for (int a = 0; a < 3; a++) { const double v = __shfl(pv, a, 64); s1[a] += v; s2[a] += v * v; }
        }
        npts++; newp++; header_dirty = true;
        if (newp > 5) {   // m_update_size_threshold_: the refit over all retained points falls due
            n_ref++; n_ref_pts += npts;
            const double dn = (double)npts;
            double vmin = 1e300, mag = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) { const double mu = s1[a] / dn; const double v = s2[a] / dn - mu * mu; vmin = fmin(vmin, v); mag += s2[a] / dn; }
            const double err = 1e-7 + 1e-12 * mag;   // rounding of the one-pass variance (and of the exact fit's own eigenvalue), generously
            newp = 0;
            if (vmin + err < (double)m.planer_threshold) last_refit_n = npts;   // certainly still planar: fit later, over these npts points
            else {
                write_header();   // (wave_init_plane finds the points through the chunk ids in the node record)
                last_refit_n = 0;
                const bool planar = wave_init_plane(m, root, npts, wq);
                if (!planar) {   // the node turns non-planar: later points take the general route (children)
                    if (w.lane == 0) node_set_flags(m, w.root, root, flags, flags & ~NF_PLANE);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    flags &= ~NF_PLANE;
                    if (npts >= m.max_points_size) {   // the fullness test of the same UpdateOctoTree call still runs (voxel_loc.cpp:245-250)
                        flags &= ~NF_UPDATE_EN;
                        write_header();
                        if (w.lane == 0) { node_free_points(m, root); nd.newpts = 0; }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        npts = 0; newp = 0;
                    }
                    j++;
                    break;
                }
            }
        }
        if (npts >= m.max_points_size) {   // the node is full: last fit (if one is pending), then it stops updating and drops its points
            if (last_refit_n) { write_header(); (void)wave_init_plane(m, root, last_refit_n, wq); last_refit_n = 0; }
            flags &= ~NF_UPDATE_EN;
            write_header();
            if (w.lane == 0) { node_free_points(m, root); nd.newpts = 0; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            npts = 0; newp = 0;
            j++;
            break;
        }
    }
    if (header_dirty) write_header();
    if (last_refit_n) (void)wave_init_plane(m, root, last_refit_n, wq);
    if (w.lane == 0 && w.stats && n_ref) { atomicAdd((unsigned long long*)&w.stats[0], (unsigned long long)n_ref); atomicAdd((unsigned long long*)&w.stats[1], (unsigned long long)n_ref_pts); }
    return j;
}

// The same batch treatment for a planar node of ANY size at ANY layer (deep octrees: velodyne.yaml keeps up to 1000 points in a node, and a refit runs over
// all of them -- six refits of a 37-point batch were 0.25 ms of a lone wavefront): the points are appended through the node's chunk table in memory
// (wave_push_point), the running per-axis sums decide the intermediate refits by the diagonal bound exactly as above, and only the LAST refit of the batch
// -- the one the matcher can observe -- is computed, over the points the reference's last refit saw.  Returns the number of points consumed; the node's
// header in memory is current on return (wave_update_point continues from it when the node turned non-planar or filled up).
__device__ int wave_replay_planar_node(const RegMapDev& m, const int node, const int* order, const int cnt, const double* __restrict__ pt_data, const WaveCtx& w) {
    NodeRec& nd = m.nodes[node];
    int flags = nd.flags;
    const int want = NF_INIT | NF_PLANE | NF_UPDATE_EN;
    if ((flags & want) == (NF_INIT | NF_PLANE)) return cnt;   // a full planar node (m_update_enable_ == false) drops every point
    if ((flags & want) != want) return 0;
    int npts = nd.npts, newp = nd.newpts;
    double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
    const bool will_refit = newp + cnt > 5;
    if (will_refit) {
        for (int i = w.lane; i < npts; i += 64) {
            const double* q = node_point_ptr(m, node, i);
#pragma unroll
            for (int a = 0; a < 3; a++) { const double v = q[a]; s1[a] += v; s2[a] += v * v; }
        }
#pragma unroll
        for (int a = 0; a < 3; a++) { s1[a] = wave_sum(s1[a]); s2[a] = wave_sum(s2[a]); }
    }
    WaveCtx wq = w; wq.stats = nullptr;   // refits are counted here (deferred ones included), not by wave_init_plane
    int last_refit_n = 0, n_ref = 0, j = 0;
    long long n_ref_pts = 0;
    auto write_counts = [&]() {
        if (w.lane == 0) { nd.npts = npts; nd.newpts = newp; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (; j < cnt; j++) {
        const double* src = pt_data + (size_t)order[j] * IM_PT_DOUBLES;
        if (!wave_push_point(m, node, npts, src, w)) { write_counts(); return j; }   // pool exhausted (flag raised): stop here
        if (will_refit) {
            const double pv = w.lane < IM_PT_DOUBLES ? src[w.lane] : 0.0;
#pragma unroll
            for (int a = 0; a < 3; a++) { const double v = __shfl(pv, a, 64); s1[a] += v; s2[a] += v * v; }
        }
        npts++; newp++;
        if (newp > 5) {   // m_update_size_threshold_: the refit over all retained points falls due
            n_ref++; n_ref_pts += npts;
            const double dn = (double)npts;
            double vmin = 1e300, mag = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) { const double mu = s1[a] / dn; const double v = s2[a] / dn - mu * mu; vmin = fmin(vmin, v); mag += s2[a] / dn; }
            const double err = 1e-7 + 1e-12 * mag;
            newp = 0;
            if (vmin + err < (double)m.planer_threshold) last_refit_n = npts;   // certainly still planar: fit later, over these npts points
            else {
                write_counts();
                last_refit_n = 0;
                const bool planar = wave_init_plane(m, node, npts, wq);
                if (!planar) {   // the node turns non-planar: later points take the general route (children)
                    if (w.lane == 0) node_set_flags(m, w.root, node, flags, flags & ~NF_PLANE, w.shared);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    flags &= ~NF_PLANE;
                    // the fullness test of the same UpdateOctoTree call still runs (voxel_loc.cpp:245-250): a node that goes non-planar exactly when it
                    // fills up stops updating and drops its points -- at max_layer it would otherwise keep refitting
                    if (npts >= m.max_points_size) {
                        write_counts();
                        if (w.lane == 0) { nd.flags = nd.flags & ~NF_UPDATE_EN; node_free_points(m, node); nd.newpts = 0; }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        npts = 0; newp = 0;
                    }
                    j++;
                    break;
                }
            }
        }
        if (npts >= m.max_points_size) {   // the node is full: last fit (if one is pending), then it stops updating and drops its points
            write_counts();
            if (last_refit_n) { (void)wave_init_plane(m, node, last_refit_n, wq); last_refit_n = 0; }
            if (w.lane == 0) { nd.flags = nd.flags & ~NF_UPDATE_EN; node_free_points(m, node); nd.newpts = 0; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            npts = 0; newp = 0;
            j++;
            if (w.lane == 0 && w.stats && n_ref) { atomicAdd((unsigned long long*)&w.stats[0], (unsigned long long)n_ref); atomicAdd((unsigned long long*)&w.stats[1], (unsigned long long)n_ref_pts); }
            return j;
        }
    }
    write_counts();
    if (last_refit_n) (void)wave_init_plane(m, node, last_refit_n, wq);
    if (w.lane == 0 && w.stats && n_ref) { atomicAdd((unsigned long long*)&w.stats[0], (unsigned long long)n_ref); atomicAdd((unsigned long long*)&w.stats[1], (unsigned long long)n_ref_pts); }
    return j;
}

// one wavefront per touched root voxel.  mode 0 = updateVoxelMap (sequential replay), mode 1 = buildVoxelMap (bucket all, then init)
__global__ __launch_bounds__(256) void replay_kernel(RegMapDev m, const uint32_t* __restrict__ sorted_slot, const int32_t* __restrict__ sorted_idx,
                                                      const double* __restrict__ pt_data, int n, const int32_t* __restrict__ seg_start,
                                                      const int32_t* __restrict__ nseg, int mode, int64_t* stats) {
    __shared__ int stacks[4][48];
    const int wv = threadIdx.x >> 6;
    const int seg = blockIdx.x * 4 + wv;
    if (seg >= *nseg) return;
    WaveCtx w; w.lane = threadIdx.x & 63; w.stats = stats;
    const int start = seg_start[seg];
    const uint32_t slot = sorted_slot[start];
    const int root = m.htab[slot].root;
    if (root < 0) return;
    w.root = root;
    if (mode == 0) {
        for (int j = start; j < n && sorted_slot[j] == slot; j++)
            wave_update_point(m, root, pt_data + (size_t)sorted_idx[j] * IM_PT_DOUBLES, stacks[wv], w);
    } else {
        int cnt = m.nodes[root].npts;
        for (int j = start; j < n && sorted_slot[j] == slot; j++) {
            if (!wave_push_point(m, root, cnt, pt_data + (size_t)sorted_idx[j] * IM_PT_DOUBLES, w)) return;
            cnt++;
        }
        if (w.lane == 0) { m.nodes[root].npts = cnt; m.nodes[root].newpts = cnt; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (cnt > m.init_size[0]) wave_init_octo_tree(m, root, stacks[wv], w);
    }
}

#define RL_CAP 64    /* points of one scan falling into one root voxel that are ordered in LDS (a down-sampled scan puts <= ~8 into a voxel); longer lists take the
                        global-scratch path.  Kept small on purpose: LDS is what limits how many workgroups of the three concurrent chains fit a CU */
#define DBG_FUSED_OFF 64          /* IMMESH_DEBUG buffer: [0,64) counters, then 16384 x 8 trace words of replay_fused_kernel, then 8 x 512 x 8 of the residual passes */
#define DBG_FUSED_RECS 16384
#define DBG_RES_OFF (DBG_FUSED_OFF + DBG_FUSED_RECS * 8)
#define DBG_TOTAL_WORDS (DBG_RES_OFF + 8 * 512 * 8)
#define RF_PTS 128   /* a settled root's retained points + this scan's points live in registers (two per lane) */
// ---- the map update in two launches ---------------------------------------------------------------------------------------------------------
//   replay_fused_kernel   one wavefront per touched root voxel.  The common state of a settled map -- an initialised, planar, update-enabled root
//                         whose points fit two per lane -- is handled completely: list gather + ordering, the OctoTree::UpdateOctoTree state
//                         machine of the batch (voxel_loc.cpp:240-262) evaluated in registers, the append, and the ONE observable plane fit
//                         (only the last refit of a batch survives: the matcher reads the map between scans) with the points still in registers.
//                         Refits that fall due earlier in the batch only decide "still planar?" -- by the diagonal bound (lambda_min <= smallest
//                         diagonal entry of the covariance) or, when that does not decide, by the exact eigenvalues.  Nothing is written before
//                         every decision of the batch is known, so whatever this path cannot finish is handed over untouched;
//   replay_list_kernel    the work list of handed-over voxels (new / subdivided / filling-up roots, a few dozen per scan): the general state machine.
// Round 2 ran this as three launches (light -> list -> refit: three plane-fit latencies in a row, 40 + 57 + 18 us).
__global__ __launch_bounds__(256, 3) void replay_fused_kernel(RegMapDev m, const int32_t* __restrict__ pt_next, const unsigned long long* __restrict__ sort_key,
                                                            const double* __restrict__ pt_data, int64_t* stats, uint32_t* __restrict__ general_list,
                                                            unsigned long long* __restrict__ dbg, unsigned long long* flag_dev, unsigned long long* flag_host,
                                                            unsigned long long flag_seq) {
    // the registration launch before this one left the scan in the mesher's world buffer and consumed the input clouds (RpEpilogue): say so
    if (flag_dev && blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(flag_dev, flag_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag_host, flag_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __shared__ unsigned long long skey[4][RL_CAP];
    __shared__ int sidx[4][RL_CAP];
    __shared__ int order[4][RL_CAP];
    __builtin_amdgcn_s_setprio(REG_UPD_PRIO);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // A resident grid strides over the touched voxels (four wavefronts per workgroup: one-wavefront workgroups were measured dispatch-bound, ~130
    // workgroups per us; one workgroup per four down-sampled POINTS -- the count of touched voxels is only known on the device -- kept the
    // dispatcher busy for 15 us placing 2000 workgroups, 40 % of whose wavefronts found nothing to do, and the mesher's launches waiting behind them).
    const int t_first = blockIdx.x * 4 + wv, t_stride = gridDim.x * 4;
    uint32_t slot_first = m.touched[2 * (size_t)t_first];    // (read beside the counter, not behind it: the list has room for every wavefront of the grid)
    int root_first = (int)m.touched[2 * (size_t)t_first + 1];
    const int n_touched = m.counters[7];
#define FDBG(k) do { if (dbg) { const unsigned long long _t = __builtin_readcyclecounter(); if (tr) ((unsigned int*)tr)[4 + (k)] = (unsigned int)(_t - tprev); tprev = _t; } } while (0)
#define FEND(state, nref) do { if (tr) { tr[1] = __builtin_amdgcn_s_memrealtime(); tr[7] = (unsigned long long)((unsigned)cnt | ((unsigned)(nref) << 8) | ((unsigned)(state) << 16)); } } while (0)
    auto process = [&](const int t, const uint32_t slot, int root) __attribute__((always_inline)) {
    // IMMESH_DEBUG: one trace record per touched voxel (plain stores, no contention): [0] start, [1] end (s_memrealtime, 100 MHz),
    // [2..6] ten 32-bit cycle counts (root known, node line 0, chunk table, list head, list, sort, load, decide, commit, plane), [7] cnt | n_ref << 8 | state << 16
    unsigned long long* const tr = (dbg && lane == 0 && t < DBG_FUSED_RECS) ? dbg + DBG_FUSED_OFF + (size_t)t * 8 : nullptr;
    unsigned long long tprev = dbg ? __builtin_readcyclecounter() : 0;
    if (tr) { tr[0] = __builtin_amdgcn_s_memrealtime(); tr[2] = tr[3] = tr[4] = tr[5] = tr[6] = tr[7] = 0; }
    if (root < 0) root = m.htab[slot].root;
    if (root < 0) return;
    FDBG(0);
    NodeRec& nd = m.nodes[root];
    // node header and the head of the point list: independent loads, one latency
    const int flags = nd.flags, layer = nd.layer;
    const int npts = nd.npts, newp = nd.newpts;
    if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FDBG(1); }   // (debug build only: the three loads are timed one by one)
    int chunks[IM_INLINE_CHUNKS];
#pragma unroll
    for (int k = 0; k < IM_INLINE_CHUNKS; k++) chunks[k] = nd.chunks[k];
    if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FDBG(2); }
    int cnt = 0;
    int ihead = (int)(unsigned int)(m.slot_head[slot] & 0xFFFFFFFFull);
    if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FDBG(3); }
    for (int i = ihead; i >= 0; i = pt_next[i]) {
        if (cnt < RL_CAP && lane == 0) { skey[wv][cnt] = sort_key[i]; sidx[wv][cnt] = i; }
        cnt++;
    }
    FDBG(4);
    const int want = NF_INIT | NF_PLANE | NF_UPDATE_EN;
    if (layer == 0 && (flags & want) == (NF_INIT | NF_PLANE)) { FEND(1, 0); return; }   // a full planar root (m_update_enable_ == false) drops every point
    const int ntot = npts + cnt;
    bool hand_over = cnt > RL_CAP || (flags & want) != want || layer != 0 || newp > 5 || ntot > RF_PTS || ntot >= m.max_points_size;
    if (!hand_over) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int e = lane; e < cnt; e += 64) {  // rank sort: (key, index) pairs are unique -- std::sort(pv_list, var_contrast) restricted to this voxel, ties by scan index
            const unsigned long long k = skey[wv][e];
            const int id = sidx[wv][e];
            int rank = 0;
            for (int f = 0; f < cnt; f++) { const unsigned long long kf = skey[wv][f]; rank += (kf < k || (kf == k && sidx[wv][f] < id)) ? 1 : 0; }
            order[wv][rank] = id;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        FDBG(5);
        auto chunk_of = [&](int ci) { int c = chunks[0];
#pragma unroll
            for (int k = 1; k < IM_INLINE_CHUNKS; k++) { const int ck = chunks[k]; c = (ci == k) ? ck : c; }   // (by value: a conditional on lvalues selects ADDRESSES and pins the array to memory)
            return c; };
        // point q of the voxel AFTER the append: q < npts retained, else this scan's point order[q - npts].  Lane i holds q = i and q = i + 64.
        const int n_ref = (newp + cnt) / 6;   // refits falling due inside the batch: new_points counts 0..5 between refits (m_update_size_threshold_ = 5)
        double P0[9], P1[9];
#pragma unroll
        for (int k = 0; k < 9; k++) { P0[k] = 0; P1[k] = 0; }
        {
            const int q0 = lane, q1 = lane + 64;
            const double* s0 = nullptr; const double* s1 = nullptr;
            if (q0 < ntot && (q0 >= npts || n_ref > 0))
                s0 = q0 < npts ? m.chunk_data + ((size_t)chunk_of(q0 >> 4) * IM_CHUNK_PTS + (q0 & 15)) * IM_PT_DOUBLES : pt_data + (size_t)order[wv][q0 - npts] * IM_PT_DOUBLES;
            if (q1 < ntot && (q1 >= npts || n_ref > 0))
                s1 = q1 < npts ? m.chunk_data + ((size_t)chunk_of(q1 >> 4) * IM_CHUNK_PTS + (q1 & 15)) * IM_PT_DOUBLES : pt_data + (size_t)order[wv][q1 - npts] * IM_PT_DOUBLES;
            if (s0) {
#pragma unroll
                for (int k = 0; k < 9; k++) P0[k] = s0[k]; }
            if (s1) {
#pragma unroll
                for (int k = 0; k < 9; k++) P1[k] = s1[k]; }
        }
        if (dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FDBG(6);
        // ---- decisions (nothing is written yet)
        PlaneFit fit;
        long long n_ref_pts = 0;
        int n_last = 0;
        for (int r = 0; r < n_ref && !hand_over; r++) {
            const int nr = npts + 6 * (r + 1) - newp;      // points the node holds when the r-th refit of the batch runs
            n_ref_pts += nr;
            auto fe = [&](auto&& fn) __attribute__((always_inline)) { if (lane < nr) fn(P0); if (lane + 64 < nr) fn(P1); };
            if (r + 1 < n_ref) {
                // an intermediate refit: only its verdict matters.  lambda_min <= smallest diagonal entry of the covariance decides most of them
                double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
                fe([&](const double* q) __attribute__((always_inline)) {
#pragma unroll
                    for (int a = 0; a < 3; a++) { s1[a] += q[a]; s2[a] += q[a] * q[a]; } });
#pragma unroll
                for (int a = 0; a < 3; a++) { s1[a] = wave_sum_fast(s1[a]); s2[a] = wave_sum_fast(s2[a]); }
                const double dn = (double)nr;
                double vmin = 1e300, mag = 0;
#pragma unroll
                for (int a = 0; a < 3; a++) { const double mu = s1[a] / dn; const double v = s2[a] / dn - mu * mu; vmin = fmin(vmin, v); mag += s2[a] / dn; }
                if (vmin + (1e-7 + 1e-12 * mag) < (double)m.planer_threshold) continue;   // certainly still planar
            }
            fit_eigen(fe, nr, m.planer_threshold, fit);
            if (!fit.planar) hand_over = true;   // the root turns non-planar in the middle of the batch: the general kernel's job (children)
            n_last = nr;
        }
        FDBG(7);
        if (!hand_over) {
            // ---- commit: chunks for the new points, the points, the header, the plane
            const int c_first = (npts + IM_CHUNK_PTS - 1) >> 4, c_last = (ntot - 1) >> 4;   // chunk slots first touched by this batch
            if (cnt > 0 && c_first <= c_last) {
                int c = 0;
                const bool need = lane >= c_first && lane <= c_last && lane < IM_INLINE_CHUNKS && chunk_of(lane) < 0;
                if (need) c = alloc_chunk(m);
                if (__ballot(need && c < 0)) return;   // pool exhausted: alloc_chunk has raised the capacity flag, the update fails as a whole
#pragma unroll
                for (int k = 0; k < IM_INLINE_CHUNKS; k++) { const int ck = __shfl(c, k, 64); const int nk = __shfl(need ? 1 : 0, k, 64); if (nk) chunks[k] = ck; }
            }
            {
                const int q0 = lane, q1 = lane + 64;
                if (q0 >= npts && q0 < ntot) { double* d = m.chunk_data + ((size_t)chunk_of(q0 >> 4) * IM_CHUNK_PTS + (q0 & 15)) * IM_PT_DOUBLES;
#pragma unroll
                    for (int k = 0; k < 9; k++) d[k] = P0[k]; }
                if (q1 >= npts && q1 < ntot) { double* d = m.chunk_data + ((size_t)chunk_of(q1 >> 4) * IM_CHUNK_PTS + (q1 & 15)) * IM_PT_DOUBLES;
#pragma unroll
                    for (int k = 0; k < 9; k++) d[k] = P1[k]; }
            }
            if (lane == 0) {
                nd.npts = ntot; nd.newpts = (newp + cnt) % 6;
#pragma unroll
                for (int k = 0; k < IM_INLINE_CHUNKS; k++) nd.chunks[k] = chunks[k];
                // refit counters: 64 shards, one 128-byte line each (thousands of wavefronts adding to ONE address serialise at the memory side: ~30 ns per atomic)
                if (n_ref) { int64_t* sh = stats + 16 + (size_t)(t & 63) * 16; atomicAdd((unsigned long long*)&sh[0], (unsigned long long)n_ref); atomicAdd((unsigned long long*)&sh[1], (unsigned long long)n_ref_pts); }
            }
            FDBG(8);
            if (n_ref > 0) {   // (n_last is the last refit's point count: fit holds its eigen-decomposition)
                double pv[21];
                auto fe = [&](auto&& fn) __attribute__((always_inline)) { if (lane < n_last) fn(P0); if (lane + 64 < n_last) fn(P1); };
                fit_plane_var(fe, n_last, fit, pv);
                fit_store(nd, fit, pv, lane);
            }
            FDBG(9);
            FEND(2, n_ref);
            return;
        }
    }
    if (lane == 0) general_list[atomicAdd(&m.counters[10], 1)] = slot;
    FEND(3, 0);
    };
    for (int t = t_first; t < n_touched; t += t_stride) {
        if (t != t_first) { slot_first = m.touched[2 * (size_t)t]; root_first = (int)m.touched[2 * (size_t)t + 1]; }
        process(t, slot_first, root_first);
    }
#undef FDBG
#undef FEND
}

__device__ bool replay_split_root(const RegMapDev& m, const int root, const int* order, const int cnt, const double* __restrict__ pt_data, const WaveCtx& w);
__device__ bool replay_split_leaves(const RegMapDev& m, const int root, const int* order, const int cnt, const double* __restrict__ pt_data, const WaveCtx& w, int* s_key, int* s_first);
// updateVoxelMap without any global sort: one wavefront per root voxel of the work list gathers that voxel's points of this scan from its
// list, orders them as std::sort(pv_list, var_contrast) would (ascending covariance norm, ties by scan index) and replays them through the
// general state machine.  The grid is FIXED and strides over the list (a few dozen voxels per scan on a settled map, every touched voxel while
// a map is being surveyed): round 2 launched one wavefront per down-sampled point to find ~80 work items.
__global__ __launch_bounds__(256) void replay_list_kernel(RegMapDev m, const int32_t* __restrict__ pt_next, const unsigned long long* __restrict__ sort_key,
                                                           const double* __restrict__ pt_data, int64_t* stats, int32_t* __restrict__ big_idx, int32_t* __restrict__ big_order,
                                                           unsigned long long* __restrict__ dbg, const uint32_t* __restrict__ work, const int32_t* __restrict__ n_work) {
    __shared__ unsigned long long skey[4][RL_CAP];
    __shared__ int sidx[4][RL_CAP];
    __shared__ int order[4][RL_CAP];
    __shared__ int stacks[4][48];
    __builtin_amdgcn_s_setprio(REG_UPD_PRIO);   // map growth is on the pose chain too (the next scan's registration waits for it)
    const int wv = threadIdx.x >> 6;
    const int nw = *n_work;
    WaveCtx w; w.lane = threadIdx.x & 63; w.stats = stats;
    for (int t = blockIdx.x * 4 + wv; t < nw; t += gridDim.x * 4) {
        const uint32_t slot = work[t];   // the voxels the fused kernel handed over (replay_fused_kernel)
        const int root = m.htab[slot].root;
        if (root < 0) continue;
        w.root = root;
        const unsigned long long tdbg0 = dbg ? __builtin_readcyclecounter() : 0;
        __builtin_amdgcn_wave_barrier();   // (the previous voxel's reads of the per-wave LDS lists are done)
        int cnt = 0;
        for (int i = (int)(unsigned int)(m.slot_head[slot] & 0xFFFFFFFFull); i >= 0; i = pt_next[i]) {
            if (cnt < RL_CAP && w.lane == 0) { skey[wv][cnt] = sort_key[i]; sidx[wv][cnt] = i; }
            cnt++;
        }
        if (cnt > RL_CAP) {
            // dense or un-down-sampled scans (the reference's updateVoxelMap takes any count): the list is ordered in global scratch instead of
            // LDS -- a segment of big_idx / big_order (n entries each; the lists of a scan partition its points) claimed with one atomic
            int base = 0;
            if (w.lane == 0) base = atomicAdd(&m.counters[9], cnt);
            base = __shfl(base, 0, 64);
            if (w.lane == 0) { int k = 0; for (int i = (int)(unsigned int)(m.slot_head[slot] & 0xFFFFFFFFull); i >= 0; i = pt_next[i]) big_idx[base + k++] = i; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int e = w.lane; e < cnt; e += 64) {
                const int id = big_idx[base + e];
                const unsigned long long k = sort_key[id];
                int rank = 0;
                for (int f = 0; f < cnt; f++) { const int idf = big_idx[base + f]; const unsigned long long kf = sort_key[idf]; rank += (kf < k || (kf == k && idf < id)) ? 1 : 0; }
                big_order[base + rank] = id;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (replay_split_root(m, root, big_order + base, cnt, pt_data, w)) continue;   // (a subdivided root of a deep octree: its octants go to replay_sub_kernel)
            int jb = wave_replay_planar_root(m, root, big_order + base, cnt, pt_data, w);
            if (jb == 0) jb = wave_replay_planar_node(m, root, big_order + base, cnt, pt_data, w);
            for (int j = jb; j < cnt; j++) wave_update_point(m, root, pt_data + (size_t)big_order[base + j] * IM_PT_DOUBLES, stacks[wv], w);
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int e = w.lane; e < cnt; e += 64) {  // rank sort: (key, index) pairs are unique
            const unsigned long long k = skey[wv][e];
            const int id = sidx[wv][e];
            int rank = 0;
            for (int f = 0; f < cnt; f++) { const unsigned long long kf = skey[wv][f]; rank += (kf < k || (kf == k && sidx[wv][f] < id)) ? 1 : 0; }
            order[wv][rank] = id;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // settled planar roots take the register-resident fast path; whatever it does not consume goes through the general state machine
        // (a subdivided root of a deep octree: its points go to replay_sub_kernel -- grouped by the LEAF each one descends to (round 6), long lists by octant)
        if (replay_split_leaves(m, root, order[wv], cnt, pt_data, w, (int*)skey[wv], sidx[wv])) continue;
        if (replay_split_root(m, root, order[wv], cnt, pt_data, w)) continue;
        const unsigned long long tdbg1 = dbg ? __builtin_readcyclecounter() : 0;
        int jf = wave_replay_planar_root(m, root, order[wv], cnt, pt_data, w);
        if (jf == 0) jf = wave_replay_planar_node(m, root, order[wv], cnt, pt_data, w);   // (planar roots that do not fit the register path: deep-octree configurations)
        for (int j = jf; j < cnt; j++) wave_update_point(m, root, pt_data + (size_t)order[wv][j] * IM_PT_DOUBLES, stacks[wv], w);
        if (dbg && w.lane == 0) {   // IMMESH_DEBUG: slowest voxel of each kind (cycles << 16 | points), totals
            const unsigned long long t2 = __builtin_readcyclecounter();
            atomicMax(&dbg[jf == cnt ? 8 : 9], ((t2 - tdbg0) << 16) | (unsigned long long)cnt);
            atomicAdd(&dbg[jf == cnt ? 10 : 11], t2 - tdbg1);
            atomicAdd(&dbg[jf == cnt ? 12 : 13], 1ull);
            atomicAdd(&dbg[14], tdbg1 - tdbg0);
        }
    }
}

// Deep octrees (velodyne.yaml: 3 m roots, four layers, ~30 points of a scan per root, up to 60): the general state machine replays a root's points one
// by one -- descent, push, a refit every sixth point -- and the densest root sets the kernel's time (61 points: 950 k cycles = 0.4 ms of the 0.6 ms scan).
// A SUBDIVIDED root (initialised, not a plane, below max_layer) only routes: UpdateOctoTree hands the point to the child of its octant (voxel_loc.cpp:
// 263-288), so the eight octants are independent state machines and only the order of the points WITHIN an octant matters.  replay_list_kernel
// therefore cuts such a root's ordered list into its octants' lists (replay_split_root) and replay_sub_kernel replays each with a wavefront of its own.
// What the octants share is the root's flat list of planar descendants: edited under the root's lock (node_set_flags, shared).
// Round 6: the same cut, all the way down.  An octant of a subdivided 3 m root still holds ~30 of a scan's points, and replay_sub_kernel replayed them ONE BY
// ONE through three more levels of dependent header reads (wave_update_point: ~8 us a point -- the launch was the densest octant's chain, 0.11 ms per scan).
// Every point's destination is known before anything is written: the node its descent through the tree AS IT STANDS stops at -- a plane, a node still
// filling up, a last-layer node, a former plane that still holds its points, or the parent of a child that does not exist yet.  Destinations are disjoint
// subtrees, so they are independent state machines exactly like the octants, and whatever a destination's batch changes (a leaf that fills up and is cut, a
// plane that turns non-planar) happens below it.  Each lane descends for its own point (read-only, in parallel), the points are grouped by destination in
// the voxel's replay order, and every group becomes a work item for replay_sub_kernel: a planar leaf takes its batch in one go (wave_replay_planar_node),
// anything else replays from ITS node instead of from the octant.  Lists of up to RL_CAP points (the LDS tables of the caller); longer ones take the octant cut.
#define SPLIT_MISSING 0x40000000   /* group key of "child `oct` of `node` does not exist": node | oct << 26 | this bit (node ids stay below 2^26 on deep-octree maps, checked) */
__device__ __noinline__ bool replay_split_leaves(const RegMapDev& m, const int root, const int* order, const int cnt, const double* __restrict__ pt_data, const WaveCtx& w, int* s_key, int* s_first) {
    const NodeRec& nr = m.nodes[root];
    const int flags = nr.flags;
    if (!m.split_general || cnt < 2 || cnt > RL_CAP || !(flags & NF_INIT) || (flags & NF_PLANE) || nr.layer >= m.max_layer || m.cap_nodes >= (1 << 26)) return false;
    if (w.lane == 0 && nr.npts != 0) node_free_points(m, root);   // what UpdateOctoTree does on its first visit after the cut (voxel_loc.cpp:263-266)
    const int lane = w.lane;
    // ---- descent: lane j < cnt follows point order[j] down the tree as it stands
    int key = -1;
    if (lane < cnt) {
        const double* q = pt_data + (size_t)order[lane] * IM_PT_DOUBLES;
        const double qx = q[0], qy = q[1], qz = q[2];
        int nd = root;
        for (int depth = 0; depth < 8; depth++) {
            const NodeRec& r = m.nodes[nd];
            const int f = r.flags, layer = r.layer, npts = r.npts;
            // a routing node: initialised, not a plane, below the last layer, its buffer already dropped (everything else is a destination)
            if (!((f & NF_INIT) && !(f & NF_PLANE) && layer < m.max_layer && (npts == 0 || nd == root))) { key = nd; break; }
            const int oct = 4 * (qx > r.center[0] ? 1 : 0) + 2 * (qy > r.center[1] ? 1 : 0) + (qz > r.center[2] ? 1 : 0);   // octant_of
            const int child = r.child[oct];
            if (child < 0) { key = nd | (oct << 26) | SPLIT_MISSING; break; }
            nd = child;
        }
        if (key < 0) key = nd;   // (deeper than any octree of the library: treat as a destination)
    }
    if (lane < cnt) s_key[lane] = key;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- grouping in replay order: first = lowest position with my key, rank = earlier positions with my key, size = positions with my key
    int first = lane, rank = 0, size = 0;
    if (lane < cnt) {
        first = -1;
        for (int f = 0; f < cnt; f++) {
            const bool same = s_key[f] == key;
            if (same && first < 0) first = f;
            if (same && f < lane) rank++;
            if (same) size++;
        }
        s_first[lane] = first;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int before = 0;   // points whose group starts before mine
    if (lane < cnt) for (int f = 0; f < cnt; f++) before += s_first[f] < first ? 1 : 0;
    const bool leader = lane < cnt && first == lane;
    const unsigned long long lm = __ballot(leader);
    const int n_items = (int)__popcll(lm);
    int seg = 0, item0 = 0;
    if (lane == 0) { seg = atomicAdd(&m.counters[11], cnt); item0 = atomicAdd(&m.counters[12], n_items); }
    seg = __shfl(seg, 0, 64); item0 = __shfl(item0, 0, 64);
    if (lane < cnt) m.sub_order[seg + before + rank] = order[lane];
    if (leader) {
        const int it = item0 + (int)__popcll(lm & ((1ull << lane) - 1ull));
        // item: start node (for a missing child: its parent -- the general state machine creates the child with the first point and routes the rest), root
        const int start = (key & SPLIT_MISSING) ? (key & ((1 << 26) - 1)) : key;
        m.sub_items[2 * (size_t)it] = (unsigned long long)(unsigned int)start | ((unsigned long long)(unsigned int)root << 32);
        m.sub_items[2 * (size_t)it + 1] = (unsigned long long)(unsigned int)(seg + before) | ((unsigned long long)(unsigned int)size << 32);
    }
    return true;
}
__device__ bool replay_split_root(const RegMapDev& m, const int root, const int* order, const int cnt, const double* __restrict__ pt_data, const WaveCtx& w) {
    const NodeRec& nr = m.nodes[root];
    const int flags = nr.flags;
    if (!m.split_general || cnt < 8 || !(flags & NF_INIT) || (flags & NF_PLANE) || nr.layer >= m.max_layer) return false;
    if (w.lane == 0 && nr.npts != 0) node_free_points(m, root);   // what UpdateOctoTree does on its first visit after the cut (voxel_loc.cpp:263-266)
    const double ctr[3] = {nr.center[0], nr.center[1], nr.center[2]};
    int child[8], n_oct[8], base[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { child[k] = nr.child[k]; n_oct[k] = 0; }
    for (int b0 = 0; b0 < cnt; b0 += 64) {   // counts per octant
        const int j = b0 + w.lane;
        const int oct = j < cnt ? octant_of(pt_data + (size_t)order[j] * IM_PT_DOUBLES, ctr) : -1;
#pragma unroll
        for (int k = 0; k < 8; k++) n_oct[k] += __popcll(__ballot(oct == k));
    }
    int total = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) total += n_oct[k];
    int seg = 0, item0 = 0, n_items = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) n_items += n_oct[k] > 0 ? 1 : 0;
    // every child the octants need exists BEFORE anything is reserved: a failed make_child (node pool exhausted) must not leave reserved work items
    // that replay_sub_kernel would read unwritten
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (n_oct[k] > 0) {
            int c = child[k];
            if (w.lane == 0 && c < 0) c = make_child(m, root, k);
            c = __shfl(c, 0, 64);
            if (c < 0) return true;   // node pool exhausted (flag raised by node_alloc): the update fails as a whole
            child[k] = c;
        }
    }
    if (w.lane == 0) { seg = atomicAdd(&m.counters[11], total); item0 = atomicAdd(&m.counters[12], n_items); }
    seg = __shfl(seg, 0, 64); item0 = __shfl(item0, 0, 64);
    {
        int run = seg, it = item0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            base[k] = run; run += n_oct[k];
            if (n_oct[k] > 0) {
                const int c = child[k];
                if (w.lane == 0) {
                    m.sub_items[2 * (size_t)it] = (unsigned long long)(unsigned int)c | ((unsigned long long)(unsigned int)root << 32);
                    m.sub_items[2 * (size_t)it + 1] = (unsigned long long)(unsigned int)base[k] | ((unsigned long long)(unsigned int)n_oct[k] << 32);
                }
                it++;
            }
        }
    }
    int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b0 = 0; b0 < cnt; b0 += 64) {   // the octants' lists, each in the voxel's replay order
        const int j = b0 + w.lane;
        const int id = j < cnt ? order[j] : -1;
        const int oct = j < cnt ? octant_of(pt_data + (size_t)id * IM_PT_DOUBLES, ctr) : -1;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned long long mk = __ballot(oct == k);
            if (oct == k) m.sub_order[base[k] + fill[k] + __popcll(mk & ((1ull << w.lane) - 1ull))] = id;
            fill[k] += __popcll(mk);
        }
    }
    return true;
}
__global__ __launch_bounds__(256) void replay_sub_kernel(RegMapDev m, const double* __restrict__ pt_data, int64_t* stats) {
    __shared__ int stacks[4][48];
    __builtin_amdgcn_s_setprio(REG_UPD_PRIO);
    const int wv = threadIdx.x >> 6;
    const int n_items = m.counters[12];
    WaveCtx w; w.lane = threadIdx.x & 63; w.stats = stats; w.shared = true;
    for (int t = blockIdx.x * 4 + wv; t < n_items; t += gridDim.x * 4) {
        const unsigned long long a = m.sub_items[2 * (size_t)t], b = m.sub_items[2 * (size_t)t + 1];
        const int child = (int)(unsigned int)a, first = (int)(unsigned int)b, cnt = (int)(unsigned int)(b >> 32);
        w.root = (int)(unsigned int)(a >> 32);
        for (int j = wave_replay_planar_node(m, child, m.sub_order + first, cnt, pt_data, w); j < cnt; j++)
            wave_update_point(m, child, pt_data + (size_t)m.sub_order[first + j] * IM_PT_DOUBLES, stacks[wv], w);
    }
}

// merge chunk ids freed by the previous kernel into the ready stack
__global__ void merge_free_kernel(RegMapDev m) {
    const int np = m.counters[3];
    const int base = m.counters[2];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) m.free_ready[base + i] = m.free_pending[i];
}
__global__ void merge_free_finish_kernel(RegMapDev m) { m.counters[2] += m.counters[3]; m.counters[3] = 0; m.counters[7] = 0; m.counters[11] = 0; m.counters[12] = 0; }
// Tail of the per-scan map update, one launch: chunks freed by the replay kernel join the free list, the per-update counters reset, and the map
// counters (node / chunk usage, capacity flag) go straight to pinned host memory -- the next scan's residual passes are queued right behind it.
__global__ __launch_bounds__(256) void merge_free_tail_kernel(RegMapDev m, int32_t* __restrict__ host_counters) { map_update_tail(m, host_counters); }

// =====================================================================================================================
// introspection
// =====================================================================================================================
__global__ void dump_planes_kernel(RegMapDev m, PlaneRecDev* out, long long cap, unsigned long long* count) {
    const int nn = m.counters[0];
    for (int nd = blockIdx.x * blockDim.x + threadIdx.x; nd < nn; nd += gridDim.x * blockDim.x) {
        const int f = m.nodes[nd].flags;
        if (!(f & NF_INIT)) continue;
        const unsigned long long idx = atomicAdd(count, 1ull);
        if ((long long)idx >= cap || out == nullptr) continue;
        PlaneRecDev& r = out[idx];
        const unsigned long long pk = m.nodes[nd].key;
        r.key[0] = (long long)(pk & IM_KEY_MASK) - IM_KEY_BIAS;
        r.key[1] = (long long)((pk >> 21) & IM_KEY_MASK) - IM_KEY_BIAS;
        r.key[2] = (long long)((pk >> 42) & IM_KEY_MASK) - IM_KEY_BIAS;
        r.layer = m.nodes[nd].layer; r.path = m.nodes[nd].path; r.is_plane = (f & NF_PLANE) ? 1 : 0; r.n_points = m.nodes[nd].npts;
        r.update_enable = (f & NF_UPDATE_EN) ? 1 : 0; r.new_points = m.nodes[nd].newpts;
        r.radius = m.nodes[nd].radius; r.min_eig = m.nodes[nd].min_eig; r.d = m.nodes[nd].d; r.pad = 0;
        for (int k = 0; k < 3; k++) { r.center[k] = m.nodes[nd].p_center[k]; r.normal[k] = m.nodes[nd].p_normal[k]; }
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) r.plane_var[a * 6 + b] = (f & NF_PLANE) ? m.nodes[nd].p_var[(a <= b ? sym21_index(a, b) : sym21_index(b, a))] : 0.0;
    }
}

// fill helpers
__global__ void fill_u64_kernel(unsigned long long* p, unsigned long long v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void iota_kernel(int32_t* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// pcl-shaped clouds consumed in place (immesh_process_scan_strided): point i at src + i * stride bytes -- x, y, z first, the intensity (if asked for) at
// int_off -- into the packed xyz / xyzI layout the kernels read.  One thread per point; rows of a 32- or 48-byte stride are read as whole 16-byte words.
__global__ void unpack_strided_kernel(const unsigned char* __restrict__ src, int n, int stride, int int_off, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char* p = src + (size_t)i * stride;
    float x, y, z, w = 0.f;
    if ((stride & 15) == 0 && ((uintptr_t)src & 15) == 0) {
        const float4 a = *(const float4*)p;
        x = a.x; y = a.y; z = a.z;
        if (int_off >= 0) w = *(const float*)(p + int_off);
    } else {
        x = *(const float*)p; y = *(const float*)(p + 4); z = *(const float*)(p + 8);
        if (int_off >= 0) w = *(const float*)(p + int_off);
    }
    if (int_off >= 0) ((float4*)out)[i] = make_float4(x, y, z, w);
    else { out[(size_t)i * 3] = x; out[(size_t)i * 3 + 1] = y; out[(size_t)i * 3 + 2] = z; }
}
void launch_unpack_strided(hipStream_t s, const void* src, int n, int stride_bytes, int intensity_off_bytes, float* out) {
    KLAUNCH(unpack_strided_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned char*)src, n, stride_bytes, intensity_off_bytes, out);
}

// ---- launchers (called from the host layer) -------------------------------------------------------------------------
void launch_residual(hipStream_t s, const RegMapDev& m, const RegIterArgs& a, RegState* rs, const float* pts, int n, double* partials, unsigned int* done_counter,
                     double* out48, double* reg_out, double ticket, int8_t* o_match, int32_t* o_node, float* o_dis, double* o_rinv, double* o_normal) {
    const int nb = (n + 63) / 64;
    if (!(a.pad & 4) && m.cap_nodes < (1 << 26)) KLAUNCH(residual_kernel<true>, dim3(nb), dim3(64), 0, s, m, a, rs, pts, n, partials, done_counter, out48, reg_out, ticket, o_match, o_node, o_dis, o_rinv, o_normal);
    else KLAUNCH(residual_kernel<false>, dim3(nb), dim3(64), 0, s, m, a, rs, pts, n, partials, done_counter, out48, reg_out, ticket, o_match, o_node, o_dis, o_rinv, o_normal);
}
void launch_residual_persistent(hipStream_t s, const RegMapDev& m, const RegIterArgs& a, RegState* rs, const float* pts, int n, double* slots, double* slots_next,
                                int32_t* host_counters, double* reg_out, double ticket, int8_t* o_match, int32_t* o_node, float* o_dis, double* o_rinv, double* o_normal,
                                const RpEpilogue& ep, int max_blocks) {
    // resident grid: at most max_blocks (<= RP_MAX_BLOCKS) four-wavefront blocks + the tail's -- half of what the device holds resident, so that two
    // contexts registering at the same time can never wait for each other's CUs
    const int nb = std::max(1, std::min((n + 255) / 256, std::min(max_blocks, RP_MAX_BLOCKS)));
    if (!(a.pad & 4) && m.cap_nodes < (1 << 26)) KLAUNCH(residual_persistent_kernel<true>, dim3(nb + 1), dim3(256), 0, s, m, a, rs, pts, n, slots, slots_next, a.max_iter * RP_MAX_BLOCKS * RES_NR, host_counters, reg_out, ticket, o_match, o_node,
            o_dis, o_rinv, o_normal, ep);
    else KLAUNCH(residual_persistent_kernel<false>, dim3(nb + 1), dim3(256), 0, s, m, a, rs, pts, n, slots, slots_next, a.max_iter * RP_MAX_BLOCKS * RES_NR, host_counters, reg_out, ticket, o_match, o_node,
            o_dis, o_rinv, o_normal, ep);
}
// workgroups of residual_persistent_kernel the device holds resident at once (registers + LDS decide: one per CU on an MI355X)
int residual_persistent_resident_blocks(int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, residual_persistent_kernel<true>, 256, 0) != hipSuccess) { (void)hipGetLastError(); per_cu = 1; }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); cus = 0; }
    return std::max(1, per_cu) * cus;
}
void launch_ekf_step(hipStream_t s, const RegIterArgs& a, RegState* rs, const double* sums48, double* reg_out, double ticket) {
    KLAUNCH(ekf_step_kernel, dim3(1), dim3(64), 0, s, a, rs, sums48, reg_out, ticket);
}
void launch_point_var(hipStream_t s, const RegMapDev& m, const ScanParams& sp, const ScanParams* spd, const float* pts, int n, int stride, int mode, double* pt_data,
                      unsigned long long* sort_key, uint32_t* slot, int32_t* pt_next, const float* raw_xyzi, float* world_xyzi, int n_raw) {
    const int nb_pv = (n + 255) / 256, nb_x = raw_xyzi ? (n_raw + 255) / 256 : 0;
    KLAUNCH(point_var_kernel, dim3(nb_pv + nb_x), dim3(256), 0, s, m, sp, spd, pts, n, stride, mode, pt_data, sort_key, slot, pt_next, (const float4*)raw_xyzi,
            (float4*)world_xyzi, n_raw, nb_pv);
}
void launch_replay_lists(hipStream_t s, const RegMapDev& m, const int32_t* pt_next, const unsigned long long* sort_key, const double* pt_data, int n,
                         int64_t* stats, int32_t* host_counters, int32_t* big_idx, int32_t* big_order, uint32_t* general_list, unsigned long long* dbg, bool with_tail,
                         unsigned long long* flag_dev, unsigned long long* flag_host, unsigned long long flag_seq) {
    static const int fused_wgs = [] { const char* e = getenv("IMMESH_FUSED_WGS"); const int v = e ? atoi(e) : 0; return v >= 32 ? v : 768; }();   // (measurement knob, tools/r06_fused.sh)
    KLAUNCH(replay_fused_kernel, dim3(std::min((n + 3) / 4, fused_wgs)), dim3(256), 0, s, m, pt_next, sort_key, pt_data, stats, general_list, dbg, flag_dev, flag_host, flag_seq);
    // the work list's length is only known on the device: a fixed grid strides over it (sized for the map-building case, where every touched voxel is on it)
    // points per workgroup of the list kernel's grid.  Two-layer maps (avia.yaml): 128 -- its ~1 100 general voxels are bound by the slowest voxel's chain, more
    // workgroups only get in the mesher's way (sweep 64 / 32 / 16 / 8: 5 700-5 930 scans/s against 5 920-5 930, round 6).  Deep octrees (velodyne.yaml: every
    // touched root is on the list and most are split here): 32 -- the launch was 252 wavefronts working through 960 roots, 0.124 -> 0.094 ms per scan.
    static const int list_div_env = [] { const char* e = getenv("IMMESH_LIST_DIV"); const int v = e ? atoi(e) : 0; return v >= 4 ? v : 0; }();   // (measurement knob, tools/r06_c4div.sh)
    const int list_div = list_div_env ? list_div_env : (m.split_general ? 32 : 128);
    const int nb_list = std::min(std::max((n + list_div - 1) / list_div, 32), 4096);
    KLAUNCH(replay_list_kernel, dim3(nb_list), dim3(256), 0, s, m, pt_next, sort_key, pt_data, stats, big_idx, big_order, dbg, (const uint32_t*)general_list,
            (const int32_t*)(m.counters + 10));
    // (items are leaf groups since round 6 -- a few points each, thousands per scan: one wavefront per ~4 points instead of per 16)
    if (m.split_general) KLAUNCH(replay_sub_kernel, dim3(std::min(std::max((n + 15) / 16, 32), 2048)), dim3(256), 0, s, m, pt_data, stats);
    if (with_tail) KLAUNCH(merge_free_tail_kernel, dim3(1), dim3(256), 0, s, m, host_counters);
}
void launch_map_update_tail(hipStream_t s, const RegMapDev& m, int32_t* host_counters) { KLAUNCH(merge_free_tail_kernel, dim3(1), dim3(256), 0, s, m, host_counters); }
void launch_segment_heads(hipStream_t s, const uint32_t* sorted_slot, int n, int32_t* seg_start, int32_t* nseg) {
    (void)hipMemsetAsync(nseg, 0, sizeof(int32_t), s);
    KLAUNCH(segment_heads_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sorted_slot, n, seg_start, nseg);
}
void launch_replay(hipStream_t s, const RegMapDev& m, const uint32_t* sorted_slot, const int32_t* sorted_idx, const double* pt_data, int n,
                   const int32_t* seg_start, const int32_t* nseg, int max_segments, int mode, int64_t* stats) {
    KLAUNCH(replay_kernel, dim3((max_segments + 3) / 4), dim3(256), 0, s, m, sorted_slot, sorted_idx, pt_data, n, seg_start, nseg, mode, stats);
    KLAUNCH(merge_free_kernel, dim3(64), dim3(256), 0, s, m);
    KLAUNCH(merge_free_finish_kernel, dim3(1), dim3(1), 0, s, m);
}
void launch_dump_planes(hipStream_t s, const RegMapDev& m, PlaneRecDev* out, long long cap, unsigned long long* count) {
    (void)hipMemsetAsync(count, 0, sizeof(unsigned long long), s);
    KLAUNCH(dump_planes_kernel, dim3(1024), dim3(256), 0, s, m, out, cap, count);
}
void launch_fill_u64(hipStream_t s, unsigned long long* p, unsigned long long v, size_t n) {
    KLAUNCH(fill_u64_kernel, dim3(2048), dim3(256), 0, s, p, v, n);
}
void launch_iota(hipStream_t s, int32_t* p, int n) { KLAUNCH(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, n); }
