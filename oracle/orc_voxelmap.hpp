// ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
// CPU restatement of ImMesh's voxel-map registration path.
// Every function cites the reference file:line it follows (paths relative to /root/reference).
// PARITY UNPINNED vs the real binary: the reference has no tests/golden vectors and cannot be
// built here (needs Eigen/PCL/ROS).  Pinned instead by analytic KATs + numpy cross-checks in tests/.
#pragma once
#include "orc_linalg.hpp"
#include <unordered_map>
#include <cstdio>

namespace orc {

struct Config {
    double voxel_size = 0.5;          // voxel/max_voxel_size        config/avia.yaml:53
    int max_layer = 2;                // voxel/max_layer             :54
    int layer_init[5] = {5, 5, 5, 5, 5};  // voxel/layer_init_size  :55
    int max_points_size = 100;        // voxel/max_points_size       :56
    double planer_threshold = 0.01;   // voxel/min_eigen_value       :50
    double dept_err = 0.02, beam_err = 0.05;  // noise_model/ranging_cov, angle_cov :48-49
    int calib_laser = 0;              // preprocess/calib_laser
    double sigma_num = 3.0;           // hard-coded 3.0 at voxel_mapping.cpp:1365
    int max_iter = 4;                 // mapping/max_iteration
    double extR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, extT[3] = {0, 0, 0};
    double mesh_min_spacing = 0.1, mesh_voxel = 0.4, mesh_region = 10.0;
    int mesh_append_budget = 10000;   // meshing/number_of_pts_append_to_map
};

struct State {  // StatesGroup, include/common_lib.h:199-288
    double R[9], t[3], vel[3], bg[3], ba[3], g[3];
    double cov[324];
};

struct Counters {  // per-scan counters feeding the roofline denominator
    long n_ds = 0, n_iter = 0, n_match = 0, n_plane_tests = 0, n_extra_probe = 0;
    long n_refits = 0, n_refit_pts = 0;
    long n_app = 0, n_new = 0, v_act = 0, n_v = 0, n_u = 0, t_v = 0, t_add = 0, t_rem = 0, c1 = 0, c20 = 0, n_degenerate_skips = 0;
};

// ---- a1: key quantisation, voxel_mapping.cpp:118-127 / 172-181 / 328-337 -------------------------------
struct Key {
    int64_t x, y, z;
    bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct KeyHash {  // bucket choice only (voxel_loc.hpp:117-127); any hash gives the same results
    size_t operator()(const Key& k) const {
        uint64_t h = (uint64_t)k.x * 0x9E3779B97F4A7C15ull ^ ((uint64_t)k.y * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull) ^ ((uint64_t)k.z * 0xD6E8FEB86659FD93ull);
        return (size_t)(h ^ (h >> 29));
    }
};
// quotient is formed in double (p/voxel_size), narrowed to float, decremented if negative, truncated.
inline Key key_from_quotient(const double q[3]) {
    float loc[3];
    for (int j = 0; j < 3; j++) {
        loc[j] = (float)q[j];
        if (loc[j] < 0) loc[j] -= 1.0;  // float -= double literal: computed in double, stored float
    }
    return Key{(int64_t)loc[0], (int64_t)loc[1], (int64_t)loc[2]};
}

// ---- a2 / a3 / a4 types, voxel_loc.hpp:63-177 -------------------------------------------------------------
struct PointWithVar { double p[3]; double pw[3]; double var[9]; };
struct Plane {
    double center[3] = {0, 0, 0}, normal[3] = {0, 0, 0}, cov[9] = {0}, plane_var[36] = {0};
    float radius = 0, min_eig = 1, d = 0;
    int points_size = 0;
    bool is_plane = false, is_init = false, is_update = false;
    int id = 0;
};
struct Ptpl {  // voxel_loc.hpp:63-73
    double point[3], normal[3], center[3], plane_var[36];
    int layer; double d;
};

struct VoxelMap;
struct OctoTree {
    std::vector<PointWithVar> temp_points;
    Plane plane;
    int layer, max_layer, octo_state = 0;
    OctoTree* leaves[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double voxel_center[3] = {0, 0, 0};
    const int* layer_init_num;
    float quater_length = 0, planer_threshold;
    int update_size_threshold = 5, octo_init_size, max_points_size, new_points = 0;
    bool init_octo = false, update_enable = true;
    VoxelMap* owner;
    OctoTree(VoxelMap* o, int max_layer_, int layer_, const int* lin, int mps, float pt)
        : layer(layer_), max_layer(max_layer_), layer_init_num(lin), planer_threshold(pt), max_points_size(mps), owner(o) {
        octo_init_size = layer_init_num[layer];
    }
    ~OctoTree() { for (auto* l : leaves) delete l; }
    void init_plane(const std::vector<PointWithVar>& points, Plane* plane);
    void init_octo_tree();
    void cut_octo_tree();
    void update(const PointWithVar& pv);
    OctoTree* make_child(int xyz[3]);
};

struct VoxelMap {
    Config cfg;
    int threads = 1;           // matcher loop (BuildResidualListOMP runs under OpenMP in the reference); results do not depend on it
    std::unordered_map<Key, OctoTree*, KeyHash> map;
    int plane_id = 0;          // g_plane_id, voxel_loc.cpp:43
    int g_max_points = 1000;   // voxel_loc.cpp:45
    Counters cnt;
    ~VoxelMap() { for (auto& kv : map) delete kv.second; }
};

// ---- a5: OctoTree::init_plane, voxel_loc.cpp:47-139 ---------------------------------------------------
inline void OctoTree::init_plane(const std::vector<PointWithVar>& points, Plane* pl) {
    owner->cnt.n_refits++;
    owner->cnt.n_refit_pts += (long)points.size();
    for (int i = 0; i < 36; i++) pl->plane_var[i] = 0;
    for (int i = 0; i < 9; i++) pl->cov[i] = 0;
    for (int i = 0; i < 3; i++) { pl->center[i] = 0; pl->normal[i] = 0; }
    pl->points_size = (int)points.size();
    pl->radius = 0;
    for (const auto& pv : points) {  // :55-59
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) pl->cov[i * 3 + j] += pv.p[i] * pv.p[j];
        for (int i = 0; i < 3; i++) pl->center[i] += pv.p[i];
    }
    const double n = (double)pl->points_size;
    for (int i = 0; i < 3; i++) pl->center[i] = pl->center[i] / n;  // :60
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) pl->cov[i * 3 + j] = pl->cov[i * 3 + j] / n - pl->center[i] * pl->center[j];  // :61
    double ev[3], U[9];
    sym3_eigen_jacobi(pl->cov, ev, U);  // stands in for Eigen::EigenSolver :62-66 (real parts)
    int imin = 0, imax = 0;             // minCoeff / maxCoeff: first extremal index :68-69
    for (int i = 1; i < 3; i++) { if (ev[i] < ev[imin]) imin = i; if (ev[i] > ev[imax]) imax = i; }
    if (ev[imin] < (double)planer_threshold) {  // :78  (float member promoted to double)
        const double Umin[3] = {U[0 * 3 + imin], U[1 * 3 + imin], U[2 * 3 + imin]};
        for (size_t i = 0; i < points.size(); i++) {  // :80-105
            double F[9];
            const double dp[3] = {points[i].p[0] - pl->center[0], points[i].p[1] - pl->center[1], points[i].p[2] - pl->center[2]};
            for (int m = 0; m < 3; m++) {
                if (m != imin) {
                    const double denom = n * (ev[imin] - ev[m]);
                    const double row[3] = {dp[0] / denom, dp[1] / denom, dp[2] / denom};
                    const double Um[3] = {U[0 * 3 + m], U[1 * 3 + m], U[2 * 3 + m]};
                    double S[9];  // u_m u_min^T + u_min u_m^T
                    for (int r = 0; r < 3; r++)
                        for (int c = 0; c < 3; c++) S[r * 3 + c] = Um[r] * Umin[c] + Umin[r] * Um[c];
                    for (int c = 0; c < 3; c++) F[m * 3 + c] = row[0] * S[0 * 3 + c] + row[1] * S[1 * 3 + c] + row[2] * S[2 * 3 + c];
                } else {
                    F[m * 3 + 0] = 0; F[m * 3 + 1] = 0; F[m * 3 + 2] = 0;
                }
            }
            double J[18];  // 6x3
            m3_mul(U, F, J);  // J.block<3,3>(0,0) = evecs * F
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) J[(3 + r) * 3 + c] = (r == c) ? 1.0 / n : 0.0;  // J_Q :75-76
            double JV[18];  // J * var (6x3)
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 3; c++)
                    JV[r * 3 + c] = J[r * 3 + 0] * points[i].var[0 * 3 + c] + J[r * 3 + 1] * points[i].var[1 * 3 + c] + J[r * 3 + 2] * points[i].var[2 * 3 + c];
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 6; c++)
                    pl->plane_var[r * 6 + c] += JV[r * 3 + 0] * J[c * 3 + 0] + JV[r * 3 + 1] * J[c * 3 + 1] + JV[r * 3 + 2] * J[c * 3 + 2];
        }
        for (int i = 0; i < 3; i++) pl->normal[i] = Umin[i];  // :107
        pl->min_eig = (float)ev[imin];
        pl->radius = (float)std::sqrt(ev[imax]);
        pl->d = (float)(-(pl->normal[0] * pl->center[0] + pl->normal[1] * pl->center[1] + pl->normal[2] * pl->center[2]));
        pl->is_plane = true;
        pl->is_update = true;
        if (!pl->is_init) { pl->id = owner->plane_id++; pl->is_init = true; }
    } else {
        if (!pl->is_init) { pl->id = owner->plane_id++; pl->is_init = true; }
        pl->is_update = true;
        pl->is_plane = false;
    }
}

// ---- a4: init_octo_tree / cut_octo_tree / UpdateOctoTree, voxel_loc.cpp:141-308 ------------------------
inline void OctoTree::init_octo_tree() {
    if ((int)temp_points.size() > octo_init_size) {
        init_plane(temp_points, &plane);
        if (plane.is_plane) octo_state = 0;
        else { octo_state = 1; cut_octo_tree(); }
        init_octo = true;
        new_points = 0;
    }
}
inline OctoTree* OctoTree::make_child(int xyz[3]) {  // :184-190, :280-285
    OctoTree* c = new OctoTree(owner, max_layer, layer + 1, layer_init_num, max_points_size, planer_threshold);
    for (int k = 0; k < 3; k++) c->voxel_center[k] = voxel_center[k] + (2 * xyz[k] - 1) * quater_length;  // int*float -> float, + double
    c->quater_length = quater_length / 2;
    return c;
}
inline void OctoTree::cut_octo_tree() {
    if (layer >= max_layer) { octo_state = 0; return; }
    for (size_t i = 0; i < temp_points.size(); i++) {
        int xyz[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) if (temp_points[i].p[k] > voxel_center[k]) xyz[k] = 1;
        const int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
        if (leaves[leafnum] == nullptr) leaves[leafnum] = make_child(xyz);
        leaves[leafnum]->temp_points.push_back(temp_points[i]);
        leaves[leafnum]->new_points++;
    }
    for (int i = 0; i < 8; i++) {
        OctoTree* l = leaves[i];
        if (l != nullptr && (int)l->temp_points.size() > l->octo_init_size) {
            init_plane(l->temp_points, &l->plane);
            if (l->plane.is_plane) l->octo_state = 0;
            else { l->octo_state = 1; l->cut_octo_tree(); }
            l->init_octo = true;
            l->new_points = 0;
        }
    }
}
inline void OctoTree::update(const PointWithVar& pv) {  // UpdateOctoTree :219-308 (state machine)
    if (!init_octo) {
        new_points++;
        temp_points.push_back(pv);
        if ((int)temp_points.size() > octo_init_size) init_octo_tree();
    } else if (plane.is_plane) {
        if (update_enable) {
            new_points++;
            temp_points.push_back(pv);
            if (new_points > update_size_threshold) { init_plane(temp_points, &plane); new_points = 0; }
            if ((int)temp_points.size() >= max_points_size) {
                update_enable = false;
                std::vector<PointWithVar>().swap(temp_points);
                new_points = 0;
            }
        }
    } else if (layer < max_layer) {
        if (!temp_points.empty()) std::vector<PointWithVar>().swap(temp_points);
        int xyz[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) if (pv.p[k] > voxel_center[k]) xyz[k] = 1;
        const int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
        if (leaves[leafnum] == nullptr) leaves[leafnum] = make_child(xyz);
        leaves[leafnum]->update(pv);
    } else if (update_enable) {
        new_points++;
        temp_points.push_back(pv);
        if (new_points > update_size_threshold) { init_plane(temp_points, &plane); new_points = 0; }
        if ((int)temp_points.size() > owner->g_max_points) {
            update_enable = false;
            std::vector<PointWithVar>().swap(temp_points);
        }
    }
}

inline OctoTree* new_root(VoxelMap& vm, const Key& k) {  // voxel_mapping.cpp:137-146 / 344-351
    const float voxel_size = (float)vm.cfg.voxel_size;
    OctoTree* o = new OctoTree(&vm, vm.cfg.max_layer, 0, vm.cfg.layer_init, vm.cfg.max_points_size, (float)vm.cfg.planer_threshold);
    o->quater_length = voxel_size / 4;
    o->voxel_center[0] = (0.5 + k.x) * voxel_size;
    o->voxel_center[1] = (0.5 + k.y) * voxel_size;
    o->voxel_center[2] = (0.5 + k.z) * voxel_size;
    return o;
}

// ---- a6: buildVoxelMap, voxel_mapping.cpp:110-151 (init order over voxels is irrelevant to results) -----
inline void build_voxel_map(VoxelMap& vm, const std::vector<PointWithVar>& pts) {
    const float voxel_size = (float)vm.cfg.voxel_size;
    std::vector<OctoTree*> touched;
    for (const auto& pv : pts) {
        const double q[3] = {pv.p[0] / voxel_size, pv.p[1] / voxel_size, pv.p[2] / voxel_size};
        const Key k = key_from_quotient(q);
        auto it = vm.map.find(k);
        OctoTree* o;
        if (it == vm.map.end()) { o = new_root(vm, k); vm.map[k] = o; }
        else o = it->second;
        o->temp_points.push_back(pv);
        o->new_points++;
    }
    for (auto& kv : vm.map) kv.second->init_octo_tree();  // :147-150 (re-inits every voxel in the map, as the reference does)
}
// ---- a16: updateVoxelMap, voxel_mapping.cpp:320-354 -------------------------------------------------------
inline void update_voxel_map(VoxelMap& vm, const std::vector<PointWithVar>& pts) {
    const float voxel_size = (float)vm.cfg.voxel_size;
    for (const auto& pv : pts) {
        const double q[3] = {pv.p[0] / voxel_size, pv.p[1] / voxel_size, pv.p[2] / voxel_size};
        const Key k = key_from_quotient(q);
        auto it = vm.map.find(k);
        if (it != vm.map.end()) it->second->update(pv);
        else { OctoTree* o = new_root(vm, k); vm.map[k] = o; o->update(pv); }
    }
}

// ---- a7: calcBodyVar, voxel_mapping.cpp:1221-1241.  DEG2RAD comes from PCL's pcl_macros.h (not in tree):
//      #define DEG2RAD(x) ((x)*0.017453293)
inline void calc_body_var(double pb[3], const float range_inc, const float degree_inc, double var[9]) {
    if (pb[2] == 0) pb[2] = 0.0001;
    const float range = (float)std::sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
    const float range_var = range_inc * range_inc;
    const double sdeg = std::sin((degree_inc) * 0.017453293);
    const double dvar = sdeg * sdeg;  // pow(x,2)
    double dir[3] = {pb[0], pb[1], pb[2]};
    normalize3(dir);
    double dhat[9];
    skew(dir, dhat);
    double b1[3] = {1, 1, -(dir[0] + dir[1]) / dir[2]};
    normalize3(b1);
    double b2[3];
    cross3(b1, dir, b2);
    normalize3(b2);
    // A = range * direction_hat * N  -> (range*direction_hat) * N, N = [b1 b2] (3x2)
    double rd[9];
    for (int i = 0; i < 9; i++) rd[i] = (double)range * dhat[i];
    double A[6];
    for (int i = 0; i < 3; i++) {
        A[i * 2 + 0] = rd[i * 3 + 0] * b1[0] + rd[i * 3 + 1] * b1[1] + rd[i * 3 + 2] * b1[2];
        A[i * 2 + 1] = rd[i * 3 + 0] * b2[0] + rd[i * 3 + 1] * b2[1] + rd[i * 3 + 2] * b2[2];
    }
    // var = direction*range_var*direction^T + A*direction_var*A^T  (direction_var = dvar * I2)
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const double t1 = (dir[i] * (double)range_var) * dir[j];
            const double t2 = (A[i * 2 + 0] * dvar) * A[j * 2 + 0] + (A[i * 2 + 1] * dvar) * A[j * 2 + 1];
            var[i * 3 + j] = t1 + t2;
        }
}

// ---- a8: transformLidar / pointBodyToWorld, voxel_mapping_common.cpp:709-726, :121-131 ---------------
inline void body_to_world_d(const Config& c, const double* R, const double* t, const double p[3], double out[3]) {
    double pi[3], pw[3];
    m3_vec(c.extR, p, pi);
    for (int k = 0; k < 3; k++) pi[k] += c.extT[k];
    m3_vec(R, pi, pw);
    for (int k = 0; k < 3; k++) out[k] = pw[k] + t[k];
}

// ---- a11: build_single_residual, voxel_mapping.cpp:247-318 ---------------------------------------------
// (n_tests: the caller's plane-test count -- a per-thread sum in the matcher's loop; a shared atomic here made every thread of the loop wait for one cache line)
inline void build_single_residual(VoxelMap& vm, const PointWithVar& pv, const OctoTree* oct, int layer, int max_layer, double sigma_num,
                                  bool& is_success, double& prob, Ptpl& out, long& n_tests) {
    const double radius_k = 3;
    const double* pw = pv.pw;
    if (oct->plane.is_plane) {
        const Plane& pl = oct->plane;
        n_tests++;
        const float dis_to_plane = (float)std::fabs(pl.normal[0] * pw[0] + pl.normal[1] * pw[1] + pl.normal[2] * pw[2] + pl.d);
        const float dis_to_center = (float)((pl.center[0] - pw[0]) * (pl.center[0] - pw[0]) + (pl.center[1] - pw[1]) * (pl.center[1] - pw[1]) +
                                            (pl.center[2] - pw[2]) * (pl.center[2] - pw[2]));
        const float range_dis = std::sqrt(dis_to_center - dis_to_plane * dis_to_plane);  // float arithmetic; NaN -> rejected
        if (range_dis <= radius_k * pl.radius) {
            const double J[6] = {pw[0] - pl.center[0], pw[1] - pl.center[1], pw[2] - pl.center[2], -pl.normal[0], -pl.normal[1], -pl.normal[2]};
            double tmp[6];
            for (int c = 0; c < 6; c++) {
                double s = 0;
                for (int r = 0; r < 6; r++) s += J[r] * pl.plane_var[r * 6 + c];
                tmp[c] = s;
            }
            double sigma_l = 0;
            for (int c = 0; c < 6; c++) sigma_l += tmp[c] * J[c];
            double vn[3];
            m3t_vec(pv.var, pl.normal, vn);  // n^T * var
            sigma_l += vn[0] * pl.normal[0] + vn[1] * pl.normal[1] + vn[2] * pl.normal[2];
            if (dis_to_plane < sigma_num * std::sqrt(sigma_l)) {
                is_success = true;
                const double this_prob = 1.0 / (std::sqrt(sigma_l)) * std::exp(-0.5 * dis_to_plane * dis_to_plane / sigma_l);
                if (this_prob > prob) {
                    prob = this_prob;
                    for (int k = 0; k < 3; k++) { out.point[k] = pv.p[k]; out.normal[k] = pl.normal[k]; out.center[k] = pl.center[k]; }
                    std::memcpy(out.plane_var, pl.plane_var, sizeof(out.plane_var));
                    out.d = pl.d;
                    out.layer = layer;
                }
            }
        }
        return;
    }
    if (layer < max_layer)
        for (int l = 0; l < 8; l++)
            if (oct->leaves[l] != nullptr) build_single_residual(vm, pv, oct->leaves[l], layer + 1, max_layer, sigma_num, is_success, prob, out, n_tests);
}

// ---- a10: BuildResidualListOMP, voxel_mapping.cpp:153-245 (serial here; result is order-independent) ---
inline void build_residual_list(VoxelMap& vm, const std::vector<PointWithVar>& pv_list, std::vector<Ptpl>& ptpl_list, std::vector<int>& match_idx) {
    const double voxel_size = vm.cfg.voxel_size;  // double in the matcher (:153)
    ptpl_list.clear();
    match_idx.clear();
    const long n = (long)pv_list.size();
    std::vector<Ptpl> single(pv_list.size());
    std::vector<uint8_t> good(pv_list.size(), 0);
    // the reference runs this loop under OpenMP (MP_PROC_NUM = 4, CMakeLists.txt:21-24); points are independent, the list is compacted in index order
    long n_tests = 0, n_extra = 0;
#pragma omp parallel for schedule(static) num_threads(vm.threads) if (vm.threads > 1) reduction(+ : n_tests, n_extra)
    for (long i = 0; i < n; i++) {
        const PointWithVar& pv = pv_list[(size_t)i];
        const double q[3] = {pv.pw[0] / voxel_size, pv.pw[1] / voxel_size, pv.pw[2] / voxel_size};
        float loc[3];
        for (int j = 0; j < 3; j++) { loc[j] = (float)q[j]; if (loc[j] < 0) loc[j] -= 1.0; }
        Key pos{(int64_t)loc[0], (int64_t)loc[1], (int64_t)loc[2]};
        auto it = vm.map.find(pos);
        if (it == vm.map.end()) continue;
        OctoTree* cur = it->second;
        bool ok = false;
        double prob = 0;
        build_single_residual(vm, pv, cur, 0, vm.cfg.max_layer, vm.cfg.sigma_num, ok, prob, single[(size_t)i], n_tests);
        if (!ok) {  // near-voxel retry, literal unit-mismatch quirk, :190-222
            Key nearp = pos;
            int64_t* nk[3] = {&nearp.x, &nearp.y, &nearp.z};
            for (int k = 0; k < 3; k++) {
                if (loc[k] > (cur->voxel_center[k] + cur->quater_length)) *nk[k] = *nk[k] + 1;
                else if (loc[k] < (cur->voxel_center[k] - cur->quater_length)) *nk[k] = *nk[k] - 1;
            }
            n_extra++;
            auto itn = vm.map.find(nearp);
            if (itn != vm.map.end()) build_single_residual(vm, pv, itn->second, 0, vm.cfg.max_layer, vm.cfg.sigma_num, ok, prob, single[(size_t)i], n_tests);
        }
        good[(size_t)i] = ok ? 1 : 0;
    }
    vm.cnt.n_plane_tests += n_tests; vm.cnt.n_extra_probe += n_extra;
    for (long i = 0; i < n; i++)
        if (good[(size_t)i]) { ptpl_list.push_back(single[(size_t)i]); match_idx.push_back((int)i); }
}

// ---- StatesGroup boxplus / boxminus, include/common_lib.h:249-271 -----------------------------------------
inline void state_plus(State& s, const double* d) {
    double E[9], Rn[9];
    so3_exp(d[0], d[1], d[2], E);
    m3_mul(s.R, E, Rn);
    std::memcpy(s.R, Rn, sizeof(Rn));
    for (int k = 0; k < 3; k++) { s.t[k] += d[3 + k]; s.vel[k] += d[6 + k]; s.bg[k] += d[9 + k]; s.ba[k] += d[12 + k]; s.g[k] += d[15 + k]; }
}
inline void state_minus(const State& a, const State& b, double* out) {  // a - b
    double Rt[9], rotd[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rt[i * 3 + j] = b.R[j * 3 + i];
    m3_mul(Rt, a.R, rotd);
    so3_log(rotd, out);
    for (int k = 0; k < 3; k++) { out[3 + k] = a.t[k] - b.t[k]; out[6 + k] = a.vel[k] - b.vel[k]; out[9 + k] = a.bg[k] - b.bg[k]; out[12 + k] = a.ba[k] - b.ba[k]; out[15 + k] = a.g[k] - b.g[k]; }
}

struct RegDebug {  // per-iteration intermediates exposed for parity tests
    std::vector<double> HTH, HTz;      // 36 / 6 per iteration
    std::vector<int> n_match;
    std::vector<int> match_idx_last;   // matches of the last iteration
    std::vector<double> normals_last;  // 3 per match (double normals as stored in the map)
    std::vector<float> dis_last;
    std::vector<double> rinv_last;
    double res_mean_last = 0;
    // per iteration, as it ended (tests/test_ref_lio.py compares them with the reference's own lio_state_estimation, oracle/ref_lio)
    std::vector<double> sol, G, state24, cov, res_mean;   // 18 / 324 / 24 (R, t, vel, bg, ba, g) / 324 / 1
    std::vector<int> converged, stopped;
};

struct Registration {
    VoxelMap* vm;
    std::vector<double> body_cov;   // m_body_cov_list
    std::vector<double> cross_mat;  // m_cross_mat_list
    // test tap (orc_debug_tap): the Point_with_var lists the three callers hand to buildVoxelMap / updateVoxelMap (in scan order, before the
    // var_contrast sort) / BuildResidualListOMP (first iteration of the call) -- the inputs the reference's own functions are fed in tests/test_ref_voxelmap.py
    bool tap_on = false;
    std::vector<PointWithVar> tap[3];
    explicit Registration(VoxelMap* v) : vm(v) {}

    // voxel_mapping.cpp:1302-1316
    void prepare(const float* pts, int n) {
        const Config& c = vm->cfg;
        body_cov.resize((size_t)n * 9);
        cross_mat.resize((size_t)n * 9);
        for (int i = 0; i < n; i++) {
            double p[3] = {pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2]};
            if (p[2] == 0) p[2] = 0.001;
            calc_body_var(p, (float)c.dept_err, (float)c.beam_err, &body_cov[(size_t)i * 9]);
            double pi[3];
            m3_vec(c.extR, p, pi);
            for (int k = 0; k < 3; k++) pi[k] += c.extT[k];
            skew(pi, &cross_mat[(size_t)i * 9]);
        }
    }

    // a9 + a10: world transform (f64 compute, f32 store), covariance propagation :1344-1359
    void make_pv_list(const float* pts, int n, const State& s, std::vector<PointWithVar>& pv_list) {
        const Config& c = vm->cfg;
        pv_list.resize(n);
        const double* rot_var_src = s.cov;
        double rot_var[9], t_var[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { rot_var[i * 3 + j] = rot_var_src[i * 18 + j]; t_var[i * 3 + j] = s.cov[(3 + i) * 18 + (3 + j)]; }
        for (int i = 0; i < n; i++) {
            PointWithVar& pv = pv_list[i];
            const double p[3] = {pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2]};
            double pw[3];
            body_to_world_d(c, s.R, s.t, p, pw);
            for (int k = 0; k < 3; k++) { pv.p[k] = p[k]; pv.pw[k] = (double)(float)pw[k]; }
            double cov[9], nc[9], nct[9], term2[9];
            m3_sandwich(s.R, &body_cov[(size_t)i * 9], cov);
            const double* cm = &cross_mat[(size_t)i * 9];
            for (int k = 0; k < 9; k++) nc[k] = -cm[k];
            // (-crossmat) * rot_var * (-crossmat.transpose())
            for (int r = 0; r < 3; r++)
                for (int cc = 0; cc < 3; cc++) nct[r * 3 + cc] = -cm[cc * 3 + r];
            double tmp[9];
            m3_mul(nc, rot_var, tmp);
            m3_mul(tmp, nct, term2);
            for (int k = 0; k < 9; k++) pv.var[k] = (cov[k] + term2[k]) + t_var[k];
        }
    }

    // a12-a14 + A.13: one full iterated update (lio_state_estimation, voxel_mapping.cpp:1284-1652)
    int run(const float* pts, int n, const State& state_propagat, State& state, RegDebug* dbg = nullptr) {
        const Config& c = vm->cfg;
        prepare(pts, n);
        int rematch_num = 0;
        double G[324], HTH18[324];
        std::memset(G, 0, sizeof(G));
        std::memset(HTH18, 0, sizeof(HTH18));
        std::vector<PointWithVar> pv_list;
        std::vector<Ptpl> ptpl_list;
        std::vector<int> match_idx;
        int iters = 0;
        vm->cnt.n_ds = n;
        for (int it = 0; it < c.max_iter; it++) {
            iters++;
            make_pv_list(pts, n, state, pv_list);
            if (tap_on && it == 0) tap[2] = pv_list;
            build_residual_list(*vm, pv_list, ptpl_list, match_idx);
            const int M = (int)ptpl_list.size();
            vm->cnt.n_match += M;
            // a12 :1372-1392 -- normals stored in a float cloud, residual from the unrounded world point
            std::vector<float> nrm((size_t)M * 3), dis(M);
            double total_residual = 0;
            for (int i = 0; i < M; i++) {
                double pwd[3];
                body_to_world_d(c, state.R, state.t, ptpl_list[i].point, pwd);
                const float nx = (float)ptpl_list[i].normal[0], ny = (float)ptpl_list[i].normal[1], nz = (float)ptpl_list[i].normal[2];
                const float d = (float)(pwd[0] * nx + pwd[1] * ny + pwd[2] * nz + ptpl_list[i].d);
                nrm[i * 3 + 0] = nx; nrm[i * 3 + 1] = ny; nrm[i * 3 + 2] = nz;
                dis[i] = d;
                total_residual += std::fabs(d);
            }
            // a13 :1487-1575
            double HTH[36], HTz[6];
            for (int k = 0; k < 36; k++) HTH[k] = 0;
            for (int k = 0; k < 6; k++) HTz[k] = 0;
            std::vector<double> rinv(M);
            double RextR[9], Rt[9];
            m3_mul(state.R, c.extR, RextR);
            for (int r = 0; r < 3; r++)
                for (int cc = 0; cc < 3; cc++) Rt[r * 3 + cc] = state.R[cc * 3 + r];
            for (int i = 0; i < M; i++) {
                const double pb[3] = {ptpl_list[i].point[0], ptpl_list[i].point[1], ptpl_list[i].point[2]};
                double pthis[3];
                m3_vec(c.extR, pb, pthis);
                for (int k = 0; k < 3; k++) pthis[k] += c.extT[k];
                double cm[9];
                skew(pthis, cm);
                const double nv[3] = {nrm[i * 3 + 0], nrm[i * 3 + 1], nrm[i * 3 + 2]};
                double pworld[3];
                m3_vec(state.R, pthis, pworld);
                for (int k = 0; k < 3; k++) pworld[k] += state.t[k];
                double var[9], varw[9];
                calc_body_var(pthis, (float)c.dept_err, c.calib_laser ? (float)0.01 : (float)c.beam_err, var);  // CALIB_ANGLE_COV common_lib.h:41
                m3_sandwich(RextR, var, varw);
                const Ptpl& pp = ptpl_list[i];
                const double J[6] = {pworld[0] - pp.center[0], pworld[1] - pp.center[1], pworld[2] - pp.center[2], -pp.normal[0], -pp.normal[1], -pp.normal[2]};
                double tmp[6];
                for (int cc = 0; cc < 6; cc++) { double s = 0; for (int r = 0; r < 6; r++) s += J[r] * pp.plane_var[r * 6 + cc]; tmp[cc] = s; }
                double sigma_l = 0;
                for (int cc = 0; cc < 6; cc++) sigma_l += tmp[cc] * J[cc];
                double vn[3];
                m3t_vec(varw, nv, vn);
                const double nvn = vn[0] * nv[0] + vn[1] * nv[1] + vn[2] * nv[2];
                const double ri = 1.0 / (sigma_l + nvn);
                rinv[i] = ri;
                // A = point_crossmat * R^T * norm_vec
                double T1[9], A[3];
                m3_mul(cm, Rt, T1);
                m3_vec(T1, nv, A);
                const double H[6] = {A[0], A[1], A[2], nv[0], nv[1], nv[2]};
                double HR[6];
                for (int k = 0; k < 6; k++) HR[k] = H[k] * ri;
                const double meas = -(double)dis[i];
                for (int r = 0; r < 6; r++) {
                    for (int cc = 0; cc < 6; cc++) HTH[r * 6 + cc] += HR[r] * H[cc];
                    HTz[r] += HR[r] * meas;
                }
            }
            if (dbg) {
                dbg->HTH.insert(dbg->HTH.end(), HTH, HTH + 36);
                dbg->HTz.insert(dbg->HTz.end(), HTz, HTz + 6);
                dbg->n_match.push_back(M);
                dbg->match_idx_last = match_idx;
                dbg->normals_last.resize((size_t)M * 3);
                for (int i = 0; i < M; i++) for (int k = 0; k < 3; k++) dbg->normals_last[i * 3 + k] = ptpl_list[i].normal[k];
                dbg->dis_last = dis;
                dbg->rinv_last = rinv;
                dbg->res_mean_last = M ? total_residual / M : 0;
            }
            // a14 :1585-1646
            for (int r = 0; r < 6; r++)
                for (int cc = 0; cc < 6; cc++) HTH18[r * 18 + cc] = HTH[r * 6 + cc];
            double covinv[324], S[324], K1[324];
            inv_gauss_jordan(state.cov, covinv, 18);
            for (int k = 0; k < 324; k++) S[k] = HTH18[k] + covinv[k];
            inv_gauss_jordan(S, K1, 18);
            for (int r = 0; r < 18; r++)
                for (int cc = 0; cc < 6; cc++) {
                    double s = 0;
                    for (int k = 0; k < 6; k++) s += K1[r * 18 + k] * HTH[k * 6 + cc];
                    G[r * 18 + cc] = s;
                }
            double vec[18], sol[18];
            state_minus(state_propagat, state, vec);
            for (int r = 0; r < 18; r++) {
                double s1 = 0, s2 = 0;
                for (int k = 0; k < 6; k++) { s1 += K1[r * 18 + k] * HTz[k]; s2 += G[r * 18 + k] * vec[k]; }
                sol[r] = (s1 + vec[r]) - s2;
            }
            state_plus(state, sol);
            const double rn = std::sqrt(sol[0] * sol[0] + sol[1] * sol[1] + sol[2] * sol[2]);
            const double tn = std::sqrt(sol[3] * sol[3] + sol[4] * sol[4] + sol[5] * sol[5]);
            const bool converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
            if (converged || ((rematch_num == 0) && (it == (c.max_iter - 2)))) rematch_num++;
            const bool stop_now = rematch_num >= 2 || (it == c.max_iter - 1);
            auto tap_iteration = [&]() {
                if (!dbg) return;
                dbg->sol.insert(dbg->sol.end(), sol, sol + 18);
                dbg->G.insert(dbg->G.end(), G, G + 324);
                dbg->state24.insert(dbg->state24.end(), state.R, state.R + 9);
                const double* parts[5] = {state.t, state.vel, state.bg, state.ba, state.g};
                for (const double* q : parts) dbg->state24.insert(dbg->state24.end(), q, q + 3);
                dbg->cov.insert(dbg->cov.end(), state.cov, state.cov + 324);
                dbg->res_mean.push_back(M ? total_residual / M : 0);
                dbg->converged.push_back(converged ? 1 : 0);
                dbg->stopped.push_back(stop_now ? 1 : 0);
            };
            if (!stop_now) tap_iteration();
            if (stop_now) {
                double IG[324], nc[324];
                for (int r = 0; r < 18; r++)
                    for (int cc = 0; cc < 18; cc++) IG[r * 18 + cc] = ((r == cc) ? 1.0 : 0.0) - G[r * 18 + cc];
                for (int r = 0; r < 18; r++)
                    for (int cc = 0; cc < 18; cc++) { double s = 0; for (int k = 0; k < 18; k++) s += IG[r * 18 + k] * state.cov[k * 18 + cc]; nc[r * 18 + cc] = s; }
                std::memcpy(state.cov, nc, sizeof(nc));
                tap_iteration();
                break;
            }
        }
        vm->cnt.n_iter += iters;
        return iters;
    }

    // a6: voxel_map_init, voxel_mapping.cpp:1243-1281 (uses the FULL undistorted scan; crossmat of the lidar-frame point)
    void map_init(const float* pts_raw, int n, const State& s) {
        const Config& c = vm->cfg;
        std::vector<PointWithVar> pv_list(n);
        double rot_var[9], t_var[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { rot_var[i * 3 + j] = s.cov[i * 18 + j]; t_var[i * 3 + j] = s.cov[(3 + i) * 18 + (3 + j)]; }
        for (int i = 0; i < n; i++) {
            double p[3] = {pts_raw[i * 3 + 0], pts_raw[i * 3 + 1], pts_raw[i * 3 + 2]};
            double pw[3];
            body_to_world_d(c, s.R, s.t, p, pw);
            PointWithVar& pv = pv_list[i];
            for (int k = 0; k < 3; k++) { pv.p[k] = (double)(float)pw[k]; pv.pw[k] = 0; }
            double var[9];
            calc_body_var(p, (float)c.dept_err, (float)c.beam_err, var);  // may set p[2]=1e-4 before the crossmat below
            double cm[9], nc[9], a[9], tmp[9], b[9];
            skew(p, cm);
            for (int k = 0; k < 9; k++) nc[k] = -cm[k];
            m3_sandwich(s.R, var, a);
            m3_mul(nc, rot_var, tmp);
            m3_mul_bt(tmp, nc, b);
            for (int k = 0; k < 9; k++) pv.var[k] = (a[k] + b[k]) + t_var[k];
        }
        if (tap_on) tap[0] = pv_list;
        build_voxel_map(*vm, pv_list);
    }

    // a15: map_incremental_grow, ImMesh_mesh_reconstruction.cpp:377-424 (needs prepare() of this scan: run() did it)
    void map_grow(const float* pts, int n, const State& s) {
        const Config& c = vm->cfg;
        std::vector<PointWithVar> pv_list(n);
        double RextR[9], rot_var[9], t_var[9];
        m3_mul(s.R, c.extR, RextR);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { rot_var[i * 3 + j] = s.cov[i * 18 + j]; t_var[i * 3 + j] = s.cov[(3 + i) * 18 + (3 + j)]; }
        for (int i = 0; i < n; i++) {
            const double p[3] = {pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2]};
            double pw[3];
            body_to_world_d(c, s.R, s.t, p, pw);
            PointWithVar& pv = pv_list[i];
            for (int k = 0; k < 3; k++) { pv.p[k] = (double)(float)pw[k]; pv.pw[k] = 0; }
            const double* cm = &cross_mat[(size_t)i * 9];
            double nc[9], a[9], tmp[9], b[9];
            for (int k = 0; k < 9; k++) nc[k] = -cm[k];
            m3_sandwich(RextR, &body_cov[(size_t)i * 9], a);
            m3_mul(nc, rot_var, tmp);
            m3_mul_bt(tmp, nc, b);
            for (int k = 0; k < 9; k++) pv.var[k] = (a[k] + b[k]) + t_var[k];
        }
        if (tap_on) tap[1] = pv_list;
        // var_contrast (voxel_mapping.cpp:49): ascending ||diag(var)||; std::sort ties broken by original index here
        std::vector<double> key(n);
        for (int i = 0; i < n; i++) key[i] = std::sqrt(pv_list[i].var[0] * pv_list[i].var[0] + pv_list[i].var[4] * pv_list[i].var[4] + pv_list[i].var[8] * pv_list[i].var[8]);
        std::vector<int> order(n);
        for (int i = 0; i < n; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a_, int b_) { return key[a_] < key[b_]; });
        std::vector<PointWithVar> sorted(n);
        for (int i = 0; i < n; i++) sorted[i] = pv_list[order[i]];
        update_voxel_map(*vm, sorted);
    }
};

}  // namespace orc
