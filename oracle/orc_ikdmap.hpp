// ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
// CPU restatement of the LEGACY registration path of ImMesh (`voxel_map_en = false`, dead in every shipped config):
//   map        KD_TREE::Build / Add_Points(.., downsample_on = true)          include/ikd-Tree/ikd_Tree.cpp:283-310, 493-602
//   matcher    "Old map ICP" branch of Voxel_mapping::lio_state_estimation     src/voxel_mapping.cpp:1400-1480
//   plane fit  esti_plane (5 neighbours, float colPivHouseholderQr)            include/common_lib.h:356-402
//   H / EKF    src/voxel_mapping.cpp:1487-1650 with R_inv = 1 / LASER_POINT_COV
// The tree itself is not restated -- only what it computes: the exact k nearest points in float arithmetic, and the box-downsample insert
// (one survivor per downsample_size box: the point nearest to the box centre, a new point winning ties).  Both are checked against the
// reference's own ikd-Tree (oracle/_ref) in tests/test_ikdmap.py.  Conventions where the reference is order-dependent or unpinned:
//   * "points inside the box" = points with the same floor(x / downsample_size) cell (differs from the float box test only within an ulp of a face)
//   * equal k-NN distances are ordered by insertion sequence
//   * Eigen's colPivHouseholderQr is restated from its published algorithm with sequential float sums (Eigen is not in the image: unpinned)
#pragma once
#include "orc_voxelmap.hpp"
#include <unordered_map>

namespace orc {

struct IkdPt { float x, y, z; long id; };

inline float ikd_dist(float ax, float ay, float az, float bx, float by, float bz) {   // KD_TREE::calc_dist, ikd_Tree.cpp:1722-1728
    return (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz);
}

struct IkdMap {
    float ds = 0.5f;                                  // downsample_size (set_downsample_param(m_filter_size_map_min), voxel_mapping.cpp:1909)
    std::unordered_map<uint64_t, std::vector<IkdPt>> cells;
    long next_id = 0;
    size_t count = 0;
    static uint64_t key(long x, long y, long z) { return (((uint64_t)(x + (1 << 20)) & 0x1FFFFF) << 42) | (((uint64_t)(y + (1 << 20)) & 0x1FFFFF) << 21) | ((uint64_t)(z + (1 << 20)) & 0x1FFFFF); }
    long cell(float v) const { return (long)std::floor(v / ds); }
    void clear() { cells.clear(); next_id = 0; count = 0; localmap_initialized = false; }
    void insert_raw(float x, float y, float z) { cells[key(cell(x), cell(y), cell(z))].push_back({x, y, z, next_id++}); count++; }
    void build(const float* xyz, int n) { clear(); for (int i = 0; i < n; i++) insert_raw(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]); }   // Build keeps every point
    // Add_Points(PointToAdd, true), ikd_Tree.cpp:493-545; returns the number of points actually added (tmp_counter)
    int add_points(const float* xyz, int n) {
        int added = 0;
        for (int i = 0; i < n; i++) {
            const float px = xyz[i * 3], py = xyz[i * 3 + 1], pz = xyz[i * 3 + 2];
            if (count == 0) { insert_raw(px, py, pz); continue; }   // Root_Node == nullptr: Build({PointToAdd[0]}) -- as written, with the FIRST point; only reachable with i == 0
            float vmin[3], vmax[3], mid[3];
            const float p[3] = {px, py, pz};
            for (int a = 0; a < 3; a++) {
                vmin[a] = (float)(std::floor(p[a] / ds) * ds);
                vmax[a] = vmin[a] + ds;
                mid[a] = (float)(vmin[a] + (vmax[a] - vmin[a]) / 2.0);
            }
            std::vector<IkdPt>& box = cells[key(cell(px), cell(py), cell(pz))];
            float min_dist = ikd_dist(px, py, pz, mid[0], mid[1], mid[2]);
            int best = -1;                                                    // -1: the new point
            for (size_t k = 0; k < box.size(); k++) {
                const float d = ikd_dist(box[k].x, box[k].y, box[k].z, mid[0], mid[1], mid[2]);
                if (d < min_dist) { min_dist = d; best = (int)k; }
            }
            const bool same = best < 0 || (std::fabs(px - box[best].x) < 1e-6 && std::fabs(py - box[best].y) < 1e-6 && std::fabs(pz - box[best].z) < 1e-6);
            if (box.size() > 1 || same) {
                IkdPt keep = best < 0 ? IkdPt{px, py, pz, 0} : box[best];
                count -= box.size();
                box.clear();                                                   // Delete_by_range(box)
                keep.id = next_id++;                                           // Add_by_point: a fresh insertion
                box.push_back(keep);
                count++; added++;
            }
        }
        return added;
    }
    // exact k nearest in float arithmetic, ascending (d2, insertion id); brute force (the checker only sees small maps)
    int knn(float qx, float qy, float qz, int k, IkdPt* out, float* d2) const {
        std::vector<std::pair<std::pair<float, long>, IkdPt>> best;
        for (const auto& kv : cells)
            for (const IkdPt& p : kv.second) {
                const float d = ikd_dist(qx, qy, qz, p.x, p.y, p.z);
                if ((int)best.size() < k || std::make_pair(d, p.id) < best.back().first) {
                    best.push_back({{d, p.id}, p});
                    std::sort(best.begin(), best.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
                    if ((int)best.size() > k) best.pop_back();
                }
            }
        for (size_t i = 0; i < best.size(); i++) { out[i] = best[i].second; d2[i] = best[i].first.first; }
        return (int)best.size();
    }
    // KD_TREE::Delete_Point_Boxes -> Delete_by_range (ikd_Tree.cpp:655-690, 1340-1420): a point goes when min <= p < max on every axis
    int delete_boxes(const float* boxes /* nb x 6: min xyz, max xyz */, int nb) {
        int removed = 0;
        for (auto& kv : cells) {
            std::vector<IkdPt>& v = kv.second;
            size_t w = 0;
            for (size_t k = 0; k < v.size(); k++) {
                bool in = false;
                for (int b = 0; b < nb && !in; b++) {
                    const float* q = boxes + b * 6;
                    in = q[0] <= v[k].x && q[3] > v[k].x && q[1] <= v[k].y && q[4] > v[k].y && q[2] <= v[k].z && q[5] > v[k].z;
                }
                if (in) removed++; else v[w++] = v[k];
            }
            v.resize(w);
        }
        count -= (size_t)removed;
        return removed;
    }
    // Voxel_mapping::laser_map_fov_segment (src/voxel_mapping_common.cpp:214-288): the local-map cube follows the sensor; what falls out is deleted
    bool localmap_initialized = false;
    float lm_min[3] = {0, 0, 0}, lm_max[3] = {0, 0, 0};
    int fov_segment(const double* pos_lid, double cube_len, float detection_range, std::vector<float>* boxes_out = nullptr) {
        const float MOV_THRESHOLD = 1.5f;   // src/voxel_mapping.hpp:137
        if (boxes_out) boxes_out->clear();
        if (!localmap_initialized) {
            for (int i = 0; i < 3; i++) { lm_min[i] = (float)(pos_lid[i] - cube_len / 2.0); lm_max[i] = (float)(pos_lid[i] + cube_len / 2.0); }
            localmap_initialized = true;
            return 0;
        }
        float dist[3][2];
        bool need_move = false;
        for (int i = 0; i < 3; i++) {
            dist[i][0] = (float)std::fabs(pos_lid[i] - (double)lm_min[i]);
            dist[i][1] = (float)std::fabs(pos_lid[i] - (double)lm_max[i]);
            if (dist[i][0] <= MOV_THRESHOLD * detection_range || dist[i][1] <= MOV_THRESHOLD * detection_range) need_move = true;
        }
        if (!need_move) return 0;
        float nmin[3], nmax[3];
        std::memcpy(nmin, lm_min, sizeof(nmin)); std::memcpy(nmax, lm_max, sizeof(nmax));
        const float mov_dist = (float)std::max((cube_len - 2.0 * MOV_THRESHOLD * detection_range) * 0.5 * 0.9, double(detection_range * (MOV_THRESHOLD - 1)));
        std::vector<float> boxes;
        for (int i = 0; i < 3; i++) {
            float b[6] = {lm_min[0], lm_min[1], lm_min[2], lm_max[0], lm_max[1], lm_max[2]};
            if (dist[i][0] <= MOV_THRESHOLD * detection_range) {
                nmax[i] -= mov_dist; nmin[i] -= mov_dist;
                b[i] = lm_max[i] - mov_dist;
                boxes.insert(boxes.end(), b, b + 6);
            } else if (dist[i][1] <= MOV_THRESHOLD * detection_range) {
                nmax[i] += mov_dist; nmin[i] += mov_dist;
                b[3 + i] = lm_min[i] + mov_dist;
                boxes.insert(boxes.end(), b, b + 6);
            }
        }
        std::memcpy(lm_min, nmin, sizeof(nmin)); std::memcpy(lm_max, nmax, sizeof(nmax));
        if (boxes_out) *boxes_out = boxes;
        return boxes.empty() ? 0 : delete_boxes(boxes.data(), (int)boxes.size() / 6);
    }
    void dump(std::vector<float>& xyz) const {   // ascending (cell key, insertion id)
        std::vector<std::pair<std::pair<uint64_t, long>, IkdPt>> all;
        for (const auto& kv : cells) for (const IkdPt& p : kv.second) all.push_back({{kv.first, p.id}, p});
        std::sort(all.begin(), all.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        xyz.clear();
        for (const auto& e : all) { xyz.push_back(e.second.x); xyz.push_back(e.second.y); xyz.push_back(e.second.z); }
    }
};

// x = A.colPivHouseholderQr().solve(b) for a 5x3 float system (Eigen ColPivHouseholderQR::computeInPlace + _solve_impl)
inline void qr_solve_5x3f(const float Ain[5][3], const float bin[5], float x[3]) {
    const int rows = 5, cols = 3, size = 3;
    float A[5][3], b[5], hc[3], nrmU[3], nrmD[3];
    int perm[3] = {0, 1, 2};
    for (int r = 0; r < rows; r++) { for (int c = 0; c < cols; c++) A[r][c] = Ain[r][c]; b[r] = bin[r]; }
    const float eps = 1.1920929e-07f;
    float maxn = 0;
    for (int c = 0; c < cols; c++) { float s = 0; for (int r = 0; r < rows; r++) s += A[r][c] * A[r][c]; nrmU[c] = nrmD[c] = std::sqrt(s); maxn = std::max(maxn, nrmU[c]); }
    const float th0 = maxn * eps / (float)rows, threshold_helper = th0 * th0, downdate = std::sqrt(eps);
    int nonzero = size;
    float maxpivot = 0;
    for (int k = 0; k < size; k++) {
        int big = k;
        for (int c = k + 1; c < cols; c++) if (nrmU[c] > nrmU[big]) big = c;
        const float bigsq = nrmU[big] * nrmU[big];
        if (nonzero == size && bigsq < threshold_helper * (float)(rows - k)) nonzero = k;
        if (big != k) {
            for (int r = 0; r < rows; r++) std::swap(A[r][k], A[r][big]);
            std::swap(nrmU[k], nrmU[big]); std::swap(nrmD[k], nrmD[big]); std::swap(perm[k], perm[big]);
        }
        // makeHouseholderInPlace on A[k..rows-1][k]
        float tailsq = 0;
        for (int r = k + 1; r < rows; r++) tailsq += A[r][k] * A[r][k];
        const float c0 = A[k][k];
        float beta, tau;
        if (tailsq <= 1.17549435e-38f) { tau = 0; beta = c0; for (int r = k + 1; r < rows; r++) A[r][k] = 0; }
        else {
            beta = std::sqrt(c0 * c0 + tailsq);
            if (c0 >= 0) beta = -beta;
            for (int r = k + 1; r < rows; r++) A[r][k] = A[r][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        hc[k] = tau;
        A[k][k] = beta;
        if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
        if (tau != 0)
            for (int c = k + 1; c < cols; c++) {   // applyHouseholderOnTheLeft on the trailing columns
                float tmp = 0;
                for (int r = k + 1; r < rows; r++) tmp += A[r][k] * A[r][c];
                tmp += A[k][c];
                A[k][c] -= tau * tmp;
                for (int r = k + 1; r < rows; r++) A[r][c] -= tau * A[r][k] * tmp;
            }
        for (int j = k + 1; j < cols; j++)
            if (nrmU[j] != 0) {
                float temp = std::fabs(A[k][j]) / nrmU[j];
                temp = (1.0f + temp) * (1.0f - temp);
                temp = temp < 0 ? 0 : temp;
                const float q = nrmU[j] / nrmD[j];
                const float temp2 = temp * (q * q);
                if (temp2 <= downdate) {
                    float s = 0;
                    for (int r = k + 1; r < rows; r++) s += A[r][j] * A[r][j];
                    nrmD[j] = std::sqrt(s); nrmU[j] = nrmD[j];
                } else nrmU[j] *= std::sqrt(temp);
            }
    }
    (void)maxpivot;
    for (int k = 0; k < nonzero; k++) {   // c = Q^T b
        if (hc[k] == 0) continue;
        float tmp = 0;
        for (int r = k + 1; r < rows; r++) tmp += A[r][k] * b[r];
        tmp += b[k];
        b[k] -= hc[k] * tmp;
        for (int r = k + 1; r < rows; r++) b[r] -= hc[k] * A[r][k] * tmp;
    }
    float c[3] = {0, 0, 0};
    for (int i = nonzero - 1; i >= 0; i--) {   // upper-triangular back substitution
        float s = b[i];
        for (int j = i + 1; j < nonzero; j++) s -= A[i][j] * c[j];
        c[i] = s / A[i][i];
    }
    x[0] = x[1] = x[2] = 0;
    for (int i = 0; i < nonzero; i++) x[perm[i]] = c[i];
}

// esti_plane(pca_result, points_near, 0.05f), include/common_lib.h:356-402
inline bool esti_plane5(const IkdPt* pt, float threshold, float pabcd[4]) {
    float A[5][3], b[5], nv[3];
    for (int j = 0; j < 5; j++) { A[j][0] = pt[j].x; A[j][1] = pt[j].y; A[j][2] = pt[j].z; b[j] = -1.0f; }
    qr_solve_5x3f(A, b, nv);
    const float n = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    pabcd[0] = nv[0] / n; pabcd[1] = nv[1] / n; pabcd[2] = nv[2] / n;
    pabcd[3] = (float)(1.0 / n);
    for (int j = 0; j < 5; j++)
        if (std::fabs(pabcd[0] * pt[j].x + pabcd[1] * pt[j].y + pabcd[2] * pt[j].z + pabcd[3]) > threshold) return false;
    return true;
}

struct IkdRegResult { int n_iter = 0, n_match = 0; double res_mean = 0; std::vector<int> match_idx; std::vector<float> normals_pd2; double HTH[36], HTz[6]; };

// "Old map ICP": one call = the whole iterated update of lio_state_estimation with m_use_new_map == false
inline void ikd_register(const IkdMap& map, const Config& cfg, const float* body, int n, const State& prior, State& st, double laser_point_cov, IkdRegResult& out) {
    std::vector<uint8_t> selected(n, 1);
    std::vector<IkdPt> near((size_t)n * 5);
    std::vector<int> near_n(n, 0);
    std::vector<float> normvec((size_t)n * 4, 0.f);
    std::vector<double> res_last(n, 1000.0);
    bool nearest_search_en = true;
    int rematch_num = 0;
    double G[324];
    std::memset(G, 0, sizeof(G));
    for (int it = 0; it < cfg.max_iter; it++) {
        out.n_iter = it + 1;
        double RextR[9];
        m3_mul(st.R, cfg.extR, RextR);
        for (int i = 0; i < n; i++) {
            const double pb[3] = {(double)body[i * 3], (double)body[i * 3 + 1], (double)body[i * 3 + 2]};
            double pi[3], pw[3];
            m3_vec(cfg.extR, pb, pi);
            for (int a = 0; a < 3; a++) pi[a] += cfg.extT[a];
            m3_vec(st.R, pi, pw);
            const float wx = (float)(pw[0] + st.t[0]), wy = (float)(pw[1] + st.t[1]), wz = (float)(pw[2] + st.t[2]);   // pointBodyToWorld: f64 compute, f32 store
            if (nearest_search_en) {
                float d2[5];
                near_n[i] = map.knn(wx, wy, wz, 5, &near[(size_t)i * 5], d2);
                selected[i] = (near_n[i] == 5 && !(d2[4] > 5)) ? 1 : 0;
                if (near_n[i] < 5) selected[i] = selected[i] && 0;   // (pointSearchSqDis keeps its stale size-5 contents in the reference; fewer than 5 map points never happens in practice)
            }
            if (!selected[i] || near_n[i] < 5) continue;
            float pabcd[4];
            selected[i] = 0;
            if (esti_plane5(&near[(size_t)i * 5], 0.05f, pabcd)) {
                const float pd2 = pabcd[0] * wx + pabcd[1] * wy + pabcd[2] * wz + pabcd[3];
                const float s = (float)(1 - 0.9 * std::fabs(pd2) / std::sqrt(std::sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2])));
                if (s > 0.9) {
                    selected[i] = 1;
                    normvec[(size_t)i * 4 + 0] = pabcd[0]; normvec[(size_t)i * 4 + 1] = pabcd[1]; normvec[(size_t)i * 4 + 2] = pabcd[2]; normvec[(size_t)i * 4 + 3] = pd2;
                    res_last[i] = std::fabs(pd2);
                }
            }
        }
        out.match_idx.clear(); out.normals_pd2.clear();
        double total = 0;
        std::memset(out.HTH, 0, sizeof(out.HTH)); std::memset(out.HTz, 0, sizeof(out.HTz));
        const double r_inv = 1.0 / laser_point_cov;
        for (int i = 0; i < n; i++) {
            if (!(selected[i] && res_last[i] <= 2.0)) continue;
            out.match_idx.push_back(i);
            for (int a = 0; a < 4; a++) out.normals_pd2.push_back(normvec[(size_t)i * 4 + a]);
            total += res_last[i];
            const double pb[3] = {(double)body[i * 3], (double)body[i * 3 + 1], (double)body[i * 3 + 2]};
            double pt[3], cm[9], T1[9], A[3];
            m3_vec(cfg.extR, pb, pt);
            for (int a = 0; a < 3; a++) pt[a] += cfg.extT[a];
            skew(pt, cm);
            const double nv[3] = {(double)normvec[(size_t)i * 4], (double)normvec[(size_t)i * 4 + 1], (double)normvec[(size_t)i * 4 + 2]};
            m3_mul_bt(cm, st.R, T1);
            m3_vec(T1, nv, A);
            const double H[6] = {A[0], A[1], A[2], nv[0], nv[1], nv[2]};
            const double meas = -(double)normvec[(size_t)i * 4 + 3];
            for (int r = 0; r < 6; r++) {
                const double hr = H[r] * r_inv;
                for (int c = 0; c < 6; c++) out.HTH[r * 6 + c] += hr * H[c];
                out.HTz[r] += hr * meas;
            }
        }
        out.n_match = (int)out.match_idx.size();
        out.res_mean = out.n_match ? total / out.n_match : 0.0;
        // EKF step (voxel_mapping.cpp:1585-1646)
        double HTH18[324], covinv[324], S[324], K1[324];
        std::memset(HTH18, 0, sizeof(HTH18));
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) HTH18[r * 18 + c] = out.HTH[r * 6 + c];
        inv_gauss_jordan(st.cov, covinv, 18);
        for (int k = 0; k < 324; k++) S[k] = HTH18[k] + covinv[k];
        inv_gauss_jordan(S, K1, 18);
        for (int r = 0; r < 18; r++)
            for (int c = 0; c < 6; c++) { double s = 0; for (int k = 0; k < 6; k++) s += K1[r * 18 + k] * out.HTH[k * 6 + c]; G[r * 18 + c] = s; }
        double vec[18], sol[18];
        state_minus(prior, st, vec);
        for (int r = 0; r < 18; r++) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 6; k++) { s1 += K1[r * 18 + k] * out.HTz[k]; s2 += G[r * 18 + k] * vec[k]; }
            sol[r] = (s1 + vec[r]) - s2;
        }
        state_plus(st, sol);
        const double rn = std::sqrt(sol[0] * sol[0] + sol[1] * sol[1] + sol[2] * sol[2]);
        const double tn = std::sqrt(sol[3] * sol[3] + sol[4] * sol[4] + sol[5] * sol[5]);
        const bool converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
        nearest_search_en = false;                                             // :1626-1632 rematch judgement
        if (converged || ((rematch_num == 0) && (it == (cfg.max_iter - 2)))) { nearest_search_en = true; rematch_num++; }
        if (rematch_num >= 2 || (it == cfg.max_iter - 1)) {
            double IG[324], nc[324];
            for (int r = 0; r < 18; r++) for (int c = 0; c < 18; c++) IG[r * 18 + c] = ((r == c) ? 1.0 : 0.0) - G[r * 18 + c];
            for (int r = 0; r < 18; r++)
                for (int c = 0; c < 18; c++) { double s = 0; for (int k = 0; k < 18; k++) s += IG[r * 18 + k] * st.cov[k * 18 + c]; nc[r * 18 + c] = s; }
            std::memcpy(st.cov, nc, sizeof(nc));
            break;
        }
    }
}

}  // namespace orc
