// ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
// CPU restatement of ImMesh's incremental mesher.
// Sequential-deterministic mode: one scan fully meshed before the next, active voxels visited in
// ascending (x,y,z) key order.  Every function cites the reference file:line it follows.
#pragma once
#include "orc_voxelmap.hpp"
#include "orc_delaunay.hpp"
#include <map>
#include <set>
#include <unordered_set>
#include <array>

namespace orc {

struct K3 {
    long x, y, z;
    bool operator==(const K3& o) const { return x == o.x && y == o.y && z == o.z; }
    bool operator<(const K3& o) const { return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z); }
};
struct K3Hash {
    size_t operator()(const K3& k) const {
        uint64_t h = (uint64_t)k.x * 0x9E3779B97F4A7C15ull ^ ((uint64_t)k.y * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull) ^ ((uint64_t)k.z * 0xD6E8FEB86659FD93ull);
        return (size_t)(h ^ (h >> 31));
    }
};
typedef std::array<int, 3> Tri;  // sorted ids (Triangle ctor sorts, triangle.hpp:27-33)
struct TriHash { size_t operator()(const Tri& t) const { return K3Hash()(K3{t[0], t[1], t[2]}); } };

struct MeshVertex { double pos[3]; double smooth[3]; };  // RGB_pts m_pos / m_pos_aft_smooth, pointcloud_rgbd.hpp:90-91
struct MeshVoxel {                                       // RGB_Voxel, pointcloud_rgbd.hpp:168-227
    long pos[3];
    long meshing_times = 0, new_added = 0;
    std::vector<int> pts;
    double short_axis[3] = {0, 0, 0};
};

struct MeshScanOut {
    int vtx_base = 0;
    std::vector<float> new_vtx;         // xyz per new vertex (ids vtx_base, vtx_base+1, ...)
    std::vector<int> tri_add, tri_rem, tri_upd;  // sorted triplets, lexicographically sorted, deduped; tri_upd = surviving triangles whose flip CHANGED
    std::vector<uint8_t> flip_add, flip_upd;
    std::vector<int> smooth_ids;        // vertices whose smoothed position was (re)set this scan (ascending)
    std::vector<double> smooth_xyz;
    int v_act = 0;
    std::vector<int> n_u_list;          // neighbourhood size of every triangulated voxel, in voxel order (diagnostics)
};

struct Mesher {
    Config cfg;
    Counters* cnt = nullptr;
    int threads = 1;   // voxel-parallel neighbourhood search + triangulation (the reference's TBB pool, maximum_thread_for_rec_mesh = 12); results do not depend on it
    std::vector<MeshVertex> verts;                      // m_rgb_pts_vec (index = vertex id)
    std::unordered_map<K3, int, K3Hash> grid;           // m_hashmap_3d_pts : dedupe cell -> vertex id
    std::unordered_map<K3, int, K3Hash> voxel_of;       // m_hashmap_voxels : key -> index in voxels
    std::vector<MeshVoxel> voxels;                      // m_voxel_vec
    // Triangle_manager (triangle.hpp:115-395): every triplet ever inserted (m_triangle_hash) + live adjacency (m_map_pt_triangle)
    std::unordered_map<Tri, int, TriHash> tri_flip;     // triplet -> m_index_flip (entry persists after erase)
    std::unordered_map<int, std::set<Tri>> adj;         // vertex -> live triangles

    // exact k-NN on float xyz, restating KD_TREE::Nearest_Search semantics (include/ikd-Tree/ikd_Tree.cpp:440-476, 1096-1279):
    // d2 = (dx*dx + dy*dy) + dz*dz in float (calc_dist :1722), results ascending.  Only neighbours with
    // sqrt(d2) < r_max are ever used by the callers, so the search is restricted to the voxel lists overlapping the ball.
    struct NN { float d2; int id; };
    void knn_radius(const float q[3], double r, std::vector<NN>& out, long* inspected) const {
        out.clear();
        const double vs = cfg.mesh_voxel;
        const double rr = r * 1.001 + 1e-6;  // superset radius; callers apply the reference's exact float tests
        long lo[3], hi[3];  // voxel index of x is round(x/vs) (monotone), so the ball maps to an index box
        for (int a = 0; a < 3; a++) { lo[a] = (long)std::round(((double)q[a] - rr) / vs); hi[a] = (long)std::round(((double)q[a] + rr) / vs); }
        for (long x = lo[0]; x <= hi[0]; x++)
            for (long y = lo[1]; y <= hi[1]; y++)
                for (long z = lo[2]; z <= hi[2]; z++) {
                    auto it = voxel_of.find(K3{x, y, z});
                    if (it == voxel_of.end()) continue;
                    for (int id : voxels[it->second].pts) {
                        const float px = (float)verts[id].pos[0], py = (float)verts[id].pos[1], pz = (float)verts[id].pos[2];
                        const float d2 = (q[0] - px) * (q[0] - px) + (q[1] - py) * (q[1] - py) + (q[2] - pz) * (q[2] - pz);
                        if (inspected) (*inspected)++;
                        if ((double)std::sqrt(d2) < rr) out.push_back(NN{d2, id});
                    }
                }
        std::sort(out.begin(), out.end(), [](const NN& a, const NN& b) { return a.d2 != b.d2 ? a.d2 < b.d2 : a.id < b.id; });
    }
    // k nearest; exact for every neighbour with sqrt(d2) < r_max (farther ones are never used by the callers).
    // Two-stage: if >= k neighbours already lie within r_max/2 they ARE the global k nearest.
    void knn(const float q[3], int k, double r_max, std::vector<NN>& out, long* inspected = nullptr) const {
        knn_radius(q, r_max * 0.5, out, inspected);
        if ((int)out.size() < k) knn_radius(q, r_max, out, inspected);
        if ((int)out.size() > k) out.resize(k);
    }

    // ---- a17: Global_map::append_points_to_global_map, pointcloud_rgbd.cpp:411-552 -------------------------
    // pts: xyzI float per point (world frame).  Returns the recent-visited voxel set (indices), cleared every call because
    // m_recent_visited_voxel_activated_time == 0 (ImMesh_node.cpp:272).
    void append(const float* pts, int n, int step, std::vector<int>& recent, MeshScanOut& out) {
        std::unordered_set<int> recent_set;
        recent.clear();
        out.vtx_base = (int)verts.size();
        std::vector<NN> nn;
        for (long pi = 0; pi < n; pi += step) {
            const float* p = pts + 4 * pi;
            if (cnt) cnt->n_app++;
            const int gx = (int)std::round(p[0] / cfg.mesh_min_spacing), gy = (int)std::round(p[1] / cfg.mesh_min_spacing), gz = (int)std::round(p[2] / cfg.mesh_min_spacing);
            const int bx = (int)std::round(p[0] / cfg.mesh_voxel), by = (int)std::round(p[1] / cfg.mesh_voxel), bz = (int)std::round(p[2] / cfg.mesh_voxel);
            const bool occupied = grid.find(K3{gx, gy, gz}) != grid.end();
            int vi;
            auto itv = voxel_of.find(K3{bx, by, bz});
            if (itv == voxel_of.end()) {
                vi = (int)voxels.size();
                MeshVoxel v; v.pos[0] = bx; v.pos[1] = by; v.pos[2] = bz;
                voxels.push_back(v);
                voxel_of[K3{bx, by, bz}] = vi;
            } else vi = itv->second;
            if (recent_set.insert(vi).second) recent.push_back(vi);
            if (occupied) continue;
            if (!verts.empty()) {  // m_kdtree.Root_Node != nullptr
                knn(p, 1, cfg.mesh_min_spacing, nn, cnt ? &cnt->c1 : nullptr);
                if (!nn.empty() && (double)std::sqrt(nn[0].d2) < cfg.mesh_min_spacing) continue;
            }
            MeshVertex mv;
            for (int k = 0; k < 3; k++) { mv.pos[k] = p[k]; mv.smooth[k] = p[k]; }  // set_pos, pointcloud_rgbd.cpp:59-66
            const int id = (int)verts.size();
            verts.push_back(mv);
            grid[K3{gx, gy, gz}] = id;
            voxels[vi].pts.push_back(id);
            voxels[vi].new_added++;
            voxels[vi].meshing_times = 0;
            out.new_vtx.push_back(p[0]); out.new_vtx.push_back(p[1]); out.new_vtx.push_back(p[2]);
            if (cnt) cnt->n_new++;
        }
    }

    // ---- a20: delaunay_triangulation, mesh_rec_geometry.cpp:174-295 -----------------------------------------
    // ids ascending; returns accepted faces as vertex-id triples (unsorted within a face)
    void delaunay_triangulation(const std::vector<int>& ids, double short_axis[3], std::vector<int>& tri_ids, long* n_skipped = nullptr) {
        tri_ids.clear();
        const int n = (int)ids.size();
        if (n < 3) return;
        double c[3] = {0, 0, 0};
        for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) c[k] += verts[ids[i]].pos[k];
        for (int k = 0; k < 3; k++) c[k] /= (double)n;  // colwise().mean()
        std::vector<double> X((size_t)n * 3);
        for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) X[i * 3 + k] = verts[ids[i]].pos[k] - c[k];
        double cov[9];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double s = 0;
                for (int i = 0; i < n; i++) s += X[i * 3 + a] * X[i * 3 + b];
                cov[a * 3 + b] = s / (double)n;
            }
        double ev[3], U[9];
        sym3_eigen_jacobi(cov, ev, U);  // stands in for SelfAdjointEigenSolver::compute (:199-200), eigenvalues ascending
        int ord[3] = {0, 1, 2};
        std::stable_sort(ord, ord + 3, [&](int a, int b) { return ev[a] < ev[b]; });
        double sh[3] = {U[0 * 3 + ord[0]], U[1 * 3 + ord[0]], U[2 * 3 + ord[0]]};
        double mid[3] = {U[0 * 3 + ord[1]], U[1 * 3 + ord[1]], U[2 * 3 + ord[1]]};
        if (X[0] * sh[0] + X[1] * sh[1] + X[2] * sh[2] < 0) for (int k = 0; k < 3; k++) sh[k] *= -1;       // :204-207
        if (X[3] * mid[0] + X[4] * mid[1] + X[5] * mid[2] < 0) for (int k = 0; k < 3; k++) mid[k] *= -1;   // :208-211
        double lg[3];
        cross3(sh, mid, lg);  // long = short x mid :212
        for (int k = 0; k < 3; k++) short_axis[k] = sh[k];
        std::vector<double> xy((size_t)n * 2);
        for (int i = 0; i < n; i++) {
            xy[2 * i + 0] = X[i * 3 + 0] * lg[0] + X[i * 3 + 1] * lg[1] + X[i * 3 + 2] * lg[2];
            xy[2 * i + 1] = X[i * 3 + 0] * mid[0] + X[i * 3 + 1] * mid[1] + X[i * 3 + 2] * mid[2];
        }
        Delaunay2D dt;
        std::vector<int> faces;
        dt.run(xy.data(), n, faces);
        if (n_skipped) *n_skipped += dt.n_skipped;
        // skinny-face filter: is_face_is_ok always uses 150 (:31-57)
        auto angle = [&](int a, int b, int cc) {  // compute_angle :24-29, at a
            const double abx = xy[2 * b] - xy[2 * a], aby = xy[2 * b + 1] - xy[2 * a + 1];
            const double acx = xy[2 * cc] - xy[2 * a], acy = xy[2 * cc + 1] - xy[2 * a + 1];
            return std::acos((abx * acx + aby * acy) / (std::sqrt(abx * abx + aby * aby) * std::sqrt(acx * acx + acy * acy))) * 57.3;
        };
        for (size_t f = 0; f + 2 < faces.size(); f += 3) {
            const int a = faces[f], b = faces[f + 1], cc = faces[f + 2];
            if (angle(a, b, cc) > 150) continue;
            if (angle(b, a, cc) > 150) continue;
            if (angle(cc, a, b) > 150) continue;
            tri_ids.push_back(ids[a]); tri_ids.push_back(ids[b]); tri_ids.push_back(ids[cc]);
        }
    }

    // ---- a23: correct_triangle_index, mesh_rec_geometry.cpp:399-433 -> m_index_flip ---------------------------
    int flip_of(const Tri& t, const double cam[3], const double short_axis_in[3]) const {
        const double* A = verts[t[0]].smooth; const double* B = verts[t[1]].smooth; const double* C = verts[t[2]].smooth;
        const double ab[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, ac[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
        const double tc[3] = {cam[0] - A[0], cam[1] - A[1], cam[2] - A[2]};
        double nrm[3];
        cross3(ab, ac, nrm);
        const double nn = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        if (nn != 0) { nrm[0] /= nn; nrm[1] /= nn; nrm[2] /= nn; }  // normalize()
        else { nrm[0] = 0; nrm[1] = 0; nrm[2] = 1; }
        double sa[3] = {short_axis_in[0], short_axis_in[1], short_axis_in[2]};
        if (dot3(sa, tc) < 0) { sa[0] *= -1; sa[1] *= -1; sa[2] *= -1; }
        return (dot3(sa, nrm) < 0) ? 0 : 1;
    }

    // ---- a25: incremental_mesh_reconstruction, ImMesh_mesh_reconstruction.cpp:92-267 --------------------------
    void mesh_scan(const float* pts_world_xyzi, int n_raw, const double sensor_pos[3], MeshScanOut& out) {
        out = MeshScanOut();
        const int step = std::max(1, (int)std::round((double)(n_raw / cfg.mesh_append_budget)));  // integer division first (:111, A.11)
        std::vector<int> recent;
        append(pts_world_xyzi, n_raw, step, recent, out);
        std::sort(recent.begin(), recent.end(), [&](int a, int b) {
            return K3{voxels[a].pos[0], voxels[a].pos[1], voxels[a].pos[2]} < K3{voxels[b].pos[0], voxels[b].pos[1], voxels[b].pos[2]};
        });
        const double accept = cfg.mesh_voxel * 1.25;  // g_kd_tree_accept_pt_dis, mesh_rec_geometry.cpp:343
        std::set<Tri> all_rem;
        std::map<Tri, int> all_add, all_upd, upd_orig;
        std::map<int, std::array<double, 3>> smoothed;
        // The per-voxel work splits into a part that reads only raw vertex positions (20-NN pull, smoothed means, 2-D Delaunay) and a part that
        // depends on the order of the voxels (smoothed positions seen so far, the live triangle set, which voxel's flip wins).  The first runs
        // voxel-parallel (the reference's TBB pool), the second strictly in ascending voxel order: same results for any thread count.
        struct VoxWork { int vi; std::vector<int> ids; std::vector<std::pair<int, std::array<double, 3>>> sm; std::vector<int> tri_ids; long c20 = 0, n_skip = 0; };
        std::vector<VoxWork> work;
        for (int vi : recent) {
            MeshVoxel& vox = voxels[vi];
            if (vox.meshing_times >= 1 || vox.new_added < 0) continue;  // :132
            vox.meshing_times++;
            vox.new_added = 0;
            if (vox.pts.size() < 3) continue;  // :147-151
            out.v_act++;
            if (cnt) { cnt->v_act++; cnt->n_v += (long)vox.pts.size(); }
            VoxWork w; w.vi = vi;
            work.push_back(std::move(w));
        }
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) if (threads > 1)
        for (long wi = 0; wi < (long)work.size(); wi++) {
            VoxWork& w = work[(size_t)wi];
            MeshVoxel& vox = voxels[w.vi];
            std::vector<NN> nn;
            // a19 retrieve_neighbor_pts_kdtree, mesh_rec_geometry.cpp:336-377
            std::set<long> rel;
            for (int id : vox.pts) {
                const float q[3] = {(float)verts[id].pos[0], (float)verts[id].pos[1], (float)verts[id].pos[2]};
                knn(q, 20, accept * 2, nn, &w.c20);
                double sv[3] = {0, 0, 0};
                int sc = 0;
                for (const NN& e : nn) {
                    const float d = std::sqrt(e.d2);
                    if (d < accept) rel.insert(e.id);
                    if (d < accept * 2) { sc++; for (int k = 0; k < 3; k++) sv[k] += verts[e.id].pos[k]; }
                }
                for (int k = 0; k < 3; k++) sv[k] /= (double)sc;
                w.sm.push_back({id, {sv[0], sv[1], sv[2]}});
            }
            w.ids.assign(rel.begin(), rel.end());
            delaunay_triangulation(w.ids, vox.short_axis, w.tri_ids, &w.n_skip);
        }
        for (VoxWork& w : work) {
            MeshVoxel& vox = voxels[w.vi];
            for (const auto& e : w.sm) {
                for (int k = 0; k < 3; k++) verts[e.first].smooth[k] = e.second[k];  // smooth_factor == 1.0
                smoothed[e.first] = e.second;
            }
            const std::vector<int>& ids = w.ids;
            const std::vector<int>& tri_ids = w.tri_ids;
            const std::set<long> rel(ids.begin(), ids.end());
            out.n_u_list.push_back((int)ids.size());
            if (cnt) { cnt->c20 += w.c20; cnt->n_u += (long)ids.size(); cnt->n_degenerate_skips += w.n_skip; }
            // a21 find_relative_triangulation_combination, triangle.hpp:223-246
            std::set<Tri> old;
            for (int id : ids) {
                auto it = adj.find(id);
                if (it == adj.end()) continue;
                for (const Tri& t : it->second)
                    if (rel.count(t[0]) && rel.count(t[1]) && rel.count(t[2])) old.insert(t);
            }
            // a22 triangle_compare, mesh_rec_geometry.cpp:137-172
            std::set<Tri> fresh;
            for (size_t f = 0; f + 2 < tri_ids.size(); f += 3) {
                Tri t = {tri_ids[f], tri_ids[f + 1], tri_ids[f + 2]};
                std::sort(t.begin(), t.end());
                fresh.insert(t);
            }
            if (cnt) cnt->t_v += (long)fresh.size();   // (the faces are a set: on degenerate input the triangulation may hold one twice)
            for (const Tri& t : old) if (!fresh.count(t)) all_rem.insert(t);
            for (const Tri& t : fresh) {
                const int fl = flip_of(t, sensor_pos, vox.short_axis);
                if (old.count(t)) {  // existing: flip rewritten in place (:203-205); reported only when the scan changes it
                    if (!upd_orig.count(t)) upd_orig[t] = tri_flip[t];
                    all_upd[t] = fl; tri_flip[t] = fl;
                }
                else all_add[t] = fl;                                     // later voxel (ascending key order) wins on duplicates
            }
        }
        // commit: all removes, then all adds (ImMesh_mesh_reconstruction.cpp:228-244)
        for (const Tri& t : all_rem) {
            for (int k = 0; k < 3; k++) { auto it = adj.find(t[k]); if (it != adj.end()) it->second.erase(t); }
            out.tri_rem.insert(out.tri_rem.end(), t.begin(), t.end());
        }
        for (const auto& kv : all_add) {
            const Tri& t = kv.first;
            tri_flip[t] = kv.second;
            for (int k = 0; k < 3; k++) adj[t[k]].insert(t);
            out.tri_add.insert(out.tri_add.end(), t.begin(), t.end());
            out.flip_add.push_back((uint8_t)kv.second);
        }
        for (const auto& kv : all_upd) {
            if (kv.second == upd_orig[kv.first]) continue;
            out.tri_upd.insert(out.tri_upd.end(), kv.first.begin(), kv.first.end());
            out.flip_upd.push_back((uint8_t)kv.second);
        }
        for (const auto& kv : smoothed) {
            out.smooth_ids.push_back(kv.first);
            for (int k = 0; k < 3; k++) out.smooth_xyz.push_back(kv.second[k]);
        }
        if (cnt) { cnt->t_add += (long)out.tri_add.size() / 3; cnt->t_rem += (long)out.tri_rem.size() / 3; }
    }

    // ---- save_to_ply_file (mesh_rec_geometry.cpp:71-131) + Global_map::smooth_pts (pointcloud_rgbd.cpp:932-958) ------------------------
    // vertices: smooth_factor == 0 -> raw; else pt*(1-f) + f * (sum of the 2nd..k-th nearest closer than accept) / valid (0/0 -> NaN as the reference).
    // faces: live triangles, winding from m_index_flip, ordered by sorted triplet.
    void export_mesh(double smooth_factor, int knn_k, std::vector<float>& vtx, std::vector<int>& faces) const {
        const double accept = cfg.mesh_voxel * 1.25;
        vtx.resize(verts.size() * 3);
        std::vector<NN> nn;
        for (size_t i = 0; i < verts.size(); i++) {
            const double* p = verts[i].pos;
            double out[3] = {p[0], p[1], p[2]};
            if (smooth_factor != 0) {
                const float q[3] = {(float)p[0], (float)p[1], (float)p[2]};
                knn(q, knn_k, accept * 2, nn);
                double s[3] = {0, 0, 0}, valid = 0.0;
                for (size_t k = 1; k < nn.size(); k++)
                    if ((double)std::sqrt(nn[k].d2) < accept) { for (int a = 0; a < 3; a++) s[a] += verts[nn[k].id].pos[a]; valid += 1.0; }
                for (int a = 0; a < 3; a++) out[a] = p[a] * (1.0 - smooth_factor) + s[a] * smooth_factor / valid;
            }
            for (int a = 0; a < 3; a++) vtx[i * 3 + a] = (float)out[a];
        }
        std::vector<int> live;
        live_triangles(live);
        faces.resize(live.size());
        for (size_t f = 0; f + 2 < live.size(); f += 3) {
            const Tri t = {live[f], live[f + 1], live[f + 2]};
            const bool keep = tri_flip.at(t) != 0;
            faces[f] = t[0]; faces[f + 1] = keep ? t[1] : t[2]; faces[f + 2] = keep ? t[2] : t[1];
        }
    }

    // ---- Global_map::smooth_pts (pointcloud_rgbd.cpp:932-958) for ONE vertex, as the renderer calls it (mesh_rec_display.cpp:86-90: smooth_factor =
    // g_ply_smooth_factor, knn = g_ply_smooth_k, maximum_smooth_dis = g_kd_tree_accept_pt_dis): the knn nearest vertices of the WHOLE map, the first one
    // (the vertex itself) skipped, those closer than maximum_smooth_dis (<= 0: 0.8 x the mesh voxel, :940-943) averaged; nobody close -> 0/0 = NaN.
    // The reference also stores the value in the point (set_smooth_pos); here the function is const -- the caller's mirror does that.
    void smooth_pts(int id, double smooth_factor, int knn_k, double maximum_smooth_dis, double out[3]) const {
        if (maximum_smooth_dis <= 0) maximum_smooth_dis = cfg.mesh_voxel * 0.8;
        const double* p = verts[id].pos;
        const float q[3] = {(float)p[0], (float)p[1], (float)p[2]};
        std::vector<NN> nn;
        knn(q, knn_k, cfg.mesh_voxel * 1.25 * 2, nn);   // (what lies beyond the search radius 2.5 x voxel is beyond every maximum_smooth_dis the entry accepts)
        double s[3] = {0, 0, 0}, valid = 0.0;
        for (size_t k = 1; k < nn.size(); k++)
            if ((double)std::sqrt(nn[k].d2) < maximum_smooth_dis) { for (int a = 0; a < 3; a++) s[a] += verts[nn[k].id].pos[a]; valid += 1.0; }
        for (int a = 0; a < 3; a++) out[a] = p[a] * (1.0 - smooth_factor) + s[a] * smooth_factor / valid;
    }
    // RGB_pts::get_pos(1) as unparse_triangle_set_to_vector reads it (mesh_rec_display.cpp:86-98): a vertex the mesher has smoothed serves its
    // m_pos_aft_smooth, any other is smoothed on the spot
    void display_vertex(int id, double smooth_factor, int knn_k, double maximum_smooth_dis, float out[3]) const {
        const MeshVertex& v = verts[id];
        const bool smoothed = v.smooth[0] != v.pos[0] || v.smooth[1] != v.pos[1] || v.smooth[2] != v.pos[2];
        double o[3];
        if (smoothed) { o[0] = v.smooth[0]; o[1] = v.smooth[1]; o[2] = v.smooth[2]; }
        else smooth_pts(id, smooth_factor, knn_k, maximum_smooth_dis, o);
        for (int a = 0; a < 3; a++) out[a] = (float)o[a];
    }

    size_t live_triangle_count() const {
        size_t s = 0;
        for (const auto& kv : adj) s += kv.second.size();
        return s / 3;
    }
    void live_triangles(std::vector<int>& out) const {
        std::set<Tri> all;
        for (const auto& kv : adj) for (const Tri& t : kv.second) all.insert(t);
        out.clear();
        for (const Tri& t : all) out.insert(out.end(), t.begin(), t.end());
    }
};

}  // namespace orc
