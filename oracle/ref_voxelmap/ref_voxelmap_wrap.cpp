// ORACLE / TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE'S OWN voxel-map code, compiled from where it lies under /root/reference
// behind an Eigen-shaped stub (stubs/mini_eigen.hpp) -- the recipe is oracle/Makefile (target ref -> _ref/libref_voxelmap.so):
//   src/voxel_loc.hpp, src/voxel_loc.cpp            whole files: OctoTree::init_plane / init_octo_tree / cut_octo_tree / UpdateOctoTree (rows a4, a5)
//   src/voxel_mapping.cpp:49                        var_contrast                                                     (a15)
//   src/voxel_mapping.cpp:110-354                   buildVoxelMap, BuildResidualListOMP, build_single_residual, updateVoxelMap   (a6, a10, a11, a16)
//   src/voxel_mapping.cpp:1221-1241                 calcBodyVar                                                      (a7)
// The excerpts of voxel_mapping.cpp are cut out by line range into _ref/vm_src/ at BUILD time (sed; the directory is removed after the compile) and
// #included below -- nothing of the reference is copied into the repository.  What is pinned: the reference's logic (key quantisation, the octree
// state machine, which points a fit sees, the near-voxel retry with its unit mismatch, the float gates, the plane-covariance formula as written).
// What is NOT: Eigen's arithmetic (the stub's products are plain sums; EigenSolver is the oracle's Jacobi).
#include "voxel_loc.hpp"      // the reference's (via -I /root/reference/src)
#include <algorithm>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include "../../include/immesh_c_api.h"
using std::unordered_map;
typedef unsigned int uint;
#ifndef MP_PROC_NUM
#define MP_PROC_NUM 4          /* CMakeLists.txt:21-24 */
#endif
// declarations of src/voxel_mapping.hpp:80-105,129
void build_single_residual( const Point_with_var &pv, const OctoTree *current_octo, const int current_layer, const int max_layer, const double sigma_num, bool &is_sucess, double &prob, ptpl &single_ptpl );
#include "vm_var_contrast.inc"     // voxel_mapping.cpp:49
#include "vm_map_and_matcher.inc"  // voxel_mapping.cpp:110-354
#include "vm_calc_body_var.inc"    // voxel_mapping.cpp:1221-1241

namespace {
struct RefVoxelMap {
    float voxel_size; int max_layer; std::vector<int> layer_init; int max_points_size; float planer_threshold;
    std::unordered_map<VOXEL_LOC, OctoTree*> map;
};
void fill(std::vector<Point_with_var>& v, const double* p_body, const double* p_world, const double* var9, int n, bool point_is_world) {
    v.resize(n);
    for (int i = 0; i < n; i++) {
        Point_with_var& pv = v[i];
        for (int k = 0; k < 3; k++) { pv.m_point[k] = point_is_world ? p_world[i * 3 + k] : (p_body ? p_body[i * 3 + k] : 0.0); pv.m_point_world[k] = p_world ? p_world[i * 3 + k] : 0.0; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) pv.m_var(r, c) = var9[i * 9 + r * 3 + c];
    }
}
void dump_node(const VOXEL_LOC& k, const OctoTree* n, int path, int depth, immesh_plane_rec* out, int64_t cap, int64_t& cnt) {
    if (n->m_init_octo_) {
        if (cnt < cap && out) {
            immesh_plane_rec& r = out[cnt];
            std::memset(&r, 0, sizeof(r));
            const Plane& p = *n->m_plane_ptr_;
            r.key[0] = k.x; r.key[1] = k.y; r.key[2] = k.z;
            r.layer = n->m_layer_; r.path = path; r.is_plane = p.m_is_plane ? 1 : 0; r.n_points = (int)n->m_temp_points_.size();
            r.update_enable = n->m_update_enable_ ? 1 : 0; r.new_points = n->m_new_points_;
            r.radius = p.m_radius; r.min_eig = p.m_min_eigen_value; r.d = p.m_d;
            for (int i = 0; i < 3; i++) { r.center[i] = p.m_center(i); r.normal[i] = p.m_normal(i); }
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) r.plane_var[i * 6 + j] = p.m_plane_var(i, j);
        }
        cnt++;
    }
    for (int l = 0; l < 8; l++)
        if (n->m_leaves_[l]) dump_node(k, n->m_leaves_[l], path | (l << (3 * depth)), depth + 1, out, cap, cnt);
}
}  // namespace

extern "C" {
void* rv_create(double voxel_size, int max_layer, const int* layer_init5, int max_points_size, double planer_threshold) {
    RefVoxelMap* m = new RefVoxelMap();
    m->voxel_size = (float)voxel_size; m->max_layer = max_layer; m->layer_init.assign(layer_init5, layer_init5 + 5); m->max_points_size = max_points_size;
    m->planer_threshold = (float)planer_threshold;
    return m;
}
void rv_destroy(void* p) { delete (RefVoxelMap*)p; }   // (the reference leaks its octrees: so does this)
// buildVoxelMap (voxel_mapping.cpp:110): pv.m_point = WORLD point, m_var = world covariance (voxel_map_init :1264-1277)
void rv_build(void* p, const double* pts_world, const double* var9, int n) {
    RefVoxelMap* m = (RefVoxelMap*)p;
    std::vector<Point_with_var> v;
    fill(v, nullptr, pts_world, var9, n, true);
    buildVoxelMap(v, m->voxel_size, m->max_layer, m->layer_init, m->max_points_size, m->planer_threshold, m->map);
}
// map_incremental_grow's tail (ImMesh_mesh_reconstruction.cpp:405-407): std::sort(pv_list, var_contrast) -- here stable, i.e. ties keep scan order, the
// deterministic tie-break of the checker -- then updateVoxelMap (voxel_mapping.cpp:320)
void rv_update(void* p, const double* pts_world, const double* var9, int n, int sort_by_var) {
    RefVoxelMap* m = (RefVoxelMap*)p;
    std::vector<Point_with_var> v;
    fill(v, nullptr, pts_world, var9, n, true);
    if (sort_by_var) std::stable_sort(v.begin(), v.end(), [](const Point_with_var& a, const Point_with_var& b) { return var_contrast(const_cast<Point_with_var&>(a), const_cast<Point_with_var&>(b)); });
    updateVoxelMap(v, m->voxel_size, m->max_layer, m->layer_init, m->max_points_size, m->planer_threshold, m->map);
}
int64_t rv_dump(void* p, immesh_plane_rec* out, int64_t cap) {
    RefVoxelMap* m = (RefVoxelMap*)p;
    int64_t cnt = 0;
    for (const auto& kv : m->map) dump_node(kv.first, kv.second, 0, 0, out, cap, cnt);
    return cnt;
}
int64_t rv_root_voxels(void* p) { return (int64_t)((RefVoxelMap*)p)->map.size(); }
// BuildResidualListOMP (voxel_mapping.cpp:153) on a pv_list {m_point = body point, m_point_world, m_var}: returns the number of matches; per match the
// index of its point (recovered from ptpl::point, which the matcher copies from pv.m_point: the caller's body points are replaced by (index, 0, 0)),
// normal, centre, d, layer, plane_var
int rv_residual_list(void* p, const double* pts_world, const double* var9, int n, double voxel_size, double sigma_num, int32_t* match_idx, double* normals, double* centers,
                     double* d, int32_t* layer, double* plane_var36) {
    RefVoxelMap* m = (RefVoxelMap*)p;
    std::vector<Point_with_var> v;
    fill(v, nullptr, pts_world, var9, n, false);
    for (int i = 0; i < n; i++) { v[i].m_point[0] = (double)i; v[i].m_point[1] = 0; v[i].m_point[2] = 0; }
    std::vector<ptpl> list;
    std::vector<Eigen::Vector3d> non_match;
    BuildResidualListOMP(m->map, voxel_size, sigma_num, m->max_layer, v, list, non_match);
    for (size_t k = 0; k < list.size(); k++) {
        const ptpl& q = list[k];
        if (match_idx) match_idx[k] = (int32_t)q.point[0];
        for (int a = 0; a < 3; a++) { if (normals) normals[k * 3 + a] = q.normal(a); if (centers) centers[k * 3 + a] = q.center(a); }
        if (d) d[k] = q.d;
        if (layer) layer[k] = q.layer;
        if (plane_var36) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) plane_var36[k * 36 + i * 6 + j] = q.plane_var(i, j);
    }
    return (int)list.size();
}
// calcBodyVar (voxel_mapping.cpp:1221): pb is modified in place (z == 0 -> 1e-4), as in the reference
void rv_calc_body_var(double* pb3, float range_inc, float degree_inc, double* var9) {
    Eigen::Vector3d pb(pb3[0], pb3[1], pb3[2]);
    Eigen::Matrix3d var;
    calcBodyVar(pb, range_inc, degree_inc, var);
    for (int k = 0; k < 3; k++) pb3[k] = pb[k];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) var9[r * 3 + c] = var(r, c);
}
// VOXEL_LOC of a point as buildVoxelMap / updateVoxelMap derive it (float voxel size) -- through a one-point map
void rv_key(const double* p3, double voxel_size, int64_t* key3) {
    std::unordered_map<VOXEL_LOC, OctoTree*> mp;
    std::vector<Point_with_var> v(1);
    for (int k = 0; k < 3; k++) v[0].m_point[k] = p3[k];
    std::vector<int> li = {5, 5, 5, 5, 5};
    updateVoxelMap(v, (float)voxel_size, 0, li, 100, 0.01f, mp);
    const VOXEL_LOC& k = mp.begin()->first;
    key3[0] = k.x; key3[1] = k.y; key3[2] = k.z;
}
}  // extern "C"
