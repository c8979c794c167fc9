// ORACLE / TEST INFRASTRUCTURE ONLY.  C entry points around the reference's own mesh geometry helpers, cut out of src/meshing/mesh_rec_geometry.cpp by
// line range at BUILD time (oracle/Makefile, target ref -> _ref/libref_meshgeom.so; the excerpts live in _ref/mg_src/ only for the duration of the
// compile) and compiled behind the Eigen / CGAL shaped stubs of stub_pointcloud_rgbd.hpp, next to the reference's real triangle.hpp / tools_kd_hash.hpp:
//   :24-29    compute_angle          :31-57   is_face_is_ok (always 150 degrees, `* 57.3`)            (a20)
//   :137-172  triangle_compare                                                                         (a22)
//   :174-295  delaunay_triangulation: centre, covariance, axis choice and sign flips, 2-D projection, angle filter, triplet emission
//             -- the CGAL triangulation itself is the oracle's Bowyer-Watson behind a CGAL-shaped class                                  (a20)
//   :399-433  correct_triangle_index                                                                   (a23)
#include "triangle.hpp"      // the reference's (symlinked next to the stub of pointcloud_rgbd.hpp)
#include <cstdint>
#include <cstring>
using std::cout; using std::endl;
Global_map g_map_rgb_pts_mesh;          // ImMesh_node.cpp:108
bool is_face_is_ok( Common_tools::Delaunay2::Face &face, double maximum_angle );
#include "mg_angle.inc"                 // mesh_rec_geometry.cpp:24-57
#include "mg_triangle_compare.inc"      // :137-172
#include "mg_delaunay.inc"              // :174-295
#include "mg_correct_index.inc"         // :399-433

extern "C" {
double rg_compute_angle(const double* pa, const double* pb, const double* pc) {
    vec_2 a(pa[0], pa[1]), b(pb[0], pb[1]), c(pc[0], pc[1]);
    return compute_angle(a, b, c);
}
// delaunay_triangulation of n vertices (positions f64, ids = m_pt_index).  axes_io: long, mid, short (short == 0 on entry: computed, as for a voxel that
// is meshed for the first time).  Returns the number of ints written to tris_out (3 per accepted face, vertex ids in the emission order).
int64_t rg_delaunay(const double* pos, const int64_t* ids, int n, double* axes_io, int64_t* tris_out, int64_t cap) {
    std::vector<RGB_pt_ptr> v(n);
    for (int i = 0; i < n; i++) { v[i] = std::make_shared<RGB_pts>(); for (int k = 0; k < 3; k++) v[i]->m_pos[k] = pos[i * 3 + k]; v[i]->m_pt_index = (int)ids[i]; }
    vec_3 lg(axes_io[0], axes_io[1], axes_io[2]), mid(axes_io[3], axes_io[4], axes_io[5]), sh(axes_io[6], axes_io[7], axes_io[8]);
    std::set<long> hull, inner;
    std::vector<long> t = delaunay_triangulation(v, lg, mid, sh, hull, inner);
    for (int k = 0; k < 3; k++) { axes_io[k] = lg(k); axes_io[3 + k] = mid(k); axes_io[6 + k] = sh(k); }
    for (size_t i = 0; i < t.size() && (int64_t)i < cap; i++) tris_out[i] = t[i];
    return (int64_t)t.size();
}
// correct_triangle_index on a triangle whose three vertices have the given SMOOTHED positions: returns m_index_flip, writes the normal
int rg_flip(const double* a, const double* b, const double* c, const double* cam, const double* short_axis, double* normal_out) {
    g_map_rgb_pts_mesh.m_rgb_pts_vec.clear();
    const double* src[3] = {a, b, c};
    for (int i = 0; i < 3; i++) { auto p = std::make_shared<RGB_pts>(); for (int k = 0; k < 3; k++) p->m_pos_aft_smooth[k] = src[i][k]; g_map_rgb_pts_mesh.m_rgb_pts_vec.push_back(p); }
    Triangle_ptr t = std::make_shared<Triangle>(0, 1, 2);
    correct_triangle_index(t, vec_3(cam[0], cam[1], cam[2]), vec_3(short_axis[0], short_axis[1], short_axis[2]));
    if (normal_out) for (int k = 0; k < 3; k++) normal_out[k] = t->m_normal(k);
    return t->m_index_flip;
}
// triangle_compare: old = the triangles find_relative_triangulation_combination returned, fresh = delaunay_triangulation's output (id triples, any order
// inside a triple).  Outputs sorted-triplet lists: to remove, to add, existing.  Returns counts through n_out[3].
void rg_triangle_compare(const int32_t* old_tris, int n_old, const int64_t* fresh, int n_fresh, int32_t* rem, int32_t* add, int32_t* exist, int32_t* n_out) {
    Triangle_set old_set, res_rem, res_add, ex;
    for (int i = 0; i < n_old; i++) old_set.insert(std::make_shared<Triangle>(old_tris[i * 3], old_tris[i * 3 + 1], old_tris[i * 3 + 2]));
    std::vector<long> f(fresh, fresh + (size_t)n_fresh * 3);
    triangle_compare(old_set, f, res_rem, res_add, &ex);
    auto dump = [](const Triangle_set& s, int32_t* out) { std::vector<std::array<int, 3>> v; for (auto& t : s) v.push_back({t->m_tri_pts_id[0], t->m_tri_pts_id[1], t->m_tri_pts_id[2]}); std::sort(v.begin(), v.end());
                                                          v.erase(std::unique(v.begin(), v.end()), v.end()); int k = 0; for (auto& t : v) { out[k * 3] = t[0]; out[k * 3 + 1] = t[1]; out[k * 3 + 2] = t[2]; k++; } return k; };
    n_out[0] = dump(res_rem, rem); n_out[1] = dump(res_add, add); n_out[2] = dump(ex, exist);
}
}  // extern "C"
