// Thin C wrapper that compiles the reference's own in-tree ikd-Tree
// (/root/reference/include/ikd-Tree/ikd_Tree.cpp, included from where it lies)
// so the oracle's exact-kNN restatement can be validated against it.
// TEST INFRASTRUCTURE ONLY -- never linked into the product.
#include "ikd_Tree.cpp"
#include <cstdint>
typedef KD_TREE<ikdTree_PointType> Tree;
extern "C" {
void* ref_ikd_create() { return new Tree(0.5f, 0.6f, 0.2f); }
void ref_ikd_destroy(void* t) { delete (Tree*)t; }
// mirrors Global_map::append_points_to_global_map's m_kdtree.Add_Point(pt,false) (pointcloud_rgbd.cpp:540)
void ref_ikd_add(void* t, const float* xyz, long idx) {
    ikdTree_PointType p(xyz[0], xyz[1], xyz[2]); p.m_pt_idx = idx;
    ((Tree*)t)->Add_Point(p, false);
}
int ref_ikd_has_root(void* t) { return ((Tree*)t)->Root_Node != nullptr; }
// Nearest_Search(point,k,...) (ikd_Tree.cpp:440): ascending distance; returns count found
int ref_ikd_knn(void* t, const float* xyz, int k, long* idx_out, float* d2_out) {
    ikdTree_PointType p(xyz[0], xyz[1], xyz[2]);
    Tree::PointVector pts; std::vector<float> d;
    ((Tree*)t)->Nearest_Search(p, k, pts, d);
    for (size_t i = 0; i < pts.size(); i++) { idx_out[i] = pts[i].m_pt_idx; d2_out[i] = d[i]; }
    return (int)pts.size();
}
int ref_ikd_size(void* t) { return ((Tree*)t)->size(); }
// the legacy registration map: set_downsample_param + Build (voxel_mapping.cpp:1906-1914), Add_Points(.., true)
// (ImMesh_mesh_reconstruction.cpp:439), flatten for a dump of the surviving points
int ref_ikd_delete_boxes(void* t, const float* boxes, int nb) {   // Delete_Point_Boxes, as laser_map_fov_segment calls it
    std::vector<BoxPointType> v;
    for (int b = 0; b < nb; b++) { BoxPointType q; for (int a = 0; a < 3; a++) { q.vertex_min[a] = boxes[b * 6 + a]; q.vertex_max[a] = boxes[b * 6 + 3 + a]; } v.push_back(q); }
    return ((Tree*)t)->Delete_Point_Boxes(v);
}
int ref_ikd_validnum(void* t) { return ((Tree*)t)->validnum(); }
void ref_ikd_set_downsample(void* t, float ds) { ((Tree*)t)->set_downsample_param(ds); }
void ref_ikd_build(void* t, const float* xyz, int n) {
    Tree::PointVector v;
    for (int i = 0; i < n; i++) { ikdTree_PointType p(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]); p.m_pt_idx = i; v.push_back(p); }
    ((Tree*)t)->Build(v);
}
int ref_ikd_add_points_ds(void* t, const float* xyz, int n) {
    Tree::PointVector v;
    for (int i = 0; i < n; i++) { ikdTree_PointType p(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]); p.m_pt_idx = -1; v.push_back(p); }
    return ((Tree*)t)->Add_Points(v, true);
}
int ref_ikd_flatten(void* t, float* xyz, int cap) {
    Tree::PointVector v;
    ((Tree*)t)->flatten(((Tree*)t)->Root_Node, v, NOT_RECORD);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) { xyz[i * 3] = v[i].x; xyz[i * 3 + 1] = v[i].y; xyz[i * 3 + 2] = v[i].z; }
    return (int)v.size();
}
int ref_ikd_knn_xyz(void* t, const float* xyz, int k, float* nn_xyz, float* d2_out) {
    ikdTree_PointType p(xyz[0], xyz[1], xyz[2]);
    Tree::PointVector pts; std::vector<float> d;
    ((Tree*)t)->Nearest_Search(p, k, pts, d);
    for (size_t i = 0; i < pts.size(); i++) { nn_xyz[i * 3] = pts[i].x; nn_xyz[i * 3 + 1] = pts[i].y; nn_xyz[i * 3 + 2] = pts[i].z; d2_out[i] = d[i]; }
    return (int)pts.size();
}
}
