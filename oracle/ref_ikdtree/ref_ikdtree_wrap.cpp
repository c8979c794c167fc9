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
}
