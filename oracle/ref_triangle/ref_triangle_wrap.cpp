// ORACLE -- TEST INFRASTRUCTURE ONLY.  C wrapper around the reference's own Triangle_manager, compiled from /root/reference (oracle/Makefile,
// target ref -> oracle/_ref/libref_triangle.so).  It pins rows a21 / a22 / a24 of SURVEY 8(a) to reference code: the per-scan diff lists
// (remove, then add) produced by the oracle and by the HIP path are applied to the REAL Triangle_manager exactly as
// incremental_mesh_reconstruction does (src/ImMesh_mesh_reconstruction.cpp:228-244) and its live set / vertex adjacency are read back.
#include "triangle.hpp"
#include <cstdint>

struct RtCtx {
    Global_map map;
    Triangle_manager mgr;
};

extern "C" {

void* rt_create(double region_size) {
    RtCtx* c = new RtCtx();
    c->mgr.m_pointcloud_map = &c->map;
    c->mgr.m_region_size = region_size;
    return c;
}
void rt_destroy(void* p) { delete (RtCtx*)p; }

// Global_map::m_rgb_pts_vec grows by the scan's new vertices (ids are positions in that vector)
void rt_append_vertices(void* p, const float* xyz, int64_t n) {
    RtCtx* c = (RtCtx*)p;
    for (int64_t i = 0; i < n; i++) {
        auto pt = std::make_shared<RGB_pts>();
        pt->m_pos[0] = xyz[i * 3 + 0]; pt->m_pos[1] = xyz[i * 3 + 1]; pt->m_pos[2] = xyz[i * 3 + 2];
        c->map.m_rgb_pts_vec.push_back(pt);
    }
}

// the "Voxel-wise mesh push" of one frame: all removals, then all insertions (insert_triangle(.., build_triangle_map = 1)) + m_index_flip.
// Returns the number of removal entries that named a triangle the manager does not know (must be 0).
int64_t rt_commit(void* p, const int32_t* tri_rem, int64_t n_rem, const int32_t* tri_add, const uint8_t* flip_add, int64_t n_add, int32_t frame_idx) {
    RtCtx* c = (RtCtx*)p;
    int64_t unknown = 0;
    Triangle_set rem;
    for (int64_t i = 0; i < n_rem; i++) {
        Triangle_ptr t = c->mgr.find_triangle(tri_rem[i * 3 + 0], tri_rem[i * 3 + 1], tri_rem[i * 3 + 2]);
        if (t == nullptr) unknown++; else rem.insert(t);
    }
    c->mgr.remove_triangle_list(rem, frame_idx);
    for (int64_t i = 0; i < n_add; i++) {
        Triangle_ptr t = c->mgr.insert_triangle(tri_add[i * 3 + 0], tri_add[i * 3 + 1], tri_add[i * 3 + 2], 1, frame_idx);
        t->m_index_flip = flip_add ? flip_add[i] : 0;
    }
    return unknown;
}
void rt_set_flips(void* p, const int32_t* tri, const uint8_t* flip, int64_t n) {   // correct_triangle_index on existing triangles
    RtCtx* c = (RtCtx*)p;
    for (int64_t i = 0; i < n; i++) {
        Triangle_ptr t = c->mgr.find_triangle(tri[i * 3 + 0], tri[i * 3 + 1], tri[i * 3 + 2]);
        if (t != nullptr) t->m_index_flip = flip[i];
    }
}

// live set = union of the per-region sets (what the renderer / save_to_ply_file walk); out: triplets + flip, cap entries
int64_t rt_live(void* p, int32_t* out_tri, uint8_t* out_flip, int64_t cap) {
    RtCtx* c = (RtCtx*)p;
    std::vector<Triangle_set> lists;
    c->mgr.get_all_triangle_list(lists, nullptr, 0);
    int64_t n = 0;
    for (auto& s : lists)
        for (auto& t : s) {
            if (n < cap && out_tri) { out_tri[n * 3 + 0] = t->m_tri_pts_id[0]; out_tri[n * 3 + 1] = t->m_tri_pts_id[1]; out_tri[n * 3 + 2] = t->m_tri_pts_id[2]; if (out_flip) out_flip[n] = (uint8_t)t->m_index_flip; }
            n++;
        }
    return n;
}
int64_t rt_live_size(void* p) { return ((RtCtx*)p)->mgr.get_triangle_list_size(); }

// find_relative_triangulation_combination (triangle.hpp:223-246): live triangles with all three vertices in the set
int64_t rt_find_relative(void* p, const int32_t* ids, int64_t n, int32_t* out_tri, int64_t cap) {
    RtCtx* c = (RtCtx*)p;
    std::set<int> s(ids, ids + n);
    Triangle_set r = c->mgr.find_relative_triangulation_combination(s);
    int64_t k = 0;
    for (auto& t : r) {
        if (k < cap) { out_tri[k * 3 + 0] = t->m_tri_pts_id[0]; out_tri[k * 3 + 1] = t->m_tri_pts_id[1]; out_tri[k * 3 + 2] = t->m_tri_pts_id[2]; }
        k++;
    }
    return k;
}

}  // extern "C"
