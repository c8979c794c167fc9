// Test driver of the drop-in (tests/test_gpu_dropin.py): plays the role of service_LiDAR_update (src/voxel_mapping.cpp:1959-1973) around the
// replaced bodies.  Input file (little endian): int32 n_scans; per scan: int32 n_raw, int32 n_ds, double prior[348], float raw[n_raw*4]
// (x y z intensity, body frame), float down[n_ds*3].  Scan 0 initialises the map (voxel_map_init) with state = prior.  Output file: per scan
// double state[348], int32 effct_feat_num, int32 n_vertices, int32 n_live_triangles, uint64 order-independent hash of the live (triplet, flip) set.
#include "immesh_ref_shapes.hpp"
#include <cstdio>
#include <cstring>
Global_map g_map_rgb_pts_mesh;
Triangle_manager g_triangles_manager;
static void load_state(const double* o, StatesGroup& s) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) s.rot_end(r, c) = o[r * 3 + c];
    for (int i = 0; i < 3; i++) { s.pos_end(i) = o[9 + i]; s.vel_end(i) = o[12 + i]; s.bias_g(i) = o[15 + i]; s.bias_a(i) = o[18 + i]; s.gravity(i) = o[21 + i]; }
    for (int r = 0; r < 18; r++) for (int c = 0; c < 18; c++) s.cov(r, c) = o[24 + r * 18 + c];
}
int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: shim_main <in> <out>\n"); return 2; }
    FILE* fi = std::fopen(argv[1], "rb"); FILE* fo = std::fopen(argv[2], "wb");
    if (!fi || !fo) return 2;
    Voxel_mapping vm;
    vm.m_extT(0) = 0.04165; vm.m_extT(1) = 0.02326; vm.m_extT(2) = -0.0284;   // config/avia.yaml:40
    vm.immesh_shim_init();
    int32_t n_scans = 0;
    if (std::fread(&n_scans, 4, 1, fi) != 1) return 2;
    for (int k = 0; k < n_scans; k++) {
        int32_t n_raw, n_ds; double prior[348];
        if (std::fread(&n_raw, 4, 1, fi) != 1 || std::fread(&n_ds, 4, 1, fi) != 1 || std::fread(prior, 8, 348, fi) != 348) return 2;
        std::vector<float> raw((size_t)n_raw * 4), down((size_t)n_ds * 3);
        if (std::fread(raw.data(), 4, raw.size(), fi) != raw.size() || std::fread(down.data(), 4, down.size(), fi) != down.size()) return 2;
        vm.m_feats_undistort->resize(n_raw); vm.m_feats_down_body->resize(n_ds);
        for (int i = 0; i < n_raw; i++) { PointType& p = vm.m_feats_undistort->points[i]; p.x = raw[4 * i]; p.y = raw[4 * i + 1]; p.z = raw[4 * i + 2]; p.intensity = raw[4 * i + 3]; }
        for (int i = 0; i < n_ds; i++) { PointType& p = vm.m_feats_down_body->points[i]; p.x = down[3 * i]; p.y = down[3 * i + 1]; p.z = down[3 * i + 2]; }
        StatesGroup propagat;
        load_state(prior, propagat);
        vm.state = propagat;
        if (k == 0) { if (!vm.voxel_map_init()) return 3; }                       // first frame: the map is initialised and the frame ends (src/voxel_mapping.cpp:1896-1904)
        else { vm.lio_state_estimation(propagat); vm.map_incremental_grow(); }    // :1956-1973
        double st[348];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) st[r * 3 + c] = vm.state.rot_end(r, c);
        for (int i = 0; i < 3; i++) { st[9 + i] = vm.state.pos_end(i); st[12 + i] = vm.state.vel_end(i); st[15 + i] = vm.state.bias_g(i); st[18 + i] = vm.state.bias_a(i); st[21 + i] = vm.state.gravity(i); }
        for (int r = 0; r < 18; r++) for (int c = 0; c < 18; c++) st[24 + r * 18 + c] = vm.state.cov(r, c);
        unsigned long long h = 0;
        for (auto& t : g_triangles_manager.m_live) {
            unsigned long long x = ((unsigned long long)(unsigned)t->m_tri_pts_id[0] * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)(unsigned)t->m_tri_pts_id[1] << 21) ^ ((unsigned long long)(unsigned)t->m_tri_pts_id[2] << 42) ^ (unsigned long long)(t->m_index_flip & 1);
            x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
            h += x;
        }
        const int32_t eff = vm.m_effct_feat_num, nv = (int32_t)g_map_rgb_pts_mesh.m_rgb_pts_vec.size(), nl = (int32_t)g_triangles_manager.m_live.size();
        std::fwrite(st, 8, 348, fo); std::fwrite(&eff, 4, 1, fo); std::fwrite(&nv, 4, 1, fo); std::fwrite(&nl, 4, 1, fo); std::fwrite(&h, 8, 1, fo);
    }
    std::fclose(fi); std::fclose(fo);
    save_to_ply_file("/tmp/immesh_dropin_test.ply", 0.0, 20);
    return 0;
}
