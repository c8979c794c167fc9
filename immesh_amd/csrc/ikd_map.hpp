// Legacy registration path (`voxel_map_en = false`, dead in every shipped config): the device point map that stands in
// for the ikd-Tree of map points (KD_TREE::Build / Add_Points(.., true) / Nearest_Search, include/ikd-Tree/ikd_Tree.cpp:283-310, 440-476,
// 493-602).  What the tree computes is kept, not the tree: a hash grid with downsample_size cells -- Add_Points' box-downsample leaves at
// most one point per cell (the one nearest to the cell centre, a new point winning ties), Build may leave a few -- which doubles as the
// spatial index of the exact 5-NN search (expanding cubes of cells until the 5th distance is covered).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IKD_CELL_PTS 8           /* points per cell (Build keeps every point of the first down-sampled scan) */
#define IKD_KNN 5                /* NUM_MATCH_POINTS, include/common_lib.h:42 */

struct IkdMapDev {
    unsigned long long* keys;    // packed 3 x 21-bit cell index, open addressing
    int32_t* count;              // points in the cell
    float4* pts;                 // [slot][IKD_CELL_PTS]: xyz + insertion id (as float bits)
    unsigned long long* best;    // per-batch: min over the batch's points of (dist-to-centre bits << 32 | ~batch index)
    int32_t* stamp;              // batch sequence number that last touched the cell
    int32_t* touched;            // slots touched by the current batch
    int32_t* counters;           // [0] points in the map, [1] touched cells, [2] overflow flag, [3] next insertion id
    uint64_t mask;
    int32_t cap_cells;
    float ds;                    // downsample_size
    int32_t seq;                 // batch sequence number
};

struct IkdHost {
    IkdMapDev m;
    bool ready = false;
    float* d_near = nullptr;      // [n][5][3] neighbours of the last search
    int32_t* d_near_n = nullptr;  // [n]
    int8_t* d_sel = nullptr;      // [n] m_point_selected_surf
    float* d_norm = nullptr;      // [n][4] normal + pd2 (m_normvec)
    double* d_part = nullptr;     // block partials
    double* d_out = nullptr;      // 48 sums
    float* d_q = nullptr;         // query / staging
    int64_t cap_pts = 0;
    bool localmap_initialized = false;   // m_localmap_Initialized / m_LocalMap_Points (laser_map_fov_segment)
    float lm_min[3] = {0, 0, 0}, lm_max[3] = {0, 0, 0};
};
struct IkdBoxes { float b[3][6]; int n; };   // at most one slab per axis

void launch_ikd_build(hipStream_t s, const IkdMapDev& m, const float* xyz, int n);
void launch_ikd_add(hipStream_t s, const IkdMapDev& m, const float* xyz, int n);
void launch_ikd_delete_boxes(hipStream_t s, const IkdMapDev& m, const IkdBoxes& boxes, int32_t* n_deleted);
void launch_ikd_dump(hipStream_t s, const IkdMapDev& m, float* xyz, long long cap, unsigned long long* count);
// mode 0: k-NN only (nn_xyz / d2 out); mode 1: search + plane fit + gates; mode 2: plane fit + gates on the stored neighbours
struct IkdMatchParams { double R[9], t[3], extR[9], extT[3]; double r_inv; };
void launch_ikd_match(hipStream_t s, const IkdMapDev& m, const IkdMatchParams& mp, const float* body_or_query, int n, int mode, float* near_xyz, int32_t* near_n,
                      int8_t* sel, float* normvec, float* d2_out);
void launch_ikd_reduce(hipStream_t s, const IkdMatchParams& mp, const float* body, int n, const int8_t* sel, const float* normvec, double* partials, double* out48);
