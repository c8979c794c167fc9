/* immesh_c_api.h -- C ABI of the MI355X-native ImMesh per-scan hot path (libimmesh_hip.so).
 *
 * The reference (hku-mars/ImMesh) has no plugin/FFI boundary: its hot path is reached through C++ members of
 * `class Voxel_mapping` and a few free functions.  Each entry point below replaces one of those call sites; the
 * file:line cited is the reference symbol a drop-in shim forwards from (paths relative to the reference tree).
 * INTEGRATION.md shows the shim.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Conventions
 *   - all matrices row-major; points are float xyz (3 floats) or xyzI (4 floats) exactly as the reference's
 *     pcl::PointXYZINormal / pcl::PointXYZI clouds hold them.
 *   - `state` = 348 doubles: R[9] t[3] vel[3] bias_g[3] bias_a[3] gravity[3] cov[18*18]   (StatesGroup,
 *     include/common_lib.h:199-288; cov block order rot,pos,vel,bg,ba,g).
 *   - input point pointers may be HOST or DEVICE (HIP) memory; the library detects which (hipPointerGetAttributes)
 *     and uses device buffers in place.  Small in/out arrays (state, HTH, ...) are host memory.
 *   - return 0 on success, negative IMMESH_E_* otherwise; immesh_last_error() gives text.  There is NO CPU
 *     fallback: without a usable HIP device immesh_create() fails.
 *   - one ctx per scan thread (reference thread A, service_LiDAR_update) ; calls on one ctx must be serialised.
 */
#ifndef IMMESH_C_API_H
#define IMMESH_C_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IMMESH_STATE_DOUBLES 348
#define IMMESH_E_INVAL (-1)
#define IMMESH_E_NODEV (-2)
#define IMMESH_E_NOMEM (-3)
#define IMMESH_E_CAPACITY (-4)
#define IMMESH_E_HIP (-5)
#define IMMESH_NOT_READY 1   /* immesh_mesh_collect_begin: no finished job within the timeout (not an error) */

typedef struct immesh_ctx immesh_ctx;

/* Parameters the reference reads once in Voxel_mapping::read_ros_parameters (src/voxel_mapping_common.cpp:625-707)
 * and ImMesh_node.cpp:254-272. */
typedef struct immesh_config {
    double voxel_size;        /* voxel/max_voxel_size                          config/avia.yaml:53 */
    int32_t max_layer;        /* voxel/max_layer                               :54 */
    int32_t layer_init[5];    /* voxel/layer_init_size                         :55 */
    int32_t max_points_size;  /* voxel/max_points_size                         :56 */
    double planer_threshold;  /* voxel/min_eigen_value                         :50 */
    double dept_err;          /* noise_model/ranging_cov                       :48 */
    double beam_err;          /* noise_model/angle_cov                         :49 */
    int32_t calib_laser;      /* preprocess/calib_laser (KITTI)                config/velodyne.yaml:26 */
    double sigma_num;         /* 3.0, hard-coded at src/voxel_mapping.cpp:1365 */
    int32_t max_iter;         /* mapping/max_iteration -> NUM_MAX_ITERATIONS   :3 */
    double extR[9];           /* mapping/extrinsic_R                           :41 */
    double extT[3];           /* mapping/extrinsic_T                           :40 */
    double mesh_min_spacing;  /* meshing/points_minimum_scale * distance_scale (ImMesh_node.cpp:254-270) */
    double mesh_voxel;        /* meshing/voxel_resolution * distance_scale */
    double mesh_region;       /* meshing/region_size * distance_scale */
    int32_t mesh_append_budget; /* meshing/number_of_pts_append_to_map */
    int32_t device;           /* HIP device ordinal */
    /* capacities of the HBM-resident pools (0 = library default) */
    int64_t cap_root_voxels;  /* root voxels in the registration map hash */
    int64_t cap_nodes;        /* octree nodes (roots + children) */
    int64_t cap_point_chunks; /* 16-point chunks backing OctoTree::m_temp_points_ */
    int64_t cap_vertices;     /* mesh vertices (Global_map::m_rgb_pts_vec) */
    int64_t cap_triangles;    /* distinct triangles ever inserted (Triangle_manager::m_triangle_hash) */
    int64_t cap_scan_points;  /* largest scan (raw points) */
    /* multi-GPU sharding of the registration map (0 / 1 = off): this context keeps the root voxels whose 2^shard_brick_log2-voxel brick
     * hashes to shard_rank (plus a one-voxel halo) and matches only the scan points falling into voxels it owns; see immesh_set_allreduce */
    int32_t shard_rank;
    int32_t shard_world;
    int32_t shard_brick_log2;  /* 0 = default 5: 32^3-voxel bricks */
    int32_t shard_mesh;        /* 1 = the mesher is sharded too (owner-computed admission / kNN / Delaunay per mesh-voxel brick, boundary band by all-gather, see immesh_set_allgather); 0 = every context meshes whatever it is handed */
    /* which rank owns brick (bx, by, bz): 0 = LATTICE COLOURING, owner = (bx + 3 by + 5 bz) mod shard_world -- along every axis consecutive bricks cycle
     * through all ranks (1, 3, 5 are units mod 2 / 4 / 8), so any axis-aligned surface patch a scan touches is dealt out evenly and neighbouring bricks never
     * share an owner (round 5: the busiest of 8 ranks owns 0.137 of a configs[4] scan instead of 0.170, the idlest 0.117 instead of 0.083); 1 = hash(brick) mod shard_world (rounds 1-4).  The colouring needs 3 and 5 to be units
     * mod shard_world: a world divisible by 3 or 5 (3, 5, 6, 10, 12, 15 ...) uses the hash whatever this field says */
    int32_t shard_scheme;
    int32_t reserved_;
} immesh_config;

void immesh_default_config(immesh_config* cfg); /* avia.yaml + mapping_avia.launch values */

immesh_ctx* immesh_create(const immesh_config* cfg); /* NULL on failure (see immesh_create_error) */
const char* immesh_create_error(void);
void immesh_destroy(immesh_ctx* ctx);
const char* immesh_last_error(immesh_ctx* ctx);

/* ---- registration map ------------------------------------------------------------------------------------- */
/* bool Voxel_mapping::voxel_map_init()              src/voxel_mapping.cpp:1243  (+ buildVoxelMap :110)
 * pts_body_xyz = m_feats_undistort (ALL raw points of the first scan, lidar frame), n x 3 float. */
int immesh_map_build(immesh_ctx* ctx, const float* pts_body_xyz, int64_t n, const double* state);

/* void Voxel_mapping::lio_state_estimation(StatesGroup&)   src/voxel_mapping.cpp:1284
 * One full iterated-EKF update.  pts_down_body_xyz = m_feats_down_body (n_ds x 3).  state_prior = state_propagat,
 * state_inout = `state` (in: prior, out: posterior incl. covariance).  Optional outputs (may be NULL):
 *   n_iter_out, n_match_out (m_effct_feat_num of the last iteration), res_mean_out (m_res_mean_last),
 *   eff_pts_body  [n_ds*3]  m_laserCloudOri (matched body points, match order = ascending scan index),
 *   eff_norm_dis  [n_ds*4]  m_corr_normvect  (float normal xyz + residual in .intensity). */
int immesh_register(immesh_ctx* ctx, const float* pts_down_body_xyz, int32_t n_ds, const double* state_prior, double* state_inout,
                    int32_t* n_iter_out, int32_t* n_match_out, double* res_mean_out, float* eff_pts_body, float* eff_norm_dis);

/* m_laserCloudOri / m_corr_normvect of the LAST registration on the context (immesh_register, immesh_process_scan), for the one consumer outside
 * lio_state_estimation -- publish_effect_world, src/voxel_mapping_common.cpp:533-546: matched body points (n x 3) and float normal + residual (n x 4),
 * match order = ascending scan index; capacity n_ds each, either may be NULL.  Valid until the next registration call; copies device -> host. */
int immesh_last_matches(immesh_ctx* ctx, float* eff_pts_body, float* eff_norm_dis, int32_t cap, int32_t* n_out);

/* One matcher + H-build pass at a fixed state: BuildResidualListOMP (src/voxel_mapping.cpp:153) + the residual /
 * Jacobian loops (:1372-1392, :1487-1575) reduced to HTH = H^T R^-1 H (6x6) and HTz = H^T R^-1 z (6).
 * Optional per-match outputs (capacity n_ds each, may be NULL): match_idx (scan index), normals (3 doubles),
 * dis (float residual), r_inv. */
int immesh_residuals(immesh_ctx* ctx, const float* pts_down_body_xyz, int32_t n_ds, const double* state, double* HTH36, double* HTz6,
                     int32_t* n_match, int32_t* match_idx, double* normals, float* dis, double* r_inv);

/* void Voxel_mapping::map_incremental_grow()  (voxel-map half)   src/ImMesh_mesh_reconstruction.cpp:377-408
 * + updateVoxelMap src/voxel_mapping.cpp:320.  Must follow immesh_register()/immesh_residuals() of the same scan
 * only in the sense that it recomputes the per-point body covariances itself (no hidden dependency). */
int immesh_map_update(immesh_ctx* ctx, const float* pts_down_body_xyz, int32_t n_ds, const double* state);

/* ---- meshing ---------------------------------------------------------------------------------------------- */
/* void incremental_mesh_reconstruction(cloud, q, t, frame_idx)   src/ImMesh_mesh_reconstruction.cpp:92
 * pts_world_xyzi = world_lidar_full (n_raw x 4 float).  Runs append + per-voxel retriangulation + diff + commit on
 * the device; results stay in the ctx until the next call and are read with immesh_mesh_sizes / immesh_mesh_fetch. */
int immesh_mesh_scan(immesh_ctx* ctx, const float* pts_world_xyzi, int32_t n_raw, const double* sensor_pos, int32_t frame_idx);
/* void reconstruct_mesh_from_pointcloud(cloud, double)   src/ImMesh_mesh_reconstruction.cpp:328-345  (offline entry of ImMesh_node.cpp:235-244)
 * VoxelGrid(leaf, the reference passes 0.01) of the whole cloud, then one incremental_mesh_reconstruction call with the identity pose.  Set
 * mesh_append_budget >= the cloud size (config/offline_pointcloud.yaml:71: 50000000) so that every point is offered. */
int immesh_reconstruct_mesh_from_pointcloud(immesh_ctx* ctx, const float* pts_xyzi, int32_t n, double leaf);
/* Asynchronous meshing (immesh_process_scan with do_mesh == 2): the scan is queued for the mesher's own stream / worker thread -- the
 * counterpart of the reference's service_reconstruct_mesh thread (ImMesh_mesh_reconstruction.cpp:272-310) -- and the call returns after
 * registration + map update.  Jobs run strictly in submission order; at most two are outstanding (a third submission blocks) and the
 * results of the two newest jobs are kept.  immesh_mesh_wait() blocks until the newest submitted job has finished and makes its
 * results the ones immesh_mesh_sizes / immesh_mesh_fetch / immesh_last_timing report; it returns that job's status. */
int immesh_mesh_wait(immesh_ctx* ctx);

/* The service-thread side of asynchronous meshing -- the counterpart of service_reconstruct_mesh popping g_rec_mesh_data_package_list
 * (src/ImMesh_mesh_reconstruction.cpp:272-310).  ONE other thread may call immesh_mesh_collect_begin / immesh_mesh_sizes / immesh_mesh_fetch /
 * immesh_mesh_world_scan / immesh_mesh_collect_end on a context while the scan thread keeps calling immesh_process_scan(.., IMMESH_MESH_ASYNC):
 *   immesh_mesh_collect_enable(ctx, 1)   once, before the first asynchronous scan.  From then on jobs are handed out strictly in submission order and a
 *                                        scan whose mesh job would reuse the result buffers of a job not yet collected (two sets, alternating) blocks in
 *                                        immesh_process_scan until the collector has caught up -- no frame is dropped (the reference drops frames only
 *                                        above 1e5 queued packages, :289-294).  The scan thread must then not use immesh_mesh_wait / the synchronous modes.
 *   immesh_mesh_collect_begin(ctx, timeout_ms, &ordinal)   waits (at most timeout_ms) for the oldest job not yet collected, makes its results the ones
 *                                        immesh_mesh_sizes / immesh_mesh_fetch return; IMMESH_NOT_READY on timeout, else the job's status.  ordinal =
 *                                        1 for the first job submitted on the context, 2 for the second, ...
 *   immesh_mesh_collect_end(ctx)         releases the job's result buffers. */
int immesh_mesh_collect_enable(immesh_ctx* ctx, int32_t on);
int immesh_mesh_collect_begin(immesh_ctx* ctx, int32_t timeout_ms, int64_t* job_ordinal);
int immesh_mesh_collect_end(immesh_ctx* ctx);

typedef struct immesh_mesh_sizes_t {
    int32_t vtx_base;   /* id of the first vertex appended by this scan */
    int32_t n_new_vtx;  /* vertices appended (ids vtx_base .. vtx_base+n_new_vtx-1) */
    int32_t n_add;      /* triangles inserted   (Triangle_manager::insert_triangle, triangle.hpp:330) */
    int32_t n_rem;      /* triangles erased     (remove_triangle_list, triangle.hpp:212) */
    int32_t n_upd;      /* existing triangles whose m_index_flip was rewritten to a different value (correct_triangle_index) */
    int32_t n_smooth;   /* vertices whose smoothed position changed (RGB_pts::set_smooth_pos) */
    int32_t n_voxels_meshed;
    int32_t reserved;
} immesh_mesh_sizes_t;
int immesh_mesh_sizes(immesh_ctx* ctx, immesh_mesh_sizes_t* sizes);
/* Diagnostics: the neighbourhood size n_u (vertices of the voxel + its 20-NN union, retrieve_neighbor_pts_kdtree  src/meshing/mesh_rec_geometry.cpp:336-377)
 * of every voxel the newest finished job triangulated, in that job's voxel order.  out may be NULL to query the count. */
int immesh_mesh_neighbourhood_sizes(immesh_ctx* ctx, int32_t* out, int32_t cap, int32_t* n_out);
/* Diagnostics (parity tests): the world-frame scan the newest finished job meshed -- world_lidar_full as transformLidar left it
 * (src/voxel_mapping_common.cpp:709-726: f64 compute, f32 store), i.e. for immesh_process_scan the cloud the registration launch's epilogue wrote
 * with the posterior pose.  out_xyzi: cap_pts x 4 floats (host), may be NULL to query the count.  A full-pipeline run is compared exactly by
 * feeding this cloud to a second mesher (tests/test_gpu_parity_fullsize.py). */
int immesh_mesh_world_scan(immesh_ctx* ctx, float* out_xyzi, int32_t cap_pts, int32_t* n_out);
/* All lists are sorted (triplets: ids ascending inside a triplet, triplets lexicographic; smooth ids ascending).
 * Any pointer may be NULL to skip that list. */
int immesh_mesh_fetch(immesh_ctx* ctx, float* new_vtx_xyz, int32_t* tri_add, uint8_t* flip_add, int32_t* tri_rem, int32_t* tri_upd,
                      uint8_t* flip_upd, int32_t* smooth_ids, double* smooth_xyz);

/* ---- mesh export: the consumer after the path --------------------------------------------------------------- */
/* void save_to_ply_file(std::string ply_file, double smooth_factor, double knn)   src/meshing/mesh_rec_geometry.cpp:71-131
 * Vertices: smooth_factor == 0 -> raw positions; else Global_map::smooth_pts (pointcloud_rgbd.cpp:932-958) with knn = 20 (g_ply_smooth_k) and
 * maximum distance 1.25 x mesh voxel: pt*(1-f) + f * mean of the 2nd..20th nearest vertices closer than that (none -> NaN, as the reference).
 * Faces: every live triangle, (v0,v1,v2) when m_index_flip != 0 else (v0,v2,v1), ordered lexicographically by sorted triplet.
 * immesh_mesh_export leaves both arrays on the device (immesh_mesh_export_fetch copies them out); immesh_save_ply writes the binary
 * little-endian PLY layout of pcl::io::savePLYFileBinary (float x y z; list uchar int vertex_indices).  The map is not modified. */
int immesh_mesh_export(immesh_ctx* ctx, double smooth_factor, int32_t knn, int64_t* n_vtx_out, int64_t* n_faces_out);
int immesh_mesh_export_fetch(immesh_ctx* ctx, float* vtx_xyz, int32_t* faces);
int immesh_save_ply(immesh_ctx* ctx, const char* path, double smooth_factor, int32_t knn);

/* ---- the renderer's consumer of the map (SURVEY 8(b) "service_refresh_and_synchronize_triangle ... Global_map::smooth_pts", 8(f) rank 3 "GL sync") ---- */
/* vec_3 Global_map::smooth_pts( RGB_pt_ptr&, double smooth_factor, double knn, double maximum_smooth_dis )   src/meshing/r3live/pointcloud_rgbd.cpp:932-958
 * as the renderer calls it for every triangle vertex with m_smoothed == false (src/meshing/mesh_rec_display.cpp:86-90; vertices of voxels that never
 * reached 3 points are never smoothed by the mesher, src/ImMesh_mesh_reconstruction.cpp:147-151, but are pulled into their neighbours' triangulations).
 * In the reference the function searches the HOST ikd-Tree, which a drop-in never feeds (zero neighbours -> 0/0 -> a NaN vertex in the GL buffer): the
 * search runs here, on the device's map.  For each of the n vertex ids: the knn nearest vertices of the whole map, the nearest (the vertex itself)
 * skipped, those closer than maximum_smooth_dis (<= 0: 0.8 x the mesh voxel; at most 2.5 x the mesh voxel, the reach of the device's 20-NN pull;
 * the renderer passes g_kd_tree_accept_pt_dis = 1.25 x) averaged: out = pt (1 - f) + f mean; nobody close -> NaN as in the reference.  knn must be 20
 * (g_ply_smooth_k).  The map is NOT modified (the reference also stores the value in the point -- RGB_pts::set_smooth_pos -- which is the host mirror's
 * business: INTEGRATION.md).  Thread-safe against a running scan loop: may be called from a third thread (the renderer's) while immesh_process_scan /
 * immesh_mesh_collect_* run on theirs; the query reads the map between two mesh jobs.  Ids must be vertices the caller has been handed (immesh_mesh_fetch). */
int immesh_smooth_pts(immesh_ctx* ctx, const int32_t* vertex_ids, int32_t n, double smooth_factor, int32_t knn, double maximum_smooth_dis, double* out_xyz);
/* The float vertex positions unparse_triangle_set_to_vector puts into the GL buffer (src/meshing/mesh_rec_display.cpp:78-103): RGB_pts::get_pos(1) AFTER the
 * on-demand smoothing above -- m_pos_aft_smooth where the mesher has smoothed the vertex, smooth_pts' value where it has not -- cast to float, for a
 * batch of ids (3 per triangle of a region's Triangle_set, in the caller's order): one call per refreshed region instead of one smooth_pts per vertex. */
int immesh_mesh_display_vertices(immesh_ctx* ctx, const int32_t* vertex_ids, int32_t n, double smooth_factor, int32_t knn, double maximum_smooth_dis, float* out_xyz);

/* ---- whole scan (what service_LiDAR_update does per scan, src/voxel_mapping.cpp:1959-1973) ---------------- */
/* lio_state_estimation + map_incremental_grow (+ world transform of the full scan and incremental_mesh_reconstruction
 * when do_mesh != 0).  pts_raw_body_xyzi = m_feats_undistort (n_raw x 4).  Everything stays on the device between
 * stages; mesh results are read with immesh_mesh_sizes / immesh_mesh_fetch.
 * do_mesh: IMMESH_MESH_OFF = registration + map update only, IMMESH_MESH_SYNC = mesh synchronously, IMMESH_MESH_ASYNC = queue the mesh job
 * and return (see immesh_mesh_wait).  IMMESH_MESH_ASYNC, or IMMESH_SCAN_NOWAIT or-ed in, also returns without waiting for the map update: the
 * pose is final when the call returns, map growth finishes on the stream ahead of the next call's work, and a capacity error of that update is
 * reported by the next call on the context (immesh_last_timing / immesh_counters wait for it). */
#define IMMESH_MESH_OFF 0
#define IMMESH_MESH_SYNC 1
#define IMMESH_MESH_ASYNC 2
#define IMMESH_SCAN_NOWAIT 0x10
int immesh_process_scan(immesh_ctx* ctx, const float* pts_down_body_xyz, int32_t n_ds, const float* pts_raw_body_xyzi, int32_t n_raw,
                        const double* state_prior, double* state_inout, int32_t frame_idx, int32_t do_mesh, int32_t* n_iter_out,
                        int32_t* n_match_out);

/* The same call on the clouds AS THE REFERENCE HOLDS THEM (src/voxel_mapping.cpp:1959-1973: m_feats_down_body, m_feats_undistort are pcl clouds of
 * PointType = pcl::PointXYZINormal, 48 bytes a point: x y z at 0, intensity at 32; pcl::PointXYZI: 32 bytes, intensity at 16): point i of a cloud lies at
 * base + i * stride_bytes, x y z first, the raw cloud's intensity at raw_intensity_offset_bytes.  Consumed in place -- host clouds are packed in ONE pass into
 * the library's pinned staging and copied asynchronously, device clouds are gathered by a kernel -- so the caller keeps no packed copy and may reuse its
 * clouds as soon as the call returns (no immesh_inputs_consumed needed).  Results identical to immesh_process_scan on the packed clouds, bit for bit. */
int immesh_process_scan_strided(immesh_ctx* ctx, const void* pts_down_body, int32_t n_ds, int32_t down_stride_bytes, const void* pts_raw_body, int32_t n_raw,
                                int32_t raw_stride_bytes, int32_t raw_intensity_offset_bytes, const double* state_prior, double* state_inout, int32_t frame_idx,
                                int32_t do_mesh, int32_t* n_iter_out, int32_t* n_match_out);

/* ---- legacy registration path ------------------------------------------------------------------------------- */
/* `voxel_map_en = false` -- dead in every shipped config, kept behind these separate entry points: the ikd-Tree of map points and the "Old map
 * ICP" matcher.  The tree is replaced by what it computes: exact float k-NN and the box-downsample insert (one survivor per downsample_size
 * box, the point nearest to the box centre, a new point winning ties), on a device hash grid.
 *   m_ikdtree.set_downsample_param(filter_size_map_min); m_ikdtree.Build(feats_down_world)   src/voxel_mapping.cpp:1906-1914 */
int immesh_ikd_build(immesh_ctx* ctx, const float* pts_world_xyz, int32_t n, double downsample_size);
/*   m_ikdtree.Add_Points(feats_down_world, true)   src/ImMesh_mesh_reconstruction.cpp:426-443 (KD_TREE::Add_Points, ikd_Tree.cpp:493-545) */
int immesh_ikd_add_points(immesh_ctx* ctx, const float* pts_world_xyz, int32_t n);
/*   Voxel_mapping::lio_state_estimation with m_use_new_map == false: 5-NN + esti_plane (include/common_lib.h:356-402) + gates
 *   (src/voxel_mapping.cpp:1400-1480), H rows with R_inv = 1 / laser_point_cov and the iterated EKF incl. the re-match rule (:1487-1650).
 *   match_idx (n_ds) / normals_pd2 (n_ds x 4: plane normal + signed distance) = m_laserCloudOri / m_corr_normvect of the last iteration; may be NULL. */
int immesh_ikd_register(immesh_ctx* ctx, const float* pts_down_body_xyz, int32_t n_ds, const double* state_prior, double* state_inout,
                        double laser_point_cov, int32_t* n_iter, int32_t* n_match, double* res_mean, int32_t* match_idx, float* normals_pd2);
/*   m_ikdtree.Delete_Point_Boxes(boxes)   include/ikd-Tree/ikd_Tree.cpp:655-690: boxes = nb x 6 floats (min xyz, max xyz), a point goes when
 *   min <= p < max on every axis.  immesh_ikd_fov_segment = Voxel_mapping::laser_map_fov_segment (src/voxel_mapping_common.cpp:214-288): the
 *   local-map cube (side cube_len) follows pos_lid; when the sensor comes within 1.5 x detection_range of a face the cube is shifted and the
 *   slabs that fall out are deleted.  The cube lives in the context (reset by immesh_ikd_build). */
int immesh_ikd_delete_boxes(immesh_ctx* ctx, const float* boxes, int32_t nb, int32_t* n_deleted);
int immesh_ikd_fov_segment(immesh_ctx* ctx, const double* pos_lid, double cube_len, double detection_range, int32_t* n_deleted);
/*   KD_TREE::validnum / flatten / Nearest_Search(point, 5, ..) -- introspection for parity (dump order unspecified; k-NN ascending distance) */
int immesh_ikd_size(immesh_ctx* ctx, int64_t* n);
int immesh_ikd_dump(immesh_ctx* ctx, float* xyz, int64_t cap, int64_t* n_out);
int immesh_ikd_knn(immesh_ctx* ctx, const float* q_xyz, int32_t nq, float* nn_xyz /* nq x 5 x 3 */, float* d2 /* nq x 5 */, int32_t* n_found /* nq */);

/* ---- before the path: sensor decode ----------------------------------------------------------------------------- */
/* void Preprocess::avia_handler(const livox_ros_driver::CustomMsg::ConstPtr&)   src/preprocess.cpp:139-232, feature_enabled == false (every
 * shipped config).  wire_points: msg->points as serialised on the wire, n x 19 bytes, little endian
 * {uint32 offset_time [ns]; float32 x, y, z; uint8 reflectivity, tag, line}.  Points 1 .. n-1 with line < n_scans are counted; every
 * point_filter_num-th of them is kept when reflectivity > 4 and x^2+y^2+z^2 > blind^2.  out_xyzit: n_out x 5 floats
 * (x, y, z, intensity = reflectivity, curvature = offset_time / 1e6 [ms]) in arrival order -- the layout immesh_undistort consumes.
 * out may be a host or a device pointer (capacity n); NULL leaves the cloud in the context (immesh_decode_result). */
int immesh_decode_livox(immesh_ctx* ctx, const uint8_t* wire_points, int32_t n, int32_t n_scans, int32_t point_filter_num, double blind,
                        float* out_xyzit, int32_t* n_out);
/* void Preprocess::velodyne_handler(const sensor_msgs::PointCloud2::ConstPtr&)   src/preprocess.cpp:497-526.  data: msg->data (n points of
 * point_step bytes); off_*: byte offsets of the float32 fields x, y, z, intensity (msg->fields).  Keeps the points whose elevation
 * atan(z / sqrt(x^2+y^2)) lies in [-24.33, 2] degrees and whose HDL-64 scan id is in [0, 50]; curvature is 0 (the handler does not set it). */
int immesh_decode_velodyne(immesh_ctx* ctx, const uint8_t* data, int32_t n, int32_t point_step, int32_t off_x, int32_t off_y, int32_t off_z,
                           int32_t off_intensity, int32_t n_scans, float* out_xyzit, int32_t* n_out);
const float* immesh_decode_result(immesh_ctx* ctx);

/* ---- before the path: motion undistortion ---------------------------------------------------------------- */
typedef struct immesh_imu_sample {   /* sensor_msgs/Imu: header.stamp, angular_velocity, linear_acceleration */
    double t;
    double gyr[3];
    double acc[3];
} immesh_imu_sample;
typedef struct immesh_imu_ctx {      /* the ImuProcess members UndistortPcl reads and carries to the next scan (src/IMU_Processing.h) */
    double last_lidar_end_time;      /* last_lidar_end_time_ */
    double acc_s_last[3];
    double angvel_last[3];
    immesh_imu_sample last_imu;      /* last_imu_ */
    double mean_acc_norm;            /* mean_acc.norm() of IMU_init */
    double cov_gyr[3];
    double cov_acc[3];
    double cov_bias_gyr[3];
    double cov_bias_acc[3];
    double lid_rot_to_imu[9];        /* Lid_rot_to_IMU, row-major */
    double lid_offset_to_imu[3];     /* Lid_offset_to_IMU */
} immesh_imu_ctx;
/* void ImuProcess::UndistortPcl(LidarMeasureGroup&, StatesGroup&, PointCloudXYZI&)   src/IMU_Processing.cpp:755-958   (LiDAR-only packages:
 * is_lidar_end == true).  pts_xyzit: n x 5 floats (x, y, z, intensity, curvature = offset from lidar_beg_time in milliseconds), the package's
 * cloud in arrival order.  imu: the package's samples (meas.imu) in time order.  The forward propagation of state + covariance over the IMU
 * samples is 18x18 host algebra (as in the reference); on the device the points are sorted by offset time (std::sort(time_list); equal stamps
 * keep their arrival order) and each one is moved into the scan-end frame by the pose of its IMU interval (:925-955, including the repeated
 * compensation of the earliest point that the reference's loop performs).
 * out_xyzi: host or device, n x 4, time order; NULL leaves the cloud in the context (immesh_undistort_result) for immesh_downsample /
 * immesh_process_scan.  state_inout: 348 doubles, propagated to the scan end.  last_update_time: LidarMeasureGroup::last_update_time, in/out. */
int immesh_undistort(immesh_ctx* ctx, const float* pts_xyzit, int32_t n, const immesh_imu_sample* imu, int32_t n_imu, double lidar_beg_time,
                     double* last_update_time, immesh_imu_ctx* imu_ctx, double* state_inout, float* out_xyzi);
const float* immesh_undistort_result(immesh_ctx* ctx);

/* ---- multi-GPU: voxel-hash sharded registration map ------------------------------------------------------------ */
/* One context per GPU / process, all configured with the same shard_world.  Every rank is handed the same scans; each matches the points
 * whose root voxel it owns and replays only its own voxels (+ halo) in immesh_map_update.  After every residual pass the library calls
 * `fn(buf, n, user)` which must sum buf[0..n) over all ranks in place (n = 46: H^T R^-1 H, H^T R^-1 z, 4 counters) and return 0 -- e.g.
 * torch.distributed.all_reduce / ncclAllReduce over xGMI.  The 18-state update then runs identically on every rank. */
typedef int (*immesh_allreduce_fn)(double* buf, int32_t n, void* user);
int immesh_set_allreduce(immesh_ctx* ctx, immesh_allreduce_fn fn, void* user);
/* Sharded mesher (shard_world > 1, shard_mesh = 1; shard_brick_log2 >= 2).  Every rank is handed the same world-frame scans.  Mesh voxels are owned in
 * bricks of 2^shard_brick_log2 voxels per axis (owner: immesh_config::shard_scheme); the OWNER of a brick tests the candidates falling into it against
 * the map and decides them (Global_map::append_points_to_global_map, pointcloud_rgbd.cpp:411-552), searches its voxels' neighbourhoods and triangulates
 * them.  Only the BOUNDARY BAND travels, by all-gather:
 *   1. admission, in rounds: a rank's candidates that survived the test against the map and lie within min_spacing of another rank's brick, then the
 *      decisions (every accept; the rejects of band candidates) -- until no rank has an undecided candidate (two rounds unless a dependency chain
 *      crosses a brick face twice).  Every rank then commits the same new vertices: ids are the serial ones (the prefix sum over the accept flags in
 *      scan-index order is the same on every rank); the 16 bytes per vertex are the one thing that stays replicated;
 *   2. this scan's smoothed positions of the voxels that lie within reach (2 voxel indices = 1.25 voxel + rounding) of another rank's brick;
 *   3. the triangle marks (add / keep / remove + flip word) of triangles with a vertex within reach of another rank's brick.
 * A rank keeps of what it receives only what lies within reach of ITS bricks: its triangle store and smoothed positions cover its bricks + halo.
 * immesh_mesh_sizes / immesh_mesh_fetch return THIS RANK'S PART of the result lists: the triangles whose smallest vertex lies in its bricks, the
 * smoothed vertices of its own voxels (new_vtx: all new vertices, identical on every rank).  The union of the ranks' lists is the unsharded list,
 * every entry exactly once (tests/test_gpu_sharded.py: 2 and 4 ranks).  n_triangles_live of immesh_counters adds up over the ranks likewise.
 * Every exchange is ONE all-gather of a fixed-size block per rank (16-byte header {records, aux} + the first records); a second one, padded to the
 * largest count, only when a rank had more records than the block holds.
 * cb gathers `bytes` bytes from every rank into recv (world x bytes, rank order); equal `bytes` on all ranks.  RCCL: ncclAllGather. */
typedef int (*immesh_allgather_fn)(const void* send, int64_t bytes, void* recv, void* user);
int immesh_set_allgather(immesh_ctx* ctx, immesh_allgather_fn cb, void* user);
/* RCCL inside the library: after immesh_rccl_init the sharded paths issue their collectives themselves, on the context's own HIP streams and on
 * device-resident buffers (ncclAllReduce of the 46 sums between a residual pass and the in-kernel EKF update: the scan is enqueued without a host
 * round trip; ncclAllGather of the mesher's exchange records) -- the callbacks above are then unused.  librccl.so is opened on first use.
 *   immesh_rccl_unique_id: ncclGetUniqueId, called by ONE rank; the 128 bytes travel to the others by whatever launcher the application uses
 *   (MPI, torch.distributed, a file).   immesh_rccl_init: ncclCommInitRank(world = shard_world, rank = shard_rank) -- collective over the ranks. */
int immesh_rccl_unique_id(uint8_t id_out[128]);
int immesh_rccl_init(immesh_ctx* ctx, const uint8_t id[128]);
const char* immesh_rccl_error(void);
/* SURVEY 8(e) "Scan ... broadcast once per scan": the rank that holds the scan (root: pts = n points of `stride` = 3 or 4 floats, host or device) hands it
 * to every rank of the sharded job; the other ranks pass pts = NULL.  Collective -- every rank calls it, in the same order.  On return *dev_out points at the
 * scan in THIS context's device memory (two buffers per stride used in turn: valid until the SECOND next broadcast of that stride, so a mesh job that still
 * reads scan k asynchronously has a whole scan of slack -- wait for it, immesh_mesh_wait, before broadcasting scan k + 2) and *n_out is its length: feed them to
 * immesh_process_scan / immesh_mesh_scan, which take device pointers as they are.  Every rank returns the SAME code: the header (points, floats per point,
 * each rank's cap_scan_points) is all-gathered before anything is decided -- IMMESH_E_INVAL when the root's scan is unusable, IMMESH_E_CAPACITY when it
 * exceeds some rank's cap_scan_points -- so no rank is left waiting in a collective.  RCCL (immesh_rccl_init): ncclAllGather of the headers + ncclBroadcast
 * of the points on the registration stream; otherwise through the all-gather callback (immesh_set_allgather).  Stubbed collectives
 * (immesh_stub_collectives): nobody sends, every rank must be handed the scan itself.  Unsharded context: a plain copy into the buffer. */
int immesh_broadcast_scan(immesh_ctx* ctx, const float* pts, int32_t n, int32_t stride, int32_t root, const float** dev_out, int32_t* n_out);
/* payload bytes this rank has contributed to the mesher's all-gathers, and the number of collective calls, since create */
int immesh_shard_traffic(immesh_ctx* ctx, int64_t* bytes, int64_t* calls);

/* Capacity planning on one GPU: run ONE rank of a sharded job alone.  The data-path collectives become local no-ops (the all-reduce leaves this rank's
 * partial sums, the all-gather delivers only its own records), so the context holds and processes exactly this rank's share of the map -- HBM use and
 * per-scan time of the share can be measured without the node (bench.py --dry-run-rank).  The poses / meshes are those of the rank's sub-problem. */
int immesh_stub_collectives(immesh_ctx* ctx);
/* device memory the context has allocated (registration map, mesh map, scratch), bytes */
int immesh_device_bytes(immesh_ctx* ctx, int64_t* bytes);

/* rank owning root voxel key3 under cfg's shard settings (host mirror of the kernels' ownership function) */
int immesh_shard_owner(const immesh_config* cfg, const int64_t* key3);

/* ---- the step before the path (host-side 18x18 algebra, no device work) ------------------------------ */
/* pcl::VoxelGrid (m_downSizeFilterSurf.filter, src/voxel_mapping.cpp:1888-1891) on the device: pts = n points of `stride` (3 or 4) floats, host or
 * device; result = one float32 centroid per occupied leaf, ordered by linear leaf index (SURVEY A.15 spec).  *n_out = number of leaves.  out_xyz
 * (host or device, may be NULL) receives n_out x 3 floats; the result also stays on the device (immesh_downsample_result) so that it can be fed
 * to immesh_register / immesh_process_scan without leaving HBM.  Not between immesh_downsample_begin and immesh_downsample_end of the same context
 * (IMMESH_E_INVAL: the two share the leaf table and the parameter block). */
int immesh_downsample(immesh_ctx* ctx, const float* pts, int32_t n, int32_t stride, double leaf, float* out_xyz, int32_t cap_out, int32_t* n_out);
const float* immesh_downsample_result(immesh_ctx* ctx);
/* The same VoxelGrid as an asynchronous pair: _begin enqueues the whole down-sampling of scan k+1 on the pre-processing stream and returns at once, so it
 * runs beside scan k's registration; _end waits for it and hands back the leaf count and the device-resident result (n_out x 3 floats, valid until the
 * second _begin after it: the results alternate between two buffers).  A cloud the hashed form gives up on (a leaf above 2048 points, a leaf index
 * beyond +-2^20) is redone by the radix pipeline inside _end -- the result is always the one immesh_downsample gives.  Call _begin(k+1) BEFORE
 * immesh_process_scan(k): behind an asynchronous immesh_process_scan the sequence is held (on the device, at most 150 us) until the next registration
 * launch is running, so that it keeps the registration company and not the map update of the scan before. */
int immesh_downsample_begin(immesh_ctx* ctx, const float* pts, int32_t n, int32_t stride, double leaf);
int immesh_downsample_end(immesh_ctx* ctx, int32_t* n_out, const float** dev_xyz);
/* The device buffers handed to an ASYNCHRONOUS immesh_process_scan (pts_down, pts_raw) are still read for a few microseconds after the call has
 * returned with the pose (the map update's preparation and the full-scan transform run behind the last registration pass).  The library's own
 * pre-processing entry points order themselves behind that; an application that refills such a buffer with its own kernels / copies calls this first:
 * it returns once the last asynchronous scan has consumed its input clouds.  (The reference has no counterpart: its scan thread owns the clouds until
 * map_incremental_grow returns, src/ImMesh_mesh_reconstruction.cpp:377-444, called at src/voxel_mapping.cpp:1973.) */
int immesh_inputs_consumed(immesh_ctx* ctx);

/* ImuProcess::Forward_without_imu   src/IMU_Processing.cpp:486-553 : constant-velocity prior (state + covariance) for the next scan. */
int immesh_forward_without_imu(const double* state_in, double dt, double cov_gyr, double cov_acc, double* state_out);

/* ---- introspection (parity tests, roofline denominators) -------------------------------------------------- */
typedef struct immesh_plane_rec {  /* one initialised octree node */
    int64_t key[3];      /* root voxel key (VOXEL_LOC) */
    int32_t layer;       /* 0 = root */
    int32_t path;        /* child indices from the root, 3 bits per level, first level in the low bits */
    int32_t is_plane;    /* Plane::m_is_plane */
    int32_t n_points;    /* retained m_temp_points_.size() */
    int32_t update_enable;
    int32_t new_points;
    float radius, min_eig, d;
    float pad;
    double center[3], normal[3];
    double plane_var[36];
} immesh_plane_rec;
/* writes up to cap records (unordered); *n_out = number of initialised nodes in the map */
int immesh_dump_planes(immesh_ctx* ctx, immesh_plane_rec* out, int64_t cap, int64_t* n_out);

typedef struct immesh_counters_t {  /* cumulative since create / last reset; SURVEY.md 8(d) symbols */
    int64_t n_ds, n_iter, n_match, n_plane_tests, n_extra_probe, n_refits, n_refit_pts;
    int64_t n_app, n_new, v_act, n_v, n_u, t_v, t_add, t_rem, c1, c20;
    int64_t n_root_voxels, n_nodes, n_vertices, n_triangles_live;
    int64_t n_degenerate_skips;   /* neighbourhood points delaunay_triangulation did NOT insert: no live triangle's circumdisk contains them -- a duplicate in the 2-D projection or exact
                                   * co-circularity (the deterministic rule of oracle/orc_delaunay.hpp, not CGAL's symbolic perturbation).  0 on noisy scans; lattices produce some */
} immesh_counters_t;
int immesh_counters(immesh_ctx* ctx, immesh_counters_t* out, int32_t reset);

/* Diagnostics: scans whose one-launch registration (a resident grid whose workgroups gather each other's partial sums) could not get all of its
 * workgroups onto the device within its spin bound and were registered by the per-pass launch chain instead -- same result to rounding, more
 * launches.  Non-zero only when other work has filled the device (more than two contexts registering large scans at once). */
int immesh_registration_fallbacks(immesh_ctx* ctx, int64_t* n);

/* timing of the last immesh_process_scan, milliseconds from HIP events on the ctx stream:
 * [0] total  [1] register  [2] map update  [3] mesh
 * [1] and [2] are taken for synchronous calls only: an asynchronous call (IMMESH_MESH_ASYNC or IMMESH_SCAN_NOWAIT) keeps the stage events out of
 * the stream -- every event record is a barrier packet between two kernels of the pose chain -- and reports zeros for them. */
int immesh_last_timing(immesh_ctx* ctx, float ms[4]);

/* per-kernel timing (HIP events on the ctx stream around every launch); off by default.  bench.py's roofline leg. */
typedef struct immesh_kernel_stat {
    char name[56];
    int64_t launches;
    double total_ms;
} immesh_kernel_stat;
int immesh_profile_enable(immesh_ctx* ctx, int32_t on);
int immesh_profile_read(immesh_ctx* ctx, immesh_kernel_stat* out, int32_t cap, int32_t* n_out, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif
