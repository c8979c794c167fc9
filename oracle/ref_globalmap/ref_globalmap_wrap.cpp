// ORACLE / TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE'S OWN incremental mesher, compiled from where it lies under /root/reference
// behind Eigen / PCL / CGAL / TBB shaped stubs -- the recipe is oracle/Makefile (target ref -> _ref/libref_globalmap.so):
//   include/ikd-Tree/ikd_Tree.{h,cpp}                      whole files: the vertex kd-tree (exact kNN, Add_Point)                    (rows a17, a18)
//   src/tools/tools_kd_hash.hpp                            whole file: Hash_map_3d (dedupe grid, mesh-voxel hash)                    (a26)
//   src/meshing/r3live/triangle.hpp, triangle.cpp          whole files: Triangle, Triangle_manager                                   (a21, a24)
//   src/meshing/r3live/pointcloud_rgbd.hpp:71-233          class RGB_pts, class RGB_Voxel                                            (a26)
//   src/meshing/r3live/pointcloud_rgbd.cpp:59-87           RGB_pts::set_pos / set_smooth_pos / get_pos
//   src/meshing/r3live/pointcloud_rgbd.cpp:256-267         Global_map::set_minimum_dis / set_voxel_resolution
//   src/meshing/r3live/pointcloud_rgbd.cpp:395-552         retrieve_pts_in_voxels, voxels_recent_visited, Global_map::append_points_to_global_map   (a17)
//   src/meshing/mesh_rec_geometry.cpp:24-57, 137-172, 174-295, 334-377, 379-397, 399-433
//                                                          compute_angle / is_face_is_ok, triangle_compare, delaunay_triangulation (everything around
//                                                          the CGAL call), retrieve_neighbor_pts_kdtree, remove_outlier_pts, correct_triangle_index
//                                                                                                                                    (a19, a20, a22, a23)
//   src/meshing/r3live/pointcloud_rgbd.cpp:932-958         Global_map::smooth_pts (round 6): the renderer's / the PLY export's smoothing, on the REAL ikd-Tree   (8(f) rank 3)
//   src/meshing/mesh_rec_geometry.cpp:71-131               save_to_ply_file (round 6), whole: vertices through smooth_pts, every live triangle of every region with
//                                                          its winding; pcl::io::savePLYFileBinary is a recorder (the file format is PCL's, not the reference's)
//   src/ImMesh_mesh_reconstruction.cpp:92-267              incremental_mesh_reconstruction, whole: the append step, the voxel selection, the per-voxel
//                                                          pull / triangulate / compare / orient, the commit (all removes, then all adds)   (a25)
// The excerpts are cut out by line range into _ref/gm_src/ at BUILD time (sed; the directory is removed after the compile), whole files are symlinked:
// nothing of the reference is copied into the repository.
// What is pinned: the reference's logic, end to end, frame after frame.  What is NOT: Eigen (stub), CGAL's Delaunay_triangulation_2 (the oracle's
// Bowyer-Watson stands in), TBB (a sequential loop over the voxels in ascending key order: the checker's determinism rule).
#include "pointcloud_rgbd.hpp"     // ref_globalmap/stub_pointcloud_rgbd.hpp (symlinked under that name): pulls in the reference's triangle.hpp, ikd_Tree.h
#include <tbb/tbb.h>               // ref_globalmap/stubs
#include <cstdint>
#include <cstring>
#include <map>
#include "ikd_Tree.cpp"            // the reference's kd-tree implementation (templates + explicit instantiations)

// ---- globals the bodies name (src/ImMesh_node.cpp:93-124, src/meshing/r3live/pointcloud_rgbd.cpp:53-57, src/ImMesh_mesh_reconstruction.cpp:46-64) ----
double g_initial_camera_exp_tim = 1.0;
double g_voxel_resolution = 0.1;
double g_global_map_minimum_dis = 0.01;
std::vector< RGB_pt_ptr > *g_rgb_pts_vec = nullptr;
int    appending_pts_frame = ( int ) 5e3;
double g_meshing_voxel_size = 0.4;
Global_map       g_map_rgb_pts_mesh( 0 );
Triangle_manager g_triangles_manager;
int    g_current_frame = -1;
bool   g_flag_pause = false;
FILE * g_fp_cost_time = nullptr;
double g_vx_map_frame_cost_time = 0;
static double g_LiDAR_frame_avg_time = 0;
std::mutex g_mutex_append_map;
std::mutex g_mutex_reconstruct_mesh;
// mesh_rec_geometry.hpp:32-33: a frame's points + a dynamic pose vector (Eigen::Matrix< double, -1, 1 >)
struct PoseVecX
{
    std::vector< double > a;
    PoseVecX &operator=( const Eigen::Matrix< double, 7, 1 > &v ) { a.assign( v.a, v.a + 7 ); return *this; }
    int size() const { return ( int ) a.size(); }
    struct Blk { const PoseVecX &v; int r0; operator vec_3() const { return vec_3( v.a[ r0 ], v.a[ r0 + 1 ], v.a[ r0 + 2 ] ); } };
    Blk block( int r0, int, int nr, int nc ) const { if ( nr != 3 || nc != 1 || ( int ) a.size() < r0 + 3 ) std::abort(); return Blk{ *this, r0 }; }
};
typedef std::vector< std::pair< std::vector< vec_4 >, PoseVecX > > LiDAR_frame_pts_and_pose_vec;
LiDAR_frame_pts_and_pose_vec g_eigen_vec_vec;
#define ANSI_COLOR_RED_BOLD ""
#define ANSI_COLOR_RESET ""
#define scope_color( a )
using std::cout; using std::endl;

#include "rgbd_pts_members.inc"            // pointcloud_rgbd.cpp:59-84
#include "rgbd_set_sizes.inc"              // pointcloud_rgbd.cpp:256-267
#include "rgbd_append.inc"                 // pointcloud_rgbd.cpp:395-552
#include "mg_angle.inc"                    // mesh_rec_geometry.cpp:24-57
#include "mg_triangle_compare.inc"         // :137-172

// ---- test taps: the two per-voxel calls of the frame loop are entered through recorders (the reference's definitions keep their bodies, under
// another name); the calls arrive in the order the voxels are processed
namespace ref_gm {
struct FlipCall { int tri[ 3 ]; int flip; int voxel_rank; };
struct Tap { std::vector< int > n_u; std::vector< long > voxel_keys; std::vector< FlipCall > flips; int rank = -1; };
inline Tap &tap() { static Tap t; return t; }
} // namespace ref_gm
#define delaunay_triangulation ref_delaunay_triangulation_body
#include "mg_delaunay.inc"                 // :174-295
#undef delaunay_triangulation
#include "mg_neighbor_pts.inc"             // :334-377
#include "mg_remove_outlier.inc"           // :379-397
#define correct_triangle_index ref_correct_triangle_index_body
#include "mg_correct_index.inc"            // :399-433
#undef correct_triangle_index
std::vector< long > delaunay_triangulation( std::vector< RGB_pt_ptr > &rgb_pt_vec, vec_3 &long_axis, vec_3 &mid_axis, vec_3 &short_axis, std::set< long > &convex_hull_index, std::set< long > &inner_hull_index )
{
    ref_gm::Tap &t = ref_gm::tap();
    t.rank++;
    t.n_u.push_back( ( int ) rgb_pt_vec.size() );
    return ref_delaunay_triangulation_body( rgb_pt_vec, long_axis, mid_axis, short_axis, convex_hull_index, inner_hull_index );
}
void correct_triangle_index( Triangle_ptr &ptr, const vec_3 &camera_center, const vec_3 &_short_axis )
{
    ref_correct_triangle_index_body( ptr, camera_center, _short_axis );
    ref_gm::Tap &t = ref_gm::tap();
    t.flips.push_back( ref_gm::FlipCall{ { ptr->m_tri_pts_id[ 0 ], ptr->m_tri_pts_id[ 1 ], ptr->m_tri_pts_id[ 2 ] }, ( int ) ptr->m_index_flip, t.rank } );
}
#include "mr_incremental_mesh_reconstruction.inc"   // ImMesh_mesh_reconstruction.cpp:92-267
#include "rgbd_smooth_pts.inc"                      // pointcloud_rgbd.cpp:932-958
// ---- what save_to_ply_file touches of PCL's mesh I/O: shapes + a recorder (the checker compares arrays, not PCL's file writer)
// (g_kd_tree_accept_pt_dis: the reference's own global, mesh_rec_geometry.cpp:335 in mg_neighbor_pts.inc; retrieve_neighbor_pts_kdtree sets it to 1.25 x the mesh voxel, :343)
namespace pcl {
struct PCLPointCloud2 { std::vector< float > xyz; };
struct Vertices { std::vector< int > vertices; };
struct PolygonMesh { PCLPointCloud2 cloud; std::vector< Vertices > polygons; };
inline void toPCLPointCloud2( const PointCloud< PointXYZ > &c, PCLPointCloud2 &out ) { out.xyz.resize( c.points.size() * 3 ); for ( size_t i = 0; i < c.points.size(); i++ ) { out.xyz[ i * 3 ] = c.points[ i ].x; out.xyz[ i * 3 + 1 ] = c.points[ i ].y; out.xyz[ i * 3 + 2 ] = c.points[ i ].z; } }
namespace io {
struct PlyTap { std::vector< float > xyz; std::vector< int > faces; };
inline PlyTap &ply_tap() { static PlyTap t; return t; }
inline int savePLYFileBinary( const std::string &, const PolygonMesh &m ) { PlyTap &t = ply_tap(); t.xyz = m.cloud.xyz; t.faces.clear(); for ( const Vertices &v : m.polygons ) t.faces.insert( t.faces.end(), v.vertices.begin(), v.vertices.end() ); return 0; }
inline int savePCDFileBinary( const std::string &, const PointCloud< PointXYZ > & ) { return 0; }
} // namespace io
} // namespace pcl
#include "mg_save_to_ply.inc"                       // mesh_rec_geometry.cpp:71-131

extern "C" {
// ImMesh_node.cpp:255-272: the mesher's set-up from the launch parameters
void rg_init( double minimum_pts, double voxel_size, double region_size, int appending_pts, int max_frames )
{
    g_meshing_voxel_size = voxel_size;
    appending_pts_frame = appending_pts;
    g_current_frame = -3e8;
    g_triangles_manager.m_pointcloud_map = &g_map_rgb_pts_mesh;
    g_map_rgb_pts_mesh.set_minimum_dis( minimum_pts );
    g_map_rgb_pts_mesh.set_voxel_resolution( g_meshing_voxel_size );
    g_triangles_manager.m_region_size = region_size;
    g_map_rgb_pts_mesh.m_recent_visited_voxel_activated_time = 0;
    g_eigen_vec_vec.resize( ( size_t ) max_frames + 2 );          /* ImMesh_node.cpp:276 sizes it for the run */
}
// one frame through incremental_mesh_reconstruction (ImMesh_mesh_reconstruction.cpp:92); returns the number of voxels triangulated
int rg_frame( const float *pts_world_xyzi, int n, const double *R9, const double *t3, int frame_idx )
{
    pcl::PointCloud< pcl::PointXYZI >::Ptr c( new pcl::PointCloud< pcl::PointXYZI > );
    c->points.resize( ( size_t ) n );
    for ( int i = 0; i < n; i++ ) { pcl::PointXYZI p; p.x = pts_world_xyzi[ i * 4 ]; p.y = pts_world_xyzi[ i * 4 + 1 ]; p.z = pts_world_xyzi[ i * 4 + 2 ]; p.intensity = pts_world_xyzi[ i * 4 + 3 ]; c->points[ i ] = p; }
    Eigen::Matrix3d R; for ( int i = 0; i < 9; i++ ) R.a[ i ] = R9[ i ];
    ref_gm::Tap &t = ref_gm::tap();
    t.n_u.clear(); t.flips.clear(); t.rank = -1;
    incremental_mesh_reconstruction( c, Eigen::Quaterniond( R ), Eigen::Vector3d( t3[ 0 ], t3[ 1 ], t3[ 2 ] ), frame_idx );
    return ( int ) t.n_u.size();
}
int64_t rg_n_vertices() { return ( int64_t ) g_map_rgb_pts_mesh.m_rgb_pts_vec.size(); }
void rg_vertices( double *pos, double *smooth, int64_t cap )
{
    const int64_t n = rg_n_vertices();
    for ( int64_t i = 0; i < n && i < cap; i++ )
    {
        const RGB_pt_ptr &p = g_map_rgb_pts_mesh.m_rgb_pts_vec[ i ];
        for ( int k = 0; k < 3; k++ ) { if ( pos ) pos[ i * 3 + k ] = p->m_pos[ k ]; if ( smooth ) smooth[ i * 3 + k ] = p->m_pos_aft_smooth[ k ]; }
    }
}
// live set = union of the per-region sets (what the renderer / save_to_ply_file walk): triplets + m_index_flip
int64_t rg_live( int32_t *out_tri, uint8_t *out_flip, int64_t cap )
{
    std::vector< Triangle_set > lists;
    g_triangles_manager.get_all_triangle_list( lists, nullptr, 0 );
    int64_t n = 0;
    for ( auto &s : lists )
        for ( auto &t : s )
        {
            if ( n < cap && out_tri ) { for ( int k = 0; k < 3; k++ ) out_tri[ n * 3 + k ] = t->m_tri_pts_id[ k ]; if ( out_flip ) out_flip[ n ] = ( uint8_t ) t->m_index_flip; }
            n++;
        }
    return n;
}
// the taps of the newest frame: neighbourhood sizes in voxel order; every correct_triangle_index call (triplet, resulting flip, rank of its voxel)
int rg_n_u( int32_t *out, int cap ) { ref_gm::Tap &t = ref_gm::tap(); for ( int i = 0; i < ( int ) t.n_u.size() && i < cap; i++ ) out[ i ] = t.n_u[ i ]; return ( int ) t.n_u.size(); }
int64_t rg_flip_calls( int32_t *tri, int32_t *flip, int32_t *rank, int64_t cap )
{
    ref_gm::Tap &t = ref_gm::tap();
    for ( int64_t i = 0; i < ( int64_t ) t.flips.size() && i < cap; i++ ) { for ( int k = 0; k < 3; k++ ) tri[ i * 3 + k ] = t.flips[ i ].tri[ k ]; flip[ i ] = t.flips[ i ].flip; rank[ i ] = t.flips[ i ].voxel_rank; }
    return ( int64_t ) t.flips.size();
}
// the mesh voxels visited by the newest frame (Global_map::m_voxels_recent_visited) as keys + (m_meshing_times, m_new_added_pts_count, points)
int64_t rg_recent_voxels( int64_t *keys3, int32_t *state3, int64_t cap )
{
    int64_t n = 0;
    for ( const auto &v : g_map_rgb_pts_mesh.m_voxels_recent_visited )
    {
        if ( n < cap && keys3 ) { for ( int k = 0; k < 3; k++ ) keys3[ n * 3 + k ] = v->m_pos[ k ]; if ( state3 ) { state3[ n * 3 ] = ( int32_t ) v->m_meshing_times; state3[ n * 3 + 1 ] = ( int32_t ) v->m_new_added_pts_count; state3[ n * 3 + 2 ] = ( int32_t ) v->m_pts_in_grid.size(); } }
        n++;
    }
    return n;
}
int64_t rg_n_voxels() { return ( int64_t ) g_map_rgb_pts_mesh.m_voxel_vec.size(); }
// Global_map::smooth_pts for one vertex, with what it stores in the point put back (the function under test is the value; a stored smoothed position
// would change what the following frames' correct_triangle_index reads)
void rg_smooth_pts( int id, double smooth_factor, double knn, double maximum_smooth_dis, double *out3 )
{
    RGB_pt_ptr &p = g_map_rgb_pts_mesh.m_rgb_pts_vec[ id ];
    const bool was = p->m_smoothed; const double keep[ 3 ] = { p->m_pos_aft_smooth[ 0 ], p->m_pos_aft_smooth[ 1 ], p->m_pos_aft_smooth[ 2 ] };
    const vec_3 v = g_map_rgb_pts_mesh.smooth_pts( p, smooth_factor, knn, maximum_smooth_dis );
    for ( int k = 0; k < 3; k++ ) { out3[ k ] = v( k ); p->m_pos_aft_smooth[ k ] = keep[ k ]; }
    p->m_smoothed = was;
}
// save_to_ply_file, whole (call it LAST: smooth_pts leaves every vertex smoothed); returns the number of faces, the arrays through rg_ply_fetch
int64_t rg_save_ply( double smooth_factor, double knn )
{
    save_to_ply_file( std::string( "/dev/null" ), smooth_factor, knn );
    return ( int64_t ) pcl::io::ply_tap().faces.size() / 3;
}
void rg_ply_fetch( float *xyz, int32_t *faces )
{
    pcl::io::PlyTap &t = pcl::io::ply_tap();
    if ( xyz ) std::memcpy( xyz, t.xyz.data(), t.xyz.size() * 4 );
    if ( faces ) for ( size_t i = 0; i < t.faces.size(); i++ ) faces[ i ] = t.faces[ i ];
}
}  // extern "C"
