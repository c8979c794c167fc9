// ORACLE / TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE'S OWN per-scan registration bodies, compiled from where they lie under
// /root/reference behind Eigen / PCL / ROS shaped stubs -- the recipe is oracle/Makefile (target ref -> _ref/libref_lio.so):
//   include/common_lib.h:199-288                struct StatesGroup (operator+ / += / -: boxplus / boxminus on SO(3) x R^15)
//   include/so3_math.h                          whole file (Exp / Log / RotMtoEuler / SKEW_SYM_MATRX), symlinked into the include path
//   src/voxel_loc.hpp, src/voxel_loc.cpp        whole files (OctoTree)
//   src/voxel_mapping.cpp:49, 110-354, 1221-1241   var_contrast, buildVoxelMap / BuildResidualListOMP / build_single_residual / updateVoxelMap, calcBodyVar
//   src/voxel_mapping.cpp:1243-1281             Voxel_mapping::voxel_map_init                                              (row a6)
//   src/voxel_mapping.cpp:1284-1399 + 1481-1652 Voxel_mapping::lio_state_estimation: per-point body covariance + cross matrices, covariance propagation
//                                               (a9), matcher call, residual -> clouds (a12), H / R^-1 build (a13), the 18-state iterated update, the
//                                               rematch / stop logic and the covariance update (a14).  Lines 1400-1480 -- the `else` branch of
//                                               `if ( m_use_new_map )`, the legacy ikd-Tree matcher (a27, dead in every shipped configuration) -- are
//                                               left out: they need esti_plane's Eigen QR
//   src/voxel_mapping.hpp:326-343               the pointBodyToWorld member templates
//   src/voxel_mapping_common.cpp:121-131, 709-726   Voxel_mapping::pointBodyToWorld( PointType ), Voxel_mapping::transformLidar   (a8)
//   src/ImMesh_mesh_reconstruction.cpp:67-80, 377-444   Rec_mesh_data_package, Voxel_mapping::map_incremental_grow               (a15 + the hand-over)
// The excerpts are cut out by line range into _ref/lio_src/ at BUILD time (sed; the directory is removed after the compile) and #included below:
// nothing of the reference is copied into the repository.  `class Voxel_mapping` below is a host struct that provides the members those bodies name
// (same names and types as src/voxel_mapping.hpp:149-285) and nothing else.
// What is pinned: the reference's logic -- which operands, which formulas, the order of the sums over the matches, float / double narrowing, the
// rematch and stop decisions.  What is NOT: Eigen's arithmetic (stub products are plain k-ascending sums; the 18 x 18 inverse() is Gauss-Jordan).
#include "voxel_loc.hpp"      // the reference's (via -I /root/reference/src); pulls in stubs/common_lib.h -> StatesGroup, so3_math.h
#include <algorithm>
#include <chrono>
#include <cstring>
#include <list>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <omp.h>
#include "../../include/immesh_c_api.h"
typedef unsigned int uint;
#ifndef MP_PROC_NUM
#define MP_PROC_NUM 4          /* CMakeLists.txt:21-24 */
#endif
M3D Eye3d = M3D::Identity();   /* src/voxel_mapping_common.cpp: the globals common_lib.h declares */
V3D Zero3d( 0, 0, 0 );

// ---- the shapes of what the bodies touch besides Eigen / PCL --------------------------------------------------------------------------------
struct Preprocess { bool calib_laser = false; };                                   /* src/preprocess.h: only this member is read (:1525) */
namespace geometry_msgs { struct Quaternion { double x = 0, y = 0, z = 0, w = 1; }; }
namespace tf { inline geometry_msgs::Quaternion createQuaternionMsgFromRollPitchYaw( double, double, double ) { return geometry_msgs::Quaternion(); } }
template < typename P > struct KD_TREE { template < typename V > int Add_Points( V &, bool ) { return 0; } };   /* only in the `!m_use_new_map` tail of map_incremental_grow */
// free functions of src/voxel_mapping.hpp:80-105,129
void build_single_residual( const Point_with_var &pv, const OctoTree *current_octo, const int current_layer, const int max_layer, const double sigma_num, bool &is_sucess, double &prob, ptpl &single_ptpl );
#include "vm_var_contrast.inc"     // voxel_mapping.cpp:49
#include "vm_map_and_matcher.inc"  // voxel_mapping.cpp:110-354
#include "vm_calc_body_var.inc"    // voxel_mapping.cpp:1221-1241

class Voxel_mapping
{
  public:
    // src/voxel_mapping.hpp:149-285 (the members the compiled bodies name)
    V3D    m_extT = Zero3d;
    M3D    m_extR = Eye3d;
    int    NUM_MAX_ITERATIONS = 0;
    int    m_feats_down_size = 0, m_effct_feat_num = 0;
    double m_res_mean_last = 0.05;
    double m_total_distance = 0;
    double m_solve_time = 0, m_solve_const_H_time = 0, m_kdtree_search_time = 0;
    double m_max_voxel_size, m_min_eigen_value = 0.003;
    double m_beam_err = 0.03, m_dept_err = 0.05;
    bool   m_use_new_map = true;
    std::unordered_map< VOXEL_LOC, OctoTree * > m_feat_map;
    std::vector< M3D > m_cross_mat_list;
    std::vector< M3D > m_body_cov_list;
    int                m_max_points_size;
    int                m_max_layer;
    std::vector< int > m_layer_init_size;
    vector< bool >          m_point_selected_surf;
    vector< vector< int > > m_pointSearchInd_surf;
    vector< PointVector >   m_Nearest_Points;
    double                  m_total_residual;
    double                  LASER_POINT_COV = 0.001;
    bool                    m_flg_EKF_inited = true, m_flg_EKF_converged = false, m_EKF_stop_flg = 0;
    PointCloudXYZI::Ptr m_feats_undistort = nullptr;
    PointCloudXYZI::Ptr m_feats_down_body = nullptr;
    PointCloudXYZI::Ptr m_feats_down_world = nullptr;
    PointCloudXYZI::Ptr m_laserCloudOri = nullptr;
    PointCloudXYZI::Ptr m_corr_normvect = nullptr;
    shared_ptr< Preprocess > m_p_pre = nullptr;
    KD_TREE< PointType > m_ikdtree;
    V3D             m_euler_cur;
    V3D             m_position_last = Zero3d;
    StatesGroup state;
    geometry_msgs::Quaternion m_geo_Quat;
    int m_meshing_maximum_thread_for_rec_mesh = 12;
    Voxel_mapping()
    {   // src/voxel_mapping.hpp:295-312
        m_feats_undistort = PointCloudXYZI().makeShared();
        m_feats_down_body = PointCloudXYZI().makeShared();
        m_feats_down_world = PointCloudXYZI().makeShared();
        m_laserCloudOri = PointCloudXYZI( 100000, 1 ).makeShared();
        m_corr_normvect = PointCloudXYZI( 100000, 1 ).makeShared();
        m_p_pre = std::make_shared< Preprocess >();
    }
    void pointBodyToWorld( const PointType &pi, PointType &po );
#include "vmh_point_body_to_world.inc"   // src/voxel_mapping.hpp:326-343
    void transformLidar( const Eigen::Matrix3d rot, const Eigen::Vector3d t, const PointCloudXYZI::Ptr &input_cloud, pcl::PointCloud< pcl::PointXYZI >::Ptr &trans_cloud );
    bool voxel_map_init();
    void lio_state_estimation( StatesGroup &state_propagat );
    void map_incremental_grow();
    void dump_lio_state_to_log( FILE * ) {}
};

// ---- the test tap: what every iteration of lio_state_estimation held when it ended -----------------------------------------------------------
// omp_get_wtime() is read three times per iteration (:1485 solve_start, :1577, :1648); the third read comes after the covariance update, with the
// iteration's locals G, H_T_H, I_STATE (function scope), K_1 and solution (loop scope) alive and entered in ref_lio::live() in that order.
namespace ref_lio {
struct IterRec { double HTH[ 36 ], HTz[ 6 ], sol[ 18 ], G[ 324 ], state[ 24 ], cov[ 324 ]; int n_match; double res_mean; int converged, stop; };
struct Tap { Voxel_mapping *vm = nullptr; int reads = 0; std::vector< IterRec > iters; };
inline Tap &tap() { static thread_local Tap t; return t; }
inline void store_state( const StatesGroup &s, double *o )
{
    for ( int i = 0; i < 9; i++ ) o[ i ] = s.rot_end.a[ i ];
    for ( int k = 0; k < 3; k++ ) { o[ 9 + k ] = s.pos_end[ k ]; o[ 12 + k ] = s.vel_end[ k ]; o[ 15 + k ] = s.bias_g[ k ]; o[ 18 + k ] = s.bias_a[ k ]; o[ 21 + k ] = s.gravity[ k ]; }
}
inline double wtime_hook()
{
    Tap &t = tap();
    if ( t.vm && ( ++t.reads % 3 ) == 0 )
    {
        const std::vector< Traced > &lv = live();
        IterRec r;
        std::memset( &r, 0, sizeof( r ) );
        if ( lv.size() == 5 && lv[ 0 ].rows == 18 && lv[ 1 ].cols == 18 && lv[ 4 ].cols == 1 )
        {
            const double *G = lv[ 0 ].data, *H = lv[ 1 ].data, *sol = lv[ 4 ].data;
            for ( int i = 0; i < 6; i++ ) for ( int j = 0; j < 6; j++ ) r.HTH[ i * 6 + j ] = H[ i * 18 + j ];
            std::memcpy( r.G, G, sizeof( r.G ) );
            std::memcpy( r.sol, sol, sizeof( r.sol ) );
        }
        else
            std::abort();   // the function's locals are not what this tap was written for
        const std::vector< double > &hz = Eigen::last_dyn_matvec();
        for ( int k = 0; k < 6 && k < ( int ) hz.size(); k++ ) r.HTz[ k ] = hz[ k ];
        store_state( t.vm->state, r.state );
        std::memcpy( r.cov, t.vm->state.cov.a, sizeof( r.cov ) );
        r.n_match = t.vm->m_effct_feat_num;
        r.res_mean = t.vm->m_res_mean_last;
        r.converged = t.vm->m_flg_EKF_converged ? 1 : 0;
        r.stop = t.vm->m_EKF_stop_flg ? 1 : 0;
        t.iters.push_back( r );
    }
    return 0.0;
}
} // namespace ref_lio
#define omp_get_wtime() ref_lio::wtime_hook()

// ---- globals of src/ImMesh_mesh_reconstruction.cpp:53-64, 82-83, 311-315, 347 that map_incremental_grow names ----------------------------------
bool   g_flag_pause = false;
double g_LiDAR_frame_start_time = 0, g_vx_map_frame_cost_time = 0;
FILE * g_fp_lio_state = nullptr;
int    g_frame_idx = 0;
#include "mr_data_package.inc"           // ImMesh_mesh_reconstruction.cpp:67-80: struct Rec_mesh_data_package
std::mutex                          g_mutex_data_package_lock;
std::list< Rec_mesh_data_package > g_rec_mesh_data_package_list;
void start_mesh_threads( int = 20 ) {}   // (:315: starts the mesher's thread pool -- the mesher is pinned elsewhere)
void open_log_file() {}                  // (:347)

#include "vmc_point_body_to_world.inc"   // voxel_mapping_common.cpp:121-131
#include "vmc_transform_lidar.inc"       // voxel_mapping_common.cpp:709-726
#include "vm_voxel_map_init.inc"         // voxel_mapping.cpp:1243-1281
#include "vm_lio_head.inc"               // voxel_mapping.cpp:1284-1399
#include "vm_lio_tail.inc"               // voxel_mapping.cpp:1481-1652
#include "mr_map_incremental_grow.inc"   // ImMesh_mesh_reconstruction.cpp:377-444
#undef omp_get_wtime

namespace {
void load_state( const double *s, StatesGroup &st )
{
    for ( int i = 0; i < 9; i++ ) st.rot_end.a[ i ] = s[ i ];
    for ( int k = 0; k < 3; k++ ) { st.pos_end[ k ] = s[ 9 + k ]; st.vel_end[ k ] = s[ 12 + k ]; st.bias_g[ k ] = s[ 15 + k ]; st.bias_a[ k ] = s[ 18 + k ]; st.gravity[ k ] = s[ 21 + k ]; }
    for ( int i = 0; i < 324; i++ ) st.cov.a[ i ] = s[ 24 + i ];
}
void store_full( const StatesGroup &st, double *s ) { ref_lio::store_state( st, s ); for ( int i = 0; i < 324; i++ ) s[ 24 + i ] = st.cov.a[ i ]; }
void fill_cloud( PointCloudXYZI::Ptr &c, const float *xyz, int n, int stride )
{
    c->points.resize( ( size_t ) n );
    for ( int i = 0; i < n; i++ )
    {
        PointType p;
        p.x = xyz[ ( size_t ) i * stride + 0 ]; p.y = xyz[ ( size_t ) i * stride + 1 ]; p.z = xyz[ ( size_t ) i * stride + 2 ];
        p.intensity = stride > 3 ? xyz[ ( size_t ) i * stride + 3 ] : 0.f;
        c->points[ i ] = p;
    }
}
void dump_node( const VOXEL_LOC &k, const OctoTree *n, int path, int depth, immesh_plane_rec *out, int64_t cap, int64_t &cnt )
{
    if ( n->m_init_octo_ )
    {
        if ( cnt < cap && out )
        {
            immesh_plane_rec &r = out[ cnt ];
            std::memset( &r, 0, sizeof( r ) );
            const Plane &p = *n->m_plane_ptr_;
            r.key[ 0 ] = k.x; r.key[ 1 ] = k.y; r.key[ 2 ] = k.z;
            r.layer = n->m_layer_; r.path = path; r.is_plane = p.m_is_plane ? 1 : 0; r.n_points = ( int ) n->m_temp_points_.size();
            r.update_enable = n->m_update_enable_ ? 1 : 0; r.new_points = n->m_new_points_;
            r.radius = p.m_radius; r.min_eig = p.m_min_eigen_value; r.d = p.m_d;
            for ( int i = 0; i < 3; i++ ) { r.center[ i ] = p.m_center( i ); r.normal[ i ] = p.m_normal( i ); }
            for ( int i = 0; i < 6; i++ ) for ( int j = 0; j < 6; j++ ) r.plane_var[ i * 6 + j ] = p.m_plane_var( i, j );
        }
        cnt++;
    }
    for ( int l = 0; l < 8; l++ )
        if ( n->m_leaves_[ l ] ) dump_node( k, n->m_leaves_[ l ], path | ( l << ( 3 * depth ) ), depth + 1, out, cap, cnt );
}
} // namespace

extern "C" {
// the configuration the node reads from yaml (voxel_mapping_common.cpp:600-700) -- same fields as immesh_config
void *rl_create( const immesh_config *c )
{
    Voxel_mapping *v = new Voxel_mapping();
    v->m_max_voxel_size = c->voxel_size; v->m_max_layer = c->max_layer; v->m_layer_init_size.assign( c->layer_init, c->layer_init + 5 );
    v->m_max_points_size = c->max_points_size; v->m_min_eigen_value = c->planer_threshold;
    v->m_dept_err = c->dept_err; v->m_beam_err = c->beam_err; v->m_p_pre->calib_laser = c->calib_laser != 0; v->NUM_MAX_ITERATIONS = c->max_iter;
    for ( int i = 0; i < 9; i++ ) v->m_extR.a[ i ] = c->extR[ i ];
    for ( int k = 0; k < 3; k++ ) v->m_extT[ k ] = c->extT[ k ];
    return v;
}
void rl_destroy( void *p ) { delete ( Voxel_mapping * ) p; }
void rl_set_state( void *p, const double *s348 ) { load_state( s348, ( ( Voxel_mapping * ) p )->state ); }
void rl_get_state( void *p, double *s348 ) { store_full( ( ( Voxel_mapping * ) p )->state, s348 ); }
// voxel_map_init (voxel_mapping.cpp:1243) on m_feats_undistort = the raw scan, at the state set before
int rl_map_init( void *p, const float *pts_raw_xyz, int n )
{
    Voxel_mapping *v = ( Voxel_mapping * ) p;
    fill_cloud( v->m_feats_undistort, pts_raw_xyz, n, 3 );
    return v->voxel_map_init() ? 0 : -1;
}
// lio_state_estimation (voxel_mapping.cpp:1284) on m_feats_down_body = the down-sampled scan: state in / out through rl_set_state / rl_get_state.
// Returns the number of iterations run; rl_iter hands out what each held when it ended.
int rl_lio( void *p, const float *pts_down_xyz, int n_ds, const double *state_propagat348 )
{
    Voxel_mapping *v = ( Voxel_mapping * ) p;
    fill_cloud( v->m_feats_down_body, pts_down_xyz, n_ds, 3 );
    v->m_feats_down_size = n_ds;                                   /* service_LiDAR_update, voxel_mapping.cpp:1891 */
    StatesGroup prop;
    load_state( state_propagat348, prop );
    ref_lio::Tap &t = ref_lio::tap();
    t.vm = v; t.reads = 0; t.iters.clear();
    v->lio_state_estimation( prop );
    t.vm = nullptr;
    return ( int ) t.iters.size();
}
int rl_iter( int k, double *HTH36, double *HTz6, double *sol18, double *G324, double *state24, double *cov324, int32_t *n_match, double *res_mean, int32_t *flags2 )
{
    ref_lio::Tap &t = ref_lio::tap();
    if ( k < 0 || k >= ( int ) t.iters.size() ) return -1;
    const ref_lio::IterRec &r = t.iters[ k ];
    if ( HTH36 ) std::memcpy( HTH36, r.HTH, sizeof( r.HTH ) );
    if ( HTz6 ) std::memcpy( HTz6, r.HTz, sizeof( r.HTz ) );
    if ( sol18 ) std::memcpy( sol18, r.sol, sizeof( r.sol ) );
    if ( G324 ) std::memcpy( G324, r.G, sizeof( r.G ) );
    if ( state24 ) std::memcpy( state24, r.state, sizeof( r.state ) );
    if ( cov324 ) std::memcpy( cov324, r.cov, sizeof( r.cov ) );
    if ( n_match ) *n_match = r.n_match;
    if ( res_mean ) *res_mean = r.res_mean;
    if ( flags2 ) { flags2[ 0 ] = r.converged; flags2[ 1 ] = r.stop; }
    return 0;
}
// m_laserCloudOri / m_corr_normvect of the last iteration: body point + sqrt(R_inv) (intensity, :1558), normal + residual (intensity, :1388)
int rl_last_matches( void *p, float *eff_pts_body_xyzi, float *eff_norm_dis, int cap )
{
    Voxel_mapping *v = ( Voxel_mapping * ) p;
    const int M = v->m_effct_feat_num;
    if ( M > cap ) return M;
    for ( int i = 0; i < M; i++ )
    {
        const PointType &a = v->m_laserCloudOri->points[ i ], &b = v->m_corr_normvect->points[ i ];
        if ( eff_pts_body_xyzi ) { eff_pts_body_xyzi[ i * 4 + 0 ] = a.x; eff_pts_body_xyzi[ i * 4 + 1 ] = a.y; eff_pts_body_xyzi[ i * 4 + 2 ] = a.z; eff_pts_body_xyzi[ i * 4 + 3 ] = a.intensity; }
        if ( eff_norm_dis ) { eff_norm_dis[ i * 4 + 0 ] = b.x; eff_norm_dis[ i * 4 + 1 ] = b.y; eff_norm_dis[ i * 4 + 2 ] = b.z; eff_norm_dis[ i * 4 + 3 ] = b.intensity; }
    }
    return M;
}
// map_incremental_grow (ImMesh_mesh_reconstruction.cpp:377) at the current state: needs the m_cross_mat_list / m_body_cov_list the newest rl_lio left
// (as in the reference) and m_feats_undistort = the raw scan; hands back the world-frame full scan it queued for the mesher (:413-416)
int rl_grow( void *p, const float *pts_raw_xyzi, int n_raw, float *world_full_xyzi )
{
    Voxel_mapping *v = ( Voxel_mapping * ) p;
    fill_cloud( v->m_feats_undistort, pts_raw_xyzi, n_raw, 4 );
    g_rec_mesh_data_package_list.clear();
    v->map_incremental_grow();
    if ( g_rec_mesh_data_package_list.size() != 1 ) return -1;
    const Rec_mesh_data_package &pk = g_rec_mesh_data_package_list.back();
    if ( ( int ) pk.m_frame_pts->size() != n_raw ) return -2;
    if ( world_full_xyzi )
        for ( int i = 0; i < n_raw; i++ )
        {
            const pcl::PointXYZI &q = pk.m_frame_pts->points[ i ];
            world_full_xyzi[ i * 4 + 0 ] = q.x; world_full_xyzi[ i * 4 + 1 ] = q.y; world_full_xyzi[ i * 4 + 2 ] = q.z; world_full_xyzi[ i * 4 + 3 ] = q.intensity;
        }
    return 0;
}
int64_t rl_dump( void *p, immesh_plane_rec *out, int64_t cap )
{
    Voxel_mapping *v = ( Voxel_mapping * ) p;
    int64_t cnt = 0;
    for ( const auto &kv : v->m_feat_map ) dump_node( kv.first, kv.second, 0, 0, out, cap, cnt );
    return cnt;
}
int64_t rl_root_voxels( void *p ) { return ( int64_t )( ( Voxel_mapping * ) p )->m_feat_map.size(); }
// StatesGroup's boxminus / boxplus as the reference writes them (common_lib.h:249-258, 260-273)
void rl_state_minus( const double *a348, const double *b348, double *out18 )
{
    StatesGroup a, b; load_state( a348, a ); load_state( b348, b );
    Eigen::Matrix< double, 18, 1 > d = a - b;
    for ( int i = 0; i < 18; i++ ) out18[ i ] = d[ i ];
}
void rl_state_plus( double *s348, const double *add18 )
{
    StatesGroup a; load_state( s348, a );
    Eigen::Matrix< double, 18, 1 > d; for ( int i = 0; i < 18; i++ ) d[ i ] = add18[ i ];
    a += d;
    store_full( a, s348 );
}
}  // extern "C"
