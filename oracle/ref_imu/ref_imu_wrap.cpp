// ORACLE / TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE'S OWN IMU pre-processing, compiled from where it lies under /root/reference behind
// Eigen / PCL / ROS shaped stubs -- the recipe is oracle/Makefile (target ref -> _ref/libref_imu.so):
//   src/IMU_Processing.cpp:48         time_list (the offset-time comparator of the point sort)
//   src/IMU_Processing.cpp:486-553    ImuProcess::Forward_without_imu   (the constant-velocity prior of a LiDAR-only stream: state + covariance propagation)
//   src/IMU_Processing.cpp:755-958    ImuProcess::UndistortPcl          (forward propagation over the package's IMU samples, frame-end prediction, backward
//                                                                        compensation of every point into the scan-end frame)
//   include/common_lib.h:199-288, 302-319   struct StatesGroup, set_pose6d
//   include/so3_math.h                whole file (Exp( ang_vel, dt ), SKEW_SYM_MATRX), symlinked into the include path
// The excerpts are cut out by line range into _ref/imu_src/ at BUILD time (sed; the directory is removed after the compile) and #included below: nothing of
// the reference is copied into the repository.  `class ImuProcess` below is a host class with the members the two bodies name (same names and types as
// src/IMU_Processing.h:80-152); LidarMeasureGroup / MeasureGroup / sensor_msgs::Imu / Pose6D are shapes of what the bodies touch.
// What is pinned: the reference's logic -- which samples are integrated over which dt, the F_x / cov_w blocks, the frame-end prediction's
// branches, which pose compensates which point, the repeated compensation of the earliest point.  What is NOT: Eigen's arithmetic (stub products are plain
// k-ascending sums) and std::sort's order among EQUAL offset times (unstable in the reference; the tests use distinct stamps).
#include <Eigen/Core>              /* ref_voxelmap/stubs */
#include <pcl/common/io.h>         /* ref_voxelmap/stubs */
#include <so3_math.h>              /* the reference's (symlink in _ref/imu_src) */
#include <algorithm>
#include <cstring>
#include <deque>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <vector>
#include "../../include/immesh_c_api.h"
using namespace std;
using namespace Eigen;
#define DIM_STATE ( 18 )           /* include/common_lib.h:36 */
#define INIT_COV ( 0.0000001 )
#define G_m_s2 ( 9.81 )            /* :35 */
#define VEC_FROM_ARRAY( v ) v[ 0 ], v[ 1 ], v[ 2 ]                                              /* :45 */
#define MAT_FROM_ARRAY( v ) v[ 0 ], v[ 1 ], v[ 2 ], v[ 3 ], v[ 4 ], v[ 5 ], v[ 6 ], v[ 7 ], v[ 8 ]   /* :46 */
#ifndef MAX
#define MAX( a, b ) ( ( a ) > ( b ) ? ( a ) : ( b ) )
#endif
typedef pcl::PointXYZINormal PointType;
typedef pcl::PointCloud< PointType > PointCloudXYZI;
typedef Eigen::Vector3d V3D;
typedef Eigen::Matrix3d M3D;
#define MD( a, b ) Eigen::Matrix< double, ( a ), ( b ) >
#define VD( a ) Eigen::Matrix< double, ( a ), 1 >
M3D Eye3d = M3D::Identity();
V3D Zero3d( 0, 0, 0 );
struct Pose6D { double offset_time; double acc[ 3 ], gyr[ 3 ], vel[ 3 ], pos[ 3 ], rot[ 9 ]; };   /* msg/Pose6D.msg */
#include "imu_states_group.inc"    /* include/common_lib.h:199-288 */
#include "imu_set_pose6d.inc"      /* include/common_lib.h:302-319 */
namespace ros { struct Time { double t = 0; double toSec() const { return t; } }; }
namespace sensor_msgs {
struct Imu { struct { ros::Time stamp; } header; struct { double x = 0, y = 0, z = 0; } angular_velocity, linear_acceleration; typedef std::shared_ptr< const Imu > ConstPtr; };
typedef Imu::ConstPtr ImuConstPtr;
}
struct MeasureGroup { double img_offset_time = 0; deque< sensor_msgs::Imu::ConstPtr > imu; };          /* include/common_lib.h:130-142 */
struct LidarMeasureGroup                                                                            /* include/common_lib.h:144-167 */
{
    double lidar_beg_time = 0, last_update_time = 0;
    PointCloudXYZI::Ptr lidar;
    std::deque< MeasureGroup > measures;
    bool is_lidar_end = false;
    int  lidar_scan_index_now = 0;
};
class ImuProcess
{
  public:
    void UndistortPcl( LidarMeasureGroup &lidar_meas, StatesGroup &state_inout, PointCloudXYZI &pcl_out );
    void Forward_without_imu( LidarMeasureGroup &meas, StatesGroup &state_inout, PointCloudXYZI &pcl_out );
    ofstream fout_imu;
    V3D cov_acc, cov_gyr, cov_bias_gyr, cov_bias_acc;
    double first_lidar_time = 0;
    sensor_msgs::ImuConstPtr last_imu_;
    vector< Pose6D > IMUpose;
    M3D Lid_rot_to_IMU;
    V3D Lid_offset_to_IMU, mean_acc, angvel_last, acc_s_last;
    double last_lidar_end_time_ = 0, time_last_scan = 0;
    bool b_first_frame_ = true;
};
#include "imu_time_list.inc"             // IMU_Processing.cpp:48
#include "imu_forward_without_imu.inc"   // IMU_Processing.cpp:486-553
#include "imu_undistort_pcl.inc"         // IMU_Processing.cpp:755-958

static void load_state( const double *s, StatesGroup &g )
{
    for ( int r = 0; r < 3; r++ ) for ( int c = 0; c < 3; c++ ) g.rot_end( r, c ) = s[ r * 3 + c ];
    for ( int i = 0; i < 3; i++ ) { g.pos_end( i ) = s[ 9 + i ]; g.vel_end( i ) = s[ 12 + i ]; g.bias_g( i ) = s[ 15 + i ]; g.bias_a( i ) = s[ 18 + i ]; g.gravity( i ) = s[ 21 + i ]; }
    for ( int r = 0; r < 18; r++ ) for ( int c = 0; c < 18; c++ ) g.cov( r, c ) = s[ 24 + r * 18 + c ];
}
static void store_state( const StatesGroup &g, double *s )
{
    for ( int r = 0; r < 3; r++ ) for ( int c = 0; c < 3; c++ ) s[ r * 3 + c ] = g.rot_end( r, c );
    for ( int i = 0; i < 3; i++ ) { s[ 9 + i ] = g.pos_end( i ); s[ 12 + i ] = g.vel_end( i ); s[ 15 + i ] = g.bias_g( i ); s[ 18 + i ] = g.bias_a( i ); s[ 21 + i ] = g.gravity( i ); }
    for ( int r = 0; r < 18; r++ ) for ( int c = 0; c < 18; c++ ) s[ 24 + r * 18 + c ] = g.cov( r, c );
}
static sensor_msgs::Imu::ConstPtr mk_imu( const immesh_imu_sample &q )
{
    auto m = std::make_shared< sensor_msgs::Imu >();
    m->header.stamp.t = q.t;
    m->angular_velocity.x = q.gyr[ 0 ]; m->angular_velocity.y = q.gyr[ 1 ]; m->angular_velocity.z = q.gyr[ 2 ];
    m->linear_acceleration.x = q.acc[ 0 ]; m->linear_acceleration.y = q.acc[ 1 ]; m->linear_acceleration.z = q.acc[ 2 ];
    return m;
}
extern "C" {
// Forward_without_imu over dt = lidar_beg_time - time_last_scan (first_frame != 0: the reference's 0.1 s); state348 in / out
int ri_forward_without_imu( double *state348, double dt, int first_frame, const double *cov_gyr3, const double *cov_acc3 )
{
    ImuProcess ip;
    for ( int i = 0; i < 3; i++ ) { ip.cov_gyr( i ) = cov_gyr3[ i ]; ip.cov_acc( i ) = cov_acc3[ i ]; }
    ip.b_first_frame_ = first_frame != 0; ip.time_last_scan = 100.0;
    LidarMeasureGroup meas;
    meas.lidar_beg_time = 100.0 + dt;
    meas.lidar = PointCloudXYZI::Ptr( new PointCloudXYZI() );
    PointType p; p.x = 1; p.y = 2; p.z = 3; p.curvature = 50.0f;
    meas.lidar->points.push_back( p );
    StatesGroup st; load_state( state348, st );
    PointCloudXYZI out;
    ip.Forward_without_imu( meas, st, out );
    store_state( st, state348 );
    return ( int ) out.points.size();
}
// UndistortPcl on one LiDAR-only package (is_lidar_end == true): the interface of immesh_undistort / orc_undistort
int ri_undistort( const float *pts_xyzit, int n, const immesh_imu_sample *imu, int n_imu, double lidar_beg_time, double *last_update_time, immesh_imu_ctx *ic, double *state348, float *out_xyzi )
{
    ImuProcess ip;
    for ( int i = 0; i < 3; i++ )
    {
        ip.cov_gyr( i ) = ic->cov_gyr[ i ]; ip.cov_acc( i ) = ic->cov_acc[ i ]; ip.cov_bias_gyr( i ) = ic->cov_bias_gyr[ i ]; ip.cov_bias_acc( i ) = ic->cov_bias_acc[ i ];
        ip.acc_s_last( i ) = ic->acc_s_last[ i ]; ip.angvel_last( i ) = ic->angvel_last[ i ]; ip.Lid_offset_to_IMU( i ) = ic->lid_offset_to_imu[ i ];
        for ( int j = 0; j < 3; j++ ) ip.Lid_rot_to_IMU( i, j ) = ic->lid_rot_to_imu[ i * 3 + j ];
    }
    ip.mean_acc = V3D( ic->mean_acc_norm, 0, 0 );                 /* only its norm is read (:825) */
    ip.last_imu_ = mk_imu( ic->last_imu );
    ip.last_lidar_end_time_ = ic->last_lidar_end_time;
    LidarMeasureGroup meas;
    meas.lidar_beg_time = lidar_beg_time; meas.last_update_time = *last_update_time; meas.is_lidar_end = true;
    meas.lidar = PointCloudXYZI::Ptr( new PointCloudXYZI() );
    for ( int i = 0; i < n; i++ ) { PointType p; p.x = pts_xyzit[ i * 5 ]; p.y = pts_xyzit[ i * 5 + 1 ]; p.z = pts_xyzit[ i * 5 + 2 ]; p.intensity = pts_xyzit[ i * 5 + 3 ]; p.curvature = pts_xyzit[ i * 5 + 4 ]; meas.lidar->points.push_back( p ); }
    MeasureGroup mg;
    for ( int i = 0; i < n_imu; i++ ) mg.imu.push_back( mk_imu( imu[ i ] ) );
    meas.measures.push_back( mg );
    StatesGroup st; load_state( state348, st );
    PointCloudXYZI out;
    ip.UndistortPcl( meas, st, out );
    store_state( st, state348 );
    *last_update_time = meas.last_update_time;
    ic->last_lidar_end_time = ip.last_lidar_end_time_;
    for ( int i = 0; i < 3; i++ ) { ic->acc_s_last[ i ] = ip.acc_s_last( i ); ic->angvel_last[ i ] = ip.angvel_last( i ); }
    ic->last_imu.t = ip.last_imu_->header.stamp.toSec();
    ic->last_imu.gyr[ 0 ] = ip.last_imu_->angular_velocity.x; ic->last_imu.gyr[ 1 ] = ip.last_imu_->angular_velocity.y; ic->last_imu.gyr[ 2 ] = ip.last_imu_->angular_velocity.z;
    ic->last_imu.acc[ 0 ] = ip.last_imu_->linear_acceleration.x; ic->last_imu.acc[ 1 ] = ip.last_imu_->linear_acceleration.y; ic->last_imu.acc[ 2 ] = ip.last_imu_->linear_acceleration.z;
    for ( size_t i = 0; i < out.points.size() && ( int ) i < n; i++ ) { const PointType &p = out.points[ i ]; out_xyzi[ i * 4 ] = p.x; out_xyzi[ i * 4 + 1 ] = p.y; out_xyzi[ i * 4 + 2 ] = p.z; out_xyzi[ i * 4 + 3 ] = p.intensity; }
    return ( int ) out.points.size();
}
}  // extern "C"
