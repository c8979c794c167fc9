// ORACLE / TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE'S OWN sensor decoders, compiled from where they lie under /root/reference behind
// ROS / PCL shaped stubs -- the recipe is oracle/Makefile (target ref -> _ref/libref_preprocess.so):
//   src/preprocess.cpp:139-232   Preprocess::avia_handler      (Livox CustomMsg -> pl_surf: line / point_filter_num / reflectivity / blind gates, offset_time -> curvature)
//   src/preprocess.cpp:497-528   Preprocess::velodyne_handler  (PointCloud2 of velodyne_ros::Point -> pl_surf: the elevation-angle / scanID gates)
// The excerpts are cut out by line range into _ref/pp_src/ at BUILD time (sed; the directory is removed after the compile) and #included below: nothing of
// the reference is copied into the repository.  `class Preprocess` below is a host class with the members those two bodies name (same names and types as
// src/preprocess.h:151-195); the message types are the shapes of livox_ros_driver::CustomMsg / sensor_msgs::PointCloud2 + pcl::fromROSMsg as far as the
// bodies touch them.  What is pinned: which points survive, in which order, with which float values -- against oracle/orc_imu.hpp's
// decode_livox / decode_velodyne, which the HIP decoders are compared with on the GPU.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>
#include <omp.h>
#include <pcl/common/io.h>      /* ref_voxelmap/stubs: pcl::PointXYZINormal, pcl::PointCloud */
typedef unsigned int uint;
typedef pcl::PointXYZINormal PointType;                 /* include/common_lib.h:58 */
typedef pcl::PointCloud< PointType > PointCloudXYZI;   /* include/common_lib.h:59; the stub's PointCloud has clear / reserve / resize / push_back / operator[] / size */
using std::vector;
struct orgtype { double range = 0, dista = 0; };      /* src/preprocess.h:84-101: the two members the (disabled) feature branch writes */
namespace livox_ros_driver {
struct CustomPoint { uint32_t offset_time; float x, y, z; uint8_t reflectivity, tag, line; };
struct CustomMsg { uint32_t point_num = 0; std::vector< CustomPoint > points; typedef std::shared_ptr< const CustomMsg > ConstPtr; };
}
namespace velodyne_ros { struct Point { float x, y, z, intensity, time; uint16_t ring; }; }   /* src/preprocess.h:29-38 */
namespace sensor_msgs { struct PointCloud2 { std::vector< velodyne_ros::Point > pts; typedef std::shared_ptr< const PointCloud2 > ConstPtr; }; }
namespace pcl {
inline void fromROSMsg( const sensor_msgs::PointCloud2 &m, PointCloud< velodyne_ros::Point > &c ) { c.points = m.pts; }
}
class Preprocess
{
  public:
    PointCloudXYZI    pl_full, pl_corn, pl_surf;
    PointCloudXYZI    pl_buff[ 128 ];
    vector< orgtype > typess[ 128 ];
    int               lidar_type = 0, point_filter_num = 1, N_SCANS = 6, time_unit = 0;
    double            blind = 0, blind_sqr = 0;
    bool              feature_enabled = false, given_offset_time = false, calib_laser = false;
    double            vx = 0, vy = 0, vz = 0;
    void avia_handler( const livox_ros_driver::CustomMsg::ConstPtr &msg );
    void velodyne_handler( const sensor_msgs::PointCloud2::ConstPtr &msg );
    void give_feature( pcl::PointCloud< PointType > &, vector< orgtype > & ) {}   /* feature extraction: off in every shipped configuration */
};
#include "pp_avia_handler.inc"       // preprocess.cpp:139-232
#include "pp_velodyne_handler.inc"   // preprocess.cpp:497-528

static int emit( const PointCloudXYZI &c, float *out5, int cap )
{
    const int n = ( int ) c.size();
    for ( int i = 0; i < n && i < cap; i++ ) { const PointType &p = c[ i ]; out5[ i * 5 ] = p.x; out5[ i * 5 + 1 ] = p.y; out5[ i * 5 + 2 ] = p.z; out5[ i * 5 + 3 ] = p.intensity; out5[ i * 5 + 4 ] = p.curvature; }
    return n;
}
extern "C" {
// wire: n x 19 bytes {u32 offset_time; f32 x, y, z; u8 reflectivity, tag, line} (the layout immesh_decode_livox takes); out5: x y z intensity curvature
int rp_avia( const uint8_t *wire, int n, int n_scans, int point_filter_num, double blind, float *out5, int cap )
{
    auto msg = std::make_shared< livox_ros_driver::CustomMsg >();
    msg->point_num = ( uint32_t ) n;
    msg->points.resize( ( size_t ) n );
    for ( int i = 0; i < n; i++ )
    {
        const uint8_t *p = wire + ( size_t ) i * 19;
        livox_ros_driver::CustomPoint &q = msg->points[ i ];
        std::memcpy( &q.offset_time, p, 4 ); std::memcpy( &q.x, p + 4, 4 ); std::memcpy( &q.y, p + 8, 4 ); std::memcpy( &q.z, p + 12, 4 );
        q.reflectivity = p[ 16 ]; q.tag = p[ 17 ]; q.line = p[ 18 ];
    }
    static Preprocess pp;
    pp.N_SCANS = n_scans; pp.point_filter_num = point_filter_num; pp.blind = blind; pp.blind_sqr = blind * blind; pp.feature_enabled = false;   /* Preprocess::set, preprocess.cpp:70-76 + :119 */
    pp.avia_handler( msg );
    return emit( pp.pl_surf, out5, cap );
}
int rp_velodyne( const uint8_t *data, int n, int step, int ox, int oy, int oz, int oi, int n_scans, float *out5, int cap )
{
    auto msg = std::make_shared< sensor_msgs::PointCloud2 >();
    msg->pts.resize( ( size_t ) n );
    for ( int i = 0; i < n; i++ )
    {
        const uint8_t *p = data + ( size_t ) i * step;
        velodyne_ros::Point &q = msg->pts[ i ];
        std::memcpy( &q.x, p + ox, 4 ); std::memcpy( &q.y, p + oy, 4 ); std::memcpy( &q.z, p + oz, 4 ); std::memcpy( &q.intensity, p + oi, 4 ); q.time = 0; q.ring = 0;
    }
    static Preprocess pp;
    pp.N_SCANS = n_scans;
    pp.velodyne_handler( msg );
    return emit( pp.pl_surf, out5, cap );
}
}  // extern "C"
