// drop_in/stubs/real_class -- what the REFERENCE'S OWN `class Voxel_mapping` (src/voxel_mapping.hpp:132-415, cut out by line range at build time:
// drop_in/Makefile target `realclass`) names besides the Eigen / PCL shapes of immesh_ref_shapes.hpp: ROS handles and messages, OpenCV, the ikd-Tree, the
// IMU / pre-processing types -- as empty or minimal SHAPES, so that both shims are compiled against the reference's member names and types instead of
// this repository's re-declaration of them (VERDICT r05 missing #5).  A compile check: nothing built from this header is run.
#pragma once
#include <condition_variable>
#include <cstdio>
#include <ctime>
#include <deque>
#include <fstream>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
using namespace std;      // include/common_lib.h:22-23: the class names mutex, string, vector, deque, shared_ptr, ofstream unqualified
using namespace Eigen;
#define ROOT_DIR ""       /* CMakeLists.txt: -DROOT_DIR */
#define MAXN ( 360000 )   /* src/voxel_mapping.hpp:60 */
#define LIDAR_SP_LEN ( 2 ) /* include/common_lib.h:38 */
typedef Eigen::Matrix3f M3F;
typedef Eigen::Vector3f V3F;
typedef std::vector< PointType > PointVector;                                   /* include/types.h */
struct BoxPointType { float vertex_min[ 3 ], vertex_max[ 3 ]; };               /* include/ikd-Tree/ikd_Tree.h:37-41 */
template < typename P > class KD_TREE {};                                      /* include/ikd-Tree/ikd_Tree.h:75 */
namespace pcl { template < typename P > class VoxelGrid {}; }
namespace cv { class Mat {}; }
namespace ros { class NodeHandle {}; class Publisher {}; }
namespace sensor_msgs { struct Imu { typedef std::shared_ptr< const Imu > ConstPtr; }; struct PointCloud2 { typedef std::shared_ptr< const PointCloud2 > ConstPtr; }; }
namespace livox_ros_driver { struct CustomMsg { typedef std::shared_ptr< const CustomMsg > ConstPtr; }; }
namespace geometry_msgs { struct Quaternion { double x = 0, y = 0, z = 0, w = 1; }; struct PoseStamped {}; }
namespace nav_msgs { struct Path {}; struct Odometry {}; }
struct LidarMeasureGroup {};                                                    /* include/common_lib.h:144-167 */
struct Preprocess { bool calib_laser = false; };                                /* src/preprocess.h:151-195: the one member the replaced bodies read (:170) */
class VOXEL_LOC { public: int64_t x = 0, y = 0, z = 0; bool operator==( const VOXEL_LOC &o ) const { return x == o.x && y == o.y && z == o.z; } };   /* src/voxel_loc.hpp:60-77 */
namespace std { template <> struct hash< VOXEL_LOC > { size_t operator()( const VOXEL_LOC &s ) const { return ( size_t ) ( ( s.z * 116101 + s.y ) * 116101 + s.x ); } }; }
class OctoTree;                                                                 /* src/voxel_loc.hpp:128 */
