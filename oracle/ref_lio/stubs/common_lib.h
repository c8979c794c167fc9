// ORACLE / TEST INFRASTRUCTURE ONLY -- stands in for include/common_lib.h when lio_state_estimation / voxel_map_init / map_incremental_grow are
// compiled from where they lie (oracle/Makefile: _ref/libref_lio.so).  Unlike ref_voxelmap's stand-in this one carries the reference's OWN
// StatesGroup (include/common_lib.h:199-288, cut by line range at build time -> lio_states_group.inc) and the reference's OWN so3_math.h
// (Exp / Log / RotMtoEuler / SKEW_SYM_MATRX: symlinked into the include path, it only needs <Eigen/Core>); the macros below are common_lib.h's.
#pragma once
#include <Eigen/Core>
#include <pcl/common/io.h>
#include <so3_math.h>              /* the reference's (symlink in _ref/lio_src) */
#include <iostream>
#include <unordered_map>
using namespace std;               /* include/common_lib.h:22-23 */
using namespace Eigen;
#define DIM_STATE ( 18 )           /* include/common_lib.h:36 */
#define INIT_COV ( 0.0000001 )     /* :40 */
#define CALIB_ANGLE_COV ( 0.01 )   /* :41 */
#define NUM_MATCH_POINTS ( 5 )     /* :42 */
#define VEC_FROM_ARRAY( v ) v[ 0 ], v[ 1 ], v[ 2 ]   /* :45 */
#define HASH_P 116101              /* :52 */
#define MAX_N 10000000000          /* :53 */
typedef pcl::PointXYZINormal PointType;             /* include/types.h:7-18 */
typedef pcl::PointCloud< PointType > PointCloudXYZI;
typedef std::vector< PointType, Eigen::aligned_allocator< PointType > > PointVector;
typedef Eigen::Vector3d V3D;
typedef Eigen::Matrix3d M3D;
typedef Eigen::Vector3f V3F;
// include/types.h:21-22 -- MD / VD name the 18-state locals of lio_state_estimation (G, H_T_H, I_STATE, K_1, solution): here they are matrices that
// also enter themselves in a per-thread list while they live, so that the wrapper can read the locals of every iteration (see ref_lio_wrap.cpp)
namespace ref_lio {
struct Traced { int rows, cols; double* data; };
inline std::vector< Traced > &live() { static thread_local std::vector< Traced > v; return v; }
template < int R, int C > struct TracedMatrix : Eigen::Matrix< double, R, C >
{
    typedef Eigen::Matrix< double, R, C > Base;
    void enter() { live().push_back( Traced{ R, C, this->a } ); }
    TracedMatrix() { enter(); }
    TracedMatrix( const Base &b ) : Base( b ) { enter(); }
    TracedMatrix( const TracedMatrix &b ) : Base( b ) { enter(); }
    TracedMatrix &operator=( const Base &b ) { Base::operator=( b ); return *this; }
    TracedMatrix &operator=( const TracedMatrix &b ) { Base::operator=( b ); return *this; }
    ~TracedMatrix() { auto &v = live(); for ( size_t i = v.size(); i-- > 0; ) if ( v[ i ].data == this->a ) { v.erase( v.begin() + i ); break; } }
};
} // namespace ref_lio
#define MD( a, b ) ref_lio::TracedMatrix< ( a ), ( b ) >
#define VD( a ) ref_lio::TracedMatrix< ( a ), 1 >
extern M3D Eye3d;                  /* include/common_lib.h:55-58 */
extern V3D Zero3d;
#include "lio_states_group.inc"    /* include/common_lib.h:199-288: struct StatesGroup */
