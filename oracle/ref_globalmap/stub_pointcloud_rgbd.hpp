// ORACLE / TEST INFRASTRUCTURE ONLY -- stands in for src/meshing/r3live/pointcloud_rgbd.hpp when the reference's mesher bodies are compiled from where
// they lie (oracle/Makefile: _ref/libref_globalmap.so).  The recipe symlinks it as `pointcloud_rgbd.hpp` next to the reference's own triangle.hpp /
// tools_kd_hash.hpp.  class RGB_pts and class RGB_Voxel are THE REFERENCE'S (pointcloud_rgbd.hpp:71-233, cut by line range at build time ->
// rgbd_classes.inc); struct Global_map below is a host struct with the members the compiled bodies name (same names and types as
// pointcloud_rgbd.hpp:234-298) minus the camera / image / thread members (OpenCV, Image_frame are not in this image).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <memory>
#include <mutex>
#include <numeric>
#include <set>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "tools_eigen.hpp"                 /* ref_globalmap/stubs */
#include <pcl/point_types.h>               /* ref_globalmap/stubs */
#include "ref_meshgeom/stub_cgal.hpp"      /* Common_tools::Timer / Delaunay2, CGAL::convex_hull_2 (-I oracle/) */
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
namespace boost { namespace serialization { class access; } }
namespace Common_tools { struct Triangle_2 {}; }   /* src/tools/tools_graphics.hpp: only as the element type of an unused member vector */
#include "tools_kd_hash.hpp"               /* the reference's (symlink) */
#include "triangle.hpp"                    /* the reference's (symlink) */
#include "ikd_Tree.h"                      /* the reference's (-I include/ikd-Tree) */
extern double g_initial_camera_exp_tim;    /* pointcloud_rgbd.hpp:66-69 */
extern double g_voxel_resolution;
#include "rgbd_classes.inc"                /* pointcloud_rgbd.hpp:71-233: RGB_pts, RGB_pt_ptr, RGB_Voxel, retrieve_pts_in_voxels, KDtree_pt(_vector) */
struct Global_map
{
    std::vector< RGB_pt_ptr >    m_rgb_pts_vec;
    std::vector< RGB_voxel_ptr > m_voxel_vec;
    std::shared_ptr< std::mutex > m_mutex_m_box_recent_hitted;
    KD_TREE< KDtree_pt >          m_kdtree;
    double                        m_recent_visited_voxel_activated_time = 0.0;
    bool                          m_in_appending_pts = 0;
    Hash_map_3d< long, RGB_pt_ptr >                    m_hashmap_3d_pts;
    Hash_map_3d< long, std::shared_ptr< RGB_Voxel > >  m_hashmap_voxels;
    std::unordered_set< std::shared_ptr< RGB_Voxel > > m_voxels_recent_visited;
    double                                             m_minimum_pts_size = 0.05;
    double                                             m_voxel_resolution = 0.1;
    void set_minimum_dis( double minimum_dis );
    void set_voxel_resolution( double minimum_dis );
    Global_map( int = 1 ) { m_mutex_m_box_recent_hitted = std::make_shared< std::mutex >(); }   /* pointcloud_rgbd.cpp:269-275 (the vectors' reserve( 1e9 ) is not copied) */
    vec_3 smooth_pts( RGB_pt_ptr &rgb_pt, double smooth_factor, double knn = 20, double maximum_smooth_dis = 0 );   /* pointcloud_rgbd.hpp:287 */
    template < typename T >
    int append_points_to_global_map( pcl::PointCloud< T > &pc_in, double added_time, std::vector< RGB_pt_ptr > *pts_added_vec = nullptr, int step = 1, int disable_append = 0 );
};
