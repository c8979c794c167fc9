// ORACLE / TEST INFRASTRUCTURE ONLY -- stands in for oneTBB (scheduling only): tbb::parallel_for_each runs the body sequentially, over the
// voxels in ASCENDING (x, y, z) KEY order -- the determinism rule of the checker (SURVEY 8(c) "determinism caveats": the reference iterates an
// unordered_set of shared_ptr in pointer-hash order on however many threads, and the order decides which smoothed positions a flip sees).
#pragma once
#include <algorithm>
#include <vector>
namespace tbb {
template < typename It, typename F > void parallel_for_each( It b, It e, F f )
{
    typedef typename std::iterator_traits< It >::value_type V;
    std::vector< V > v( b, e );
    std::sort( v.begin(), v.end(), []( const V &p, const V &q ) {
        if ( p->m_pos[ 0 ] != q->m_pos[ 0 ] ) return p->m_pos[ 0 ] < q->m_pos[ 0 ];
        if ( p->m_pos[ 1 ] != q->m_pos[ 1 ] ) return p->m_pos[ 1 ] < q->m_pos[ 1 ];
        return p->m_pos[ 2 ] < q->m_pos[ 2 ];
    } );
    for ( const V &x : v ) f( x );
}
// tbb::parallel_for over a blocked_range (save_to_ply_file, mesh_rec_geometry.cpp:81-98): one sequential pass over the whole range
template < typename T > struct blocked_range { T b_, e_; blocked_range( T b, T e, T = 1 ) : b_( b ), e_( e ) {} T begin() const { return b_; } T end() const { return e_; } };
template < typename R, typename F > void parallel_for( const R &r, F f ) { f( r ); }
} // namespace tbb
