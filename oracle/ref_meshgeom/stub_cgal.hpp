// ORACLE / TEST INFRASTRUCTURE ONLY -- CGAL-SHAPED classes for the reference's mesh code compiled from where it lies (oracle/Makefile:
// _ref/libref_meshgeom.so, _ref/libref_globalmap.so): Common_tools::Delaunay2 / D2_Point / Convex_hull_traits_2 (src/tools/tools_graphics.hpp:20-41) and
// CGAL::convex_hull_2 as far as delaunay_triangulation (mesh_rec_geometry.cpp:174-295) uses them.  The triangulation forwards to the oracle's
// restatement (orc_delaunay.hpp): what is pinned is the reference's code AROUND the CGAL call, not CGAL.
#pragma once
#include <algorithm>
#include <vector>
#include "orc_delaunay.hpp"   /* (-I oracle/) */

namespace Common_tools {
struct Timer { void tic(const char* = nullptr) {} double toc(const char* = nullptr, int = 0) { return 0.0; } };   // src/tools/tools_timer.hpp
struct D2_Point { double px, py; D2_Point(double x = 0, double y = 0) : px(x), py(y) {} double x() const { return px; } double y() const { return py; } };
// CGAL::Delaunay_triangulation_2<Simple_cartesian<double>, vertex info = unsigned> (src/tools/tools_graphics.hpp:20-41) as far as delaunay_triangulation uses it
class Delaunay2 {
  public:
    struct Vertex { D2_Point p; long inf; const D2_Point& point() const { return p; } long info() const { return inf; } };
    struct Face { const Vertex* v[3] = {nullptr, nullptr, nullptr}; const Vertex* vertex(int i) const { return v[i]; } };
    typedef std::vector<Face>::iterator Finite_faces_iterator;
    std::vector<Vertex> verts; std::vector<Face> faces;
    template <class It> void insert(It b, It e) {
        for (It it = b; it != e; ++it) verts.push_back(Vertex{it->first, (long)it->second});
        std::vector<double> xy(verts.size() * 2);
        for (size_t i = 0; i < verts.size(); i++) { xy[2 * i] = verts[i].p.x(); xy[2 * i + 1] = verts[i].p.y(); }
        orc::Delaunay2D dt;
        std::vector<int> f;
        dt.run(xy.data(), (int)verts.size(), f);
        for (size_t k = 0; k + 2 < f.size(); k += 3) { Face fc; fc.v[0] = &verts[f[k]]; fc.v[1] = &verts[f[k + 1]]; fc.v[2] = &verts[f[k + 2]]; faces.push_back(fc); }
    }
    size_t number_of_faces() const { return faces.size(); }
    Finite_faces_iterator finite_faces_begin() { return faces.begin(); }
    Finite_faces_iterator finite_faces_end() { return faces.end(); }
};
struct Convex_hull_traits_2 { const std::vector<D2_Point>* pts; explicit Convex_hull_traits_2(const std::vector<D2_Point>* p) : pts(p) {} };
}  // namespace Common_tools
namespace CGAL {
inline const std::vector<Common_tools::D2_Point>* make_property_map(const std::vector<Common_tools::D2_Point>& v) { return &v; }
// indices of the convex hull (monotone chain); its only consumers downstream are dead code
template <class It, class Out> void convex_hull_2(It b, It e, Out out, const Common_tools::Convex_hull_traits_2& tr) {
    std::vector<std::size_t> idx(b, e);
    const auto& P = *tr.pts;
    std::sort(idx.begin(), idx.end(), [&](std::size_t i, std::size_t j) { return P[i].x() != P[j].x() ? P[i].x() < P[j].x() : P[i].y() < P[j].y(); });
    auto cr = [&](std::size_t o, std::size_t a, std::size_t c) { return (P[a].x() - P[o].x()) * (P[c].y() - P[o].y()) - (P[a].y() - P[o].y()) * (P[c].x() - P[o].x()); };
    std::vector<std::size_t> h(2 * idx.size() + 2);
    int k = 0;
    for (std::size_t i = 0; i < idx.size(); i++) { while (k >= 2 && cr(h[k - 2], h[k - 1], idx[i]) <= 0) k--; h[k++] = idx[i]; }
    for (int i = (int)idx.size() - 2, t = k + 1; i >= 0; i--) { while (k >= t && cr(h[k - 2], h[k - 1], idx[i]) <= 0) k--; h[k++] = idx[i]; }
    for (int i = 0; i + 1 < k; i++) *out++ = h[i];
}
}  // namespace CGAL
