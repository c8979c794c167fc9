// ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
// 2-D Delaunay triangulation standing in for CGAL::Delaunay_triangulation_2<Simple_cartesian<double>>
// (src/tools/tools_graphics.hpp:20-39, used at src/meshing/mesh_rec_geometry.cpp:253-287).  CGAL is a third-party
// dependency that is NOT under /root/reference (find_package(CGAL REQUIRED), CMakeLists.txt:59, version unpinned);
// this restates its published algorithm: incremental insertion (points pre-sorted along a space-filling curve,
// as CGAL's insert(range) does), point location by walking, Bowyer-Watson cavity via the in-circle predicate,
// an "infinite vertex" closing the hull.  Predicates are the plain-double Simple_cartesian formulas
// (SURVEY.md A.14): no filtering, no exact fallback.
// For points in general position the Delaunay triangulation is unique, so the face set equals CGAL's; exact
// co-circular/collinear ties are resolved differently (CGAL: symbolic perturbation) -- cross-checks against
// scipy.spatial.Delaunay (Qhull) use jittered data.
// DEGENERATE INPUT (round 5: regular lattices are tested, not avoided).  On cocircular / collinear points the plain-double determinants are
// rounding noise around zero: a far-away triangle may test in-disk, one next to p may not, the orientation tests of an edge p lies on may disagree
// from its two sides.  (CGAL with Simple_cartesian< double > is no better off -- the reference is not robust here; what the checker fixes is a
// DETERMINISTIC rule that keeps the triangulation usable and that the HIP path reproduces bit for bit.)  The rule, on the bare SET of live triangles:
//   G = every live triangle with in_disk(t, p); empty -> the point is not inserted.
//   If G's boundary (directed edges whose twin is not an edge of G) has |G| + 2 edges, G is one disk: cavity = G.
//   Otherwise cavity = the part of G that is edge-connected to the START SET: the finite triangles of G that contain p (all three orientations
//   >= 0); if there is none (p outside the hull) the ghost of G that sees p best (largest orient2d(a, b, p); ties: smallest sorted vertex
//   triple); if G has neither, its triangle with the smallest sorted vertex triple.
//   Cavity out, one new triangle (a, b, p) per boundary edge of the cavity in.
// The HIP path does exactly this at every insertion (it keeps nothing but the set).  The checker normally runs the usual linked algorithm -- walk to
// the containing triangle, flood over the neighbour links -- which gives the same cavity whenever every determinant it evaluated was decisively
// non-zero: then the flood's result C is the true cavity, anything else in G is a stray piece not adjacent to C, so G is either C or fails the
// |G| + 2 test, and the start set lies in C.  As soon as one evaluated determinant is within rounding of zero (|det| <= 1e-11 x the sum of the
// absolute values of its terms -- four orders of magnitude above the rounding error, so never on real, noisy scans) the checker leaves the linked
// mode for this point set and applies the rule above literally.
#pragma once
#include <vector>
#include <cstdint>
#include <algorithm>
#include <cmath>

namespace orc {

inline double orient2d(const double* p, const double* q, const double* r) {
    return (q[0] - p[0]) * (r[1] - p[1]) - (r[0] - p[0]) * (q[1] - p[1]);
}
// side_of_oriented_circle(p,q,r,t) > 0  <=> t inside circle through ccw p,q,r  (CGAL kernel_ftC2.h formula)
inline double incircle2d(const double* p, const double* q, const double* r, const double* t) {
    const double qpx = q[0] - p[0], qpy = q[1] - p[1], rpx = r[0] - p[0], rpy = r[1] - p[1], tpx = t[0] - p[0], tpy = t[1] - p[1];
    return (qpx * tpy - qpy * tpx) * (rpx * (r[0] - q[0]) + rpy * (r[1] - q[1])) - (tpx * (t[0] - q[0]) + tpy * (t[1] - q[1])) * (qpx * rpy - qpy * rpx);
}

struct Delaunay2D {
    static constexpr int INF = -1;  // the infinite vertex
    struct T { int v[3]; int n[3]; bool alive; };  // n[i] = neighbour across the edge opposite v[i]
    const double* xy = nullptr;
    std::vector<T> tris;
    std::vector<int> free_list;
    int last = 0;
    long n_skipped = 0;             // points of this run that were not inserted (no triangle's disk contains them: a duplicate in the projection / exact co-circularity)
    bool force_link_free = false;   // tests: every insertion by the link-free rule (what the HIP path does throughout) -- the result must not depend on it

    const double* P(int i) const { return xy + 2 * i; }
    mutable bool suspect = false;   // a determinant evaluated since the flag was cleared was within rounding of zero (see the header)
    double orient_s(const double* p, const double* q, const double* r) const {
        const double t1 = (q[0] - p[0]) * (r[1] - p[1]), t2 = (r[0] - p[0]) * (q[1] - p[1]);
        const double o = t1 - t2;
        if (std::fabs(o) <= 1e-11 * (std::fabs(t1) + std::fabs(t2))) suspect = true;
        return o;
    }

    // does the (possibly infinite) triangle's "circumdisk" contain point p ?
    bool in_disk(const T& t, const double* p) const {
        for (int i = 0; i < 3; i++)
            if (t.v[i] == INF) {  // ghost (a,b,INF) with a->b a hull edge seen from outside: half-plane test
                const int a = t.v[(i + 1) % 3], b = t.v[(i + 2) % 3];
                const double o = orient_s(P(a), P(b), p);
                if (o > 0) return true;
                if (o < 0) return false;
                // collinear with the hull edge: inside iff strictly between a and b
                const double* A = P(a); const double* B = P(b);
                const double d = (p[0] - A[0]) * (B[0] - A[0]) + (p[1] - A[1]) * (B[1] - A[1]);
                const double l = (B[0] - A[0]) * (B[0] - A[0]) + (B[1] - A[1]) * (B[1] - A[1]);
                return d > 0 && d < l;
            }
        {   // incircle2d with the magnitude of its two terms alongside (same expression, same bits)
            const double* a = P(t.v[0]); const double* q = P(t.v[1]); const double* r = P(t.v[2]);
            const double qpx = q[0] - a[0], qpy = q[1] - a[1], rpx = r[0] - a[0], rpy = r[1] - a[1], tpx = p[0] - a[0], tpy = p[1] - a[1];
            const double m1 = (qpx * tpy - qpy * tpx) * (rpx * (r[0] - q[0]) + rpy * (r[1] - q[1])), m2 = (tpx * (p[0] - q[0]) + tpy * (p[1] - q[1])) * (qpx * rpy - qpy * rpx);
            const double val = m1 - m2;
            const double mag = (std::fabs(qpx * tpy) + std::fabs(qpy * tpx)) * (std::fabs(rpx * (r[0] - q[0])) + std::fabs(rpy * (r[1] - q[1]))) +
                               (std::fabs(tpx * (p[0] - q[0])) + std::fabs(tpy * (p[1] - q[1]))) * (std::fabs(qpx * rpy) + std::fabs(qpy * rpx));
            if (std::fabs(val) <= 1e-11 * mag) suspect = true;
            return val > 0;
        }
    }
    int new_tri(int a, int b, int c) {
        int id;
        if (!free_list.empty()) { id = free_list.back(); free_list.pop_back(); }
        else { id = (int)tris.size(); tris.push_back(T()); }
        T& t = tris[id];
        t.v[0] = a; t.v[1] = b; t.v[2] = c; t.n[0] = t.n[1] = t.n[2] = -1; t.alive = true;
        return id;
    }
    // walk from `last` toward p; returns a triangle whose disk contains p (or -1 if p duplicates a vertex)
    int locate(const double* p) {
        int cur = last;
        if (!tris[cur].alive) { for (cur = 0; cur < (int)tris.size() && !tris[cur].alive; cur++) {} }
        // if we start in a ghost, step to its finite neighbour first
        for (int guard = 0; guard < (int)tris.size() * 4 + 16; guard++) {
            const T& t = tris[cur];
            int gi = -1;
            for (int i = 0; i < 3; i++) if (t.v[i] == INF) gi = i;
            if (gi >= 0) {
                if (in_disk(t, p)) return cur;
                cur = t.n[gi];  // the finite triangle across the hull edge
                continue;
            }
            bool moved = false;
            for (int i = 0; i < 3; i++) {
                const int a = t.v[(i + 1) % 3], b = t.v[(i + 2) % 3];
                if (orient_s(P(a), P(b), p) < 0) { cur = t.n[i]; moved = true; break; }
            }
            if (!moved) return cur;  // inside or on the boundary of a finite triangle
        }
        return -2;   // the walk did not settle (degenerate input): the caller chooses the start set by the canonical rule
    }
    // closed containment of p in a live triangle of G (see the header): ghosts count as containing
    bool contains(const T& t, const double* p) const {
        if (t.v[0] == INF || t.v[1] == INF || t.v[2] == INF) return true;
        return orient2d(P(t.v[0]), P(t.v[1]), p) >= 0 && orient2d(P(t.v[1]), P(t.v[2]), p) >= 0 && orient2d(P(t.v[2]), P(t.v[0]), p) >= 0;
    }
    static unsigned long long canon_key(const T& t) {   // sorted vertex triple, the infinite vertex last
        unsigned v[3];
        for (int i = 0; i < 3; i++) v[i] = t.v[i] == INF ? 0xFFFFu : (unsigned)t.v[i];
        std::sort(v, v + 3);
        return ((unsigned long long)v[0] << 32) | ((unsigned long long)v[1] << 16) | (unsigned long long)v[2];
    }

    // triangulate n points; out = finite faces as local index triples (ccw)
    void run(const double* xy_, int n, std::vector<int>& out) {
        xy = xy_; out.clear(); tris.clear(); free_list.clear(); n_skipped = 0;
        if (n < 3) return;
        // insertion order: Hilbert-like (Morton) order over the bounding box, ties by index
        double mn[2] = {xy[0], xy[1]}, mx[2] = {xy[0], xy[1]};
        for (int i = 1; i < n; i++) for (int k = 0; k < 2; k++) { mn[k] = std::min(mn[k], xy[2 * i + k]); mx[k] = std::max(mx[k], xy[2 * i + k]); }
        const double ext = std::max(mx[0] - mn[0], mx[1] - mn[1]);
        std::vector<std::pair<uint32_t, int>> order(n);
        for (int i = 0; i < n; i++) {
            uint32_t q[2];
            for (int k = 0; k < 2; k++) { double f = ext > 0 ? (xy[2 * i + k] - mn[k]) / ext : 0; q[k] = (uint32_t)std::min(65535.0, std::max(0.0, f * 65535.0)); }
            uint32_t code = 0;
            for (int b = 0; b < 16; b++) code |= ((q[0] >> b) & 1u) << (2 * b) | ((q[1] >> b) & 1u) << (2 * b + 1);
            order[i] = {code, i};
        }
        std::sort(order.begin(), order.end());
        // first non-degenerate triangle: take order[0], the next distinct point, the next non-collinear point
        int i0 = order[0].second, i1 = -1, i2 = -1;
        std::vector<char> used(n, 0);
        for (int k = 1; k < n && i1 < 0; k++) { int c = order[k].second; if (P(c)[0] != P(i0)[0] || P(c)[1] != P(i0)[1]) i1 = c; }
        if (i1 < 0) return;
        for (int k = 1; k < n && i2 < 0; k++) { int c = order[k].second; if (c != i1 && orient2d(P(i0), P(i1), P(c)) != 0) i2 = c; }
        if (i2 < 0) return;
        if (orient2d(P(i0), P(i1), P(i2)) < 0) std::swap(i1, i2);
        used[i0] = used[i1] = used[i2] = 1;
        const int t0 = new_tri(i0, i1, i2);
        // ghosts: for finite edge a->b (ccw interior on the left) the ghost is (b,a,INF)
        const int g0 = new_tri(i2, i1, INF), g1 = new_tri(i0, i2, INF), g2 = new_tri(i1, i0, INF);
        tris[t0].n[0] = g0; tris[t0].n[1] = g1; tris[t0].n[2] = g2;
        tris[g0].n[2] = t0; tris[g1].n[2] = t0; tris[g2].n[2] = t0;
        // ghost-ghost adjacency: ghost (b,a,INF): n[0] is across edge (a,INF), n[1] across (INF,b)
        auto link_ghosts = [&](int ga, int gb) {  // ga=(b,a,INF), gb=(c,b,INF) share vertex b:  ga.n[1] (opp a... ) set below
            (void)ga; (void)gb;
        };
        (void)link_ghosts;
        // g0=(i2,i1,INF): edge opposite v0=i2 is (i1,INF); opposite v1=i1 is (INF,i2)
        // neighbour across (i1,INF) is the ghost containing i1 besides g0 -> g2=(i1,i0,INF); across (INF,i2) -> g1=(i0,i2,INF)
        tris[g0].n[0] = g2; tris[g0].n[1] = g1;
        tris[g1].n[0] = g0; tris[g1].n[1] = g2;  // g1=(i0,i2,INF): opp i0 -> edge (i2,INF) shared with g0; opp i2 -> (INF,i0) shared with g2
        tris[g2].n[0] = g1; tris[g2].n[1] = g0;  // g2=(i1,i0,INF): opp i1 -> edge (i0,INF) shared with g1; opp i0 -> (INF,i1) shared with g0
        last = t0;

        std::vector<int> cavity, stack;
        std::vector<char> incav;
        struct BE { int a, b, outer; };  // boundary edge a->b (ccw around the cavity), triangle outside
        std::vector<BE> boundary;
        // LINKED mode (the normal case): walk to the containing triangle, flood over the neighbour links, fan, re-link.  It is left for good -- for the
        // rest of this point set -- the first time a determinant comes out within rounding of zero, the walk does not end in a triangle whose disk
        // holds p, or the cavity is not a disk; from then on every insertion follows the header's rule literally, on the bare set of triangles.
        bool linked = !force_link_free;
        auto edge_key = [](int x, int y) { return ((unsigned long long)(unsigned)(x + 1) << 32) | (unsigned long long)(unsigned)(y + 1); };
        for (int oi = 0; oi < n; oi++) {
            const int pi = order[oi].second;
            if (used[pi]) continue;
            const double* p = P(pi);
            if (linked) {
                suspect = false;
                const int seed = locate(p);
                if (seed >= 0 && in_disk(tris[seed], p)) {
                    cavity.clear(); stack.clear(); stack.push_back(seed);
                    incav.assign(tris.size(), 0);
                    incav[seed] = 1;
                    boundary.clear();
                    while (!stack.empty()) {
                        const int c = stack.back(); stack.pop_back();
                        cavity.push_back(c);
                        for (int i = 0; i < 3; i++) {
                            const int nb = tris[c].n[i];
                            if (incav[nb]) continue;
                            if (in_disk(tris[nb], p)) { incav[nb] = 1; stack.push_back(nb); }
                            else boundary.push_back(BE{tris[c].v[(i + 1) % 3], tris[c].v[(i + 2) % 3], nb});
                        }
                    }
                    // boundary edges found while a neighbour was not yet in the cavity may later be absorbed: filter
                    {
                        std::vector<BE> b2;
                        for (auto& e : boundary) if (!incav[e.outer]) b2.push_back(e);
                        boundary.swap(b2);
                    }
                    bool disk = !suspect && boundary.size() == cavity.size() + 2;
                    if (disk) {   // one simple loop: every vertex starts exactly one boundary edge
                        std::vector<int> starts;
                        for (auto& e : boundary) starts.push_back(e.a);
                        std::sort(starts.begin(), starts.end());
                        for (size_t k = 1; k < starts.size(); k++) if (starts[k] == starts[k - 1]) disk = false;
                    }
                    if (disk) {
                        for (int c : cavity) { tris[c].alive = false; free_list.push_back(c); }
                        // fan: new triangle (a,b,p) per boundary edge; link to outer and to each other
                        std::vector<int> created(boundary.size());
                        for (size_t k = 0; k < boundary.size(); k++) {
                            const BE& e = boundary[k];
                            const int nt = new_tri(e.a, e.b, pi);
                            created[k] = nt;
                            tris[nt].n[2] = e.outer;
                            T& o = tris[e.outer];
                            for (int i = 0; i < 3; i++) {  // outer's edge (b,a)
                                const int oa = o.v[(i + 1) % 3], ob = o.v[(i + 2) % 3];
                                if (oa == e.b && ob == e.a) o.n[i] = nt;
                            }
                        }
                        for (size_t k = 0; k < boundary.size(); k++)
                            for (size_t m = 0; m < boundary.size(); m++) {
                                if (boundary[k].b == boundary[m].a) {  // triangle k's edge (b,p) [opp a = n[0]] meets triangle m's edge (p,a) [opp b = n[1]]
                                    tris[created[k]].n[0] = created[m];
                                    tris[created[m]].n[1] = created[k];
                                }
                            }
                        used[pi] = 1;
                        last = created.empty() ? last : created[0];
                        continue;
                    }
                }
                linked = false;   // nothing has been changed for this point yet: it is inserted link-free below
            }
            // ---- link-free insertion on the set of live triangles
            std::vector<int> G;
            for (int i = 0; i < (int)tris.size(); i++) if (tris[i].alive && in_disk(tris[i], p)) G.push_back(i);
            if (G.empty()) { n_skipped++; continue; }      // duplicate of a vertex / on every circle: not inserted (counted: immesh_counters_t::n_degenerate_skips)
            auto boundary_of = [&](const std::vector<int>& S, std::vector<std::pair<int, int>>& bd) {
                std::vector<unsigned long long> ek;
                for (int t : S) for (int i = 0; i < 3; i++) ek.push_back(edge_key(tris[t].v[(i + 1) % 3], tris[t].v[(i + 2) % 3]));
                std::sort(ek.begin(), ek.end());
                bd.clear();
                for (int t : S) for (int i = 0; i < 3; i++) {
                    const int x = tris[t].v[(i + 1) % 3], y = tris[t].v[(i + 2) % 3];
                    if (!std::binary_search(ek.begin(), ek.end(), edge_key(y, x))) bd.push_back({x, y});
                }
            };
            std::vector<std::pair<int, int>> bd;
            boundary_of(G, bd);
            std::vector<int> cav = G;
            if (bd.size() != G.size() + 2) {
                std::vector<char> mem(G.size(), 0);
                bool any = false;
                for (size_t g = 0; g < G.size(); g++) {
                    const T& t = tris[G[g]];
                    if (t.v[0] != INF && t.v[1] != INF && t.v[2] != INF && contains(t, p)) { mem[g] = 1; any = true; }
                }
                if (!any) {   // p outside the hull: the ghost that sees it best; no ghost either: the smallest vertex triple
                    long best = -1; double best_o = 0;
                    for (size_t g = 0; g < G.size(); g++) {
                        const T& t = tris[G[g]];
                        int gi = -1;
                        for (int i = 0; i < 3; i++) if (t.v[i] == INF) gi = i;
                        if (gi < 0) continue;
                        const double o = orient2d(P(t.v[(gi + 1) % 3]), P(t.v[(gi + 2) % 3]), p);
                        if (best < 0 || o > best_o || (o == best_o && canon_key(t) < canon_key(tris[G[best]]))) { best = (long)g; best_o = o; }
                    }
                    if (best < 0) {
                        best = 0;
                        for (size_t g = 1; g < G.size(); g++) if (canon_key(tris[G[g]]) < canon_key(tris[G[best]])) best = (long)g;
                    }
                    mem[best] = 1;
                }
                for (bool changed = true; changed;) {
                    changed = false;
                    for (size_t g = 0; g < G.size(); g++) {
                        if (mem[g]) continue;
                        const T& t = tris[G[g]];
                        bool join = false;
                        for (size_t f = 0; f < G.size() && !join; f++) {
                            if (!mem[f]) continue;
                            const T& u = tris[G[f]];
                            for (int i = 0; i < 3 && !join; i++)
                                for (int j = 0; j < 3; j++)
                                    if (t.v[(i + 1) % 3] == u.v[(j + 2) % 3] && t.v[(i + 2) % 3] == u.v[(j + 1) % 3]) { join = true; break; }
                        }
                        if (join) { mem[g] = 1; changed = true; }
                    }
                }
                cav.clear();
                for (size_t g = 0; g < G.size(); g++) if (mem[g]) cav.push_back(G[g]);
                boundary_of(cav, bd);
            }
            for (int c : cav) { tris[c].alive = false; free_list.push_back(c); }
            for (auto& e : bd) new_tri(e.first, e.second, pi);
            used[pi] = 1;
        }
        for (const T& t : tris)
            if (t.alive && t.v[0] != INF && t.v[1] != INF && t.v[2] != INF) { out.push_back(t.v[0]); out.push_back(t.v[1]); out.push_back(t.v[2]); }
    }
};

}  // namespace orc
