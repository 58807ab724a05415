// Host-side mesh set-up (include/tdgl_host_mesh.h): Delaunay triangulation and Voronoi dual mesh.
// Plain C++17, no HIP; built into libtdgl_mesh.so with -ffp-contract=off (the predicates' floating-point
// filters and the bit-for-bit agreement of the dual mesh with its NumPy predecessor both need every
// product and sum rounded on its own).
//
// Delaunay: points are inserted in order of distance from the circumcentre of a seed triangle, so every
// new point lies outside the convex hull of the points before it; it is connected to the hull edges it
// sees, and edges that fail the in-circle test are flipped (Lawson).  (The order is computed in floating
// point: a point that turns out to lie inside the hull or on it is located by a walk and inserted into its
// triangle or onto its edge; only a point that coincides with another one is left out, and reported.)  The hull is a doubly linked ring
// with an angular hash to find a visible edge in O(1).  Orientation and in-circle signs are exact: a
// forward error bound decides almost always (Shewchuk's stage-A bounds), otherwise the determinant is
// evaluated in exact expansion arithmetic.
#include "tdgl_host_mesh.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// exact arithmetic on expansions (sums of non-overlapping doubles, increasing magnitude)
// ---------------------------------------------------------------------------------------------
using Expansion = std::vector<double>;

inline void two_sum(double a, double b, double &s, double &e) {
    s = a + b;
    const double bv = s - a;
    e = (a - (s - bv)) + (b - bv);
}
inline void quick_two_sum(double a, double b, double &s, double &e) {  // |a| >= |b|
    s = a + b;
    e = b - (s - a);
}
inline void two_prod(double a, double b, double &p, double &e) {
    p = a * b;
    e = std::fma(a, b, -p);
}

// e + b
Expansion grow(const Expansion &e, double b) {
    Expansion h;
    h.reserve(e.size() + 1);
    double q = b;
    for (double ei : e) {
        double s, err;
        two_sum(q, ei, s, err);
        if (err != 0.0) h.push_back(err);
        q = s;
    }
    if (q != 0.0 || h.empty()) h.push_back(q);
    return h;
}
Expansion add(const Expansion &e, const Expansion &f) {
    Expansion h = e;
    for (double fi : f) h = grow(h, fi);
    return h;
}
Expansion negate(Expansion e) {
    for (double &v : e) v = -v;
    return e;
}
// e * b
Expansion scale(const Expansion &e, double b) {
    Expansion h;
    if (e.empty() || b == 0.0) return Expansion{0.0};
    h.reserve(2 * e.size());
    double q, lo;
    two_prod(e[0], b, q, lo);
    if (lo != 0.0) h.push_back(lo);
    for (size_t i = 1; i < e.size(); ++i) {
        double t, tl, s, err;
        two_prod(e[i], b, t, tl);
        two_sum(q, tl, s, err);
        if (err != 0.0) h.push_back(err);
        quick_two_sum(t, s, q, err);
        if (err != 0.0) h.push_back(err);
    }
    if (q != 0.0 || h.empty()) h.push_back(q);
    return h;
}
Expansion mul(const Expansion &e, const Expansion &f) {
    Expansion h{0.0};
    for (double fi : f) h = add(h, scale(e, fi));
    return h;
}
Expansion diff(double a, double b) {  // a - b, exactly
    double s, e;
    two_sum(a, -b, s, e);
    Expansion h;
    if (e != 0.0) h.push_back(e);
    h.push_back(s);
    return h;
}
inline double sign_of(const Expansion &e) {  // the largest component carries the sign
    for (size_t i = e.size(); i-- > 0;)
        if (e[i] != 0.0) return e[i];
    return 0.0;
}

constexpr double EPS = 1.1102230246251565e-16;  // 2^-53
constexpr double ORIENT_BOUND = (3.0 + 16.0 * EPS) * EPS;
constexpr double INCIRCLE_BOUND = (10.0 + 96.0 * EPS) * EPS;

double orient_exact(const double *a, const double *b, const double *c) {
    const Expansion acx = diff(a[0], c[0]), acy = diff(a[1], c[1]);
    const Expansion bcx = diff(b[0], c[0]), bcy = diff(b[1], c[1]);
    return sign_of(add(mul(acx, bcy), negate(mul(acy, bcx))));
}

// > 0: a, b, c counter-clockwise; < 0: clockwise; 0: collinear.  Sign exact.
inline double orient(const double *a, const double *b, const double *c) {
    const double l = (a[0] - c[0]) * (b[1] - c[1]);
    const double r = (a[1] - c[1]) * (b[0] - c[0]);
    const double det = l - r;
    double sum;
    if (l > 0.0) {
        if (r <= 0.0) return det;
        sum = l + r;
    } else if (l < 0.0) {
        if (r >= 0.0) return det;
        sum = -l - r;
    } else {
        return det;
    }
    const double bound = ORIENT_BOUND * sum;
    if (det >= bound || -det >= bound) return det;
    return orient_exact(a, b, c);
}

double incircle_exact(const double *a, const double *b, const double *c, const double *d) {
    const Expansion adx = diff(a[0], d[0]), ady = diff(a[1], d[1]);
    const Expansion bdx = diff(b[0], d[0]), bdy = diff(b[1], d[1]);
    const Expansion cdx = diff(c[0], d[0]), cdy = diff(c[1], d[1]);
    const Expansion al = add(mul(adx, adx), mul(ady, ady));
    const Expansion bl = add(mul(bdx, bdx), mul(bdy, bdy));
    const Expansion cl = add(mul(cdx, cdx), mul(cdy, cdy));
    const Expansion bc = add(mul(bdx, cdy), negate(mul(cdx, bdy)));
    const Expansion ca = add(mul(cdx, ady), negate(mul(adx, cdy)));
    const Expansion ab = add(mul(adx, bdy), negate(mul(bdx, ady)));
    return sign_of(add(add(mul(al, bc), mul(bl, ca)), mul(cl, ab)));
}

// > 0: d strictly inside the circle through a, b, c (counter-clockwise); < 0 outside; 0 on it.  Sign exact.
inline double incircle(const double *a, const double *b, const double *c, const double *d) {
    const double adx = a[0] - d[0], ady = a[1] - d[1];
    const double bdx = b[0] - d[0], bdy = b[1] - d[1];
    const double cdx = c[0] - d[0], cdy = c[1] - d[1];
    const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy;
    const double cdxady = cdx * ady, adxcdy = adx * cdy;
    const double adxbdy = adx * bdy, bdxady = bdx * ady;
    const double al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
    const double det = al * (bdxcdy - cdxbdy) + bl * (cdxady - adxcdy) + cl * (adxbdy - bdxady);
    const double permanent = (std::fabs(bdxcdy) + std::fabs(cdxbdy)) * al + (std::fabs(cdxady) + std::fabs(adxcdy)) * bl +
                             (std::fabs(adxbdy) + std::fabs(bdxady)) * cl;
    const double bound = INCIRCLE_BOUND * permanent;
    if (det > bound || -det > bound) return det;
    return incircle_exact(a, b, c, d);
}

// ---------------------------------------------------------------------------------------------
// sweep-hull Delaunay
// ---------------------------------------------------------------------------------------------
struct Triangulator {
    const double *xy;
    int n;
    std::vector<int> tri;   // 3 t: start vertex of every half-edge; half-edge h runs tri[h] -> tri[next(h)]
    std::vector<int> twin;  // 3 t: the opposite half-edge, -1 on the hull
    std::vector<int> hull_next, hull_prev, hull_edge, hash;  // hull ring; hull_edge[v]: the half-edge v -> hull_next[v]
    std::vector<int> stack;
    double cx = 0.0, cy = 0.0;
    int hash_size = 0;

    const double *pt(int i) const { return xy + 2 * (size_t)i; }
    static int next(int h) { return h % 3 == 2 ? h - 2 : h + 1; }
    static int prev(int h) { return h % 3 == 0 ? h + 2 : h - 1; }

    int hash_key(const double *p) const {
        // monotone in the angle of p around the sweep centre, without atan2
        const double dx = p[0] - cx, dy = p[1] - cy;
        const double s = std::fabs(dx) + std::fabs(dy);
        const double t = s > 0.0 ? dx / s : 0.0;           // [-1, 1]
        const double a = (dy > 0.0 ? 3.0 - t : 1.0 + t) / 4.0;  // [0, 1]
        int k = (int)std::floor(a * hash_size);
        return k >= hash_size ? hash_size - 1 : (k < 0 ? 0 : k);
    }
    void link(int a, int b) {
        twin[a] = b;
        if (b >= 0) twin[b] = a;
        else hull_edge[tri[a]] = a;  // a hull edge changed its slot: the ring entry of its start vertex follows
    }
    int add_triangle(int a, int b, int c, int ta, int tb, int tc) {
        const int h = (int)tri.size();
        tri.push_back(a);
        tri.push_back(b);
        tri.push_back(c);
        twin.push_back(-1);
        twin.push_back(-1);
        twin.push_back(-1);
        link(h, ta);
        link(h + 1, tb);
        link(h + 2, tc);
        return h;
    }
    // Lawson flips starting at half-edge `h`.  During insertion the vertex opposite `h` in its own triangle is the
    // new point and only the two edges that end up opposite it need another look; `all_sides` re-examines all four
    // edges around a flipped pair (the closing sweep).  Returns the number of flips.
    int64_t legalize(int h, bool all_sides = false) {
        int64_t flips = 0;
        stack.clear();
        stack.push_back(h);
        while (!stack.empty()) {
            const int a = stack.back();
            stack.pop_back();
            const int b = twin[a];
            if (b < 0) continue;
            const int a1 = next(a), a2 = prev(a), b1 = next(b), b2 = prev(b);
            const int u = tri[a], v = tri[a1], w = tri[a2], x = tri[b2];
            if (!(incircle(pt(u), pt(v), pt(w), pt(x)) > 0.0)) continue;
            // replace edge u-v by x-w: triangle of a becomes (x, v, w), triangle of b becomes (w, u, x)
            const int ta2 = twin[a2], tb2 = twin[b2];
            tri[a] = x;
            tri[b] = w;
            link(a, tb2);
            link(b, ta2);
            link(a2, b2);
            // the edges now opposite the new point w: x -> v (slot a) and u -> x (slot b1)
            ++flips;
            stack.push_back(a);
            stack.push_back(b1);
            if (all_sides) {
                stack.push_back(a1);
                stack.push_back(b);
            }
        }
        return flips;
    }

    int new_slots(int a, int b, int c) {  // three half-edges a -> b -> c -> a without twins yet
        const int h = (int)tri.size();
        tri.push_back(a);
        tri.push_back(b);
        tri.push_back(c);
        twin.push_back(-1);
        twin.push_back(-1);
        twin.push_back(-1);
        return h;
    }
    // The triangle that contains p (p sees no hull edge, so it is inside the hull or on its boundary): a walk from
    // half-edge `h`, always across an edge that has p strictly on its far side; in a Delaunay triangulation such a
    // walk ends.  Returns a half-edge of the triangle, or -1 (cannot happen with exact predicates).
    int locate(const double *p, int h) const {
        const int64_t limit = 2 * (int64_t)tri.size() + 16;
        for (int64_t step = 0; step < limit; ++step) {
            const int t = h - h % 3;
            int cross = -1;
            for (int k = 0; k < 3 && cross < 0; ++k) {
                const int e = t + k;
                if (orient(pt(tri[e]), pt(tri[next(e)]), p) < 0.0) cross = e;
            }
            if (cross < 0) return t;
            if (twin[cross] < 0) return -1;
            h = twin[cross];
        }
        for (int t = 0; t < (int)tri.size(); t += 3) {  // (a walk that does not end: look at every triangle)
            bool in = true;
            for (int k = 0; k < 3 && in; ++k) in = !(orient(pt(tri[t + k]), pt(tri[next(t + k)]), p) < 0.0);
            if (in) return t;
        }
        return -1;
    }
    // Insert point i, which sees no hull edge.  0: inserted; 1: it coincides with a point of the triangulation
    // (left out); -1: not located.
    int insert_inside(int i, int start_edge) {
        const double *p = pt(i);
        const int t = locate(p, start_edge);
        if (t < 0) return -1;
        int zero_edge = -1, zeros = 0;
        for (int k = 0; k < 3; ++k)
            if (orient(pt(tri[t + k]), pt(tri[next(t + k)]), p) == 0.0) zero_edge = t + k, ++zeros;
        if (zeros >= 2) return 1;  // on two edge lines of its triangle: a vertex
        if (zeros == 0) {
            // (a, b, c) -> (a, b, p), (b, c, p), (c, a, p)
            const int h0 = t, h1 = t + 1, h2 = t + 2;
            const int b = tri[h1], c = tri[h2], a = tri[h0];
            const int t1 = twin[h1], t2 = twin[h2];
            tri[h2] = i;
            const int g = new_slots(b, c, i), k = new_slots(c, a, i);
            link(g, t1);
            link(k, t2);
            twin[h1] = g + 2, twin[g + 2] = h1;  // b -> p | p -> b
            twin[g + 1] = k + 2, twin[k + 2] = g + 1;  // c -> p | p -> c
            twin[h2] = k + 1, twin[k + 1] = h2;  // p -> a | a -> p
            legalize(h0);
            legalize(g);
            legalize(k);
            return 0;
        }
        // on the edge e = u -> v of (u, v, w): (u, p, w) + (p, v, w); the triangle across, (v, u, x), likewise
        const int e = zero_edge, e1 = next(e), e2 = prev(e);
        const int v = tri[e1], w = tri[e2], u = tri[e];
        const int f = twin[e], t_vw = twin[e1];
        tri[e1] = i;                           // e: u -> p, e1: p -> w, e2: w -> u
        const int nn = new_slots(i, v, w);     // p -> v, v -> w, w -> p
        link(nn + 1, t_vw);
        twin[e1] = nn + 2, twin[nn + 2] = e1;
        if (f >= 0) {
            const int f1 = next(f), f2 = prev(f);
            const int x = tri[f2];
            const int t_ux = twin[f1];
            tri[f1] = i;                        // f: v -> p, f1: p -> x, f2: x -> v
            const int m = new_slots(i, u, x);   // p -> u, u -> x, x -> p
            link(m + 1, t_ux);
            twin[f1] = m + 2, twin[m + 2] = f1;
            twin[e] = m, twin[m] = e;           // u -> p | p -> u
            twin[nn] = f, twin[f] = nn;         // p -> v | v -> p
            legalize(e2);
            legalize(nn + 1);
            legalize(f2);
            legalize(m + 1);
        } else {
            // a hull edge: p joins the ring between u and v
            link(e, -1);   // hull_edge[u] = e
            link(nn, -1);  // hull_edge[p] = nn
            hull_next[u] = i;
            hull_prev[i] = u;
            hull_next[i] = v;
            hull_prev[v] = i;
            hash[hash_key(p)] = i;
            legalize(e2);
            legalize(nn + 1);
        }
        return 0;
    }

    int run(int64_t *out, int64_t *n_out) {
        // bounding box, seed triangle
        double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 2; ++k) {
                const double v = xy[2 * (size_t)i + k];
                if (!std::isfinite(v)) return TDGL_MESH_ERR_ARG;
                lo[k] = std::min(lo[k], v);
                hi[k] = std::max(hi[k], v);
            }
        const double mx = 0.5 * (lo[0] + hi[0]), my = 0.5 * (lo[1] + hi[1]);
        auto dist2 = [&](const double *p, double x, double y) { return (p[0] - x) * (p[0] - x) + (p[1] - y) * (p[1] - y); };
        int i0 = 0;
        double best = INFINITY;
        for (int i = 0; i < n; ++i) {
            const double d = dist2(pt(i), mx, my);
            if (d < best) best = d, i0 = i;
        }
        int i1 = -1;
        best = INFINITY;
        for (int i = 0; i < n; ++i) {
            const double d = dist2(pt(i), pt(i0)[0], pt(i0)[1]);
            if (i != i0 && d > 0.0 && d < best) best = d, i1 = i;
        }
        if (i1 < 0) return TDGL_MESH_ERR_DEGENERATE;
        auto circumradius2 = [&](const double *a, const double *b, const double *c) {
            const double bx = b[0] - a[0], by = b[1] - a[1], ex = c[0] - a[0], ey = c[1] - a[1];
            const double bl = bx * bx + by * by, el = ex * ex + ey * ey, d = bx * ey - by * ex;
            if (d == 0.0) return (double)INFINITY;
            const double x = (ey * bl - by * el) * (0.5 / d), y = (bx * el - ex * bl) * (0.5 / d);
            return x * x + y * y;
        };
        int i2 = -1;
        best = INFINITY;
        for (int i = 0; i < n; ++i) {
            if (i == i0 || i == i1) continue;
            const double r = circumradius2(pt(i0), pt(i1), pt(i));
            if (r < best && orient(pt(i0), pt(i1), pt(i)) != 0.0) best = r, i2 = i;
        }
        if (i2 < 0 || !std::isfinite(best)) return TDGL_MESH_ERR_DEGENERATE;
        if (orient(pt(i0), pt(i1), pt(i2)) < 0.0) std::swap(i1, i2);
        {
            const double *a = pt(i0), *b = pt(i1), *c = pt(i2);
            const double bx = b[0] - a[0], by = b[1] - a[1], ex = c[0] - a[0], ey = c[1] - a[1];
            const double bl = bx * bx + by * by, el = ex * ex + ey * ey, d = bx * ey - by * ex;
            cx = a[0] + (ey * bl - by * el) * (0.5 / d);
            cy = a[1] + (bx * el - ex * bl) * (0.5 / d);
        }
        // insertion order: distance from the seed circumcentre (ties: index)
        std::vector<double> dist((size_t)n);
        for (int i = 0; i < n; ++i) dist[i] = dist2(pt(i), cx, cy);
        std::vector<int> order((size_t)n);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return dist[a] < dist[b] || (dist[a] == dist[b] && a < b); });

        hash_size = std::max(16, (int)std::ceil(std::sqrt((double)n)));
        hash.assign((size_t)hash_size, -1);
        hull_next.assign((size_t)n, -1);
        hull_prev.assign((size_t)n, -1);
        hull_edge.assign((size_t)n, -1);
        tri.reserve(6 * (size_t)n);
        twin.reserve(6 * (size_t)n);

        hull_next[i0] = hull_prev[i2] = i1;
        hull_next[i1] = hull_prev[i0] = i2;
        hull_next[i2] = hull_prev[i1] = i0;
        add_triangle(i0, i1, i2, -1, -1, -1);  // link() fills hull_edge
        hash[hash_key(pt(i0))] = i0;
        hash[hash_key(pt(i1))] = i1;
        hash[hash_key(pt(i2))] = i2;

        int64_t skipped = 0;
        for (int k = 0; k < n; ++k) {
            const int i = order[k];
            if (i == i0 || i == i1 || i == i2) continue;
            const double *p = pt(i);
            // a hull vertex near p's direction that is still on the hull
            int start = -1;
            const int key = hash_key(p);
            for (int j = 0; j < hash_size; ++j) {
                start = hash[(key + j) % hash_size];
                if (start >= 0 && hull_next[start] != start) break;
                start = -1;
            }
            if (start < 0) return TDGL_MESH_ERR_DEGENERATE;  // cannot happen: the hull is never empty
            start = hull_prev[start];
            // first hull edge e -> q that p sees (p strictly on its right)
            int e = start, q = hull_next[e];
            bool found = true;
            while (!(orient(pt(e), pt(q), p) < 0.0)) {
                e = q;
                if (e == start) {
                    found = false;
                    break;
                }
                q = hull_next[e];
            }
            if (!found) {
                // p sees no hull edge: it is not outside the hull after all -- it repeats a point, or rounding in the
                // sweep order let a point farther out go first.  Insert it into the triangle that contains it.
                if (insert_inside(i, hull_edge[start]) != 0) ++skipped;
                continue;
            }
            // walk back: the search may have started inside the visible chain
            for (;;) {
                const int ep = hull_prev[e];
                if (!(orient(pt(ep), pt(e), p) < 0.0)) break;
                e = ep;
            }
            q = hull_next[e];
            // fan over the visible chain e -> q -> ...
            int h = add_triangle(e, i, q, -1, -1, hull_edge[e]);  // e->p and p->q are hull edges now
            legalize(h + 2);
            int first = e;
            for (;;) {
                const int q2 = hull_next[q];
                if (!(orient(pt(q), pt(q2), p) < 0.0)) break;
                h = add_triangle(q, i, q2, hull_edge[i], -1, hull_edge[q]);
                legalize(h + 2);
                hull_next[q] = q;  // q left the hull
                q = q2;
            }
            hull_prev[i] = first;
            hull_next[first] = i;
            hull_next[i] = q;
            hull_prev[q] = i;
            hash[hash_key(p)] = i;
            hash[hash_key(pt(first))] = first;
        }
        // closing sweeps over every interior edge until nothing flips (normally the first one finds nothing)
        const int nh = (int)tri.size();
        for (int round = 0; round < 64; ++round) {
            int64_t flips = 0;
            for (int h = 0; h < nh; ++h)
                if (twin[h] > h) flips += legalize(h, true);
            if (!flips) break;
        }
        const int64_t nt = nh / 3;
        for (int h = 0; h < nh; ++h) out[h] = tri[h];
        *n_out = nt;
        return skipped ? TDGL_MESH_ERR_SKIPPED : TDGL_MESH_OK;
    }
};

}  // namespace

extern "C" int tdgl_host_delaunay(int64_t n, const double *xy, int64_t *triangles, int64_t *n_triangles) try {
    if (!xy || !triangles || !n_triangles || n < 3 || n > (int64_t)300000000) return TDGL_MESH_ERR_ARG;
    Triangulator t;
    t.xy = xy;
    t.n = (int)n;
    return t.run(triangles, n_triangles);
} catch (...) {  // std::bad_alloc: no exception crosses the C boundary
    return TDGL_MESH_ERR_RESOURCES;
}

extern "C" int tdgl_host_is_delaunay(int64_t n, const double *xy, int64_t n_triangles, const int64_t *triangles) try {
    if (!xy || !triangles || n < 3 || n_triangles < 1) return TDGL_MESH_ERR_ARG;
    // edge -> (triangle, opposite vertex) through a sort of directed edges
    struct Rec {
        int64_t lo, hi, opp;
        int64_t t;
    };
    std::vector<Rec> recs;
    recs.reserve(3 * (size_t)n_triangles);
    for (int64_t t = 0; t < n_triangles; ++t)
        for (int k = 0; k < 3; ++k) {
            const int64_t a = triangles[3 * t + k], b = triangles[3 * t + (k + 1) % 3], c = triangles[3 * t + (k + 2) % 3];
            if (a < 0 || a >= n || b < 0 || b >= n) return TDGL_MESH_ERR_INDEX;
            recs.push_back(Rec{std::min(a, b), std::max(a, b), c, t});
        }
    std::sort(recs.begin(), recs.end(), [](const Rec &x, const Rec &y) { return x.lo != y.lo ? x.lo < y.lo : x.hi < y.hi; });
    for (size_t i = 0; i + 1 < recs.size(); ++i) {
        if (recs[i].lo != recs[i + 1].lo || recs[i].hi != recs[i + 1].hi) continue;
        const int64_t *ta = triangles + 3 * recs[i].t;
        const double *a = xy + 2 * ta[0], *b = xy + 2 * ta[1], *c = xy + 2 * ta[2];
        const double *d = xy + 2 * recs[i + 1].opp;
        const double o = orient(a, b, c);
        if (o == 0.0) return 0;
        const double s = o > 0.0 ? incircle(a, b, c, d) : incircle(a, c, b, d);
        if (s > 0.0) return 0;
    }
    return 1;
} catch (...) {  // std::bad_alloc: no exception crosses the C boundary
    return TDGL_MESH_ERR_RESOURCES;
}

extern "C" int tdgl_host_dual_mesh(int64_t n, const double *xy, int64_t nt, const int64_t *tri, int64_t *n_edges, int64_t *edges,
                                   uint8_t *is_boundary, int64_t *tri_edge, double *centers, double *directions,
                                   double *edge_lengths, double *cc, double *dual, double *areas, uint8_t *suspicious) try {
    if (!xy || !tri || !n_edges || !edges || !is_boundary || !tri_edge || !centers || !directions || !edge_lengths || !cc || !dual ||
        !areas || !suspicious || n < 3 || nt < 1)
        return TDGL_MESH_ERR_ARG;
    for (int64_t i = 0; i < 3 * nt; ++i)
        if (tri[i] < 0 || tri[i] >= n) return TDGL_MESH_ERR_INDEX;
    // --- unique edges, ascending by (lower, upper) site: bucket by the lower site, order each bucket ----------
    struct Half {
        int64_t hi, slot;  // upper site; 3 t + k
    };
    std::vector<int64_t> start((size_t)n + 1, 0);
    for (int64_t t = 0; t < nt; ++t)
        for (int k = 0; k < 3; ++k) ++start[(size_t)std::min(tri[3 * t + k], tri[3 * t + (k + 1) % 3]) + 1];
    for (int64_t i = 0; i < n; ++i) start[(size_t)i + 1] += start[(size_t)i];
    std::vector<Half> halves((size_t)(3 * nt));
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t t = 0; t < nt; ++t)  // ascending slot inside every bucket
            for (int k = 0; k < 3; ++k) {
                const int64_t a = tri[3 * t + k], b = tri[3 * t + (k + 1) % 3];
                halves[(size_t)fill[(size_t)std::min(a, b)]++] = Half{std::max(a, b), 3 * t + k};
            }
    }
    int64_t m = 0;
    std::vector<int64_t> t_a, t_b;
    t_a.reserve((size_t)(2 * nt));
    t_b.reserve((size_t)(2 * nt));
    for (int64_t lo = 0; lo < n; ++lo) {
        Half *first = halves.data() + start[(size_t)lo], *last = halves.data() + start[(size_t)lo + 1];
        // a handful of entries: insertion sort by upper site, stable in the slot
        for (Half *i = first + 1; i < last; ++i) {
            const Half v = *i;
            Half *j = i;
            for (; j > first && (j - 1)->hi > v.hi; --j) *j = *(j - 1);
            *j = v;
        }
        for (Half *i = first; i < last;) {
            Half *j = i;
            while (j < last && j->hi == i->hi) ++j;
            edges[2 * m] = lo;
            edges[2 * m + 1] = i->hi;
            is_boundary[m] = (j - i) == 1;
            for (Half *k = i; k < j; ++k) tri_edge[k->slot] = m;
            t_a.push_back(i->slot / 3);                      // lowest (triangle, local edge) first, as a stable sort gives
            t_b.push_back(j - i > 1 ? (j - 1)->slot / 3 : -1);  // (a third triangle on an edge: the last one, like the NumPy scatter)
            ++m;
            i = j;
        }
    }
    *n_edges = m;
    // --- edge midpoints, directions, lengths ----------------------------------------------------------------
    for (int64_t e = 0; e < m; ++e) {
        const double *s0 = xy + 2 * edges[2 * e], *s1 = xy + 2 * edges[2 * e + 1];
        centers[2 * e] = (s0[0] + s1[0]) / 2.0;
        centers[2 * e + 1] = (s0[1] + s1[1]) / 2.0;
        const double d0 = s1[0] - s0[0], d1 = s1[1] - s0[1];
        directions[2 * e] = d0;
        directions[2 * e + 1] = d1;
        edge_lengths[e] = std::sqrt(d0 * d0 + d1 * d1);
    }
    // --- circumcentres --------------------------------------------------------------------------------
    for (int64_t t = 0; t < nt; ++t) {
        const double *p0 = xy + 2 * tri[3 * t], *p1 = xy + 2 * tri[3 * t + 1], *p2 = xy + 2 * tri[3 * t + 2];
        const double u0 = p1[0] - p0[0], u1 = p1[1] - p0[1], v0 = p2[0] - p0[0], v1 = p2[1] - p0[1];
        const double uu = u0 * u0 + u1 * u1, vv = v0 * v0 + v1 * v1;
        const double det = 2 * u0 * v1 - 2 * u1 * v0;
        cc[2 * t] = (v1 * uu - u1 * vv) / det + p0[0];
        cc[2 * t + 1] = (u0 * vv - v0 * uu) / det + p0[1];
    }
    // --- dual edge lengths ------------------------------------------------------------------------------
    for (int64_t e = 0; e < m; ++e) {
        const double *a = cc + 2 * t_a[(size_t)e];
        double dx, dy;
        if (t_b[(size_t)e] >= 0) {
            const double *b = cc + 2 * t_b[(size_t)e];
            dx = a[0] - b[0];
            dy = a[1] - b[1];
        } else {
            dx = a[0] - centers[2 * e];
            dy = a[1] - centers[2 * e + 1];
        }
        dual[e] = std::sqrt(dx * dx + dy * dy);
    }
    // --- cell areas: signed kites, in the summation order of the NumPy construction --------------------
    std::fill(areas, areas + n, 0.0);
    std::fill(suspicious, suspicious + n, (uint8_t)0);
    std::vector<double> contrib((size_t)nt);
    for (int k = 0; k < 3; ++k) {
        const int ip = k, iq = (k + 1) % 3, ir = (k + 2) % 3;
        for (int64_t t = 0; t < nt; ++t) {
            const double *p = xy + 2 * tri[3 * t + ip], *q = xy + 2 * tri[3 * t + iq], *r = xy + 2 * tri[3 * t + ir];
            const double d0 = q[0] - p[0], d1 = q[1] - p[1];
            const double length = std::sqrt(d0 * d0 + d1 * d1);
            const double mid0 = 0.5 * (p[0] + q[0]), mid1 = 0.5 * (p[1] + q[1]);
            const double n0 = -d1 / length, n1 = d0 / length;
            const double s = (r[0] - p[0]) * n0 + (r[1] - p[1]) * n1;
            const double side = s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : s);  // numpy.sign (0 -> 0, nan -> nan)
            const double h = ((cc[2 * t] - mid0) * n0 + (cc[2 * t + 1] - mid1) * n1) * side;
            contrib[(size_t)t] = 0.25 * length * h;
            if (h < -1e-14 * length) suspicious[tri[3 * t + ip]] = suspicious[tri[3 * t + iq]] = 1;
        }
        for (int64_t t = 0; t < nt; ++t) areas[tri[3 * t + ip]] += contrib[(size_t)t];
        for (int64_t t = 0; t < nt; ++t) areas[tri[3 * t + iq]] += contrib[(size_t)t];
    }
    return TDGL_MESH_OK;
} catch (...) {  // std::bad_alloc: no exception crosses the C boundary
    return TDGL_MESH_ERR_RESOURCES;
}
