// Delaunay triangulation of the valid pixel centres in the (lat, lon) plane — the triangulation scipy.interpolate.griddata
// (method='linear' | 'cubic', reference auromat/resample.py:323-326) gets from Qhull (scipy.spatial.Delaunay: "Qbb Qc Qz Q12 Qt").
//
// Host code (the reference's triangulation is host code too: Qhull).  Qhull is a third-party dependency of the reference that is
// absent from /root/reference; what it computes is the Delaunay triangulation of the point set, which is UNIQUE whenever no four
// points are cocircular and no three collinear on the hull — so any exact algorithm reproduces it, triangle for triangle
// (tests/test_delaunay_cpu.py: equal to scipy.spatial.Delaunay's simplices on every fixture of the reference's 'cubic' outputs).
// Where four points ARE cocircular to the last bit (an undistorted lattice) Qhull's choice of diagonal follows from its facet
// merging, and where hull points are collinear up to rounding (straight rows of such a lattice) it keeps or merges slivers of
// 1e-16 as its roundoff tolerances fall: neither is reproduced — this triangulation is the exact one of the doubles given.
// Projected camera grids are curved by many orders of magnitude more than that.
//
// Algorithm: incremental Bowyer-Watson with ghost triangles for the hull (no super-triangle: the hull is exact), points inserted
// along a Hilbert curve (the previous point is a neighbour: the walk to the new point takes a step or two), orientation and in-circle tests in double precision behind Shewchuk's static error bounds with a binary128 evaluation
// behind them (orientation: exact there; in-circle: to 1e-33 relative, and anything below ITS error bound counts as cocircular).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/auromat_hip.h"

namespace {

// How often the predicates had to leave double precision, and how often the wider evaluation still could not tell (a tie: four
// points cocircular, three collinear, as far as 113 bits can see).  Zero ties = the triangulation is the unique one.
struct predicate_stats {
    long long orient_wide, orient_zero, incircle_wide, incircle_zero;
};
thread_local predicate_stats* g_stats = nullptr;

constexpr int kInf = -1;                // the vertex at infinity of a ghost triangle
constexpr double kEps = 1.1102230246251565e-16;          // 2^-53

struct pt {
    double x, y;
};

// > 0: a, b, c counter-clockwise
inline double orient2d(const pt& a, const pt& b, const pt& c) {
    const double l = (a.x - c.x) * (b.y - c.y), r = (a.y - c.y) * (b.x - c.x);
    const double det = l - r;
    const double bound = (3.0 + 16.0 * kEps) * kEps * (std::fabs(l) + std::fabs(r));
    if (det > bound || -det > bound) return det;
    // (differences of doubles of comparable size and their products fit 113 bits: exact)
    const __float128 L = ((__float128)a.x - c.x) * ((__float128)b.y - c.y), R = ((__float128)a.y - c.y) * ((__float128)b.x - c.x);
    const __float128 d = L - R;
    if (g_stats) {
        ++g_stats->orient_wide;
        if (d == 0) ++g_stats->orient_zero;
    }
    return d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0);
}

// > 0: d strictly inside the circle through a, b, c (counter-clockwise)
inline double incircle(const pt& a, const pt& b, const pt& c, const pt& d) {
    const double adx = a.x - d.x, ady = a.y - d.y, bdx = b.x - d.x, bdy = b.y - d.y, cdx = c.x - d.x, cdy = c.y - d.y;
    const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy, alift = adx * adx + ady * ady;
    const double cdxady = cdx * ady, adxcdy = adx * cdy, blift = bdx * bdx + bdy * bdy;
    const double adxbdy = adx * bdy, bdxady = bdx * ady, clift = cdx * cdx + cdy * cdy;
    const double det = alift * (bdxcdy - cdxbdy) + blift * (cdxady - adxcdy) + clift * (adxbdy - bdxady);
    const double permanent = (std::fabs(bdxcdy) + std::fabs(cdxbdy)) * alift + (std::fabs(cdxady) + std::fabs(adxcdy)) * blift +
                             (std::fabs(adxbdy) + std::fabs(bdxady)) * clift;
    const double bound = (10.0 + 96.0 * kEps) * kEps * permanent;
    if (det > bound || -det > bound) return det;
    typedef __float128 q;
    const q Adx = (q)a.x - d.x, Ady = (q)a.y - d.y, Bdx = (q)b.x - d.x, Bdy = (q)b.y - d.y, Cdx = (q)c.x - d.x, Cdy = (q)c.y - d.y;
    const q t1 = Bdx * Cdy, t2 = Cdx * Bdy, t3 = Cdx * Ady, t4 = Adx * Cdy, t5 = Adx * Bdy, t6 = Bdx * Ady;
    const q al = Adx * Adx + Ady * Ady, bl = Bdx * Bdx + Bdy * Bdy, cl = Cdx * Cdx + Cdy * Cdy;
    const q D = al * (t1 - t2) + bl * (t3 - t4) + cl * (t5 - t6);
    auto ab = [](q v) { return v < 0 ? -v : v; };
    const q P = (ab(t1) + ab(t2)) * al + (ab(t3) + ab(t4)) * bl + (ab(t5) + ab(t6)) * cl;
    const q B = P * (q)1.6e-32;                       // ~ 16 x 2^-112 x the permanent
    if (g_stats) ++g_stats->incircle_wide;
    if (D > B) return 1.0;
    if (-D > B) return -1.0;
    if (g_stats) ++g_stats->incircle_zero;
    return 0.0;                                        // cocircular as far as 113 bits can tell
}

}  // namespace

struct amt_delaunay {
    std::vector<pt> p;
    std::vector<int> v;          // 3 per triangle: vertices, counter-clockwise; kInf marks a ghost (hull edge + point at infinity)
    std::vector<int> adj;        // 3 per triangle: the triangle across the edge opposite vertex k
    std::vector<char> dead;
    std::vector<int> free_list;
    std::vector<int> vert_tri;   // a live triangle incident to each vertex (-1: not inserted — a duplicate point)
    // scratch of insert()
    std::vector<int> cavity, stack, edge_from, edge_to, touched;
    std::vector<char> in_cavity;
    int last;                    // a live finite triangle near the latest point
    int n_dup;
    predicate_stats stats;       // of build()
    // compacted result
    std::vector<int> tri;        // 3 per finite triangle
    std::vector<int> nbr;        // 3 per finite triangle: finite neighbour opposite vertex k or -1
    std::vector<int> slot_of;    // triangle slot -> index in `tri` (-1: ghost / dead)
    std::vector<int64_t> indptr;
    std::vector<int> indices;

    int new_tri(int a, int b, int c) {
        int t;
        if (!free_list.empty()) {
            t = free_list.back();
            free_list.pop_back();
            dead[t] = 0;
        } else {
            t = (int)dead.size();
            dead.push_back(0);
            v.resize(v.size() + 3);
            adj.resize(adj.size() + 3);
            in_cavity.push_back(0);
        }
        v[3 * t] = a, v[3 * t + 1] = b, v[3 * t + 2] = c;
        adj[3 * t] = adj[3 * t + 1] = adj[3 * t + 2] = -1;
        return t;
    }

    // does point q conflict with triangle t (must t go when q is inserted)?
    bool conflicts(int t, const pt& q) const {
        const int* w = &v[3 * t];
        for (int k = 0; k < 3; ++k) {
            if (w[k] != kInf) continue;
            // ghost: the hull edge a -> b with the outside on its left
            const pt &a = p[w[(k + 1) % 3]], &b = p[w[(k + 2) % 3]];
            const double o = orient2d(a, b, q);
            if (o > 0) return true;
            if (o < 0) return false;
            // on the line of the hull edge: in conflict when strictly between its end points
            const double dx = b.x - a.x, dy = b.y - a.y;
            const double s = (q.x - a.x) * dx + (q.y - a.y) * dy;
            return s > 0 && s < dx * dx + dy * dy;
        }
        return incircle(p[w[0]], p[w[1]], p[w[2]], q) > 0;
    }

    // a triangle in conflict with q: visibility walk from `last` (terminates on Delaunay triangulations)
    int locate(const pt& q) const {
        int t = last;
        int64_t guard = (int64_t)dead.size() * 4 + 64;
        for (;;) {
            if (--guard < 0) return -1;
            const int* w = &v[3 * t];
            if (w[0] == kInf || w[1] == kInf || w[2] == kInf) {
                if (conflicts(t, q)) return t;
                // a ghost that is not in conflict: q lies beyond one of the hull edge's ends (or inside): go round the hull
                int k = w[0] == kInf ? 0 : (w[1] == kInf ? 1 : 2);
                const pt &a = p[w[(k + 1) % 3]], &b = p[w[(k + 2) % 3]];
                if (orient2d(a, b, q) < 0) {
                    t = adj[3 * t + k];           // inside the hull as seen from this edge: back into the finite triangle
                    continue;
                }
                // collinear, outside the segment: towards the nearer end
                const double dx = b.x - a.x, dy = b.y - a.y;
                const double s = (q.x - a.x) * dx + (q.y - a.y) * dy;
                t = adj[3 * t + (s <= 0 ? (k + 2) % 3 : (k + 1) % 3)];      // across (inf, a) or (b, inf)
                continue;
            }
            int k;
            for (k = 0; k < 3; ++k) {
                const pt &a = p[w[(k + 1) % 3]], &b = p[w[(k + 2) % 3]];
                if (orient2d(a, b, q) < 0) break;
            }
            if (k == 3) return t;                 // inside or on the border of t: t is in conflict (or q duplicates a vertex)
            t = adj[3 * t + k];
        }
    }

    bool insert(int iq) {
        const pt q = p[iq];
        const int t0 = locate(q);
        if (t0 < 0) return false;
        if (!conflicts(t0, q)) {
            // q coincides with a vertex of t0 (its circle passes through q): a duplicate point, left out like Qhull's coplanar points
            ++n_dup;
            return true;
        }
        cavity.clear();
        stack.clear();
        stack.push_back(t0);
        in_cavity[t0] = 1;
        while (!stack.empty()) {
            const int t = stack.back();
            stack.pop_back();
            cavity.push_back(t);
            for (int k = 0; k < 3; ++k) {
                const int n = adj[3 * t + k];
                if (n >= 0 && !in_cavity[n] && conflicts(n, q)) {
                    in_cavity[n] = 1;
                    stack.push_back(n);
                }
            }
        }
        // new triangles: one per edge of the cavity's border, (a, b, q) with the cavity on the left of a -> b
        touched.clear();
        const int np = (int)p.size();
        auto key = [np](int vtx) { return vtx == kInf ? np : vtx; };
        int first_new = -1;
        for (size_t ci = 0; ci < cavity.size(); ++ci) {
            const int t = cavity[ci];
            for (int k = 0; k < 3; ++k) {
                const int n = adj[3 * t + k];
                if (n >= 0 && in_cavity[n]) continue;
                const int a = v[3 * t + (k + 1) % 3], b = v[3 * t + (k + 2) % 3];
                const int T = new_tri(a, b, iq);
                adj[3 * T + 2] = n;                               // opposite q: the triangle outside the cavity
                if (n >= 0)
                    for (int m = 0; m < 3; ++m)
                        if (adj[3 * n + m] == t && v[3 * n + (m + 1) % 3] == b && v[3 * n + (m + 2) % 3] == a) adj[3 * n + m] = T;
                edge_from[key(a)] = T;                            // the new triangle whose border edge STARTS at a
                edge_to[key(b)] = T;                              // ... ENDS at b
                touched.push_back(key(a));
                touched.push_back(key(b));
                if (a != kInf) vert_tri[a] = T;
                if (b != kInf) vert_tri[b] = T;
                first_new = T;
            }
        }
        // link the fan: triangle (a, b, q): across (b, q) [opposite a] lies the triangle that starts at b, across (q, a)
        // [opposite b] the one that ends at a (the cavity's border is one closed loop: every vertex starts and ends one edge)
        for (size_t i = 0; i < touched.size(); i += 2) {
            const int T = edge_from[touched[i]];
            const int a = v[3 * T], b = v[3 * T + 1];
            adj[3 * T] = edge_from[key(b)];
            adj[3 * T + 1] = edge_to[key(a)];
        }
        for (size_t ci = 0; ci < cavity.size(); ++ci) {
            const int t = cavity[ci];
            in_cavity[t] = 0;
            dead[t] = 1;
            free_list.push_back(t);
        }
        vert_tri[iq] = first_new;
        // the next walk starts at a finite triangle of the fan
        last = first_new;
        for (size_t i = 0; i < touched.size(); i += 2) {
            const int T = edge_from[touched[i]];
            if (v[3 * T] != kInf && v[3 * T + 1] != kInf) {
                last = T;
                break;
            }
        }
        return true;
    }

    // Insertion order: along a Hilbert curve through the points' bounding box.  In row-major order every new point of a
    // partly filled row destroys and rebuilds the fan of skinny triangles between that row's end and the rest of the row
    // above (O(width) per point, 36 us per point on a 1400 x 2000 lattice); along the curve the inserted set is a union of
    // squares at every moment, the hull stays short and a point's cavity small.  The triangulation does not depend on it.
    static uint32_t hilbert_d(uint32_t x, uint32_t y) {          // 16-bit x, y -> position on the curve of order 16
        uint32_t d = 0;
        for (uint32_t s = 1u << 15; s > 0; s >>= 1) {
            const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
            d += s * s * ((3u * rx) ^ ry);
            if (ry == 0) {
                if (rx == 1) {
                    x = 65535u - x;
                    y = 65535u - y;
                }
                const uint32_t t = x;
                x = y;
                y = t;
            }
        }
        return d;
    }

    bool build() {
        const int n = (int)p.size();
        vert_tri.assign(n, -1);
        edge_from.assign(n + 1, -1);
        edge_to.assign(n + 1, -1);
        n_dup = 0;
        if (n < 3) return false;
        double x0 = p[0].x, x1 = p[0].x, y0 = p[0].y, y1 = p[0].y;
        for (int i = 1; i < n; ++i) {
            x0 = std::min(x0, p[i].x), x1 = std::max(x1, p[i].x);
            y0 = std::min(y0, p[i].y), y1 = std::max(y1, p[i].y);
        }
        const double sx = x1 > x0 ? 65535.0 / (x1 - x0) : 0.0, sy = y1 > y0 ? 65535.0 / (y1 - y0) : 0.0;
        std::vector<uint64_t> keyed((size_t)n);
        for (int i = 0; i < n; ++i)
            keyed[(size_t)i] = ((uint64_t)hilbert_d((uint32_t)((p[i].x - x0) * sx), (uint32_t)((p[i].y - y0) * sy)) << 32) | (uint32_t)i;
        std::sort(keyed.begin(), keyed.end());
        auto at = [&](int k) { return (int)(uint32_t)keyed[(size_t)k]; };
        // first triangle: the first point of the order, the first that differs from it and the first not collinear with them
        const int a0 = at(0);
        int k1 = 1;
        while (k1 < n && p[at(k1)].x == p[a0].x && p[at(k1)].y == p[a0].y) ++k1;
        if (k1 >= n) return false;
        int k2 = k1 + 1;
        while (k2 < n && orient2d(p[a0], p[at(k1)], p[at(k2)]) == 0) ++k2;
        if (k2 >= n) return false;
        int a = a0, b = at(k1), c = at(k2);
        if (orient2d(p[a], p[b], p[c]) < 0) std::swap(b, c);
        const size_t guess = (size_t)n * 2 + 16;
        v.reserve(3 * guess), adj.reserve(3 * guess), dead.reserve(guess), in_cavity.reserve(guess);
        const int t = new_tri(a, b, c);
        // ghosts: across the edge opposite vertex k of t, i.e. (v[k+1], v[k+2]), lies (v[k+2], v[k+1], inf)
        int g[3];
        for (int k = 0; k < 3; ++k) {
            g[k] = new_tri(v[3 * t + (k + 2) % 3], v[3 * t + (k + 1) % 3], kInf);
            adj[3 * t + k] = g[k];
            adj[3 * g[k] + 2] = t;
        }
        for (int k = 0; k < 3; ++k) {
            // ghost g[k] = (x, y, inf): across (y, inf) [opposite x] the ghost that starts at y, across (inf, x) [opposite y]
            // the ghost that ends at x
            const int x = v[3 * g[k]], y = v[3 * g[k] + 1];
            for (int m = 0; m < 3; ++m) {
                if (v[3 * g[m]] == y) adj[3 * g[k]] = g[m];
                if (v[3 * g[m] + 1] == x) adj[3 * g[k] + 1] = g[m];
            }
        }
        vert_tri[a] = vert_tri[b] = vert_tri[c] = t;
        last = t;
        for (int k = 0; k < n; ++k) {
            const int i = at(k);
            if (i == a || i == b || i == c) continue;
            if (!insert(i)) return false;
        }
        return true;
    }

    void compact() {
        const int nt = (int)dead.size();
        slot_of.assign(nt, -1);
        tri.clear();
        for (int t = 0; t < nt; ++t) {
            if (dead[t] || v[3 * t] == kInf || v[3 * t + 1] == kInf || v[3 * t + 2] == kInf) continue;
            slot_of[t] = (int)(tri.size() / 3);
            tri.push_back(v[3 * t]), tri.push_back(v[3 * t + 1]), tri.push_back(v[3 * t + 2]);
        }
        nbr.assign(tri.size(), -1);
        for (int t = 0; t < nt; ++t) {
            if (slot_of[t] < 0) continue;
            for (int k = 0; k < 3; ++k) {
                const int n = adj[3 * t + k];
                nbr[3 * slot_of[t] + k] = n >= 0 ? slot_of[n] : -1;
            }
        }
        // vertex -> neighbouring vertices (scipy.spatial.Delaunay.vertex_neighbor_vertices; in no particular order).  Every
        // triangle lists its edges counter-clockwise, so an inner edge {x, y} comes up once as x -> y and once as y -> x (from
        // the triangle on its other side); a hull edge comes up once and gets its reverse here: no sorting, no duplicates
        const int n = (int)p.size();
        const size_t m = tri.size() / 3;
        indptr.assign((size_t)n + 1, 0);
        for (size_t t = 0; t < m; ++t)
            for (int k = 0; k < 3; ++k) {
                ++indptr[(size_t)tri[3 * t + (k + 1) % 3] + 1];                       // edge opposite vertex k: v[k+1] -> v[k+2]
                if (nbr[3 * t + k] < 0) ++indptr[(size_t)tri[3 * t + (k + 2) % 3] + 1];
            }
        for (int i = 0; i < n; ++i) indptr[(size_t)i + 1] += indptr[(size_t)i];
        indices.resize((size_t)indptr[(size_t)n]);
        std::vector<int64_t> fill(indptr.begin(), indptr.end() - 1);
        for (size_t t = 0; t < m; ++t)
            for (int k = 0; k < 3; ++k) {
                const int x = tri[3 * t + (k + 1) % 3], y = tri[3 * t + (k + 2) % 3];
                indices[(size_t)fill[(size_t)x]++] = y;
                if (nbr[3 * t + k] < 0) indices[(size_t)fill[(size_t)y]++] = x;
            }
    }
};

extern "C" {

int amt_delaunay_create(const double* xy, int64_t n, amt_delaunay** out) {
    if (xy == nullptr || out == nullptr || n < 3 || n > 2000000000ll) return AMT_EINVAL;
    *out = nullptr;
    amt_delaunay* d = new (std::nothrow) amt_delaunay();
    if (d == nullptr) return AMT_ENOMEM;
    try {
        d->p.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            d->p[(size_t)i].x = xy[2 * i], d->p[(size_t)i].y = xy[2 * i + 1];
            if (!(std::isfinite(xy[2 * i]) && std::isfinite(xy[2 * i + 1]))) {
                delete d;
                return AMT_EINVAL;
            }
        }
        d->stats = predicate_stats{0, 0, 0, 0};
        g_stats = &d->stats;
        const bool built = d->build();
        g_stats = nullptr;
        if (!built) {
            delete d;
            return AMT_EINVAL;         // fewer than three points that are not collinear
        }
        d->compact();
    } catch (const std::bad_alloc&) {
        delete d;
        return AMT_ENOMEM;
    }
    *out = d;
    return AMT_OK;
}

int amt_delaunay_destroy(amt_delaunay* d) {
    if (d == nullptr) return AMT_EINVAL;
    delete d;
    return AMT_OK;
}

int amt_delaunay_sizes(const amt_delaunay* d, int64_t* n_triangles, int64_t* n_neighbours, int64_t* n_duplicates) {
    if (d == nullptr) return AMT_EINVAL;
    if (n_triangles) *n_triangles = (int64_t)(d->tri.size() / 3);
    if (n_neighbours) *n_neighbours = (int64_t)d->indices.size();
    if (n_duplicates) *n_duplicates = d->n_dup;
    return AMT_OK;
}

int amt_delaunay_stats(const amt_delaunay* d, int64_t* stats4) {
    if (d == nullptr || stats4 == nullptr) return AMT_EINVAL;
    stats4[0] = d->stats.orient_wide, stats4[1] = d->stats.orient_zero;
    stats4[2] = d->stats.incircle_wide, stats4[3] = d->stats.incircle_zero;
    return AMT_OK;
}

int amt_delaunay_triangles(const amt_delaunay* d, int32_t* simplices, int32_t* neighbours) {
    if (d == nullptr) return AMT_EINVAL;
    if (simplices) std::memcpy(simplices, d->tri.data(), d->tri.size() * sizeof(int32_t));
    if (neighbours) std::memcpy(neighbours, d->nbr.data(), d->nbr.size() * sizeof(int32_t));
    return AMT_OK;
}

int amt_delaunay_vertex_neighbours(const amt_delaunay* d, int64_t* indptr, int32_t* indices) {
    if (d == nullptr || indptr == nullptr || indices == nullptr) return AMT_EINVAL;
    std::memcpy(indptr, d->indptr.data(), d->indptr.size() * sizeof(int64_t));
    std::memcpy(indices, d->indices.data(), d->indices.size() * sizeof(int32_t));
    return AMT_OK;
}

int amt_delaunay_locate(const amt_delaunay* d, const double* targets, int64_t m, int32_t* vertices, double* centroids,
                        uint8_t* has_neighbour) {
    if (d == nullptr || targets == nullptr || vertices == nullptr || centroids == nullptr || has_neighbour == nullptr || m < 0)
        return AMT_EINVAL;
    const size_t nt = d->tri.size() / 3;
    if (nt == 0) return AMT_EINVAL;
    int t = 0;
    for (int64_t i = 0; i < m; ++i) {
        const pt q = {targets[2 * i], targets[2 * i + 1]};
        int32_t* vo = vertices + 3 * i;
        vo[0] = vo[1] = vo[2] = -1;
        for (int k = 0; k < 6; ++k) centroids[6 * i + k] = 0;
        has_neighbour[3 * i] = has_neighbour[3 * i + 1] = has_neighbour[3 * i + 2] = 0;
        if (!(std::isfinite(q.x) && std::isfinite(q.y))) continue;
        // visibility walk over the finite triangles; leaving through a hull edge means "outside" — unless another way round
        // exists (the hull is convex: a point beyond a hull edge is outside)
        int64_t guard = (int64_t)nt * 4 + 64;
        bool inside = false;
        for (;;) {
            if (--guard < 0) break;
            const int* w = &d->tri[3 * (size_t)t];
            int k;
            for (k = 0; k < 3; ++k)
                if (orient2d(d->p[w[(k + 1) % 3]], d->p[w[(k + 2) % 3]], q) < 0) break;
            if (k == 3) {
                inside = true;
                break;
            }
            const int n = d->nbr[3 * (size_t)t + k];
            if (n < 0) break;
            t = n;
        }
        if (!inside) continue;
        const int* w = &d->tri[3 * (size_t)t];
        for (int k = 0; k < 3; ++k) {
            vo[k] = w[k];
            const int n = d->nbr[3 * (size_t)t + k];
            if (n >= 0) {
                const int* u = &d->tri[3 * (size_t)n];
                has_neighbour[3 * i + k] = 1;
                centroids[6 * i + 2 * k] = (d->p[u[0]].x + d->p[u[1]].x + d->p[u[2]].x) / 3;
                centroids[6 * i + 2 * k + 1] = (d->p[u[0]].y + d->p[u[1]].y + d->p[u[2]].y) / 3;
            }
        }
    }
    return AMT_OK;
}

}  // extern "C"
