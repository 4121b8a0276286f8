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
//
// Large point sets are built in parallel (build_parallel): the plane is cut into vertical strips of equal point counts, every strip
// is triangulated by the algorithm above on a thread of its own, and neighbouring strips are joined the way divide-and-conquer
// Delaunay algorithms join their halves — the lower and upper common tangents of the two hulls, the gap between the facing hull
// chains filled by triangles that each take the next vertex of one chain, then Lawson's flips from the seam until every edge is
// locally Delaunay again (a few per seam vertex).  The Delaunay triangulation is unique (see above), so the result is the
// sequential one; the tests compare the two triangle for triangle.  AMT_DELAUNAY_THREADS (default: up to 16),
// AMT_DELAUNAY_PARALLEL_MIN (points from which on the parallel build is used; default 200 000).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/auromat_hip.h"

#ifdef AMT_DELAUNAY_STANDALONE
// (this file alone as a host library — tools/asan_delaunay.py builds it with g++ under the sanitizers —: the one function it takes
// from the rest of the library)
extern "C" int amt_host_threads(int wanted, int* cores, int* local_ranks) {
    if (cores) *cores = (int)std::thread::hardware_concurrency();
    if (local_ranks) *local_ranks = 1;
    return wanted < 1 ? 1 : wanted;
}
#endif

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

// f(j) for j = 0 .. parts - 1, each on a thread of its own (the calling thread takes part 0)
// (an exception — std::bad_alloc — thrown on one of them is thrown again here, after all have ended)
template <typename F>
void on_threads(int parts, F f) {
    std::vector<std::exception_ptr> failed((size_t)parts);
    auto guarded = [&](int j) {
        try {
            f(j);
        } catch (...) {
            failed[(size_t)j] = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int j = 1; j < parts; ++j) pool.emplace_back(guarded, j);
    } catch (...) {                                            // (no more threads to be had: the rest on this one)
        for (int j = (int)pool.size() + 1; j < parts; ++j) guarded(j);
    }
    guarded(0);
    for (auto& th : pool) th.join();
    for (auto& e : failed)
        if (e) std::rethrow_exception(e);
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
    int n_strips = 1;            // build_parallel: strips triangulated side by side, flips spent on joining them
    int64_t n_flips = 0;
    int n_threads = 1;           // threads compact() may use
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

    // ---- hull access through the ghosts: ghost (a, b, inf) is the hull edge a -> b with the outside on its left, i.e. the ring
    // of ghosts runs CLOCKWISE round the hull
    int inf_index(int t) const { return v[3 * t] == kInf ? 0 : (v[3 * t + 1] == kInf ? 1 : (v[3 * t + 2] == kInf ? 2 : -1)); }
    int ghost_start(int g) const { return v[3 * g + (inf_index(g) + 1) % 3]; }
    int ghost_end(int g) const { return v[3 * g + (inf_index(g) + 2) % 3]; }
    int ghost_next(int g) const { return adj[3 * g + (inf_index(g) + 1) % 3]; }       // the ghost that starts where g ends
    int ghost_prev(int g) const { return adj[3 * g + (inf_index(g) + 2) % 3]; }       // the ghost that ends where g starts
    int ghost_inner(int g) const { return adj[3 * g + inf_index(g)]; }                // the finite triangle behind the hull edge
    int slot_towards(int t, int n) const {                                            // k with adj[3t + k] == n
        for (int k = 0; k < 3; ++k)
            if (adj[3 * t + k] == n) return k;
        return -1;
    }
    void kill(int t) {
        dead[t] = 1;
        free_list.push_back(t);
    }

    // flips the edge opposite vertex k of finite triangle t when the vertex across it lies inside t's circle; pushes the four
    // outer edges of the quadrilateral.  Returns whether it flipped.
    bool flip_if_needed(int t, int k, std::vector<int>& queue) {
        if (dead[t] || inf_index(t) >= 0) return false;
        const int n = adj[3 * t + k];
        if (n < 0 || dead[n] || inf_index(n) >= 0) return false;
        const int m = slot_towards(n, t);
        if (m < 0) return false;
        const int a = v[3 * t + k], b = v[3 * t + (k + 1) % 3], c = v[3 * t + (k + 2) % 3], d = v[3 * n + m];
        if (!(incircle(p[a], p[b], p[c], p[d]) > 0)) return false;
        // t = (a, b, c), n = (d, c, b) in some rotation: the shared edge is b - c.  After the flip: t = (a, b, d), n = (a, d, c)
        const int t_ab = adj[3 * t + (k + 2) % 3];            // across a - b (opposite c)
        const int t_ca = adj[3 * t + (k + 1) % 3];            // across c - a (opposite b)
        const int n_bd = adj[3 * n + (m + 1) % 3];            // n = (d, c, b) from m: opposite c is the edge b - d
        const int n_dc = adj[3 * n + (m + 2) % 3];            // opposite b: the edge d - c
        v[3 * t] = a, v[3 * t + 1] = b, v[3 * t + 2] = d;
        adj[3 * t] = n_bd, adj[3 * t + 1] = n, adj[3 * t + 2] = t_ab;          // opposite a: b - d; opposite b: d - a; opposite d: a - b
        v[3 * n] = a, v[3 * n + 1] = d, v[3 * n + 2] = c;
        adj[3 * n] = n_dc, adj[3 * n + 1] = t_ca, adj[3 * n + 2] = t;          // opposite a: d - c; opposite d: c - a; opposite c: a - d
        if (n_bd >= 0) adj[3 * n_bd + slot_towards(n_bd, n)] = t;
        if (t_ca >= 0) adj[3 * t_ca + slot_towards(t_ca, t)] = n;
        vert_tri[a] = t, vert_tri[b] = t, vert_tri[d] = t, vert_tri[c] = n;
        queue.push_back(t), queue.push_back(0);               // b - d
        queue.push_back(t), queue.push_back(2);               // a - b
        queue.push_back(n), queue.push_back(0);               // d - c
        queue.push_back(n), queue.push_back(1);               // c - a
        return true;
    }

    // Joins two components of the structure — A strictly to the left of B (every x of A below every x of B) —, each a complete
    // triangulation with its ring of ghosts; ga / gb: any ghost of A / B.  Returns a ghost of the union, or -1 when the gap
    // between the hulls cannot be filled without a degenerate triangle (collinear runs across the seam: the caller then builds
    // sequentially).
    int join(int ga, int gb) {
        // the rightmost vertex of A and the leftmost of B (ties: the lower one), as ghosts that START there
        auto extreme = [&](int g0, bool rightmost) {
            int best = g0, g = g0;
            do {
                const pt &q = p[ghost_start(g)], &b = p[ghost_start(best)];
                if (rightmost ? (q.x > b.x || (q.x == b.x && q.y < b.y)) : (q.x < b.x || (q.x == b.x && q.y < b.y))) best = g;
                g = ghost_next(g);
            } while (g != g0);
            return best;
        };
        // Common tangents.  A vertex of A is held as the ghost that starts at it (ea), one of B as the ghost that ends at it (eb).
        // Downwards on A's right side is clockwise (next), on B's left side counter-clockwise (prev); upwards the other way.
        auto strictly_between = [&](const pt& x, const pt& a, const pt& b) {      // x on the line a b: closer to b than a is?
            const double dx = b.x - a.x, dy = b.y - a.y;
            const double s = (x.x - a.x) * dx + (x.y - a.y) * dy;
            return s > 0 && s < dx * dx + dy * dy;
        };
        auto tangent = [&](bool lower, int& ea, int& eb, int& moves_a, int& moves_b) {
            int64_t guard = (int64_t)dead.size() * 2 + 64;
            for (;;) {
                if (--guard < 0) return false;
                const int a = ghost_start(ea), b = ghost_end(eb);
                bool moved = false;
                // A's candidate: the next vertex downwards (lower) / upwards (upper)
                {
                    const int cand_g = lower ? ghost_next(ea) : ghost_prev(ea);
                    const int c = ghost_start(cand_g);
                    if (c != a) {
                        const double o = orient2d(p[a], p[b], p[c]);
                        if (lower ? (o < 0) : (o > 0)) ea = cand_g, moved = true;
                        else if (o == 0 && strictly_between(p[c], p[a], p[b])) ea = cand_g, moved = true;
                    }
                }
                if (moved) {
                    ++moves_a;
                    continue;
                }
                {
                    const int cand_g = lower ? ghost_prev(eb) : ghost_next(eb);
                    const int c = ghost_end(cand_g);
                    if (c != b) {
                        const double o = orient2d(p[a], p[b], p[c]);
                        if (lower ? (o < 0) : (o > 0)) eb = cand_g, moved = true;
                        else if (o == 0 && strictly_between(p[c], p[b], p[a])) eb = cand_g, moved = true;
                    }
                }
                if (!moved) return true;
                ++moves_b;
            }
        };
        auto ring_size = [&](int g0) {
            int count = 0, g = g0;
            do {
                ++count;
                g = ghost_next(g);
            } while (g != g0);
            return count;
        };
        const int ra = extreme(ga, true);
        int lb = extreme(gb, false);
        lb = ghost_prev(lb);                                   // ... as the ghost that ENDS at B's leftmost vertex
        int lo_a = ra, lo_b = lb, up_a = ra, up_b = lb;
        int moves_a = 0, moves_b = 0;
        if (!tangent(true, lo_a, lo_b, moves_a, moves_b) || !tangent(false, up_a, up_b, moves_a, moves_b)) return -1;
        const int a0 = ghost_start(lo_a), b0 = ghost_end(lo_b), a1 = ghost_start(up_a), b1 = ghost_end(up_b);
        // The facing chains: A's hull edges from a1 clockwise down to a0, B's from b0 clockwise up to b1.  When both tangents touch
        // a hull in ONE vertex the chain is empty if the walks never left the starting vertex — or the whole ring: that hull lies
        // inside the wedge the two tangents make, only that vertex of it is on the union's hull.
        int rem_a = 0, rem_b = 0;
        if (a0 == a1) {
            rem_a = moves_a > 0 ? ring_size(lo_a) : 0;
        } else {
            for (int g = up_a; ghost_start(g) != a0; g = ghost_next(g)) ++rem_a;
        }
        if (b0 == b1) {
            rem_b = moves_b > 0 ? ring_size(lo_b) : 0;
        } else {
            for (int g = ghost_next(lo_b);; g = ghost_next(g)) {
                ++rem_b;
                if (g == up_b) break;
            }
        }
        const bool whole_a = a0 == a1 && rem_a > 0, whole_b = b0 == b1 && rem_b > 0;
        if (whole_a && whole_b) return -1;                     // (cannot be: two hulls apart)
        // Dry run of the zip (no changes yet): every step needs a candidate strictly to the left of the base edge a -> b
        {
            int ga_cur = ghost_prev(lo_a), gb_cur = ghost_next(lo_b), ra_left = rem_a, rb_left = rem_b;
            int a = a0, b = b0;
            while (ra_left > 0 || rb_left > 0) {
                const int a_up = ra_left > 0 ? ghost_start(ga_cur) : -1, b_up = rb_left > 0 ? ghost_end(gb_cur) : -1;
                const bool va = ra_left > 0 && orient2d(p[a], p[b], p[a_up]) > 0, vb = rb_left > 0 && orient2d(p[a], p[b], p[b_up]) > 0;
                if (!va && !vb) return -1;
                bool take_b = vb;
                if (va && vb) take_b = !(incircle(p[a], p[b], p[b_up], p[a_up]) > 0);
                if (take_b) b = b_up, gb_cur = ghost_next(gb_cur), --rb_left; else a = a_up, ga_cur = ghost_prev(ga_cur), --ra_left;
            }
            if (a != a1 || b != b1) return -1;
        }
        // The ghosts outside the facing chains that the new hull edges link to (none on a side whose whole ring is consumed)
        const int below_b = lo_b;                              // ends at b0
        const int below_a = lo_a;                              // starts at a0 (lo_a runs on clockwise from a0, away from the gap)
        const int above_a = ghost_prev(up_a);                  // ends at a1
        const int above_b = ghost_next(up_b);                  // starts at b1
        const int g_low = new_tri(b0, a0, kInf);               // hull edge b0 -> a0, outside (below) on its left
        int prev_tri = g_low, prev_slot = 2;                   // the triangle below the current base and its slot for it
        std::vector<int> queue;
        int cur_a_ghost = ghost_prev(lo_a), cur_b_ghost = ghost_next(lo_b);        // the facing-chain ghosts next to be consumed
        int a = a0, b = b0;
        while (rem_a > 0 || rem_b > 0) {
            const int a_up = rem_a > 0 ? ghost_start(cur_a_ghost) : -1, b_up = rem_b > 0 ? ghost_end(cur_b_ghost) : -1;
            const bool va = rem_a > 0 && orient2d(p[a], p[b], p[a_up]) > 0, vb = rem_b > 0 && orient2d(p[a], p[b], p[b_up]) > 0;
            bool take_b = vb;
            if (va && vb) take_b = !(incircle(p[a], p[b], p[b_up], p[a_up]) > 0);
            const int x = take_b ? b_up : a_up;
            // (the ghost to be consumed is read BEFORE new_tri, which may hand out the slot of one consumed earlier)
            const int g = take_b ? cur_b_ghost : cur_a_ghost, inner = ghost_inner(g);
            if (take_b) cur_b_ghost = ghost_next(g), --rem_b; else cur_a_ghost = ghost_prev(g), --rem_a;
            const int T = new_tri(a, b, x);
            adj[3 * T + 2] = prev_tri;                         // opposite x: the base a - b
            adj[3 * prev_tri + prev_slot] = T;
            if (take_b) {
                // edge b - x (opposite a) was B's hull edge b -> x: its ghost goes, the finite triangle behind it is T's neighbour
                adj[3 * T] = inner;
                adj[3 * inner + slot_towards(inner, g)] = T;
                kill(g);
                prev_tri = T, prev_slot = 1;                   // opposite b: the edge x - a = the next base (a, x)
                b = x;
            } else {
                // edge x - a (opposite b) was A's hull edge x -> a
                adj[3 * T + 1] = inner;
                adj[3 * inner + slot_towards(inner, g)] = T;
                kill(g);
                prev_tri = T, prev_slot = 0;                   // opposite a: the edge b - x = the next base (x, b)
                a = x;
            }
            vert_tri[v[3 * T]] = T, vert_tri[v[3 * T + 1]] = T, vert_tri[v[3 * T + 2]] = T;
            for (int k = 0; k < 3; ++k) queue.push_back(T), queue.push_back(k);
        }
        const int g_up = new_tri(a1, b1, kInf);                // hull edge a1 -> b1, outside (above) on its left
        adj[3 * g_up + 2] = prev_tri;
        adj[3 * prev_tri + prev_slot] = g_up;
        // ring: g_low = (b0, a0, inf): across (a0, inf) [slot 0] the ghost that starts at a0, across (inf, b0) [slot 1] the one that
        // ends at b0; g_up = (a1, b1, inf): slot 0 the ghost that starts at b1, slot 1 the one that ends at a1
        auto set_prev = [&](int g, int to) { adj[3 * g + (inf_index(g) + 2) % 3] = to; };
        auto set_next = [&](int g, int to) { adj[3 * g + (inf_index(g) + 1) % 3] = to; };
        if (whole_a) {
            adj[3 * g_low] = g_up, adj[3 * g_up + 1] = g_low;           // a0 = a1 is A's only vertex on the hull
        } else {
            adj[3 * g_low] = below_a, set_prev(below_a, g_low);
            adj[3 * g_up + 1] = above_a, set_next(above_a, g_up);
        }
        if (whole_b) {
            adj[3 * g_up] = g_low, adj[3 * g_low + 1] = g_up;
        } else {
            adj[3 * g_low + 1] = below_b, set_next(below_b, g_low);
            adj[3 * g_up] = above_b, set_prev(above_b, g_up);
        }
        // Lawson: flip until every edge near the seam is locally Delaunay (the guard only trips on a structure that is not a
        // triangulation)
        int64_t flips = 0;
        const int64_t flip_limit = (int64_t)dead.size() * 16 + 1024;
        while (!queue.empty()) {
            const int k = queue.back();
            queue.pop_back();
            const int t = queue.back();
            queue.pop_back();
            if (flip_if_needed(t, k, queue) && ++flips > flip_limit) return -1;
        }
        n_flips += flips;
        return g_up;
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

    // (debugging aid, AMT_DELAUNAY_VALIDATE: every link of the structure is mutual and names the same edge from both sides)
    bool validate(const char* when) const {
        const int nt = (int)dead.size();
        for (int t = 0; t < nt; ++t) {
            if (dead[t]) continue;
            for (int k = 0; k < 3; ++k) {
                const int n = adj[3 * t + k];
                const int x = v[3 * t + (k + 1) % 3], y = v[3 * t + (k + 2) % 3];
                if (n < 0 || n >= nt || dead[n]) {
                    std::fprintf(stderr, "[delaunay] %s: triangle %d (%d %d %d) slot %d -> %d (dead or none)\n", when, t, v[3 * t], v[3 * t + 1],
                                 v[3 * t + 2], k, n);
                    return false;
                }
                int m = -1;
                for (int j = 0; j < 3; ++j)
                    if (adj[3 * n + j] == t && v[3 * n + (j + 1) % 3] == y && v[3 * n + (j + 2) % 3] == x) m = j;
                if (m < 0) {
                    std::fprintf(stderr, "[delaunay] %s: triangle %d (%d %d %d) slot %d -> %d (%d %d %d | %d %d %d): not mutual\n", when, t, v[3 * t],
                                 v[3 * t + 1], v[3 * t + 2], k, n, v[3 * n], v[3 * n + 1], v[3 * n + 2], adj[3 * n], adj[3 * n + 1], adj[3 * n + 2]);
                    return false;
                }
            }
            if (inf_index(t) < 0 && !(orient2d(p[v[3 * t]], p[v[3 * t + 1]], p[v[3 * t + 2]]) > 0)) {
                std::fprintf(stderr, "[delaunay] %s: triangle %d (%d %d %d) is not counter-clockwise\n", when, t, v[3 * t], v[3 * t + 1], v[3 * t + 2]);
                return false;
            }
        }
        return true;
    }

    // Parallel build (see the head of the file).  false: not applicable / a degenerate seam — the caller builds sequentially.
    bool build_parallel(int threads, int min_points) {
        const int n = (int)p.size();
        if (threads < 2 || n < min_points) return false;
        static const bool debug = std::getenv("AMT_DELAUNAY_DEBUG") != nullptr;             // phase times on stderr
        static const bool check = std::getenv("AMT_DELAUNAY_VALIDATE") != nullptr;       // every link checked after every join
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto t_begin = now();
        auto lap = [&](const char* what) {
            if (!debug) return;
            const auto t = now();
            std::fprintf(stderr, "[delaunay] %-12s %.3f s\n", what, std::chrono::duration<double>(t - t_begin).count());
            t_begin = t;
        };
        int K = std::min(threads, n / std::max(256, min_points / 4));          // (a strip has a quarter of the threshold at least)
        if (K < 2) return false;
        // splitters: x values from a sorted sample; a point goes to the strip of the last splitter <= its x, so that equal x values
        // never sit on both sides of a cut
        std::vector<double> sample;
        const int step = std::max(1, n / 65536);
        for (int i = 0; i < n; i += step) sample.push_back(p[i].x);
        std::sort(sample.begin(), sample.end());
        std::vector<double> cut;
        for (int j = 1; j < K; ++j) {
            const double c = sample[(size_t)((int64_t)sample.size() * j / K)];
            if (cut.empty() ? c > sample.front() : c > cut.back()) cut.push_back(c);
        }
        K = (int)cut.size() + 1;
        if (K < 2) return false;
        auto strip_of = [&](double x) { return (int)(std::upper_bound(cut.begin(), cut.end(), x) - cut.begin()); };
        // (every thread sorts a contiguous piece of the points into strips; a strip's members are the pieces' lists one after the
        // other, i.e. in increasing order of the point index, as one pass over all points would leave them)
        std::vector<std::vector<int>> members((size_t)K);
        {
            const int P = std::max(1, std::min(threads, n / 262144));
            std::vector<std::vector<std::vector<int>>> mine((size_t)P, std::vector<std::vector<int>>((size_t)K));
            on_threads(P, [&](int q) {
                const int a = (int)((int64_t)n * q / P), b = (int)((int64_t)n * (q + 1) / P);
                for (int j = 0; j < K; ++j) mine[(size_t)q][(size_t)j].reserve((size_t)(b - a) / K + 1024);
                for (int i = a; i < b; ++i) mine[(size_t)q][(size_t)strip_of(p[i].x)].push_back(i);
            });
            on_threads(std::min(K, threads), [&](int t) {
                for (int j = t; j < K; j += std::min(K, threads)) {
                    size_t total = 0;
                    for (int q = 0; q < P; ++q) total += mine[(size_t)q][(size_t)j].size();
                    members[(size_t)j].reserve(total);
                    for (int q = 0; q < P; ++q) members[(size_t)j].insert(members[(size_t)j].end(), mine[(size_t)q][(size_t)j].begin(), mine[(size_t)q][(size_t)j].end());
                }
            });
        }
        for (int j = 0; j < K; ++j)
            if (members[(size_t)j].size() < 64) return false;
        lap("partition");
        // every strip on a thread of its own
        std::vector<amt_delaunay> part((size_t)K);
        std::vector<char> ok((size_t)K, 0);
        std::vector<predicate_stats> part_stats((size_t)K, predicate_stats{0, 0, 0, 0});
        auto work = [&](int j) {
            amt_delaunay& d = part[(size_t)j];
            const std::vector<int>& mine = members[(size_t)j];
            d.p.resize(mine.size());
            for (size_t i = 0; i < mine.size(); ++i) d.p[i] = p[(size_t)mine[i]];
            g_stats = &part_stats[(size_t)j];
            bool built = false;
            try {
                built = d.build();
            } catch (const std::bad_alloc&) {
                built = false;
            }
            g_stats = nullptr;
            ok[(size_t)j] = built ? 1 : 0;
        };
        {
            predicate_stats* mine = g_stats;
            on_threads(K, work);
            g_stats = mine;
        }
        for (int j = 0; j < K; ++j)
            if (!ok[(size_t)j]) return false;
        lap("strips");
        // one structure: triangle ids shifted by the strips before, vertex ids back to the caller's (every strip copies itself)
        std::vector<size_t> offs((size_t)K + 1, 0);
        for (int j = 0; j < K; ++j) offs[(size_t)j + 1] = offs[(size_t)j] + part[(size_t)j].dead.size();
        const size_t total = offs[(size_t)K];
        v.resize(3 * total), adj.resize(3 * total), dead.resize(total), in_cavity.assign(total, 0);
        free_list.clear();
        vert_tri.assign((size_t)n, -1);
        edge_from.assign((size_t)n + 1, -1);
        edge_to.assign((size_t)n + 1, -1);
        n_dup = 0;
        std::vector<int> ghost_of((size_t)K, -1);
        std::vector<std::vector<int>> freed((size_t)K);
        on_threads(K, [&](int j) {
            const amt_delaunay& d = part[(size_t)j];
            const std::vector<int>& mine = members[(size_t)j];
            const size_t nt = d.dead.size(), off = offs[(size_t)j];
            int ghost = -1;
            for (size_t t = 0; t < nt; ++t) {
                dead[off + t] = d.dead[t];
                if (d.dead[t]) {
                    freed[(size_t)j].push_back((int)(off + t));
                    v[3 * (off + t)] = v[3 * (off + t) + 1] = v[3 * (off + t) + 2] = -2;
                    adj[3 * (off + t)] = adj[3 * (off + t) + 1] = adj[3 * (off + t) + 2] = -1;
                    continue;
                }
                for (int k = 0; k < 3; ++k) {
                    const int w = d.v[3 * t + k], a = d.adj[3 * t + k];
                    v[3 * (off + t) + k] = w == kInf ? kInf : mine[(size_t)w];
                    adj[3 * (off + t) + k] = a < 0 ? -1 : (int)(off + a);
                    if (w == kInf) ghost = (int)(off + t);
                }
            }
            ghost_of[(size_t)j] = ghost;
            for (size_t i = 0; i < mine.size(); ++i)
                if (d.vert_tri[i] >= 0) vert_tri[(size_t)mine[i]] = (int)(off + d.vert_tri[i]);
        });
        for (int j = 0; j < K; ++j) {
            free_list.insert(free_list.end(), freed[(size_t)j].begin(), freed[(size_t)j].end());
            n_dup += part[(size_t)j].n_dup;
            stats.orient_wide += part_stats[(size_t)j].orient_wide, stats.orient_zero += part_stats[(size_t)j].orient_zero;
            stats.incircle_wide += part_stats[(size_t)j].incircle_wide, stats.incircle_zero += part_stats[(size_t)j].incircle_zero;
        }
        part.clear();
        lap("concatenate");
        if (check && !validate("before the joins")) return false;
        // join the strips, left to right
        int g = ghost_of[0];
        for (int j = 1; j < K; ++j) {
            if (g < 0 || ghost_of[(size_t)j] < 0) return false;
            g = join(g, ghost_of[(size_t)j]);
            if (g < 0) return false;
            if (check && !validate("after a join")) return false;
        }
        lap("join");
        if (debug) std::fprintf(stderr, "[delaunay] %d strips, %lld flips\n", K, (long long)n_flips);
        n_strips = K;
        n_threads = threads;
        last = ghost_inner(g);
        return true;
    }

    void compact() {
        static const bool debug = std::getenv("AMT_DELAUNAY_DEBUG") != nullptr;
        auto t_begin = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!debug) return;
            const auto t = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[delaunay] %-12s %.3f s\n", what, std::chrono::duration<double>(t - t_begin).count());
            t_begin = t;
        };
        const int nt = (int)dead.size();
        const int T = std::max(1, std::min(n_threads, nt / 65536));
        auto chunk = [&](int64_t count, int j) { return std::pair<int64_t, int64_t>(count * j / T, count * (j + 1) / T); };
        auto finite = [&](int t) { return !dead[t] && v[3 * t] != kInf && v[3 * t + 1] != kInf && v[3 * t + 2] != kInf; };
        slot_of.resize((size_t)nt);
        std::vector<int64_t> first((size_t)T + 1, 0);
        on_threads(T, [&](int j) {
            const auto r = chunk(nt, j);
            int64_t c = 0;
            for (int64_t t = r.first; t < r.second; ++t) c += finite((int)t) ? 1 : 0;
            first[(size_t)j + 1] = c;
        });
        for (int j = 0; j < T; ++j) first[(size_t)j + 1] += first[(size_t)j];
        tri.resize(3 * (size_t)first[(size_t)T]);
        on_threads(T, [&](int j) {
            const auto r = chunk(nt, j);
            int64_t at = first[(size_t)j];
            for (int64_t t = r.first; t < r.second; ++t) {
                if (!finite((int)t)) {
                    slot_of[(size_t)t] = -1;
                    continue;
                }
                slot_of[(size_t)t] = (int)at;
                tri[3 * (size_t)at] = v[3 * t], tri[3 * (size_t)at + 1] = v[3 * t + 1], tri[3 * (size_t)at + 2] = v[3 * t + 2];
                ++at;
            }
        });
        lap("triangles");
        nbr.resize(tri.size());
        on_threads(T, [&](int j) {
            const auto r = chunk(nt, j);
            for (int64_t t = r.first; t < r.second; ++t) {
                const int s0 = slot_of[(size_t)t];
                if (s0 < 0) continue;
                for (int k = 0; k < 3; ++k) {
                    const int n = adj[3 * t + k];
                    nbr[3 * (size_t)s0 + k] = n >= 0 ? slot_of[(size_t)n] : -1;
                }
            }
        });
        lap("neighbours");
    }

    // `tri` / `nbr` (the finite triangles, renumbered) are made when somebody asks for them — amt_delaunay_triangles, the vertex
    // lists —; point location alone ('linear') walks the build's own structure (v / adj: ghosts and dead slots skipped)
    bool compacted = false;
    int first_finite = -1;          // the finite triangle with the lowest slot: where every piece of targets starts its walk
    // vertex -> neighbouring vertices, made when somebody asks for them ('linear' never does)
    bool lists_made = false;
    void build_vertex_lists() {
        static const bool debug = std::getenv("AMT_DELAUNAY_DEBUG") != nullptr;
        auto t_begin = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!debug) return;
            const auto t = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[delaunay] %-12s %.3f s\n", what, std::chrono::duration<double>(t - t_begin).count());
            t_begin = t;
        };
        const int nt = (int)dead.size();
        const int T = std::max(1, std::min(n_threads, nt / 65536));
        auto chunk = [&](int64_t count, int j) { return std::pair<int64_t, int64_t>(count * j / T, count * (j + 1) / T); };
        // vertex -> neighbouring vertices (scipy.spatial.Delaunay.vertex_neighbor_vertices).  Every triangle lists its edges
        // counter-clockwise, so an inner edge {x, y} comes up once as x -> y and once as y -> x (from the triangle on its other
        // side); a hull edge comes up once and gets its reverse here: no duplicates.  Threads take pieces of the triangles and
        // count / place with atomic adds; every vertex's list is then sorted, so that the lists — and with them the order in
        // which the relaxation sums a point's neighbours — do not depend on how the triangulation was built.
        const int n = (int)p.size();
        const size_t m = tri.size() / 3;
        indptr.assign((size_t)n + 1, 0);
        auto tri_chunk = [&](int j) { return std::pair<size_t, size_t>(m * (size_t)j / (size_t)T, m * ((size_t)j + 1) / (size_t)T); };
        on_threads(T, [&](int j) {
            const auto r = tri_chunk(j);
            for (size_t t = r.first; t < r.second; ++t)
                for (int k = 0; k < 3; ++k) {
                    __atomic_fetch_add(&indptr[(size_t)tri[3 * t + (k + 1) % 3] + 1], (int64_t)1, __ATOMIC_RELAXED);      // v[k+1] -> v[k+2]
                    if (nbr[3 * t + k] < 0) __atomic_fetch_add(&indptr[(size_t)tri[3 * t + (k + 2) % 3] + 1], (int64_t)1, __ATOMIC_RELAXED);
                }
        });
        for (int i = 0; i < n; ++i) indptr[(size_t)i + 1] += indptr[(size_t)i];
        indices.resize((size_t)indptr[(size_t)n]);
        std::vector<int64_t> fill(indptr.begin(), indptr.end() - 1);
        on_threads(T, [&](int j) {
            const auto r = tri_chunk(j);
            for (size_t t = r.first; t < r.second; ++t)
                for (int k = 0; k < 3; ++k) {
                    const int x = tri[3 * t + (k + 1) % 3], y = tri[3 * t + (k + 2) % 3];
                    indices[(size_t)__atomic_fetch_add(&fill[(size_t)x], (int64_t)1, __ATOMIC_RELAXED)] = y;
                    if (nbr[3 * t + k] < 0) indices[(size_t)__atomic_fetch_add(&fill[(size_t)y], (int64_t)1, __ATOMIC_RELAXED)] = x;
                }
        });
        on_threads(T, [&](int j) {
            const auto r = chunk(n, j);
            for (int64_t i = r.first; i < r.second; ++i) std::sort(indices.begin() + indptr[(size_t)i], indices.begin() + indptr[(size_t)i + 1]);
        });
        lap("vertex CSR");
    }
};

namespace {
// (a handle is const to its readers; the lists are its cache.  One lock for all handles: the build is what takes the time)
// -> AMT_OK, or AMT_ENOMEM / AMT_EHIP when the lists could not be made (no memory for ~140 MB of lists at full frame size, no
// thread to be had): nothing escapes into the C ABI, and `lists_made` stays false so that a later call tries again
int ensure_compact(const amt_delaunay* d) {
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    amt_delaunay* m = const_cast<amt_delaunay*>(d);
    if (m->compacted) return AMT_OK;
    try {
        m->compact();
        m->compacted = true;
    } catch (const std::bad_alloc&) {
        m->tri.clear(), m->nbr.clear();
        return AMT_ENOMEM;
    } catch (...) {
        m->tri.clear(), m->nbr.clear();
        return AMT_EHIP;
    }
    return AMT_OK;
}

int ensure_vertex_lists(const amt_delaunay* d) {
    if (int rc = ensure_compact(d)) return rc;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    amt_delaunay* m = const_cast<amt_delaunay*>(d);
    if (m->lists_made) return AMT_OK;
    try {
        m->build_vertex_lists();
        m->lists_made = true;
    } catch (const std::bad_alloc&) {
        m->indptr.clear(), m->indices.clear();
        return AMT_ENOMEM;
    } catch (...) {
        m->indptr.clear(), m->indices.clear();
        return AMT_EHIP;
    }
    return AMT_OK;
}
}  // namespace

extern "C" {

int amt_delaunay_create_threads(const double* xy, int64_t n, int32_t threads, int64_t parallel_min, amt_delaunay** out) {
    if (xy == nullptr || out == nullptr || n < 3 || n > 2000000000ll) return AMT_EINVAL;
    *out = nullptr;
    amt_delaunay* d = new (std::nothrow) amt_delaunay();
    if (d == nullptr) return AMT_ENOMEM;
    if (threads <= 0) {
        static const int env_threads = [] {
            const char* e = std::getenv("AMT_DELAUNAY_THREADS");
            const int v = e ? std::atoi(e) : 0;
            return v > 0 ? v : amt_host_threads(16, nullptr, nullptr);      // (this rank's share of the host's cores)
        }();
        threads = env_threads;
    }
    if (parallel_min <= 0) {
        static const int64_t env_min = [] {
            const char* e = std::getenv("AMT_DELAUNAY_PARALLEL_MIN");
            const long long v = e ? std::atoll(e) : 0;
            return (int64_t)(v > 0 ? v : 200000);
        }();
        parallel_min = env_min;
    }
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
        bool built = false;
        try {
            built = d->build_parallel(threads, (int)std::min<int64_t>(parallel_min, 2000000000ll));
        } catch (const std::bad_alloc&) {
            built = false;
        }
        if (!built) {
            // (small inputs, one thread, a degenerate seam: the sequential build on a fresh structure)
            std::vector<pt> pts;
            pts.swap(d->p);
            *d = amt_delaunay();
            d->p.swap(pts);
            d->stats = predicate_stats{0, 0, 0, 0};
            g_stats = &d->stats;
            built = d->build();
        }
        g_stats = nullptr;
        if (!built) {
            delete d;
            return AMT_EINVAL;         // fewer than three points that are not collinear
        }
        // (no compaction here: see `compacted`)
        const int slots = (int)d->dead.size();
        for (int t = 0; t < slots && d->first_finite < 0; ++t)
            if (!d->dead[t] && d->v[3 * t] != kInf && d->v[3 * t + 1] != kInf && d->v[3 * t + 2] != kInf) d->first_finite = t;
        if (d->first_finite < 0) {
            delete d;
            return AMT_EINVAL;
        }
    } catch (const std::bad_alloc&) {
        g_stats = nullptr;
        delete d;
        return AMT_ENOMEM;
    }
    *out = d;
    return AMT_OK;
}

int amt_delaunay_create(const double* xy, int64_t n, amt_delaunay** out) { return amt_delaunay_create_threads(xy, n, 0, 0, out); }

int amt_delaunay_build_info(const amt_delaunay* d, int64_t* info2) {
    if (d == nullptr || info2 == nullptr) return AMT_EINVAL;
    info2[0] = d->n_strips, info2[1] = d->n_flips;
    return AMT_OK;
}

int amt_delaunay_destroy(amt_delaunay* d) {
    if (d == nullptr) return AMT_EINVAL;
    delete d;
    return AMT_OK;
}

int amt_delaunay_sizes(const amt_delaunay* d, int64_t* n_triangles, int64_t* n_neighbours, int64_t* n_duplicates) {
    if (d == nullptr) return AMT_EINVAL;
    if (n_triangles) {
        if (int rc = ensure_compact(d)) return rc;
        *n_triangles = (int64_t)(d->tri.size() / 3);
    }
    if (n_neighbours) {
        if (int rc = ensure_vertex_lists(d)) return rc;
        *n_neighbours = (int64_t)d->indices.size();
    }
    if (n_duplicates) *n_duplicates = d->n_dup;
    return AMT_OK;
}

int amt_delaunay_slots(const amt_delaunay* d, const int32_t** vertices, const uint8_t** dead, int64_t* n_slots) {
    if (d == nullptr || vertices == nullptr || dead == nullptr || n_slots == nullptr) return AMT_EINVAL;
    static_assert(sizeof(int) == sizeof(int32_t) && sizeof(char) == sizeof(uint8_t), "the structure's arrays as the ABI names them");
    *vertices = reinterpret_cast<const int32_t*>(d->v.data());
    *dead = reinterpret_cast<const uint8_t*>(d->dead.data());
    *n_slots = (int64_t)d->dead.size();
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
    if (int rc = ensure_compact(d)) return rc;
    if (simplices) std::memcpy(simplices, d->tri.data(), d->tri.size() * sizeof(int32_t));
    if (neighbours) std::memcpy(neighbours, d->nbr.data(), d->nbr.size() * sizeof(int32_t));
    return AMT_OK;
}

int amt_delaunay_vertex_neighbours(const amt_delaunay* d, int64_t* indptr, int32_t* indices) {
    if (d == nullptr || indptr == nullptr || indices == nullptr) return AMT_EINVAL;
    if (int rc = ensure_vertex_lists(d)) return rc;
    std::memcpy(indptr, d->indptr.data(), d->indptr.size() * sizeof(int64_t));
    std::memcpy(indices, d->indices.data(), d->indices.size() * sizeof(int32_t));
    return AMT_OK;
}

int amt_delaunay_locate(const amt_delaunay* d, const double* targets, int64_t m, int32_t* vertices, double* centroids,
                        uint8_t* has_neighbour) {
    if (d == nullptr || targets == nullptr || vertices == nullptr || centroids == nullptr || has_neighbour == nullptr || m < 0)
        return AMT_EINVAL;
    // The walk runs on the build's own structure (slots of v / adj: ghost triangles carry kInf, dead slots are never reached from a
    // live one) — the same triangles in the same order with the same corner order as the compacted lists, so the answers are those
    // a walk over `tri` / `nbr` gives (every piece starts at the finite triangle with the lowest slot = compacted triangle 0).
    const size_t nt = d->dead.size();
    if (nt == 0 || d->first_finite < 0) return AMT_EINVAL;
    const int* V = d->v.data();
    const int* A = d->adj.data();
    auto ghost = [&](int t) { return V[3 * (size_t)t] == kInf || V[3 * (size_t)t + 1] == kInf || V[3 * (size_t)t + 2] == kInf; };
    // (targets in pieces on threads: every piece walks on from its own previous target; the answers do not depend on the pieces —
    // a target on an edge or a vertex gets the triangle the walk from ITS predecessor reaches first, so pieces are cut at fixed
    // multiples of 1024 targets whatever the number of threads)
    const int64_t piece = 1024, pieces = (m + piece - 1) / piece;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(d->n_threads, pieces), 16));
    try {
    on_threads(T, [&](int part) {
    for (int64_t pc = part; pc < pieces; pc += T) {
    int t = d->first_finite;
    for (int64_t i = pc * piece; i < std::min(m, (pc + 1) * piece); ++i) {
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
            const int* w = &V[3 * (size_t)t];
            int k;
            for (k = 0; k < 3; ++k)
                if (orient2d(d->p[w[(k + 1) % 3]], d->p[w[(k + 2) % 3]], q) < 0) break;
            if (k == 3) {
                inside = true;
                break;
            }
            const int n = A[3 * (size_t)t + k];
            if (n < 0 || ghost(n)) break;
            t = n;
        }
        if (!inside) continue;
        const int* w = &V[3 * (size_t)t];
        for (int k = 0; k < 3; ++k) {
            vo[k] = w[k];
            const int n = A[3 * (size_t)t + k];
            if (n >= 0 && !ghost(n)) {
                const int* u = &V[3 * (size_t)n];
                has_neighbour[3 * i + k] = 1;
                centroids[6 * i + 2 * k] = (d->p[u[0]].x + d->p[u[1]].x + d->p[u[2]].x) / 3;
                centroids[6 * i + 2 * k + 1] = (d->p[u[0]].y + d->p[u[1]].y + d->p[u[2]].y) / 3;
            }
        }
    }
    }
    });
    } catch (const std::bad_alloc&) {
        return AMT_ENOMEM;          // (nothing of it may leave through the C ABI)
    } catch (...) {
        return AMT_EHIP;
    }
    return AMT_OK;
}

}  // extern "C"
