// rf_star.hpp -- the Delaunay star of one point, built on its own (no shared mesh, no atomics).
//
// What it replaces: the reference triangulates on the GPU by growing ONE global tetrahedral mesh
// (src/delaunay/delaunay.cu:273-370: sample_initial_tets -> growth_iteration* -> find_adjacency, with
// delete_violations.cu for the incremental case): every iteration sorts and de-duplicates global tet / face tables.
// The tracer consumes only the NEIGHBOUR LISTS of that mesh (point_adjacency, the Voronoi faces of each cell).
// Here every point computes its own list independently -- one lane, one star -- which needs no global structure at
// all:
//
//   * the star of p_i is kept as its link: a triangulated sphere whose vertices are the neighbours found so far and
//     whose triangles (a,b,c) stand for the tetrahedra (i,a,b,c); the point at infinity is an ordinary link vertex
//     (local slot 0), so hull points need no special casing: a "ghost" triangle (a,b,inf) is the hull facet (i,a,b);
//   * inserting a candidate q removes the triangles whose tetrahedron q conflicts with (q strictly inside the
//     circumsphere; for a ghost: strictly beyond the facet's plane) and fans the hole to q (Bowyer-Watson on the link);
//   * a triangle is FINAL once its open circumball (half-space) is known to hold no point at all.  That is asked of
//     the reference's own AABB tree (build_aabb_tree: an implicit balanced tree over the kd-ordered points) by a
//     stackless nearest-child-first traversal which returns the conflicting point closest to p_i, or none;
//   * candidates ("seeds") only make the queries cheap -- the nearest points of the lane's kd-block, or the previous
//     neighbour list on an incremental rebuild -- they never decide the result: every final triangle is certified.
//
// Since a Delaunay edge can only disappear when points are added, a rejected candidate never has to be looked at
// again, and the loop "take an uncertified triangle, query, insert or certify" terminates with exactly the Delaunay
// neighbours of p_i in the full point set.
//
// Predicates: a float4 sphere cached per triangle settles all but the points within a few 1e-6 of the sphere; those
// go to the fp64 determinant with a forward error bound, and what that cannot sign is evaluated EXACTLY in 384-bit
// integer arithmetic (fp32 inputs are dyadic rationals: scaled to a common grid they are 62-bit integers).  The
// reference carries Shewchuk's expansion arithmetic for the same purpose (src/delaunay/shewchuk.cuh, predicate.cuh).
//
// The includer defines RF_STAR_FN / RF_STAR_NOINLINE (function qualifiers) and RF_STAR_NOUNROLL (loop pragma): the
// HIP translation unit makes them __device__, tests/host_harness compiles the very same text for the host to check
// the logic without a GPU.
#pragma once

#include <math.h>
#include <stdint.h>

#if !defined(RF_STAR_FN) || !defined(RF_STAR_NOINLINE) || !defined(RF_STAR_NOUNROLL)
#error "define RF_STAR_FN, RF_STAR_NOINLINE and RF_STAR_NOUNROLL before including rf_star.hpp"
#endif

#ifndef RF_STAR_ANY   // true while any lane of the wave says so (the kernels: a ballot); a star on its own: itself
#define RF_STAR_ANY(x) (x)
#endif
#ifndef RF_STAR_TRACE_QUERY   // instrumentation hook of the host harness: (tree nodes of the query, found a point)
#define RF_STAR_TRACE_QUERY(nodes, found) ((void)0)
#endif
#ifndef RF_STAR_TRACE_SWEEP   // ... (how a star's sweep ended: 0 done, 3 out of budget; points offered; rounds; tree nodes)
#define RF_STAR_TRACE_SWEEP(how, candidates, rounds, nodes) ((void)0)
#endif

namespace rf {
namespace star {

// The points a star's first candidates come from when it has no previous neighbour list: its 64-point block of the
// kd-order.  The last block of a cloud may hold one, two or three points (n % 64 in 1..3) -- fewer than a first
// tetrahedron needs --, so a short last block takes the LAST 64 points of the cloud instead (n >= 32 always; below 64
// points the one block is the cloud).  Shared by the kernels and the host harness.
RF_STAR_FN void seed_window(uint32_t n, uint32_t i, uint32_t &first, uint32_t &count) {
    first = i & ~63u;
    count = n - first < 64u ? n - first : 64u;
    if (count < 64u && n >= 64u) {
        first = n - 64u;
        count = 64u;
    }
}


constexpr uint32_t kInfinity = 0xFFFFFFFFu;   // global id of the vertex at infinity; also "no point"
constexpr int kLeafBits = 2;                  // the search tests buckets of 4 consecutive points directly

enum Status : int {
    kOk = 0,
    kOverflow = 1,     // more link vertices / triangles than this instance holds (retried with the large instance)
    kDegenerate = 2,   // no non-coplanar starting tetrahedron among the seeds
    kBroken = 3,       // the link stopped being a sphere (cospherical input: the caller perturbs, like the reference)
    kDuplicate = 4,    // another point has the same coordinates
    kPending = 5,      // a query ran out of budget: the star is redone in the second pass (see HullSet)
};

// Where the expensive queries go.  Nearly all queries touch a few dozen tree nodes, but a few per thousand stars --
// those at the rim of the cloud -- ask about regions the tree cannot bound: the half-space behind a facet of the
// convex hull, or the ball of a flat tetrahedron between rim points, which is so large that it acts like one.  Such a
// region holds no point, yet the boxes of the tree poke into it all along that side of the cloud: 10^5 nodes for one
// query, in one lane, while the rest of the launch has long finished.  So the first pass gives every query `budget`
// tree nodes (a query with a conflicting point nearby finds it within that) and parks the star when a query runs
// out with nothing found; the second pass (delaunay_star_coop_kernel) gives each parked star a whole wave, the 64
// lanes testing 64 boxes or points of one query at a time.  There, ghost queries do not use the tree at all: whatever
// lies beyond a plane, a vertex of the convex hull lies beyond it too, and the hull's vertices are among `ids` -- the
// points whose first-pass star kept a ghost or did not finish.
struct HullSet {
    const uint32_t *ids;   // null in the first pass
    uint32_t count;
    uint32_t budget;       // first pass: tree nodes per query (0xFFFFFFFF = unbounded)
};

enum TriFlag : uint8_t { kCertified = 1, kMarked = 2, kSlow = 4, kGhost = 8 };

// reference layout of build_aabb_tree (src/aabb_tree/aabb_tree.cuh:22-30): level d (2^d nodes, node k covering the
// points [k << (depth-d), (k+1) << (depth-d))) starts at node index 2^depth - 2^(d+1); a node is {min[3], max[3]}.
struct Tree {
    const float *nodes;
    uint32_t n;       // points
    uint32_t depth;   // log2(pow2_round_up(n))
};

RF_STAR_FN const float *tree_node(const Tree &tr, uint32_t d, uint32_t k) {
    return tr.nodes + 6 * ((size_t)(1u << tr.depth) - (size_t)(1u << (d + 1)) + k);
}

// ---- exact arithmetic ---------------------------------------------------------------------------------------------

// Everything in this section is the rare path (a few predicates in a million): it is written for a small
// register footprint -- values live in memory, loops stay rolled, helpers are real calls -- so that the kernels
// around it keep their occupancy.
struct Big {   // 384-bit two's complement
    uint64_t w[6];
};

RF_STAR_FN void mul64(uint64_t a, uint64_t b, uint64_t &hi, uint64_t &lo) {
    const uint64_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
    const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    const uint64_t mid = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
    lo = (p00 & 0xFFFFFFFFull) | (mid << 32);
    hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}

RF_STAR_NOINLINE void big_set(Big *r, int64_t v) {
    r->w[0] = (uint64_t)v;
    const uint64_t fill = v < 0 ? ~0ull : 0ull;
    RF_STAR_NOUNROLL
    for (int k = 1; k < 6; ++k) r->w[k] = fill;
}

// r = a + b or a - b (r may alias a or b)
RF_STAR_NOINLINE void big_addsub(Big *r, const Big *a, const Big *b, int subtract) {
    uint64_t carry = subtract ? 1 : 0;
    const uint64_t flip = subtract ? ~0ull : 0ull;
    RF_STAR_NOUNROLL
    for (int k = 0; k < 6; ++k) {
        const uint64_t x = a->w[k], y = b->w[k] ^ flip;
        const uint64_t s = x + y;
        const uint64_t c1 = s < x;
        const uint64_t s2 = s + carry;
        const uint64_t c2 = s2 < s;
        r->w[k] = s2;
        carry = c1 | c2;
    }
}

// r = low 384 bits of a * b: exact whenever the product fits (r must not alias a or b)
RF_STAR_NOINLINE void big_mul(Big *r, const Big *a, const Big *b) {
    RF_STAR_NOUNROLL
    for (int k = 0; k < 6; ++k) r->w[k] = 0;
    RF_STAR_NOUNROLL
    for (int i = 0; i < 6; ++i) {
        uint64_t carry = 0;
        const uint64_t ai = a->w[i];
        RF_STAR_NOUNROLL
        for (int j = 0; i + j < 6; ++j) {
            uint64_t hi, lo;
            mul64(ai, b->w[j], hi, lo);
            const uint64_t s = r->w[i + j] + lo;
            const uint64_t c1 = s < lo;
            const uint64_t s2 = s + carry;
            const uint64_t c2 = s2 < carry;
            r->w[i + j] = s2;
            carry = hi + c1 + c2;
        }
    }
}

RF_STAR_NOINLINE int big_sign(const Big *a) {
    if (a->w[5] >> 63) return -1;
    uint64_t any = 0;
    RF_STAR_NOUNROLL
    for (int k = 0; k < 6; ++k) any |= a->w[k];
    return any ? 1 : 0;
}

// fp32 -> mantissa * 2^exponent
RF_STAR_FN void decompose(float f, int32_t &m, int32_t &e) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    const int32_t be = (int32_t)((u >> 23) & 0xFF);
    int32_t frac = (int32_t)(u & 0x7FFFFF);
    if (be == 0) {
        e = -149;
    } else {
        frac |= 0x800000;
        e = be - 150;
    }
    m = (u >> 31) ? -frac : frac;
}

// `count` coordinates onto one integer grid (62 bits).  LIMIT OF THE "EXACT" PATH: a coordinate whose last bit lies
// more than 38 binary places below the last bit of the coarsest coordinate of the SAME predicate is snapped to that
// predicate's grid (truncated) -- e.g. a point at 1e-9 tested together with points around 1e3.  For such inputs the
// predicates are exact for the snapped coordinates only, and because the grid depends on which points share the
// predicate, two stars may sign the same configuration differently; that surfaces as unmatched edges in the symmetry
// check (TriangulationFailedError), never as a silently wrong list.  A float32 foam whose coordinates span fewer than
// 38 + 24 binary orders of magnitude (every scene the tracer can render: its fp16 face offsets need far less) never
// gets there.  Widening Big to the 277-bit exponent spread of fp32 would remove the limit at ~3x the rare path's cost.
RF_STAR_NOINLINE void to_grid(const float *c, int count, int64_t *out) {
    int32_t emax = -100000, emin = 100000;
    RF_STAR_NOUNROLL
    for (int k = 0; k < count; ++k) {
        int32_t m, e;
        decompose(c[k], m, e);
        if (m != 0) {
            emax = e > emax ? e : emax;
            emin = e < emin ? e : emin;
        }
    }
    const int32_t elo = emin > emax - 38 ? emin : emax - 38;
    RF_STAR_NOUNROLL
    for (int k = 0; k < count; ++k) {
        int32_t m, e;
        decompose(c[k], m, e);
        const int32_t sh = e - elo;
        int64_t v = 0;
        if (m != 0) {
            if (sh >= 0) v = (int64_t)m << sh;
            else if (sh > -40) v = (int64_t)m >> (-sh);
        }
        out[k] = v;
    }
}

// r = a * b - c * d
RF_STAR_NOINLINE void big_minor(Big *r, int64_t a, int64_t b, int64_t c, int64_t d) {
    Big x, y, t;
    big_set(&x, a);
    big_set(&y, b);
    big_mul(r, &x, &y);
    big_set(&x, c);
    big_set(&y, d);
    big_mul(&t, &x, &y);
    big_addsub(r, r, &t, 1);
}

// r = det[x; y; z]
RF_STAR_NOINLINE void big_det3(Big *r, const int64_t *x, const int64_t *y, const int64_t *z) {
    Big m, f, t;
    big_set(r, 0);
    RF_STAR_NOUNROLL
    for (int k = 0; k < 3; ++k) {   // cofactors of x[k]: cyclic, so every sign is +
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        big_minor(&m, y[k1], z[k2], y[k2], z[k1]);
        big_set(&f, x[k]);
        big_mul(&t, &f, &m);
        big_addsub(r, r, &t, 0);
    }
}

// sign of det[a-i; b-i; c-i], exact.  p = {i, a, b, c} as 12 floats.
RF_STAR_NOINLINE int exact_orient(const float *p) {
    int64_t g[12];
    to_grid(p, 12, g);
    RF_STAR_NOUNROLL
    for (int k = 3; k < 12; ++k) g[k] -= g[k % 3];
    Big r;
    big_det3(&r, g + 3, g + 6, g + 9);
    return big_sign(&r);
}

// sign of the 4x4 determinant with rows (X, |X|^2), X = a-i, b-i, c-i, q-i, exact.  p = {i, a, b, c, q}.
RF_STAR_NOINLINE int exact_insphere(const float *p) {
    int64_t g[15];
    to_grid(p, 15, g);
    RF_STAR_NOUNROLL
    for (int k = 3; k < 15; ++k) g[k] -= g[k % 3];
    const int64_t *row = g + 3;   // rows A, B, C, Q
    Big r, lift, d, t, x, y;
    big_set(&r, 0);
    // expansion along the lifted column: row v has sign (-1)^(v + 3), its minor the other three rows in order
    RF_STAR_NOUNROLL
    for (int v = 0; v < 4; ++v) {
        big_set(&lift, 0);
        RF_STAR_NOUNROLL
        for (int k = 0; k < 3; ++k) {
            big_set(&x, row[3 * v + k]);
            big_set(&y, row[3 * v + k]);
            big_mul(&t, &x, &y);
            big_addsub(&lift, &lift, &t, 0);
        }
        const int o0 = v == 0 ? 1 : 0, o1 = v <= 1 ? 2 : 1, o2 = v <= 2 ? 3 : 2;
        big_det3(&d, row + 3 * o0, row + 3 * o1, row + 3 * o2);
        big_mul(&t, &lift, &d);
        big_addsub(&r, &r, &t, (v & 1) ? 0 : 1);
    }
    return big_sign(&r);
}

// ---- fp64 filters ---------------------------------------------------------------------------------------------------

struct D3 {
    double x, y, z;
};

RF_STAR_FN D3 diff(float ax, float ay, float az, float bx, float by, float bz) {
    return D3{(double)ax - (double)bx, (double)ay - (double)by, (double)az - (double)bz};
}

RF_STAR_FN double det3(const D3 &x, const D3 &y, const D3 &z, double &perm) {
    const double a = y.y * z.z, b = y.z * z.y, c = y.x * z.z, d = y.z * z.x, e = y.x * z.y, f = y.y * z.x;
    perm = fabs(x.x) * (fabs(a) + fabs(b)) + fabs(x.y) * (fabs(c) + fabs(d)) + fabs(x.z) * (fabs(e) + fabs(f));
    return x.x * (a - b) - x.y * (c - d) + x.z * (e - f);
}

// rounding: every difference, product and sum below rounds once; a term of the 3x3 expansion carries at most 6
// roundings, the 4x4 one at most 16 (relative to the sum of absolute values of its terms, "perm"); 2^-53 per
// rounding.  The bounds leave a factor of two on top of that count.
constexpr double kOrientBound = 2.0e-15;     // > 2 * 8 * 2^-53
constexpr double kInsphereBound = 8.0e-15;   // > 2 * 32 * 2^-53

// sign of det[a-i; b-i; c-i]
RF_STAR_FN int orient_sign(const float *pi, const float *pa, const float *pb, const float *pc) {
    const D3 A = diff(pa[0], pa[1], pa[2], pi[0], pi[1], pi[2]);
    const D3 B = diff(pb[0], pb[1], pb[2], pi[0], pi[1], pi[2]);
    const D3 C = diff(pc[0], pc[1], pc[2], pi[0], pi[1], pi[2]);
    double perm;
    const double d = det3(A, B, C, perm);
    if (fabs(d) > kOrientBound * perm) return d > 0 ? 1 : -1;
    const float p[12] = {pi[0], pi[1], pi[2], pa[0], pa[1], pa[2], pb[0], pb[1], pb[2], pc[0], pc[1], pc[2]};
    return exact_orient(p);
}

// sign of the insphere determinant (negative = q strictly inside the sphere through i, a, b, c when
// det[a-i; b-i; c-i] > 0)
RF_STAR_FN int insphere_sign(const float *pi, const float *pa, const float *pb, const float *pc, const float *pq) {
    const D3 A = diff(pa[0], pa[1], pa[2], pi[0], pi[1], pi[2]);
    const D3 B = diff(pb[0], pb[1], pb[2], pi[0], pi[1], pi[2]);
    const D3 C = diff(pc[0], pc[1], pc[2], pi[0], pi[1], pi[2]);
    const D3 Q = diff(pq[0], pq[1], pq[2], pi[0], pi[1], pi[2]);
    const double la = A.x * A.x + A.y * A.y + A.z * A.z, lb = B.x * B.x + B.y * B.y + B.z * B.z;
    const double lc = C.x * C.x + C.y * C.y + C.z * C.z, lq = Q.x * Q.x + Q.y * Q.y + Q.z * Q.z;
    double pabc, pabq, pacq, pbcq;
    const double dabc = det3(A, B, C, pabc), dabq = det3(A, B, Q, pabq);
    const double dacq = det3(A, C, Q, pacq), dbcq = det3(B, C, Q, pbcq);
    const double det = (lq * dabc - lc * dabq) + (lb * dacq - la * dbcq);
    const double perm = lq * pabc + lc * pabq + lb * pacq + la * pbcq;
    if (fabs(det) > kInsphereBound * perm) return det > 0 ? 1 : -1;
    const float p[15] = {pi[0], pi[1], pi[2], pa[0], pa[1], pa[2], pb[0], pb[1], pb[2],
                         pc[0], pc[1], pc[2], pq[0], pq[1], pq[2]};
    return exact_insphere(p);
}

// ---- the star -------------------------------------------------------------------------------------------------------

// One record per link vertex / link triangle.
struct Vert {
    uint32_t g;      // global id; slot 0 of a star is the point at infinity
    float x, y, z;
    uint32_t use;    // triangles using the slot; 0 = free
};

template <typename Idx>
struct TriT {
    Idx a, b, c;           // link triangle (a,b,c) with det[a-i; b-i; c-i] > 0
    uint8_t f;             // TriFlag bits
    float sx, sy, sz, sr;  // finite: circumcentre relative to p_i and squared radius; ghost: outward normal of the facet
};

// V link vertices, T link triangles (a closed link of v vertices has 2v - 4), Idx wide enough for a vertex slot;
// HOLE = room of the list of hole triangles an insertion may be given (0: the insertion scans the flags instead)
template <int V, int T, typename Idx = uint8_t, int HOLE = 0>
struct Star {
    static constexpr int kV = V, kT = T, kHole = HOLE;
    using Index = Idx;
    uint32_t self;
    int status;
    int nt;
    float p[3];
    Vert v[V];
    TriT<Idx> t[T];
};

template <typename S>
RF_STAR_FN void star_reset(S &s, uint32_t self, const float *p) {
    s.self = self;
    s.status = kOk;
    s.nt = 0;
    s.p[0] = p[0];
    s.p[1] = p[1];
    s.p[2] = p[2];
    for (int k = 0; k < S::kV; ++k) s.v[k].use = 0;
    s.v[0].g = kInfinity;
    s.v[0].x = s.v[0].y = s.v[0].z = 0.0f;
}

template <typename S>
RF_STAR_FN void vertex_xyz(const S &s, int slot, float *out) {
    out[0] = s.v[slot].x;
    out[1] = s.v[slot].y;
    out[2] = s.v[slot].z;
}

// the two finite vertices (u, v) of a ghost triangle, rotated so that infinity comes last
template <typename S>
RF_STAR_FN void ghost_edge(const S &s, int t, int &u, int &v) {
    const int a = s.t[t].a, b = s.t[t].b, c = s.t[t].c;
    if (c == 0) {
        u = a;
        v = b;
    } else if (a == 0) {
        u = b;
        v = c;
    } else {
        u = c;
        v = a;
    }
}

template <typename S>
RF_STAR_FN void set_sphere(S &s, int t) {
    const int a = s.t[t].a, b = s.t[t].b, c = s.t[t].c;
    uint8_t flags = 0;
    if (a == 0 || b == 0 || c == 0) {
        int u, v;
        ghost_edge(s, t, u, v);
        const D3 A = diff(s.v[u].x, s.v[u].y, s.v[u].z, s.p[0], s.p[1], s.p[2]);
        const D3 B = diff(s.v[v].x, s.v[v].y, s.v[v].z, s.p[0], s.p[1], s.p[2]);
        const double nx = A.y * B.z - A.z * B.y, ny = A.z * B.x - A.x * B.z, nz = A.x * B.y - A.y * B.x;
        const double nm = fmax(fabs(nx), fmax(fabs(ny), fabs(nz)));
        const double am = fmax(fabs(A.x), fmax(fabs(A.y), fabs(A.z)));
        const double bm = fmax(fabs(B.x), fmax(fabs(B.y), fabs(B.z)));
        flags = kGhost;
        if (!(nm > 1e-6 * am * bm)) flags |= kSlow;   // i, u, v nearly collinear: the float normal means nothing
        // scale does not matter for a half-space: keep the normal in float range
        const double sc = nm > 0 ? 1.0 / nm : 0.0;
        s.t[t].sx = (float)(nx * sc);
        s.t[t].sy = (float)(ny * sc);
        s.t[t].sz = (float)(nz * sc);
        s.t[t].sr = -1.0f;
    } else {
        const D3 A = diff(s.v[a].x, s.v[a].y, s.v[a].z, s.p[0], s.p[1], s.p[2]);
        const D3 B = diff(s.v[b].x, s.v[b].y, s.v[b].z, s.p[0], s.p[1], s.p[2]);
        const D3 C = diff(s.v[c].x, s.v[c].y, s.v[c].z, s.p[0], s.p[1], s.p[2]);
        const D3 bc{B.y * C.z - B.z * C.y, B.z * C.x - B.x * C.z, B.x * C.y - B.y * C.x};
        const D3 ca{C.y * A.z - C.z * A.y, C.z * A.x - C.x * A.z, C.x * A.y - C.y * A.x};
        const D3 ab{A.y * B.z - A.z * B.y, A.z * B.x - A.x * B.z, A.x * B.y - A.y * B.x};
        const double det = A.x * bc.x + A.y * bc.y + A.z * bc.z;
        const double la = A.x * A.x + A.y * A.y + A.z * A.z, lb = B.x * B.x + B.y * B.y + B.z * B.z;
        const double lc = C.x * C.x + C.y * C.y + C.z * C.z;
        const double pm = (fabs(A.x) + fabs(A.y) + fabs(A.z)) * (fabs(B.x) + fabs(B.y) + fabs(B.z)) *
                          (fabs(C.x) + fabs(C.y) + fabs(C.z));
        // relative error of the fp64 determinant (and with it of the centre): ~ 8 roundings of 2^-53 over det / pm
        const double kappa = det > 0 ? 1e-15 * pm / det : 1.0;
        if (kappa < 0.05) {
            const double h = 0.5 / det;
            const double cx = (la * bc.x + lb * ca.x + lc * ab.x) * h;
            const double cy = (la * bc.y + lb * ca.y + lc * ab.y) * h;
            const double cz = (la * bc.z + lb * ca.z + lc * ab.z) * h;
            double r2 = cx * cx + cy * cy + cz * cz;
            if (kappa > 1e-8) {
                // a sliver: the centre is too vague for the float filter -- every conflict test of this triangle takes
                // the determinant -- but a ball padded by its uncertainty still bounds the search
                flags |= kSlow;
                r2 *= (1.0 + 4.0 * kappa) * (1.0 + 4.0 * kappa);
            }
            s.t[t].sx = (float)cx;
            s.t[t].sy = (float)cy;
            s.t[t].sz = (float)cz;
            s.t[t].sr = (float)r2;
            if (!(s.t[t].sr < 3.0e38f)) {
                flags |= kSlow;
                s.t[t].sr = 3.4e38f;
            }
        } else {   // numerically flat: no ball at all
            flags |= kSlow;
            s.t[t].sx = s.t[t].sy = s.t[t].sz = 0.0f;
            s.t[t].sr = 3.4e38f;
        }
    }
    s.t[t].f = flags;
}

// does the point q (global coordinates) conflict with triangle t?  Exact.
template <typename S>
RF_STAR_FN bool conflict(const S &s, int t, const float *q) {
    const float qx = q[0] - s.p[0], qy = q[1] - s.p[1], qz = q[2] - s.p[2];
    const uint8_t f = s.t[t].f;
    if (f & kGhost) {
        if (!(f & kSlow)) {
            const float tx = s.t[t].sx * qx, ty = s.t[t].sy * qy, tz = s.t[t].sz * qz;
            const float d = tx + ty + tz;
            const float u = 4e-6f * (fabsf(tx) + fabsf(ty) + fabsf(tz)) + 1e-37f;
            if (d > u) return true;
            if (d < -u) return false;
        }
        int a, b;
        ghost_edge(s, t, a, b);
        float pa[3], pb[3];
        vertex_xyz(s, a, pa);
        vertex_xyz(s, b, pb);
        return orient_sign(s.p, pa, pb, q) > 0;
    }
    if (!(f & kSlow)) {
        const float dx = qx - s.t[t].sx, dy = qy - s.t[t].sy, dz = qz - s.t[t].sz;
        const float d2 = dx * dx + dy * dy + dz * dz, r2 = s.t[t].sr;
        const float u = 4e-6f * (d2 + r2) + 1e-37f;
        if (d2 > r2 + u) return false;
        if (d2 < r2 - u) return true;
    }
    float pa[3], pb[3], pc[3];
    vertex_xyz(s, s.t[t].a, pa);
    vertex_xyz(s, s.t[t].b, pb);
    vertex_xyz(s, s.t[t].c, pc);
    return insphere_sign(s.p, pa, pb, pc, q) < 0;
}

template <typename S>
RF_STAR_FN bool has_directed_edge(const S &s, int t, int u, int v) {
    const int a = s.t[t].a, b = s.t[t].b, c = s.t[t].c;
    return (a == u && b == v) || (b == u && c == v) || (c == u && a == v);
}

// first tetrahedron (i, a, b, c) + its three ghosts.  false: the four points are coplanar.
template <typename S>
RF_STAR_FN bool star_init(S &s, uint32_t ga, const float *pa, uint32_t gb, const float *pb, uint32_t gc,
                          const float *pc) {
    const int o = orient_sign(s.p, pa, pb, pc);
    if (o == 0) return false;
    const float *q1 = o > 0 ? pa : pb, *q2 = o > 0 ? pb : pa;
    s.v[1].g = o > 0 ? ga : gb;
    s.v[2].g = o > 0 ? gb : ga;
    s.v[3].g = gc;
    s.v[1].x = q1[0]; s.v[1].y = q1[1]; s.v[1].z = q1[2];
    s.v[2].x = q2[0]; s.v[2].y = q2[1]; s.v[2].z = q2[2];
    s.v[3].x = pc[0]; s.v[3].y = pc[1]; s.v[3].z = pc[2];
    // (1,2,3), (2,1,inf), (3,2,inf), (1,3,inf): every directed edge once, its reverse once
    const int tri[4][3] = {{1, 2, 3}, {2, 1, 0}, {3, 2, 0}, {1, 3, 0}};
    for (int t = 0; t < 4; ++t) {
        s.t[t].a = (typename S::Index)tri[t][0];
        s.t[t].b = (typename S::Index)tri[t][1];
        s.t[t].c = (typename S::Index)tri[t][2];
    }
    s.nt = 4;
    s.v[0].use = 3;
    s.v[1].use = s.v[2].use = s.v[3].use = 3;
    for (int t = 0; t < 4; ++t) set_sphere(s, t);
    return true;
}

// What conflict() decides from the cached sphere alone, without a branch: +1 q conflicts with triangle t, -1 it does
// not, 0 the float filter cannot tell (exact_conflict then).  qx, qy, qz = q - p_i.
template <typename S>
RF_STAR_FN int filter_conflict(const S &s, int t, float qx, float qy, float qz) {
    const uint8_t f = s.t[t].f;
    const float sx = s.t[t].sx, sy = s.t[t].sy, sz = s.t[t].sz, sr = s.t[t].sr;
    const float tx = sx * qx, ty = sy * qy, tz = sz * qz;
    const float dg = tx + ty + tz;
    const float ug = 4e-6f * (fabsf(tx) + fabsf(ty) + fabsf(tz)) + 1e-37f;
    const float dx = qx - sx, dy = qy - sy, dz = qz - sz;
    const float d2 = dx * dx + dy * dy + dz * dz;
    const float ub = 4e-6f * (d2 + sr) + 1e-37f;
    const bool ghost = (f & kGhost) != 0;
    const bool in = ghost ? dg > ug : d2 < sr - ub;
    const bool out = ghost ? dg < -ug : d2 > sr + ub;
    return (f & kSlow) ? 0 : (in ? 1 : (out ? -1 : 0));
}

// ... and what it decides when the filter cannot: the determinants (fp64 with an error bound, then exact)
template <typename S>
RF_STAR_FN bool exact_conflict(const S &s, int t, const float *q) {
    if (s.t[t].f & kGhost) {
        int a, b;
        ghost_edge(s, t, a, b);
        float pa[3], pb[3];
        vertex_xyz(s, a, pa);
        vertex_xyz(s, b, pb);
        return orient_sign(s.p, pa, pb, q) > 0;
    }
    float pa[3], pb[3], pc[3];
    vertex_xyz(s, s.t[t].a, pa);
    vertex_xyz(s, s.t[t].b, pb);
    vertex_xyz(s, s.t[t].c, pc);
    return insphere_sign(s.p, pa, pb, pc, q) < 0;
}

// lowest free vertex slot (never 0: the point at infinity), or -1.  The scan looks at 64 slots per trip with neither a
// branch nor a store inside the trip, so that its loads are in flight together: a loop that leaves at the first hit waits
// for every record in turn, and in the kernels a record is a round trip to L2 or beyond (the stars live in scratch
// memory).  The same form of the search for the first uncertified triangle was measured and is no faster (the
// certified ones come first, the loop leaves early): profiles/r06.
template <typename S>
RF_STAR_FN int free_slot(const S &s) {
    for (int base = 0; base < S::kV; base += 64) {
        const int count = S::kV - base < 64 ? S::kV - base : 64;
        unsigned long long free_mask = 0ull;
        for (int k = 0; k < count; ++k) free_mask |= (unsigned long long)(s.v[base + k].use == 0) << k;
        if (base == 0) free_mask &= ~1ull;
        if (free_mask) return base + __builtin_ctzll(free_mask);
    }
    return -1;
}

// Bowyer-Watson on the link, in two steps so that a block of threads can share the first one.
// star_mark: flag the triangles q conflicts with; returns how many, and lists the first `cap` of them (ascending) in
// `hole`.  The kernels' stars live in scratch memory, where a load is a round trip to L2 or beyond: the first loop
// classifies 64 triangles at a time by the float filter with neither a store nor a branch in it, so that the loads of
// different triangles are in flight together (a loop that flags as it goes waits for every triangle in turn); the few
// the filter cannot decide and the flags follow.
template <typename S>
RF_STAR_FN int star_mark(S &s, const float *q, uint32_t *hole, int cap) {
    const float qx = q[0] - s.p[0], qy = q[1] - s.p[1], qz = q[2] - s.p[2];
    int marked = 0;
    for (int base = 0; base < s.nt; base += 64) {
        const int count = s.nt - base < 64 ? s.nt - base : 64;
        unsigned long long in = 0ull, unsure = 0ull;
        for (int k = 0; k < count; ++k) {
            const int c = filter_conflict(s, base + k, qx, qy, qz);
            in |= (unsigned long long)(c > 0) << k;
            unsure |= (unsigned long long)(c == 0) << k;
        }
        while (unsure) {
            const int k = __builtin_ctzll(unsure);
            unsure &= unsure - 1ull;
            if (exact_conflict(s, base + k, q)) in |= 1ull << k;
        }
        while (in) {
            const int k = __builtin_ctzll(in);
            in &= in - 1ull;
            s.t[base + k].f |= kMarked;
            if (marked < cap) hole[marked] = (uint32_t)(base + k);
            ++marked;
        }
    }
    return marked;
}

// star_apply: the flagged triangles (`marked` of them; `hole` = their indices if the caller collected them, else
// null; `ascending`: the list is in ascending order) come out and the hole is fanned to q.  Returns `marked`, or -1 on
// failure.
template <typename S>
RF_STAR_FN int star_apply(S &s, uint32_t gq, const float *q, int marked, const uint32_t *hole, bool ascending = false) {
    const int nt0 = s.nt;
    const int slot = free_slot(s);
    if (slot < 0) {
        s.status = kOverflow;
        return -1;
    }
    s.v[slot].g = gq;
    s.v[slot].x = q[0];
    s.v[slot].y = q[1];
    s.v[slot].z = q[2];
    // every directed edge of the hole whose reverse is not in the hole lies on its boundary: fan it to q
    int nt = nt0;
    const int outer = hole ? marked : nt0;
    for (int o = 0; o < outer; ++o) {
        const int t = hole ? (int)hole[o] : o;
        if (!(s.t[t].f & kMarked)) continue;
        const int v3[4] = {s.t[t].a, s.t[t].b, s.t[t].c, s.t[t].a};
        for (int e = 0; e < 3; ++e) {
            const int u = v3[e], v = v3[e + 1];
            bool inner = false;
            for (int i2 = 0; i2 < outer && !inner; ++i2) {
                const int t2 = hole ? (int)hole[i2] : i2;
                inner = (s.t[t2].f & kMarked) && t2 != t && has_directed_edge(s, t2, v, u);
            }
            if (inner) continue;
            if (nt >= S::kT) {
                s.status = kOverflow;
                return -1;
            }
            s.t[nt].a = (typename S::Index)u;
            s.t[nt].b = (typename S::Index)v;
            s.t[nt].c = (typename S::Index)slot;
            s.t[nt].f = 0;
            ++s.v[u].use;
            ++s.v[v].use;
            ++s.v[slot].use;
            ++nt;
        }
    }
    const int added = nt - nt0;
    for (int t = nt0; t < nt; ++t) set_sphere(s, t);
    // drop the hole: move the last live triangle into every marked slot
    int vanished = 0;   // link vertices that were interior to the hole
    if (hole && ascending) {
        // from the highest marked slot down: whatever lies above it is live by then, so the last triangle can go straight
        // into it -- `marked` steps instead of a pass over all the triangles
        for (int o = marked - 1; o >= 0; --o) {
            const int t = (int)hole[o];
            vanished += (--s.v[s.t[t].a].use == 0) + (--s.v[s.t[t].b].use == 0) + (--s.v[s.t[t].c].use == 0);
            --nt;
            if (t != nt) s.t[t] = s.t[nt];
        }
    } else {
        int t = 0;
        while (t < nt) {
            if (!(s.t[t].f & kMarked)) {
                ++t;
                continue;
            }
            vanished += (--s.v[s.t[t].a].use == 0) + (--s.v[s.t[t].b].use == 0) + (--s.v[s.t[t].c].use == 0);
            --nt;
            if (t != nt) s.t[t] = s.t[nt];
        }
    }
    s.nt = nt;
    // a hole that is a disc of m triangles around k interior vertices has m + 2 - 2k boundary edges
    if (added != marked + 2 - 2 * vanished) {
        s.status = kBroken;
        return -1;
    }
    return marked;
}

// Returns the number of triangles removed (0: q is not a neighbour), -1 on failure.
template <typename S>
RF_STAR_FN int star_insert(S &s, uint32_t gq, const float *q) {
    // the hole as a list when it is small (nearly always: a handful of triangles), the flags otherwise
    constexpr int kList = S::kHole > 0 ? S::kHole : 24;
    uint32_t hole[kList];
    const int marked = star_mark(s, q, hole, kList);
    if (marked == 0) return 0;
    return star_apply(s, gq, q, marked, marked <= kList ? hole : nullptr, true);
}

RF_STAR_FN float box_dist2(const float *nd, float x, float y, float z) {
    const float dx = fmaxf(fmaxf(nd[0] - x, x - nd[3]), 0.0f);
    const float dy = fmaxf(fmaxf(nd[1] - y, y - nd[4]), 0.0f);
    const float dz = fmaxf(fmaxf(nd[2] - z, z - nd[5]), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

// The region a query asks about (triangle t of the star of p_i), in the form the tree walk tests boxes and points with.
struct Region {
    float px, py, pz;      // p_i
    float nx, ny, nz;      // centre of the ball relative to p_i / outward normal of a ghost's plane
    float cx, cy, cz;      // centre of the ball, absolute
    float r2, rp2;         // squared radius as cached in the triangle / padded for the roundings of centre and box distance
    uint32_t g0, g1, g2;   // the triangle's own vertices
    uint8_t f;
    bool ghost, ball;
};

// Depth-first walk of the subtree rooted at node k0 of level d0 and then, one after the other, of the sibling subtrees
// of p_i's ancestors at the levels whose bits are set in `reach` (deepest first); nearer child first, boxes pruned
// against the region and against the best candidate so far; keeps the conflicting point closest to p_i in (best,
// best_id, out_q).  Returns false when the query's budget of tree nodes ran out (see HullSet).
template <typename S>
RF_STAR_FN bool search_subtree(S &s, const Tree &tr, const float *pts, int t, const Region &R, uint32_t d0, uint32_t k0,
                               uint32_t reach, float &best, uint32_t &best_id, float *out_q, uint32_t &visited,
                               uint32_t &spent, uint32_t budget) {
    const uint32_t leaf_depth = tr.depth - kLeafBits;
    uint32_t ld = 0, vidx = 0, flip = 0;   // coordinates inside the subtree: level and position below (d0, k0)
    for (;;) {
        const uint32_t depth = d0 + ld;
        const uint32_t idx = (k0 << ld) | (vidx ^ flip);
        const uint32_t first = idx << (tr.depth - depth);
        bool descend = false;
        if (first < tr.n) {
            const float *nd = tree_node(tr, depth, idx);
            ++visited;
            if (++spent > budget) return false;
            bool ok = box_dist2(nd, R.px, R.py, R.pz) < best;
            if (ok && R.ball) ok = box_dist2(nd, R.cx, R.cy, R.cz) <= R.rp2;
            if (ok && R.ghost && !(R.f & kSlow)) {
                // largest value of n . (x - p) over the box
                const float ax = R.nx > 0 ? nd[3] - R.px : nd[0] - R.px, ay = R.ny > 0 ? nd[4] - R.py : nd[1] - R.py;
                const float az = R.nz > 0 ? nd[5] - R.pz : nd[2] - R.pz;
                const float tx = R.nx * ax, ty = R.ny * ay, tz = R.nz * az;
                ok = tx + ty + tz > -4e-6f * (fabsf(tx) + fabsf(ty) + fabsf(tz));
            }
            if (ok) {
                if (depth < leaf_depth) {
                    descend = true;
                } else {
                    const uint32_t end = first + (1u << kLeafBits) < tr.n ? first + (1u << kLeafBits) : tr.n;
                    for (uint32_t k = first; k < end; ++k) {
                        if (k == s.self || k == R.g0 || k == R.g1 || k == R.g2) continue;
                        const float q[3] = {pts[3 * (size_t)k], pts[3 * (size_t)k + 1], pts[3 * (size_t)k + 2]};
                        const float dx = q[0] - R.px, dy = q[1] - R.py, dz = q[2] - R.pz;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        if (d2 == 0.0f && q[0] == R.px && q[1] == R.py && q[2] == R.pz) s.status = kDuplicate;
                        if (!(d2 < best)) continue;
                        // the float filter of conflict(), from the copy of the sphere this walk holds in registers
                        // (the star lives in scratch: every lane asks about another triangle here, and a scattered
                        // scratch read costs a cache line per dword)
                        if (R.ball && !(R.f & kSlow)) {
                            const float ex = dx - R.nx, ey = dy - R.ny, ez = dz - R.nz;
                            const float e2 = ex * ex + ey * ey + ez * ez, r2 = R.r2;
                            if (e2 > r2 + 4e-6f * (e2 + r2) + 1e-37f) continue;
                            if (!(e2 < r2 - (4e-6f * (e2 + r2) + 1e-37f)) && !conflict(s, t, q)) continue;
                        } else if (!conflict(s, t, q)) {
                            continue;
                        }
                        best = d2;
                        best_id = k;
                        out_q[0] = q[0];
                        out_q[1] = q[1];
                        out_q[2] = q[2];
                    }
                }
            }
        }
        if (descend) {
            // the children of a level-d node are separated along axis d % 3 (kd-order): nearer one first
            const uint32_t dim = depth % 3;
            const float *left = tree_node(tr, depth + 1, 2 * idx);
            const float pd = dim == 0 ? R.px : (dim == 1 ? R.py : R.pz);
            const uint32_t right_first = pd > left[3 + dim] ? 1u : 0u;
            ++ld;
            vidx <<= 1;
            flip = (flip << 1) | right_first;
            continue;
        }
        ++vidx;
        uint32_t up = (uint32_t)__builtin_ctz(vidx);
        up = up < ld ? up : ld;
        ld -= up;
        vidx >>= up;
        flip >>= up;
        if (ld == 0) {
            // this subtree is done: on to the next one the caller listed -- the sibling of the ancestor of p_i at the
            // deepest level whose bit is set in `reach` -- in the SAME loop: one trip is one node for every lane of a
            // wave, whichever subtree each of them is in (a loop per subtree made the lanes wait for each other at
            // every level: 2M points from scratch 502 ms against 355)
            if (reach == 0u) break;
            d0 = 31u - (uint32_t)__builtin_clz(reach);
            reach &= ~(1u << d0);
            k0 = (s.self >> (tr.depth - d0)) ^ 1u;
            vidx = flip = 0;
        }
    }
    return true;
}

// The point in strict conflict with triangle t that is closest to p_i, or kInfinity if there is none.  `visited`
// counts tree nodes (instrumentation).
//
// A finite ball passes through p_i, so what it can hold sits near p_i's own leaf of the tree: the walk goes BOTTOM-UP
// (RF_STAR_BOTTOM_UP, default) -- p_i's bucket first, then the sibling subtree of every ancestor of that bucket, from
// the leaves to the root.  Every point of the cloud lies in exactly one of these subtrees, so nothing is missed whatever
// the boxes look like (an incremental rebuild walks the boxes of points that moved after they were sorted: siblings may
// overlap then, which this does not care about).  Against the root-first walk it visits one box per level instead of
// two -- the ancestors themselves are never tested -- and, what matters more for a kernel that waits on dependent loads
// (waves parked 79 % of their cycles), the siblings' addresses follow from the lane's own index: the
// first phase tests all of them against the ball with independent loads and leaves a bit per level, the second walks
// into the few that the ball reaches.  Half-spaces (ghosts) and the balls of flat tetrahedra, which no box bounds,
// still start at the root.
#ifndef RF_STAR_BOTTOM_UP
#define RF_STAR_BOTTOM_UP 1
#endif
template <typename S>
RF_STAR_FN uint32_t star_search(S &s, const Tree &tr, const float *pts, int t, const HullSet &hull,
                                float *out_q, uint32_t &visited, uint32_t top_level = 1u) {
    Region R;
    R.f = s.t[t].f;
    R.ghost = (R.f & kGhost) != 0;
    R.ball = !R.ghost && s.t[t].sr < 3.0e38f;
    R.px = s.p[0];
    R.py = s.p[1];
    R.pz = s.p[2];
    R.g0 = s.v[s.t[t].a].g;
    R.g1 = s.v[s.t[t].b].g;
    R.g2 = s.v[s.t[t].c].g;
    if (R.ghost && hull.ids) {
        // second pass: whatever lies beyond a plane, a vertex of the convex hull lies beyond it too, and the hull's
        // vertices are among the few points whose first-pass star kept a ghost
        float best = 3.4e38f;
        uint32_t best_id = kInfinity;
        for (uint32_t h = 0; h < hull.count; ++h) {
            const uint32_t k = hull.ids[h];
            if (k == s.self || k == R.g0 || k == R.g1 || k == R.g2) continue;
            const float q[3] = {pts[3 * (size_t)k], pts[3 * (size_t)k + 1], pts[3 * (size_t)k + 2]};
            const float dx = q[0] - R.px, dy = q[1] - R.py, dz = q[2] - R.pz;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 == 0.0f && q[0] == R.px && q[1] == R.py && q[2] == R.pz) s.status = kDuplicate;
            if (!(d2 < best) || !conflict(s, t, q)) continue;
            best = d2;
            best_id = k;
            out_q[0] = q[0];
            out_q[1] = q[1];
            out_q[2] = q[2];
        }
        visited += hull.count;
        return best_id;
    }
    const uint32_t budget = hull.budget;
    uint32_t spent = 0;
    R.nx = s.t[t].sx;
    R.ny = s.t[t].sy;
    R.nz = s.t[t].sz;
    // ball in absolute coordinates, radius padded for the roundings of centre and box distance
    R.cx = R.px + R.nx;
    R.cy = R.py + R.ny;
    R.cz = R.pz + R.nz;
    R.rp2 = 3.4e38f;
    R.r2 = s.t[t].sr;
    if (R.ball) {
        const float r = sqrtf(R.r2);
        const float pad = 4e-7f * (fabsf(R.px) + fabsf(R.py) + fabsf(R.pz) + fabsf(R.nx) + fabsf(R.ny) + fabsf(R.nz) + r);
        const float rp = (r + pad) * 1.000002f;
        R.rp2 = rp * rp;
    }
    float best = 3.4e38f;
    uint32_t best_id = kInfinity;
    bool in_budget = true;
    const uint32_t leaf_depth = tr.depth - kLeafBits;
    if (RF_STAR_BOTTOM_UP && R.ball && leaf_depth >= 1u && leaf_depth < 32u) {
        // phase 1: which ancestors' siblings does the ball reach?  (independent loads: addresses from s.self alone)
        // (`top_level`: no ball of this star reaches the sibling of an ancestor above that level -- star_top_level)
        uint32_t reach = 0;
        top_level = top_level < 1u ? 1u : top_level;   // (d >= 0 would never end)
        for (uint32_t d = leaf_depth; d >= top_level; --d) {
            const uint32_t sib = (s.self >> (tr.depth - d)) ^ 1u;
            if ((sib << (tr.depth - d)) >= tr.n) continue;
            if (box_dist2(tree_node(tr, d, sib), R.cx, R.cy, R.cz) <= R.rp2) reach |= 1u << d;
        }
        visited += leaf_depth + 1u - top_level;
        spent += leaf_depth + 1u - top_level;
        // phase 2: p_i's own bucket, then the reached siblings from the nearest level up (one loop: see search_subtree)
        in_budget = search_subtree(s, tr, pts, t, R, leaf_depth, s.self >> kLeafBits, reach, best, best_id, out_q,
                                   visited, spent, budget);
    } else {
        in_budget = search_subtree(s, tr, pts, t, R, 0u, 0u, 0u, best, best_id, out_q, visited, spent, budget);
    }
    // out of budget (see HullSet): any conflicting point found so far will do; with none, the star waits for the
    // second pass
    if (!in_budget && best_id == kInfinity) s.status = kPending;
    return best_id;
}

// Seeds of a star that has no previous neighbour list (a from-scratch build): its RF_STAR_KNN_SEEDS nearest points from a
// walk of the tree; 0 = the nearest 12 of its 64-point block of the kd-order (rounds 2-3).  Measured (2 M / 500 k points
// from scratch, profiles/r04/k_delaunay_knn_seeds_ab.log): block seeds 340 / 110.5 ms, 2099 / 1966 tree nodes per point;
// 12 tree neighbours 318 / 108.7 ms, 2031 / 1896 nodes, 15.1 instead of 17.1 insertions per point; 16: 338 / 105.7; 24
// (what the owner-certified prototype of round 3 used): 372 / 111.1 -- every seed is inserted, and half of 24 are not
// Delaunay neighbours.  Seeds never decide the result.
#ifndef RF_STAR_KNN_SEEDS
#define RF_STAR_KNN_SEEDS 12
#endif

// The K nearest points of p_self (nearest first), for the seeds of a star that has no previous neighbour list: the same
// bottom-up walk as star_search -- p_self's bucket, then the sibling subtree of every ancestor, one flat loop -- with a
// box entered only while it is nearer than the K-th candidate so far.  Returns the number found (K unless n <= K).
template <int K>
RF_STAR_FN int star_knn_up(const Tree &tr, const float *pts, uint32_t self, uint32_t *out, uint32_t &visited) {
    const float px = pts[3 * (size_t)self], py = pts[3 * (size_t)self + 1], pz = pts[3 * (size_t)self + 2];
    float bd[K];
    int n = 0;
    const uint32_t leaf_depth = tr.depth - kLeafBits;
    uint32_t d0 = leaf_depth, k0 = self >> kLeafBits;        // the subtree being walked: p_self's bucket first
    uint32_t next = leaf_depth;                              // then the sibling of the ancestor at this level, upwards
    uint32_t ld = 0, vidx = 0, flip = 0;
    for (;;) {
        const uint32_t depth = d0 + ld;
        const uint32_t idx = (k0 << ld) | (vidx ^ flip);
        const uint32_t first = idx << (tr.depth - depth);
        bool descend = false;
        if (first < tr.n) {
            const float *nd = tree_node(tr, depth, idx);
            ++visited;
            if (n < K || box_dist2(nd, px, py, pz) < bd[K - 1]) {
                if (depth < leaf_depth) {
                    descend = true;
                } else {
                    const uint32_t end = first + (1u << kLeafBits) < tr.n ? first + (1u << kLeafBits) : tr.n;
                    for (uint32_t j = first; j < end; ++j) {
                        if (j == self) continue;
                        const float dx = pts[3 * (size_t)j] - px, dy = pts[3 * (size_t)j + 1] - py;
                        const float dz = pts[3 * (size_t)j + 2] - pz;
                        const float d = dx * dx + dy * dy + dz * dz;
                        if (n == K && !(d < bd[K - 1])) continue;
                        int pos = n < K ? n++ : K - 1;       // sorted insertion, nearest first
                        while (pos > 0 && bd[pos - 1] > d) {
                            bd[pos] = bd[pos - 1];
                            out[pos] = out[pos - 1];
                            --pos;
                        }
                        bd[pos] = d;
                        out[pos] = j;
                    }
                }
            }
        }
        if (descend) {
            const uint32_t dim = depth % 3;
            const float *left = tree_node(tr, depth + 1, 2 * idx);
            const float pd = dim == 0 ? px : (dim == 1 ? py : pz);
            const uint32_t right_first = pd > left[3 + dim] ? 1u : 0u;
            ++ld;
            vidx <<= 1;
            flip = (flip << 1) | right_first;
            continue;
        }
        ++vidx;
        uint32_t up = (uint32_t)__builtin_ctz(vidx);
        up = up < ld ? up : ld;
        ld -= up;
        vidx >>= up;
        flip >>= up;
        if (ld == 0) {
            if (next == 0u) break;
            d0 = next;
            k0 = (self >> (tr.depth - d0)) ^ 1u;
            --next;
            vidx = flip = 0;
        }
    }
    return n;
}

// Seeds nearest first (insertion sort; CAP >= nseeds).  A previous neighbour list comes in ascending index order, i.e.
// sweeping across the star from one side of the kd-order to the other: the link stays lopsided until the last seeds
// arrive and every insertion re-triangulates a large hole.  Nearest first, each seed changes a few triangles only.
#ifndef RF_STAR_SORT_SEEDS
#define RF_STAR_SORT_SEEDS 1
#endif
template <int CAP>
RF_STAR_FN void sort_seeds(const float *pts, const float *p, uint32_t *seeds, int nseeds) {
    float d2[CAP];
    for (int k = 0; k < nseeds; ++k) {
        const uint32_t g = seeds[k];
        const float dx = pts[3 * (size_t)g] - p[0], dy = pts[3 * (size_t)g + 1] - p[1], dz = pts[3 * (size_t)g + 2] - p[2];
        const float d = dx * dx + dy * dy + dz * dz;
        int pos = k;
        while (pos > 0 && d2[pos - 1] > d) {
            d2[pos] = d2[pos - 1];
            seeds[pos] = seeds[pos - 1];
            --pos;
        }
        d2[pos] = d;
        seeds[pos] = g;
    }
}

// Starting tetrahedron: the first seed triple that is not coplanar with p_i.  Sets kDegenerate if there is none.
template <typename S>
RF_STAR_FN void star_first_tet(S &s, const float *pts, const uint32_t *seeds, int nseeds, int *used) {
    used[0] = used[1] = used[2] = -1;
    for (int a = 0; a < nseeds && used[0] < 0; ++a)
        for (int b = a + 1; b < nseeds && used[0] < 0; ++b)
            for (int c = b + 1; c < nseeds; ++c) {
                const float *pa = pts + 3 * (size_t)seeds[a], *pb = pts + 3 * (size_t)seeds[b];
                const float *pc = pts + 3 * (size_t)seeds[c];
                if (star_init(s, seeds[a], pa, seeds[b], pb, seeds[c], pc)) {
                    used[0] = a;
                    used[1] = b;
                    used[2] = c;
                    break;
                }
            }
    if (used[0] < 0) s.status = kDegenerate;
}

// First tetrahedron + all the seeds (any order; nearest first is cheapest).
template <typename S>
RF_STAR_FN void star_seed(S &s, const float *pts, const uint32_t *seeds, int nseeds, uint32_t &inserted) {
    int used[3];
    star_first_tet(s, pts, seeds, nseeds, used);
    if (s.status != kOk) return;
    for (int k = 0; k < nseeds; ++k) {
        if (k == used[0] || k == used[1] || k == used[2]) continue;
        const float *q = pts + 3 * (size_t)seeds[k];
        if (q[0] == s.p[0] && q[1] == s.p[1] && q[2] == s.p[2]) {
            s.status = kDuplicate;
            return;
        }
        if (star_insert(s, seeds[k], q) < 0) return;
        ++inserted;
    }
}

// ---- the sweep: one range query certifies the triangles of a star whose balls it covers -----------------------------
// A circumball of the star passes through p_i, so it lies inside the ball of radius 2 r around p_i (r = its own radius).
// And the region the star's balls cover only shrinks: the triangle (u, v, q) an insertion fans from a boundary edge
// (u, v) of its hole has its ball inside ball(T1) u ball(T2), T1 the removed and T2 the kept triangle on that edge (the
// three spheres share the circle through p_i, u, v; q lies inside T1's and not inside T2's; half-spaces of ghosts are
// the limit case) -- which is also why a point that was offered to the link and did not conflict never conflicts later.
// So ONE walk of the tree that offers every point within R of p_i to the link certifies, when it is done, every
// triangle with 2 r <= R: a point inside such a ball lies within R, was offered, and conflicts with nothing that is
// left.  That replaces a tree walk per triangle (27 for an average star, plus one per insertion) by one walk that meets
// ~45 points, 16 of them the link's own vertices (scripts/model_star_churn.py has the numbers for a random foam).
// R is 2 max r, except that a ball more than kSweepSpread times the star's smallest does not count: ghosts (the convex
// hull), numerically flat triangles and the huge balls of the cloud's rim -- regions that reach far outside the cloud,
// where a ball around p_i would cover half of it -- keep their own queries (the per-triangle loop of star_build).
// The walk is taken in rounds: it pauses when its list is full, the list is offered to the link -- nearest subtrees
// first, so the star is close to final after the first round --, R shrinks to what the balls need now, and the walk
// goes on from where it stopped (whatever it passed was within the larger R).  Rounds keep the wave together: all lanes
// walk, then all lanes insert.  RF_STAR_SWEEP=0 compiles the sweep out (A/B).
#ifndef RF_STAR_SWEEP
#define RF_STAR_SWEEP 0   // measured out on the GPU (profiles/r06): 148-184 ms against 145 for 2 M points (see below)
#endif
#ifndef RF_STAR_SWEEP_CAP
#define RF_STAR_SWEEP_CAP 64
#endif
#ifndef RF_STAR_SWEEP_SPREAD
#define RF_STAR_SWEEP_SPREAD 25.0f   // (largest / smallest radius)^2 a star's balls may span and still all be swept
#endif
#ifndef RF_STAR_SWEEP_ON          // run-time switch of the host harness (tests compare both ways); a constant in the kernels
#define RF_STAR_SWEEP_ON 1
#endif
constexpr int kSweepCap = RF_STAR_SWEEP_CAP;
constexpr float kSweepSpread = RF_STAR_SWEEP_SPREAD;

RF_STAR_FN uint32_t vertex_hash(uint32_t g) { return (g * 0x9E3779B1u) >> 26; }

// squared radius of the sweep the star asks for now: (2 r)^2 of its largest ball that counts, with room for the
// roundings of the cached radius, of the boxes' and of the points' distances; 0: no finite triangle
template <typename S>
RF_STAR_FN float sweep_reach(const S &s) {
    float lo = 3.4e38f, hi = 0.0f;
    for (int t = 0; t < s.nt; ++t) {
        const float r = s.t[t].sr;   // ghosts: -1
        if (r >= 0.0f) {
            lo = fminf(lo, r);
            hi = fmaxf(hi, r);
        }
    }
    if (!(lo < 3.0e38f)) return 0.0f;
    float reach = 4.0f * fminf(hi, kSweepSpread * lo) * 1.0001f + 1e-37f;
#if defined(RF_STAR_SWEEP_FAR)
    // ... nor more than RF_STAR_SWEEP_FAR times the distance of the farthest link vertex (2 r >= that distance for every
    // ball at the vertex): the walk of a star with one large ball stays as short as its neighbours' in the wave
    float far2 = 0.0f;
    for (int k = 1; k < S::kV; ++k) {
        const float dx = s.v[k].x - s.p[0], dy = s.v[k].y - s.p[1], dz = s.v[k].z - s.p[2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        far2 = s.v[k].use != 0 ? fmaxf(far2, d2) : far2;
    }
    reach = fminf(reach, (float)(RF_STAR_SWEEP_FAR) * (float)(RF_STAR_SWEEP_FAR) * far2);
#endif
    return reach;
}

// where a sweep's walk stands: the subtree it is in (d0, k0), the position inside it, the ancestors' siblings still to
// visit (a bit per level), the tree nodes spent so far
struct SweepWalk {
    uint32_t d0, k0, ld, vidx, flip, reach, spent;
    bool done;
};

// Start: p_i's bucket, then the sibling subtree of every ancestor that a ball of squared radius r2 reaches (the walk of
// star_knn_up / search_subtree; independent loads: the addresses follow from s.self alone).
template <typename S>
RF_STAR_FN void sweep_begin(const S &s, const Tree &tr, float r2, SweepWalk &w, uint32_t &visited) {
    const uint32_t leaf_depth = tr.depth - kLeafBits;
    w.reach = 0;
    for (uint32_t d = leaf_depth; d >= 1u; --d) {
        const uint32_t sib = (s.self >> (tr.depth - d)) ^ 1u;
        if ((sib << (tr.depth - d)) >= tr.n) continue;
        if (box_dist2(tree_node(tr, d, sib), s.p[0], s.p[1], s.p[2]) <= r2) w.reach |= 1u << d;
    }
    visited += leaf_depth;
    w.spent = leaf_depth;
    w.d0 = leaf_depth;
    w.k0 = s.self >> kLeafBits;
    w.ld = w.vidx = w.flip = 0;
    w.done = false;
}

// The next points within sqrt(r2) of p_i (the link's own vertices among them) until the walk is done or `out` has no
// room for another bucket.  The leaf does nothing but measure and store: whatever a lane does rarely the wave does in
// nearly every trip, so telling the link's vertices from new points is left to the caller's (convergent) loop over the
// list.  Returns how many were written, or -1 when the walk has taken more than `budget` tree nodes.
template <typename S>
RF_STAR_FN int sweep_collect(S &s, const Tree &tr, const float *pts, float r2, SweepWalk &w, uint32_t *out, int cap,
                             uint32_t budget, uint32_t &visited) {
    const float px = s.p[0], py = s.p[1], pz = s.p[2];
    const uint32_t leaf_depth = tr.depth - kLeafBits;
    int count = 0;
    bool duplicate = false, over = false;
    while (!w.done && !over && count + (1 << kLeafBits) <= cap) {
        const uint32_t depth = w.d0 + w.ld;
        const uint32_t idx = (w.k0 << w.ld) | (w.vidx ^ w.flip);
        const uint32_t first = idx << (tr.depth - depth);
        bool descend = false;
        if (first < tr.n) {
            const float *nd = tree_node(tr, depth, idx);
            ++visited;
            over = ++w.spent > budget;
            if (!over && box_dist2(nd, px, py, pz) <= r2) {
                if (depth < leaf_depth) {
                    descend = true;
                } else {
                    const uint32_t end = first + (1u << kLeafBits) < tr.n ? first + (1u << kLeafBits) : tr.n;
                    for (uint32_t k = first; k < end; ++k) {
                        const float dx = pts[3 * (size_t)k] - px, dy = pts[3 * (size_t)k + 1] - py;
                        const float dz = pts[3 * (size_t)k + 2] - pz;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        if (k == s.self || !(d2 <= r2)) continue;
                        duplicate |= d2 == 0.0f;   // (or an underflow: looked at again below)
                        out[count++] = k;
                    }
                }
            }
        }
        if (descend) {
            const uint32_t dim = depth % 3;
            const float *left = tree_node(tr, depth + 1, 2 * idx);
            const float pd = dim == 0 ? px : (dim == 1 ? py : pz);
            const uint32_t right_first = pd > left[3 + dim] ? 1u : 0u;
            ++w.ld;
            w.vidx <<= 1;
            w.flip = (w.flip << 1) | right_first;
            continue;
        }
        ++w.vidx;
        uint32_t up = (uint32_t)__builtin_ctz(w.vidx);
        up = up < w.ld ? up : w.ld;
        w.ld -= up;
        w.vidx >>= up;
        w.flip >>= up;
        if (w.ld == 0) {
            if (w.reach == 0u) {
                w.done = true;
            } else {
                w.d0 = 31u - (uint32_t)__builtin_clz(w.reach);
                w.reach &= ~(1u << w.d0);
                w.k0 = (s.self >> (tr.depth - w.d0)) ^ 1u;
                w.vidx = w.flip = 0;
            }
        }
    }
    if (duplicate)
        for (int c = 0; c < count; ++c) {
            const float *q = pts + 3 * (size_t)out[c];
            if (q[0] == px && q[1] == py && q[2] == pz) s.status = kDuplicate;
        }
    return over ? -1 : count;
}

// is the ball of triangle t inside a swept reach of squared radius r2?
template <typename S>
RF_STAR_FN bool swept_ball(const S &s, int t, float r2) {
    const float r = s.t[t].sr;
    return r >= 0.0f && 4.0f * r * 1.0001f + 1e-37f <= r2;
}

// Offers every point within reach of the star's (ordinary) balls to the link and certifies the triangles whose balls
// that covers.  Returns the squared radius that was swept (0: nothing); triangles the star gets LATER are final too if
// their ball lies inside it.  The star may fail on the way (s.status).  The rounds are the WAVE's (RF_STAR_ANY): all
// lanes walk, then all lanes offer their lists.
template <typename S>
RF_STAR_FN float star_sweep(S &s, const Tree &tr, const float *pts, uint32_t budget, uint32_t &visited,
                            uint32_t &inserted) {
    float r2 = sweep_reach(s);
    bool more = r2 > 0.0f && tr.depth > (uint32_t)kLeafBits;
    bool swept = more;
    SweepWalk w;
    w.done = true;
    if (more) sweep_begin(s, tr, r2, w, visited);
    uint32_t cand[kSweepCap];
    int offered = 0, rounds = 0;
    while (RF_STAR_ANY(more)) {
        if (more) {
            ++rounds;
            const int n = sweep_collect(s, tr, pts, r2, w, cand, kSweepCap, budget, visited);
            if (s.status != kOk || n < 0) {   // failed / out of budget: what was inserted stands, nothing is certified here
                more = swept = false;
            } else {
                // the link's vertices are in the list too: a 64-bit filter over a hash of the id, the slots only on a
                // filter hit (a point this loop inserts is not a candidate again: the filter need not follow the link)
                unsigned long long filter = 0ull;
                int top = 1;
                for (int k = 1; k < S::kV; ++k) {   // (no branch: see free_slot)
                    const bool used = s.v[k].use != 0;
                    filter |= (unsigned long long)used << vertex_hash(s.v[k].g);
                    top = used ? k + 1 : top;
                }
                bool failed = false;
                for (int c = 0; c < n; ++c) {
                    const uint32_t k = cand[c];
                    bool vertex = failed;
                    if ((filter >> vertex_hash(k)) & 1ull)
                        for (int v = 1; v < top; ++v) vertex |= (s.v[v].use != 0) & (s.v[v].g == k);
                    if (!vertex) {
                        const float q[3] = {pts[3 * (size_t)k], pts[3 * (size_t)k + 1], pts[3 * (size_t)k + 2]};
                        const int r = star_insert(s, k, q);
                        failed = r < 0;   // overflow / broken: s.status says which
                        inserted += r > 0;
                    }
                }
                offered += n;
                if (failed) more = swept = false;
                else if (w.done) more = false;
                // the balls the star has now need less: the rest of the walk is shorter (never longer: what the walk
                // passed was measured against the reach of that time)
                else r2 = fminf(r2, sweep_reach(s));
            }
        }
    }
    RF_STAR_TRACE_SWEEP(swept ? 0 : 3, offered, rounds, w.spent);
    if (!swept || s.status != kOk) return 0.0f;
    // every point within sqrt(r2) of p_i has been offered: a ball inside that reach is empty
    for (int t = 0; t < s.nt; ++t)
        if (swept_ball(s, t, r2)) s.t[t].f |= kCertified;
    return r2;
}

// The highest level of the tree (smallest d >= 1) at which the sibling of p_i's ancestor lies within reach of ANY ball the
// star has or will have; 1 when the star has a ghost or a flat triangle, whose regions no ball around p_i bounds.  A ball
// passes through p_i, so it lies within 2 r of it, and the region the balls cover only shrinks as points go in (see the
// sweep above): what the ball of radius 2 max r around p_i does not reach now, no later query of this star has to test.
// A query of the bottom-up search tests the sibling of every ancestor against its own ball first -- 19 boxes for 2 M
// points, 27 queries a star: a third of the nodes the first pass visits; with this bound it tests the levels near the
// leaves that the star's neighbourhood spans.  In the kernels the bound is made the WAVE's (the lowest of its 64 stars,
// consecutive points of the kd-order: RF_STAR_UNIFORM_MIN, defined by the includer): the loop over the levels stays a loop
// of the wave with a scalar trip count.
#ifndef RF_STAR_UNIFORM_MIN
#define RF_STAR_UNIFORM_MIN(x) (x)
#endif
#ifndef RF_STAR_TOP_LEVEL
#define RF_STAR_TOP_LEVEL 1
#endif
template <typename S>
RF_STAR_FN uint32_t star_top_level(const S &s, const Tree &tr, uint32_t &visited) {
    if (s.v[0].use != 0 || tr.depth <= (uint32_t)kLeafBits) return 1u;
    float hi = 0.0f;
    for (int t = 0; t < s.nt; ++t) hi = fmaxf(hi, s.t[t].sr);
    if (!(hi < 1.0e37f)) return 1u;
    const float r2 = 4.0f * hi * 1.0001f + 1e-37f;
    const uint32_t leaf_depth = tr.depth - kLeafBits;
    uint32_t top = leaf_depth + 1u;
    for (uint32_t d = leaf_depth; d >= 1u; --d) {
        const uint32_t sib = (s.self >> (tr.depth - d)) ^ 1u;
        if ((sib << (tr.depth - d)) >= tr.n) continue;
        if (box_dist2(tree_node(tr, d, sib), s.p[0], s.p[1], s.p[2]) <= r2) top = d;
    }
    visited += leaf_depth;
    return top;
}

// Build the star from `nseeds` candidate points, then certify every triangle.
template <typename S>
RF_STAR_FN void star_build(S &s, const Tree &tr, const float *pts, const HullSet &hull,
                           const uint32_t *seeds, int nseeds, uint32_t &visited, uint32_t &inserted) {
    star_seed(s, pts, seeds, nseeds, inserted);
    if (s.status != kOk) return;
#if defined(RF_STAR_EXPERIMENT_STAGE) && RF_STAR_EXPERIMENT_STAGE == 1   // timing only: seeds, nothing certified
    return;
#endif
#if RF_STAR_SWEEP
    float swept = 0.0f;
    if (RF_STAR_SWEEP_ON) {
        swept = star_sweep(s, tr, pts, hull.budget, visited, inserted);
        if (s.status != kOk) return;
    }
#endif
    // The loop is the WAVE's: every lane whose star is still open takes one of its open triangles per trip, and the lanes
    // meet again at the end of the trip (RF_STAR_ANY: true while any lane of the wave says so; defined by the includer,
    // on the host a star is on its own).  Written with a `return` / `break` per lane instead, where the lanes of a wave
    // meet again is the compiler's choice, and it changed with unrelated edits: the same star code ran the first pass in
    // 163 or in 311 ms (profiles/r06/r_delaunay_stages_ab.log).
#if RF_STAR_TOP_LEVEL
    const uint32_t top_level = RF_STAR_UNIFORM_MIN(star_top_level(s, tr, visited));
#else
    const uint32_t top_level = 1u;
#endif
    bool open = true;
    while (RF_STAR_ANY(open)) {
        if (open) {
            int t = -1;
            for (int k = 0; k < s.nt; ++k)
                if (!(s.t[k].f & kCertified)) {
                    t = k;
                    break;
                }
            if (t < 0) {
                open = false;
            }
#if RF_STAR_SWEEP
            // a triangle that came with an insertion after the sweep: final if its ball lies inside what was swept
            else if (swept > 0.0f && swept_ball(s, t, swept)) {
                s.t[t].f |= kCertified;
            }
#endif
            else {
                float q[3];
                const uint32_t before = visited;
                const uint32_t j = star_search(s, tr, pts, t, hull, q, visited, top_level);
                RF_STAR_TRACE_QUERY(visited - before, j != kInfinity);
                if (s.status != kOk) {
                    open = false;
                } else if (j == kInfinity) {
                    s.t[t].f |= kCertified;
                } else if (star_insert(s, j, q) <= 0) {
                    if (s.status == kOk) s.status = kBroken;   // the search said "conflict", the insertion must agree
                    open = false;
                } else {
                    ++inserted;
                }
            }
        }
    }
}

// Neighbours in ascending order (finite link vertices); returns the count.  *hull = the star has a ghost.
template <typename S>
RF_STAR_FN int star_neighbours(const S &s, uint32_t *out, int stride, bool *hull) {
    int n = 0;
    for (int k = 1; k < S::kV; ++k) {
        if (s.v[k].use == 0) continue;
        const uint32_t g = s.v[k].g;
        int pos = n;
        while (pos > 0 && out[(size_t)(pos - 1) * stride] > g) {
            out[(size_t)pos * stride] = out[(size_t)(pos - 1) * stride];
            --pos;
        }
        out[(size_t)pos * stride] = g;
        ++n;
    }
    *hull = s.v[0].use != 0;
    return n;
}

}  // namespace star
}  // namespace rf
