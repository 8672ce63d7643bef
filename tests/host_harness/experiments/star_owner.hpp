// star_owner.hpp -- RESEARCH CODE, not part of the product: the "every tetrahedron certified once" two-pass build of
// DESIGN.md section 7 (owner-certified stars), as header functions over radfoam_amd/csrc/rf_star.hpp.  Host-validated
// (tests/test_delaunay.py::test_host_owner_build_equals_qhull through tests/host_harness/star_host.cpp) and measured on
// the GPU in round 3 (profiles/r03/b_delaunay_owner_*.json): 1915 / 990 tree nodes per point instead of 2649, but
// 444 / 428 ms instead of 374 ms for the 2M-point foam from scratch (incremental: 258 / 265 against 267) -- the
// records' round trip through memory and the second launch cost more than the saved tree walks.  The kernels were
// therefore removed from the product library; this file keeps the algorithm for the record.
// Include after rf_star.hpp (same RF_STAR_FN / RF_STAR_NOINLINE / RF_STAR_NOUNROLL macros).
#pragma once

namespace rf {
namespace star {

// ---- every tetrahedron certified once (EXPERIMENTAL: host-validated; the default build above
// certifies a tetrahedron in the star of each of its four vertices) ------------------------------------------------------
//
//   pass 1  star_certify_owned: a star certifies only the triangles it owns -- those whose tetrahedron (i,a,b,c) has i
//           as its lowest vertex, the point at infinity counting as the highest -- and is then saved as a StarRecord;
//   pass 2  star_close: a star reloaded from its record closes its other triangles with the owner's certificate where
//           the owner's record holds the same tetrahedron (an empty ball is a fact about the point set, whoever
//           established it), and asks the tree otherwise.  Records are read-only in pass 2.

constexpr uint32_t kNoVertex = 0xFFFFFFFEu;

template <int V, int T>
struct StarRecord {   // what a star is, without what can be recomputed (coordinates, spheres, use counts)
    int32_t status;
    uint32_t nt;
    uint32_t vg[V];                    // global id per slot, kNoVertex = free; slot 0 = infinity
    uint8_t a[T], b[T], c[T], f[T];    // f: kCertified only
};

template <typename S>
RF_STAR_FN bool star_owns(const S &s, int t) {
    const uint32_t i = s.self;
    return s.v[s.t[t].a].g > i && s.v[s.t[t].b].g > i && s.v[s.t[t].c].g > i;
}

template <typename S, typename R>
RF_STAR_FN void star_save(const S &s, R &r) {
    r.status = s.status;
    r.nt = (uint32_t)s.nt;
    for (int k = 0; k < S::kV; ++k) r.vg[k] = s.v[k].use ? s.v[k].g : kNoVertex;
    for (int t = 0; t < s.nt; ++t) {
        r.a[t] = (uint8_t)s.t[t].a;
        r.b[t] = (uint8_t)s.t[t].b;
        r.c[t] = (uint8_t)s.t[t].c;
        r.f[t] = (uint8_t)(s.t[t].f & kCertified);
    }
}

template <typename S, typename R>
RF_STAR_FN void star_load(S &s, const R &r, uint32_t self, const float *pts) {
    star_reset(s, self, pts + 3 * (size_t)self);
    s.status = r.status;
    s.nt = (int)r.nt;
    for (int k = 1; k < S::kV; ++k) {
        const uint32_t g = r.vg[k];
        s.v[k].g = g;
        if (g != kNoVertex) {
            s.v[k].x = pts[3 * (size_t)g];
            s.v[k].y = pts[3 * (size_t)g + 1];
            s.v[k].z = pts[3 * (size_t)g + 2];
        }
    }
    for (int t = 0; t < s.nt; ++t) {
        s.t[t].a = (typename S::Index)r.a[t];
        s.t[t].b = (typename S::Index)r.b[t];
        s.t[t].c = (typename S::Index)r.c[t];
        ++s.v[r.a[t]].use;
        ++s.v[r.b[t]].use;
        ++s.v[r.c[t]].use;
    }
    for (int t = 0; t < s.nt; ++t) {
        set_sphere(s, t);
        s.t[t].f |= r.f[t];
    }
}

// does the record hold the tetrahedron (its owner, g0, g1, g2) with a certificate?
template <typename R>
RF_STAR_FN bool record_certifies(const R &r, int V, uint32_t g0, uint32_t g1, uint32_t g2) {
    if (r.status != kOk) return false;
    int a = -1, b = -1, c = -1;
    for (int k = 0; k < V; ++k) {
        const uint32_t g = r.vg[k];
        a = g == g0 ? k : a;
        b = g == g1 ? k : b;
        c = g == g2 ? k : c;
    }
    if (a < 0 || b < 0 || c < 0) return false;
    for (uint32_t t = 0; t < r.nt; ++t) {
        const int x = r.a[t], y = r.b[t], z = r.c[t];
        if ((x == a || y == a || z == a) && (x == b || y == b || z == b) && (x == c || y == c || z == c))
            return (r.f[t] & kCertified) != 0;
    }
    return false;
}

// pass 1: seeds, then certification of the owned triangles only
template <typename S>
RF_STAR_FN void star_certify_owned(S &s, const Tree &tr, const float *pts, const HullSet &hull, const uint32_t *seeds,
                                   int nseeds, uint32_t &visited, uint32_t &inserted) {
    star_seed(s, pts, seeds, nseeds, inserted);
    while (s.status == kOk) {
        int t = -1;
        for (int k = 0; k < s.nt; ++k)
            if (!(s.t[k].f & kCertified) && star_owns(s, k)) {
                t = k;
                break;
            }
        if (t < 0) break;
        float q[3];
        const uint32_t j = star_search(s, tr, pts, t, hull, q, visited);
        if (s.status != kOk) return;
        if (j == kInfinity) {
            s.t[t].f |= kCertified;
            continue;
        }
        if (star_insert(s, j, q) <= 0) {
            if (s.status == kOk) s.status = kBroken;
            return;
        }
        ++inserted;
    }
}

// pass 2: `certified(owner, g0, g1, g2)` answers from the owner's pass-1 record
template <typename S, typename Lookup>
RF_STAR_FN void star_close(S &s, const Tree &tr, const float *pts, const HullSet &hull, const Lookup &certified,
                           uint32_t &visited, uint32_t &inserted, uint32_t &closed_by_owner) {
    while (s.status == kOk) {
        int t = -1;
        for (int k = 0; k < s.nt; ++k)
            if (!(s.t[k].f & kCertified)) {
                t = k;
                break;
            }
        if (t < 0) break;
        const uint32_t g0 = s.v[s.t[t].a].g, g1 = s.v[s.t[t].b].g, g2 = s.v[s.t[t].c].g;
        const uint32_t lo = g0 < g1 ? (g0 < g2 ? g0 : g2) : (g1 < g2 ? g1 : g2);
        if (lo < s.self) {
            // seen from `lo`, the same tetrahedron has the vertices i and the two others of this triangle
            const uint32_t o1 = g0 == lo ? g1 : g0, o2 = g2 == lo ? g1 : g2;
            if (certified(lo, s.self, o1, o2)) {
                s.t[t].f |= kCertified;
                ++closed_by_owner;
                continue;
            }
        }
        float q[3];
        const uint32_t j = star_search(s, tr, pts, t, hull, q, visited);
        if (s.status != kOk) return;
        if (j == kInfinity) {
            s.t[t].f |= kCertified;
            continue;
        }
        if (star_insert(s, j, q) <= 0) {
            if (s.status == kOk) s.status = kBroken;
            return;
        }
        ++inserted;
    }
}

// the K nearest points of p_self by a walk of the tree (seeds for a star: better than the kd-block's, at ~130 nodes)
template <int K>
RF_STAR_FN int star_knn(const Tree &tr, const float *pts, uint32_t self, uint32_t *out, uint32_t &visited) {
    const float px = pts[3 * (size_t)self], py = pts[3 * (size_t)self + 1], pz = pts[3 * (size_t)self + 2];
    float bd[K];
    int n = 0;
    const uint32_t leaf_depth = tr.depth - kLeafBits;
    uint32_t depth = 0, vidx = 0, flip = 0;
    for (;;) {
        const uint32_t idx = vidx ^ flip, first = idx << (tr.depth - depth);
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
                        int pos = n < K ? n++ : K - 1;   // sorted insertion, nearest first
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
            ++depth;
            vidx <<= 1;
            flip = (flip << 1) | right_first;
            continue;
        }
        ++vidx;
        uint32_t up = (uint32_t)__builtin_ctz(vidx);
        up = up < depth ? up : depth;
        depth -= up;
        vidx >>= up;
        flip >>= up;
        if (depth == 0) break;
    }
    return n;
}


}  // namespace star
}  // namespace rf
