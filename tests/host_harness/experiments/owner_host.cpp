// owner_host.cpp -- EXPERIMENT (host only, not product, not part of the test suite): certify every tetrahedron once,
// with ONE exchange and no iteration.
//
//   pass 1  every star is built from its seeds and certifies, against the AABB tree, only the triangles it OWNS:
//           those whose tetrahedron (i,a,b,c) has i as its lowest vertex (the point at infinity counts as the highest);
//           points found are inserted as usual.  The other triangles stay open.
//   pass 2  every star, starting from a copy of its pass-1 state, closes its open triangles: a triangle owned by
//           vertex a is closed for free if a's pass-1 star holds the same tetrahedron with its certificate -- an empty
//           ball is a fact about the point set, whoever established it -- and goes to the tree otherwise.
// Exact by construction (every final triangle carries a certificate), no rounds, the stars of pass 1 are read-only in
// pass 2.  `extra_seeds` > 0 adds that many true nearest neighbours (tree search) to the seeds of a star.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <vector>

#define RF_STAR_FN static inline
#define RF_STAR_NOINLINE static __attribute__((noinline))
#define RF_STAR_NOUNROLL
#include "../../../radfoam_amd/csrc/rf_star.hpp"
#include "star_owner.hpp"

using namespace rf::star;
using S64 = Star<64, 124>;

static int slot_of(const S64 &s, uint32_t g) {
    for (int k = 0; k < S64::kV; ++k)
        if (s.v[k].use && s.v[k].g == g) return k;
    return -1;
}

static bool certified_tet(const S64 &s, uint32_t g0, uint32_t g1, uint32_t g2) {
    const int a = slot_of(s, g0), b = slot_of(s, g1), c = slot_of(s, g2);
    if (a < 0 || b < 0 || c < 0) return false;
    for (int t = 0; t < s.nt; ++t) {
        const int x = s.t[t].a, y = s.t[t].b, z = s.t[t].c;
        if ((x == a || y == a || z == a) && (x == b || y == b || z == b) && (x == c || y == c || z == c))
            return (s.t[t].f & kCertified) != 0;
    }
    return false;
}

// nearest `k` points by a best-first walk of the tree (plain k-NN, only used to study better seeding)
static int knn_tree(const Tree &tr, const float *pts, uint32_t self, int k, uint32_t *out, uint32_t &visited) {
    const float *p = pts + 3 * (size_t)self;
    std::vector<std::pair<float, uint32_t>> best;   // max-heap by distance
    auto worst = [&]() { return (int)best.size() < k ? 3.4e38f : best.front().first; };
    const uint32_t leaf_depth = tr.depth - kLeafBits;
    uint32_t depth = 0, vidx = 0, flip = 0;
    for (;;) {
        const uint32_t idx = vidx ^ flip, first = idx << (tr.depth - depth);
        bool descend = false;
        if (first < tr.n) {
            const float *nd = tree_node(tr, depth, idx);
            ++visited;
            if (box_dist2(nd, p[0], p[1], p[2]) < worst()) {
                if (depth < leaf_depth) descend = true;
                else {
                    const uint32_t end = std::min(first + (1u << kLeafBits), tr.n);
                    for (uint32_t j = first; j < end; ++j) {
                        if (j == self) continue;
                        const float dx = pts[3 * j] - p[0], dy = pts[3 * j + 1] - p[1], dz = pts[3 * j + 2] - p[2];
                        const float d = dx * dx + dy * dy + dz * dz;
                        if (d < worst()) {
                            if ((int)best.size() == k) { std::pop_heap(best.begin(), best.end()); best.pop_back(); }
                            best.emplace_back(d, j);
                            std::push_heap(best.begin(), best.end());
                        }
                    }
                }
            }
        }
        if (descend) {
            const uint32_t dim = depth % 3;
            const float *left = tree_node(tr, depth + 1, 2 * idx);
            const uint32_t rf = p[dim] > left[3 + dim] ? 1u : 0u;
            ++depth; vidx <<= 1; flip = (flip << 1) | rf;
            continue;
        }
        ++vidx;
        uint32_t up = (uint32_t)__builtin_ctz(vidx);
        up = up < depth ? up : depth;
        depth -= up; vidx >>= up; flip >>= up;
        if (depth == 0) break;
    }
    std::sort_heap(best.begin(), best.end());
    for (size_t i = 0; i < best.size(); ++i) out[i] = best[i].second;
    return (int)best.size();
}

static bool owned(const S64 &s, int t) {
    const uint32_t i = s.self;
    return s.v[s.t[t].a].g > i && s.v[s.t[t].b].g > i && s.v[s.t[t].c].g > i;   // infinity = 0xFFFFFFFF
}

extern "C" int owner_host(const float *pts, uint32_t n, const float *tree, uint32_t depth, uint32_t knn,
                          uint32_t extra_seeds, uint32_t *rows, int stride, uint32_t *degree, double *stats) {
    Tree tr{tree, n, depth};
    const HullSet all{nullptr, 0, 0xFFFFFFFFu};
    std::vector<S64> p1(n), p2(n);
    std::atomic<long> nodes1{0}, nodes2{0}, nodes_knn{0}, ins{0}, free_hits{0}, misses{0}, owned2{0}, failed{0};
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t i = 0; i < n; ++i) {
        S64 &s = p1[i];
        star_reset(s, i, pts + 3 * (size_t)i);
        uint32_t seeds[96];
        int ns = 0;
        uint32_t vis = 0, in = 0;
        if (extra_seeds) {
            ns = knn_tree(tr, pts, i, (int)extra_seeds, seeds, vis);
            nodes_knn += vis;
            vis = 0;
        } else {
            const uint32_t b0 = i & ~63u, b1 = b0 + 64 < n ? b0 + 64 : n;
            float d2[64];
            for (uint32_t k = b0; k < b1; ++k) {
                const float dx = pts[3 * k] - pts[3 * i], dy = pts[3 * k + 1] - pts[3 * i + 1], dz = pts[3 * k + 2] - pts[3 * i + 2];
                d2[k - b0] = k == i ? 3.4e38f : dx * dx + dy * dy + dz * dz;
            }
            for (uint32_t r = 0; r < knn; ++r) {
                int best = -1;
                for (uint32_t k = 0; k < b1 - b0; ++k)
                    if (d2[k] < 3.4e38f && (best < 0 || d2[k] < d2[best])) best = (int)k;
                if (best < 0) break;
                seeds[ns++] = b0 + (uint32_t)best;
                d2[best] = 3.4e38f;
            }
        }
        star_seed(s, pts, seeds, ns, in);
        // pass 1: owned triangles only
        while (s.status == kOk) {
            int t = -1;
            for (int k = 0; k < s.nt; ++k)
                if (!(s.t[k].f & kCertified) && owned(s, k)) { t = k; break; }
            if (t < 0) break;
            float q[3];
            const uint32_t j = star_search(s, tr, pts, t, all, q, vis);
            if (j == kInfinity) { s.t[t].f |= kCertified; continue; }
            if (star_insert(s, j, q) <= 0) { if (s.status == kOk) s.status = kBroken; break; }
            ++in;
        }
        nodes1 += vis;
        ins += in;
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t i = 0; i < n; ++i) {
        S64 &s = p2[i];
        s = p1[i];
        uint32_t vis = 0, in = 0;
        while (s.status == kOk) {
            int t = -1;
            for (int k = 0; k < s.nt; ++k)
                if (!(s.t[k].f & kCertified)) { t = k; break; }
            if (t < 0) break;
            const uint32_t g[3] = {s.v[s.t[t].a].g, s.v[s.t[t].b].g, s.v[s.t[t].c].g};
            const uint32_t lo = std::min(g[0], std::min(g[1], g[2]));
            if (lo < i) {
                // the owner's certificate, if it has one: the same tetrahedron seen from `lo` has vertices i and the
                // two others of this triangle
                const uint32_t o1 = g[0] == lo ? g[1] : g[0], o2 = g[2] == lo ? g[1] : g[2];
                if (p1[lo].status == kOk && certified_tet(p1[lo], i, o1, o2)) {
                    s.t[t].f |= kCertified;
                    free_hits++;
                    continue;
                }
                misses++;
            } else {
                owned2++;
            }
            float q[3];
            const uint32_t j = star_search(s, tr, pts, t, all, q, vis);
            if (j == kInfinity) { s.t[t].f |= kCertified; continue; }
            if (star_insert(s, j, q) <= 0) { if (s.status == kOk) s.status = kBroken; break; }
            ++in;
        }
        nodes2 += vis;
        ins += in;
        if (s.status != kOk) { failed++; degree[i] = 0; continue; }
        bool h;
        degree[i] = (uint32_t)star_neighbours(s, rows + (size_t)i * stride, 1, &h);
    }
    stats[0] = (double)nodes1 / n; stats[1] = (double)nodes2 / n; stats[2] = (double)nodes_knn / n; stats[3] = (double)ins / n;
    stats[4] = (double)free_hits / n; stats[5] = (double)misses / n; stats[6] = (double)owned2 / n; stats[7] = (double)failed;
    return 0;
}
