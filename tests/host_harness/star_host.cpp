// star_host.cpp -- TEST HARNESS, not product: compiles radfoam_amd/csrc/rf_star.hpp (the per-point Delaunay star the
// HIP kernels of rf_delaunay.hip run, one lane per point) for the host, so that its logic -- link surgery, the
// tree search, the filtered / exact predicates -- can be checked against Qhull and Python big integers without a
// GPU.  Only tests/ loads the library this builds; nothing under radfoam_amd/ does.
#include <cstdint>
#include <cstring>
#include <vector>

#define RF_STAR_FN static inline
#define RF_STAR_NOINLINE static __attribute__((noinline))
#define RF_STAR_NOUNROLL
#include "../../radfoam_amd/csrc/rf_star.hpp"

using namespace rf::star;

// what delaunay_star_kernel does per lane: seeds = the `knn` nearest points of the lane's 64-point kd-block, or
// the old neighbour list when one is given
template <int V, int T>
static int one_star(const float *pts, uint32_t n, const Tree &tr, uint32_t i, uint32_t knn, const uint32_t *old_adj,
                    const uint32_t *old_off, uint32_t *row, int stride, uint32_t *degree, uint8_t *hull,
                    uint32_t *visited, uint32_t *inserted) {
    static thread_local Star<V, T> s;
    star_reset(s, i, pts + 3 * (size_t)i);
    uint32_t seeds[256];
    int ns = 0;
    if (old_adj) {
        for (uint32_t e = old_off[i]; e < old_off[i + 1] && ns < 256; ++e) seeds[ns++] = old_adj[e];
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
    uint32_t vis = 0, ins = 0;
    star_build(s, tr, pts, seeds, ns, vis, ins);
    visited[i] = vis;
    inserted[i] = ins;
    if (s.status != kOk) return s.status;
    bool h;
    degree[i] = (uint32_t)star_neighbours(s, row, 1, &h);
    hull[i] = h;
    (void)stride;
    return kOk;
}

extern "C" {

int star_host_exact_orient(const float *p12) { return exact_orient(p12); }
int star_host_exact_insphere(const float *p15) { return exact_insphere(p15); }
int star_host_orient_sign(const float *p12) { return orient_sign(p12, p12 + 3, p12 + 6, p12 + 9); }
int star_host_insphere_sign(const float *p15) { return insphere_sign(p15, p15 + 3, p15 + 6, p15 + 9, p15 + 12); }

int star_host_delaunay(const float *pts, uint32_t n, const float *tree, uint32_t depth, uint32_t knn,
                       const uint32_t *old_adj, const uint32_t *old_off, uint32_t *rows, int stride,
                       uint32_t *degree, uint8_t *hull, int *status, uint32_t *visited, uint32_t *inserted) {
    Tree tr{tree, n, depth};
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : bad)
    for (uint32_t i = 0; i < n; ++i) {
        int st = stride >= 64 ? one_star<64, 124>(pts, n, tr, i, knn, old_adj, old_off, rows + (size_t)i * stride, stride,
                                                  degree, hull, visited, inserted)
                              : kOverflow;
        if (st == kOverflow && stride >= 250)
            st = one_star<250, 496>(pts, n, tr, i, knn, old_adj, old_off, rows + (size_t)i * stride, stride, degree,
                                    hull, visited, inserted);
        status[i] = st;
        bad += st != kOk;
    }
    return bad;
}

}  // extern "C"
