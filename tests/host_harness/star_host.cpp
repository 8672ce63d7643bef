// star_host.cpp -- TEST HARNESS, not product: compiles radfoam_amd/csrc/rf_star.hpp (the per-point Delaunay star the
// HIP kernels of rf_delaunay.hip run, one lane per point) for the host, so that its logic -- link surgery, the
// tree search, the filtered / exact predicates -- can be checked against Qhull and Python big integers without a
// GPU.  Only tests/ loads the library this builds; nothing under radfoam_amd/ does.
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <vector>
#include <chrono>

// per-query trace (tree nodes, found-a-point) of the star being built, for the SIMT-cost model in star_host.py
#include <utility>
static thread_local std::vector<std::pair<uint32_t, int>> *g_trace = nullptr;
#define RF_STAR_TRACE_QUERY(nodes, found) \
    do {                                  \
        if (g_trace) g_trace->emplace_back((uint32_t)(nodes), (int)(found)); \
    } while (0)

// how the sweeps of the first-pass (small) stars ended: [0] done, [1] ghost, [2] flat triangle, [3] list / budget exceeded, [4] candidates of the done ones
static long long g_sweep_stats[5];
static thread_local uint32_t *g_sweep_rec = nullptr;   // star_host_sweep_trace: {how, offered, rounds, nodes} of the star being built
#define RF_STAR_TRACE_SWEEP(how, candidates, rounds, nodes)                    \
    do {                                                                       \
        __atomic_fetch_add(&g_sweep_stats[how], 1, __ATOMIC_RELAXED);          \
        if ((how) == 0) __atomic_fetch_add(&g_sweep_stats[4], (long long)(candidates), __ATOMIC_RELAXED); \
        if (g_sweep_rec) {                                                     \
            g_sweep_rec[0] = (uint32_t)(how);                                  \
            g_sweep_rec[1] = (uint32_t)(candidates);                           \
            g_sweep_rec[2] = (uint32_t)(rounds);                               \
            g_sweep_rec[3] = (uint32_t)(nodes);                                \
        }                                                                      \
    } while (0)
#define RF_STAR_SWEEP 1   // compiled in here whatever the kernels' default is: the tests run it both ways
static int g_star_sweep = 0;   // star_host_set_sweep: the sweep of rf_star.hpp on / off (off: what the kernels ship; tests run both)
#define RF_STAR_SWEEP_ON g_star_sweep
#define RF_STAR_FN static inline
#define RF_STAR_NOINLINE static __attribute__((noinline))
#define RF_STAR_NOUNROLL
#include "../../radfoam_amd/csrc/rf_star.hpp"
#include "experiments/star_owner.hpp"   // research code: the owner-certified two-pass build (not in the product)

using namespace rf::star;

static uint32_t *g_star_ns = nullptr;   // per-star build time (instrumentation)

// what delaunay_star_kernel / delaunay_star_big_kernel do per lane: seeds = the `knn` nearest points of the lane's
// 64-point kd-block, or the old neighbour list when one is given
template <typename StarT>
static int one_star(const float *pts, uint32_t n, const Tree &tr, const HullSet &hull, uint32_t i, uint32_t knn,
                    const uint32_t *old_adj, const uint32_t *old_off, uint32_t *row, uint32_t *degree, uint8_t *ghost,
                    uint32_t *visited, uint32_t *inserted) {
    static thread_local StarT s;
    constexpr int V = StarT::kV;
    star_reset(s, i, pts + 3 * (size_t)i);
    static thread_local uint32_t seeds[4096];
    int ns = 0;
    if (old_adj) {
        for (uint32_t e = old_off[i]; e < old_off[i + 1] && ns < V - 1; ++e) seeds[ns++] = old_adj[e];
        if (ns < 3) ns = 0;   // nothing usable in the previous list: the block's points, as gather_seeds does
        else if (RF_STAR_SORT_SEEDS) sort_seeds<V>(pts, pts + 3 * (size_t)i, seeds, ns);
    }
    if (ns == 0) {
        uint32_t b0, bc;
        seed_window(n, i, b0, bc);
        const uint32_t b1 = b0 + bc;
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
    const auto t0 = std::chrono::steady_clock::now();
    // as delaunay_star_kernel: a star without a previous list takes its seeds from a walk of the tree
    if (!old_adj && RF_STAR_KNN_SEEDS > 0) ns = star_knn_up<(RF_STAR_KNN_SEEDS > 0 ? RF_STAR_KNN_SEEDS : 1)>(tr, pts, i, seeds, vis);
    star_build(s, tr, pts, hull, seeds, ns, vis, ins);
    g_star_ns[i] += (uint32_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    visited[i] += vis;
    inserted[i] += ins;
    if (s.status != kOk) return s.status;
    bool h;
    degree[i] = (uint32_t)star_neighbours(s, row, 1, &h);
    ghost[i] = h;
    return kOk;
}

extern "C" {

int star_host_exact_orient(const float *p12) { return exact_orient(p12); }
int star_host_exact_insphere(const float *p15) { return exact_insphere(p15); }
int star_host_orient_sign(const float *p12) { return orient_sign(p12, p12 + 3, p12 + 6, p12 + 9); }
int star_host_insphere_sign(const float *p15) { return insphere_sign(p15, p15 + 3, p15 + 6, p15 + 9, p15 + 12); }

int star_host_delaunay(const float *pts, uint32_t n, const float *tree, uint32_t depth, uint32_t knn,
                       const uint32_t *old_adj, const uint32_t *old_off, uint32_t *rows, int stride,
                       uint32_t *degree, uint8_t *hull, int *status, uint32_t *visited, uint32_t *inserted,
                       uint32_t ghost_budget) {
    if (stride < 250) return -1;
    static std::vector<uint32_t> ns_store;
    ns_store.assign(n, 0);
    g_star_ns = ns_store.data();
    Tree tr{tree, n, depth};
    // first pass: the small instance, ghost queries on a budget
    const HullSet first{nullptr, 0, ghost_budget};
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t i = 0; i < n; ++i) {
        visited[i] = inserted[i] = 0;
        hull[i] = 0;
        status[i] = one_star<Star<64, 124>>(pts, n, tr, first, i, knn, old_adj, old_off, rows + (size_t)i * stride, degree,
                                      hull, visited, inserted);
    }
    // hull candidates: stars that kept a ghost or were parked; second pass: the large instance for parked and
    // overflowing stars, ghost queries through the candidate list
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < n; ++i)
        if (hull[i] || status[i] != kOk) ids.push_back(i);   // anything not known to be interior
    const HullSet second{ids.data(), (uint32_t)ids.size(), 0xFFFFFFFFu};
    int bad = 0;
    {
        size_t parked = 0, over = 0;
        for (uint32_t i = 0; i < n; ++i) parked += status[i] == kPending, over += status[i] == kOverflow;
        fprintf(stderr, "[star_host] first pass: %zu parked, %zu overflow, %zu hull candidates\n", parked, over, ids.size());
    }
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad)
    for (uint32_t i = 0; i < n; ++i) {
        if (status[i] == kPending || status[i] == kOverflow)
            status[i] = one_star<Star<250, 496>>(pts, n, tr, second, i, knn, old_adj, old_off, rows + (size_t)i * stride,
                                           degree, hull, visited, inserted);
        // third tier (delaunay_star_huge_kernel): hubs with up to 4095 neighbours
        if (status[i] == kOverflow && stride >= 4096)
            status[i] = one_star<Star<4096, 8188, uint16_t, 1024>>(pts, n, tr, second, i, knn, old_adj, old_off,
                                                                   rows + (size_t)i * stride, degree, hull, visited,
                                                                   inserted);
        bad += status[i] != kOk;
    }
    return bad;
}

// The experimental two-pass build (experiments/star_owner.hpp: star_certify_owned / star_close), as the owner kernels (removed from the product in round 3) of
// rf_delaunay.hip run it: pass 1 certifies owned triangles and saves records, pass 2 closes the rest from the owners'
// records or the tree; stars parked in either pass go to the large instance with the hull candidates, as in the
// default build.  tree_knn = 0: seeds from the kd-block; 24: the 24 nearest points by a tree walk.
int star_host_delaunay_owner(const float *pts, uint32_t n, const float *tree, uint32_t depth, uint32_t knn,
                             uint32_t tree_knn, uint32_t budget, const uint32_t *old_adj, const uint32_t *old_off,
                             uint32_t *rows, int stride, uint32_t *degree, uint8_t *hull, int *status, double *stats) {
    if (stride < 250) return -1;
    using Small = Star<64, 124>;
    using Rec = StarRecord<64, 124>;
    Tree tr{tree, n, depth};
    const HullSet first{nullptr, 0, budget};
    std::vector<Rec> rec(n);
    std::vector<uint32_t> ns_store(n, 0);
    g_star_ns = ns_store.data();
    double nodes1 = 0, nodes2 = 0, nodes_knn = 0, ins = 0, closed = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nodes1, nodes_knn, ins)
    for (uint32_t i = 0; i < n; ++i) {
        static thread_local Small s;
        star_reset(s, i, pts + 3 * (size_t)i);
        uint32_t seeds[64];
        int ns = 0;
        uint32_t vis = 0, in = 0;
        if (old_adj) {   // incremental: the previous lists
            for (uint32_t e = old_off[i]; e < old_off[i + 1] && ns < 63; ++e) seeds[ns++] = old_adj[e];
        } else if (tree_knn) {
            ns = star_knn<24>(tr, pts, i, seeds, vis);
            nodes_knn += vis;
            vis = 0;
        } else {
            uint32_t b0, bc;
        seed_window(n, i, b0, bc);
        const uint32_t b1 = b0 + bc;
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
        star_certify_owned(s, tr, pts, first, seeds, ns, vis, in);
        star_save(s, rec[i]);
        nodes1 += vis;
        ins += in;
    }
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nodes2, ins, closed)
    for (uint32_t i = 0; i < n; ++i) {
        hull[i] = 0;
        status[i] = rec[i].status;
        if (rec[i].status != kOk) continue;
        static thread_local Small s;
        star_load(s, rec[i], i, pts);
        uint32_t vis = 0, in = 0, cl = 0;
        star_close(s, tr, pts, first,
                   [&](uint32_t owner, uint32_t g0, uint32_t g1, uint32_t g2) { return record_certifies(rec[owner], 64, g0, g1, g2); },
                   vis, in, cl);
        nodes2 += vis;
        ins += in;
        closed += cl;
        status[i] = s.status;
        if (s.status != kOk) continue;
        bool h;
        degree[i] = (uint32_t)star_neighbours(s, rows + (size_t)i * stride, 1, &h);
        hull[i] = h;
    }
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < n; ++i)
        if (hull[i] || status[i] != kOk) ids.push_back(i);
    const HullSet second{ids.data(), (uint32_t)ids.size(), 0xFFFFFFFFu};
    std::vector<uint32_t> vis2(n, 0), ins2(n, 0);
    int bad = 0;
    long redone = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad, redone)
    for (uint32_t i = 0; i < n; ++i) {
        if (status[i] == kPending || status[i] == kOverflow) {
            status[i] = one_star<Star<250, 496>>(pts, n, tr, second, i, knn, old_adj, old_off, rows + (size_t)i * stride,
                                                 degree, hull, vis2.data(), ins2.data());
            redone++;
        }
        bad += status[i] != kOk;
    }
    g_star_ns = nullptr;
    stats[0] = nodes1 / n; stats[1] = nodes2 / n; stats[2] = nodes_knn / n; stats[3] = ins / n; stats[4] = closed / n;
    stats[5] = (double)redone;
    return bad;
}

// Per star of [first, first + count), first-pass instance: out[k] = {how the sweep ended (0 done, 3 out of budget, 9 none),
// points offered, rounds, tree nodes of the walk, queries of the per-triangle loop after it, their tree nodes (sum),
// the largest of them, insertions}.  For the lockstep cost model of the sweep (scripts/model_star_sweep.py).
int star_host_sweep_trace(const float *pts, uint32_t n, const float *tree, uint32_t depth, uint32_t knn, uint32_t budget,
                          const uint32_t *old_adj, const uint32_t *old_off, uint32_t first, uint32_t count, uint32_t *out) {
    Tree tr{tree, n, depth};
    const HullSet pass{nullptr, 0, budget};
    std::vector<uint32_t> row(4096), deg(n), vis(n), ins(n);
    std::vector<uint8_t> ghost(n);
    std::vector<uint32_t> ns_store(n, 0);
    g_star_ns = ns_store.data();
    for (uint32_t k = 0; k < count && first + k < n; ++k) {
        std::vector<std::pair<uint32_t, int>> tr_q;
        uint32_t *rec = out + 8 * (size_t)k;
        rec[0] = 9;
        rec[1] = rec[2] = rec[3] = 0;
        g_trace = &tr_q;
        g_sweep_rec = rec;
        ins[first + k] = 0;
        one_star<Star<64, 124>>(pts, n, tr, pass, first + k, knn, old_adj, old_off, row.data(), deg.data(), ghost.data(),
                                vis.data(), ins.data());
        g_trace = nullptr;
        g_sweep_rec = nullptr;
        rec[4] = (uint32_t)tr_q.size();
        rec[5] = rec[6] = 0;
        for (auto &q : tr_q) {
            rec[5] += q.first;
            rec[6] = q.first > rec[6] ? q.first : rec[6];
        }
        rec[7] = ins[first + k];
    }
    g_star_ns = nullptr;
    return 0;
}

const uint32_t *star_host_times() { return g_star_ns; }

void star_host_set_sweep(int on) { g_star_sweep = on; }
void star_host_sweep_stats(long long *out, int reset) {
    for (int k = 0; k < 5; ++k) {
        out[k] = g_sweep_stats[k];
        if (reset) g_sweep_stats[k] = 0;
    }
}

// First-pass query traces of the stars [first, first + count): out[k * cap + q] = tree nodes of query q of star k
// (bit 31: the query found a point), lengths[k] = number of queries.  For the lockstep cost model.
int star_host_trace(const float *pts, uint32_t n, const float *tree, uint32_t depth, uint32_t knn, uint32_t budget,
                    uint32_t first, uint32_t count, uint32_t cap, uint32_t *out, uint32_t *lengths) {
    Tree tr{tree, n, depth};
    const HullSet pass{nullptr, 0, budget};
    std::vector<uint32_t> row(4096), deg(n), vis(n), ins(n);
    std::vector<uint8_t> ghost(n);
    std::vector<uint32_t> ns_store(n, 0);
    g_star_ns = ns_store.data();
    for (uint32_t k = 0; k < count && first + k < n; ++k) {
        std::vector<std::pair<uint32_t, int>> tr_q;
        g_trace = &tr_q;
        one_star<Star<64, 124>>(pts, n, tr, pass, first + k, knn, nullptr, nullptr, row.data(), deg.data(), ghost.data(),
                                vis.data(), ins.data());
        g_trace = nullptr;
        lengths[k] = (uint32_t)(tr_q.size() < cap ? tr_q.size() : cap);
        for (uint32_t q = 0; q < lengths[k]; ++q) out[(size_t)k * cap + q] = tr_q[q].first | (tr_q[q].second ? 0x80000000u : 0u);
    }
    g_star_ns = nullptr;
    return 0;
}

}  // extern "C"
