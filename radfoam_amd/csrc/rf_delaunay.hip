// rf_delaunay.hip -- Delaunay neighbour lists on the GPU, one lane per point (the "later" half:
// the triangulation itself, not only tetrahedra -> CSR), plus the AABB tree of the reference's radfoam.build_aabb_tree.
//
// Reference: Triangulation::rebuild, src/delaunay/delaunay.cu:273-370 (sort_points -> build_aabb_tree ->
// sample_initial_tets -> growth_iteration until the frontier is empty -> find_adjacency; delete_violations.cu for
// incremental = true) and build_aabb_tree, src/aabb_tree/aabb_tree.cu:192-283.  The reference grows one global mesh
// through sorted tet / face tables; what the tracer needs from it is point_adjacency(+_offsets).  Here every point
// builds its own star (csrc/rf_star.hpp explains the algorithm and why it is exact) against the same AABB tree:
//
//   aabb_level_kernel        the tree, level by level, in the reference's layout (bit-identical boxes)
//   delaunay_star_kernel     a wave = 64 consecutive points of the kd-order = one subtree; the block's points sit in
//                            LDS and give every lane its first candidates (the 12 nearest of the block), or -- on an
//                            incremental rebuild -- the lane's previous neighbour list does; star state (64 link
//                            vertices, 124 triangles with cached spheres: 3.6 KB) lives in the lane's scratch
//                            every query has a budget of tree nodes; a star whose query runs out is parked
//   delaunay_star_coop_kernel second pass, one WAVE per star: the parked stars (rim of the cloud: half-spaces behind
//                            hull facets and the huge balls of flat tetrahedra, 10^5 tree nodes a query) and stars
//                            with more than 63 neighbours (room for 249); a few per thousand points
//   delaunay_star_huge_kernel third pass: hubs with more than 249 neighbours (room for 4095), a block per star, shared
//                            insertions; usually nothing to do
//   csr_*                    degrees -> offsets (rocPRIM scan) -> adjacency, ascending per point like find_adjacency's
//   symmetry_kernel          j in N(i) <=> i in N(j): exact predicates make independent stars agree; cospherical
//                            input can break that, which is reported (the reference throws "ambiguous triangulation")
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <stdint.h>

#include "../../include/radfoam_hip.h"
#include "rf_foam.hpp"
#include "rf_host.hpp"

#define RF_STAR_FN __device__ __forceinline__
#define RF_STAR_NOINLINE __device__ __noinline__
#define RF_STAR_NOUNROLL _Pragma("nounroll")
#define RF_STAR_ANY(x) (__ballot(x) != 0ull)
// the smallest value over the lanes of the wave, as a wave-uniform (scalar) value: rf_star.hpp, star_top_level.  A lane that
// has left contributes whatever its register holds -- a smaller bound at worst, which costs tests and never misses one.
static __device__ __forceinline__ uint32_t rf_wave_min(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
        v = o < v ? o : v;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
#define RF_STAR_UNIFORM_MIN(x) rf_wave_min(x)
#include "rf_star.hpp"

namespace rf {

constexpr int kSmallV = 64, kSmallT = 124;
constexpr int kBigV = 250, kBigT = 496;
constexpr int kBlockSeeds = 12;
constexpr uint32_t kBigFlag = 0x80000000u;   // first word of the row of a second-pass star: flag | row in big_rows

constexpr int kHugeV = 4096, kHugeT = 8188, kHugeHole = 1024;   // third instance: hubs
constexpr int kHugeStars = 64;                                   // ... of which a build may have this many
constexpr int kHugeWaves = 4;
constexpr uint32_t kHugeFlag = 0x40000000u;   // first word of the row of a third-pass star: flag | row in huge_rows

using SmallStar = star::Star<kSmallV, kSmallT>;
using BigStar = star::Star<kBigV, kBigT>;
using HugeStar = star::Star<kHugeV, kHugeT, uint16_t, kHugeHole>;

static inline uint32_t tree_depth_of(uint32_t n) {
    uint32_t d = 0;
    while ((1u << d) < n) ++d;
    return d;
}

// ---- AABB tree ------------------------------------------------------------------------------------------------------

// deepest level: node k = box of points 2k and 2k+1; indices past the end repeat the last point
// (build_leaves_kernel, aabb_tree.cu:192-219)
__global__ __launch_bounds__(256) void aabb_leaf_kernel(const float *__restrict__ pts, uint32_t n, uint32_t count,
                                                        float *__restrict__ level) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const uint32_t i0 = 2 * k < n ? 2 * k : n - 1, i1 = 2 * k + 1 < n ? 2 * k + 1 : i0;
    float *o = level + 6 * (size_t)k;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = pts[3 * (size_t)i0 + c], b = pts[3 * (size_t)i1 + c];
        o[c] = fminf(a, b);
        o[3 + c] = fmaxf(a, b);
    }
}

__global__ __launch_bounds__(256) void aabb_level_kernel(const float *__restrict__ below, uint32_t count,
                                                         float *__restrict__ level) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const float *a = below + 12 * (size_t)k, *b = a + 6;
    float *o = level + 6 * (size_t)k;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = fminf(a[c], b[c]);
        o[3 + c] = fmaxf(a[3 + c], b[3 + c]);
    }
}

// ---- stars ----------------------------------------------------------------------------------------------------------

struct StarCounters {
    uint32_t overflow;     // stars handed to the second pass (too large for the small instance, or parked)
    uint32_t hull;         // entries of the hull candidate list
    uint32_t huge;         // stars the second pass handed on to the third (more than 249 neighbours)
    uint32_t failed[6];    // stars by star::Status (index 0 unused)
    uint32_t asymmetric;   // directed edges without their reverse
    uint32_t adjacency;    // E
    uint32_t nodes_lo, nodes_hi;   // tree nodes visited (64-bit)
    uint32_t inserted;     // link insertions
#ifdef RF_COOP_SECTIONS        // instrumentation: wave clocks (/1024) of the second pass's slowest block, by section
    uint32_t sec_max[4];       // seed, queries, surgery, whole block
    uint32_t sec_who;          // the star of the slowest block
    uint32_t sec_rounds, sec_nt, sec_deg;
#endif
};

template <int V, int T>
__device__ __forceinline__ int gather_seeds(const float *__restrict__ pts, uint32_t n, uint32_t i,
                                            const uint32_t *__restrict__ seed_adj,
                                            const uint32_t *__restrict__ seed_off, const float *block_pts,
                                            uint32_t block_first, uint32_t block_count, uint32_t *seeds, int cap) {
    int ns = 0;
    if (seed_adj) {
        const uint32_t e0 = seed_off[i], e1 = seed_off[i + 1];
        for (uint32_t e = e0; e < e1 && ns < cap; ++e) {
            const uint32_t j = seed_adj[e];
            if (j < n && j != i) seeds[ns++] = j;
        }
        if (ns >= 3) {
#if RF_STAR_SORT_SEEDS
            star::sort_seeds<V>(pts, pts + 3 * (size_t)i, seeds, ns);
#endif
            return ns;
        }
        ns = 0;
    }
    // the nearest points of the block, by repeated selection of the next (distance, index) pair
    const float px = pts[3 * (size_t)i], py = pts[3 * (size_t)i + 1], pz = pts[3 * (size_t)i + 2];
    float last_d = -1.0f;
    uint32_t last_k = 0;
    for (int r = 0; r < kBlockSeeds && r < cap; ++r) {
        float best_d = 3.4e38f;
        uint32_t best_k = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < block_count; ++k) {
            if (block_first + k == i) continue;
            const float dx = block_pts[3 * k] - px, dy = block_pts[3 * k + 1] - py, dz = block_pts[3 * k + 2] - pz;
            const float d = dx * dx + dy * dy + dz * dz;
            const bool after = d > last_d || (d == last_d && k > last_k && r > 0);
            if (after && d < best_d) {
                best_d = d;
                best_k = k;
            }
        }
        if (best_k == 0xFFFFFFFFu) break;
        seeds[ns++] = block_first + best_k;
        last_d = best_d;
        last_k = best_k;
    }
    return ns;
}

// First pass: one lane per point, a wave = 64 consecutive points of the kd-order; the lane's star is a private
// (scratch) record.
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(64, WAVES_PER_SIMD) void delaunay_star_kernel(
    const float *__restrict__ pts, uint32_t n, const float *__restrict__ tree, uint32_t depth,
    const uint32_t *__restrict__ seed_adj, const uint32_t *__restrict__ seed_off, uint32_t *__restrict__ rows,
    uint32_t *__restrict__ degree, uint32_t *__restrict__ overflow_list, uint32_t *__restrict__ hull_list,
    uint32_t budget, StarCounters *__restrict__ counters) {
    __shared__ float block_pts[64 * 3];
    // candidates of a lane without a previous list: the block's own points (a short last block: the last 64 points)
    uint32_t block_first, block_count;
    star::seed_window(n, blockIdx.x * 64u, block_first, block_count);
    for (uint32_t k = threadIdx.x; k < 3 * block_count; k += 64) block_pts[k] = pts[3 * (size_t)block_first + k];
    __syncthreads();
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;

    SmallStar s;
    star::star_reset(s, i, pts + 3 * (size_t)i);
    uint32_t seeds[kSmallV];
    const int ns = gather_seeds<kSmallV, kSmallT>(pts, n, i, seed_adj, seed_off, block_pts, block_first, block_count,
                                                  seeds, kSmallV - 1);
    const star::Tree tr{tree, n, depth};
    const star::HullSet first_pass{nullptr, 0u, budget};
    uint32_t visited = 0, inserted = 0;
#if RF_STAR_KNN_SEEDS > 0
    // a star without a previous list is seeded with its K nearest points from a walk of the tree (rf_star.hpp:
    // RF_STAR_KNN_SEEDS) instead of the nearest of its 64-point kd-block
    int ns_used = ns;
    if (!seed_adj) ns_used = star::star_knn_up<RF_STAR_KNN_SEEDS>(tr, pts, i, seeds, visited);
    star::star_build(s, tr, pts, first_pass, seeds, ns_used, visited, inserted);
#else
    star::star_build(s, tr, pts, first_pass, seeds, ns, visited, inserted);
#endif
    atomicAdd(&counters->inserted, inserted);
    const uint32_t old = atomicAdd(&counters->nodes_lo, visited);
    if (old + visited < old) atomicAdd(&counters->nodes_hi, 1u);

    uint32_t *row = rows + (size_t)i * kSmallV;
    if (s.status == star::kOverflow || s.status == star::kPending) {
        overflow_list[atomicAdd(&counters->overflow, 1u)] = i;
        hull_list[atomicAdd(&counters->hull, 1u)] = i;   // not known to be interior: a hull candidate
        // what this pass found so far are the second pass's first candidates
        uint32_t c = 0;
        for (int k = 1; k < kSmallV; ++k)
            if (s.v[k].use) row[c++] = s.v[k].g;
        degree[i] = c;
        return;
    }
    if (s.status != star::kOk) {
        atomicAdd(&counters->failed[s.status], 1u);
        degree[i] = 0;
        return;
    }
    bool hull;
    const int deg = star::star_neighbours(s, row, 1, &hull);   // ascending, straight into the row
    if (hull) hull_list[atomicAdd(&counters->hull, 1u)] = i;   // a vertex of the convex hull
    degree[i] = (uint32_t)deg;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}

struct CoopResult {
    uint32_t id;
    float q[3];
    float d2;        // squared distance of q from the star's point (the waves that share a query compare by it)
    bool duplicate;
};

// One wave answers one query: the tree is walked six levels at a time -- the 64 descendants of a node six levels down are
// consecutive in memory: a box per lane, a ballot keeps what the region touches -- from a stack of {level, first node,
// ballot} entries in LDS.  kCoopUnroll blocks of 64 boxes are requested per trip (the children of up to that many set
// bits of the entry on top), so that their loads are in flight together: the expensive queries of this pass -- the
// half-space-like balls of the rim, 10^5 boxes that all have to be looked at -- are chains of such trips, each a
// round trip to L2 (one block per trip: 76 / 31 ms for the second pass of the 2 M-point foam from scratch /
// incrementally; four: see profiles/README.md r06).
#ifndef RF_COOP_UNROLL
#define RF_COOP_UNROLL 4
#endif
constexpr int kCoopUnroll = RF_COOP_UNROLL;
constexpr int kCoopStack = 40;   // entries: at most kCoopUnroll per level below an entry that is still open, 6 levels

struct CoopStack {
    unsigned long long mask[kCoopStack];
    uint32_t base[kCoopStack];
    uint32_t level[kCoopStack];
};

template <typename S>
__device__ CoopResult coop_search(const S &s, const star::Tree &tr, const float *__restrict__ pts, int t,
                                  const uint32_t *__restrict__ hull_ids, uint32_t hull_count, uint32_t &visited,
                                  CoopStack *stack, uint32_t part = 0u, uint32_t parts = 1u) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint8_t f = s.t[t].f;
    const bool ghost = (f & star::kGhost) != 0;
    const bool ball = !ghost && s.t[t].sr < 3.0e38f;
    const float px = s.p[0], py = s.p[1], pz = s.p[2];
    const float nx = s.t[t].sx, ny = s.t[t].sy, nz = s.t[t].sz;
    const float cx = px + nx, cy = py + ny, cz = pz + nz;
    float rp2 = 3.4e38f;
    if (ball) {
        const float r = sqrtf(s.t[t].sr);
        const float pad = 4e-7f * (fabsf(px) + fabsf(py) + fabsf(pz) + fabsf(nx) + fabsf(ny) + fabsf(nz) + r);
        const float rp = (r + pad) * 1.000002f;
        rp2 = rp * rp;
    }
    const uint32_t g0 = s.v[s.t[t].a].g, g1 = s.v[s.t[t].b].g, g2 = s.v[s.t[t].c].g;
    float best = 3.4e38f;        // wave-uniform: prunes
    float my_d2 = 3.4e38f;       // this lane's candidate
    uint32_t my_id = star::kInfinity;
    float my_q[3] = {0.0f, 0.0f, 0.0f};
    bool duplicate = false;

    auto try_point = [&](uint32_t k) {
        if (k == s.self || k == g0 || k == g1 || k == g2) return;
        const float q[3] = {pts[3 * (size_t)k], pts[3 * (size_t)k + 1], pts[3 * (size_t)k + 2]};
        const float dx = q[0] - px, dy = q[1] - py, dz = q[2] - pz;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 == 0.0f && q[0] == px && q[1] == py && q[2] == pz) duplicate = true;
        if (!(d2 < my_d2) || !(d2 < best) || !star::conflict(s, t, q)) return;
        my_d2 = d2;
        my_id = k;
        my_q[0] = q[0];
        my_q[1] = q[1];
        my_q[2] = q[2];
    };

    // `parts` waves share this query: wave `part` takes every parts-th block of the hull candidates / every parts-th
    // subtree below the first level of the walk (the top of the tree is looked at by all of them); the caller keeps
    // the nearest of their answers
    if (ghost && hull_ids) {
        for (uint32_t base = 64u * part; base < hull_count; base += 64u * parts)
            if (base + lane < hull_count) try_point(hull_ids[base + lane]);
        visited += hull_count / parts;
    } else {
        const uint32_t leaf_depth = tr.depth - star::kLeafBits;
        const uint32_t d0 = leaf_depth % 6;
        // does the region touch the box?
        auto touches = [&](const float *nd) {
            bool ok = star::box_dist2(nd, px, py, pz) < best;
            if (ok && ball) ok = star::box_dist2(nd, cx, cy, cz) <= rp2;
            if (ok && ghost && !(f & star::kSlow)) {
                const float ax = nx > 0 ? nd[3] - px : nd[0] - px, ay = ny > 0 ? nd[4] - py : nd[1] - py;
                const float az = nz > 0 ? nd[5] - pz : nd[2] - pz;
                const float tx = nx * ax, ty = ny * ay, tz = nz * az;
                ok = tx + ty + tz > -4e-6f * (fabsf(tx) + fabsf(ty) + fabsf(tz));
            }
            return ok;
        };
        // the points of a leaf bucket
        auto bucket = [&](uint32_t first) {
            const uint32_t end = first + (1u << star::kLeafBits) < tr.n ? first + (1u << star::kLeafBits) : tr.n;
            for (uint32_t k = first; k < end; ++k) try_point(k);
        };
        volatile __attribute__((address_space(3))) CoopStack *st =
            (volatile __attribute__((address_space(3))) CoopStack *)stack;
        int sp = 0;
        {   // level 0: all 2^d0 nodes of depth d0
            const uint32_t width = 1u << d0;
            const uint32_t first = lane << (tr.depth - d0);
            bool ok = lane < width && first < tr.n;
            if (ok) ok = touches(star::tree_node(tr, d0, lane));
            visited += width;
            if (d0 == leaf_depth) {
                if (ok && part == 0u) bucket(first);
            } else {
                const unsigned long long m = __ballot(ok);
                if (m) {
                    if (lane == 0) {
                        st->mask[0] = m;
                        st->base[0] = 0u;
                        st->level[0] = 0u;
                    }
                    sp = 1;
                }
            }
        }
        while (sp > 0) {
            unsigned long long m = st->mask[sp - 1];
            const uint32_t pbase = st->base[sp - 1], plevel = st->level[sp - 1];
            // the children of up to kCoopUnroll nodes of the entry on top: blocks of 64 boxes at the next level
            uint32_t cbase[kCoopUnroll];
            int blocks = 0;
#pragma unroll
            for (int u = 0; u < kCoopUnroll; ++u) {
                cbase[u] = 0u;
                if (m) {
                    cbase[u] = (pbase + (uint32_t)__builtin_ctzll(m)) << 6;
                    m &= m - 1ull;
                    blocks = u + 1;
                }
            }
            if (m == 0ull) --sp;
            else if (lane == 0) st->mask[sp - 1] = m;
            const uint32_t level = plevel + 1u, depth = d0 + 6u * level;
            const bool leaves = depth == leaf_depth;
            float box[kCoopUnroll][6];
            bool ok[kCoopUnroll];
#pragma unroll
            for (int u = 0; u < kCoopUnroll; ++u) {   // the requests of all the blocks first
                const uint32_t idx = cbase[u] + lane;
                ok[u] = u < blocks && ((idx << (tr.depth - depth)) < tr.n);
                if (ok[u]) {
                    const float *nd = star::tree_node(tr, depth, idx);
#pragma unroll
                    for (int c = 0; c < 6; ++c) box[u][c] = nd[c];
                }
            }
            visited += 64u * (uint32_t)blocks;
            bool found = false;
#pragma unroll
            for (int u = 0; u < kCoopUnroll; ++u) {
                if (u >= blocks) break;   // wave-uniform
                if (ok[u]) ok[u] = touches(box[u]);
                // the subtrees below the first level are dealt to the waves that share the query
                if (plevel == 0u && parts > 1u) ok[u] = ok[u] && (lane % parts) == part;
                if (leaves) {
                    if (ok[u]) bucket((cbase[u] + lane) << (tr.depth - depth));
                    found = true;
                } else {
                    const unsigned long long cm = __ballot(ok[u]);
                    if (cm) {
                        if (lane == 0) {
                            st->mask[sp] = cm;
                            st->base[sp] = cbase[u];
                            st->level[sp] = level;
                        }
                        ++sp;
                    }
                }
            }
            if (found) best = wave_min(my_d2);
        }
    }
    // the nearest candidate of the wave, the lower index on ties
    const float d = wave_min(my_d2);
    const unsigned long long holders = __ballot(my_id != star::kInfinity && my_d2 == d);
    CoopResult r;
    r.duplicate = __ballot(duplicate) != 0;
    r.id = star::kInfinity;
    r.q[0] = r.q[1] = r.q[2] = 0.0f;
    r.d2 = d;
    if (holders) {
        const int src = (int)__builtin_ctzll(holders);
        r.id = __shfl(my_id, src, 64);
        r.q[0] = __shfl(my_q[0], src, 64);
        r.q[1] = __shfl(my_q[1], src, 64);
        r.q[2] = __shfl(my_q[2], src, 64);
    }
    return r;
}

// kCoopWaves waves per second-pass star, kCoopGroup of them on each query: kCoopWaves / kCoopGroup of the star's queries run
// at the same time.  The launch lasts as long as its slowest star -- a rim star whose ~40 rounds each wait for a query of
// 10^5-10^6 boxes answered by ONE wave (issue-bound: 76 % of that block's clocks, -DRF_COOP_SECTIONS=1) --, so the waves
// that share a query shorten the launch although they do not shorten the work.
template <int kCoopWaves, int kCoopGroup>
__global__ __launch_bounds__(64 * kCoopWaves) void delaunay_star_coop_kernel(
    const float *__restrict__ pts, uint32_t n, const float *__restrict__ tree, uint32_t depth,
    const uint32_t *__restrict__ seed_adj, const uint32_t *__restrict__ seed_off,
    const uint32_t *__restrict__ overflow_list, const uint32_t *__restrict__ hull_list, uint32_t hull_count,
    uint32_t count, const uint32_t *__restrict__ rows, uint32_t *__restrict__ big_rows,
    uint32_t *__restrict__ degree, uint32_t *__restrict__ huge_list, StarCounters *__restrict__ counters) {
    __shared__ BigStar s;
    __shared__ uint32_t seeds[kBigV];
    constexpr int kQueries = kCoopWaves / kCoopGroup;
    __shared__ int pick[kQueries];
    __shared__ CoopResult found[kCoopWaves];
    __shared__ CoopStack stacks[kCoopWaves];
    const uint32_t w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    if (w >= count) return;
    const uint32_t i = overflow_list[w];
    const star::Tree tr{tree, n, depth};
    uint32_t visited = 0, inserted = 0;
#ifdef RF_COOP_SECTIONS
    unsigned long long c_seed = 0, c_query = 0, c_surgery = 0, c_rounds = 0;
    const unsigned long long c_start = __builtin_readcyclecounter();
#endif
    if (tid == 0) {
        star::star_reset(s, i, pts + 3 * (size_t)i);
        uint32_t block_first, block_count;
        star::seed_window(n, i, block_first, block_count);
        int ns = (int)degree[i];   // the link vertices the first pass got to
        if (ns >= 3) {
            for (int k = 0; k < ns; ++k) seeds[k] = rows[(size_t)i * kSmallV + k];
        } else {
            ns = gather_seeds<kBigV, kBigT>(pts, n, i, seed_adj, seed_off, pts + 3 * (size_t)block_first, block_first,
                                            block_count, seeds, kBigV - 1);
        }
        star::star_seed(s, pts, seeds, ns, inserted);
#ifdef RF_COOP_SECTIONS
        c_seed = __builtin_readcyclecounter() - c_start;
#endif
    }
    for (;;) {
#ifdef RF_COOP_SECTIONS
        const unsigned long long r0 = __builtin_readcyclecounter();
#endif
        // a round: every wave takes one uncertified triangle; certification is a statement about the whole point
        // set, so it holds whatever the other waves' answers do to the link afterwards
        if (tid == 0) {
            int k = 0;
            for (int v = 0; v < kQueries; ++v) {
                pick[v] = -1;
                if (s.status != star::kOk) continue;
                while (k < s.nt && (s.t[k].f & star::kCertified)) ++k;
                if (k < s.nt) pick[v] = k++;
            }
        }
        __syncthreads();
        if (pick[0] < 0) break;
        const int t = pick[wave / kCoopGroup];
        if (t >= 0) {
            const CoopResult r = coop_search(s, tr, pts, t, hull_list, hull_count, visited, &stacks[wave],
                                             wave % kCoopGroup, (uint32_t)kCoopGroup);
            if ((tid & 63u) == 0) found[wave] = r;
        }
        __syncthreads();
#ifdef RF_COOP_SECTIONS
        const unsigned long long r1 = __builtin_readcyclecounter();
        c_query += r1 - r0;
        c_rounds++;
#endif
        if (tid == 0) {
            // the answer of a query: the nearest of its waves' (the lower index on ties), in the first wave's slot
            for (int v = 0; v < kQueries; ++v) {
                if (pick[v] < 0) continue;
                CoopResult &a = found[v * kCoopGroup];
                for (int g = 1; g < kCoopGroup; ++g) {
                    const CoopResult &b = found[v * kCoopGroup + g];
                    a.duplicate = a.duplicate || b.duplicate;
                    if (b.id != star::kInfinity && (a.id == star::kInfinity || b.d2 < a.d2 || (b.d2 == a.d2 && b.id < a.id))) {
                        a.id = b.id;
                        a.d2 = b.d2;
                        a.q[0] = b.q[0];
                        a.q[1] = b.q[1];
                        a.q[2] = b.q[2];
                    }
                }
                if (v) found[v] = found[v * kCoopGroup];
            }
            // certifications first (triangle indices are still those of the round), insertions after
            for (int v = 0; v < kQueries; ++v) {
                if (pick[v] < 0) continue;
                if (found[v].duplicate) s.status = star::kDuplicate;
                else if (found[v].id == star::kInfinity) s.t[pick[v]].f |= star::kCertified;
            }
            for (int v = 0; v < kQueries && s.status == star::kOk; ++v) {
                if (pick[v] < 0 || found[v].id == star::kInfinity) continue;
                // 0 = the triangle this point conflicted with went with an earlier insertion of the round, and the
                // point conflicts with nothing that replaced it
                if (star::star_insert(s, found[v].id, found[v].q) > 0) ++inserted;
            }
        }
        __syncthreads();
#ifdef RF_COOP_SECTIONS
        c_surgery += __builtin_readcyclecounter() - r1;
#endif
    }
#ifdef RF_COOP_SECTIONS
    if (tid == 0) {
        const unsigned long long whole = __builtin_readcyclecounter() - c_start;
        const uint32_t w1024 = (uint32_t)(whole >> 10);
        if (atomicMax(&counters->sec_max[3], w1024) < w1024) {   // (racy on purpose: a diagnostic)
            counters->sec_max[0] = (uint32_t)(c_seed >> 10);
            counters->sec_max[1] = (uint32_t)(c_query >> 10);
            counters->sec_max[2] = (uint32_t)(c_surgery >> 10);
            counters->sec_who = i;
            counters->sec_rounds = (uint32_t)c_rounds;
            counters->sec_nt = (uint32_t)s.nt;
            counters->sec_deg = (uint32_t)degree[i];
        }
    }
#endif
    if ((tid & 63u) == 0) {
        const uint32_t old = atomicAdd(&counters->nodes_lo, visited);
        if (old + visited < old) atomicAdd(&counters->nodes_hi, 1u);
    }
    if (tid != 0) return;
    atomicAdd(&counters->inserted, inserted);
    if (s.status == star::kOverflow) {
        // a hub: the third pass takes it, with what this pass found as its first candidates
        const uint32_t slot = atomicAdd(&counters->huge, 1u);
        if (slot < (uint32_t)kHugeStars) {
            huge_list[slot] = w;
            uint32_t c = 0;
            for (int k = 1; k < kBigV; ++k)
                if (s.v[k].use) big_rows[(size_t)w * kBigV + c++] = s.v[k].g;
            degree[i] = c;
            return;
        }
    }
    if (s.status != star::kOk) {
        atomicAdd(&counters->failed[s.status], 1u);
        degree[i] = 0;
        return;
    }
    bool hull;
    degree[i] = (uint32_t)star::star_neighbours(s, big_rows + (size_t)w * kBigV, 1, &hull);
}

// ---- third pass: hubs ------------------------------------------------------------------------------------------------
// A point inside an empty shell of thousands of points (a floater inside a densely sampled surface) has thousands of
// Delaunay neighbours.  Such a star (room for 4095; state in global memory, 280 KB) gets a block of four waves: the
// waves answer its open queries like the second pass does, and every insertion is shared -- all threads test the
// link's triangles against the new point and list the hole, thread 0 re-triangulates it from that list.

template <typename S>
__device__ int block_insert(S &s, uint32_t gq, const float *q, uint32_t *hole, int *nhole) {
    if (threadIdx.x == 0) *nhole = 0;
    __syncthreads();
    for (int t = (int)threadIdx.x; t < s.nt; t += (int)blockDim.x)
        if (star::conflict(s, t, q)) {
            s.t[t].f |= star::kMarked;
            const int k = atomicAdd(nhole, 1);
            if (k < S::kHole) hole[k] = (uint32_t)t;
        }
    __syncthreads();
    int r = 0;
    if (threadIdx.x == 0 && *nhole > 0) r = star::star_apply(s, gq, q, *nhole, *nhole <= S::kHole ? hole : nullptr);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(64 * kHugeWaves) void delaunay_star_huge_kernel(
    const float *__restrict__ pts, uint32_t n, const float *__restrict__ tree, uint32_t depth,
    const uint32_t *__restrict__ overflow_list, const uint32_t *__restrict__ hull_list, uint32_t hull_count,
    const uint32_t *__restrict__ huge_list, const uint32_t *__restrict__ big_rows, HugeStar *__restrict__ arena,
    uint32_t *__restrict__ huge_rows, uint32_t *__restrict__ rows, uint32_t *__restrict__ degree,
    StarCounters *__restrict__ counters) {
    __shared__ uint32_t hole[kHugeHole];
    __shared__ int nhole;
    __shared__ int pick[kHugeWaves];
    __shared__ CoopResult found[kHugeWaves];
    __shared__ CoopStack stacks[kHugeWaves];
    __shared__ int ok;
    const uint32_t b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const uint32_t total = counters->huge;
    // (hubs beyond kHugeStars were already counted as failed by the second pass, which found no slot for them)
    if (b >= total || b >= (uint32_t)kHugeStars) return;
    const uint32_t w = huge_list[b], i = overflow_list[w];
    HugeStar &s = arena[b];
    const star::Tree tr{tree, n, depth};
    const uint32_t *seeds = big_rows + (size_t)w * kBigV;
    const int ns = (int)degree[i];
    uint32_t visited = 0, inserted = 0;
    int used[3] = {-1, -1, -1};
    if (tid == 0) {
        star::star_reset(s, i, pts + 3 * (size_t)i);
        star::star_first_tet(s, pts, seeds, ns, used);
        pick[0] = used[0];
        pick[1] = used[1];
        pick[2] = used[2];
        ok = s.status == star::kOk;
    }
    __syncthreads();
    used[0] = pick[0];
    used[1] = pick[1];
    used[2] = pick[2];
    for (int k = 0; k < ns && ok; ++k) {
        if (k == used[0] || k == used[1] || k == used[2]) continue;
        const float *q = pts + 3 * (size_t)seeds[k];
        const int r = block_insert(s, seeds[k], q, hole, &nhole);
        if (tid == 0) {
            inserted += r > 0;
            if (r < 0) ok = 0;
        }
        __syncthreads();
    }
    for (;;) {
        if (tid == 0) {
            int k = 0;
            for (int v = 0; v < kHugeWaves; ++v) {
                pick[v] = -1;
                if (!ok || s.status != star::kOk) continue;
                while (k < s.nt && (s.t[k].f & star::kCertified)) ++k;
                if (k < s.nt) pick[v] = k++;
            }
        }
        __syncthreads();
        if (pick[0] < 0) break;
        const int t = pick[wave];
        if (t >= 0) {
            const CoopResult r = coop_search(s, tr, pts, t, hull_list, hull_count, visited, &stacks[wave]);
            if ((tid & 63u) == 0) found[wave] = r;
        }
        __syncthreads();
        if (tid == 0)
            for (int v = 0; v < kHugeWaves; ++v) {
                if (pick[v] < 0) continue;
                if (found[v].duplicate) s.status = star::kDuplicate;
                else if (found[v].id == star::kInfinity) s.t[pick[v]].f |= star::kCertified;
            }
        __syncthreads();
        for (int v = 0; v < kHugeWaves; ++v) {
            if (pick[v] < 0 || found[v].id == star::kInfinity || s.status != star::kOk) continue;   // block-uniform
            const int r = block_insert(s, found[v].id, found[v].q, hole, &nhole);
            if (tid == 0) inserted += r > 0;
        }
        __syncthreads();
    }
    if ((tid & 63u) == 0) {
        const uint32_t old = atomicAdd(&counters->nodes_lo, visited);
        if (old + visited < old) atomicAdd(&counters->nodes_hi, 1u);
    }
    if (tid != 0) return;
    atomicAdd(&counters->inserted, inserted);
    if (s.status != star::kOk) {
        atomicAdd(&counters->failed[s.status], 1u);
        degree[i] = 0;
        return;
    }
    bool hull;
    degree[i] = (uint32_t)star::star_neighbours(s, huge_rows + (size_t)b * kHugeV, 1, &hull);
    rows[(size_t)i * kSmallV] = b | kHugeFlag;
}

// rows of the large instance are addressed through rows[i * kSmallV] = position in the overflow list
__global__ __launch_bounds__(256) void mark_big_rows_kernel(const uint32_t *__restrict__ overflow_list, uint32_t count,
                                                            uint32_t *__restrict__ rows) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < count) rows[(size_t)overflow_list[w] * kSmallV] = w | kBigFlag;
}

struct RowTables {
    const uint32_t *rows, *big_rows, *huge_rows;
};

__device__ __forceinline__ const uint32_t *row_of(const RowTables &T, uint32_t i, uint32_t deg) {
    const uint32_t *row = T.rows + (size_t)i * kSmallV;
    if (deg && (row[0] & kBigFlag)) return T.big_rows + (size_t)(row[0] & ~kBigFlag) * kBigV;
    if (deg && (row[0] & kHugeFlag)) return T.huge_rows + (size_t)(row[0] & ~kHugeFlag) * kHugeV;
    return row;
}

__global__ __launch_bounds__(256) void csr_gather_kernel(RowTables tables,
                                                         const uint32_t *__restrict__ degree,
                                                         const uint32_t *__restrict__ offsets, uint32_t n,
                                                         uint32_t capacity, uint32_t *__restrict__ adjacency,
                                                         StarCounters *__restrict__ counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t deg = degree[i], off = offsets[i];
    if (i == n - 1) counters->adjacency = off + deg;
    if (off + deg > capacity) return;
    const uint32_t *row = row_of(tables, i, deg);
    for (uint32_t k = 0; k < deg; ++k) adjacency[off + k] = row[k];
}

__global__ __launch_bounds__(256) void symmetry_kernel(RowTables tables,
                                                       const uint32_t *__restrict__ degree, uint32_t n,
                                                       StarCounters *__restrict__ counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t deg = degree[i];
    const uint32_t *row = row_of(tables, i, deg);
    uint32_t missing = 0;
    for (uint32_t k = 0; k < deg; ++k) {
        const uint32_t j = row[k];
        const uint32_t dj = degree[j];
        const uint32_t *rj = row_of(tables, j, dj);
        uint32_t lo = 0, hi = dj;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rj[mid] < i) lo = mid + 1; else hi = mid;
        }
        missing += !(lo < dj && rj[lo] == i);
    }
    if (missing) atomicAdd(&counters->asymmetric, missing);
}

__global__ void last_offset_kernel(const uint32_t *__restrict__ degree, uint32_t *__restrict__ offsets, uint32_t n) {
    offsets[n] = offsets[n - 1] + degree[n - 1];
}

// ---- kd-order (sort_points, src/aabb_tree/aabb_tree.cu:62-190) ------------------------------------------------------
// With P = pow2_round_up(N): sort everything by x, every consecutive P/2 segment by y, every P/4 segment by z, ...
// cycling the axis down to segments of 2; ties keep their order (stable sorts).  The reference runs a CUB
// segmented / global / block radix sort per level; here a level is one rocPRIM radix sort of 64-bit keys
// (segment index << 32 | order-preserving bits of the coordinate) carrying the permutation.

__device__ __forceinline__ uint32_t orderable(float f) {   // aabb_tree.cu:88-96
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u ^ 0x80000000u);
}

__global__ __launch_bounds__(256) void kd_keys_kernel(const float *__restrict__ pts, const uint32_t *__restrict__ perm,
                                                      uint32_t n, uint32_t seg_shift, uint32_t dim,
                                                      unsigned long long *__restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((unsigned long long)(i >> seg_shift) << 32) | orderable(pts[3 * (size_t)perm[i] + dim]);
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t *__restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

__global__ __launch_bounds__(256) void kd_gather_kernel(const float *__restrict__ pts, const uint32_t *__restrict__ perm,
                                                        uint32_t n, uint32_t *__restrict__ perm_out,
                                                        float *__restrict__ sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = perm[i];
    perm_out[i] = j;
    sorted[3 * (size_t)i] = pts[3 * (size_t)j];
    sorted[3 * (size_t)i + 1] = pts[3 * (size_t)j + 1];
    sorted[3 * (size_t)i + 2] = pts[3 * (size_t)j + 2];
}

struct KdLayout {
    size_t keys_a, keys_b, perm_a, perm_b, temp, temp_bytes, total;
};

static KdLayout kd_layout(uint32_t n) {
    KdLayout L{};
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
    L.temp_bytes = bytes;
    size_t at = 0;
    auto take = [&](size_t b) {
        const size_t here = at;
        at += align_up(b, 256);
        return here;
    };
    L.keys_a = take((size_t)n * 8);
    L.keys_b = take((size_t)n * 8);
    L.perm_a = take((size_t)n * 4);
    L.perm_b = take((size_t)n * 4);
    L.temp = take(bytes);
    L.total = at;
    return L;
}

struct DelaunayLayout {
    size_t rows, degree, overflow, hull, counters, big_rows, huge_list, huge_rows, huge_arena, scan_temp, scan_bytes, total;
    uint32_t second_rows;   // stars the second pass has result rows for
};

static size_t scan_temp_bytes(uint32_t n) {
    size_t scan_bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)n,
                                  rocprim::plus<uint32_t>(), (hipStream_t)0);
    return scan_bytes;
}

static uint32_t default_second_rows(uint32_t n) { return n / 32 < 1024 ? 1024 : n / 32; }

static DelaunayLayout delaunay_layout(uint32_t n, uint32_t second_rows, size_t scan_bytes) {
    DelaunayLayout L{};
    L.scan_bytes = scan_bytes;
    L.second_rows = second_rows;
    size_t at = 0;
    auto take = [&](size_t bytes) {
        const size_t here = at;
        at += align_up(bytes, 256);
        return here;
    };
    L.rows = take((size_t)n * kSmallV * 4);
    L.degree = take((size_t)n * 4);
    L.overflow = take((size_t)n * 4);
    L.hull = take((size_t)n * 4);
    L.counters = take(sizeof(StarCounters));
    L.big_rows = take((size_t)second_rows * kBigV * 4);
    L.huge_list = take((size_t)kHugeStars * 4);
    L.huge_rows = take((size_t)kHugeStars * kHugeV * 4);
    L.huge_arena = take((size_t)kHugeStars * sizeof(HugeStar));
    L.scan_temp = take(scan_bytes);
    L.total = at;
    return L;
}

}  // namespace rf

using namespace rf;

extern "C" {

int rf_build_aabb_tree(const float *points, uint32_t num_points, float *aabb_tree, void *stream) {
    g_err[0] = 0;
    if (!points || !aabb_tree || num_points == 0) return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_aabb_tree: null pointer or no points");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t depth = tree_depth_of(num_points);
    if (depth == 0) return RF_OK;   // one point: the reference's tree has a single, never written, entry
    const size_t p2 = (size_t)1 << depth;
    uint32_t count = (uint32_t)(p2 >> 1);
    float *level = aabb_tree;   // level depth-1 starts at node 0
    hipLaunchKernelGGL(aabb_leaf_kernel, dim3((count + 255u) / 256u), dim3(256), 0, s, points, num_points, count, level);
    for (uint32_t d = depth - 1; d-- > 0;) {
        float *above = aabb_tree + 6 * (p2 - ((size_t)1 << (d + 1)));
        count >>= 1;
        hipLaunchKernelGGL(aabb_level_kernel, dim3((count + 255u) / 256u), dim3(256), 0, s, level, count, above);
        level = above;
    }
    return check_launch("rf_build_aabb_tree");
}

size_t rf_kd_order_workspace_bytes(uint32_t num_points) { return kd_layout(num_points).total; }

int rf_kd_order(const float *points, uint32_t num_points, uint32_t *permutation, float *sorted_points, void *workspace,
                size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (num_points == 0) return RF_OK;
    if (!points || !permutation || !sorted_points) return fail(RF_ERR_INVALID_ARGUMENT, "rf_kd_order: null pointer");
    const KdLayout L = kd_layout(num_points);
    if (!workspace || workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_kd_order_workspace_bytes()");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(workspace);
    unsigned long long *keys_a = reinterpret_cast<unsigned long long *>(base + L.keys_a);
    unsigned long long *keys_b = reinterpret_cast<unsigned long long *>(base + L.keys_b);
    uint32_t *perm_a = reinterpret_cast<uint32_t *>(base + L.perm_a);
    uint32_t *perm_b = reinterpret_cast<uint32_t *>(base + L.perm_b);
    const uint32_t blocks = (num_points + 255u) / 256u;
    hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(256), 0, s, perm_a, num_points);
    const uint32_t depth = tree_depth_of(num_points);
    uint32_t dim = 0;
    for (uint32_t seg_shift = depth; seg_shift >= 1; --seg_shift) {   // segments of 2^seg_shift points
        hipLaunchKernelGGL(kd_keys_kernel, dim3(blocks), dim3(256), 0, s, points, perm_a, num_points, seg_shift, dim,
                           keys_a);
        size_t bytes = L.temp_bytes;
        const unsigned end_bit = 32u + (depth - seg_shift) + 1u;
        if (rocprim::radix_sort_pairs(base + L.temp, bytes, keys_a, keys_b, perm_a, perm_b, (size_t)num_points, 0,
                                      end_bit > 64u ? 64u : end_bit, s) != hipSuccess)
            return fail(RF_ERR_LAUNCH, "rf_kd_order: radix sort failed");
        uint32_t *t = perm_a;
        perm_a = perm_b;
        perm_b = t;
        dim = (dim + 1) % 3;
    }
    hipLaunchKernelGGL(kd_gather_kernel, dim3(blocks), dim3(256), 0, s, points, perm_a, num_points, permutation,
                       sorted_points);
    return check_launch("rf_kd_order");
}

size_t rf_delaunay_workspace_bytes(uint32_t num_points) {
    return delaunay_layout(num_points, default_second_rows(num_points), scan_temp_bytes(num_points)).total;
}

size_t rf_delaunay_workspace_bytes_for(uint32_t num_points, uint32_t second_pass_stars) {
    const uint32_t rows = second_pass_stars > num_points ? num_points : second_pass_stars;
    return delaunay_layout(num_points, rows < 1024 ? 1024 : rows, scan_temp_bytes(num_points)).total;
}

int rf_delaunay_adjacency(const float *points, uint32_t num_points, const float *aabb_tree,
                          const uint32_t *seed_adjacency, const uint32_t *seed_offsets, uint32_t *point_adjacency,
                          uint32_t adjacency_capacity, uint32_t *point_adjacency_offsets, uint32_t *info,
                          void *workspace, size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (!points || !aabb_tree || !point_adjacency || !point_adjacency_offsets || !info)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_delaunay_adjacency: null pointer");
    if ((seed_adjacency == nullptr) != (seed_offsets == nullptr))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_delaunay_adjacency: seed_adjacency and seed_offsets go together");
    if (num_points < 32)
        return fail(RF_ERR_INVALID_ARGUMENT, "Delaunay triangulation does not support less than 32 points");
    // the workspace decides how many stars the second pass has rows for: the default size, or whatever larger one
    // the caller came back with (rf_delaunay_workspace_bytes_for) after being told how many were needed
    const size_t scan_bytes = scan_temp_bytes(num_points);
    uint32_t second_rows = default_second_rows(num_points);
    if (!workspace || workspace_bytes < delaunay_layout(num_points, second_rows, scan_bytes).total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_delaunay_workspace_bytes()");
    for (uint32_t lo = second_rows, hi = num_points > second_rows ? num_points : second_rows; lo < hi;) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;   // the largest row count this workspace holds
        if (delaunay_layout(num_points, mid, scan_bytes).total <= workspace_bytes) second_rows = lo = mid;
        else hi = mid - 1;
    }
    const DelaunayLayout L = delaunay_layout(num_points, second_rows, scan_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(workspace);
    uint32_t *rows = reinterpret_cast<uint32_t *>(base + L.rows);
    uint32_t *degree = reinterpret_cast<uint32_t *>(base + L.degree);
    uint32_t *overflow = reinterpret_cast<uint32_t *>(base + L.overflow);
    uint32_t *hull = reinterpret_cast<uint32_t *>(base + L.hull);
    StarCounters *counters = reinterpret_cast<StarCounters *>(base + L.counters);
    uint32_t *big_rows = reinterpret_cast<uint32_t *>(base + L.big_rows);
    uint32_t *huge_list = reinterpret_cast<uint32_t *>(base + L.huge_list);
    uint32_t *huge_rows = reinterpret_cast<uint32_t *>(base + L.huge_rows);
    HugeStar *huge_arena = reinterpret_cast<HugeStar *>(base + L.huge_arena);
    const RowTables tables{rows, big_rows, huge_rows};
    const uint32_t depth = tree_depth_of(num_points);
    const uint32_t blocks = (num_points + 63u) / 64u;

    if (hipMemsetAsync(counters, 0, sizeof(StarCounters), s) != hipSuccess) return check_launch("rf_delaunay_adjacency");
    // first pass: a query gets this many tree nodes before its star is parked for the second pass (rf_star.hpp,
    // HullSet); RF_DELAUNAY_BUDGET=0 disables the parking (tuning / A-B only)
    static const uint32_t ghost_budget = [] {
        const char *e = getenv("RF_DELAUNAY_BUDGET");
        const long v = e ? atol(e) : 512;
        return v <= 0 ? 0xFFFFFFFFu : (uint32_t)v;
    }();
    // RF_DELAUNAY_WAVES (4 or 6 waves per SIMD; tuning only)
    static const int waves = [] {
        const char *e = getenv("RF_DELAUNAY_WAVES");
        return e ? atoi(e) : 6;
    }();
#define RF_LAUNCH_STARS(W)                                                                                        \
    hipLaunchKernelGGL(delaunay_star_kernel<W>, dim3(blocks), dim3(64), 0, s, points, num_points, aabb_tree, depth, \
                       seed_adjacency, seed_offsets, rows, degree, overflow, hull, ghost_budget, counters)
    if (waves <= 4) RF_LAUNCH_STARS(4);
    else RF_LAUNCH_STARS(6);
#undef RF_LAUNCH_STARS
    StarCounters host{};
    if (hipMemcpyAsync(&host, counters, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return check_launch("rf_delaunay_adjacency: star kernel");
    if (host.overflow > L.second_rows) {
        // more second-pass stars than this workspace has rows for: info[2] says how many, the caller comes back
        // with rf_delaunay_workspace_bytes_for(num_points, info[2])
        for (int k = 0; k < 12; ++k) info[k] = 0;
        info[2] = host.overflow;
        return fail(RF_ERR_WORKSPACE, "rf_delaunay_adjacency: more stars need the second pass than the workspace has rows for");
    }
    if (host.overflow) {
        // ghost queries of the second pass scan the hull candidates; a cloud with most of its points on its hull (a
        // sphere shell) keeps the tree instead
        const bool use_list = host.hull <= 65536u;
        static const int coop_waves = [] {   // tuning only
            const char *e = getenv("RF_DELAUNAY_COOP_WAVES");
            return e ? atoi(e) : 8;
        }();
#define RF_LAUNCH_COOP(W, G)                                                                                         \
    hipLaunchKernelGGL((delaunay_star_coop_kernel<W, G>), dim3(host.overflow), dim3(64 * W), 0, s, points, num_points, \
                       aabb_tree, depth, seed_adjacency, seed_offsets, overflow,                                     \
                       use_list ? hull : (const uint32_t *)nullptr, host.hull, host.overflow, rows, big_rows, degree, \
                       huge_list, counters)
        static const int coop_group = [] {   // tuning only: waves per query
            const char *e = getenv("RF_DELAUNAY_COOP_GROUP");
            return e ? atoi(e) : 8;
        }();
        // (8 x 8: all the waves of a block on one query at a time -- measured 120 / 192 ms for the whole build of 2 M points,
        // incrementally / from scratch, against 135 / 233 with four queries of one wave each: profiles/r06/t_coop_groups.log)
        if (coop_waves >= 16) RF_LAUNCH_COOP(16, 16);
        else if (coop_waves >= 8 && coop_group >= 8) RF_LAUNCH_COOP(8, 8);
        else if (coop_waves >= 8) RF_LAUNCH_COOP(8, 4);
        else if (coop_waves >= 4 && coop_group >= 4) RF_LAUNCH_COOP(4, 4);
        else if (coop_waves >= 4) RF_LAUNCH_COOP(4, 1);
        else RF_LAUNCH_COOP(1, 1);
#undef RF_LAUNCH_COOP
        hipLaunchKernelGGL(mark_big_rows_kernel, dim3((host.overflow + 255u) / 256u), dim3(256), 0, s, overflow,
                           host.overflow, rows);
        // third pass: whatever the second handed on (usually nothing: the blocks look at the counter and leave)
        hipLaunchKernelGGL(delaunay_star_huge_kernel, dim3(kHugeStars), dim3(64 * kHugeWaves), 0, s, points, num_points,
                           aabb_tree, depth, overflow, use_list ? hull : (const uint32_t *)nullptr, host.hull, huge_list,
                           big_rows, huge_arena, huge_rows, rows, degree, counters);
    }
    size_t bytes = L.scan_bytes;
    if (rocprim::exclusive_scan(base + L.scan_temp, bytes, degree, point_adjacency_offsets, 0u, (size_t)num_points,
                                rocprim::plus<uint32_t>(), s) != hipSuccess)
        return fail(RF_ERR_LAUNCH, "rf_delaunay_adjacency: scan failed");
    hipLaunchKernelGGL(last_offset_kernel, dim3(1), dim3(1), 0, s, degree, point_adjacency_offsets, num_points);
    const uint32_t pb = (num_points + 255u) / 256u;
    hipLaunchKernelGGL(csr_gather_kernel, dim3(pb), dim3(256), 0, s, tables, degree, point_adjacency_offsets,
                       num_points, adjacency_capacity, point_adjacency, counters);
    hipLaunchKernelGGL(symmetry_kernel, dim3(pb), dim3(256), 0, s, tables, degree, num_points, counters);
    if (hipMemcpyAsync(&host, counters, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return check_launch("rf_delaunay_adjacency: assembly");
    info[0] = host.adjacency;
    info[1] = host.failed[star::kOverflow] + host.failed[star::kDegenerate] + host.failed[star::kBroken] +
              host.failed[star::kPending];
    info[2] = host.overflow;
    info[3] = host.failed[star::kDuplicate];
    info[4] = host.asymmetric;
    info[5] = host.nodes_lo;
    info[6] = host.nodes_hi;
    info[7] = host.inserted;
    info[8] = host.failed[star::kDegenerate];
    info[9] = host.failed[star::kBroken];
    info[10] = host.failed[star::kOverflow];
    info[11] = host.hull;
#ifdef RF_COOP_SECTIONS
    fprintf(stderr, "[coop sections] slowest block: star %u (first-pass link %u vertices, %u triangles at the end) whole %u seed %u queries %u surgery %u kcycles, %u rounds; hull candidates %u, second-pass stars %u\n",
            host.sec_who, host.sec_deg, host.sec_nt, host.sec_max[3], host.sec_max[0], host.sec_max[1], host.sec_max[2], host.sec_rounds, host.hull, host.overflow);
#endif
    return check_launch("rf_delaunay_adjacency");
}

}  // extern "C"
