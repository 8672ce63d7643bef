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
//   delaunay_star_big_kernel stars that outgrow that (hub points, long hull facets) are redone with room for 250
//                            neighbours in a global arena, a few hundred of 2M points
//   csr_*                    degrees -> offsets (rocPRIM scan) -> adjacency, ascending per point like find_adjacency's
//   symmetry_kernel          j in N(i) <=> i in N(j): exact predicates make independent stars agree; cospherical
//                            input can break that, which is reported (the reference throws "ambiguous triangulation")
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
#include "rf_star.hpp"

namespace rf {

constexpr int kSmallV = 64, kSmallT = 124;
constexpr int kBigV = 250, kBigT = 496;
constexpr int kBlockSeeds = 12;
constexpr uint32_t kBigFlag = 0x80000000u;   // degree word of a star kept by the large instance: flag | arena slot

using SmallStar = star::Star<kSmallV, kSmallT>;
using BigStar = star::Star<kBigV, kBigT>;

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
    uint32_t overflow;     // stars handed to the large instance
    uint32_t failed[5];    // stars by star::Status (index 0 unused)
    uint32_t asymmetric;   // directed edges without their reverse
    uint32_t adjacency;    // E
    uint32_t nodes_lo, nodes_hi;   // tree nodes visited (64-bit)
    uint32_t inserted;     // link insertions
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
        if (ns >= 3) return ns;
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

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(64, WAVES_PER_SIMD) void delaunay_star_kernel(const float *__restrict__ pts, uint32_t n,
                                                           const float *__restrict__ tree, uint32_t depth,
                                                           const uint32_t *__restrict__ seed_adj,
                                                           const uint32_t *__restrict__ seed_off,
                                                           uint32_t *__restrict__ rows, uint32_t *__restrict__ degree,
                                                           uint32_t *__restrict__ overflow_list,
                                                           StarCounters *__restrict__ counters) {
    __shared__ float block_pts[64 * 3];
    const uint32_t block_first = blockIdx.x * 64u;
    const uint32_t block_count = n - block_first < 64u ? n - block_first : 64u;
    for (uint32_t k = threadIdx.x; k < 3 * block_count; k += 64) block_pts[k] = pts[3 * (size_t)block_first + k];
    __syncthreads();
    const uint32_t i = block_first + threadIdx.x;
    if (i >= n) return;

    SmallStar s;
    star::star_reset(s, i, pts + 3 * (size_t)i);
    uint32_t seeds[kSmallV];
    const int ns = gather_seeds<kSmallV, kSmallT>(pts, n, i, seed_adj, seed_off, block_pts, block_first, block_count,
                                                  seeds, kSmallV - 1);
    const star::Tree tr{tree, n, depth};
    uint32_t visited = 0, inserted = 0;
    star::star_build(s, tr, pts, seeds, ns, visited, inserted);
    atomicAdd(&counters->inserted, inserted);
    const uint32_t old = atomicAdd(&counters->nodes_lo, visited);
    if (old + visited < old) atomicAdd(&counters->nodes_hi, 1u);

    if (s.status == star::kOverflow) {
        const uint32_t slot = atomicAdd(&counters->overflow, 1u);
        overflow_list[slot] = i;
        degree[i] = 0;
        return;
    }
    if (s.status != star::kOk) {
        atomicAdd(&counters->failed[s.status], 1u);
        degree[i] = 0;
        return;
    }
    bool hull;
    uint32_t nb[kSmallV];
    const int deg = star::star_neighbours(s, nb, 1, &hull);
    uint32_t *row = rows + (size_t)i * kSmallV;
    for (int k = 0; k < deg; ++k) row[k] = nb[k];
    degree[i] = (uint32_t)deg;
}

// one lane per star the small instance could not hold; state and result rows in global memory
__global__ __launch_bounds__(64) void delaunay_star_big_kernel(const float *__restrict__ pts, uint32_t n,
                                                               const float *__restrict__ tree, uint32_t depth,
                                                               const uint32_t *__restrict__ seed_adj,
                                                               const uint32_t *__restrict__ seed_off,
                                                               const uint32_t *__restrict__ overflow_list,
                                                               uint32_t first, uint32_t count,
                                                               BigStar *__restrict__ arena,
                                                               uint32_t *__restrict__ big_rows,
                                                               uint32_t *__restrict__ degree,
                                                               StarCounters *__restrict__ counters) {
    const uint32_t w = blockIdx.x * 64u + threadIdx.x;
    if (w >= count) return;
    const uint32_t i = overflow_list[first + w];
    BigStar &s = arena[w];
    star::star_reset(s, i, pts + 3 * (size_t)i);
    const uint32_t block_first = i & ~63u;
    const uint32_t block_count = n - block_first < 64u ? n - block_first : 64u;
    uint32_t *seeds = big_rows + (size_t)(first + w) * kBigV;   // the result row doubles as the seed list
    const int ns = gather_seeds<kBigV, kBigT>(pts, n, i, seed_adj, seed_off, pts + 3 * (size_t)block_first,
                                              block_first, block_count, seeds, kBigV - 1);
    const star::Tree tr{tree, n, depth};
    uint32_t visited = 0, inserted = 0;
    star::star_build(s, tr, pts, seeds, ns, visited, inserted);
    if (s.status != star::kOk) {
        atomicAdd(&counters->failed[s.status], 1u);
        degree[i] = 0;
        return;
    }
    bool hull;
    const int deg = star::star_neighbours(s, seeds, 1, &hull);
    degree[i] = (uint32_t)deg;
    // the slot is found again through the overflow list position: rows[i][0] keeps it
}

// rows of the large instance are addressed through rows[i * kSmallV] = position in the overflow list
__global__ __launch_bounds__(256) void mark_big_rows_kernel(const uint32_t *__restrict__ overflow_list, uint32_t count,
                                                            uint32_t *__restrict__ rows) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < count) rows[(size_t)overflow_list[w] * kSmallV] = w | kBigFlag;
}

__device__ __forceinline__ const uint32_t *row_of(const uint32_t *rows, const uint32_t *big_rows, uint32_t i,
                                                  uint32_t deg) {
    const uint32_t *row = rows + (size_t)i * kSmallV;
    if (deg && (row[0] & kBigFlag)) return big_rows + (size_t)(row[0] & ~kBigFlag) * kBigV;
    return row;
}

__global__ __launch_bounds__(256) void csr_gather_kernel(const uint32_t *__restrict__ rows,
                                                         const uint32_t *__restrict__ big_rows,
                                                         const uint32_t *__restrict__ degree,
                                                         const uint32_t *__restrict__ offsets, uint32_t n,
                                                         uint32_t capacity, uint32_t *__restrict__ adjacency,
                                                         StarCounters *__restrict__ counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t deg = degree[i], off = offsets[i];
    if (i == n - 1) counters->adjacency = off + deg;
    if (off + deg > capacity) return;
    const uint32_t *row = row_of(rows, big_rows, i, deg);
    for (uint32_t k = 0; k < deg; ++k) adjacency[off + k] = row[k];
}

__global__ __launch_bounds__(256) void symmetry_kernel(const uint32_t *__restrict__ rows,
                                                       const uint32_t *__restrict__ big_rows,
                                                       const uint32_t *__restrict__ degree, uint32_t n,
                                                       StarCounters *__restrict__ counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t deg = degree[i];
    const uint32_t *row = row_of(rows, big_rows, i, deg);
    uint32_t missing = 0;
    for (uint32_t k = 0; k < deg; ++k) {
        const uint32_t j = row[k];
        const uint32_t dj = degree[j];
        const uint32_t *rj = row_of(rows, big_rows, j, dj);
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
    size_t rows, degree, overflow, counters, big_rows, arena, scan_temp, scan_bytes, total;
    uint32_t arena_stars;
};

static DelaunayLayout delaunay_layout(uint32_t n) {
    DelaunayLayout L{};
    size_t scan_bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)n,
                                  rocprim::plus<uint32_t>(), (hipStream_t)0);
    L.scan_bytes = scan_bytes;
    // room for one star in 64 in the large instance; a batch loop covers more
    L.arena_stars = n / 64 < 1024 ? 1024 : n / 64;
    size_t at = 0;
    auto take = [&](size_t bytes) {
        const size_t here = at;
        at += align_up(bytes, 256);
        return here;
    };
    L.rows = take((size_t)n * kSmallV * 4);
    L.degree = take((size_t)n * 4);
    L.overflow = take((size_t)n * 4);
    L.counters = take(sizeof(StarCounters));
    L.big_rows = take((size_t)L.arena_stars * kBigV * 4);
    L.arena = take((size_t)L.arena_stars * sizeof(BigStar));
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

size_t rf_delaunay_workspace_bytes(uint32_t num_points) { return delaunay_layout(num_points).total; }

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
    const DelaunayLayout L = delaunay_layout(num_points);
    if (!workspace || workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_delaunay_workspace_bytes()");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(workspace);
    uint32_t *rows = reinterpret_cast<uint32_t *>(base + L.rows);
    uint32_t *degree = reinterpret_cast<uint32_t *>(base + L.degree);
    uint32_t *overflow = reinterpret_cast<uint32_t *>(base + L.overflow);
    StarCounters *counters = reinterpret_cast<StarCounters *>(base + L.counters);
    uint32_t *big_rows = reinterpret_cast<uint32_t *>(base + L.big_rows);
    BigStar *arena = reinterpret_cast<BigStar *>(base + L.arena);
    const uint32_t depth = tree_depth_of(num_points);
    const uint32_t blocks = (num_points + 63u) / 64u;

    if (hipMemsetAsync(counters, 0, sizeof(StarCounters), s) != hipSuccess) return check_launch("rf_delaunay_adjacency");
    // RF_DELAUNAY_WAVES (4, 6 or 8 waves per SIMD; tuning only): more waves hide more of the tree's load latency,
    // fewer keep more of the star out of scratch
    static const int waves = [] {
        const char *e = getenv("RF_DELAUNAY_WAVES");
        return e ? atoi(e) : 6;
    }();
#define RF_LAUNCH_STARS(W)                                                                                        \
    hipLaunchKernelGGL(delaunay_star_kernel<W>, dim3(blocks), dim3(64), 0, s, points, num_points, aabb_tree, depth, \
                       seed_adjacency, seed_offsets, rows, degree, overflow, counters)
    if (waves <= 4) RF_LAUNCH_STARS(4);
    else if (waves >= 8) RF_LAUNCH_STARS(8);
    else RF_LAUNCH_STARS(6);
#undef RF_LAUNCH_STARS
    StarCounters host{};
    if (hipMemcpyAsync(&host, counters, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return check_launch("rf_delaunay_adjacency: star kernel");
    if (host.overflow > L.arena_stars) {
        // more hub stars than the arena holds: not a point cloud this instance is sized for
        info[0] = 0; info[1] = host.overflow; info[2] = host.overflow;
        for (int k = 3; k < 8; ++k) info[k] = 0;
        return fail(RF_ERR_WORKSPACE, "rf_delaunay_adjacency: more stars need the large instance than its arena holds");
    }
    if (host.overflow) {
        hipLaunchKernelGGL(delaunay_star_big_kernel, dim3((host.overflow + 63u) / 64u), dim3(64), 0, s, points,
                           num_points, aabb_tree, depth, seed_adjacency, seed_offsets, overflow, 0u, host.overflow,
                           arena, big_rows, degree, counters);
        hipLaunchKernelGGL(mark_big_rows_kernel, dim3((host.overflow + 255u) / 256u), dim3(256), 0, s, overflow,
                           host.overflow, rows);
    }
    size_t bytes = L.scan_bytes;
    if (rocprim::exclusive_scan(base + L.scan_temp, bytes, degree, point_adjacency_offsets, 0u, (size_t)num_points,
                                rocprim::plus<uint32_t>(), s) != hipSuccess)
        return fail(RF_ERR_LAUNCH, "rf_delaunay_adjacency: scan failed");
    hipLaunchKernelGGL(last_offset_kernel, dim3(1), dim3(1), 0, s, degree, point_adjacency_offsets, num_points);
    const uint32_t pb = (num_points + 255u) / 256u;
    hipLaunchKernelGGL(csr_gather_kernel, dim3(pb), dim3(256), 0, s, rows, big_rows, degree, point_adjacency_offsets,
                       num_points, adjacency_capacity, point_adjacency, counters);
    hipLaunchKernelGGL(symmetry_kernel, dim3(pb), dim3(256), 0, s, rows, big_rows, degree, num_points, counters);
    if (hipMemcpyAsync(&host, counters, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return check_launch("rf_delaunay_adjacency: assembly");
    info[0] = host.adjacency;
    info[1] = host.failed[star::kOverflow] + host.failed[star::kDegenerate] + host.failed[star::kBroken];
    info[2] = host.overflow;
    info[3] = host.failed[star::kDuplicate];
    info[4] = host.asymmetric;
    info[5] = host.nodes_lo;
    info[6] = host.nodes_hi;
    info[7] = host.inserted;
    return check_launch("rf_delaunay_adjacency");
}

}  // extern "C"
