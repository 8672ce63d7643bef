// rf_adjacency.hip -- the two sort-based helpers of the library (rocPRIM device radix sort):
//   * rf_build_ray_order: a coherent processing order for shuffled ray batches (bottom of the file);
//   * rf_build_adjacency: CSR point adjacency from the tetrahedra of a Delaunay triangulation, the
//     on-"wire" format the tracer consumes.
//
// Reference: find_adjacency, src/delaunay/delaunay.cu:140-229 -- edges of every tet, merge sort,
// unique, both directions, stable sort by source, offsets where the source changes.  The result
// there is, per point, its neighbours in ascending order (stable merge of two ascending runs).
// Here: every tet emits its 12 DIRECTED edges as 64-bit keys (source << 32 | target); one radix sort
// of the keys (rocPRIM device primitive -- a plain library sort, nothing to hand-write), unique,
// and the CSR falls out: adjacency = the low words, offsets = lower bounds of (p << 32).  Same
// lists in the same order.  Points that no tet references get an empty range (the reference leaves
// 0xFFFFFFFF in their offsets; a Delaunay triangulation of all points has none).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <stdint.h>

#include "../../include/radfoam_hip.h"
#include "rf_foam.hpp"
#include "rf_host.hpp"

namespace rf {

using u64 = unsigned long long;
constexpr u64 kBadEdge = ~0ull;   // sorts behind every real key

__global__ __launch_bounds__(256) void tet_edges_kernel(const uint32_t *__restrict__ tets, uint32_t num_tets,
                                                        uint32_t num_points, u64 *__restrict__ keys) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= num_tets) return;
    const uint4 v = reinterpret_cast<const uint4 *>(tets)[t];
    const uint32_t id[4] = {v.x, v.y, v.z, v.w};
    u64 *out = keys + 12ull * t;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (a == b) continue;
            const bool ok = id[a] < num_points && id[b] < num_points && id[a] != id[b];
            out[k++] = ok ? ((u64)id[a] << 32) | (u64)id[b] : kBadEdge;
        }
}

// offsets[p] = first sorted unique key >= p << 32, for p = 0..num_points (the last one is E)
__global__ __launch_bounds__(256) void csr_offsets_kernel(const u64 *__restrict__ keys,
                                                          const uint32_t *__restrict__ num_unique,
                                                          uint32_t num_points, uint32_t *__restrict__ offsets,
                                                          uint32_t *__restrict__ adjacency_size) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > num_points) return;
    const u64 want = (u64)p << 32;
    uint32_t lo = 0, hi = *num_unique;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    offsets[p] = lo;
    if (p == num_points) *adjacency_size = lo;
}

__global__ __launch_bounds__(256) void csr_targets_kernel(const u64 *__restrict__ keys,
                                                          const uint32_t *__restrict__ adjacency_size,
                                                          uint32_t *__restrict__ adjacency) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < *adjacency_size) adjacency[k] = (uint32_t)(keys[k] & 0xFFFFFFFFull);
}

// ---- ray ordering ---------------------------------------------------------------------------------
// key = entry cell << 32 | Morton(u, v), (u, v) the 16-bit octahedral coordinates of the direction

__device__ __forceinline__ uint32_t spread16(uint32_t x) {
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

__global__ __launch_bounds__(256) void ray_keys_kernel(const float *__restrict__ rays,
                                                       const uint32_t *__restrict__ start, uint32_t num_rays,
                                                       u64 *__restrict__ keys, uint32_t *__restrict__ index) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_rays) return;
    const float dx = rays[6 * (size_t)i + 3], dy = rays[6 * (size_t)i + 4], dz = rays[6 * (size_t)i + 5];
    const float l1 = fabsf(dx) + fabsf(dy) + fabsf(dz);
    float px = 0.0f, py = 0.0f;
    if (l1 > 0.0f && l1 == l1) {
        px = dx / l1;
        py = dy / l1;
        if (dz < 0.0f) {
            const float ox = (1.0f - fabsf(py)) * (px >= 0.0f ? 1.0f : -1.0f);
            const float oy = (1.0f - fabsf(px)) * (py >= 0.0f ? 1.0f : -1.0f);
            px = ox;
            py = oy;
        }
    }
    const uint32_t u = (uint32_t)fminf(fmaxf((px * 0.5f + 0.5f) * 65535.0f, 0.0f), 65535.0f);
    const uint32_t v = (uint32_t)fminf(fmaxf((py * 0.5f + 0.5f) * 65535.0f, 0.0f), 65535.0f);
    keys[i] = ((u64)start[i] << 32) | (u64)(spread16(u) | (spread16(v) << 1));
    index[i] = i;
}

struct RayOrderLayout {
    size_t keys_a, keys_b, idx_a, temp, temp_bytes, total;
};

static RayOrderLayout ray_order_layout(uint32_t num_rays) {
    RayOrderLayout L{};
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (u64 *)nullptr, (u64 *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)num_rays, 0, 64, (hipStream_t)0);
    L.temp_bytes = bytes;
    L.keys_a = 0;
    L.keys_b = align_up((size_t)num_rays * 8, 256);
    L.idx_a = L.keys_b + align_up((size_t)num_rays * 8, 256);
    L.temp = L.idx_a + align_up((size_t)num_rays * 4, 256);
    L.total = L.temp + align_up(bytes, 256);
    return L;
}

struct AdjacencyLayout {
    size_t keys_a, keys_b, count, temp, temp_bytes, total;
};

static AdjacencyLayout adjacency_layout(uint32_t num_tets) {
    AdjacencyLayout L{};
    const size_t n = 12ull * num_tets;
    size_t sort_bytes = 0, unique_bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, sort_bytes, (u64 *)nullptr, (u64 *)nullptr, n, 0, 64, (hipStream_t)0);
    (void)rocprim::unique(nullptr, unique_bytes, (u64 *)nullptr, (u64 *)nullptr, (uint32_t *)nullptr, n,
                          rocprim::equal_to<u64>(), (hipStream_t)0);
    L.temp_bytes = sort_bytes > unique_bytes ? sort_bytes : unique_bytes;
    L.keys_a = 0;
    L.keys_b = align_up(n * 8, 256);
    L.count = L.keys_b + align_up(n * 8, 256);
    L.temp = L.count + 256;
    L.total = L.temp + align_up(L.temp_bytes, 256);
    return L;
}

}  // namespace rf

using namespace rf;

extern "C" {

size_t rf_ray_order_workspace_bytes(uint32_t num_rays) { return ray_order_layout(num_rays).total; }

int rf_build_ray_order(const float *rays, const uint32_t *start_point_index, uint32_t num_rays,
                       uint32_t *ray_order, void *workspace, size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (num_rays == 0) return RF_OK;
    if (!rays || !start_point_index || !ray_order)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_ray_order: null pointer");
    const RayOrderLayout L = ray_order_layout(num_rays);
    if (!workspace || workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_ray_order_workspace_bytes()");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(workspace);
    u64 *keys_a = reinterpret_cast<u64 *>(base + L.keys_a);
    u64 *keys_b = reinterpret_cast<u64 *>(base + L.keys_b);
    uint32_t *idx_a = reinterpret_cast<uint32_t *>(base + L.idx_a);
    hipLaunchKernelGGL(ray_keys_kernel, dim3((num_rays + 255u) / 256u), dim3(256), 0, s, rays, start_point_index,
                       num_rays, keys_a, idx_a);
    size_t bytes = L.temp_bytes;
    if (rocprim::radix_sort_pairs(base + L.temp, bytes, keys_a, keys_b, idx_a, ray_order, (size_t)num_rays, 0, 64,
                                  s) != hipSuccess)
        return fail(RF_ERR_LAUNCH, "rf_build_ray_order: radix sort failed");
    return check_launch("rf_build_ray_order");
}

size_t rf_adjacency_workspace_bytes(uint32_t num_tets) { return adjacency_layout(num_tets).total; }

int rf_build_adjacency(const uint32_t *tets, uint32_t num_tets, uint32_t num_points, uint32_t *point_adjacency,
                       uint32_t *point_adjacency_offsets, uint32_t *point_adjacency_size, void *workspace,
                       size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (!point_adjacency_offsets || !point_adjacency_size || (num_tets && (!tets || !point_adjacency)))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_adjacency: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (num_tets == 0) {
        (void)hipMemsetAsync(point_adjacency_offsets, 0, ((size_t)num_points + 1) * 4, s);
        (void)hipMemsetAsync(point_adjacency_size, 0, 4, s);
        return RF_OK;
    }
    const AdjacencyLayout L = adjacency_layout(num_tets);
    if (!workspace || workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_adjacency_workspace_bytes()");
    char *base = static_cast<char *>(workspace);
    u64 *keys_a = reinterpret_cast<u64 *>(base + L.keys_a);
    u64 *keys_b = reinterpret_cast<u64 *>(base + L.keys_b);
    uint32_t *count = reinterpret_cast<uint32_t *>(base + L.count);
    void *temp = base + L.temp;
    const size_t n = 12ull * num_tets;
    hipLaunchKernelGGL(tet_edges_kernel, dim3((num_tets + 255u) / 256u), dim3(256), 0, s, tets, num_tets, num_points,
                       keys_a);
    size_t bytes = L.temp_bytes;
    if (rocprim::radix_sort_keys(temp, bytes, keys_a, keys_b, n, 0, 64, s) != hipSuccess)
        return fail(RF_ERR_LAUNCH, "rf_build_adjacency: radix sort failed");
    bytes = L.temp_bytes;
    if (rocprim::unique(temp, bytes, keys_b, keys_a, count, n, rocprim::equal_to<u64>(), s) != hipSuccess)
        return fail(RF_ERR_LAUNCH, "rf_build_adjacency: unique failed");
    hipLaunchKernelGGL(csr_offsets_kernel, dim3((num_points + 256u) / 256u), dim3(256), 0, s, keys_a, count,
                       num_points, point_adjacency_offsets, point_adjacency_size);
    hipLaunchKernelGGL(csr_targets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys_a,
                       point_adjacency_size, point_adjacency);
    return check_launch("rf_build_adjacency");
}

}  // extern "C"
