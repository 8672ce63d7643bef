// rf_scene_ops.hip -- the reference's per-iteration work on either side of the tracer
// (SURVEY.md 8(f)), as gfx950 kernels behind the same C-ABI.
//
//   pack_attributes_kernel / _backward   RadFoamScene.get_trace_data, radfoam_model/scene.py:202-217:
//                                        attributes = cat[att_dc, att_sh, scale * softplus(density, beta=10)]
//                                        .to(attr_dtype) -- three torch ops (two concatenations and a cast, each
//                                        re-materialising N*A scalars) fused into one pass, plus its backward
//   nearest_point_kernel                 radfoam.nn for camera positions (torch_bindings/triangulation_bindings.cpp
//                                        :142-181, src/aabb_tree/aabb_tree.cu:343-415): exact, brute force --
//                                        24 MB of points per pass is microseconds of HBM time for the handful
//                                        of cameras scene.py:224-234 / benchmark.py:89 ask about
//   farthest_neighbor_kernel             src/delaunay/triangulation_ops.cu:9-44 (densification statistics)
//
// All HBM-bound streaming / gather kernels; no LDS tiling is needed beyond the block reductions.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/radfoam_hip.h"
#include "rf_foam.hpp"
#include "rf_host.hpp"
#include "rf_math.hpp"

namespace rf {

// log(1 + e) for e >= 0 without the cancellation of forming 1 + e first (Kahan: log(u) * e / (u - 1))
__device__ __forceinline__ float log1p_(float e) {
    const float u = 1.0f + e;
    if (u == 1.0f) return e;
    return log_(u) * (e / (u - 1.0f));
}

constexpr float kSoftplusBeta = 10.0f;        // scene.py:203  F.softplus(self.density, beta=10)
constexpr float kSoftplusThreshold = 20.0f;   // torch default: linear above beta * x = 20

__device__ __forceinline__ float softplus_(float x) {
    const float z = x * kSoftplusBeta;
    return z > kSoftplusThreshold ? x : log1p_(exp_(z)) / kSoftplusBeta;
}

// d softplus / dx, the way torch's softplus_backward evaluates it: z / (z + 1) with z = exp(beta x)
__device__ __forceinline__ float softplus_grad_(float x) {
    const float z = x * kSoftplusBeta;
    if (z > kSoftplusThreshold) return 1.0f;
    const float ez = exp_(z);
    return ez / (ez + 1.0f);
}

template <typename T>
__device__ __forceinline__ T to_attr(float v);
template <>
__device__ __forceinline__ float to_attr<float>(float v) { return v; }
template <>
__device__ __forceinline__ uint16_t to_attr<uint16_t>(float v) { return float_to_half_bits(v); }

// one thread per output scalar: writes are fully coalesced, the three sources are read in runs
template <typename T>
__global__ __launch_bounds__(256) void pack_attributes_kernel(const float *__restrict__ att_dc,
                                                              const float *__restrict__ att_sh,
                                                              const float *__restrict__ density, float scale,
                                                              uint32_t num_points, uint32_t attr_dim,
                                                              T *__restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)num_points * attr_dim;
    if (idx >= total) return;
    const uint32_t row = (uint32_t)(idx / attr_dim), col = (uint32_t)(idx - (size_t)row * attr_dim);
    const uint32_t nsh = attr_dim - 4u;   // SH coefficients beyond the DC term
    float v;
    if (col < 3u) v = att_dc[(size_t)row * 3u + col];
    else if (col < attr_dim - 1u) v = att_sh[(size_t)row * nsh + (col - 3u)];
    else v = scale * softplus_(density[row]);
    out[idx] = to_attr<T>(v);
}

__global__ __launch_bounds__(256) void pack_attributes_backward_kernel(
    const float *__restrict__ density, float scale, uint32_t num_points, uint32_t attr_dim,
    const float *__restrict__ attr_grad, float *__restrict__ att_dc_grad, float *__restrict__ att_sh_grad,
    float *__restrict__ density_grad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)num_points * attr_dim;
    if (idx >= total) return;
    const uint32_t row = (uint32_t)(idx / attr_dim), col = (uint32_t)(idx - (size_t)row * attr_dim);
    const uint32_t nsh = attr_dim - 4u;
    const float g = attr_grad[idx];
    if (col < 3u) att_dc_grad[(size_t)row * 3u + col] = g;
    else if (col < attr_dim - 1u) att_sh_grad[(size_t)row * nsh + (col - 3u)] = g;
    else density_grad[row] = (g * scale) * softplus_grad_(density[row]);
}

// ---- nearest point --------------------------------------------------------------------------------
// key = squared distance bits << 32 | index: for non-negative floats the bit pattern orders like
// the value, so a 64-bit integer minimum is the nearest point, the lowest index among exact ties.
constexpr uint32_t kNnPerThread = 8;

__device__ __forceinline__ unsigned long long min_u64(unsigned long long a, unsigned long long b) {
    return a < b ? a : b;
}

__global__ __launch_bounds__(256) void nearest_point_kernel(const float *__restrict__ points, uint32_t num_points,
                                                            const float *__restrict__ queries,
                                                            uint32_t num_queries,
                                                            unsigned long long *__restrict__ best) {
    __shared__ unsigned long long s_wave[4];
    const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * kNnPerThread;
    float px[kNnPerThread], py[kNnPerThread], pz[kNnPerThread];
#pragma unroll
    for (uint32_t j = 0; j < kNnPerThread; ++j) {
        const uint32_t i = base + j;
        const bool ok = i < num_points;
        px[j] = ok ? points[3 * (size_t)i] : 0.0f;
        py[j] = ok ? points[3 * (size_t)i + 1] : 0.0f;
        pz[j] = ok ? points[3 * (size_t)i + 2] : 0.0f;
    }
    for (uint32_t q = 0; q < num_queries; ++q) {
        const float qx = queries[3 * (size_t)q], qy = queries[3 * (size_t)q + 1], qz = queries[3 * (size_t)q + 2];
        unsigned long long key = ~0ull;
#pragma unroll
        for (uint32_t j = 0; j < kNnPerThread; ++j) {
            const float dx = px[j] - qx, dy = py[j] - qy, dz = pz[j] - qz;
            const float d2 = dot3(dx, dy, dz, dx, dy, dz);
            const unsigned long long k = ((unsigned long long)f2bits(d2) << 32) | (unsigned long long)(base + j);
            // NaN distances (non-finite input) never win; out-of-range slots neither
            key = (base + j < num_points && d2 == d2) ? min_u64(key, k) : key;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = min_u64(key, o);
        }
        if ((threadIdx.x & 63u) == 0u) s_wave[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long m = min_u64(min_u64(s_wave[0], s_wave[1]), min_u64(s_wave[2], s_wave[3]));
            if (m != ~0ull) atomicMin(best + q, m);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void nearest_point_finish_kernel(const unsigned long long *__restrict__ best,
                                                                   uint32_t num_queries,
                                                                   uint32_t *__restrict__ indices) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < num_queries) indices[q] = (uint32_t)(best[q] & 0xFFFFFFFFull);
}

// ---- nearest point through the AABB tree ---------------------------------------------------------------
// radfoam.nn as the reference answers it (nn_kernel, src/aabb_tree/aabb_tree.cu:343-415 over aabb_tree.cuh:154-276:
// a warp of 32 per query, k candidates kept by a warp merge sort, ballot-driven stack).  Here k = 1 is all the callers
// ask for (the entry cell of a ray origin), so a LANE takes a query and walks the implicit tree depth first, nearer
// child first, with its stack in private memory: O(log N) boxes per query instead of the N points of the brute-force
// kernel above, which stays the choice for a handful of camera positions.  Same answer as that kernel, bit for bit:
// the point distance is the same expression, (distance, index) pairs are compared as one 64-bit key (lowest index among
// exact ties), and a box is skipped only when its distance is strictly larger than the best so far -- fp32 rounding is
// monotone through the subtraction, the squares and the fused sums, so a box's distance never exceeds that of a point
// inside it.  Tree layout (build_aabb_tree, aabb_tree.cu:192-292): P = pow2_round_up(N) nodes of {min[3], max[3]};
// level d (2^d nodes) starts at node P - 2^(d+1); the deepest level pairs the points (2k, 2k+1).
__device__ __forceinline__ float box_distance2(const float *__restrict__ node, float qx, float qy, float qz) {
    const float dx = __builtin_fmaxf(__builtin_fmaxf(node[0] - qx, qx - node[3]), 0.0f);
    const float dy = __builtin_fmaxf(__builtin_fmaxf(node[1] - qy, qy - node[4]), 0.0f);
    const float dz = __builtin_fmaxf(__builtin_fmaxf(node[2] - qz, qz - node[5]), 0.0f);
    return dot3(dx, dy, dz, dx, dy, dz);
}

__global__ __launch_bounds__(256) void nearest_point_tree_kernel(const float *__restrict__ points, uint32_t num_points,
                                                                 const float *__restrict__ tree, uint32_t pow2,
                                                                 uint32_t depth, const float *__restrict__ queries,
                                                                 uint32_t num_queries, uint32_t *__restrict__ indices) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= num_queries) return;
    const float qx = queries[3 * (size_t)q], qy = queries[3 * (size_t)q + 1], qz = queries[3 * (size_t)q + 2];
    unsigned long long best = ~0ull;          // distance bits << 32 | index
    float best_d2 = __builtin_inff();
    auto try_point = [&](uint32_t i) {
        if (i >= num_points) return;
        const float dx = points[3 * (size_t)i] - qx, dy = points[3 * (size_t)i + 1] - qy, dz = points[3 * (size_t)i + 2] - qz;
        const float d2 = dot3(dx, dy, dz, dx, dy, dz);
        if (d2 == d2) {
            const unsigned long long k = ((unsigned long long)f2bits(d2) << 32) | (unsigned long long)i;
            if (k < best) {
                best = k;
                best_d2 = d2;
            }
        }
    };
    if (depth == 0) {   // one point
        try_point(0u);
        indices[q] = (uint32_t)(best & 0xFFFFFFFFull);
        return;
    }
    uint32_t stack[32];                       // (level << 26) | node of that level; level < 32, node < 2^26
    int sp = 0;
    stack[sp++] = 0u;                         // the root: level 0, node 0
    while (sp > 0) {
        const uint32_t e = stack[--sp];
        const uint32_t d = e >> 26, k = e & 0x03FFFFFFu;
        const float *node = tree + 6 * (size_t)(pow2 - (2u << d) + k);
        if (box_distance2(node, qx, qy, qz) > best_d2) continue;   // (a NaN query compares false: it visits everything)
        if (d + 1u == depth) {                // the deepest level: its two points
            try_point(2u * k);
            try_point(2u * k + 1u);
            continue;
        }
        const uint32_t c = 2u * k;
        const float *lo = tree + 6 * (size_t)(pow2 - (4u << d) + c);
        const float d0 = box_distance2(lo, qx, qy, qz), d1 = box_distance2(lo + 6, qx, qy, qz);
        const uint32_t near = d1 < d0 ? c + 1u : c, far = d1 < d0 ? c : c + 1u;
        const float dn = d1 < d0 ? d1 : d0, df = d1 < d0 ? d0 : d1;
        if (!(df > best_d2)) stack[sp++] = ((d + 1u) << 26) | far;
        if (!(dn > best_d2)) stack[sp++] = ((d + 1u) << 26) | near;
    }
    indices[q] = (uint32_t)(best & 0xFFFFFFFFull);
}

// ---- farthest neighbour -----------------------------------------------------------------------------
// reference: farthest_neighbor_kernel, triangulation_ops.cu:9-44 -- the first strict maximum of the
// neighbour distances (UINT32_MAX when there is none), and the mean half-distance; the reference's
// `sum += 0.5 * distance` promotes to double for the addition, restated as such.
__global__ __launch_bounds__(256) void farthest_neighbor_kernel(const float *__restrict__ points,
                                                                const uint32_t *__restrict__ adj,
                                                                const uint32_t *__restrict__ offsets,
                                                                uint32_t num_points,
                                                                uint32_t *__restrict__ indices,
                                                                float *__restrict__ cell_radius) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_points) return;
    const float px = points[3 * (size_t)i], py = points[3 * (size_t)i + 1], pz = points[3 * (size_t)i + 2];
    const uint32_t b = offsets[i], e = offsets[i + 1];
    uint32_t far = kNone;
    float sum = 0.0f, best = 0.0f;
    for (uint32_t f = b; f < e; ++f) {
        const uint32_t q = adj[f];
        const float dx = points[3 * (size_t)q] - px, dy = points[3 * (size_t)q + 1] - py,
                    dz = points[3 * (size_t)q + 2] - pz;
        const float d = sqrtf(dot3(dx, dy, dz, dx, dy, dz));
        sum = (float)((double)sum + 0.5 * (double)d);
        if (d > best) {
            best = d;
            far = q;
        }
    }
    indices[i] = far;
    cell_radius[i] = sum / (float)(e - b);
}

// fp32 accumulator -> attribute type, one scalar per thread (rf_cast_accumulator)
template <typename T>
__global__ __launch_bounds__(256) void cast_accumulator_kernel(const float *__restrict__ src, size_t count,
                                                               T *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = to_attr<T>(src[i]);
}

}  // namespace rf

using namespace rf;

// ------------------------------------------------------------------------------------------
// radfoam.BatchFetcher on the device (src/utils/batch_fetcher.cpp:60-77).  The reference gathers rows of a HOST array on
// a worker thread and uploads them, four batches ahead; with 288 GB of HBM the whole training set (rays, colours, alphas
// of every view: a few GB) lives on the device and a batch is one gather kernel: thread per 4-byte word of the batch,
// row j of batch b taken from  randint(make_rng(b * batch_size + j), 0, n)  (random.h:13-57, restated below: the
// hash-prospector mixer, then x / (0xffffffff / n) clamped to n - 1) or from (b * batch_size + j) % n without shuffling
// -- the same index sequence, so fetchers over rays / colours / alphas stay aligned as the reference's do.
__device__ __forceinline__ uint32_t rng_mix(uint32_t x) {
    x ^= x >> 17;
    x *= 0xED5AD4BBu;
    x ^= x >> 11;
    x *= 0xAC4C1B51u;
    x ^= x >> 15;
    x *= 0x31848BABu;
    x ^= x >> 14;
    return x;
}

__global__ __launch_bounds__(256) void fetch_batch_kernel(const uint32_t *__restrict__ data, uint64_t num_elements,
                                                          uint32_t words_per_row, uint64_t first, uint32_t batch_size,
                                                          int shuffle, uint32_t *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= (uint64_t)batch_size * words_per_row) return;
    const uint32_t j = (uint32_t)(t / words_per_row), w = (uint32_t)(t - (uint64_t)j * words_per_row);
    const uint64_t seq = first + j;                                  // batch_idx * batch_size + j
    uint64_t row;
    if (shuffle) {
        const uint32_t n = (uint32_t)num_elements;
        uint32_t x = rng_mix((uint32_t)seq ^ 0x2815DB5Bu) / (0xFFFFFFFFu / n);
        row = x < n - 1u ? x : n - 1u;
    } else {
        row = seq % num_elements;
    }
    out[t] = data[row * words_per_row + w];
}


extern "C" {

int rf_pack_attributes(int sh_degree, int attr_type, uint32_t num_points, const float *att_dc,
                       const float *att_sh, const float *density, float activation_scale, void *attributes,
                       void *stream) {
    g_err[0] = 0;
    const uint32_t A = attribute_dim(sh_degree);
    if (A == 0 || (attr_type != RF_ATTR_FLOAT32 && attr_type != RF_ATTR_FLOAT16))
        return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported SH degree or attribute type");
    if (num_points == 0) return RF_OK;
    if (!att_dc || !density || !attributes || (A > 4 && !att_sh))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_pack_attributes: null pointer");
    const size_t total = (size_t)num_points * A;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (attr_type == RF_ATTR_FLOAT16)
        hipLaunchKernelGGL(pack_attributes_kernel<uint16_t>, grid, block, 0, s, att_dc, att_sh, density,
                           activation_scale, num_points, A, static_cast<uint16_t *>(attributes));
    else
        hipLaunchKernelGGL(pack_attributes_kernel<float>, grid, block, 0, s, att_dc, att_sh, density,
                           activation_scale, num_points, A, static_cast<float *>(attributes));
    return check_launch("rf_pack_attributes");
}

int rf_pack_attributes_backward(int sh_degree, uint32_t num_points, const float *density,
                                float activation_scale, const float *attr_grad, float *att_dc_grad,
                                float *att_sh_grad, float *density_grad, void *stream) {
    g_err[0] = 0;
    const uint32_t A = attribute_dim(sh_degree);
    if (A == 0) return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported SH degree");
    if (num_points == 0) return RF_OK;
    if (!density || !attr_grad || !att_dc_grad || !density_grad || (A > 4 && !att_sh_grad))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_pack_attributes_backward: null pointer");
    const size_t total = (size_t)num_points * A;
    hipLaunchKernelGGL(pack_attributes_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), density, activation_scale, num_points, A, attr_grad,
                       att_dc_grad, att_sh_grad, density_grad);
    return check_launch("rf_pack_attributes_backward");
}

int rf_nearest_point(const float *points, uint32_t num_points, const float *queries, uint32_t num_queries,
                     uint32_t *indices, void *scratch, void *stream) {
    g_err[0] = 0;
    if (num_queries == 0) return RF_OK;
    if (num_points == 0) return fail(RF_ERR_INVALID_ARGUMENT, "rf_nearest_point: no points");
    if (!points || !queries || !indices || !scratch)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_nearest_point: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned long long *best = static_cast<unsigned long long *>(scratch);
    (void)hipMemsetAsync(best, 0xFF, (size_t)num_queries * 8, s);
    const uint32_t per_block = 256u * kNnPerThread;
    hipLaunchKernelGGL(nearest_point_kernel, dim3((num_points + per_block - 1u) / per_block), dim3(256), 0, s,
                       points, num_points, queries, num_queries, best);
    hipLaunchKernelGGL(nearest_point_finish_kernel, dim3((num_queries + 255u) / 256u), dim3(256), 0, s, best,
                       num_queries, indices);
    return check_launch("rf_nearest_point");
}

int rf_nearest_point_tree(const float *points, uint32_t num_points, const float *aabb_tree, const float *queries,
                          uint32_t num_queries, uint32_t *indices, void *stream) {
    g_err[0] = 0;
    if (num_queries == 0) return RF_OK;
    if (num_points == 0) return fail(RF_ERR_INVALID_ARGUMENT, "rf_nearest_point_tree: no points");
    if (num_points > (1u << 26)) return fail(RF_ERR_INVALID_ARGUMENT, "rf_nearest_point_tree: more than 2^26 points");
    if (!points || !aabb_tree || !queries || !indices)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_nearest_point_tree: null pointer");
    uint32_t pow2 = 1, depth = 0;
    while (pow2 < num_points) {
        pow2 <<= 1;
        ++depth;
    }
    hipLaunchKernelGGL(nearest_point_tree_kernel, dim3((num_queries + 255u) / 256u), dim3(256), 0,
                       static_cast<hipStream_t>(stream), points, num_points, aabb_tree, pow2, depth, queries, num_queries,
                       indices);
    return check_launch("rf_nearest_point_tree");
}

int rf_farthest_neighbor(const float *points, uint32_t num_points, const uint32_t *point_adjacency,
                         const uint32_t *point_adjacency_offsets, uint32_t *indices, float *cell_radius,
                         void *stream) {
    g_err[0] = 0;
    if (num_points == 0) return RF_OK;
    if (!points || !point_adjacency || !point_adjacency_offsets || !indices || !cell_radius)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_farthest_neighbor: null pointer");
    hipLaunchKernelGGL(farthest_neighbor_kernel, dim3((num_points + 255u) / 256u), dim3(256), 0,
                       static_cast<hipStream_t>(stream), points, point_adjacency, point_adjacency_offsets,
                       num_points, indices, cell_radius);
    return check_launch("rf_farthest_neighbor");
}


static int fetch_batch_launch(const char *who, const void *data, uint64_t num_elements, uint32_t stride_bytes,
                              uint64_t first, uint32_t count, int shuffle, void *out, void *stream) {
    g_err[0] = 0;
    if (count == 0) return RF_OK;
    if (!data || !out) return fail(RF_ERR_INVALID_ARGUMENT, "%s: null pointer", who);
    if (num_elements == 0) return fail(RF_ERR_INVALID_ARGUMENT, "%s: no elements", who);
    if (num_elements > 0xFFFFFFFFull) return fail(RF_ERR_INVALID_ARGUMENT, "Too many elements");
    if (stride_bytes == 0 || (stride_bytes & 3u))
        return fail(RF_ERR_INVALID_ARGUMENT, "%s: stride must be a positive multiple of 4 bytes", who);
    const uint32_t words = stride_bytes / 4u;
    const uint64_t total = (uint64_t)count * words;
    if (total > 0xFFFFFFFFull * 256ull) return fail(RF_ERR_INVALID_ARGUMENT, "%s: batch too large", who);
    hipLaunchKernelGGL(fetch_batch_kernel, dim3((unsigned)((total + 255u) / 256u)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint32_t *>(data), num_elements, words,
                       first, count, shuffle, static_cast<uint32_t *>(out));
    return check_launch(who);
}

int rf_fetch_batch(const void *data, uint64_t num_elements, uint32_t stride_bytes, uint32_t batch_index,
                   uint32_t batch_size, int shuffle, void *out, void *stream) {
    return fetch_batch_launch("rf_fetch_batch", data, num_elements, stride_bytes,
                              (uint64_t)batch_index * (uint64_t)batch_size, batch_size, shuffle, out, stream);
}

int rf_fetch_batch_range(const void *data, uint64_t num_elements, uint32_t stride_bytes, uint64_t first_sequence,
                         uint32_t count, int shuffle, void *out, void *stream) {
    return fetch_batch_launch("rf_fetch_batch_range", data, num_elements, stride_bytes, first_sequence, count, shuffle,
                              out, stream);
}


int rf_adjacency_size(uint32_t num_points, const uint32_t *point_adjacency_offsets, uint32_t *point_adjacency_size,
                      void *stream) {
    g_err[0] = 0;
    if (!point_adjacency_offsets || !point_adjacency_size)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_adjacency_size: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(point_adjacency_size, point_adjacency_offsets + num_points, sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(RF_ERR_LAUNCH, "rf_adjacency_size: %s", hipGetErrorString(e));
    return RF_OK;
}

int rf_cast_accumulator(const float *src, size_t count, int attr_type, void *dst, void *stream) {
    g_err[0] = 0;
    if (count == 0) return RF_OK;
    if (!src || !dst) return fail(RF_ERR_INVALID_ARGUMENT, "rf_cast_accumulator: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    if (attr_type == RF_ATTR_FLOAT16)
        hipLaunchKernelGGL(cast_accumulator_kernel<uint16_t>, grid, block, 0, s, src, count, static_cast<uint16_t *>(dst));
    else if (attr_type == RF_ATTR_FLOAT32)
        hipLaunchKernelGGL(cast_accumulator_kernel<float>, grid, block, 0, s, src, count, static_cast<float *>(dst));
    else
        return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported attribute type");
    return check_launch("rf_cast_accumulator");
}

}  // extern "C"
