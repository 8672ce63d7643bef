// rf_tile_prior.hip -- a cost prior for the tiles of a frame that has never been traced.
//
// Which tiles are still running when a launch drains decides its last half millisecond (rf_kernels.hip, dealt_tile): a
// frame traced AGAIN takes its tiles longest-first per XCD from its own step counts (rf_launch_opts.tile_cost), but
// benchmark.py:95-139 renders another camera every frame and a training view is new every time -- rays the pipeline has
// not seen run under the static dealing (forward 5.46 against 4.93 ms on the north-star frame, render 2.91 against 2.55).
// This file estimates the step count of a tile's longest ray WITHOUT tracing anything:
//
//   rf_build_cost_grid     a coarse grid (res^3 voxels over the points' bounding box) of {lambda, sigma}: lambda = the cells
//                          a line crosses per unit length in that voxel (1.455 n^(1/3) for a Poisson-Voronoi foam of n
//                          points per unit volume -- the same constant SURVEY.md 8(d) uses for its density scale), sigma =
//                          the mean density of the voxel's cells.  One pass of atomics over the points; rebuilt when the
//                          foam is repacked after a triangulation rebuild (it need not follow every optimiser step);
//   rf_estimate_tile_cost  one thread per 16x16 tile marches five of the tile's rays (from the ray tensor, or cast from the
//                          camera) through the grid: steps += lambda dl, optical depth += sigma dl, until the transmittance
//                          falls below the weight threshold or the ray leaves the box; the tile costs its longest ray.
//
// The estimate only has to RANK tiles (the host turns it into a block -> tile table with the same rules it applies to a
// measured cost map); any order gives the same results, bit for bit.  No counterpart in the reference.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include "../../include/radfoam_hip.h"
#include "rf_host.hpp"
#include "rf_tiles.hpp"

namespace rf {

constexpr uint32_t kGridHeaderFloats = 16;   // lo[3], hi[3], inv_voxel[3], voxel_volume, res, 5 spare

// order-preserving float <-> uint32 (for atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float unordered(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// scratch[0..2] = ordered min, scratch[3..5] = ordered max (initialised by the host call)
__global__ __launch_bounds__(256) void grid_bbox_kernel(const float *__restrict__ points, uint32_t n,
                                                        uint32_t *__restrict__ scratch) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = points[3 * (size_t)i + k];
            if (v == v && fabsf(v) < INFINITY) {
                lo[k] = fminf(lo[k], v);
                hi[k] = fmaxf(hi[k], v);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
        }
    }
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            atomicMin(scratch + k, ordered(lo[k]));
            atomicMax(scratch + 3 + k, ordered(hi[k]));
        }
    }
}

// header from the box; voxels zeroed by the host call before this
__global__ void grid_header_kernel(const uint32_t *__restrict__ scratch, uint32_t res, float *__restrict__ header) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float vol = 1.0f;
    for (int k = 0; k < 3; ++k) {
        float lo = unordered(scratch[k]), hi = unordered(scratch[3 + k]);
        if (!(hi > lo)) {                    // no points, one point, or a flat cloud: a unit box around it
            const float c = (hi == hi && lo == lo && fabsf(lo) < INFINITY) ? lo : 0.0f;
            lo = c - 0.5f;
            hi = c + 0.5f;
        }
        const float pad = 1e-4f * (hi - lo);
        lo -= pad;
        hi += pad;
        header[k] = lo;
        header[3 + k] = hi;
        header[6 + k] = (float)res / (hi - lo);
        vol *= (hi - lo) / (float)res;
    }
    header[9] = vol;
    header[10] = (float)res;
}

template <bool HALF>
__global__ __launch_bounds__(256) void grid_scatter_kernel(const float *__restrict__ points, const void *__restrict__ attributes,
                                                           uint32_t attr_dim, uint32_t n, uint32_t res,
                                                           const float *__restrict__ header, float2 *__restrict__ voxels) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = points[3 * (size_t)i + k];
        if (!(v == v) || !(fabsf(v) < INFINITY)) return;
        int q = (int)floorf((v - header[k]) * header[6 + k]);
        c[k] = q < 0 ? 0 : (q >= (int)res ? (int)res - 1 : q);
    }
    float s;
    if constexpr (HALF)
        s = (float)reinterpret_cast<const _Float16 *>(attributes)[(size_t)i * attr_dim + attr_dim - 1];
    else
        s = reinterpret_cast<const float *>(attributes)[(size_t)i * attr_dim + attr_dim - 1];
    if (!(s == s) || !(fabsf(s) < INFINITY)) s = 0.0f;
    float2 *v = voxels + ((size_t)c[2] * res + (size_t)c[1]) * res + (size_t)c[0];
    unsafeAtomicAdd(&v->x, 1.0f);
    unsafeAtomicAdd(&v->y, s);
}

// {count, density sum} -> {cells crossed per unit length, mean density}
__global__ __launch_bounds__(256) void grid_finish_kernel(const float *__restrict__ header, uint32_t total,
                                                          float2 *__restrict__ voxels) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const float2 v = voxels[i];
    const float n = v.x / header[9];                        // points per unit volume
    voxels[i] = make_float2(1.455f * cbrtf(n), v.x > 0.0f ? v.y / v.x : 0.0f);
}

struct PriorView {
    const float *rays;        // [H][W][6] or nullptr
    rf_camera cam;            // used when rays == nullptr
    float inv_tan_half_fov;
    uint32_t width, height;
    float log_inv_threshold;  // -ln(weight_threshold)
    float max_steps;
};

__device__ __forceinline__ bool prior_ray(const PriorView &p, uint32_t x, uint32_t y, float (&o)[3], float (&d)[3]) {
    if (p.rays) {
        const float *r = p.rays + ((size_t)y * p.width + x) * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = r[k];
            d[k] = r[3 + k];
        }
    } else {
        // camera.h:56-85 (to the accuracy a prior needs; the render kernel casts its own rays)
        const float aspect = (float)p.cam.width / (float)p.cam.height;
        const float u = (2.0f * ((float)x / (float)p.cam.width) - 1.0f) * aspect;
        const float v = 1.0f - 2.0f * ((float)y / (float)p.cam.height);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = p.cam.position[k];
        if (p.cam.model == 0u) {
#pragma unroll
            for (int k = 0; k < 3; ++k) d[k] = p.inv_tan_half_fov * p.cam.forward[k] + u * p.cam.right[k] + v * p.cam.up[k];
        } else {
            const float theta = atan2f(v, u);
            const float phi = fminf(p.cam.fov * sqrtf(u * u + v * v), 3.14159f);
            const float a = sinf(phi) * cosf(theta), b = sinf(phi) * sinf(theta), c = cosf(phi);
#pragma unroll
            for (int k = 0; k < 3; ++k) d[k] = c * p.cam.forward[k] + a * p.cam.right[k] + b * p.cam.up[k];
        }
    }
    const float n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (!(n2 > 0.0f) || !(n2 < INFINITY)) return false;
    const float inv = 1.0f / sqrtf(n2);
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] *= inv;
    return true;
}

// estimated cells a ray visits before it is opaque or out of the box
__device__ __forceinline__ float prior_march(const float *__restrict__ header, const float2 *__restrict__ voxels, uint32_t res,
                                             const float (&o)[3], const float (&d)[3], float log_inv_thr, float max_steps) {
    float ta = 0.0f, tb = INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float inv = 1.0f / d[k];             // +-inf for an axis-parallel ray: the slab test still works
        float t0 = (header[k] - o[k]) * inv, t1 = (header[3 + k] - o[k]) * inv;
        if (t0 != t0 || t1 != t1) {                // 0 * inf: the origin sits on a slab plane of a parallel ray
            t0 = -INFINITY;
            t1 = INFINITY;
        }
        ta = fmaxf(ta, fminf(t0, t1));
        tb = fminf(tb, fmaxf(t0, t1));
    }
    // outside the box the foam has only its unbounded hull cells: a handful of steps
    float steps = 4.0f;
    if (!(tb > ta)) return steps;
    const float voxel = 1.0f / fmaxf(fmaxf(header[6], header[7]), header[8]);      // smallest voxel edge
    const float length = tb - ta;
    const int n = (int)fminf(ceilf(length / (0.5f * voxel)), 256.0f);
    const float h = length / (float)n;
    float tau = 0.0f;
    for (int i = 0; i < n; ++i) {
        const float t = ta + ((float)i + 0.5f) * h;
        int c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int q = (int)floorf((o[k] + t * d[k] - header[k]) * header[6 + k]);
            c[k] = q < 0 ? 0 : (q >= (int)res ? (int)res - 1 : q);
        }
        const float2 v = voxels[((size_t)c[2] * res + (size_t)c[1]) * res + (size_t)c[0]];
        // the segment in which the ray becomes opaque is only crossed in part
        const float dtau = v.y * h;
        if (tau + dtau > log_inv_thr) {
            steps += v.x * h * ((log_inv_thr - tau) / dtau) + 1.0f;
            break;
        }
        tau += dtau;
        steps += v.x * h;
        if (steps >= max_steps) break;
    }
    return fminf(steps, max_steps);
}

__global__ __launch_bounds__(256) void tile_cost_estimate_kernel(PriorView p, const float *__restrict__ header,
                                                                 const float2 *__restrict__ voxels, uint32_t res,
                                                                 uint32_t tiles_x, uint32_t tiles,
                                                                 uint32_t *__restrict__ tile_cost) {
    const uint32_t tile = blockIdx.x * 256u + threadIdx.x;
    if (tile >= tiles) return;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    // the centre of the tile and of its four 8x8 quadrants (a wave owns a quadrant)
    const uint32_t sx[5] = {8u, 4u, 12u, 4u, 12u}, sy[5] = {8u, 4u, 4u, 12u, 12u};
    float longest = 0.0f;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        uint32_t x = tx * 16u + sx[s], y = ty * 16u + sy[s];
        x = x < p.width ? x : p.width - 1u;
        y = y < p.height ? y : p.height - 1u;
        float o[3], d[3];
        if (!prior_ray(p, x, y, o, d)) continue;
        longest = fmaxf(longest, prior_march(header, voxels, res, o, d, p.log_inv_threshold, p.max_steps));
    }
    tile_cost[tile] = (uint32_t)(longest + 0.5f);
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile orders from a cost map, on the device.  Rounds 4-5 turned a forward's tile_cost into the block -> tile tables of the
// next launches with two dozen small torch operations (sort, gather, kthvalue, ...): 0.2-0.25 ms on the launch stream
// between the forward and whatever follows it -- as much as the backward's order then gained (profiles/r06/c_*).  One
// launch of 8 blocks does the same: block x owns XCD x's column of the static dealing (positions i = 0 .. rows-1, block
// b = 8 i + x), gives every position a 64-bit key {rule-dependent rank, position}, sorts the column (bitonic, in LDS) and
// writes the tiles in that order.
//   rule 1 "xcd:q"   longest first in classes of q steps, the static order within a class (forward / render of a frame
//                    whose own costs are known);
//   rule 2 "tail:n"  the static order, except that the n cheapest tiles of the frame come last, the longest of them first
//                    (backward; flat batches).  The threshold is the n-th smallest cost of the whole frame: every block
//                    histograms the frame for itself (costs clamp at 4095 for the histogram only).
// Tiles past the end of the frame (padding blocks of the last round) go last under both rules.
constexpr uint32_t kOrderRuleClasses = 1, kOrderRuleTail = 2;
constexpr uint32_t kHistBins = 4096;

struct OrderJob {
    uint32_t rule, param;
    uint32_t *out;
};

__global__ __launch_bounds__(256) void tile_orders_kernel(const uint32_t *__restrict__ cost, uint32_t num_rays, uint32_t width,
                                                          uint32_t height, uint32_t rows_pow2, OrderJob job0, OrderJob job1) {
    extern __shared__ unsigned long long s_keys[];                 // rows_pow2 entries
    __shared__ uint32_t s_hist[kHistBins];
    __shared__ uint32_t s_limit;
    const OrderJob job = blockIdx.y == 0 ? job0 : job1;
    const uint32_t x = blockIdx.x;
    const uint32_t nt = tile_count(num_rays, width, height);
    const uint32_t chunk = tile_chunk_of(width);
    const uint32_t nblocks = launch_block_count(num_rays, width, height);
    const uint32_t rows = nblocks >> 3, rounds = nblocks / (8u * chunk);
    uint32_t limit = 0;
    if (job.rule == kOrderRuleTail) {
        for (uint32_t i = threadIdx.x; i < kHistBins; i += 256u) s_hist[i] = 0u;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < nt; t += 256u) {
            const uint32_t c = cost[t];
            atomicAdd(&s_hist[c < kHistBins ? c : kHistBins - 1u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t count = job.param < 1u ? 1u : (job.param > nt ? nt : job.param), acc = 0, l = 0;
            for (l = 0; l < kHistBins; ++l) {
                acc += s_hist[l];
                if (acc >= count) break;
            }
            s_limit = l < kHistBins - 1u ? l : 0xFFFFFFFFu;          // the last bin holds everything above it too
        }
        __syncthreads();
        limit = s_limit;
    }
    const uint32_t q = job.param ? job.param : 1u;
    for (uint32_t i = threadIdx.x; i < rows_pow2; i += 256u) {
        unsigned long long key = ~0ull;                             // padding of the sort
        if (i < rows) {
            const uint32_t tile = dealt_tile(i * 8u + x, chunk, rounds);
            const bool valid = tile < nt;
            const uint32_t c = valid ? cost[tile] : 0u;
            uint32_t rank;
            if (job.rule == kOrderRuleTail) {
                const uint32_t big = rows + 1u;
                const uint32_t cc = c < (1u << 24) ? c : (1u << 24);
                rank = !valid ? big + (1u << 25) : (c <= limit ? big + ((1u << 24) - cc) : i);
            } else {
                rank = !valid ? 0xFFFFFFFEu : 0x7FFFFFFFu - (c / q < 0x7FFFFFFFu ? c / q : 0x7FFFFFFFu);
            }
            key = ((unsigned long long)rank << 32) | (unsigned long long)i;
        }
        s_keys[i] = key;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= rows_pow2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < rows_pow2; i += 256u) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const unsigned long long a = s_keys[i], b = s_keys[l];
                    const bool up = (i & k) == 0u;
                    if ((a > b) == up) {
                        s_keys[i] = b;
                        s_keys[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < rows; i += 256u) {
        const uint32_t pos = (uint32_t)(s_keys[i] & 0xFFFFFFFFull);
        job.out[i * 8u + x] = dealt_tile(pos * 8u + x, chunk, rounds);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// May a frame take the block order that ANOTHER frame's trace measured?  Only when it shows nearly the same picture: a
// camera path (a viewer, a fly-through) moves a pixel or two per frame, and the cost map with it; another camera of a
// data set does not, and its order is worse than the static dealing (profiles/r04/d_tile_order_*).  A reference record of
// the learnt frame -- five of its rays (the centre of the frame and of its quadrants) and, per ray, the distance from the
// origin to the point of its entry cell (the nearest thing the camera can see move) -- is compared with the same five rays
// of the new frame: the frames are coherent when, for every sample, the angle between the directions plus the shift of
// the origin relative to that distance stays below max_angle (radians).  Decided on the device: no synchronisation.
constexpr uint32_t kRefSamples = 5, kRefFloats = 8;

__device__ __forceinline__ void ref_sample_pixel(uint32_t s, uint32_t width, uint32_t height, uint32_t &x, uint32_t &y) {
    const uint32_t fx[5] = {2u, 1u, 3u, 1u, 3u}, fy[5] = {2u, 1u, 1u, 3u, 3u};
    x = (width * fx[s]) >> 2;
    y = (height * fy[s]) >> 2;
    x = x < width ? x : width - 1u;
    y = y < height ? y : height - 1u;
}

__global__ void tile_reference_kernel(PriorView v, const uint32_t *__restrict__ start, const float *__restrict__ points,
                                      float *__restrict__ ref) {
    const uint32_t s = threadIdx.x;
    if (s >= kRefSamples) return;
    uint32_t x, y;
    ref_sample_pixel(s, v.width, v.height, x, y);
    float o[3], d[3] = {0.0f, 0.0f, 0.0f};
    if (!prior_ray(v, x, y, o, d)) d[0] = d[1] = d[2] = 0.0f;
    // (a camera has ONE entry cell; a ray tensor one per ray)
    const float *pp = points + 3 * (size_t)start[v.rays ? (size_t)y * v.width + x : 0];
    const float dx = pp[0] - o[0], dy = pp[1] - o[1], dz = pp[2] - o[2];
    float *r = ref + s * kRefFloats;
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
    r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = sqrtf(dx * dx + dy * dy + dz * dz);
    r[7] = 0.0f;
}

__global__ __launch_bounds__(256) void tile_order_gate_kernel(PriorView v, const float *__restrict__ ref, float max_angle,
                                                              const uint32_t *__restrict__ learnt, uint32_t *__restrict__ out,
                                                              uint32_t *__restrict__ verdict) {
    const uint32_t nblocks = launch_block_count(v.width * v.height, v.width, v.height);
    const uint32_t chunk = tile_chunk_of(v.width);
    bool coherent = true;
#pragma unroll
    for (uint32_t s = 0; s < kRefSamples; ++s) {
        uint32_t x, y;
        ref_sample_pixel(s, v.width, v.height, x, y);
        float o[3], d[3];
        const float *r = ref + s * kRefFloats;
        if (!prior_ray(v, x, y, o, d)) {
            coherent = false;
            continue;
        }
        const float cx = d[1] * r[5] - d[2] * r[4], cy = d[2] * r[3] - d[0] * r[5], cz = d[0] * r[4] - d[1] * r[3];
        const float angle = atan2f(sqrtf(cx * cx + cy * cy + cz * cz), d[0] * r[3] + d[1] * r[4] + d[2] * r[5]);
        const float sx = o[0] - r[0], sy = o[1] - r[1], sz = o[2] - r[2];
        const float rel = sqrtf(sx * sx + sy * sy + sz * sz) / fmaxf(r[6], 1e-30f);
        if (!(angle + rel <= max_angle)) coherent = false;      // NaN anywhere: not coherent
    }
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b == 0 && verdict) *verdict = coherent ? 1u : 0u;
    if (b < nblocks) out[b] = coherent ? learnt[b] : dealt_tile(b, chunk, nblocks / (8u * chunk));
}

}  // namespace rf

using namespace rf;

extern "C" {

size_t rf_cost_grid_bytes(uint32_t res) {
    return (size_t)kGridHeaderFloats * 4 + (size_t)res * res * res * sizeof(float2) + 64;
}

int rf_build_cost_grid(const float *points, const void *attributes, int attr_type, uint32_t attr_dim, uint32_t num_points,
                       uint32_t res, void *grid, size_t grid_bytes, void *stream) {
    g_err[0] = 0;
    if (!points || !attributes || !grid) return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_cost_grid: null pointer");
    if (res < 2 || res > 256) return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_cost_grid: res must be in [2, 256]");
    if (attr_dim == 0) return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_cost_grid: attr_dim must be positive");
    if (attr_type != RF_ATTR_FLOAT32 && attr_type != RF_ATTR_FLOAT16)
        return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported attribute type");
    if (grid_bytes < rf_cost_grid_bytes(res)) return fail(RF_ERR_WORKSPACE, "rf_build_cost_grid: grid buffer too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *header = static_cast<float *>(grid);
    float2 *voxels = reinterpret_cast<float2 *>(header + kGridHeaderFloats);
    const uint32_t total = res * res * res;
    uint32_t *scratch = reinterpret_cast<uint32_t *>(voxels + total);     // the 64 bytes behind the voxels
    hipError_t e = hipMemsetAsync(voxels, 0, (size_t)total * sizeof(float2), s);
    if (e == hipSuccess) e = hipMemsetAsync(scratch, 0xFF, 12, s);          // ordered min: start at the top
    if (e == hipSuccess) e = hipMemsetAsync(scratch + 3, 0x00, 12, s);      // ordered max: start at the bottom
    if (e != hipSuccess) return fail(RF_ERR_LAUNCH, "rf_build_cost_grid: %s", hipGetErrorString(e));
    if (num_points) {
        const unsigned blocks = (unsigned)((num_points + 255u) / 256u);
        hipLaunchKernelGGL(grid_bbox_kernel, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, s, points, num_points, scratch);
    }
    hipLaunchKernelGGL(grid_header_kernel, dim3(1), dim3(64), 0, s, scratch, res, header);
    if (num_points) {
        const dim3 g((unsigned)((num_points + 255u) / 256u)), b(256);
        if (attr_type == RF_ATTR_FLOAT16)
            hipLaunchKernelGGL(grid_scatter_kernel<true>, g, b, 0, s, points, attributes, attr_dim, num_points, res, header, voxels);
        else
            hipLaunchKernelGGL(grid_scatter_kernel<false>, g, b, 0, s, points, attributes, attr_dim, num_points, res, header, voxels);
    }
    hipLaunchKernelGGL(grid_finish_kernel, dim3((total + 255u) / 256u), dim3(256), 0, s, header, total, voxels);
    return check_launch("rf_build_cost_grid");
}

int rf_estimate_tile_cost(const void *grid, uint32_t res, const float *rays, const rf_camera *camera, uint32_t width,
                          uint32_t height, float weight_threshold, uint32_t max_intersections, uint32_t *tile_cost,
                          void *stream) {
    g_err[0] = 0;
    if (!grid || !tile_cost) return fail(RF_ERR_INVALID_ARGUMENT, "rf_estimate_tile_cost: null pointer");
    if ((rays == nullptr) == (camera == nullptr))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_estimate_tile_cost: give the rays or the camera, not both");
    if (res < 2 || res > 256) return fail(RF_ERR_INVALID_ARGUMENT, "rf_estimate_tile_cost: res must be in [2, 256]");
    if (camera) {
        width = camera->width;
        height = camera->height;
    }
    if (width == 0 || height == 0) return RF_OK;
    PriorView p{};
    p.rays = rays;
    if (camera) {
        p.cam = *camera;
        p.inv_tan_half_fov = 1.0f / tanf(0.5f * camera->fov);
    }
    p.width = width;
    p.height = height;
    const float thr = weight_threshold > 0.0f && weight_threshold < 1.0f ? weight_threshold : 1e-30f;
    p.log_inv_threshold = -logf(thr);
    p.max_steps = (float)(max_intersections < 65535u ? max_intersections : 65535u);
    const uint32_t tiles_x = (width + 15u) >> 4, tiles = tiles_x * ((height + 15u) >> 4);
    const float *header = static_cast<const float *>(grid);
    const float2 *voxels = reinterpret_cast<const float2 *>(header + kGridHeaderFloats);
    hipLaunchKernelGGL(tile_cost_estimate_kernel, dim3((tiles + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream), p,
                       header, voxels, res, tiles_x, tiles, tile_cost);
    return check_launch("rf_estimate_tile_cost");
}

static uint32_t order_rule_id(const char *who, uint32_t rule) {
    (void)who;
    return rule == kOrderRuleClasses || rule == kOrderRuleTail ? rule : 0u;
}

int rf_build_tile_orders(const uint32_t *tile_cost, uint32_t num_rays, uint32_t image_width, uint32_t image_height,
                         uint32_t rule_a, uint32_t param_a, uint32_t *order_a, uint32_t rule_b, uint32_t param_b,
                         uint32_t *order_b, void *stream) {
    g_err[0] = 0;
    if (!tile_cost || !order_a) return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_tile_orders: null pointer");
    if (!order_rule_id("a", rule_a) || (order_b && !order_rule_id("b", rule_b)))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_tile_orders: rule must be 1 (classes) or 2 (tail)");
    if (image_width && (uint64_t)image_width * image_height != num_rays)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_tile_orders: image_width * image_height must equal num_rays");
    const uint32_t nblocks = launch_block_count(num_rays, image_width, image_height);
    if (nblocks == 0) return RF_OK;
    const uint32_t rows = nblocks >> 3;
    uint32_t pow2 = 1;
    while (pow2 < rows) pow2 <<= 1;
    if (pow2 > 8192u)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_tile_orders: more than 65536 blocks (build the order on the host)");
    OrderJob a{rule_a, param_a, order_a}, b{rule_b, param_b, order_b};
    hipLaunchKernelGGL(tile_orders_kernel, dim3(8, order_b ? 2 : 1), dim3(256), (size_t)pow2 * sizeof(unsigned long long),
                       static_cast<hipStream_t>(stream), tile_cost, num_rays, image_width, image_height, pow2, a, b);
    return check_launch("rf_build_tile_orders");
}

static int prior_view(const char *who, const float *rays, const rf_camera *camera, uint32_t width, uint32_t height,
                      PriorView &p) {
    if ((rays == nullptr) == (camera == nullptr)) return fail(RF_ERR_INVALID_ARGUMENT, "%s: give the rays or the camera, not both", who);
    p = PriorView{};
    p.rays = rays;
    if (camera) {
        p.cam = *camera;
        p.inv_tan_half_fov = 1.0f / tanf(0.5f * camera->fov);
        width = camera->width;
        height = camera->height;
    }
    if (width == 0 || height == 0) return fail(RF_ERR_INVALID_ARGUMENT, "%s: empty frame", who);
    p.width = width;
    p.height = height;
    return RF_OK;
}

int rf_tile_order_reference(const float *rays, const rf_camera *camera, const uint32_t *start_point_index, const float *points,
                            uint32_t image_width, uint32_t image_height, float *reference, void *stream) {
    g_err[0] = 0;
    if (!start_point_index || !points || !reference) return fail(RF_ERR_INVALID_ARGUMENT, "rf_tile_order_reference: null pointer");
    PriorView v;
    const int rc = prior_view("rf_tile_order_reference", rays, camera, image_width, image_height, v);
    if (rc != RF_OK) return rc;
    hipLaunchKernelGGL(tile_reference_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), v, start_point_index,
                       points, reference);
    return check_launch("rf_tile_order_reference");
}

int rf_gate_tile_order(const float *rays, const rf_camera *camera, const float *reference, uint32_t image_width,
                       uint32_t image_height, float max_angle_radians, const uint32_t *learnt_order, uint32_t *order,
                       uint32_t *verdict, void *stream) {
    g_err[0] = 0;
    if (!reference || !learnt_order || !order) return fail(RF_ERR_INVALID_ARGUMENT, "rf_gate_tile_order: null pointer");
    PriorView v;
    const int rc = prior_view("rf_gate_tile_order", rays, camera, image_width, image_height, v);
    if (rc != RF_OK) return rc;
    const uint32_t nblocks = launch_block_count(v.width * v.height, v.width, v.height);
    hipLaunchKernelGGL(tile_order_gate_kernel, dim3((nblocks + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream), v,
                       reference, max_angle_radians, learnt_order, order, verdict);
    return check_launch("rf_gate_tile_order");
}

}  // extern "C"
