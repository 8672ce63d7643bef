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

}  // extern "C"
