// ref_driver.cpp -- runs the REFERENCE's own tracer kernels on the CPU.  TEST INFRASTRUCTURE ONLY.
//
// Nothing of the reference is copied into this repository: this file #includes the reference's
// headers where they lie (/root/reference/src/tracing: tracing_utils.cuh, sh_utils.cuh, camera.h,
// pipeline.h) and the kernel definitions of pipeline.cu, which oracle/Makefile.ref extracts by line
// range into the git-ignored oracle/_ref/ at build time (forward :14-343 incl. backward,
// benchmark + prefetch_adjacent_diff_kernel :472-568; the visualization kernel needs CUDA
// surfaces and is left out).  CUDA and Eigen are replaced by the stand-ins in oracle/ref_shim/.
//
// What differs from the real CUDA build, and therefore what this can and cannot pin:
//   * the shim does every vector op as plain scalar C, one rounding per operation, and the build
//     uses -ffp-contract=off: no FMA contraction at all (nvcc contracts at its discretion);
//   * expf/logf are glibc's (or the oracle's with RFREF_CANONICAL_LIBM), not CUDA's;
//   * atomics are OpenMP atomics; threads run in a loop.
// So control flow, formulas, index arithmetic, quirks and data layout are the reference's,
// bit for bit its source; floating-point results agree with any other faithful evaluation to a
// few ulp per operation.  tests/test_reference_source.py compares the oracle against it with
// exactly that tolerance and uses it to generate tests/golden/.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "cuda_runtime.h"

thread_local rfref_uint3 threadIdx, blockIdx, blockDim, gridDim;

#include "../utils/geometry.h"   // resolved through -I/root/reference/src/tracing
#include "pipeline.h"
#include "sh_utils.cuh"
#include "tracing_utils.cuh"

namespace radfoam {
#include "pipeline_kernels.inc"
}  // namespace radfoam

using namespace radfoam;

namespace {

template <typename F>
void launch_1d(uint32_t n, uint32_t block, F &&body) {
    const int64_t blocks = ((int64_t)n + block - 1) / block;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t b = 0; b < blocks; ++b) {
        blockDim = {block, 1, 1};
        gridDim = {(unsigned)blocks, 1, 1};
        blockIdx = {(unsigned)b, 0, 0};
        for (uint32_t t = 0; t < block; ++t) {
            threadIdx = {t, 0, 0};
            body();
        }
    }
}

template <typename S, int D>
void run_forward(const TraceSettings &s, const Vec3f *points, const void *attr, const uint32_t *adj,
                 const uint32_t *off, const Vec4h *diff, const Ray *rays, uint32_t R, const uint32_t *start,
                 uint32_t nq, const float *q, void *rgba, float *qd, uint32_t *qi, uint32_t *nint, void *contrib) {
    launch_1d(R, 128, [&] {
        forward<S, D, 128>(s, points, (const S *)attr, adj, off, diff, rays, R, start, nq, q, (S *)rgba, qd, qi,
                           nint, (S *)contrib);
    });
}

template <typename S, int D>
void run_backward(const TraceSettings &s, const Vec3f *points, const void *attr, const uint32_t *adj,
                  const uint32_t *off, const Vec4h *diff, const Ray *rays, uint32_t R, const uint32_t *start,
                  uint32_t nq, const float *q, const uint32_t *qi, const void *rgba, const void *g,
                  const float *dg, const void *err, Vec3f *pg, void *ag, void *pe) {
    launch_1d(R, 128, [&] {
        backward<S, D, 128>(s, points, (const S *)attr, adj, off, diff, rays, R, start, nq, q, qi,
                            (const S *)rgba, (const S *)g, dg, (const S *)err, nullptr, pg, (S *)ag, (S *)pe);
    });
}

template <typename S, int D>
void run_benchmark(const TraceSettings &s, const Vec3f *points, const void *attr, const uint32_t *adj,
                   const uint32_t *off, const Vec4h *diff, const Camera &cam, const uint32_t *start,
                   uint32_t *out) {
    launch_1d(cam.width * cam.height, 512, [&] {
        benchmark<S, D, 512>(s, points, (const S *)attr, adj, off, diff, cam, start, out);
    });
}

#define RFREF_DISPATCH(FN, half, deg, ...)                                            \
    do {                                                                              \
        if (half) {                                                                   \
            switch (deg) {                                                            \
            case 0: FN<__half, 0>(__VA_ARGS__); break;                                \
            case 1: FN<__half, 1>(__VA_ARGS__); break;                                \
            case 2: FN<__half, 2>(__VA_ARGS__); break;                                \
            default: FN<__half, 3>(__VA_ARGS__); break;                               \
            }                                                                         \
        } else {                                                                      \
            switch (deg) {                                                            \
            case 0: FN<float, 0>(__VA_ARGS__); break;                                 \
            case 1: FN<float, 1>(__VA_ARGS__); break;                                 \
            case 2: FN<float, 2>(__VA_ARGS__); break;                                 \
            default: FN<float, 3>(__VA_ARGS__); break;                                \
            }                                                                         \
        }                                                                             \
    } while (0)

}  // namespace

extern "C" {

// prefetch_adjacent_diff (pipeline.cu:546-586); diff must hold adj_size + 32 entries of 8 bytes
void rfref_prefetch_adjacent_diff(const float *points, uint32_t num_points, uint32_t adj_size,
                                  const uint32_t *adj, const uint32_t *off, void *diff) {
    launch_1d(num_points, 256, [&] {
        prefetch_adjacent_diff_kernel((const Vec3f *)points, num_points, adj_size, adj, off, (Vec4h *)diff);
    });
}

// CUDATracingPipeline::trace_forward (pipeline.cu:595-642), including its per-call table build
void rfref_trace_forward(int sh_degree, int attr_half, float weight_threshold, uint32_t max_intersections,
                         uint32_t num_points, const float *points, const void *attributes, uint32_t adj_size,
                         const uint32_t *adj, const uint32_t *off, uint32_t num_rays, const float *rays,
                         const uint32_t *start, uint32_t nq, const float *quantiles, void *rgba, float *qdepth,
                         uint32_t *qidx, uint32_t *nint, void *contribution) {
    std::vector<uint64_t> diff((size_t)adj_size + 32, 0);
    rfref_prefetch_adjacent_diff(points, num_points, adj_size, adj, off, diff.data());
    TraceSettings s{weight_threshold, max_intersections};
    RFREF_DISPATCH(run_forward, attr_half, sh_degree, s, (const Vec3f *)points, attributes, adj, off,
                   (const Vec4h *)diff.data(), (const Ray *)rays, num_rays, start, nq, quantiles, rgba, qdepth,
                   qidx, nint, contribution);
}

// CUDATracingPipeline::trace_backward (pipeline.cu:644-700); outputs must be zero-filled
void rfref_trace_backward(int sh_degree, int attr_half, float weight_threshold, uint32_t max_intersections,
                          uint32_t num_points, const float *points, const void *attributes, uint32_t adj_size,
                          const uint32_t *adj, const uint32_t *off, uint32_t num_rays, const float *rays,
                          const uint32_t *start, uint32_t nq, const float *quantiles, const uint32_t *qidx,
                          const void *rgba, const void *rgba_grad, const float *depth_grad, const void *ray_error,
                          float *points_grad, void *attr_grad, void *point_error) {
    std::vector<uint64_t> diff((size_t)adj_size + 32, 0);
    rfref_prefetch_adjacent_diff(points, num_points, adj_size, adj, off, diff.data());
    TraceSettings s{weight_threshold, max_intersections};
    RFREF_DISPATCH(run_backward, attr_half, sh_degree, s, (const Vec3f *)points, attributes, adj, off,
                   (const Vec4h *)diff.data(), (const Ray *)rays, num_rays, start, nq, quantiles, qidx, rgba,
                   rgba_grad, depth_grad, ray_error, (Vec3f *)points_grad, attr_grad, point_error);
}

// CUDATracingPipeline::trace_benchmark (pipeline.cu:738-765); diff is the caller's table, padded
// here because the reference kernel reads up to 3 entries past it
void rfref_trace_benchmark(int sh_degree, int attr_half, float weight_threshold, uint32_t max_intersections,
                           uint32_t num_points, const float *points, const void *attributes, uint32_t adj_size,
                           const uint32_t *adj, const uint32_t *off, const void *diff_in, const float *cam_pos,
                           const float *cam_fwd, const float *cam_right, const float *cam_up, float fov,
                           uint32_t width, uint32_t height, int fisheye, uint32_t start_point, uint32_t *out) {
    (void)num_points;
    std::vector<uint64_t> diff((size_t)adj_size + 32, 0);
    std::memcpy(diff.data(), diff_in, (size_t)adj_size * 8);
    Camera cam;
    cam.position = Vec3f(cam_pos[0], cam_pos[1], cam_pos[2]);
    cam.forward = Vec3f(cam_fwd[0], cam_fwd[1], cam_fwd[2]);
    cam.right = Vec3f(cam_right[0], cam_right[1], cam_right[2]);
    cam.up = Vec3f(cam_up[0], cam_up[1], cam_up[2]);
    cam.fov = fov;
    cam.width = width;
    cam.height = height;
    cam.model = fisheye ? Fisheye : Pinhole;
    TraceSettings s{weight_threshold, max_intersections};
    RFREF_DISPATCH(run_benchmark, attr_half, sh_degree, s, (const Vec3f *)points, attributes, adj, off,
                   (const Vec4h *)diff.data(), cam, &start_point, out);
}

}  // extern "C"
