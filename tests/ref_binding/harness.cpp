// harness.cpp -- C entry points for tests/test_ref_binding.py: every call goes through the REFERENCE's virtual
// interface (radfoam::Pipeline, /root/reference/src/tracing/pipeline.h:58-131) on an object made by the
// reference's factory signature radfoam::create_pipeline (implemented by hip_pipeline.cpp).  Pointers are raw
// device pointers (torch tensor.data_ptr()).  TEST INFRASTRUCTURE ONLY.
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>

#include "pipeline.h"
#include "hip_pipeline.h"

extern "C" {
// dev_mem.cpp (HIP runtime; kept out of this translation unit, whose CUDA vocabulary is the CPU shim's)
void *rb_dev_alloc(size_t bytes);
void rb_dev_free(void *ptr);
void rb_dev_zero(void *ptr, size_t bytes, void *stream);
}

using namespace radfoam;

namespace {
thread_local char g_error[512] = "";
struct Handle {
    std::shared_ptr<Pipeline> pipeline;
};
template <typename F>
int guarded(F &&f) {
    g_error[0] = 0;
    try {
        f();
        return 0;
    } catch (const std::exception &e) {
        std::snprintf(g_error, sizeof(g_error), "%s", e.what());
        return -1;
    }
}
}  // namespace

extern "C" {

const char *rb_last_error() { return g_error; }

void *rb_create(int sh_degree, int attr_scalar_type, void *stream) {
    Handle *h = nullptr;
    guarded([&] {
        set_hip_pipeline_memory(DeviceMemoryHooks{rb_dev_alloc, rb_dev_free, rb_dev_zero, stream});
        std::shared_ptr<Pipeline> p = create_pipeline(sh_degree, static_cast<ScalarType>(attr_scalar_type));
        h = new Handle{p};
    });
    return h;
}

void rb_destroy(void *handle) { delete static_cast<Handle *>(handle); }

uint32_t rb_attribute_dim(void *handle) { return static_cast<Handle *>(handle)->pipeline->attribute_dim(); }
int rb_attribute_type(void *handle) { return (int)static_cast<Handle *>(handle)->pipeline->attribute_type(); }
int rb_scalar_float16() { return (int)Float16; }
int rb_scalar_float32() { return (int)Float32; }
int rb_scalar_float64() { return (int)Float64; }

int rb_trace_forward(void *handle, float weight_threshold, uint32_t max_intersections, uint32_t num_points,
                     const void *points, const void *attributes, uint32_t adjacency_size, const uint32_t *adjacency,
                     const uint32_t *offsets, uint32_t num_rays, const void *rays, const uint32_t *start,
                     uint32_t num_quantiles, const float *quantiles, void *rgba, float *quantile_depths,
                     uint32_t *quantile_indices, uint32_t *num_intersections, void *contribution) {
    return guarded([&] {
        TraceSettings s = default_trace_settings();
        s.weight_threshold = weight_threshold;
        s.max_intersections = max_intersections;
        static_cast<Handle *>(handle)->pipeline->trace_forward(
            s, num_points, static_cast<const Vec3f *>(points), attributes, adjacency_size, adjacency, offsets, num_rays,
            static_cast<const Ray *>(rays), start, num_quantiles, quantiles, rgba, quantile_depths, quantile_indices,
            num_intersections, contribution);
    });
}

int rb_trace_backward(void *handle, float weight_threshold, uint32_t max_intersections, uint32_t num_points,
                      const void *points, const void *attributes, uint32_t adjacency_size, const uint32_t *adjacency,
                      const uint32_t *offsets, uint32_t num_rays, const void *rays, const uint32_t *start,
                      uint32_t num_quantiles, const float *quantiles, const uint32_t *quantile_indices,
                      const void *rgba, const void *rgba_grad, const float *depth_grad, const void *ray_error,
                      void *ray_grad, void *points_grad, void *attribute_grad, void *point_error) {
    return guarded([&] {
        TraceSettings s = default_trace_settings();
        s.weight_threshold = weight_threshold;
        s.max_intersections = max_intersections;
        static_cast<Handle *>(handle)->pipeline->trace_backward(
            s, num_points, static_cast<const Vec3f *>(points), attributes, adjacency_size, adjacency, offsets, num_rays,
            static_cast<const Ray *>(rays), start, num_quantiles, quantiles, quantile_indices, rgba, rgba_grad,
            depth_grad, ray_error, static_cast<Ray *>(ray_grad), static_cast<Vec3f *>(points_grad), attribute_grad,
            point_error);
    });
}

int rb_trace_benchmark(void *handle, float weight_threshold, uint32_t max_intersections, uint32_t num_points,
                       const void *points, const void *attributes, const uint32_t *adjacency, const uint32_t *offsets,
                       const void *adjacent_diff, const float *position, const float *forward, const float *right,
                       const float *up, float fov, uint32_t width, uint32_t height, int fisheye,
                       const uint32_t *start, uint32_t *out_rgba8) {
    return guarded([&] {
        TraceSettings s = default_trace_settings();
        s.weight_threshold = weight_threshold;
        s.max_intersections = max_intersections;
        Camera cam;
        cam.position = Vec3f(position[0], position[1], position[2]);
        cam.forward = Vec3f(forward[0], forward[1], forward[2]);
        cam.right = Vec3f(right[0], right[1], right[2]);
        cam.up = Vec3f(up[0], up[1], up[2]);
        cam.fov = fov;
        cam.width = width;
        cam.height = height;
        cam.model = fisheye ? Fisheye : Pinhole;
        static_cast<Handle *>(handle)->pipeline->trace_benchmark(
            s, num_points, static_cast<const Vec3f *>(points), attributes, adjacency, offsets,
            static_cast<const Vec4h *>(adjacent_diff), cam, start, out_rgba8);
    });
}

int rb_trace_visualization_is_rejected(void *handle) {
    int rc = guarded([&] {
        static_cast<Handle *>(handle)->pipeline->trace_visualization(
            default_trace_settings(), default_visualization_settings(), Camera{}, CMapTable{}, 0, 0, nullptr, nullptr,
            nullptr, nullptr, nullptr, 0, 0, nullptr);
    });
    return rc == -1 ? 1 : 0;
}

int rb_prefetch_adjacent_diff(const void *points, uint32_t num_points, uint32_t adjacency_size,
                              const uint32_t *adjacency, const uint32_t *offsets, void *diff, void *stream) {
    return guarded([&] {
        prefetch_adjacent_diff(static_cast<const Vec3f *>(points), num_points, adjacency_size, adjacency, offsets,
                               static_cast<Vec4h *>(diff), stream);
    });
}

}  // extern "C"
