// hip_pipeline.cpp -- Seam B for real: a complete radfoam::Pipeline subclass over the C-ABI of
// include/radfoam_hip.h, compiled against the REFERENCE's own src/tracing/pipeline.h (read where it lies
// under /root/reference; oracle/Makefile.ref, target _ref/libhip_pipeline.so).  This is the file a maintainer
// of the reference would add as src/tracing/hip_pipeline.cpp and build instead of pipeline.cu on ROCm: it
// defines radfoam::create_pipeline() (pipeline.h:133, reference definition pipeline.cu:776-805) and every
// virtual of Pipeline (pipeline.h:58-131).  TEST INFRASTRUCTURE in this repository: tests/test_ref_binding.py
// drives it through the reference's virtual interface and compares with the ctypes path.
//
// It is the plain binding: like CUDATracingPipeline (pipeline.cu:588-774) it keeps no state between calls --
// the packed foam is rebuilt inside every call (the reference rebuilds its half4 table in every call too,
// pipeline.cu:613-620,667-674) and trace_backward re-walks.  INTEGRATION.md lists the rf_launch_opts a
// stateful binding would keep between the two calls of a step (foam_prepared, the hop trail, ray_order).
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "pipeline.h"        // the reference's: -I/root/reference/src/tracing
#include "hip_pipeline.h"
#include "radfoam_hip.h"     // this repository: include/radfoam_hip.h

namespace radfoam {

namespace {

DeviceMemoryHooks g_mem{nullptr, nullptr, nullptr, nullptr};

void check(int rc) {
    // every failure of the reference surfaces as std::runtime_error (cuda_helpers.h:12-19,
    // pipeline_bindings.cpp:14-70); pybind turns it into Python's RuntimeError
    if (rc != RF_OK) throw std::runtime_error(rf_last_error());
}

// device scratch with the lifetime of one call (CUDAArray<Vec4h> adjacent_diff(...) in the reference)
struct Scratch {
    void *ptr = nullptr;
    explicit Scratch(size_t bytes) {
        if (!g_mem.alloc) throw std::runtime_error("set_hip_pipeline_memory() was not called");
        ptr = bytes ? g_mem.alloc(bytes) : nullptr;
        if (bytes && !ptr) throw std::runtime_error("device allocation failed");
    }
    ~Scratch() {
        if (ptr) g_mem.free(ptr);
    }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
};

// The C-ABI accumulates contribution / attribute_grad / point_error in fp32 for both attribute types; the
// reference's buffers have the attribute type.  fp32 pipelines accumulate straight into the caller's
// buffer; fp16 ones into fp32 scratch that is rounded into the caller's buffer once (rf_cast_accumulator).
struct Accumulator {
    std::unique_ptr<Scratch> scratch;
    void *caller;
    size_t count;
    bool half;
    Accumulator(void *caller_buffer, size_t n, bool is_half) : caller(caller_buffer), count(n), half(is_half) {
        if (caller && half) {
            scratch.reset(new Scratch(n * sizeof(float)));
            g_mem.zero(scratch->ptr, n * sizeof(float), g_mem.stream);
        }
    }
    void *device() const { return !caller ? nullptr : (half ? scratch->ptr : caller); }
    void finish() const {
        if (caller && half)
            check(rf_cast_accumulator(static_cast<const float *>(scratch->ptr), count, RF_ATTR_FLOAT16, caller,
                                      g_mem.stream));
    }
};

class HIPTracingPipeline : public Pipeline {
    int sh_degree_;
    rf_attr_type attr_;

    bool half() const { return attr_ == RF_ATTR_FLOAT16; }

    rf_launch_opts options(const Scratch &ws, size_t bytes) const {
        rf_launch_opts o;
        std::memset(&o, 0, sizeof(o));
        o.workspace = ws.ptr;
        o.workspace_bytes = bytes;
        return o;   // foam_prepared = 0: packed inside the call; no trail; rays as a flat list
    }

  public:
    HIPTracingPipeline(int sh_degree, rf_attr_type attr) : sh_degree_(sh_degree), attr_(attr) {}

    void trace_forward(const TraceSettings &settings, uint32_t num_points, const Vec3f *points,
                       const void *attributes, uint32_t point_adjacency_size, const uint32_t *point_adjacency,
                       const uint32_t *point_adjacency_offsets, uint32_t num_rays, const Ray *rays,
                       const uint32_t *start_point_index, uint32_t num_depth_quantiles,
                       const float *depth_quantiles, void *ray_rgba, float *quantile_dpeths,
                       uint32_t *quantile_point_indices, uint32_t *num_intersections,
                       void *point_contribution) override {
        const size_t bytes = rf_workspace_bytes(num_points, point_adjacency_size, sh_degree_, attr_);
        Scratch ws(bytes);
        rf_launch_opts opts = options(ws, bytes);
        rf_trace_settings s{settings.weight_threshold, settings.max_intersections};
        Accumulator contribution(point_contribution, num_points, half());
        check(rf_trace_forward(sh_degree_, attr_, &s, num_points, reinterpret_cast<const float *>(points),
                               attributes, point_adjacency_size, point_adjacency, point_adjacency_offsets,
                               num_rays, reinterpret_cast<const float *>(rays), start_point_index,
                               num_depth_quantiles, depth_quantiles, ray_rgba, quantile_dpeths,
                               quantile_point_indices, num_intersections, contribution.device(), &opts,
                               g_mem.stream));
        contribution.finish();
    }

    void trace_backward(const TraceSettings &settings, uint32_t num_points, const Vec3f *points,
                        const void *attributes, uint32_t point_adjacency_size, const uint32_t *point_adjacency,
                        const uint32_t *point_adjacency_offsets, uint32_t num_rays, const Ray *rays,
                        const uint32_t *start_point_index, uint32_t num_depth_quantiles,
                        const float *depth_quantiles, const uint32_t *quantile_point_indices,
                        const void *ray_rgba, const void *ray_rgba_grad, const float *depth_grad,
                        const void *ray_error, Ray *ray_grad, Vec3f *points_grad, void *attribute_grad,
                        void *point_error) override {
        const size_t bytes = rf_workspace_bytes(num_points, point_adjacency_size, sh_degree_, attr_);
        Scratch ws(bytes);
        rf_launch_opts opts = options(ws, bytes);
        rf_trace_settings s{settings.weight_threshold, settings.max_intersections};
        Accumulator attr_grad(attribute_grad, (size_t)num_points * rf_attribute_dim(sh_degree_), half());
        Accumulator error(point_error, num_points, half());
        check(rf_trace_backward(sh_degree_, attr_, &s, num_points, reinterpret_cast<const float *>(points),
                                attributes, point_adjacency_size, point_adjacency, point_adjacency_offsets,
                                num_rays, reinterpret_cast<const float *>(rays), start_point_index,
                                num_depth_quantiles, depth_quantiles, quantile_point_indices, ray_rgba,
                                ray_rgba_grad, depth_grad, ray_error, reinterpret_cast<float *>(ray_grad),
                                reinterpret_cast<float *>(points_grad), attr_grad.device(), error.device(), &opts,
                                g_mem.stream));
        attr_grad.finish();
        error.finish();
    }

    void trace_visualization(const TraceSettings &, const VisualizationSettings &, const Camera &, CMapTable,
                             uint32_t, uint32_t, const void *, const void *, const void *, const void *,
                             const void *, uint32_t, uint64_t, const void *) override {
        throw std::runtime_error("trace_visualization is not supported on this platform (no viewer)");
    }

    void trace_benchmark(const TraceSettings &settings, uint32_t num_points, const Vec3f *points,
                         const void *attributes, const uint32_t *point_adjacency,
                         const uint32_t *point_adjacency_offsets, const Vec4h *adjacent_diff, Camera camera,
                         const uint32_t *start_point_index, uint32_t *ray_rgba) override {
        // the interface passes no adjacency size (pipeline.h:117-126): the library reads offsets[num_points]
        uint32_t adjacency_size = 0;
        check(rf_adjacency_size(num_points, point_adjacency_offsets, &adjacency_size, g_mem.stream));
        const size_t bytes = rf_workspace_bytes(num_points, adjacency_size, sh_degree_, attr_);
        Scratch ws(bytes);
        rf_launch_opts opts = options(ws, bytes);
        rf_trace_settings s{settings.weight_threshold, settings.max_intersections};
        rf_camera cam;
        for (int k = 0; k < 3; ++k) {
            cam.position[k] = camera.position.data[k];
            cam.forward[k] = camera.forward.data[k];
            cam.right[k] = camera.right.data[k];
            cam.up[k] = camera.up.data[k];
        }
        cam.fov = camera.fov;
        cam.width = camera.width;
        cam.height = camera.height;
        cam.model = camera.model == Fisheye ? 1u : 0u;
        check(rf_trace_benchmark(sh_degree_, attr_, &s, num_points, reinterpret_cast<const float *>(points),
                                 attributes, adjacency_size, point_adjacency, point_adjacency_offsets,
                                 adjacent_diff, &cam, start_point_index, ray_rgba, &opts, g_mem.stream));
    }

    uint32_t attribute_dim() const override { return rf_attribute_dim(sh_degree_); }

    ScalarType attribute_type() const override { return half() ? Float16 : Float32; }
};

}  // namespace

void set_hip_pipeline_memory(const DeviceMemoryHooks &hooks) { g_mem = hooks; }

DeviceMemoryHooks triangulation_memory() { return g_mem; }   // hip_triangulation.cpp allocates through the same hooks

// prefetch_adjacent_diff, pipeline.h:46-53 (reference definition pipeline.cu:570-586)
void prefetch_adjacent_diff(const Vec3f *points, uint32_t num_points, uint32_t point_adjacency_size,
                            const uint32_t *point_adjacency, const uint32_t *point_adjacency_offsets,
                            Vec4h *adjacent_diff, const void *stream) {
    check(rf_build_adjacent_diff(reinterpret_cast<const float *>(points), num_points, point_adjacency_size,
                                 point_adjacency, point_adjacency_offsets, adjacent_diff,
                                 const_cast<void *>(stream)));
}

// create_pipeline, pipeline.h:133 (reference definition pipeline.cu:776-805: same errors)
std::shared_ptr<Pipeline> create_pipeline(int sh_degree, ScalarType attr_type) {
    if (attr_type != Float32 && attr_type != Float16) throw std::runtime_error("Unsupported attribute type");
    if (sh_degree < 0 || sh_degree > 3) throw std::runtime_error("Unsupported SH degree");
    return std::make_shared<HIPTracingPipeline>(sh_degree, attr_type == Float16 ? RF_ATTR_FLOAT16 : RF_ATTR_FLOAT32);
}

}  // namespace radfoam
