// hip_triangulation.cpp -- Seam B for the triangulation: a radfoam::Triangulation subclass over the C-ABI of
// include/radfoam_hip.h (rf_kd_order, rf_build_aabb_tree, rf_delaunay_adjacency), compiled against the REFERENCE's
// own src/delaunay/delaunay.h (read where it lies under /root/reference; oracle/Makefile.ref, target
// _ref/libhip_pipeline.so).  This is the file a maintainer of the reference would add as
// src/delaunay/hip_delaunay.cpp and build instead of delaunay.cu on ROCm: it defines
// Triangulation::create_triangulation (delaunay.h:41-42, reference definition delaunay.cu:390-395) and the virtuals
// the training loop reads (permutation, point_adjacency(+_size, _offsets), rebuild: triangulation_bindings.cpp:20-115,
// radfoam_model/scene.py:160-200).  TEST INFRASTRUCTURE in this repository.
//
// The tetrahedra (tets / tet_adjacency / vert_to_tet: the viewer's getters) are not produced by the star
// triangulation; this binding reports none.
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>

#include "delaunay.h"        // the reference's: -I/root/reference/src/delaunay
#include "hip_pipeline.h"    // DeviceMemoryHooks
#include "radfoam_hip.h"     // this repository: include/radfoam_hip.h

namespace radfoam {

DeviceMemoryHooks triangulation_memory();   // hip_pipeline.cpp: the hooks set_hip_pipeline_memory() stored

namespace {

struct Buffer {   // CUDAArray<T> of the reference (src/utils/cuda_array.h): grows, never shrinks
    void *ptr = nullptr;
    size_t bytes = 0;
    DeviceMemoryHooks mem;
    explicit Buffer(const DeviceMemoryHooks &m) : mem(m) {}
    ~Buffer() {
        if (ptr) mem.free(ptr);
    }
    void expand(size_t want) {
        if (want <= bytes) return;
        if (ptr) mem.free(ptr);
        ptr = mem.alloc(want);
        if (!ptr) throw std::runtime_error("device allocation failed");
        bytes = want;
    }
    Buffer(const Buffer &) = delete;
    Buffer &operator=(const Buffer &) = delete;
};

class HIPTriangulation : public Triangulation {
  public:
    HIPTriangulation(const void *points, uint32_t num_points)
        : mem(triangulation_memory()), sorted(mem), perm(mem), tree(mem), adjacency(mem), next_adjacency(mem),
          offsets(mem), next_offsets(mem), workspace(mem) {
        if (!mem.alloc) throw std::runtime_error("set_hip_pipeline_memory() was not called");
        rebuild(points, num_points, false);
    }

    const uint32_t *permutation() const override { return static_cast<const uint32_t *>(perm.ptr); }
    uint32_t num_points() const override { return n; }
    const IndexedTet *tets() const override { return nullptr; }
    uint32_t num_tets() const override { return 0; }
    uint32_t num_faces() const override { return 0; }
    const uint32_t *tet_adjacency() const override { return nullptr; }
    const uint32_t *point_adjacency() const override { return static_cast<const uint32_t *>(adjacency.ptr); }
    uint32_t point_adjacency_size() const override { return adjacency_size; }
    const uint32_t *point_adjacency_offsets() const override { return static_cast<const uint32_t *>(offsets.ptr); }
    const uint32_t *vert_to_tet() const override { return nullptr; }

    // delaunay.cu:273-370.  Returns whether the points were re-sorted (the caller then applies permutation()).
    bool rebuild(const void *points, uint32_t num_points, bool incremental) override {
        if (num_points < 32) throw std::runtime_error("Delaunay triangulation does not support less than 32 points");
        const bool keep_order = incremental && num_points == n && adjacency.ptr;
        const float *pts = static_cast<const float *>(points);
        if (!keep_order) {
            sorted.expand((size_t)num_points * 12);
            perm.expand((size_t)num_points * 4);
            workspace.expand(rf_kd_order_workspace_bytes(num_points));
            check(rf_kd_order(pts, num_points, static_cast<uint32_t *>(perm.ptr), static_cast<float *>(sorted.ptr),
                              workspace.ptr, workspace.bytes, mem.stream));
            pts = static_cast<const float *>(sorted.ptr);
        }
        uint32_t p2 = 1;
        while (p2 < num_points) p2 <<= 1;
        tree.expand((size_t)p2 * 24);
        check(rf_build_aabb_tree(pts, num_points, static_cast<float *>(tree.ptr), mem.stream));

        size_t capacity = 20ull * num_points;   // the reference gives up beyond 20 tetrahedra per point
        size_t ws_bytes = rf_delaunay_workspace_bytes(num_points);
        uint32_t info[12];
        for (;;) {
            next_adjacency.expand(capacity * 4);
            next_offsets.expand(((size_t)num_points + 1) * 4);
            workspace.expand(ws_bytes);
            const int rc = rf_delaunay_adjacency(
                pts, num_points, static_cast<const float *>(tree.ptr),
                keep_order ? static_cast<const uint32_t *>(adjacency.ptr) : nullptr,
                keep_order ? static_cast<const uint32_t *>(offsets.ptr) : nullptr,
                static_cast<uint32_t *>(next_adjacency.ptr), (uint32_t)capacity, static_cast<uint32_t *>(next_offsets.ptr),
                info, workspace.ptr, workspace.bytes, mem.stream);
            if (rc == RF_ERR_WORKSPACE && info[2] > 0 &&
                rf_delaunay_workspace_bytes_for(num_points, info[2]) > workspace.bytes) {
                ws_bytes = rf_delaunay_workspace_bytes_for(num_points, info[2]);   // a cloud that is mostly rim
                continue;
            }
            check(rc);
            if (info[0] <= capacity) break;
            capacity = info[0];
        }
        // the reference's own failure type, for the caller's perturb-and-retry loop (scene.py:160-186)
        if (info[3]) throw TriangulationFailedError("duplicate points found");
        if (info[1] || info[4]) throw TriangulationFailedError("ambiguous triangulation");
        std::swap(adjacency.ptr, next_adjacency.ptr);
        std::swap(adjacency.bytes, next_adjacency.bytes);
        std::swap(offsets.ptr, next_offsets.ptr);
        std::swap(offsets.bytes, next_offsets.bytes);
        adjacency_size = info[0];
        n = num_points;
        return !keep_order;
    }

  private:
    static void check(int rc) {
        if (rc != RF_OK) throw std::runtime_error(rf_last_error());
    }
    DeviceMemoryHooks mem;
    Buffer sorted, perm, tree, adjacency, next_adjacency, offsets, next_offsets, workspace;
    uint32_t n = 0, adjacency_size = 0;
};

}  // namespace

std::unique_ptr<Triangulation> Triangulation::create_triangulation(const void *points, uint32_t num_points) {
    return std::make_unique<HIPTriangulation>(points, num_points);
}

}  // namespace radfoam

// ---- C entry points for tests/test_ref_binding.py: everything goes through the reference's virtual interface ----------

namespace {
thread_local char g_tri_error[512] = "";
thread_local int g_tri_failed_error = 0;   // 1: the exception was a TriangulationFailedError
}

extern "C" {
// dev_mem.cpp
void *rb_dev_alloc(size_t bytes);
void rb_dev_free(void *ptr);
void rb_dev_zero(void *ptr, size_t bytes, void *stream);

// the allocator the triangulation (and the pipeline) go through: hipMalloc / hipFree here, the torch caching
// allocator in a reference build
void tb_init(void *stream) {
    radfoam::set_hip_pipeline_memory(radfoam::DeviceMemoryHooks{rb_dev_alloc, rb_dev_free, rb_dev_zero, stream});
}

const char *tb_last_error() { return g_tri_error; }
int tb_last_error_is_triangulation_failed() { return g_tri_failed_error; }

void *tb_create(const void *points, uint32_t num_points) {
    g_tri_error[0] = 0;
    g_tri_failed_error = 0;
    try {
        return radfoam::Triangulation::create_triangulation(points, num_points).release();
    } catch (const radfoam::TriangulationFailedError &e) {
        g_tri_failed_error = 1;
        std::snprintf(g_tri_error, sizeof(g_tri_error), "%s", e.what());
    } catch (const std::exception &e) {
        std::snprintf(g_tri_error, sizeof(g_tri_error), "%s", e.what());
    }
    return nullptr;
}

void tb_destroy(void *h) { delete static_cast<radfoam::Triangulation *>(h); }

// 1 / 0 = rebuild()'s return value, -1 = it threw
int tb_rebuild(void *h, const void *points, uint32_t num_points, int incremental) {
    g_tri_error[0] = 0;
    g_tri_failed_error = 0;
    try {
        return static_cast<radfoam::Triangulation *>(h)->rebuild(points, num_points, incremental != 0) ? 1 : 0;
    } catch (const radfoam::TriangulationFailedError &e) {
        g_tri_failed_error = 1;
        std::snprintf(g_tri_error, sizeof(g_tri_error), "%s", e.what());
    } catch (const std::exception &e) {
        std::snprintf(g_tri_error, sizeof(g_tri_error), "%s", e.what());
    }
    return -1;
}

uint32_t tb_num_points(void *h) { return static_cast<radfoam::Triangulation *>(h)->num_points(); }
uint32_t tb_point_adjacency_size(void *h) { return static_cast<radfoam::Triangulation *>(h)->point_adjacency_size(); }
const void *tb_permutation(void *h) { return static_cast<radfoam::Triangulation *>(h)->permutation(); }
const void *tb_point_adjacency(void *h) { return static_cast<radfoam::Triangulation *>(h)->point_adjacency(); }
const void *tb_point_adjacency_offsets(void *h) { return static_cast<radfoam::Triangulation *>(h)->point_adjacency_offsets(); }
uint32_t tb_num_tets(void *h) { return static_cast<radfoam::Triangulation *>(h)->num_tets(); }

}  // extern "C"
