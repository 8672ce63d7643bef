// rf_grad_exchange.hip -- the device side of the multi-GPU gradient exchange (radfoam_amd/dist.py).
//
// The reference is a single-GPU program; the north star shards one image by rows over the GPUs of a
// node and sums the partial gradients.  A rank's rays cross only a thin wedge of the foam, so its dense
// [N][3+A] gradient buffer (248 MB for the 2M-point SH-2 foam) is almost all zeros: instead of all-reducing
// it, each rank compacts the rows that hold anything, the ranks all-gather those (a few MB each over xGMI),
// and every rank adds all of them to its dense buffer in rank order -- the same sums, bit for bit, on every
// rank.  Both kernels are plain HBM streaming: one pass over the dense buffer, one over the packed rows.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/radfoam_hip.h"
#include "rf_host.hpp"
#include "rf_wave.hpp"

namespace rf {

inline __host__ __device__ uint32_t grad_row_pitch(uint32_t attr_dim) { return (1u + 3u + attr_dim + 3u) & ~3u; }

// One lane per cell row.  A wave's 64 rows are contiguous in both buffers (64 * A floats), so although a
// lane walks its own row every cache line fetched is consumed in full; rows that hold anything reserve
// their output slots with one atomic per wave.
template <bool VEC4>
__global__ __launch_bounds__(256) void compact_grad_rows_kernel(const float *__restrict__ points_grad,
                                                                const float *__restrict__ attr_grad,
                                                                uint32_t num_points, uint32_t attr_dim,
                                                                uint32_t attr_pitch, uint32_t capacity,
                                                                uint32_t *__restrict__ count,
                                                                float *__restrict__ packed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pitch = grad_row_pitch(attr_dim);
    bool any = false;
    if (i < num_points) {
        const float *pg = points_grad + 3 * (size_t)i;
        const float *ag = attr_grad + (size_t)i * attr_pitch;
        uint32_t acc = __builtin_bit_cast(uint32_t, pg[0]) | __builtin_bit_cast(uint32_t, pg[1]) |
                       __builtin_bit_cast(uint32_t, pg[2]);
        if constexpr (VEC4) {
            const uint4 *ag4 = reinterpret_cast<const uint4 *>(ag);
            for (uint32_t k = 0; k < attr_dim / 4u; ++k) {
                const uint4 w = ag4[k];
                acc |= w.x | w.y | w.z | w.w;
            }
        } else {
            for (uint32_t k = 0; k < attr_dim; ++k) acc |= __builtin_bit_cast(uint32_t, ag[k]);
        }
        any = (acc & 0x7FFFFFFFu) != 0u;   // -0.0 alone does not make a row worth sending
    }
    const uint64_t mask = ballot(any);
    if (mask == 0ull) return;
    uint32_t base = 0;
    if (lane == 0u) base = atomicAdd(count, (uint32_t)__builtin_popcountll(mask));
    base = readlane(base, 0);
    if (!any) return;
    const uint32_t slot = base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
    if (slot >= capacity) return;
    float *dst = packed + (size_t)slot * pitch;
    const float *pg = points_grad + 3 * (size_t)i;
    const float *ag = attr_grad + (size_t)i * attr_pitch;
    dst[0] = __builtin_bit_cast(float, i);
    dst[1] = pg[0];
    dst[2] = pg[1];
    dst[3] = pg[2];
    if constexpr (VEC4) {
        const float4 *ag4 = reinterpret_cast<const float4 *>(ag);
        float4 *d4 = reinterpret_cast<float4 *>(dst + 4);
        for (uint32_t k = 0; k < attr_dim / 4u; ++k) d4[k] = ag4[k];
    } else {
        for (uint32_t k = 0; k < attr_dim; ++k) dst[4 + k] = ag[k];
        for (uint32_t k = 4u + attr_dim; k < pitch; ++k) dst[k] = 0.0f;
    }
}

// One thread per packed scalar: reads of the packed rows are coalesced, a row's 3 + A writes land in
// two contiguous runs.  MODE 0 adds, MODE 1 zeroes the addressed rows.
template <int MODE>
__global__ __launch_bounds__(256) void scatter_grad_rows_kernel(const float *__restrict__ packed, uint32_t num_rows,
                                                                uint32_t num_points, uint32_t attr_dim,
                                                                uint32_t attr_pitch,
                                                                float *__restrict__ points_grad,
                                                                float *__restrict__ attr_grad) {
    const uint32_t pitch = grad_row_pitch(attr_dim);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)num_rows * pitch) return;
    const uint32_t row = (uint32_t)(idx / pitch), col = (uint32_t)(idx - (size_t)row * pitch);
    if (col == 0u || col >= 4u + attr_dim) return;
    const uint32_t cell = __builtin_bit_cast(uint32_t, packed[(size_t)row * pitch]);
    if (cell >= num_points) return;
    float *dst = col < 4u ? points_grad + 3 * (size_t)cell + (col - 1u) : attr_grad + (size_t)cell * attr_pitch + (col - 4u);
    if constexpr (MODE == 0) {
        const float v = packed[idx];
        if (v != 0.0f) *dst = *dst + v;
    } else {
        *dst = 0.0f;
    }
}

}  // namespace rf

using namespace rf;

extern "C" {

uint32_t rf_grad_row_pitch(uint32_t attr_dim) { return grad_row_pitch(attr_dim); }

int rf_compact_grad_rows_pitched(const float *points_grad, const float *attr_grad, uint32_t num_points,
                                 uint32_t attr_dim, uint32_t attr_pitch, uint32_t capacity, uint32_t *count, float *packed,
                                 void *stream) {
    g_err[0] = 0;
    if (num_points == 0) return RF_OK;
    if (!points_grad || !attr_grad || !count || (capacity && !packed) || attr_dim == 0)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_compact_grad_rows: null pointer");
    if (attr_pitch < attr_dim) return fail(RF_ERR_INVALID_ARGUMENT, "rf_compact_grad_rows: attr_pitch below attr_dim");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((num_points + 255u) / 256u), block(256);
    // rows readable as 16-byte vectors iff the row pitch and the base keep them aligned
    const bool vec4 = (attr_dim % 4u) == 0u && (attr_pitch % 4u) == 0u && (reinterpret_cast<uintptr_t>(attr_grad) % 16u) == 0u;
    if (vec4)
        hipLaunchKernelGGL(compact_grad_rows_kernel<true>, grid, block, 0, s, points_grad, attr_grad, num_points,
                           attr_dim, attr_pitch, capacity, count, packed);
    else
        hipLaunchKernelGGL(compact_grad_rows_kernel<false>, grid, block, 0, s, points_grad, attr_grad, num_points,
                           attr_dim, attr_pitch, capacity, count, packed);
    return check_launch("rf_compact_grad_rows");
}

int rf_compact_grad_rows(const float *points_grad, const float *attr_grad, uint32_t num_points,
                         uint32_t attr_dim, uint32_t capacity, uint32_t *count, float *packed, void *stream) {
    return rf_compact_grad_rows_pitched(points_grad, attr_grad, num_points, attr_dim, attr_dim, capacity, count, packed,
                                        stream);
}

int rf_scatter_grad_rows_pitched(const float *packed, uint32_t num_rows, uint32_t num_points, uint32_t attr_dim,
                                 uint32_t attr_pitch, int mode, float *points_grad, float *attr_grad, void *stream) {
    g_err[0] = 0;
    if (num_rows == 0) return RF_OK;
    if (!packed || !points_grad || !attr_grad || attr_dim == 0)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_scatter_grad_rows: null pointer");
    if (attr_pitch < attr_dim) return fail(RF_ERR_INVALID_ARGUMENT, "rf_scatter_grad_rows: attr_pitch below attr_dim");
    if (mode != 0 && mode != 1) return fail(RF_ERR_INVALID_ARGUMENT, "rf_scatter_grad_rows: mode must be 0 or 1");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t total = (size_t)num_rows * grad_row_pitch(attr_dim);
    const dim3 grid((unsigned)((total + 255u) / 256u)), block(256);
    if (mode == 0)
        hipLaunchKernelGGL(scatter_grad_rows_kernel<0>, grid, block, 0, s, packed, num_rows, num_points, attr_dim,
                           attr_pitch, points_grad, attr_grad);
    else
        hipLaunchKernelGGL(scatter_grad_rows_kernel<1>, grid, block, 0, s, packed, num_rows, num_points, attr_dim,
                           attr_pitch, points_grad, attr_grad);
    return check_launch("rf_scatter_grad_rows");
}

int rf_scatter_grad_rows(const float *packed, uint32_t num_rows, uint32_t num_points, uint32_t attr_dim,
                         int mode, float *points_grad, float *attr_grad, void *stream) {
    return rf_scatter_grad_rows_pitched(packed, num_rows, num_points, attr_dim, attr_dim, mode, points_grad, attr_grad,
                                        stream);
}

}  // extern "C"
