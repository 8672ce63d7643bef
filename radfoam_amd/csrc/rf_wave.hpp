// rf_wave.hpp -- wave64 cross-lane primitives for gfx950 used by the tracer kernels:
// DPP / permlane-swap exchanges, an all-lanes sum, and the transposing butterfly that turns
// NV per-lane partial sums into NV wave totals (total j lands in lane j) in ~3*NV instructions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rf {

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ uint32_t readlane(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// DPP move: lanes whose source is invalid or masked off keep `old`.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                           CTRL, ROW_MASK, BANK_MASK, false));
}

// value of lane (l ^ BIT) for BIT in {1,2,4,8}
template <int BIT>
__device__ __forceinline__ float xor_lane(float x) {
    if constexpr (BIT == 1) {
        return dpp_mov<0xB1>(x, x);  // quad_perm [1,0,3,2]
    } else if constexpr (BIT == 2) {
        return dpp_mov<0x4E>(x, x);  // quad_perm [2,3,0,1]
    } else if constexpr (BIT == 4) {
        float t = dpp_mov<0x104, 0xF, 0x5>(x, x);  // row_shl:4 -> banks 0,2 read lane+4
        return dpp_mov<0x114, 0xF, 0xA>(t, x);     // row_shr:4 -> banks 1,3 read lane-4
    } else {
        static_assert(BIT == 8, "xor_lane: BIT must be 1,2,4,8");
        return dpp_mov<0x128>(x, x);  // row_ror:8
    }
}

// value of lane (l ^ BIT) for any power-of-two BIT < 64
template <int BIT>
__device__ __forceinline__ float xor_lane_any(float x) {
    if constexpr (BIT <= 8) {
        return xor_lane<BIT>(x);
    } else if constexpr (BIT == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x),
                                                  false, false);
        // r[0]: odd rows replaced by the even-row copy's data; r[1]: even rows replaced by odd rows' data
        const bool odd_row = (lane_id() & 16u) != 0u;
        return __builtin_bit_cast(float, (unsigned)(odd_row ? r[0] : r[1]));
    } else {
        static_assert(BIT == 32, "xor_lane_any: BIT must be a power of two below 64");
        auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x),
                                                  false, false);
        const bool hi = (lane_id() & 32u) != 0u;
        return __builtin_bit_cast(float, (unsigned)(hi ? r[0] : r[1]));
    }
}

// value of lane (l + BIT) for lanes whose bit BIT is clear (BIT in {1,2,4,8}; what the other lanes get is
// unspecified: 0 where the source falls outside the row).  A single DPP read with bound_ctrl, so that the compiler
// can fold it into the consuming VOP2 instruction.
template <int BIT>
__device__ __forceinline__ float from_upper_lane(float x) {
    constexpr int ctrl = BIT == 1 ? 0xB1      // quad_perm [1,0,3,2]
                         : BIT == 2 ? 0x4E    // quad_perm [2,3,0,1]
                         : BIT == 4 ? 0x104   // row_shl:4  (lane l reads l + 4)
                                    : 0x108;  // row_shl:8  (lane l reads l + 8)
    static_assert(BIT == 1 || BIT == 2 || BIT == 4 || BIT == 8, "from_upper_lane: BIT must be 1,2,4,8");
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true));
}

template <int BIT>
__device__ __forceinline__ uint32_t xor_lane_u(uint32_t x) {
    return __builtin_bit_cast(uint32_t, xor_lane_any<BIT>(__builtin_bit_cast(float, x)));
}

// a' + b' where the pair (a,b) is exchanged across lane bit 4 / bit 5:
// result(lane) = (bit clear ? a : b)(lane) + (bit clear ? a : b)(lane ^ BIT)
__device__ __forceinline__ float swap16_sum(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a),
                                              __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float swap32_sum(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a),
                                              __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// Sum over all 64 lanes, result in every lane.
__device__ __forceinline__ float wave_sum(float x) {
    x = x + xor_lane<1>(x);
    x = x + xor_lane<2>(x);
    x = x + xor_lane<4>(x);
    x = x + xor_lane<8>(x);
    x = swap16_sum(x, x);
    x = swap32_sum(x, x);
    return x;
}

// One butterfly stage across lane bit BIT on v[0..n): n >= 2 halves the number of values
// (lane with the bit clear keeps the even one of each pair), n == 1 just accumulates.
template <int BIT, int N>
__device__ __forceinline__ void butterfly_stage(float (&v)[N], int &n, uint32_t lane) {
    if (n >= 2) {
        const bool hi = (lane & (uint32_t)BIT) != 0u;
        const int half = n / 2;
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            if (k < half) {
                float a = v[2 * k], b = v[2 * k + 1];
                if constexpr (BIT == 16) {
                    v[k] = swap16_sum(a, b);
                } else if constexpr (BIT == 32) {
                    v[k] = swap32_sum(a, b);
                } else {
                    float keep = hi ? b : a;
                    float send = hi ? a : b;
                    v[k] = keep + xor_lane<BIT>(send);
                }
            }
        }
        n = half;
    } else {
        if constexpr (BIT == 16) {
            v[0] = swap16_sum(v[0], v[0]);
        } else if constexpr (BIT == 32) {
            v[0] = swap32_sum(v[0], v[0]);
        } else {
            v[0] = v[0] + xor_lane<BIT>(v[0]);
        }
    }
}

// NV (power of two, <= 64) per-lane values -> wave totals; total j is returned in lane j
// (and in every lane congruent to j modulo NV).
template <int NV>
__device__ __forceinline__ float transpose_reduce(float (&v)[NV], uint32_t lane) {
    static_assert(NV == 8 || NV == 16 || NV == 32 || NV == 64, "NV must be 8..64, power of two");
    int n = NV;
    butterfly_stage<1>(v, n, lane);
    butterfly_stage<2>(v, n, lane);
    butterfly_stage<4>(v, n, lane);
    butterfly_stage<8>(v, n, lane);
    butterfly_stage<16>(v, n, lane);
    butterfly_stage<32>(v, n, lane);
    return v[0];
}

}  // namespace rf
