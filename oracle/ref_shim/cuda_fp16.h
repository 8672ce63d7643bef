// Software __half / half2 for the CPU build of the reference kernels.  TEST INFRASTRUCTURE ONLY.
// Conversions: float -> half round-to-nearest-even, half -> float exact (as __float2half /
// __half2float).
#pragma once
#include <cstdint>
#include <cstring>

extern "C" uint16_t rfo_float_to_half(float);
extern "C" float rfo_half_to_float(uint16_t);

struct __half {
    uint16_t bits;
    __half() = default;
    __half(float f) : bits(rfo_float_to_half(f)) {}
    __half(double f) : bits(rfo_float_to_half((float)f)) {}
    __half(int i) : bits(rfo_float_to_half((float)i)) {}
    operator float() const { return rfo_half_to_float(bits); }
};
struct half2 { __half x, y; };
inline float __half2float(__half h) { return rfo_half_to_float(h.bits); }
inline __half __float2half(float f) { return __half(f); }
