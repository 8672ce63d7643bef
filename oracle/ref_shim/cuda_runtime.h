// CPU emulation of the CUDA device-code vocabulary radfoam's tracing kernels use.
// TEST INFRASTRUCTURE ONLY (oracle/_ref build).  A "kernel launch" is a loop that sets the
// thread-local built-in index variables and calls the kernel function once per thread.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <algorithm>
#include "cuda_fp16.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __constant__ static const

struct rfref_uint3 { unsigned x, y, z; };
extern thread_local rfref_uint3 threadIdx, blockIdx, blockDim, gridDim;

using std::max;
using std::min;

inline float atomicAdd(float *p, float v) {
    float old;
#pragma omp atomic capture
    { old = *p; *p += v; }
    return old;
}
inline __half atomicAdd(__half *p, __half v) {
    // half precision read-modify-write, rounded to half after every add (as the hardware does)
    __half old;
#pragma omp critical(rfref_half_atomic)
    { old = *p; *p = __half(float(*p) + float(v)); }
    return old;
}

// the tracer's only libm calls are expf / logf / fmaxf (+ tanf etc. in cast_ray, left to libm).
// RFREF_CANONICAL_LIBM routes expf/logf to the oracle's portable routines so the two can be
// compared without a libm-shaped difference.
#ifdef RFREF_CANONICAL_LIBM
extern "C" float rfo_expf(float);
extern "C" float rfo_logf(float);
#define expf rfo_expf
#define logf rfo_logf
#endif
