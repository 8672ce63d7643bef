// rf_math.hpp -- device arithmetic of the gfx950 tracer kernels.
//
// The kernels evaluate the reference's formulas (src/tracing/*.cuh, pipeline.cu) in ONE fixed
// fp32 evaluation order, documented in DESIGN.md ("canonical arithmetic"): every fused
// multiply-add is an explicit __builtin_fmaf and the translation unit is compiled with
// -ffp-contract=off, so the compiler neither adds nor removes contractions; '/' and sqrtf are
// IEEE correctly rounded (hipcc default); exp/log are the self-contained routines below (only
// +,*,fma,/ and integer ops), so results do not depend on a vendor libm.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rf {

constexpr uint32_t kNone = 0xFFFFFFFFu;

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// 3-vector dot in the pinned order e0 + (e1 + e2) with the leading product of each sum fused
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fma_(ax, bx, fma_(ay, by, az * bz));
}

__device__ __forceinline__ float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// exact fp16 -> fp32 of the low / high half of a dword (v_cvt_f32_f16, SDWA-selectable)
__device__ __forceinline__ float half_lo(uint32_t w) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu));
}
__device__ __forceinline__ float half_hi(uint32_t w) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
}
// fp32 -> fp16 round-to-nearest-even (v_cvt_f16_f32)
__device__ __forceinline__ uint16_t float_to_half_bits(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}

// exp(x): k = rint(x*log2e); r = x - k*ln2 (two-term Cody-Waite); degree-6 polynomial
// (Cephes expf coefficients); scale by 2^k with v_ldexp_f32 (one correctly rounded scaling,
// also into the denormal range).  < 1 ulp.
__device__ __forceinline__ float exp_(float x) {
    float k = __builtin_rintf(x * 1.44269502162933349609375f);
    float r = fma_(k, -0.693145751953125f, x);
    r = fma_(k, -1.42860676533018704526e-06f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    p = fma_(p, z, r);
    p = p + 1.0f;
    float res = __builtin_ldexpf(p, (int)k);
    res = (x > 88.72283935546875f) ? __builtin_inff() : res;
    res = (x < -103.97208404541015625f) ? 0.0f : res;
    res = (x != x) ? x : res;
    return res;
}

// log(x): classic k*ln2 + log1p(f) with s = f/(2+f) (msun/musl logf constants).  < 1 ulp.
// Only reached on the depth-quantile path.
__device__ __forceinline__ float log_(float x) {
    uint32_t ix = f2bits(x);
    int k = 0;
    if (ix >= 0x80000000u || ix < 0x00800000u) {
        if ((ix << 1) == 0) return -__builtin_inff();
        if (ix >= 0x80000000u) return __builtin_nanf("");
        k -= 25;
        x = x * 33554432.0f;
        ix = f2bits(x);
    }
    if (ix >= 0x7F800000u) return x;
    ix += 0x3F800000u - 0x3F3504F3u;
    k += (int)(ix >> 23) - 127;
    ix = (ix & 0x007FFFFFu) + 0x3F3504F3u;
    x = bits2f(ix);
    float f = x - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * fma_(w, 0.24279078841e+00f, 0.40000972152e+00f);
    float t2 = z * fma_(w, 0.28498786688e+00f, 0.66666662693e+00f);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    float inner = fma_(s, hfsq + R, dk * 9.0580006145e-06f);
    return fma_(dk, 6.9313812256e-01f, -((hfsq - inner) - f));
}

constexpr int sh_dim(int degree) { return (degree + 1) * (degree + 1); }

// Real SH basis values; reference: sh_coefficients<degree>, src/tracing/sh_utils.cuh:8-70.
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float (&sh)[sh_dim(DEG)]) {
    constexpr float C0 = 0.28209479177387814f;
    constexpr float C1 = 0.4886025119029199f;
    sh[0] = C0;
    if constexpr (DEG > 0) {
        sh[1] = -C1 * y;
        sh[2] = C1 * z;
        sh[3] = -C1 * x;
    }
    if constexpr (DEG > 1) {
        float xx = x * x, yy = y * y, zz = z * z;
        float xy = x * y, yz = y * z, xz = x * z;
        sh[4] = 1.0925484305920792f * xy;
        sh[5] = -1.0925484305920792f * yz;
        sh[6] = 0.31539156525252005f * ((2.0f * zz - xx) - yy);
        sh[7] = -1.0925484305920792f * xz;
        sh[8] = 0.5462742152960396f * (xx - yy);
        if constexpr (DEG > 2) {
            sh[9] = (-0.5900435899266435f * y) * fma_(3.0f, xx, -yy);
            sh[10] = (2.890611442640554f * xy) * z;
            sh[11] = (-0.4570457994644658f * y) * ((4.0f * zz - xx) - yy);
            sh[12] = (0.3731763325901154f * z) * fma_(-3.0f, yy, fma_(-3.0f, xx, 2.0f * zz));
            sh[13] = (-0.4570457994644658f * x) * ((4.0f * zz - xx) - yy);
            sh[14] = (1.445305721320277f * z) * (xx - yy);
            sh[15] = (-0.5900435899266435f * x) * fma_(-3.0f, yy, xx);
        }
    }
}

// (nx, ny, nz) / den with one shared reciprocal: the instruction sequence hipcc expands each IEEE
// fp32 divide into (v_rcp_f32, one Newton step on the reciprocal, two residual corrections of the
// quotient) without v_div_scale / v_div_fmas / v_div_fixup, which only act when an operand or the
// quotient is subnormal, huge, zero, infinite or NaN.  Bit-identical to three '/' otherwise.
__device__ __forceinline__ void div3(float nx, float ny, float nz, float den, float &qx, float &qy, float &qz) {
    float y = __builtin_amdgcn_rcpf(den);
    float e = fma_(-den, y, 1.0f);
    y = fma_(e, y, y);
    float q, r;
    q = nx * y; r = fma_(-den, q, nx); q = fma_(r, y, q); r = fma_(-den, q, nx); qx = fma_(r, y, q);
    q = ny * y; r = fma_(-den, q, ny); q = fma_(r, y, q); r = fma_(-den, q, ny); qy = fma_(r, y, q);
    q = nz * y; r = fma_(-den, q, nz); q = fma_(r, y, q); r = fma_(-den, q, nz); qz = fma_(r, y, q);
}

// div3 for a denominator that can leave the range where the short sequence is exact: numerators and
// denominator are first scaled by the same power of two (exact; what v_div_scale does for the IEEE divide) so
// that |den| is back in the normal range -- the quotients are unchanged.  Used for the compositing denominator
// T * (1 - alpha + 1e-6) of the backward pass, whose transmittance T decays into the subnormal range when a
// caller sets weight_threshold = 0; the bisector denominators (dp^2 of a face the ray crosses) never do.
// Branch-free: three selects and four v_ldexp_f32.
__device__ __forceinline__ void div3_guarded(float nx, float ny, float nz, float den, float &qx, float &qy, float &qz) {
    const uint32_t ex = (f2bits(den) >> 23) & 0xFFu;
    const int sc = ex < 67u ? 64 : (ex > 187u ? -64 : 0);
    div3(__builtin_ldexpf(nx, sc), __builtin_ldexpf(ny, sc), __builtin_ldexpf(nz, sc), __builtin_ldexpf(den, sc), qx, qy, qz);
}

// Both gradients of one bisector hit -- d(t)/d(p) and d(t)/d(q) of the ray's crossing of the bisector of (p, q) -- as
// bisector_grad(p, q, ...) and bisector_grad(q, p, ...) would return them, bit for bit, at little more than the cost of
// one: swapping p and q negates the normal, num and dp exactly (every rounding on the way is sign-symmetric) and leaves
// the midpoint and dp^2 alone, so the two calls share everything but their three numerators and the last steps of the
// divide.  The backward functor needs exactly such pairs: (prev, cur) and (cur, next) (pipeline.cu:243-262).
__device__ __forceinline__ void bisector_grad_pair(float px, float py, float pz, float qx, float qy, float qz, float ox,
                                                   float oy, float oz, float dx, float dy, float dz, float &gpx, float &gpy,
                                                   float &gpz, float &gqx, float &gqy, float &gqz) {
    float fnx = qx - px, fny = qy - py, fnz = qz - pz;
    float vx = (px + qx) / 2.0f - ox;
    float vy = (py + qy) / 2.0f - oy;
    float vz = (pz + qz) / 2.0f - oz;
    float num = dot3(vx, vy, vz, fnx, fny, fnz);
    float dp = dot3(fnx, fny, fnz, dx, dy, dz);
    float den = dp * dp;
    // div3's sequence with the reciprocal shared by six numerators
    float y = __builtin_amdgcn_rcpf(den);
    float e = fma_(-den, y, 1.0f);
    y = fma_(e, y, y);
    const float n[6] = {fma_(num, dx, dp * (ox - px)), fma_(num, dy, dp * (oy - py)), fma_(num, dz, dp * (oz - pz)),
                        fma_(num, dx, dp * (ox - qx)), fma_(num, dy, dp * (oy - qy)), fma_(num, dz, dp * (oz - qz))};
    float g[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float q = n[i] * y;
        float r = fma_(-den, q, n[i]);
        q = fma_(r, y, q);
        r = fma_(-den, q, n[i]);
        g[i] = fma_(r, y, q);
    }
    gpx = g[0]; gpy = g[1]; gpz = g[2];
    gqx = -g[3]; gqy = -g[4]; gqz = -g[5];
}

// d(t)/d(primal) of the ray/bisector(primal,opposite) hit; reference: cell_intersection_grad,
// src/tracing/tracing_utils.cuh:91-103 (uses the fp32 points, not the fp16 face table).
__device__ __forceinline__ void bisector_grad(float px, float py, float pz, float qx, float qy,
                                              float qz, float ox, float oy, float oz, float dx,
                                              float dy, float dz, float &gx, float &gy, float &gz) {
    float fnx = qx - px, fny = qy - py, fnz = qz - pz;
    float vx = (px + qx) / 2.0f - ox;
    float vy = (py + qy) / 2.0f - oy;
    float vz = (pz + qz) / 2.0f - oz;
    float num = dot3(vx, vy, vz, fnx, fny, fnz);
    float dp = dot3(fnx, fny, fnz, dx, dy, dz);
    float den = dp * dp;
    div3(fma_(num, dx, dp * (ox - px)), fma_(num, dy, dp * (oy - py)), fma_(num, dz, dp * (oz - pz)), den, gx, gy, gz);
}

}  // namespace rf
