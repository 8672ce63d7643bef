// probe: does a 16-byte global load at an 8-byte-aligned address return the right bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((aligned(8))) P { uint32_t a, b, c, d; };
__global__ void k(const uint2 *src, uint32_t off, uint32_t *out) {
    P p = *reinterpret_cast<const P *>(src + off + threadIdx.x);
    out[4 * threadIdx.x + 0] = p.a; out[4 * threadIdx.x + 1] = p.b;
    out[4 * threadIdx.x + 2] = p.c; out[4 * threadIdx.x + 3] = p.d;
}
__global__ void k6(const float *src, float *out) {
    const float *r = src + 6 * threadIdx.x;
    float s = 0; for (int i = 0; i < 6; ++i) s += r[i] * (i + 1);
    out[threadIdx.x] = s;
}
int main() {
    uint32_t h[64]; for (int i = 0; i < 64; ++i) h[i] = i;
    uint32_t *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4 * 4 * 8);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (uint32_t off = 0; off < 2; ++off) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, (const uint2 *)d, off, o);
        uint32_t r[16]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("off=%u:", off); for (int i = 0; i < 16; ++i) printf(" %u", r[i]); printf("\n");
    }
    float hf[64]; for (int i = 0; i < 64; ++i) hf[i] = (float)i;
    float *df, *of; hipMalloc(&df, sizeof(hf)); hipMalloc(&of, 64);
    hipMemcpy(df, hf, sizeof(hf), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k6, dim3(1), dim3(4), 0, 0, df, of);
    float rf[4]; hipMemcpy(rf, of, sizeof(rf), hipMemcpyDeviceToHost);
    printf("k6: %g %g %g %g (expect 70 196 322 448)\n", rf[0], rf[1], rf[2], rf[3]);
    return 0;
}
