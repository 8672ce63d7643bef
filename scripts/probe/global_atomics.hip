// Probe: scattered global fp32 atomic-add rate on gfx950 by memory scope, with a correctness check.
//   A: agent scope (unsafeAtomicAdd / default) into one shared buffer
//   B: workgroup scope into a buffer private to the XCD the block runs on (XCC_ID), reduced afterwards
// hipcc --offload-arch=gfx950 -O3 scripts/probe/global_atomics.hip -o scripts/probe/global_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kPerLane = 64;

template <int MODE>
__global__ __launch_bounds__(256) void scatter(float *buf, size_t n, size_t xcd_stride, unsigned long long seed) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 0xF;   // HW_REG_XCC_ID, 4 bits
    float *dst = MODE == 0 ? buf : buf + xcc * xcd_stride;
    unsigned long long x = seed + (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < kPerLane; ++i) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const size_t a = (size_t)((x * 0x2545F4914F6CDD1Dull) >> 20) % n;
        if (MODE == 3) {
            // locality: the 64 lanes of a wave hit a 1 KB window (8 lines) that moves every iteration
            unsigned long long w = __shfl(x, 0, 64);
            const size_t base = (size_t)((w * 0x2545F4914F6CDD1Dull) >> 20) % (n - 256);
            unsafeAtomicAdd(buf + base + (a & 255), 1.0f);
        } else if (MODE == 4) {
            // strided rows: lane -> its own 112-byte row near a moving window (attr_grad-like)
            unsigned long long w = __shfl(x, 0, 64);
            const size_t base = (size_t)((w * 0x2545F4914F6CDD1Dull) >> 20) % (n - 256 * 28);
            unsafeAtomicAdd(buf + base + (a & 255) * 28 + 27, 1.0f);
        } else if (MODE == 5 || MODE == 6) {
            // gradient rows as the flat-batch backward sends them: one instruction = 48 lanes on one 192-byte row (3 lines);
            // the rows of a block stay inside a 2 MB window (the wedge of the foam its rays cross) -- agent scope into
            // the shared buffer (5) against workgroup scope into the XCD's private copy (6)
            const unsigned long long w = __shfl(x, 0, 64);
            const size_t win = (size_t)((blockIdx.x * 0x9E3779B97F4A7C15ull) >> 20) % (n - (1u << 19) - 64);
            const size_t row = win + (((w * 0x2545F4914F6CDD1Dull) >> 20) % (1u << 13)) * 64;
            const unsigned lane = threadIdx.x & 63u;
            if (lane < 48u) {
                if (MODE == 5) unsafeAtomicAdd(buf + row + lane, 1.0f);
                else __hip_atomic_fetch_add(buf + xcc * xcd_stride + row + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else if (MODE == 0) unsafeAtomicAdd(dst + a, 1.0f);
        else if (MODE == 1) __hip_atomic_fetch_add(dst + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(dst + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

__global__ void reduce8(const float *buf, size_t n, size_t stride, float *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0;
    for (int x = 0; x < 8; ++x) s += buf[x * stride + i];
    out[i] = s;
}

__global__ void total(const float *v, size_t n, double *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += v[i];
    atomicAdd(out, s);
}

template <int MODE>
void run(const char *name, size_t n) {
    const int blocks = 256 * 64;
    const size_t stride = n;
    float *buf, *red; double *tot;
    constexpr bool kPerXcd = MODE == 1 || MODE == 2 || MODE == 6;
    hipMalloc(&buf, (kPerXcd ? 8 : 1) * n * sizeof(float));
    hipMalloc(&red, n * sizeof(float));
    hipMalloc(&tot, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(buf, 0, (kPerXcd ? 8 : 1) * n * sizeof(float));
        hipMemset(tot, 0, 8);
        hipDeviceSynchronize();
        hipEventRecord(a);
        scatter<MODE><<<blocks, 256>>>(buf, n, stride, 1234 + rep);
        hipEventRecord(b);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const float *res = buf;
    if (kPerXcd) { reduce8<<<(unsigned)((n + 255) / 256), 256>>>(buf, n, stride, red); res = red; }
    total<<<1024, 256>>>(res, n, tot);
    double h; hipMemcpy(&h, tot, 8, hipMemcpyDeviceToHost);
    const double ops = (double)blocks * 256 * kPerLane * ((MODE == 5 || MODE == 6) ? 48.0 / 64.0 : 1.0);
    printf("%-46s %8.3f ms  %7.2f G atomics/s   sum %.0f expected %.0f %s\n", name, ms, ops / ms / 1e6, h, ops,
           h == ops ? "OK" : "LOST UPDATES");
    hipFree(buf); hipFree(red); hipFree(tot);
}

int main() {
    for (size_t mb : {256}) {
        const size_t n = mb * 1024 * 1024 / 4;
        printf("-- %zu MB target buffer\n", mb);
        run<0>("agent scope, one buffer", n);
        run<1>("workgroup scope, buffer per XCD", n);
        run<2>("wavefront scope, buffer per XCD", n);
        run<3>("agent scope, wave hits a moving 1 KB window", n);
        run<4>("agent scope, 112-B rows near a moving window", n);
        run<5>("agent scope, 192-B rows (48 lanes), 2 MB window/block", n);
        run<6>("workgroup scope per XCD, 192-B rows, 2 MB window", n);
    }
    return 0;
}
