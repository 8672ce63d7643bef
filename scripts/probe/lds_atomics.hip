// Probe: LDS update throughput on gfx950 -- ds_add_f32 patterns vs float4 read-modify-write.
// hipcc --offload-arch=gfx950 -O3 scripts/probe/lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 2048;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cycles, int stride) {
    __shared__ __attribute__((aligned(16))) float s[256 * 36];
    unsigned int *su = reinterpret_cast<unsigned int *>(s);
    double *sd = reinterpret_cast<double *>(s);
    unsigned int keep = 0;
    for (int i = threadIdx.x; i < 256 * 36; i += 256) s[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v = 1.0f + lane;
    unsigned long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
        if (MODE == 0) {            // 64 lanes, consecutive addresses
            atomicAdd(&s[wave * 2304 + ((it & 31) * 64 + lane)], v);
        } else if (MODE == 1) {     // 28 lanes, consecutive addresses
            if (lane < 28) atomicAdd(&s[wave * 2304 + (it & 63) * 36 + lane], v);
        } else if (MODE == 2) {     // 64 lanes, row-strided (lane -> row lane, same column): stride 36 floats
            atomicAdd(&s[(lane * stride + (it & 31)) % (256 * 36)], v);
        } else if (MODE == 3) {     // 64 lanes, groups of 4 lanes share an address
            atomicAdd(&s[wave * 2304 + ((lane >> 2) * 36 + (it & 31))], v);
        } else if (MODE == 4) {     // 64 lanes same address
            atomicAdd(&s[wave * 2304 + (it & 31)], v);
        } else if (MODE == 5) {     // float4 RMW, one row per lane (stride 36 floats)
            volatile float4 *r = reinterpret_cast<volatile float4 *>(&s[((wave * 64 + lane) * 36) + 4 * (it & 7)]);
            float4 x;
            x.x = r->x; x.y = r->y; x.z = r->z; x.w = r->w;
            x.x += v; x.y += v; x.z += v; x.w += v;
            r->x = x.x; r->y = x.y; r->z = x.z; r->w = x.w;
        } else if (MODE == 8) {     // ds_add_u32 no return, 64 lanes consecutive
            atomicAdd(&su[wave * 2304 + ((it & 31) * 64 + lane)], 1u);
        } else if (MODE == 9) {     // ds_add_rtn_u32, 64 lanes, groups of 4 share an address
            keep += atomicAdd(&su[wave * 2304 + ((lane >> 2) * 36 + (it & 31))], 1u);
        } else if (MODE == 10) {    // ds_add_f64, 64 lanes consecutive
            atomicAdd(&sd[wave * 1152 + ((it & 15) * 64 + lane)], (double)v);
        } else if (MODE == 13) {    // ds_add_f64, row per lane (stride in doubles), same column
            atomicAdd(&sd[((lane * stride) + (it & 15)) % (128 * 36)], (double)v);
        } else if (MODE == 14) {    // ds_add_f64, groups of 4 lanes share an address, rows strided
            atomicAdd(&sd[(((lane >> 2) * stride) + (it & 15)) % (128 * 36)], (double)v);
        } else if (MODE == 15) {    // ds_add_f64, 64 lanes same address
            atomicAdd(&sd[wave * 1152 + (it & 15)], (double)v);
        } else if (MODE == 16) {    // ds_add_f64, random rows (hash of lane and it), stride in doubles
            unsigned r = ((lane * 2654435761u + it * 40503u) >> 20) % 152u;
            atomicAdd(&sd[r * stride + (it & 15)], (double)v);
        } else if (MODE == 11) {    // ds_cmpst_rtn_b32, 64 lanes distinct
            keep += atomicCAS(&su[wave * 2304 + ((it & 31) * 64 + lane)], (unsigned)it, (unsigned)it + 1u);
        } else if (MODE == 12) {    // ds_max_f32 (no return)
            __hip_atomic_fetch_max(&s[wave * 2304 + ((it & 31) * 64 + lane)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 6) {     // two records per instruction: lanes 0-27 and 32-59
            int rec = lane >> 5, c = lane & 31;
            if (c < 28) atomicAdd(&s[wave * 2304 + ((it * 2 + rec) & 63) * 36 + c], v);
        } else if (MODE == 7) {     // plain (non-atomic) RMW b32, 28 lanes consecutive
            if (lane < 28) { float *q = &s[wave * 2304 + (it & 63) * 36 + lane]; *q = *q + v; }
        }
    }
    unsigned long long t1 = clock64();
    __syncthreads();
    float acc = (float)keep;
    for (int i = threadIdx.x; i < 256 * 36; i += 256) acc += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks_per_cu, int stride = 36) {
    float *out; unsigned long long *cyc;
    int nb = 256 * blocks_per_cu;
    hipMalloc(&out, nb * 256 * sizeof(float));
    hipMalloc(&cyc, nb * sizeof(unsigned long long));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<nb, 256>>>(out, cyc, stride);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<nb, 256>>>(out, cyc, stride);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    // per-CU: blocks_per_cu blocks x 4 waves each issue kIters instructions
    double instr_per_cu = (double)blocks_per_cu * 4 * kIters;
    double ns_per_instr_cu = ms * 1e6 / instr_per_cu;
    printf("%-44s blocks/CU %d  %.3f ms  %.1f ns per wave-instruction per CU (%.1f clk @2.4GHz)\n", name, blocks_per_cu, ms,
           ns_per_instr_cu, ns_per_instr_cu * 2.4);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int b : {4}) {
        run<0>("ds_add_f32 64 lanes consecutive", b);
        run<1>("ds_add_f32 28 lanes consecutive", b);
        run<6>("ds_add_f32 2x28 lanes (two records)", b);
        run<2>("ds_add_f32 64 lanes stride 36 (row per lane)", b, 36);
        run<3>("ds_add_f32 groups of 4 lanes same address", b);
        run<4>("ds_add_f32 64 lanes same address", b);
        run<5>("float4 RMW row per lane (read+add+write)", b);
        run<8>("ds_add_u32 64 lanes consecutive", b);
        run<9>("ds_add_rtn_u32 groups of 4 same address", b);
        run<10>("ds_add_f64 64 lanes consecutive", b);
        run<13>("ds_add_f64 row per lane stride 31", b, 31);
        run<13>("ds_add_f64 row per lane stride 33", b, 33);
        run<14>("ds_add_f64 groups of 4 same addr, stride 31", b, 31);
        run<15>("ds_add_f64 64 lanes same address", b);
        run<16>("ds_add_f64 random rows of 152, stride 31", b, 31);
        run<11>("ds_cmpst_rtn_b32 64 lanes distinct", b);
        run<12>("ds_max_f32 64 lanes consecutive", b);
    }
    return 0;
}
