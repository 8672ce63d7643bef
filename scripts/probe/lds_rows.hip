// Probe: is the bank-conflict share of the image backward's LDS write-combining cache a LAYOUT problem or inherent?
// (SQ_LDS_BANK_CONFLICT is 438 M of 484 M LDS-active cycles per launch of backward_replay_cached_kernel.)
//
// The kernel's pattern (rf_kernels.hip, backward_replay_cached_kernel, SH degree 2): per lit wave-step every ACTIVE lane
// (those left after the DPP merges of same-cell lanes: 8-32 of 64) adds 28 doubles -- 27 colour gradients + the density
// gradient -- to "its" row of a table of 160 rows of 31 doubles, one ds_add_f64 per column, the same column in every lane:
//     for k in 0..27:  ds_add_f64  rows[row(lane) * 31 + k]  +=  v[k]
// where row(lane) is the hash slot of the lane's cell: effectively random.  A double covers two of the 64 four-byte banks,
// so 64 lanes need at least two passes whatever the addresses; lanes whose rows are congruent mod 32 collide on top.
//
// Variants (clocks per wave-instruction, 4 blocks x 4 waves per CU hammering the LDS as in the kernel):
//   ideal      consecutive doubles (lane i -> element i): the floor of a full-wave ds_add_f64
//   today      the pattern above, L active lanes, random rows, stride 31
//   distinct   the same, rows forced distinct mod 32 for the active lanes: what a conflict-free assignment would cost
//   rotated    stride 32, lane i adds column (k + i) mod 32 at iteration k: conflict-free for ANY rows -- but a lane would
//              have to pick v[(k + lane) % 32] from its registers by a lane-dependent index (not expressible without
//              staging v in LDS first: an upper bound on what a swizzle could buy, not a candidate)
//   u64        today's addresses with ds_add_u64 (fixed-point sums): is the integer path faster?
//   rtn        today's addresses with a returning ds_add_rtn_f64 (what a fused "claim + add" would need)
// hipcc --offload-arch=gfx950 -O3 scripts/probe/lds_rows.hip -o scripts/probe/lds_rows && scripts/probe/lds_rows
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kIters = 256;          // lit wave-steps per wave
constexpr int kCols = 28;
constexpr int kRows = 160;

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int active) {
    __shared__ __attribute__((aligned(16))) double s[kRows * 32];
    unsigned long long *su = reinterpret_cast<unsigned long long *>(s);
    for (int i = threadIdx.x; i < kRows * 32; i += 256) s[i] = 0.0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    double keep = 0.0;
    const double v = 1.0 + lane;
    for (int it = 0; it < kIters; ++it) {
        const unsigned h = hash32(lane * 0x9E3779B1u + (unsigned)it * 7919u + wave * 104729u + blockIdx.x * 31u);
        // which lanes are active this step: a pseudo-random subset of `active` lanes (the survivors of the DPP merges sit
        // at the low lane of their group; their places are irregular)
        const bool on = (hash32(lane + (unsigned)it * 64u) % 64u) < (unsigned)active;
        unsigned row = h % kRows;
        if (MODE == 2) row = (lane & 31u) + 32u * (h % (kRows / 32));      // distinct mod 32 among lanes 0..31 and 32..63
        if (MODE == 0) {
            // ideal: all 64 lanes, consecutive doubles (ignores `active`)
#pragma unroll
            for (int k = 0; k < kCols; ++k) atomicAdd(&s[(k * 64 + lane) % (kRows * 32)], v);
        } else if (MODE == 1 || MODE == 2) {
            if (on) {
                double *r = s + row * 31;
#pragma unroll
                for (int k = 0; k < kCols; ++k) atomicAdd(r + k, v);
            }
        } else if (MODE == 3) {
            if (on) {
                double *r = s + row * 32;
#pragma unroll
                for (int k = 0; k < kCols; ++k) atomicAdd(r + ((k + lane) & 31u), v);
            }
        } else if (MODE == 4) {
            if (on) {
                unsigned long long *r = su + row * 31;
#pragma unroll
                for (int k = 0; k < kCols; ++k) atomicAdd(r + k, (unsigned long long)(lane + 1u));
            }
        } else if (MODE == 5) {
            if (on) {
                double *r = s + row * 31;
#pragma unroll
                for (int k = 0; k < kCols; ++k) keep += atomicAdd(r + k, v);
            }
        }
    }
    __syncthreads();
    double acc = keep;
    for (int i = threadIdx.x; i < kRows * 32; i += 256) acc += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}

template <int MODE>
void run(const char *name, int active) {
    const int blocks_per_cu = 4, nb = 256 * blocks_per_cu;
    float *out;
    hipMalloc(&out, nb * 256 * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<MODE><<<nb, 256>>>(out, active);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<nb, 256>>>(out, active);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per CU: blocks_per_cu blocks x 4 waves x kIters steps x kCols instructions
    const double instr_per_cu = (double)blocks_per_cu * 4 * kIters * kCols;
    const double ns = ms * 1e6 / instr_per_cu;
    printf("%-58s active %2d  %7.3f ms  %6.2f ns per wave-instruction per CU = %6.1f clk @2.4 GHz  (%5.2f us per lit wave-step)\n",
           name, active, ms, ns, ns * 2.4, ns * kCols / 1e3);
    hipFree(out);
}

int main() {
    run<0>("ideal: 64 lanes, consecutive doubles", 64);
    for (int a : {8, 16, 24, 32, 48, 64}) {
        run<1>("today: random rows, stride 31 doubles", a);
        run<2>("distinct: rows distinct mod 32, stride 31", a);
        run<3>("rotated: stride 32, column (k + lane) % 32", a);
        run<4>("u64: today's addresses, ds_add_u64", a);
        run<5>("rtn: today's addresses, ds_add_rtn_f64", a);
    }
    return 0;
}
