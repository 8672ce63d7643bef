/*
 * rf_oracle.c -- CPU restatement of the radfoam tracer.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (radfoam_amd/) never does.
 *
 * PARITY PIN.  theialab/radfoam ships no tests, golden vectors or CPU tracer (SURVEY.md section
 * 4 / 8c), and its CUDA build cannot run here (no nvcc, no NVIDIA GPU, Eigen submodule empty), so
 * parity against the CUDA BINARY is unpinned.  What is pinned: this restatement agrees with the
 * reference's own kernel SOURCE TEXT -- /root/reference/src/tracing/{pipeline.cu kernels,
 * tracing_utils.cuh, sh_utils.cuh, camera.h} compiled for the CPU against stand-ins for CUDA and
 * Eigen (oracle/Makefile.ref, oracle/ref_driver.cpp, oracle/ref_shim/) -- on every case of
 * tests/golden/ and on live random cases (tests/test_reference_source.py): integer outputs
 * equal, rgba within 2e-6, gradients within 2e-4 relative.  It is further validated by a float64
 * twin, finite differences, closed-form cases and invariants (tests/test_oracle.py).  Citations
 * below are relative to /root/reference.
 *
 *   rfo_build_adjacent_diff  <- prefetch_adjacent_diff_kernel  src/tracing/pipeline.cu:546-568
 *   walk loop (trace_ray)    <- trace<>                        src/tracing/tracing_utils.cuh:8-89
 *   bisector_grad            <- cell_intersection_grad         src/tracing/tracing_utils.cuh:91-103
 *   sh_basis                 <- sh_coefficients<>              src/tracing/sh_utils.cuh:34-70
 *   sh_to_rgb                <- load_sh_as_rgb<>               src/tracing/sh_utils.cuh:72-83
 *   rfo_trace_forward        <- forward kernel                 src/tracing/pipeline.cu:14-130
 *   rfo_trace_backward       <- backward kernel                src/tracing/pipeline.cu:132-343
 *   rfo_trace_benchmark      <- benchmark kernel + cast_ray    src/tracing/pipeline.cu:472-544,
 *                               + make_rgba8                   src/tracing/camera.h:56-85,
 *                                                              src/tracing/tracing_utils.cuh:105-115
 *
 * PINNED ARITHMETIC.  The CUDA build is fp32 with nvcc's default FMA contraction,
 * IEEE divide/sqrt and CUDA's libm; none of that is reproducible bit-for-bit elsewhere.
 * This oracle pins ONE concrete evaluation so that the HIP kernels can be compared exactly:
 *   - fp32 throughout, compile with -ffp-contract=off, every fused op written as fmaf();
 *   - 3-vector dot(a,b) = fmaf(a0,b0, fmaf(a1,b1, a2*b2))  (Eigen's fixed-size redux
 *     associates e0+(e1+e2); nvcc fuses the leading product of each sum);
 *   - x*y + z*w  ->  fmaf(x,y, z*w);   a - b*c -> fmaf(-b,c,a);
 *   - IEEE correctly rounded '/', sqrtf;
 *   - expf/logf are the self-contained rf_expf/rf_logf below (+,*,fma,/ and integer ops
 *     only, so they evaluate identically on x86 and gfx950);
 *   - float->half is round-to-nearest-even, half->float exact.
 * The float64 twin (REAL=double, no half rounding of the face table) is for validating
 * the maths, not for parity.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RFO_NONE 0xFFFFFFFFu

typedef struct {
    float weight_threshold;     /* src/tracing/pipeline.h:10-20 */
    uint32_t max_intersections;
} rfo_settings;

typedef struct {
    /* exact counts for the algorithmic-bytes figure of SURVEY.md 8(d) */
    uint64_t cells_scanned;   /* cells whose face list was scanned           */
    uint64_t faces_scanned;   /* sum of face counts of those cells           */
    uint64_t hops;            /* scans that found an exit face (adj+point)   */
    uint64_t segments;        /* functor invocations (t1 > t0)               */
    uint64_t segments_lit;    /* ... with density > 1e-6 (SH row needed)     */
} rfo_stats;

/* How the nearest exit of a cell is EVALUATED (the function is always the reference's, rf_oracle_body.inc: scan_cell):
 * 0 = the way the reference writes it (every face divided, running minimum of rounded quotients); 1 = the way the HIP
 * kernels do (cross-multiplied tournament + certificate, dividing scan for contested cells) -- bit-identical by
 * construction, which tests/test_oracle.py checks over whole frames.  Process-global, set by the tests only.
 * rfo_scan_contested counts the cells mode 1 handed to the dividing scan. */
/* products at most this many floats apart do not decide a comparison (rf_kernels.hip: kTieUlps, with the derivation) */
#ifndef RFO_TIE_ULPS
#define RFO_TIE_ULPS 3u
#endif
static int rfo_scan_mode = 0;
static uint64_t rfo_scan_contested = 0;
void rfo_set_scan_mode(int mode) { rfo_scan_mode = mode == 1 ? 1 : 0; }
int rfo_get_scan_mode(void) { return rfo_scan_mode; }
uint64_t rfo_get_scan_contested(void) { return rfo_scan_contested; }
void rfo_reset_scan_contested(void) { rfo_scan_contested = 0; }

/* rfo_trace_paths: per-thread recorder of a ray's walk (scheduling studies of the kernels: scripts/model_*.py) */
static _Thread_local uint32_t *rfo_path_cells = NULL;
static _Thread_local float *rfo_path_t1 = NULL;
static _Thread_local uint32_t rfo_path_cap = 0;

/* ------------------------------------------------------------------------------------ */
/* half <-> float, software, RNE (== __float2half / __half2float)                        */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static uint16_t f2h(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) {                      /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? (0x0200u | ((ax >> 13) & 0x3FFu)) : 0));
    }
    if (ax >= 0x477FF000u) {                      /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (ax < 0x33000001u) {                       /* <= 2^-25 rounds to zero */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;    /* 24-bit significand */
    uint32_t shift;
    uint32_t hexp;
    if (e < -14) {                                /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t h;
    if (hexp == 0) {
        h = q;                                    /* q may reach 0x400 -> smallest normal */
    } else {
        h = ((hexp << 10) + (q - 0x400u));        /* carry propagates into the exponent   */
    }
    return (uint16_t)(sign | h);
}

static float h2f(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu;
    uint32_t m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * 5.9604644775390625e-08f; /* 2^-24, exact */
        return sign ? -v : v;
    }
    if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* ------------------------------------------------------------------------------------ */
/* rf_expf / rf_logf: portable, deterministic (stand-ins for CUDA expf/logf,             */
/* pipeline.cu:76,87; both within ~1 ulp of the true function)                            */

static float rf_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -103.97208404541015625f) return 0.0f;
    float k = rintf(x * 1.44269502162933349609375f);
    float r = fmaf(k, -0.693145751953125f, x);
    r = fmaf(k, -1.42860676533018704526e-06f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, z, r);
    p = p + 1.0f;
    int ki = (int)k;
    /* p in (0.70, 1.42); scale by 2^ki in two exact-or-single-rounding steps */
    if (ki > 127) {
        p = p * 1.7014118346046923e+38f; /* 2^127 */
        ki -= 127;
    } else if (ki < -126) {
        p = p * u2f((uint32_t)(ki + 100 + 127) << 23);
        return p * 7.888609052210118e-31f; /* 2^-100 */
    }
    return p * u2f((uint32_t)(ki + 127) << 23);
}

static float rf_logf(float x) {
    uint32_t ix = f2u(x);
    int k = 0;
    if (ix >= 0x80000000u || ix < 0x00800000u) {
        if ((ix << 1) == 0) return -INFINITY;        /* log(+-0) */
        if (ix >= 0x80000000u) return NAN;           /* log(<0), log(-nan) */
        k -= 25;                                     /* subnormal: scale up */
        x = x * 33554432.0f;
        ix = f2u(x);
    }
    if (ix >= 0x7F800000u) return x;                 /* inf / nan */
    /* normalise x into [sqrt(2)/2, sqrt(2)) */
    ix += 0x3F800000u - 0x3F3504F3u;
    k += (int)(ix >> 23) - 127;
    ix = (ix & 0x007FFFFFu) + 0x3F3504F3u;
    x = u2f(ix);
    float f = x - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * fmaf(w, 0.24279078841e+00f, 0.40000972152e+00f);
    float t2 = z * fmaf(w, 0.28498786688e+00f, 0.66666662693e+00f);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    /* log(x) = k*ln2_hi - ((hfsq - (s*(hfsq+R) + k*ln2_lo)) - f) */
    float inner = fmaf(s, hfsq + R, dk * 9.0580006145e-06f);
    return fmaf(dk, 6.9313812256e-01f, -((hfsq - inner) - f));
}

/* ------------------------------------------------------------------------------------ */
/* Thread-local write-combining table of gradient rows (rfo_trace_backward with several threads): open addressing keyed
 * by cell, a row = [points_grad 3 | attr_grad A | point_error 1] floats.  A thread sums its rays' contributions here
 * without any atomic and merges the table into the shared output arrays when it is 5/8 full and when its rays are done
 * (a dense copy of the outputs per thread would be threads x N x (4 + A) floats -- 64 GB for 256
 * threads on the 2 M-point foam). */
typedef struct {
    uint32_t *keys;      /* cap entries; RFO_NONE = empty */
    float *rows;         /* cap x width */
    uint32_t cap, used, width;
    int A;
    float *pg, *ag, *pe; /* the shared outputs the table is merged into (pe may be NULL) */
} rfo_wc_t;

static void rfo_wc_init(rfo_wc_t *t, uint32_t cap, int A, float *pg, float *ag, float *pe) {
    t->cap = cap;
    t->used = 0;
    t->width = (uint32_t)A + 4u;
    t->A = A;
    t->pg = pg;
    t->ag = ag;
    t->pe = pe;
    t->keys = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    t->rows = (float *)calloc((size_t)cap * t->width, sizeof(float));
    memset(t->keys, 0xFF, sizeof(uint32_t) * cap);
}

static void rfo_wc_flush(rfo_wc_t *t) {
    for (uint32_t e = 0; e < t->cap; ++e) {
        const uint32_t cell = t->keys[e];
        if (cell == RFO_NONE) continue;
        float *row = t->rows + (size_t)e * t->width;
        for (int i = 0; i < 3; ++i)
            if (row[i] != 0.0f) {
#pragma omp atomic
                t->pg[3 * (size_t)cell + i] += row[i];
            }
        for (int i = 0; i < t->A; ++i)
            if (row[3 + i] != 0.0f) {
#pragma omp atomic
                t->ag[(size_t)cell * t->A + i] += row[3 + i];
            }
        if (t->pe && row[3 + t->A] != 0.0f) {
#pragma omp atomic
            t->pe[cell] += row[3 + t->A];
        }
        memset(row, 0, sizeof(float) * t->width);
        t->keys[e] = RFO_NONE;
    }
    t->used = 0;
}

static inline float *rfo_wc_row(rfo_wc_t *t, uint32_t cell) {
    const uint32_t mask = t->cap - 1u;
    uint32_t e = (cell * 2654435761u) >> 7 & mask;
    for (;;) {
        const uint32_t k = t->keys[e];
        if (k == cell) return t->rows + (size_t)e * t->width;
        if (k == RFO_NONE) break;
        e = (e + 1u) & mask;
    }
    if (t->used * 8u >= t->cap * 5u) {      /* merge, then insert into the empty table */
        rfo_wc_flush(t);
        e = (cell * 2654435761u) >> 7 & mask;
    }
    t->keys[e] = cell;
    t->used++;
    return t->rows + (size_t)e * t->width;
}

static void rfo_wc_free(rfo_wc_t *t) {
    free(t->keys);
    free(t->rows);
}

/* ------------------------------------------------------------------------------------ */
/* float32 instance (the pinned arithmetic)                                             */

#define REAL float
#define SUF(name) name##_f32
#define R_FMA(a, b, c) fmaf((a), (b), (c))
#define R_SQRT(a) sqrtf(a)
#define R_EXP(a) rf_expf(a)
#define R_LOG(a) rf_logf(a)
#define R_MAX(a, b) fmaxf((a), (b))
#define R_INF INFINITY
#define RFO_HALF_FACES 1
#include "rf_oracle_body.inc"
#undef REAL
#undef SUF
#undef R_FMA
#undef R_SQRT
#undef R_EXP
#undef R_LOG
#undef R_MAX
#undef R_INF
#undef RFO_HALF_FACES

/* float64 twin: libm exp/log, exact (unrounded) face offsets */
#define REAL double
#define SUF(name) name##_f64
#define R_FMA(a, b, c) ((a) * (b) + (c))
#define R_SQRT(a) sqrt(a)
#define R_EXP(a) exp(a)
#define R_LOG(a) log(a)
#define R_MAX(a, b) fmax((a), (b))
#define R_INF ((double)INFINITY)
#define RFO_HALF_FACES 0
#include "rf_oracle_body.inc"

/* ------------------------------------------------------------------------------------ */
/* exported helpers                                                                      */

uint16_t rfo_float_to_half(float f) { return f2h(f); }
float rfo_half_to_float(uint16_t h) { return h2f(h); }
float rfo_expf(float x) { return rf_expf(x); }
float rfo_logf(float x) { return rf_logf(x); }
int rfo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* diff[e] = half4(points[adj[e]] - points[i], 0)       src/tracing/pipeline.cu:546-568 */
void rfo_build_adjacent_diff(const float *points, uint32_t num_points, const uint32_t *adj,
                             const uint32_t *offsets, uint16_t *diff) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)num_points; ++i) {
        const float *p = points + 3 * i;
        for (uint32_t e = offsets[i]; e < offsets[i + 1]; ++e) {
            const float *q = points + 3 * (size_t)adj[e];
            diff[4 * (size_t)e + 0] = f2h(q[0] - p[0]);
            diff[4 * (size_t)e + 1] = f2h(q[1] - p[1]);
            diff[4 * (size_t)e + 2] = f2h(q[2] - p[2]);
            diff[4 * (size_t)e + 3] = 0;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* exported entry points (C-ABI shaped like Pipeline::trace_*, src/tracing/pipeline.h:58-131) */

static int pick_threads(int num_threads) {
#ifdef _OPENMP
    /* 0: OpenMP's default; otherwise what the caller asks for (bench.py: one thread per logical core it may run on) */
    if (num_threads <= 0) num_threads = omp_get_max_threads();
    if (num_threads > 1024) num_threads = 1024;
    return num_threads;
#else
    (void)num_threads;
    return 1;
#endif
}

static float *half_to_float_array(const uint16_t *src, size_t n) {
    float *dst = (float *)malloc((n ? n : 1) * sizeof(float));
    for (size_t i = 0; i < n; ++i) dst[i] = h2f(src[i]);
    return dst;
}

static void store_attr(void *dst, size_t i, float v, int attr_half) {
    if (attr_half) ((uint16_t *)dst)[i] = f2h(v);
    else ((float *)dst)[i] = v;
}

/*
 * forward                                                 src/tracing/pipeline.cu:595-642
 * attr_half: attributes / rgba / contribution are fp16 (pipeline instances <__half,d>).
 * diff may be NULL: the table is then built here exactly as the reference does per call.
 * DEVIATION (documented in DESIGN.md): half 'contribution' is accumulated in fp32 and
 * rounded once; the reference rounds every atomicAdd to half in nondeterministic order.
 */
void rfo_trace_forward(int sh_degree, int attr_half, rfo_settings settings, uint32_t num_points,
                       const float *points, const void *attributes, uint32_t adj_size,
                       const uint32_t *adj, const uint32_t *offsets, const uint16_t *diff,
                       uint32_t num_rays, const float *rays, const uint32_t *start, uint32_t nq,
                       const float *quantiles, void *rgba, float *qdepth, uint32_t *qidx,
                       uint32_t *num_intersections, void *contribution, int num_threads,
                       rfo_stats *stats) {
    int A = 1 + 3 * (sh_degree + 1) * (sh_degree + 1);
    float *attr_f = NULL;
    uint16_t *diff_own = NULL;
    if (attr_half) attr_f = half_to_float_array((const uint16_t *)attributes, (size_t)num_points * A);
    if (!diff) {
        diff_own = (uint16_t *)malloc(((size_t)adj_size + 1) * 4 * sizeof(uint16_t));
        rfo_build_adjacent_diff(points, num_points, adj, offsets, diff_own);
    }
    foam_t_f32 fm = {sh_degree, A, settings, num_points, points,
                     attr_half ? attr_f : (const float *)attributes, adj, offsets,
                     diff ? diff : diff_own};
    int nt = pick_threads(num_threads);
    /* scatter output: ONE fp32 accumulator shared by all threads (omp atomic), like the GPU's */
    float *contrib_f = NULL;
    if (contribution) contrib_f = (float *)calloc(num_points ? num_points : 1, sizeof(float));
    rfo_stats total = {0, 0, 0, 0, 0};
#pragma omp parallel num_threads(nt)
    {
        rfo_stats local = {0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 64)
        for (int64_t r = 0; r < (int64_t)num_rays; ++r) {
            float out[4];
            float qd[64];
            uint32_t qx[64];
            uint32_t nqq = nq > 64 ? 64 : nq;
            uint32_t n = forward_ray_f32(&fm, rays + 6 * r, start[r], nqq,
                                         quantiles ? quantiles + (size_t)r * nq : NULL, out, qd, qx,
                                         contrib_f, 1, stats ? &local : NULL, nt > 1);
            for (int c = 0; c < 4; ++c) store_attr(rgba, 4 * (size_t)r + c, out[c], attr_half);
            for (uint32_t i = 0; i < nqq; ++i) {
                qdepth[(size_t)r * nq + i] = qd[i];
                qidx[(size_t)r * nq + i] = qx[i];
            }
            if (num_intersections) num_intersections[r] = n;
        }
#pragma omp critical
        {
            total.cells_scanned += local.cells_scanned;
            total.faces_scanned += local.faces_scanned;
            total.hops += local.hops;
            total.segments += local.segments;
            total.segments_lit += local.segments_lit;
        }
    }
    if (contribution) {
        for (size_t i = 0; i < num_points; ++i) store_attr(contribution, i, contrib_f[i], attr_half);
        free(contrib_f);
    }
    if (stats) *stats = total;
    free(attr_f);
    free(diff_own);
}

/*
 * backward                                                src/tracing/pipeline.cu:644-700
 * points_grad[N,3] f32, attr_grad[N,A] (attr dtype), point_error[N] (attr dtype, optional)
 * are OVERWRITTEN with the sums (the binding zero-fills them, pipeline_bindings.cpp:441-452).
 * ray_grad is not an argument: the reference allocates but never writes it.
 * strict: see backward_ray.
 */
/* The cells every ray scans (cells[r][k], k < min(n_r, cap)) and the parameter at which it leaves each (t1[r][k], +inf
 * when the walk ends there); returns nothing else: a scheduling study tool, not part of the parity surface. */
void rfo_trace_paths(int sh_degree, rfo_settings settings, uint32_t num_points, const float *points,
                     const float *attributes, uint32_t adj_size, const uint32_t *adj, const uint32_t *offsets,
                     const uint16_t *diff, uint32_t num_rays, const float *rays, const uint32_t *start, uint32_t cap,
                     uint32_t *cells, float *t1, uint32_t *num_intersections, int num_threads) {
    int A = 1 + 3 * (sh_degree + 1) * (sh_degree + 1);
    (void)adj_size;
    foam_t_f32 fm = {sh_degree, A, settings, num_points, points, attributes, adj, offsets, diff};
    int nt = pick_threads(num_threads);
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (int64_t r = 0; r < (int64_t)num_rays; ++r) {
        float rgba[4];
        rfo_path_cells = cells + (size_t)r * cap;
        rfo_path_t1 = t1 + (size_t)r * cap;
        rfo_path_cap = cap;
        num_intersections[r] = forward_ray_f32(&fm, rays + 6 * (size_t)r, start[r], 0, NULL, rgba, NULL, NULL, NULL, 1,
                                               NULL, 0);
        rfo_path_cells = NULL;
        rfo_path_t1 = NULL;
        rfo_path_cap = 0;
    }
}

void rfo_trace_backward(int sh_degree, int attr_half, rfo_settings settings, uint32_t num_points,
                        const float *points, const void *attributes, uint32_t adj_size,
                        const uint32_t *adj, const uint32_t *offsets, const uint16_t *diff,
                        uint32_t num_rays, const float *rays, const uint32_t *start, uint32_t nq,
                        const float *quantiles, const uint32_t *qidx, const void *rgba,
                        const void *rgba_grad, const float *depth_grad, const void *ray_error,
                        float *points_grad, void *attr_grad, void *point_error, int strict,
                        int num_threads) {
    int A = 1 + 3 * (sh_degree + 1) * (sh_degree + 1);
    float *attr_f = NULL, *rgba_f = NULL, *g_f = NULL, *err_f = NULL;
    uint16_t *diff_own = NULL;
    if (attr_half) {
        attr_f = half_to_float_array((const uint16_t *)attributes, (size_t)num_points * A);
        rgba_f = half_to_float_array((const uint16_t *)rgba, (size_t)num_rays * 4);
        g_f = half_to_float_array((const uint16_t *)rgba_grad, (size_t)num_rays * 4);
        if (ray_error) err_f = half_to_float_array((const uint16_t *)ray_error, num_rays);
    }
    if (!diff) {
        diff_own = (uint16_t *)malloc(((size_t)adj_size + 1) * 4 * sizeof(uint16_t));
        rfo_build_adjacent_diff(points, num_points, adj, offsets, diff_own);
    }
    foam_t_f32 fm = {sh_degree, A, settings, num_points, points,
                     attr_half ? attr_f : (const float *)attributes, adj, offsets,
                     diff ? diff : diff_own};
    const float *rgba_p = attr_half ? rgba_f : (const float *)rgba;
    const float *g_p = attr_half ? g_f : (const float *)rgba_grad;
    const float *err_p = ray_error ? (attr_half ? err_f : (const float *)ray_error) : NULL;

    int nt = pick_threads(num_threads);
    /* fp32 accumulators of the outputs; with several threads every thread sums in its own write-combining table
     * (rfo_wc_t) and merges it into these -- no atomic in the walk */
    size_t per = (size_t)num_points * (3 + A + 1);
    float *tl = (float *)calloc(per ? per : 1, sizeof(float));
    float *pg = tl;
    float *ag = pg + (size_t)num_points * 3;
    float *pe = ag + (size_t)num_points * A;
#pragma omp parallel num_threads(nt)
    {
        rfo_wc_t table;
        sink_t_f32 sink = {pg, ag, point_error ? pe : NULL, 0, NULL};
        if (nt > 1) {
            rfo_wc_init(&table, 1u << 15, A, pg, ag, point_error ? pe : NULL);
            sink.wc = &table;
        }
#pragma omp for schedule(dynamic, 64)
        for (int64_t r = 0; r < (int64_t)num_rays; ++r) {
            backward_ray_f32(&fm, rays + 6 * r, start[r], nq,
                             quantiles ? quantiles + (size_t)r * nq : NULL,
                             qidx ? qidx + (size_t)r * nq : NULL, rgba_p + 4 * r, g_p + 4 * r,
                             depth_grad ? depth_grad + (size_t)r * nq : NULL,
                             err_p ? err_p + r : NULL, &sink, strict);
        }
        if (nt > 1) {
            rfo_wc_flush(&table);
            rfo_wc_free(&table);
        }
    }
    memcpy(points_grad, pg, sizeof(float) * 3 * (size_t)num_points);
    for (size_t i = 0; i < (size_t)num_points * A; ++i) store_attr(attr_grad, i, ag[i], attr_half);
    if (point_error)
        for (size_t i = 0; i < num_points; ++i) store_attr(point_error, i, pe[i], attr_half);
    free(tl);
    free(attr_f);
    free(rgba_f);
    free(g_f);
    free(err_f);
    free(diff_own);
}

/* camera by value, as the reference's Camera struct      src/tracing/camera.h:17-26 */
typedef struct {
    float position[3];
    float forward[3];
    float right[3];
    float up[3];
    float fov;
    uint32_t width;
    uint32_t height;
    uint32_t model; /* 0 pinhole, 1 fisheye */
} rfo_camera;

/* cast_ray                                                src/tracing/camera.h:56-85 */
static void cast_ray_f32(const rfo_camera *cam, uint32_t i, uint32_t j, float *ray6) {
    float aspect = (float)cam->width / (float)cam->height;
    float x = (float)i / (float)cam->width;
    float y = (float)j / (float)cam->height;
    float u = (2.0f * x - 1.0f) * aspect;
    float v = 1.0f - 2.0f * y;
    float mask = 1.0f;
    float d[3] = {0, 0, 0};
    if (cam->model == 0) {
        float w = 1.0f / tanf(cam->fov * 0.5f);
        for (int k = 0; k < 3; ++k)
            d[k] = fmaf(v, cam->up[k], fmaf(w, cam->forward[k], u * cam->right[k]));
    } else {
        float theta = atan2f(v, u);
        float phi = cam->fov * sqrtf(fmaf(u, u, v * v));
        if (phi >= 3.14159265358979323846f) {
            phi = 3.14159265358979323846f - 1e-6f;
            mask = 0.0f;
        }
        float a = sinf(phi) * cosf(theta);
        float b = sinf(phi) * sinf(theta);
        float c = cosf(phi);
        for (int k = 0; k < 3; ++k)
            d[k] = fmaf(c, cam->forward[k], fmaf(a, cam->right[k], b * cam->up[k]));
    }
    /* Eigen normalized(): divide by sqrt(squaredNorm) when squaredNorm > 0 */
    float n2 = fmaf(d[0], d[0], fmaf(d[1], d[1], d[2] * d[2]));
    if (n2 > 0.0f) {
        float n = sqrtf(n2);
        d[0] = d[0] / n;
        d[1] = d[1] / n;
        d[2] = d[2] / n;
    }
    ray6[0] = cam->position[0];
    ray6[1] = cam->position[1];
    ray6[2] = cam->position[2];
    ray6[3] = d[0] * mask;
    ray6[4] = d[1] * mask;
    ray6[5] = d[2] * mask;
}

/* make_rgba8                                    src/tracing/tracing_utils.cuh:105-115 */
static uint32_t make_rgba8(float r, float g, float b, float a) {
    r = fmaxf(0.0f, fminf(1.0f, r));
    g = fmaxf(0.0f, fminf(1.0f, g));
    b = fmaxf(0.0f, fminf(1.0f, b));
    a = fmaxf(0.0f, fminf(1.0f, a));
    int ri = (int)(r * 255.0f), gi = (int)(g * 255.0f), bi = (int)(b * 255.0f),
        ai = (int)(a * 255.0f);
    return ((uint32_t)ai << 24) | ((uint32_t)bi << 16) | ((uint32_t)gi << 8) | (uint32_t)ri;
}

void rfo_cast_rays(const rfo_camera *cam, float *rays) {
    for (uint32_t j = 0; j < cam->height; ++j)
        for (uint32_t i = 0; i < cam->width; ++i)
            cast_ray_f32(cam, i, j, rays + 6 * ((size_t)j * cam->width + i));
}

/* benchmark                                       src/tracing/pipeline.cu:472-544,738-765
 * diff is the caller-supplied half4 table [E,4] (benchmark.py:44-54).                   */
void rfo_trace_benchmark(int sh_degree, int attr_half, rfo_settings settings, uint32_t num_points,
                         const float *points, const void *attributes, const uint32_t *adj,
                         const uint32_t *offsets, const uint16_t *diff, const rfo_camera *cam,
                         uint32_t start_point, uint32_t *out_rgba8, int num_threads) {
    int A = 1 + 3 * (sh_degree + 1) * (sh_degree + 1);
    float *attr_f = NULL;
    if (attr_half) attr_f = half_to_float_array((const uint16_t *)attributes, (size_t)num_points * A);
    foam_t_f32 fm = {sh_degree, A, settings, num_points, points,
                     attr_half ? attr_f : (const float *)attributes, adj, offsets, diff};
    int nt = pick_threads(num_threads);
    int64_t total = (int64_t)cam->width * cam->height;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t idx = 0; idx < total; ++idx) {
        uint32_t pi = (uint32_t)(idx % cam->width), pj = (uint32_t)(idx / cam->width);
        float ray[6], out[4];
        cast_ray_f32(cam, pi, pj, ray);
        float nrm = sqrtf(fmaf(ray[3], ray[3], fmaf(ray[4], ray[4], ray[5] * ray[5])));
        if (nrm < 0.1f) {
            out_rgba8[idx] = 0;
            continue;
        }
        forward_ray_f32(&fm, ray, start_point, 0, NULL, out, NULL, NULL, NULL, 0, NULL, 0);
        out_rgba8[idx] = make_rgba8(out[0], out[1], out[2], 1.0f);
    }
    free(attr_f);
}

/* ---- float64 twin (validation of the maths; faces from exact point differences) ------- */

void rfo_trace_forward_f64(int sh_degree, rfo_settings settings, uint32_t num_points,
                           const double *points, const double *attributes, const uint32_t *adj,
                           const uint32_t *offsets, uint32_t num_rays, const double *rays,
                           const uint32_t *start, uint32_t nq, const double *quantiles,
                           double *rgba, double *qdepth, uint32_t *qidx,
                           uint32_t *num_intersections, double *contribution) {
    int A = 1 + 3 * (sh_degree + 1) * (sh_degree + 1);
    foam_t_f64 fm = {sh_degree, A, settings, num_points, points, attributes, adj, offsets, NULL};
    for (uint32_t r = 0; r < num_rays; ++r) {
        uint32_t n = forward_ray_f64(&fm, rays + 6 * (size_t)r, start[r], nq,
                                     quantiles ? quantiles + (size_t)r * nq : NULL,
                                     rgba + 4 * (size_t)r, qdepth ? qdepth + (size_t)r * nq : NULL,
                                     qidx ? qidx + (size_t)r * nq : NULL, contribution, 1, NULL, 0);
        if (num_intersections) num_intersections[r] = n;
    }
}

void rfo_trace_backward_f64(int sh_degree, rfo_settings settings, uint32_t num_points,
                            const double *points, const double *attributes, const uint32_t *adj,
                            const uint32_t *offsets, uint32_t num_rays, const double *rays,
                            const uint32_t *start, uint32_t nq, const double *quantiles,
                            const uint32_t *qidx, const double *rgba, const double *rgba_grad,
                            const double *depth_grad, const double *ray_error,
                            double *points_grad, double *attr_grad, double *point_error,
                            int strict) {
    int A = 1 + 3 * (sh_degree + 1) * (sh_degree + 1);
    foam_t_f64 fm = {sh_degree, A, settings, num_points, points, attributes, adj, offsets, NULL};
    memset(points_grad, 0, sizeof(double) * 3 * (size_t)num_points);
    memset(attr_grad, 0, sizeof(double) * (size_t)A * num_points);
    if (point_error) memset(point_error, 0, sizeof(double) * (size_t)num_points);
    sink_t_f64 sink = {points_grad, attr_grad, point_error, 0, NULL};
    for (uint32_t r = 0; r < num_rays; ++r) {
        backward_ray_f64(&fm, rays + 6 * (size_t)r, start[r], nq,
                         quantiles ? quantiles + (size_t)r * nq : NULL,
                         qidx ? qidx + (size_t)r * nq : NULL, rgba + 4 * (size_t)r,
                         rgba_grad + 4 * (size_t)r,
                         depth_grad ? depth_grad + (size_t)r * nq : NULL,
                         ray_error ? ray_error + r : NULL, &sink, strict);
    }
}
