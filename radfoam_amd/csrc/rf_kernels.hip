// rf_kernels.hip -- gfx950 (MI355X) kernels of the Voronoi-foam ray tracer + the C-ABI.
//
// One lane walks one ray.  A wave64 owns an 8x8 pixel tile when the rays form an image, so its
// lanes sit in the same few cells: their face-table and cell-record gathers collapse to a
// handful of cache lines, and face-count / step-count divergence inside the wave stays small.
// Blocks are handed to XCDs in contiguous chunks of the tile order so each XCD's private L2
// sees one band of the image (= one slab of the foam).
//
// Kernels (reference counterparts in src/tracing/pipeline.cu):
//   prepare_cells_kernel   prefetch_adjacent_diff_kernel :546-568  (+ cell-record packing)
//   repack_sh_kernel       (no counterpart: aligned SH rows)
//   forward_kernel         forward :14-130 and benchmark :472-544
//   backward_kernel        backward :132-343
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/radfoam_hip.h"
#include "rf_foam.hpp"
#include "rf_math.hpp"

namespace rf {

// ------------------------------------------------------------------------------------------
// views and parameters

struct FoamView {
    const RfCell *cells;
    const uint2 *diff;      // half4 entries (8 B)
    const uint32_t *adj;
    const void *sh;         // SH rows, sh_stride scalars apart
    uint32_t sh_stride;
    uint32_t diff_count;    // readable entries (only used by CLAMP instances)
};

struct RayGrid {
    uint32_t num_rays;
    uint32_t img_w, img_h;  // 0,0: flat list
};

struct FwdParams {
    FoamView foam;
    RayGrid grid;
    rf_trace_settings settings;
    const float *rays;
    const uint32_t *start;
    uint32_t nq;
    const float *quantiles;
    void *rgba;
    float *qdepth;
    uint32_t *qidx;
    uint32_t *nint;
    float *contribution;
    unsigned long long *stats;
    // benchmark
    rf_camera cam;
    float inv_tan_half_fov;
    uint32_t *rgba8;
};

struct BwdParams {
    FoamView foam;
    RayGrid grid;
    rf_trace_settings settings;
    const float *rays;
    const uint32_t *start;
    uint32_t nq;
    const float *quantiles;
    const uint32_t *qidx;
    const void *rgba;
    const void *rgba_grad;
    const float *depth_grad;
    const void *ray_error;
    float *points_grad;
    float *attr_grad;   // fp32 accumulator [N][A]
    float *point_error; // fp32 accumulator [N]
    uint32_t attr_dim;
};

// ------------------------------------------------------------------------------------------
// block -> tile, lane -> ray

// The dispatcher places block b on XCD b%8.  Give XCD x the x-th contiguous chunk of the
// logical block order instead of every 8th block (bijection for any grid size).
__device__ __forceinline__ uint32_t xcd_chunked_block(uint32_t b, uint32_t nb) {
    uint32_t x = b & 7u, local = b >> 3;
    uint32_t q = nb >> 3, r = nb & 7u;
    return x * q + (x < r ? x : r) + local;
}

__device__ __forceinline__ bool map_ray(const RayGrid &g, uint32_t &ray) {
    uint32_t blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    uint32_t tid = threadIdx.x;
    if (g.img_w) {
        uint32_t tiles_x = (g.img_w + 15u) >> 4;
        uint32_t ty = blk / tiles_x, tx = blk - ty * tiles_x;
        uint32_t wave = tid >> 6, lane = tid & 63u;
        uint32_t x = (tx << 4) + ((wave & 1u) << 3) + (lane & 7u);
        uint32_t y = (ty << 4) + ((wave >> 1) << 3) + (lane >> 3);
        ray = y * g.img_w + x;
        return x < g.img_w && y < g.img_h;
    }
    ray = blk * 256u + tid;
    return ray < g.num_rays;
}

inline uint32_t grid_blocks(const RayGrid &g) {
    if (g.img_w) return ((g.img_w + 15u) >> 4) * ((g.img_h + 15u) >> 4);
    return (g.num_rays + 255u) / 256u;
}


// One packed cell record, fetched as two naturally aligned 16-byte loads.
struct CellRec {
    float x, y, z, s;
    uint32_t begin, end;
};

__device__ __forceinline__ CellRec load_cell(const RfCell *cells, uint32_t i) {
    const uint4 *q = reinterpret_cast<const uint4 *>(cells) + 2 * (size_t)i;
    uint4 a = q[0];
    uint4 b = q[1];
    CellRec c;
    c.x = bits2f(a.x);
    c.y = bits2f(a.y);
    c.z = bits2f(a.z);
    c.s = bits2f(a.w);
    c.begin = b.x;
    c.end = b.y;
    return c;
}

// ------------------------------------------------------------------------------------------
// the per-cell face scan                         reference: trace<>, tracing_utils.cuh:27-67

struct __attribute__((aligned(8))) FacePair {
    uint32_t a0, a1, b0, b1;
};

// Nearest exit of the ray from the cell whose faces are entries [b,e) of the face table.
// Ascending order with a strict '<' (first minimum wins), like the reference.
template <bool CLAMP>
__device__ __forceinline__ void scan_cell(const FoamView &fv, uint32_t b, uint32_t e, float Px,
                                          float Py, float Pz, float Ox, float Oy, float Oz,
                                          float dx, float dy, float dz, float &t1, uint32_t &best) {
    t1 = __builtin_inff();
    best = kNone;
    for (uint32_t f = b; f < e; f += 2) {
        FacePair fp;
        if constexpr (CLAMP) {
            // caller-owned table without padding: never read entry diff_count
            uint32_t fc = (f + 1 < fv.diff_count) ? f : fv.diff_count - 2;
            fp = *reinterpret_cast<const FacePair *>(fv.diff + fc);
            if (fc != f) {
                fp.a0 = fp.b0;
                fp.a1 = fp.b1;
            }
        } else {
            fp = *reinterpret_cast<const FacePair *>(fv.diff + f);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t w0 = j ? fp.b0 : fp.a0, w1 = j ? fp.b1 : fp.a1;
            float ox = half_lo(w0), oy = half_hi(w0), oz = half_lo(w1);
            float dp = dot3(ox, oy, oz, dx, dy, dz);
            bool cand = (dp > 0.0f) && (f + j < e);
            // wave-uniform skip: back-facing planes need neither the numerator nor the divide
            if (__builtin_amdgcn_ballot_w64(cand) != 0ull) {
                float vx = fma_(ox, 0.5f, Px) - Ox;
                float vy = fma_(oy, 0.5f, Py) - Oy;
                float vz = fma_(oz, 0.5f, Pz) - Oz;
                float t = dot3(vx, vy, vz, ox, oy, oz) / dp;
                if (cand && t < t1) {
                    t1 = t;
                    best = f + j;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// colour of a cell for this ray     reference: load_sh_as_rgb, sh_utils.cuh:72-83 (+ :50-54)

template <int DEG, bool HALF>
__device__ __forceinline__ void cell_rgb(const FoamView &fv, uint32_t cell,
                                         const float (&sh)[sh_dim(DEG)], float &r, float &g,
                                         float &b) {
    constexpr int NC = 3 * sh_dim(DEG);
    constexpr int NV = (NC + 3) / 4;
    float c[NV * 4];
    if constexpr (HALF) {
        const uint2 *row = reinterpret_cast<const uint2 *>(
            reinterpret_cast<const uint16_t *>(fv.sh) + (size_t)cell * fv.sh_stride);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            uint2 w = row[i];
            c[4 * i + 0] = half_lo(w.x);
            c[4 * i + 1] = half_hi(w.x);
            c[4 * i + 2] = half_lo(w.y);
            c[4 * i + 3] = half_hi(w.y);
        }
    } else {
        const float4 *row = reinterpret_cast<const float4 *>(
            reinterpret_cast<const float *>(fv.sh) + (size_t)cell * fv.sh_stride);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float4 w = row[i];
            c[4 * i + 0] = w.x;
            c[4 * i + 1] = w.y;
            c[4 * i + 2] = w.z;
            c[4 * i + 3] = w.w;
        }
    }
    float acc[3] = {0.5f, 0.5f, 0.5f};
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i % 3] = fma_(sh[i / 3], c[i], acc[i % 3]);
    r = __builtin_fmaxf(acc[0], 0.0f);
    g = __builtin_fmaxf(acc[1], 0.0f);
    b = __builtin_fmaxf(acc[2], 0.0f);
}

// ------------------------------------------------------------------------------------------
// camera ray                                        reference: cast_ray, camera.h:56-85

__device__ __forceinline__ void cast_ray(const rf_camera &cam, float inv_tan, uint32_t i,
                                         uint32_t j, float &dx, float &dy, float &dz) {
    float aspect = (float)cam.width / (float)cam.height;
    float x = (float)i / (float)cam.width;
    float y = (float)j / (float)cam.height;
    float u = (2.0f * x - 1.0f) * aspect;
    float v = 1.0f - 2.0f * y;
    float mask = 1.0f;
    float d[3];
    if (cam.model == 0u) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            d[k] = fma_(v, cam.up[k], fma_(inv_tan, cam.forward[k], u * cam.right[k]));
    } else {
        float theta = atan2f(v, u);
        float phi = cam.fov * sqrtf(fma_(u, u, v * v));
        if (phi >= 3.14159265358979323846f) {
            phi = 3.14159265358979323846f - 1e-6f;
            mask = 0.0f;
        }
        float a = sinf(phi) * cosf(theta);
        float bb = sinf(phi) * sinf(theta);
        float c = cosf(phi);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            d[k] = fma_(c, cam.forward[k], fma_(a, cam.right[k], bb * cam.up[k]));
    }
    float n2 = dot3(d[0], d[1], d[2], d[0], d[1], d[2]);
    if (n2 > 0.0f) {
        float n = sqrtf(n2);
        d[0] = d[0] / n;
        d[1] = d[1] / n;
        d[2] = d[2] / n;
    }
    dx = d[0] * mask;
    dy = d[1] * mask;
    dz = d[2] * mask;
}

// make_rgba8, tracing_utils.cuh:105-115
__device__ __forceinline__ uint32_t make_rgba8(float r, float g, float b, float a) {
    r = __builtin_fmaxf(0.0f, __builtin_fminf(1.0f, r));
    g = __builtin_fmaxf(0.0f, __builtin_fminf(1.0f, g));
    b = __builtin_fmaxf(0.0f, __builtin_fminf(1.0f, b));
    a = __builtin_fmaxf(0.0f, __builtin_fminf(1.0f, a));
    uint32_t ri = (uint32_t)(int)(r * 255.0f), gi = (uint32_t)(int)(g * 255.0f);
    uint32_t bi = (uint32_t)(int)(b * 255.0f), ai = (uint32_t)(int)(a * 255.0f);
    return (ai << 24) | (bi << 16) | (gi << 8) | ri;
}

// ------------------------------------------------------------------------------------------
// forward / benchmark                     reference: pipeline.cu:14-130 and :472-544
//
// Control flow: every lane keeps an `alive` flag and the wave iterates while any lane is alive
// (no per-lane `break` out of nested conditionals).  hipcc 7.2 was observed to miscompile the
// natural `for(;;){...break...}` form of this loop (the next cell's face range was dropped on
// the path through the compositing block); the flag form also is what lane compaction needs.

template <int DEG, bool HALF, bool BENCH>
__global__ __launch_bounds__(256) void forward_kernel(FwdParams p) {
    uint32_t ray;
    bool alive = map_ray(p.grid, ray);
    const FoamView &fv = p.foam;

    float Ox = 0.0f, Oy = 0.0f, Oz = 0.0f, dx = 0.0f, dy = 0.0f, dz = 1.0f;
    uint32_t cur = 0;
    if (alive) {
        if constexpr (BENCH) {
            uint32_t pi = ray % p.cam.width, pj = ray / p.cam.width;
            Ox = p.cam.position[0];
            Oy = p.cam.position[1];
            Oz = p.cam.position[2];
            cast_ray(p.cam, p.inv_tan_half_fov, pi, pj, dx, dy, dz);
            if (sqrtf(dot3(dx, dy, dz, dx, dy, dz)) < 0.1f) {
                p.rgba8[ray] = 0u;
                alive = false;
            }
            cur = p.start[0];
        } else {
            const float *rp = p.rays + (size_t)ray * 6;
            Ox = rp[0];
            Oy = rp[1];
            Oz = rp[2];
            dx = rp[3];
            dy = rp[4];
            dz = rp[5];
            float nrm = sqrtf(dot3(dx, dy, dz, dx, dy, dz));
            dx = dx / nrm;
            dy = dy / nrm;
            dz = dz / nrm;
            cur = p.start[ray];
        }
    }
    const bool valid = alive;  // lanes that own a ray and must write outputs
    float sh[sh_dim(DEG)];
    sh_basis<DEG>(dx, dy, dz, sh);

    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    uint32_t qi = 0;
    const uint32_t nq = BENCH ? 0u : p.nq;
    const float *qp = nullptr;
    float cq = 0.0f;
    if (nq && alive) {
        qp = p.quantiles + (size_t)ray * nq;
        cq = qp[0];
    }
    const float thr = p.settings.weight_threshold;
    const uint32_t max_steps = p.settings.max_intersections;

    unsigned long long st_cells = 0, st_faces = 0, st_hops = 0, st_seg = 0, st_lit = 0;
    const bool want_stats = !BENCH && p.stats != nullptr;

    float t0 = 0.0f;
    uint32_t n = 0;
    CellRec head = load_cell(fv.cells, cur);
    while (__builtin_amdgcn_ballot_w64(alive) != 0ull) {
        if (alive) {
            n++;
            if (n > max_steps) alive = false;
        }
        float t1 = __builtin_inff();
        uint32_t best = kNone;
        if (alive) {
            scan_cell<BENCH>(fv, head.begin, head.end, head.x, head.y, head.z, Ox, Oy, Oz, dx, dy, dz, t1, best);
            if (want_stats) {
                st_cells++;
                st_faces += head.end - head.begin;
            }
            if (best == kNone) alive = false;
        }
        if (alive) {
            const uint32_t nxt = fv.adj[best];
            const CellRec nhead = load_cell(fv.cells, nxt);
            if (want_stats) st_hops++;
            if (t1 > t0) {
                float s = head.s;
                float r = 0.0f, g = 0.0f, b = 0.0f;
                if (s > 1e-6f) cell_rgb<DEG, HALF>(fv, cur, sh, r, g, b);
                if (want_stats) {
                    st_seg++;
                    st_lit += (s > 1e-6f) ? 1u : 0u;
                }
                float dt = __builtin_fmaxf(t1 - t0, 0.0f);
                float alpha = 1.0f - exp_(-s * dt);
                float w = T * alpha;
                if constexpr (!BENCH) {
                    if (p.contribution) unsafeAtomicAdd(p.contribution + cur, w);
                }
                Cr = fma_(w, r, Cr);
                Cg = fma_(w, g, Cg);
                Cb = fma_(w, b, Cb);
                float Tn = T * (1.0f - alpha);
                if constexpr (!BENCH) {
                    while (qi < nq && Tn < cq) {
                        p.qdepth[(size_t)ray * nq + qi] = t0 + log_(T / cq) / s;
                        p.qidx[(size_t)ray * nq + qi] = cur;
                        qi++;
                        if (qi < nq) cq = qp[qi];
                    }
                }
                T = Tn;
                if (!(T > thr)) alive = false;
            }
            t0 = __builtin_fmaxf(t0, t1);
            cur = nxt;
            head = nhead;
        }
    }

    if (!valid) return;
    if constexpr (BENCH) {
        p.rgba8[ray] = make_rgba8(Cr, Cg, Cb, 1.0f);
    } else {
        while (qi < nq) {
            p.qdepth[(size_t)ray * nq + qi] = -1.0f;
            p.qidx[(size_t)ray * nq + qi] = kNone;
            qi++;
        }
        float a = 1.0f - T;
        if constexpr (HALF) {
            uint32_t lo = (uint32_t)float_to_half_bits(Cr) | ((uint32_t)float_to_half_bits(Cg) << 16);
            uint32_t hi = (uint32_t)float_to_half_bits(Cb) | ((uint32_t)float_to_half_bits(a) << 16);
            reinterpret_cast<uint2 *>(p.rgba)[ray] = make_uint2(lo, hi);
        } else {
            reinterpret_cast<float4 *>(p.rgba)[ray] = make_float4(Cr, Cg, Cb, a);
        }
        if (p.nint) p.nint[ray] = n;
        if (want_stats) {
            atomicAdd(p.stats + 0, st_cells);
            atomicAdd(p.stats + 1, st_faces);
            atomicAdd(p.stats + 2, st_hops);
            atomicAdd(p.stats + 3, st_seg);
            atomicAdd(p.stats + 4, st_lit);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward                                           reference: pipeline.cu:132-343
// Re-walks the ray with the same control flow as forward and scatters gradients.  All the
// reference's behaviours are kept: the accumulators of the last visited
// cell and of its exit neighbour are never flushed; the first segment's dt0/dP term is taken
// against the world origin; dL/dt0 receives depth_grad/s.
// MODE 1: one atomic per lane per value (the reference's scatter).

template <bool HALF>
__device__ __forceinline__ float load_attr_scalar(const void *base, size_t i) {
    if constexpr (HALF) {
        return (float)__builtin_bit_cast(_Float16, reinterpret_cast<const uint16_t *>(base)[i]);
    } else {
        return reinterpret_cast<const float *>(base)[i];
    }
}

template <int DEG, bool HALF>
__global__ __launch_bounds__(256) void backward_kernel(BwdParams p) {
    uint32_t ray;
    bool alive = map_ray(p.grid, ray);
    const FoamView &fv = p.foam;
    constexpr int NB = sh_dim(DEG);
    constexpr int A = 1 + 3 * NB;

    float Ox = 0.0f, Oy = 0.0f, Oz = 0.0f, dx = 0.0f, dy = 0.0f, dz = 1.0f;
    float gr = 0.0f, gg = 0.0f, gb = 0.0f, ga = 0.0f, outr = 0.0f, outg = 0.0f, outb = 0.0f, outa = 0.0f;
    float err = 0.0f;
    uint32_t cur = 0;
    const uint32_t nq = p.nq;
    uint32_t qi = 0;
    const float *qp = nullptr;
    const float *dgp = nullptr;
    float cq = 0.0f, cdg = 0.0f;
    if (alive) {
        const float *rp = p.rays + (size_t)ray * 6;
        Ox = rp[0];
        Oy = rp[1];
        Oz = rp[2];
        dx = rp[3];
        dy = rp[4];
        dz = rp[5];
        float nrm = sqrtf(dot3(dx, dy, dz, dx, dy, dz));
        dx = dx / nrm;
        dy = dy / nrm;
        dz = dz / nrm;
        gr = load_attr_scalar<HALF>(p.rgba_grad, (size_t)ray * 4 + 0);
        gg = load_attr_scalar<HALF>(p.rgba_grad, (size_t)ray * 4 + 1);
        gb = load_attr_scalar<HALF>(p.rgba_grad, (size_t)ray * 4 + 2);
        ga = load_attr_scalar<HALF>(p.rgba_grad, (size_t)ray * 4 + 3);
        outr = load_attr_scalar<HALF>(p.rgba, (size_t)ray * 4 + 0);
        outg = load_attr_scalar<HALF>(p.rgba, (size_t)ray * 4 + 1);
        outb = load_attr_scalar<HALF>(p.rgba, (size_t)ray * 4 + 2);
        outa = load_attr_scalar<HALF>(p.rgba, (size_t)ray * 4 + 3);
        if (p.ray_error) err = load_attr_scalar<HALF>(p.ray_error, ray);
        cur = p.start[ray];
        if (nq) {
            qp = p.quantiles + (size_t)ray * nq;
            dgp = p.depth_grad + (size_t)ray * nq;
            cq = qp[0];
            for (uint32_t i = 0; i < nq; ++i) {
                uint32_t ci = p.qidx[(size_t)ray * nq + i];
                if (ci != kNone) {
                    float s = load_cell(fv.cells, ci).s;
                    cdg += dgp[i] / s;
                }
            }
        }
    }
    float sh[NB];
    sh_basis<DEG>(dx, dy, dz, sh);
    const float thr = p.settings.weight_threshold;
    const uint32_t max_steps = p.settings.max_intersections;

    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    uint32_t prev = kNone;
    float ppx = 0.0f, ppy = 0.0f, ppz = 0.0f;          // prev point
    float pgx = 0.0f, pgy = 0.0f, pgz = 0.0f;          // prev_point_grad
    float cgx = 0.0f, cgy = 0.0f, cgz = 0.0f;          // current_point_grad
    float ngx = 0.0f, ngy = 0.0f, ngz = 0.0f;          // next_point_grad

    float t0 = 0.0f;
    uint32_t n = 0;
    CellRec head = load_cell(fv.cells, cur);
    while (__builtin_amdgcn_ballot_w64(alive) != 0ull) {
        if (alive) {
            n++;
            if (n > max_steps) alive = false;
        }
        float t1 = __builtin_inff();
        uint32_t best = kNone;
        if (alive) {
            scan_cell<false>(fv, head.begin, head.end, head.x, head.y, head.z, Ox, Oy, Oz, dx, dy, dz, t1, best);
            if (best == kNone) alive = false;
        }
        if (alive) {
            const uint32_t nxt = fv.adj[best];
            const CellRec nhead = load_cell(fv.cells, nxt);
            if (t1 > t0) {
                float s = head.s;
                float r = 0.0f, g = 0.0f, b = 0.0f;
                if (s > 1e-6f) cell_rgb<DEG, HALF>(fv, cur, sh, r, g, b);
                float dt = __builtin_fmaxf(t1 - t0, 0.0f);
                float alpha = 1.0f - exp_(-s * dt);
                float w = T * alpha;
                float da_ds = dt * (1.0f - alpha);
                float da_ddt = (dt > 0.0f) ? s * (1.0f - alpha) : 0.0f;

                Cr = fma_(w, r, Cr);
                Cg = fma_(w, g, Cg);
                Cb = fma_(w, b, Cb);
                if (p.point_error) unsafeAtomicAdd(p.point_error + cur, w * err);

                float dLr = gr * w, dLg = gg * w, dLb = gb * w;
                float den = T * ((1.0f - alpha) + 1e-6f);
                float dfr = r - (outr - Cr) / den;
                float dfg = g - (outg - Cg) / den;
                float dfb = b - (outb - Cb) / den;
                float dL_da = T * dot3(dfr, dfg, dfb, gr, gg, gb);
                dL_da = dL_da + ((1.0f - outa) * ga) / ((1.0f - alpha) + 1e-6f);

                float dL_ds = dL_da * da_ds;
                float dL_ddt = dL_da * da_ddt;
                float dL_dt0 = 0.0f;

                float Tn = T * (1.0f - alpha);
                while (qi < nq && Tn < cq) {
                    float gi = dgp[qi] / s;
                    dL_dt0 = dL_dt0 + gi;
                    dL_ds = dL_ds + ((-gi) * log_(T / cq)) / s;
                    cdg = cdg - gi;
                    qi++;
                    if (qi < nq) cq = qp[qi];
                }
                if (qi < nq) {
                    dL_ds = fma_(-dt, cdg, dL_ds);
                    dL_ddt = fma_(-s, cdg, dL_ddt);
                }
                dL_dt0 = dL_dt0 + (-dL_ddt);
                float dL_dt1 = dL_ddt;

                float ax = 0.0f, ay = 0.0f, az = 0.0f;  // dt0_dprev
                if (prev != kNone)
                    bisector_grad(ppx, ppy, ppz, head.x, head.y, head.z, Ox, Oy, Oz, dx, dy, dz, ax, ay, az);
                float bx, by, bz;                        // dt1_dcurrent
                bisector_grad(head.x, head.y, head.z, nhead.x, nhead.y, nhead.z, Ox, Oy, Oz, dx, dy, dz, bx, by, bz);
                float ex, ey, ez;                        // dt0_dcurrent (vs prev, or vs origin on the first segment)
                bisector_grad(head.x, head.y, head.z, ppx, ppy, ppz, Ox, Oy, Oz, dx, dy, dz, ex, ey, ez);
                float fx, fy, fz;                        // dt1_dnext
                bisector_grad(nhead.x, nhead.y, nhead.z, head.x, head.y, head.z, Ox, Oy, Oz, dx, dy, dz, fx, fy, fz);

                pgx = fma_(dL_dt0, ax, pgx);
                pgy = fma_(dL_dt0, ay, pgy);
                pgz = fma_(dL_dt0, az, pgz);
                cgx = cgx + fma_(dL_dt0, ex, dL_dt1 * bx);
                cgy = cgy + fma_(dL_dt0, ey, dL_dt1 * by);
                cgz = cgz + fma_(dL_dt0, ez, dL_dt1 * bz);
                ngx = fma_(dL_dt1, fx, ngx);
                ngy = fma_(dL_dt1, fy, ngy);
                ngz = fma_(dL_dt1, fz, ngz);

                if (prev != kNone) {
                    float *pg = p.points_grad + 3 * (size_t)prev;
                    unsafeAtomicAdd(pg + 0, pgx);
                    unsafeAtomicAdd(pg + 1, pgy);
                    unsafeAtomicAdd(pg + 2, pgz);
                }
                ppx = head.x;
                ppy = head.y;
                ppz = head.z;
                prev = cur;
                pgx = cgx;
                pgy = cgy;
                pgz = cgz;
                cgx = ngx;
                cgy = ngy;
                cgz = ngz;
                ngx = ngy = ngz = 0.0f;
                T = Tn;

                if (r == 0.0f) dLr = 0.0f;
                if (g == 0.0f) dLg = 0.0f;
                if (b == 0.0f) dLb = 0.0f;
                float *row = p.attr_grad + (size_t)cur * A;
                // all-zero colour gradients (empty cells, clamped channels) add nothing: skip them
                if (dLr != 0.0f || dLg != 0.0f || dLb != 0.0f) {
#pragma unroll
                    for (int i = 0; i < 3 * NB; ++i) {
                        float gc = (i % 3 == 0) ? dLr : ((i % 3 == 1) ? dLg : dLb);
                        unsafeAtomicAdd(row + i, sh[i / 3] * gc);
                    }
                }
                unsafeAtomicAdd(row + (A - 1), dL_ds);

                if (!(T > thr)) alive = false;
            }
            t0 = __builtin_fmaxf(t0, t1);
            cur = nxt;
            head = nhead;
        }
    }
}

// ------------------------------------------------------------------------------------------
// foam packing

template <bool HALF>
__global__ __launch_bounds__(256) void prepare_cells_kernel(
    const float *__restrict__ points, const void *__restrict__ attributes, uint32_t attr_dim,
    uint32_t num_points, const uint32_t *__restrict__ adj, const uint32_t *__restrict__ offsets,
    RfCell *__restrict__ cells, uint2 *__restrict__ diff, int write_cells, int write_diff) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_points) return;
    float px = points[3 * (size_t)i], py = points[3 * (size_t)i + 1], pz = points[3 * (size_t)i + 2];
    uint32_t b = offsets[i], e = offsets[i + 1];
    if (write_cells) {
        float s = load_attr_scalar<HALF>(attributes, (size_t)i * attr_dim + attr_dim - 1);
        float4 *c4 = reinterpret_cast<float4 *>(cells + i);
        c4[0] = make_float4(px, py, pz, s);
        reinterpret_cast<uint4 *>(c4)[1] = make_uint4(b, e, 0u, 0u);
    }
    if (write_diff) {
        for (uint32_t f = b; f < e; ++f) {
            uint32_t q = adj[f];
            float qx = points[3 * (size_t)q], qy = points[3 * (size_t)q + 1], qz = points[3 * (size_t)q + 2];
            uint32_t lo = (uint32_t)float_to_half_bits(qx - px) | ((uint32_t)float_to_half_bits(qy - py) << 16);
            uint32_t hi = (uint32_t)float_to_half_bits(qz - pz);
            diff[f] = make_uint2(lo, hi);
        }
    }
}

// Copies the 3B colour coefficients of every row into 16-B aligned rows of `stride` scalars.
template <typename T>
__global__ __launch_bounds__(256) void repack_sh_kernel(const T *__restrict__ attributes,
                                                        uint32_t attr_dim, uint32_t num_points,
                                                        uint32_t stride, T *__restrict__ out) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)num_points * stride;
    if (idx >= total) return;
    uint32_t row = (uint32_t)(idx / stride), col = (uint32_t)(idx % stride);
    out[idx] = (col < attr_dim - 1) ? attributes[(size_t)row * attr_dim + col] : T(0);
}

// ------------------------------------------------------------------------------------------
// host side

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *detail = "") {
    std::snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

static int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return RF_ERR_LAUNCH;
    }
    return RF_OK;
}

static bool valid_instance(int sh_degree, int attr_type) {
    return sh_degree >= 0 && sh_degree <= 3 && (attr_type == RF_ATTR_FLOAT32 || attr_type == RF_ATTR_FLOAT16);
}

static FoamView make_view(const FoamLayout &L, void *ws, const void *attributes, const uint32_t *adj,
                          uint32_t adj_size) {
    FoamView v;
    char *base = static_cast<char *>(ws);
    v.cells = reinterpret_cast<const RfCell *>(base + L.cells_off);
    v.diff = reinterpret_cast<const uint2 *>(base + L.diff_off);
    v.adj = adj;
    v.sh = L.sh_repacked ? static_cast<const void *>(base + L.sh_off) : attributes;
    v.sh_stride = L.sh_stride;
    v.diff_count = adj_size + kDiffPad;
    return v;
}

static int prepare_impl(int sh_degree, int attr_type, uint32_t num_points, const float *points,
                        const void *attributes, uint32_t adj_size, const uint32_t *adj,
                        const uint32_t *offsets, void *ws, size_t ws_bytes, bool write_diff,
                        hipStream_t stream) {
    const bool half = attr_type == RF_ATTR_FLOAT16;
    FoamLayout L = foam_layout(num_points, adj_size, sh_degree, half);
    if (!ws || ws_bytes < L.total) return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_workspace_bytes()");
    if (num_points == 0) return RF_OK;
    char *base = static_cast<char *>(ws);
    RfCell *cells = reinterpret_cast<RfCell *>(base + L.cells_off);
    uint2 *diff = reinterpret_cast<uint2 *>(base + L.diff_off);
    const uint32_t A = attribute_dim(sh_degree);
    dim3 grid((num_points + 255u) / 256u), block(256);
    if (half)
        hipLaunchKernelGGL(prepare_cells_kernel<true>, grid, block, 0, stream, points, attributes, A,
                           num_points, adj, offsets, cells, diff, 1, write_diff ? 1 : 0);
    else
        hipLaunchKernelGGL(prepare_cells_kernel<false>, grid, block, 0, stream, points, attributes, A,
                           num_points, adj, offsets, cells, diff, 1, write_diff ? 1 : 0);
    if (write_diff) {
        // zero the padding so over-reads past the last cell see finite values
        hipMemsetAsync(diff + adj_size, 0, (size_t)kDiffPad * 8, stream);
    }
    if (L.sh_repacked) {
        size_t total = (size_t)num_points * L.sh_stride;
        dim3 g2((unsigned)((total + 255) / 256));
        if (half)
            hipLaunchKernelGGL(repack_sh_kernel<uint16_t>, g2, block, 0, stream,
                               static_cast<const uint16_t *>(attributes), A, num_points, L.sh_stride,
                               reinterpret_cast<uint16_t *>(base + L.sh_off));
        else
            hipLaunchKernelGGL(repack_sh_kernel<float>, g2, block, 0, stream,
                               static_cast<const float *>(attributes), A, num_points, L.sh_stride,
                               reinterpret_cast<float *>(base + L.sh_off));
    }
    return check_launch("rf_prepare_foam");
}

template <template <int, bool> class Launcher, typename... Args>
static int dispatch(int sh_degree, bool half, Args &&...args) {
    switch (sh_degree * 2 + (half ? 1 : 0)) {
    case 0: return Launcher<0, false>::run(args...);
    case 1: return Launcher<0, true>::run(args...);
    case 2: return Launcher<1, false>::run(args...);
    case 3: return Launcher<1, true>::run(args...);
    case 4: return Launcher<2, false>::run(args...);
    case 5: return Launcher<2, true>::run(args...);
    case 6: return Launcher<3, false>::run(args...);
    default: return Launcher<3, true>::run(args...);
    }
}

template <int DEG, bool HALF>
struct LaunchForward {
    static int run(const FwdParams &p, bool bench, hipStream_t stream) {
        uint32_t nb = grid_blocks(p.grid);
        if (nb == 0) return RF_OK;
        if (bench)
            hipLaunchKernelGGL((forward_kernel<DEG, HALF, true>), dim3(nb), dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL((forward_kernel<DEG, HALF, false>), dim3(nb), dim3(256), 0, stream, p);
        return check_launch(bench ? "rf_trace_benchmark" : "rf_trace_forward");
    }
};

template <int DEG, bool HALF>
struct LaunchBackward {
    static int run(const BwdParams &p, hipStream_t stream) {
        uint32_t nb = grid_blocks(p.grid);
        if (nb == 0) return RF_OK;
        hipLaunchKernelGGL((backward_kernel<DEG, HALF>), dim3(nb), dim3(256), 0, stream, p);
        return check_launch("rf_trace_backward");
    }
};

static RayGrid make_grid(uint32_t num_rays, const rf_launch_opts *opts) {
    RayGrid g{num_rays, 0u, 0u};
    if (opts && opts->image_width && opts->image_height &&
        (uint64_t)opts->image_width * opts->image_height == num_rays) {
        g.img_w = opts->image_width;
        g.img_h = opts->image_height;
    }
    return g;
}

}  // namespace rf

// ------------------------------------------------------------------------------------------
// C-ABI

using namespace rf;

extern "C" {

const char *rf_last_error(void) { return g_err; }

uint32_t rf_attribute_dim(int sh_degree) { return attribute_dim(sh_degree); }

size_t rf_workspace_bytes(uint32_t num_points, uint32_t point_adjacency_size, int sh_degree,
                          int attr_type) {
    if (!valid_instance(sh_degree, attr_type)) return 0;
    return foam_layout(num_points, point_adjacency_size, sh_degree, attr_type == RF_ATTR_FLOAT16).total;
}

int rf_build_adjacent_diff(const float *points, uint32_t num_points, uint32_t point_adjacency_size,
                           const uint32_t *point_adjacency, const uint32_t *point_adjacency_offsets,
                           void *adjacent_diff, void *stream) {
    g_err[0] = 0;
    (void)point_adjacency_size;
    if (num_points == 0) return RF_OK;
    if (!points || !point_adjacency || !point_adjacency_offsets || !adjacent_diff)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_build_adjacent_diff: null pointer");
    hipLaunchKernelGGL(prepare_cells_kernel<false>, dim3((num_points + 255u) / 256u), dim3(256), 0,
                       static_cast<hipStream_t>(stream), points, static_cast<const void *>(nullptr), 0u,
                       num_points, point_adjacency, point_adjacency_offsets,
                       static_cast<RfCell *>(nullptr), static_cast<uint2 *>(adjacent_diff), 0, 1);
    return check_launch("rf_build_adjacent_diff");
}

int rf_prepare_foam(int sh_degree, int attr_type, uint32_t num_points, const float *points,
                    const void *attributes, uint32_t point_adjacency_size,
                    const uint32_t *point_adjacency, const uint32_t *point_adjacency_offsets,
                    void *workspace, size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (!valid_instance(sh_degree, attr_type))
        return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported SH degree or attribute type");
    if (num_points && (!points || !attributes || !point_adjacency || !point_adjacency_offsets))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_prepare_foam: null pointer");
    return prepare_impl(sh_degree, attr_type, num_points, points, attributes, point_adjacency_size,
                        point_adjacency, point_adjacency_offsets, workspace, workspace_bytes, true,
                        static_cast<hipStream_t>(stream));
}

int rf_trace_forward(int sh_degree, int attr_type, const rf_trace_settings *settings,
                     uint32_t num_points, const float *points, const void *attributes,
                     uint32_t point_adjacency_size, const uint32_t *point_adjacency,
                     const uint32_t *point_adjacency_offsets, uint32_t num_rays, const float *rays,
                     const uint32_t *start_point_index, uint32_t num_depth_quantiles,
                     const float *depth_quantiles, void *ray_rgba, float *quantile_depths,
                     uint32_t *quantile_point_indices, uint32_t *num_intersections,
                     void *point_contribution, const rf_launch_opts *opts, void *stream) {
    g_err[0] = 0;
    if (!valid_instance(sh_degree, attr_type))
        return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported SH degree or attribute type");
    if (!settings || !opts) return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_forward: settings/opts null");
    if (num_rays == 0) return RF_OK;
    if (!points || !attributes || !point_adjacency || !point_adjacency_offsets || !rays ||
        !start_point_index || !ray_rgba)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_forward: null pointer");
    if (num_depth_quantiles && (!depth_quantiles || !quantile_depths || !quantile_point_indices))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_forward: depth quantile buffers missing");
    const bool half = attr_type == RF_ATTR_FLOAT16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    FoamLayout L = foam_layout(num_points, point_adjacency_size, sh_degree, half);
    if (!opts->workspace || opts->workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_workspace_bytes()");
    if (!opts->foam_prepared) {
        int rc = prepare_impl(sh_degree, attr_type, num_points, points, attributes, point_adjacency_size,
                              point_adjacency, point_adjacency_offsets, opts->workspace,
                              opts->workspace_bytes, true, s);
        if (rc != RF_OK) return rc;
    }
    FwdParams p{};
    p.foam = make_view(L, opts->workspace, attributes, point_adjacency, point_adjacency_size);
    p.grid = make_grid(num_rays, opts);
    p.settings = *settings;
    p.rays = rays;
    p.start = start_point_index;
    p.nq = depth_quantiles ? num_depth_quantiles : 0u;
    p.quantiles = depth_quantiles;
    p.rgba = ray_rgba;
    p.qdepth = quantile_depths;
    p.qidx = quantile_point_indices;
    p.nint = num_intersections;
    p.contribution = static_cast<float *>(point_contribution);
    p.stats = reinterpret_cast<unsigned long long *>(opts->stats);
    return dispatch<LaunchForward>(sh_degree, half, p, false, s);
}

int rf_trace_backward(int sh_degree, int attr_type, const rf_trace_settings *settings,
                      uint32_t num_points, const float *points, const void *attributes,
                      uint32_t point_adjacency_size, const uint32_t *point_adjacency,
                      const uint32_t *point_adjacency_offsets, uint32_t num_rays, const float *rays,
                      const uint32_t *start_point_index, uint32_t num_depth_quantiles,
                      const float *depth_quantiles, const uint32_t *quantile_point_indices,
                      const void *ray_rgba, const void *ray_rgba_grad, const float *depth_grad,
                      const void *ray_error, float *ray_grad, float *points_grad, void *attribute_grad,
                      void *point_error, const rf_launch_opts *opts, void *stream) {
    g_err[0] = 0;
    (void)ray_grad;  // never written, as in the reference
    if (!valid_instance(sh_degree, attr_type))
        return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported SH degree or attribute type");
    if (!settings || !opts) return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_backward: settings/opts null");
    if (num_rays == 0) return RF_OK;
    if (!points || !attributes || !point_adjacency || !point_adjacency_offsets || !rays ||
        !start_point_index || !ray_rgba || !ray_rgba_grad || !points_grad || !attribute_grad)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_backward: null pointer");
    if (num_depth_quantiles && depth_quantiles && (!quantile_point_indices || !depth_grad))
        return fail(RF_ERR_INVALID_ARGUMENT, "depth_grad must be provided if depth_quantiles is provided");
    const bool half = attr_type == RF_ATTR_FLOAT16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    FoamLayout L = foam_layout(num_points, point_adjacency_size, sh_degree, half);
    if (!opts->workspace || opts->workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_workspace_bytes()");
    if (!opts->foam_prepared) {
        int rc = prepare_impl(sh_degree, attr_type, num_points, points, attributes, point_adjacency_size,
                              point_adjacency, point_adjacency_offsets, opts->workspace,
                              opts->workspace_bytes, true, s);
        if (rc != RF_OK) return rc;
    }
    BwdParams p{};
    p.foam = make_view(L, opts->workspace, attributes, point_adjacency, point_adjacency_size);
    p.grid = make_grid(num_rays, opts);
    p.settings = *settings;
    p.rays = rays;
    p.start = start_point_index;
    p.nq = depth_quantiles ? num_depth_quantiles : 0u;
    p.quantiles = depth_quantiles;
    p.qidx = quantile_point_indices;
    p.rgba = ray_rgba;
    p.rgba_grad = ray_rgba_grad;
    p.depth_grad = depth_grad;
    p.ray_error = ray_error;
    p.points_grad = points_grad;
    p.attr_grad = static_cast<float *>(attribute_grad);
    p.point_error = static_cast<float *>(point_error);
    p.attr_dim = attribute_dim(sh_degree);
    return dispatch<LaunchBackward>(sh_degree, half, p, s);
}

int rf_trace_benchmark(int sh_degree, int attr_type, const rf_trace_settings *settings,
                       uint32_t num_points, const float *points, const void *attributes,
                       uint32_t point_adjacency_size, const uint32_t *point_adjacency,
                       const uint32_t *point_adjacency_offsets, const void *adjacent_diff,
                       const rf_camera *camera, const uint32_t *start_point_index,
                       uint32_t *ray_rgba, const rf_launch_opts *opts, void *stream) {
    g_err[0] = 0;
    if (!valid_instance(sh_degree, attr_type))
        return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported SH degree or attribute type");
    if (!settings || !opts || !camera)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_benchmark: settings/opts/camera null");
    if (camera->model > 1u) return fail(RF_ERR_INVALID_ARGUMENT, "Invalid camera model");
    if (camera->width == 0 || camera->height == 0) return RF_OK;
    if (!points || !attributes || !point_adjacency || !point_adjacency_offsets || !adjacent_diff ||
        !start_point_index || !ray_rgba)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_benchmark: null pointer");
    if (point_adjacency_size < 2)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_benchmark: adjacency too small");
    const bool half = attr_type == RF_ATTR_FLOAT16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    FoamLayout L = foam_layout(num_points, point_adjacency_size, sh_degree, half);
    if (!opts->workspace || opts->workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_workspace_bytes()");
    if (!opts->foam_prepared) {
        // the face table is the caller's: only cell records / SH rows are packed
        int rc = prepare_impl(sh_degree, attr_type, num_points, points, attributes, point_adjacency_size,
                              point_adjacency, point_adjacency_offsets, opts->workspace,
                              opts->workspace_bytes, false, s);
        if (rc != RF_OK) return rc;
    }
    FwdParams p{};
    p.foam = make_view(L, opts->workspace, attributes, point_adjacency, point_adjacency_size);
    p.foam.diff = static_cast<const uint2 *>(adjacent_diff);
    p.foam.diff_count = point_adjacency_size;
    p.grid = RayGrid{camera->width * camera->height, camera->width, camera->height};
    p.settings = *settings;
    p.start = start_point_index;
    p.cam = *camera;
    p.inv_tan_half_fov = 1.0f / tanf(camera->fov * 0.5f);
    p.rgba8 = ray_rgba;
    return dispatch<LaunchForward>(sh_degree, half, p, true, s);
}

}  // extern "C"
