// rf_kernels.hip -- gfx950 (MI355X) kernels of the Voronoi-foam ray tracer + the C-ABI.
//
// One lane walks one ray; a wave64 owns an 8x8 pixel tile when the rays form an image, so its
// lanes sit in neighbouring cells and their gathers hit the same few cache lines.  Per step a lane
//   1. scans its cell's faces for the nearest exit: four faces per iteration (a 24-byte block of
//      planar fp16 offsets, lists padded to a multiple of four), branch-free, two faces per VALU
//      slot in packed fp32;
//   2. reads the link of the winning face: it names the next cell AND its face range, so the next
//      cell record, its face list and the colour row are requested together -- one dependent
//      round trip per hop instead of the reference's four;
//   3. composites the segment (forward) / accumulates gradients (backward).
// Forward records the cell every hop enters (the "trail"); backward replays it instead of scanning
// again (it reads no face table at all), and sums gradient rows in a block-level LDS write-combining cache before they go to
// global atomics.  Tiles are dealt to the XCDs in half-row strips, round-robin, so each XCD's
// private L2 sees whole strips of the image while all XCDs get the same mix of cheap and
// expensive image regions.
//
// Tried and measured out: staging the wave's distinct cells' face lists in
// LDS (global_load_lds DMA + ballot/readlane dedupe).  The lanes of a tile sit in 16-24
// different cells at any step, so the dedupe loop cost more than the L1-served gathers it
// replaced (forward 10.8 ms staged vs 8.8 ms direct on the 2M-point foam).
//
// Kernels (reference counterparts in src/tracing/pipeline.cu):
//   padded_*_kernel        (no counterpart: prefix sum of the face counts rounded up to 4)
//   prepare_foam_kernel    prefetch_adjacent_diff_kernel :546-568  (+ cell/face/link packing)
//   prepare_geometry_kernel  the same, points-dependent part only (adjacency unchanged)
//   adjacent_diff_kernel   prefetch_adjacent_diff_kernel :546-568  (plain half4 table)
//   repack_sh_kernel       (no counterpart: aligned SH rows for fp16 attributes of odd pitch)
//   forward_kernel         forward :14-130 and benchmark :472-544 (EAGER instances: the first six face blocks of
//                          a cell requested at the hop that enters it -- flat batches and small launches)
//   backward_kernel        backward :132-343 (re-walk); backward_replay_kernel (modes 1, 2),
//                          backward_replay_cached_kernel (mode 3, image-shaped batches) and
//                          backward_replay_direct_kernel (mode 4, flat batches): the same functor over
//                          the recorded trail
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/radfoam_hip.h"
#include "rf_foam.hpp"
#include "rf_host.hpp"
#include "rf_math.hpp"
#include "rf_tiles.hpp"
#include "rf_wave.hpp"

namespace rf {

// ------------------------------------------------------------------------------------------
// views and parameters

struct FoamView {
    const float4 *cells;        // {x, y, z, density}
    const uint16_t *geo;        // blocks of 4 faces: half x[4] y[4] z[4], offsets to the neighbours (rf_foam.hpp)
    const Link *link;           // per face: {neighbour index, its first face, its padded face count}
    const uint32_t *poff;       // [N+1] first (padded) face entry per cell
    const uint32_t *offsets;    // caller's CSR offsets (walk statistics only)
    const void *sh;             // SH rows, sh_stride scalars apart
    uint32_t sh_stride;
};

struct RayGrid {
    uint32_t num_rays;
    uint32_t img_w, img_h;  // 0,0: flat list
    const uint32_t *order;  // flat list only, optional: thread slot s traces ray order[s] (rf_build_ray_order)
    const uint32_t *tile_order;  // images only, optional: block b walks tile tile_order[b] instead of dealt_tile(b)
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
    uint8_t *visit_marks;   // statistics instance only, optional: [N] set to 1 for every cell scanned
    uint32_t *tile_cost;    // optional: [tiles] maximum over the launch of the steps of a tile's longest ray
    uint32_t *trail;        // [trail_cap][trail_slots] cell entered by each hop (optional)
    uint32_t *trail_hops;   // [trail_slots] hops taken by the ray of each thread slot
    uint32_t trail_cap, trail_slots;
    uint32_t *queue;        // forward_mode 4: one device uint32 (the persistent waves' ray queue head), in the workspace
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
    const uint32_t *trail;
    const uint32_t *trail_hops;
    uint32_t trail_cap, trail_slots;
    uint32_t attr_pitch;         // floats between two rows of attr_grad (>= A; rf_launch_opts.attr_grad_pitch)
    unsigned long long *stats;   // optional scatter counters (experiments): [0] row flushes [1] values flushed
                                 // [2] lane contributions that bypassed the block cache [3] cached lane contributions
};

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------
// block -> tile, lane -> ray

// The dispatcher places block b on XCD b%8, and every XCD has a private L2.  Tiles are dealt to
// the XCDs in chunks of `chunk` consecutive tiles of the row-major tile order (a quarter of an image
// row of tiles), a round of 8 chunks at a time.  A chunk is a contiguous strip of the image (one slab
// of the foam for that XCD's L2); dealing them gives every XCD strips from all over the image, so
// that the XCDs finish together even though rays through different image regions walk very
// different numbers of cells (measured: contiguous eighths of the image finish between 5.8 and
// 7.7 ms).  Which chunk of a round an XCD takes rotates from round to round: with a fixed assignment
// and k chunks per image row, XCD x would only ever see the same (x mod k)-th part of every row
// (measured with 16x4-tile blocks, 8 per row: one XCD per image column, 7.7 ms instead of 5.6).
// The launch is padded to whole rounds of 8 chunks; blocks whose tile falls past the end own no rays.
// That is the assignment when nothing is known about the frame.  Which tiles are still running when the launch drains
// decides its last half millisecond (a launch ends a longest-ray's chain of dependent hops after its last block
// starts), and the frame itself says which tiles are short: the forward records the step count of every tile's longest
// ray (rf_launch_opts.tile_cost) and the host turns that into an assignment for the next launches over the frame
// (rf_launch_opts.tile_order; radfoam_amd/pipeline.py: tile_order) -- every XCD keeps the tiles dealt to it here but takes
// them longest first, or keeps this order and moves the cheapest tiles to the end (1080p frame: forward 4.70 -> 4.42 ms,
// backward 3.88 -> 3.73; the render path 2.51 -> 2.26 ms).
// (dealt_tile / tile counts: rf_tiles.hpp, shared with the device-side builder of tile orders in rf_tile_prior.hip)
inline __host__ __device__ uint32_t num_tiles(const RayGrid &g) { return tile_count(g.num_rays, g.img_w, g.img_h); }

inline __host__ __device__ uint32_t tile_chunk(const RayGrid &g) { return tile_chunk_of(g.img_w); }

constexpr uint32_t kResidentBlocks = 1024;   // 256 CUs x 4 blocks: a launch of the eager forward (4 waves per SIMD) that is resident at once
constexpr int kEagerBlocks = 6;

// blocks to launch: whole rounds of 8 chunks
inline uint32_t launch_blocks(const RayGrid &g) { return launch_block_count(g.num_rays, g.img_w, g.img_h); }

// ray and trail slot of thread `tid` of block `block` of a launch of `nblocks` blocks; false when it owns no ray (slot ==
// kNone: not even a slot)
__device__ __forceinline__ bool map_slot(const RayGrid &g, uint32_t block, uint32_t tid, uint32_t nblocks, uint32_t &ray,
                                         uint32_t &slot) {
    const uint32_t chunk = tile_chunk(g);
    const uint32_t tile = g.tile_order ? g.tile_order[block] : dealt_tile(block, chunk, nblocks / (8u * chunk));
    ray = 0;
    slot = kNone;
    if (tile >= num_tiles(g)) return false;
    slot = tile * (uint32_t)kBlock + tid;
    if (g.img_w) {
        uint32_t tiles_x = (g.img_w + 15u) >> 4;
        uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        uint32_t wave = tid >> 6, lane = tid & 63u;
        // Z-order inside the wave's 8x8 tile: the 4 / 16 lanes whose addresses the texture path
        // processes together are a 2x2 / 4x4 pixel block (fewest distinct cells, i.e. cache lines)
        uint32_t lx = (lane & 1u) | ((lane >> 1) & 2u) | ((lane >> 2) & 4u);
        uint32_t ly = ((lane >> 1) & 1u) | ((lane >> 2) & 2u) | ((lane >> 3) & 4u);
        uint32_t x = (tx << 4) + ((wave & 1u) << 3) + lx;
        uint32_t y = (ty << 4) + ((wave >> 1) << 3) + ly;
        ray = y * g.img_w + x;
        return x < g.img_w && y < g.img_h;
    }
    if (slot >= g.num_rays) return false;
    ray = g.order ? g.order[slot] : slot;
    return true;
}

__device__ __forceinline__ bool map_ray(const RayGrid &g, uint32_t &ray, uint32_t &slot) {
    return map_slot(g, blockIdx.x, threadIdx.x, gridDim.x, ray, slot);
}

// ------------------------------------------------------------------------------------------
// the per-cell face scan                         reference: trace<>, tracing_utils.cuh:27-67

struct ScanResult {
    float t1;
    uint32_t k;     // winning face, relative to the cell's first face; kNone if no exit
};

// t of the ray/bisector hit for one face with offset o, as tracing_utils.cuh:57-60 forms it (and as the scans below do):
// what the trail replay recomputes for the face a hop crossed -- the same float the forward's scan produced
__device__ __forceinline__ float face_hit(float ox, float oy, float oz, float Px, float Py, float Pz,
                                          float Ox, float Oy, float Oz, float dx, float dy, float dz) {
    const float dp = dot3(ox, oy, oz, dx, dy, dz);
    const float vx = fma_(ox, 0.5f, Px) - Ox;
    const float vy = fma_(oy, 0.5f, Py) - Oy;
    const float vz = fma_(oz, 0.5f, Pz) - Oz;
    return dot3(vx, vy, vz, ox, oy, oz) / dp;
}

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

struct __attribute__((aligned(8))) GeoXY {
    uint32_t x01, x23, y01, y23;
};
struct __attribute__((aligned(8))) GeoZ {
    uint32_t z01, z23;
};

// ---- the face scan ----------------------------------------------------------------------------------------------
// WHAT is computed is the reference's scan, tracing_utils.cuh:43-67, to the bit: per face o (fp16 -> fp32)
//     dp = o.d      v = (P + o/2) - O      t = (v.o) / dp  (IEEE divide)      take if dp > 0 && t < t_1
// i.e. the exit is the face with the smallest ROUNDED quotient among those with dp > 0, the lowest index among equal
// quotients, and t_1 is that rounded quotient.  This is a total preorder on the faces, so the result does not depend on
// the order in which they are examined -- which is what lets the kernels find it without dividing every face:
//
//   filter   two faces a, b with dp > 0 are compared by cross-multiplication, p1 = RN(num_a * dp_b) against
//            p2 = RN(num_b * dp_a) (one packed operation); num and dp are the reference's own floats (same association,
//            same dot-product order as scan_faces_strict below).  Let D = |bits(p1) - bits(p2)|, the number of floats
//            from one product to the other.  If D >= 4 the order of p1 and p2 is the STRICT order of the two rounded
//            quotients RN(num_a/dp_a), RN(num_b/dp_b):  |p1 - p2| >= D ulp(p) (the smaller ulp when they straddle a
//            binade), each product is within half an ulp of its exact value, so the exact products differ by at least
//            (D - 1.5) ulp(p) >= 2.5 * 2^-24 relative; the exact quotients num_a/dp_a and num_b/dp_b differ by the same
//            relative amount (divide both products by dp_a dp_b), which is more than one ulp of a quotient (at most
//            2^-23 relative), and two reals more than an ulp apart round to different floats in their own order.
//            (Zero and subnormal products: same argument with absolute spacings, 2^-149 -- relatively coarser, so the
//            bound holds a fortiori.  Opposite signs: D >= 2^23.  The one assumption: the two QUOTIENTS are normal floats,
//            whose ulp is at most 2^-23 relative; it is not left to the caller (include/radfoam_hip.h states it) but
//            checked: scan_end sends a cell whose winning quotient has a zero exponent field to the dividing scan.)
//            A pairwise tournament over such comparisons (the nearer face of a pair, then
//            against the running best; earlier faces win ties) therefore returns the reference's face whenever NO
//            comparison it made had D <= kTieUlps = 3;
//   certify  the scan keeps the minimum D over all its comparisons (v_sad_u32 + v_min3_u32: three instructions per pair
//            of faces).  Products are formed as fma(x, y, +0), so that a zero product is always +0 and two zero
//            products are 0 apart;
//   resolve  a cell whose minimum is <= kTieUlps is scanned again by scan_faces_strict, which divides every face as the
//            reference does.  How often: the exits of a cell differ by (cell size / distance from the ray origin)
//            relative, 1/230 on the 2 M-point benchmark foam, so two of them fall within +-3 floats of each other in about
//            4 cell scans of 10^4 (measured by the CPU checker's mirror of this evaluation, the test-suite: 1.3e-4
//            at 60 k points).  Only the winner of an uncontested scan is divided, once.
//
// Face lists are padded to a multiple of four entries with copies of the block's first real face at 2x, 4x, 8x its
// offset (pad_offset, prepare kernels): a bisector twice / four / eight times as far along the same normal has the same
// sign of dp and a quotient that is never smaller (every rounding on the way is monotone), and a higher index, so a
// padding entry never wins -- under the reference's rule itself, no special case in any scan.  (An all-zero entry, used
// when the doubled offset would overflow fp16, has dp = 0 and is no candidate either; it merely sends the cell to the
// dividing scan, since two zero products are 0 apart.)
//
// Four faces per iteration, branch-free, two at a time in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32); a block arrives as
// one 16-byte and one 8-byte load (blocks are 8-byte aligned, which this hardware serves -- scripts/probe/unaligned.hip).
// The winner is tracked relative to the block being scanned (`rel`, inline constants 0..3) and rebased once per
// iteration, instead of materialising k+j per face.
// Measured out for whole frames in round 3 (profiles/r03/c_ab_scan_pipe_*; the code is in commit 488b7a4): requesting
// the first block of the next cell at hop time, from the link just read (forward 4.75 ms against 4.67), and on top of
// that software-pipelining the block loop with inline-asm loads and an explicit s_waitcnt (4.83 ms: a whole frame is
// issue-bound, six waves per SIMD already cover the latency).  Flat batches and small launches are not: scan_faces_eager.
// History: rounds 1-4 shipped a different evaluation as the default ("canonical": v = (P - O) + o/2, the
// cross-multiplied tournament WITHOUT the certificate, i.e. near-ties decided by the rounded products) and this one only
// as the dividing instance; profiles/HISTORY.md has its numbers.
constexpr uint32_t kTieUlps = 3;    // see above; must stay below scan_block's bias (64)
#ifndef RF_OUTLINE_RESCANS
#define RF_OUTLINE_RESCANS 0           // 1: the rescans of contested cells as real functions (a call) instead of inlined code
#endif
#if RF_OUTLINE_RESCANS
#define RF_RESCAN_FN __device__ __attribute__((noinline))
#else
#define RF_RESCAN_FN __device__ __forceinline__
#endif
#ifndef RF_RESOLVE_BY_DIVIDING_ALL
#define RF_RESOLVE_BY_DIVIDING_ALL 0   // 1: contested cells rescanned by scan_faces_strict (rounds 5's resolution; A/B)
#endif

__device__ __forceinline__ void load_geo_block(const uint32_t *src, GeoXY &A, GeoZ &B) {
    A = *reinterpret_cast<const GeoXY *>(src);
    B = *reinterpret_cast<const GeoZ *>(src + 4);
}

// |bits(a) - bits(b)| + bias as unsigned integers: the number of floats between two products of the same sign, huge for
// products of opposite signs, 0 for two (+)zeros; `bias` lifts the lanes that sit a block out above any threshold
__device__ __forceinline__ uint32_t bit_distance(float a, float b, uint32_t bias) {
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(bias));
    return r;
}

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t m = a < b ? a : b;
    return m < c ? m : c;
}

// the ray and the cell as the packed scan wants them (every scalar broadcast to both halves of a register pair)
struct ScanRay {
    v2f Px, Py, Pz, Ox, Oy, Oz, dx, dy, dz;
};

__device__ __forceinline__ ScanRay scan_ray(float Px, float Py, float Pz, float Ox, float Oy, float Oz, float dx, float dy,
                                            float dz) {
    ScanRay R;
    R.Px = {Px, Px}; R.Py = {Py, Py}; R.Pz = {Pz, Pz};
    R.Ox = {Ox, Ox}; R.Oy = {Oy, Oy}; R.Oz = {Oz, Oz};
    R.dx = {dx, dx}; R.dy = {dy, dy}; R.dz = {dz, dz};
    return R;
}

// running state of a scan: best exit so far as the fraction best.y / best.x (inf/1 loses against any valid face), where
// it is, and the smallest bit distance of any comparison made so far.  The winner's position is kept as the entry number
// of its block (a register, rewritten once per block) and its two low bits as WAVE MASKS in scalar registers, updated by
// scalar instructions from the comparison masks -- four vector instructions per block fewer than selecting an index per
// pair, on a kernel bound by vector issue.
struct ScanState {
    v2f best;
    uint32_t blk;              // first entry of the block that holds the best exit so far; kNone: none yet
    unsigned long long b0, b1; // per lane (bit = lane): bits 0 and 1 of its position inside that block
    uint32_t tie;
};

__device__ __forceinline__ void scan_begin(ScanState &S) {
    S.best = {1.0f, __builtin_inff()};
    S.blk = kNone;
    S.b0 = S.b1 = 0ull;
    S.tie = 0xFFFFFFFFu;
}

// num and dp of the two faces whose fp16 offsets sit in the halves of (wx, wy, wz), as tracing_utils.cuh:57-60 forms
// them: every packed lane is an independent IEEE operation, the same roundings as the scalar form
__device__ __forceinline__ void face_pair(uint32_t wx, uint32_t wy, uint32_t wz, const ScanRay &R, v2f &num, v2f &dpp) {
    const v2f half2 = {0.5f, 0.5f};
    const v2f ox = {half_lo(wx), half_hi(wx)};
    const v2f oy = {half_lo(wy), half_hi(wy)};
    const v2f oz = {half_lo(wz), half_hi(wz)};
    dpp = fma2(ox, R.dx, fma2(oy, R.dy, oz * R.dz));
    const v2f vx = fma2(ox, half2, R.Px) - R.Ox;      // (P + o/2) - O; o/2 is exact, so P + o/2 is one rounding
    const v2f vy = fma2(oy, half2, R.Py) - R.Oy;
    const v2f vz = fma2(oz, half2, R.Pz) - R.Oz;
    num = fma2(vx, ox, fma2(vy, oy, vz * oz));
}

// The four faces of the block whose first entry is `k` (wave-uniform) against the running best.  Executed by the whole
// wave; `act` = the lanes whose list has this block (the others hold stale data: they take nothing and their distances are
// biased away).  Every predicate is a wave mask (v_cmp writes one), combined by scalar instructions and applied by
// v_cndmask: the selection costs the vector unit four compares and four selects per pair, nothing else.
__device__ __forceinline__ void scan_block(ScanState &S, uint32_t k, unsigned long long act, const GeoXY &A, const GeoZ &B,
                                           const ScanRay &R) {
    const v2f zero2 = {0.0f, 0.0f};
    const uint32_t bias = __builtin_amdgcn_inverse_ballot_w64(act) ? 0u : 64u;   // (an inline constant) > kTieUlps
    unsigned long long any = 0ull;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        v2f num, dpp;
        face_pair(h ? A.x23 : A.x01, h ? A.y23 : A.y01, h ? B.z23 : B.z01, R, num, dpp);
        // the nearer face of the pair (the first on a tie, or when the second is no exit) ...
        const v2f c = fma2(__builtin_shufflevector(num, num, 1, 0), dpp, zero2);    // {num1*dp0, num0*dp1}
        const unsigned long long v0 = ballot(dpp.x > 0.0f), v1 = ballot(dpp.y > 0.0f), lt10 = ballot(c.x < c.y);
        const unsigned long long w1m = v1 & ~(v0 & ~lt10);
        const bool w1 = __builtin_amdgcn_inverse_ballot_w64(w1m);
        const v2f cand = {w1 ? num.y : num.x, w1 ? dpp.y : dpp.x};      // (num, dp) of the pair's winner
        // ... against the running best, kept as the adjacent pair (db, nb): one packed operation forms both products
        const v2f ab = fma2(cand, S.best, zero2);                        // {num_w*db, dp_w*nb}
        const unsigned long long tm = (w1m | v0) & ballot(ab.x < ab.y) & act;
        const bool take = __builtin_amdgcn_inverse_ballot_w64(tm);
        S.best.x = take ? cand.y : S.best.x;
        S.best.y = take ? cand.x : S.best.y;
        S.b0 = (S.b0 & ~tm) | (w1m & tm);
        S.b1 = h ? (S.b1 | tm) : (S.b1 & ~tm);
        any |= tm;
        S.tie = min3u(S.tie, bit_distance(c.x, c.y, bias), bit_distance(ab.x, ab.y, bias));
    }
    S.blk = __builtin_amdgcn_inverse_ballot_w64(any) ? k : S.blk;
}

RF_RESCAN_FN ScanResult scan_faces_strict(const uint16_t *blk, uint32_t cnt, float Px, float Py, float Pz,
                                                        float Ox, float Oy, float Oz, float dx, float dy, float dz);

RF_RESCAN_FN ScanResult scan_resolve(const uint16_t *blk, uint32_t cnt, const ScanState &S, const ScanRay &R);

// what a finished tournament found; `contested`: some comparison was too close for the products to decide; `strict`: the
// products cannot be trusted at all for this cell (see the fail-safe below) -- only the dividing scan may decide it
__device__ __forceinline__ ScanResult scan_end(const ScanState &S, bool &contested, bool &strict) {
    ScanResult r;
    const bool found = S.blk != kNone;
    const uint32_t pos = (__builtin_amdgcn_inverse_ballot_w64(S.b1) ? 2u : 0u) | (__builtin_amdgcn_inverse_ballot_w64(S.b0) ? 1u : 0u);
    r.k = found ? S.blk + pos : kNone;
    r.t1 = found ? S.best.y / S.best.x : __builtin_inff();
    // a quotient that overflows is no exit for the reference either (t < inf fails); nothing can be nearer: the tournament's
    // winner has the smallest exact quotient
    if (!(r.t1 < __builtin_inff())) r.k = kNone;
    // Fail-safe of the certificate's one assumption ("no quotient is subnormal", above): with a zero or subnormal winning
    // quotient two exits more than an ulp-of-a-normal apart may still round to the SAME float, where the reference keeps the
    // first and the products order them strictly.  Any comparison between two such faces ends with a winner whose quotient is
    // at most theirs in magnitude class, so looking at the winner's exponent field is enough: zero => the dividing scan decides.
    const bool tiny = found && (__builtin_bit_cast(uint32_t, r.t1) & 0x7F800000u) == 0u;
    contested = (S.tie <= kTieUlps) | tiny;
    strict = tiny | !found | (r.k == kNone);
#ifdef RF_ISA_NO_CONTESTED_PATH
    // scripts/isa_stats.py --constants only (never shipped: results would be wrong): the rescans of contested cells compiled
    // out, so that the instructions counted per hop are those of a wave-step in which no lane is contested (97.7 % of them)
    // and not the statically inlined preambles of scan_resolve / scan_faces_strict
    contested = false;
#endif
    return r;
}

// A contested cell -- some comparison of the tournament was within kTieUlps floats -- does not need every face divided
// (scan_faces_strict: ~450 VALU with 20 IEEE divides, under divergence: 2.3 % of the wave-steps of a frame wait behind one
// lane's rescan).  Whatever the close comparisons did to the tournament, its winner W is SOME face
// with dp > 0, so the reference's exit -- the smallest rounded quotient, the first among equals -- has a rounded quotient
// <= W's.  One certified comparison per face against W (the same products and bit distance as in the tournament) drops every
// face whose rounded quotient is STRICTLY larger than W's; what is left -- W itself, faces within kTieUlps floats of it and
// faces the products order before it -- is divided, in list order, with the reference's strict '<': the reference's loop
// restricted to the only faces that can win it.  Typically two or three divides instead of twenty.
RF_RESCAN_FN ScanResult scan_resolve(const uint16_t *blk, uint32_t cnt, const ScanState &S, const ScanRay &R) {
    ScanResult r;
    r.t1 = __builtin_inff();
    r.k = kNone;
    const v2f zero2 = {0.0f, 0.0f};
    const v2f nw = {S.best.y, S.best.y}, dw = {S.best.x, S.best.x};    // the winner's num and dp, in both halves
    const uint32_t *src = reinterpret_cast<const uint32_t *>(blk);
    for (uint32_t k = 0; k < cnt; k += 4) {
        GeoXY A;
        GeoZ B;
        load_geo_block(src, A, B);
        src += 6;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v2f num, dpp;
            face_pair(h ? A.x23 : A.x01, h ? A.y23 : A.y01, h ? B.z23 : B.z01, R, num, dpp);
            const v2f pf = fma2(num, dw, zero2);        // num_face * dp_W
            const v2f pw = fma2(nw, dpp, zero2);        // num_W * dp_face
            // certainly behind the winner: the products say so by more than kTieUlps floats
            const bool far0 = (pf.x > pw.x) & (bit_distance(pf.x, pw.x, 0u) > kTieUlps);
            const bool far1 = (pf.y > pw.y) & (bit_distance(pf.y, pw.y, 0u) > kTieUlps);
            const bool c0 = (dpp.x > 0.0f) & !far0, c1 = (dpp.y > 0.0f) & !far1;
            if (c0) {
                const float t = num.x / dpp.x;
                if (t < r.t1) {
                    r.t1 = t;
                    r.k = k + (uint32_t)(2 * h);
                }
            }
            if (c1) {
                const float t = num.y / dpp.y;
                if (t < r.t1) {
                    r.t1 = t;
                    r.k = k + (uint32_t)(2 * h + 1);
                }
            }
        }
    }
    return r;
}

// Nearest exit of the ray from the cell whose `cnt` (a multiple of 4, padding included) face entries start at `blk` in
// the geo table.
__device__ __forceinline__ ScanResult scan_faces(const uint16_t *blk, uint32_t cnt, float Px, float Py,
                                                 float Pz, float Ox, float Oy, float Oz, float dx,
                                                 float dy, float dz) {
    ScanState S;
    scan_begin(S);
    const ScanRay R = scan_ray(Px, Py, Pz, Ox, Oy, Oz, dx, dy, dz);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(blk);
    // the loop is wave-uniform (it runs while any lane has a block left; a lane without one sits out): the winner's
    // position bits are wave masks, i.e. scalar values, which a loop that lanes LEAVE at different trips would force into
    // vector registers
    GeoXY A = {0u, 0u, 0u, 0u};
    GeoZ B = {0u, 0u};
    uint32_t k = 0;
    unsigned long long act = ballot(0u < cnt);
    if (act != 0ull) {
        do {
            if (__builtin_amdgcn_inverse_ballot_w64(act)) {
                load_geo_block(src, A, B);
                src += 6;
            }
            scan_block(S, k, act, A, B, R);
            k += 4;
            act = ballot(k < cnt);
        } while (act != 0ull);
    }
    bool contested, strict;
    ScanResult r = scan_end(S, contested, strict);
    if (contested) {
        if (RF_RESOLVE_BY_DIVIDING_ALL || strict) r = scan_faces_strict(blk, cnt, Px, Py, Pz, Ox, Oy, Oz, dx, dy, dz);
        else r = scan_resolve(blk, cnt, S, R);
    }
    return r;
}

// The first K blocks of a cell's face list, all requested at once (a cell has 18.6 faces on average: K = 6 holds the
// whole list of most cells); lists shorter than K blocks re-request their last block (an L1 hit, not used).
template <int K>
struct GeoBlocks {
    GeoXY A[K > 0 ? K : 1];
    GeoZ B[K > 0 ? K : 1];
};

template <int K>
__device__ __forceinline__ void load_geo_blocks(const uint16_t *geo, uint32_t first, uint32_t cnt, GeoBlocks<K> &G) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(geo + (size_t)first * 3u);
    const uint32_t last = cnt ? (cnt >> 2) - 1u : 0u;
#pragma unroll
    for (int i = 0; i < K; ++i) load_geo_block(src + 6u * (last < (uint32_t)i ? last : (uint32_t)i), G.A[i], G.B[i]);
}

// scan_faces with the first K blocks already in registers (or on their way); same result
template <int K>
__device__ __forceinline__ ScanResult scan_faces_eager(const GeoBlocks<K> &G, const uint16_t *blk, uint32_t cnt, float Px,
                                                       float Py, float Pz, float Ox, float Oy, float Oz, float dx,
                                                       float dy, float dz) {
    ScanState S;
    scan_begin(S);
    const ScanRay R = scan_ray(Px, Py, Pz, Ox, Oy, Oz, dx, dy, dz);
#pragma unroll
    for (int i = 0; i < K; ++i)
        scan_block(S, (uint32_t)(4 * i), ballot((uint32_t)(4 * i) < cnt), G.A[i], G.B[i], R);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(blk) + 6 * K;
    GeoXY A = {0u, 0u, 0u, 0u};
    GeoZ B = {0u, 0u};
    uint32_t k = 4u * K;
    unsigned long long act = ballot(k < cnt);
    while (act != 0ull) {
        if (__builtin_amdgcn_inverse_ballot_w64(act)) {
            load_geo_block(src, A, B);
            src += 6;
        }
        scan_block(S, k, act, A, B, R);
        k += 4;
        act = ballot(k < cnt);
    }
    bool contested, strict;
    ScanResult r = scan_end(S, contested, strict);
    if (contested) {
        if (RF_RESOLVE_BY_DIVIDING_ALL || strict) r = scan_faces_strict(blk, cnt, Px, Py, Pz, Ox, Oy, Oz, dx, dy, dz);
        else r = scan_resolve(blk, cnt, S, R);
    }
    return r;
}

// ---- block-level LDS cache of cell entries (rf_launch_opts.forward_mode = 5; experiment) -----------------------------
// The rays of a 256-slot group of a sorted flat batch meet a cell about eleven times, tens of steps apart
// (scripts/model_train_batch.py): too far for L1, and 160 resident blocks per XCD leave each an L2 share of ~100 cells.
// A direct-mapped table in LDS keyed by cell -- {cell record, the first kEagerBlocks face blocks}, 176 bytes -- serves
// what a block re-visits (the model: 0.56 -> 0.25 fetches per visit at 304 entries), and a hit takes the second
// dependent round trip (link -> cell record + face blocks) out of the hop.  No lock, no barrier in the walk: an entry
// carries a 64-bit tag {version, cell}; a writer claims it by compare-and-swap to {version + 1, busy}, writes, and
// publishes {version + 1, cell}; a reader takes tag, data, tag again in program order (a wave's LDS operations execute in
// order) and has a hit only when both tags are the same and name its cell.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kCacheBusy = 0xFFFFFFFEu;

template <int K>
struct __attribute__((aligned(16))) CellEntry {
    unsigned long long tag;
    unsigned long long unused;
    u32x4 head;        // {x, y, z, density}
    u32x4 A[K];        // GeoXY of the first K face blocks
    u32x2 B[K];        // GeoZ
};

__device__ __forceinline__ uint32_t cache_slot(uint32_t cell, uint32_t entries) {
    return __umulhi(cell * 2654435761u, entries);
}

// true: head / G now hold the entry of `cell`
template <int K>
__device__ __forceinline__ bool cache_read(CellEntry<K> *cache, uint32_t entries, uint32_t cell, float4 &head,
                                           GeoBlocks<K> &G) {
    // (explicitly LDS-typed: the optimiser does not infer the address space of volatile accesses, and flat loads count
    // against both wait counters)
    typedef volatile __attribute__((address_space(3))) CellEntry<K> *EntryPtr;
    EntryPtr e = (EntryPtr)(cache + cache_slot(cell, entries));
    const unsigned long long t0 = e->tag;
    const u32x4 h = e->head;
    u32x4 a[K];
    u32x2 b[K];
#pragma unroll
    for (int i = 0; i < K; ++i) a[i] = e->A[i];
#pragma unroll
    for (int i = 0; i < K; ++i) b[i] = e->B[i];
    const unsigned long long t1 = e->tag;
    if ((uint32_t)t0 != cell || t1 != t0) return false;
    head = __builtin_bit_cast(float4, h);
#pragma unroll
    for (int i = 0; i < K; ++i) {
        G.A[i] = __builtin_bit_cast(GeoXY, a[i]);
        G.B[i] = __builtin_bit_cast(GeoZ, b[i]);
    }
    return true;
}

template <int K>
__device__ __forceinline__ void cache_fill(CellEntry<K> *cache, uint32_t entries, uint32_t cell, const float4 &head,
                                           const GeoBlocks<K> &G) {
    CellEntry<K> *e = cache + cache_slot(cell, entries);
    typedef volatile __attribute__((address_space(3))) CellEntry<K> *EntryPtr;
    EntryPtr v = (EntryPtr)e;
    const unsigned long long old = v->tag;
    const uint32_t holds = (uint32_t)old;
    if (holds == cell || holds == kCacheBusy) return;     // already there / another lane is writing it
    const unsigned long long version = ((old >> 32) + 1ull) << 32;
    if (atomicCAS(&e->tag, old, version | (unsigned long long)kCacheBusy) != old) return;
    v->head = __builtin_bit_cast(u32x4, head);
#pragma unroll
    for (int i = 0; i < K; ++i) v->A[i] = __builtin_bit_cast(u32x4, G.A[i]);
#pragma unroll
    for (int i = 0; i < K; ++i) v->B[i] = __builtin_bit_cast(u32x2, G.B[i]);
    v->tag = version | (unsigned long long)cell;
}

// The reference's scan evaluated the way the reference writes it (tracing_utils.cuh:43-67): EVERY face divided (IEEE), a
// running minimum of the rounded quotients, strict '<' (the first minimum wins).  It resolves the contested cells of the
// scans above, and is selectable for whole launches as rf_launch_opts.forward_mode = 3 -- the independent instance the
// filtered scan is tested against bit for bit (tests/test_gpu_parity.py); a correctly rounded divide per face makes it
// 40 % slower on a frame.
RF_RESCAN_FN ScanResult scan_faces_strict(const uint16_t *blk, uint32_t cnt, float Px, float Py, float Pz,
                                                        float Ox, float Oy, float Oz, float dx, float dy, float dz) {
    ScanResult r;
    r.t1 = __builtin_inff();
    r.k = kNone;
    const ScanRay R = scan_ray(Px, Py, Pz, Ox, Oy, Oz, dx, dy, dz);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(blk);
    for (uint32_t k = 0; k < cnt; k += 4) {
        GeoXY A;
        GeoZ B;
        load_geo_block(src, A, B);
        src += 6;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v2f num, dpp;
            face_pair(h ? A.x23 : A.x01, h ? A.y23 : A.y01, h ? B.z23 : B.z01, R, num, dpp);
            // the divides and the running minimum stay per face, in list order
            const float t0 = num.x / dpp.x, t1 = num.y / dpp.y;
            const bool take0 = (dpp.x > 0.0f) & (t0 < r.t1);
            r.t1 = take0 ? t0 : r.t1;
            r.k = take0 ? k + (uint32_t)(2 * h) : r.k;
            const bool take1 = (dpp.y > 0.0f) & (t1 < r.t1);
            r.t1 = take1 ? t1 : r.t1;
            r.k = take1 ? k + (uint32_t)(2 * h + 1) : r.k;
        }
    }
    return r;
}

// fp16-rounded offset from cell p to its neighbour q, as the face tables hold it (pack_diff)
__device__ __forceinline__ void face_offset(const float4 &p, const float4 &q, float &ox, float &oy, float &oz) {
    ox = (float)(_Float16)(q.x - p.x);
    oy = (float)(_Float16)(q.y - p.y);
    oz = (float)(_Float16)(q.z - p.z);
}

// ------------------------------------------------------------------------------------------
// colour of a cell for this ray     reference: load_sh_as_rgb, sh_utils.cuh:72-83 (+ :50-54)

struct __attribute__((packed, aligned(4))) Float4U {   // four floats at a 4-byte aligned address
    float x, y, z, w;
};

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
        // rows of odd pitch (A = 13, 49) are only 4-byte aligned: the load is declared as such (one global_load_dwordx4
        // all the same: the hardware serves it in its unaligned access mode, scripts/probe/unaligned.hip)
        const Float4U *row = reinterpret_cast<const Float4U *>(
            reinterpret_cast<const float *>(fv.sh) + (size_t)cell * fv.sh_stride);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const Float4U w = row[i];
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
// the path through the compositing block).

// Waves per SIMD the register allocation is told to aim for: 6 (80 VGPRs, no spills) for the plain
// instances up to SH degree 2 -- measured 3 % faster than leaving the choice to the compiler, which
// lands on the same occupancy with a worse schedule; 7 / 8 (72 / 64 VGPRs, a few spills) measured 1-2 % slower --,
// 5 for the plain fp32 SH-degree-3 instances (96 VGPRs, 11 dwords spilled outside the scan: 4M-point 4K frame 23.5 ->
// 22.2 ms; 6 waves: 22.8 then, 20.26 against 20.32 with the final kernel: no difference), 6 for the fp16 ones, whose
// rows are half as many registers (the render path, 1557x1038 at SH 3: 2.65 -> 2.53 ms: profiles/r03/q_final_bench_render.json against the final evidence) and 4
// where quantile / statistics code is compiled in.
#ifndef RF_FWD_WAVES_OTHER
#define RF_FWD_WAVES_OTHER 4
#endif
#ifndef RF_FWD_WAVES_MAIN
#define RF_FWD_WAVES_MAIN 6
#endif
#ifndef RF_FWD_WAVES_D3
#define RF_FWD_WAVES_D3 5
#endif
#ifndef RF_FWD_WAVES_D3_HALF
#define RF_FWD_WAVES_D3_HALF 6
#endif
// EAGER instances (the first kEagerBlocks face blocks of a cell requested at once, at the hop that enters it): 122 VGPRs
// without spills at 4 waves per SIMD up to SH degree 2; 3 waves for degree 3 and with quantile code (measured on the
// training batch: degree 3 forward 4.94 ms at 4 waves, 4.64 at 3; with quantiles 7.66 -- spills -- and 4.62).
#ifndef RF_FWD_WAVES_EAGER
#define RF_FWD_WAVES_EAGER 4
#endif
constexpr int kScanBlocks = 0, kScanEager = 1, kScanStrict = 2, kScanCached = 3;   // forward_kernel's SCAN parameter
constexpr int forward_waves(int deg, bool half, bool quant, bool stats, int scan) {
    if (scan == kScanEager || scan == kScanCached) return (deg <= 2 && !quant) ? RF_FWD_WAVES_EAGER : 3;
    if (quant || stats) return RF_FWD_WAVES_OTHER;
    return deg <= 2 ? RF_FWD_WAVES_MAIN : (half ? RF_FWD_WAVES_D3_HALF : RF_FWD_WAVES_D3);
}

// entries of the kScanCached instances' LDS table: what the blocks that share a CU leave each other of its 160 KB
#ifndef RF_CELL_CACHE_ENTRIES
#define RF_CELL_CACHE_ENTRIES 0
#endif
constexpr uint32_t cell_cache_entries(int waves) {
    return RF_CELL_CACHE_ENTRIES ? (uint32_t)RF_CELL_CACHE_ENTRIES : (waves <= 3 ? 304u : 232u);
}

template <int DEG, bool HALF, bool BENCH, bool QUANT, bool STATS, int SCAN>
__global__ __launch_bounds__(kBlock, forward_waves(DEG, HALF, QUANT, STATS, SCAN)) void forward_kernel(FwdParams p) {
    constexpr bool CACHED = SCAN == kScanCached;
    constexpr bool EAGER = SCAN == kScanEager || CACHED;
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t kEntries = cell_cache_entries(forward_waves(DEG, HALF, QUANT, STATS, SCAN));
    CellEntry<kEagerBlocks> *cache = nullptr;
    if constexpr (CACHED) {
        __shared__ CellEntry<kEagerBlocks> s_cells[kEntries];
        for (uint32_t e = threadIdx.x; e < kEntries; e += (uint32_t)kBlock) s_cells[e].tag = (unsigned long long)kNone;
        __syncthreads();    // the only barrier of the launch: before the walk
        cache = s_cells;
    }
#ifdef RF_EXPERIMENT_TIMELINE
    const unsigned long long tl_start = wall_clock64();
#endif

    uint32_t ray, slot;
    bool alive = map_ray(p.grid, ray, slot);
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
    const uint32_t nq = QUANT ? p.nq : 0u;
    const float *qp = nullptr;
    float cq = 0.0f;
    if (nq && alive) {
        qp = p.quantiles + (size_t)ray * nq;
        cq = qp[0];
    }
    const float thr = p.settings.weight_threshold;
    const uint32_t max_steps = p.settings.max_intersections;

    unsigned long long st_cells = 0, st_faces = 0, st_hops = 0, st_seg = 0, st_lit = 0;
    constexpr bool want_stats = STATS;

    float t0 = 0.0f;
    uint32_t n = 0;
    uint32_t nb = 0, cnt = 0;
    float4 head = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (alive) {
        nb = fv.poff[cur];
        cnt = fv.poff[cur + 1] - nb;
        head = fv.cells[cur];
    }
    GeoBlocks<EAGER ? kEagerBlocks : 0> GB;
    if constexpr (EAGER) load_geo_blocks(fv.geo, nb, cnt, GB);
    bool fetched = alive;   // CACHED: head / GB of the current cell came from memory, the table may want them
    uint32_t wave_steps = 0;
    uint32_t hops = 0;
    uint32_t trail_step = 0;             // wave-uniform: steps taken so far = hops of every lane still alive
    uint32_t *trail_row = p.trail;       // wave-uniform: row `trail_step` of the trail
    const uint32_t slot4 = slot * 4u;    // the lane's byte offset in a row (rf_trace_forward refuses rows of 4 GB and more)
    while (ballot(alive) != 0ull) {
        wave_steps++;
        if (alive) {
            n++;
            if (n > max_steps) alive = false;
        }
        ScanResult sr;
        sr.t1 = __builtin_inff();
        sr.k = kNone;
        const bool scanned = alive;
        if (alive) {
            if constexpr (EAGER)
                sr = scan_faces_eager(GB, fv.geo + (size_t)nb * 3u, cnt, head.x, head.y, head.z, Ox, Oy, Oz, dx, dy, dz);
            else if constexpr (SCAN == kScanStrict)
                sr = scan_faces_strict(fv.geo + (size_t)nb * 3u, cnt, head.x, head.y, head.z, Ox, Oy, Oz, dx, dy, dz);
            else
                sr = scan_faces(fv.geo + (size_t)nb * 3u, cnt, head.x, head.y, head.z, Ox, Oy, Oz, dx, dy, dz);
            if (want_stats) {
                st_cells++;
                st_faces += fv.offsets[cur + 1] - fv.offsets[cur];
                if (p.visit_marks) p.visit_marks[cur] = (uint8_t)1;
            }
            if (sr.k == kNone) alive = false;
        }
        // What compositing needs of the cell being left; the hop then loads the next cell straight into the walk's state
        // (cur, head, nb, cnt) -- with the old values dead, the exec-masked loads land in those very registers instead of
        // in temporaries that have to be copied over at the end of the step (seven moves per lane-step before).
        const float dens = head.w;
        const uint32_t cell = cur;
        if constexpr (CACHED) {
            // everything requested so far has been consumed by the scan (or belongs to lanes that are done): telling the
            // compiler so keeps its wait for the table's stores below from covering the link load as well
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            Link link{0u, 0u, 0u};
            if (alive) link = fv.link[nb + sr.k];
            // while the link is on its way: what this cell's scan fetched from memory goes into the table
            if (scanned && fetched) cache_fill(cache, kEntries, cell, head, GB);
            if (alive) {
                cur = link.nbr;
                nb = link.first;
                cnt = link.count;
                fetched = !cache_read(cache, kEntries, cur, head, GB);
                if (fetched) {
                    head = fv.cells[cur];
                    load_geo_blocks(fv.geo, nb, cnt, GB);
                }
            }
        }
        if (alive) {
            if constexpr (!CACHED) {
                const Link link = fv.link[nb + sr.k];
                cur = link.nbr;
                nb = link.first;
                cnt = link.count;
                head = fv.cells[cur];
                if constexpr (EAGER) load_geo_blocks(fv.geo, nb, cnt, GB);
            }
            if constexpr (!BENCH) {
                // trail: the cell each hop enters, for trace_backward to replay
                if (p.trail) {
                    // every lane that is still alive has hopped in every step so far: hop number = step number, the same
                    // for the whole wave -- the trail row is a scalar base, the lane adds its slot (a 32-bit offset)
                    if (trail_step < p.trail_cap)
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(trail_row) + slot4) = cur;
                    hops = trail_step + 1u;
                }
            }
        }
        if (alive) {
            const float t1 = sr.t1;
            const bool segment = t1 > t0;
            if (want_stats) st_hops++;
            if (want_stats && segment) {
                st_seg++;
                st_lit += (dens > 1e-6f) ? 1u : 0u;
            }
            // A segment through a cell of density exactly 0 changes nothing (alpha = 1 - exp(-0) = 0,
            // weight 0, transmittance unchanged): skipped -- a wave in empty space skips the block.
            if (segment && dens != 0.0f) {
                float r = 0.0f, g = 0.0f, b = 0.0f;
                if (dens > 1e-6f) cell_rgb<DEG, HALF>(fv, cell, sh, r, g, b);
                float dt = __builtin_fmaxf(t1 - t0, 0.0f);
                float alpha = 1.0f - exp_(-dens * dt);
                float w = T * alpha;
                if constexpr (!BENCH) {
                    if (p.contribution) unsafeAtomicAdd(p.contribution + cell, w);
                }
                Cr = fma_(w, r, Cr);
                Cg = fma_(w, g, Cg);
                Cb = fma_(w, b, Cb);
                float Tn = T * (1.0f - alpha);
                if constexpr (QUANT) {
                    while (qi < nq && Tn < cq) {
                        p.qdepth[(size_t)ray * nq + qi] = t0 + log_(T / cq) / dens;
                        p.qidx[(size_t)ray * nq + qi] = cell;
                        qi++;
                        if (qi < nq) cq = qp[qi];
                    }
                }
                T = Tn;
                if (!(T > thr)) alive = false;
            }
            t0 = segment ? t1 : t0;     // fmaxf(t0, t1): t0 is never NaN
        }
        trail_step++;
        if (trail_row) trail_row += p.trail_slots;
    }

    if constexpr (!BENCH) {
        if (p.trail && slot != kNone) p.trail_hops[slot] = valid ? hops : 0u;
    }
    // what the tile cost: the steps of its longest ray (feeds the tile order of the next launches over this frame)
    if (p.tile_cost && slot != kNone) {
        uint32_t longest = valid ? n : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)longest, off, 64);
            longest = other > longest ? other : longest;
        }
        if (lane == 0u) atomicMax(p.tile_cost + slot / (uint32_t)kBlock, longest);
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
#ifdef RF_EXPERIMENT_TIMELINE
        // per-block record behind the 8 counters: {XCC id | HW_ID << 8, start, end, wave steps}
        if (want_stats && threadIdx.x == 0) {
            unsigned long long *rec = p.stats + 8 + 4ull * blockIdx.x;
            rec[0] = (unsigned long long)__builtin_amdgcn_s_getreg(6164) |
                     ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 8);
            rec[1] = tl_start;
            rec[2] = wall_clock64();
            rec[3] = wave_steps;
        }
#endif
        if (want_stats) {
            atomicAdd(p.stats + 0, st_cells);
            atomicAdd(p.stats + 1, st_faces);
            atomicAdd(p.stats + 2, st_hops);
            atomicAdd(p.stats + 3, st_seg);
            atomicAdd(p.stats + 4, st_lit);
            if (lane == 0) atomicAdd(p.stats + 6, (unsigned long long)wave_steps);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Persistent waves with refill (rf_launch_opts.forward_mode = 4; experiment): the north star's third mechanism --
// "wavefront ballot / prefix-sum used to compact active rays as transmittance falls off".  A launch is as many blocks as
// the chip holds at once; a wave keeps walking, and whenever at most RF_REFILL_BELOW of its lanes are still alive the dead
// ones write their ray's outputs and take new rays from a global queue -- ONE atomicAdd for the wave (the ballot counts
// the dead lanes, the prefix count of the ballot ranks them), in the order the ordinary launch would have taken them
// (slot q = block q / 256, thread q % 256 of that launch), so the trail slots, the tile orders and every output are the
// same.  Live lanes keep their state in registers: nothing is moved, the wave is simply full again.  Results are per ray
// and do not depend on which lane walked it (bit-identical: tests/test_gpu_parity.py runs this mode too).
// Measured (profiles/r04/l_persistent_refill_ab.log; forward of the 1080p frame / of the training batch, ms): the ordinary
// launch 4.46 / 4.34; refilling when at most 0 / 16 / 32 / 48 / 56 / 63 lanes are alive 4.97 / 5.61 / 5.85 / 6.22 / 7.10 / 12.6 and
// 5.10 / 5.33 / 5.59 / 5.98 / 6.50 / 8.80 -- the more a wave refills, the slower: 92.8 % (87.7 %) of its lane-steps are live
// anyway, a refilled lane's ray comes from another tile (other cells, other cache lines), and the refill itself (ray load,
// SH basis, start cell) runs with most of the wave masked off.  Kept as a tested experiment mode; never picked by mode 0.
#ifndef RF_REFILL_BELOW
#define RF_REFILL_BELOW 16
#endif
#ifndef RF_PERSISTENT_WAVES
#define RF_PERSISTENT_WAVES 5
#endif

template <int DEG, bool HALF>
__global__ __launch_bounds__(kBlock, (DEG <= 2 ? RF_PERSISTENT_WAVES : 4)) void forward_persistent_kernel(
    FwdParams p, uint32_t *__restrict__ queue, uint32_t queue_blocks) {
    const uint32_t lane = threadIdx.x & 63u;
    const FoamView &fv = p.foam;
    const uint32_t total = queue_blocks * (uint32_t)kBlock;
    const float thr = p.settings.weight_threshold;
    const uint32_t max_steps = p.settings.max_intersections;

    bool alive = false, valid = false;
    uint32_t ray = 0, slot = kNone, cur = 0, n = 0, hops = 0, nb = 0, cnt = 0;
    float Ox = 0.0f, Oy = 0.0f, Oz = 0.0f, dx = 0.0f, dy = 0.0f, dz = 1.0f;
    float sh[sh_dim(DEG)];
    sh_basis<DEG>(dx, dy, dz, sh);
    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f, t0 = 0.0f;
    float4 head = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool more = true;      // wave-uniform: the queue may still hold rays

    for (;;) {
        unsigned long long live = ballot(alive);
        if (more && __builtin_popcountll(live) <= RF_REFILL_BELOW) {
            // ---- the dead lanes hand in their ray ...
            if (!alive && valid) {
                if (p.trail) p.trail_hops[slot] = hops;
                if (p.tile_cost) atomicMax(p.tile_cost + slot / (uint32_t)kBlock, n);
                const float a = 1.0f - T;
                if constexpr (HALF) {
                    uint32_t lo = (uint32_t)float_to_half_bits(Cr) | ((uint32_t)float_to_half_bits(Cg) << 16);
                    uint32_t hi = (uint32_t)float_to_half_bits(Cb) | ((uint32_t)float_to_half_bits(a) << 16);
                    reinterpret_cast<uint2 *>(p.rgba)[ray] = make_uint2(lo, hi);
                } else {
                    reinterpret_cast<float4 *>(p.rgba)[ray] = make_float4(Cr, Cg, Cb, a);
                }
                if (p.nint) p.nint[ray] = n;
                valid = false;
            }
            // ---- ... and take the next ones: one atomic for the wave, the ballot's prefix count ranks the lanes
            const unsigned long long dead = ~live;
            const uint32_t need = (uint32_t)__builtin_popcountll(dead);
            uint32_t base = 0;
            if (lane == 0u) base = atomicAdd(queue, need);
            base = readlane(base, 0);
            const uint32_t rank = (uint32_t)__builtin_popcountll(dead & ((1ull << lane) - 1ull));
            more = base + need < total;
            if (!alive) {
                const uint32_t q = base + rank;
                slot = kNone;
                if (q < total && map_slot(p.grid, q / (uint32_t)kBlock, q % (uint32_t)kBlock, queue_blocks, ray, slot)) {
                    const float *rp = p.rays + (size_t)ray * 6;
                    Ox = rp[0];
                    Oy = rp[1];
                    Oz = rp[2];
                    dx = rp[3];
                    dy = rp[4];
                    dz = rp[5];
                    const float nrm = sqrtf(dot3(dx, dy, dz, dx, dy, dz));
                    dx = dx / nrm;
                    dy = dy / nrm;
                    dz = dz / nrm;
                    cur = p.start[ray];
                    sh_basis<DEG>(dx, dy, dz, sh);
                    T = 1.0f;
                    Cr = Cg = Cb = 0.0f;
                    t0 = 0.0f;
                    n = 0;
                    hops = 0;
                    nb = fv.poff[cur];
                    cnt = fv.poff[cur + 1] - nb;
                    head = fv.cells[cur];
                    alive = valid = true;
                } else if (slot != kNone && p.trail) {
                    p.trail_hops[slot] = 0u;       // a slot of the launch without a ray (frame edge): nothing to replay
                }
            }
            live = ballot(alive);
        }
        if (live == 0ull) {
            if (!more) break;
            continue;
        }
        // ---- one step of every live lane: forward_kernel's body
        if (alive) {
            n++;
            if (n > max_steps) alive = false;
        }
        ScanResult sr;
        sr.t1 = __builtin_inff();
        sr.k = kNone;
        if (alive) {
            sr = scan_faces(fv.geo + (size_t)nb * 3u, cnt, head.x, head.y, head.z, Ox, Oy, Oz, dx, dy, dz);
            if (sr.k == kNone) alive = false;
        }
        uint32_t nxt = 0, nnb = 0, ncnt = 0;
        float4 nhead = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (alive) {
            const Link link = fv.link[nb + sr.k];
            nxt = link.nbr;
            nnb = link.first;
            ncnt = link.count;
            nhead = fv.cells[nxt];
            if (p.trail) {
                if (hops < p.trail_cap) p.trail[(size_t)hops * p.trail_slots + slot] = nxt;
                hops++;
            }
        }
        if (alive) {
            const float t1 = sr.t1;
            if (t1 > t0 && head.w != 0.0f) {
                const float sdens = head.w;
                float r = 0.0f, g = 0.0f, b = 0.0f;
                if (sdens > 1e-6f) cell_rgb<DEG, HALF>(fv, cur, sh, r, g, b);
                const float dt = __builtin_fmaxf(t1 - t0, 0.0f);
                const float alpha = 1.0f - exp_(-sdens * dt);
                const float w = T * alpha;
                if (p.contribution) unsafeAtomicAdd(p.contribution + cur, w);
                Cr = fma_(w, r, Cr);
                Cg = fma_(w, g, Cg);
                Cb = fma_(w, b, Cb);
                T = T * (1.0f - alpha);
                if (!(T > thr)) alive = false;
            }
            t0 = __builtin_fmaxf(t0, t1);
            cur = nxt;
            head = nhead;
            nb = nnb;
            cnt = ncnt;
        }
    }
    // the rays that ended after the last refill
    if (valid) {
        if (p.trail) p.trail_hops[slot] = hops;
        if (p.tile_cost) atomicMax(p.tile_cost + slot / (uint32_t)kBlock, n);
        const float a = 1.0f - T;
        if constexpr (HALF) {
            uint32_t lo = (uint32_t)float_to_half_bits(Cr) | ((uint32_t)float_to_half_bits(Cg) << 16);
            uint32_t hi = (uint32_t)float_to_half_bits(Cb) | ((uint32_t)float_to_half_bits(a) << 16);
            reinterpret_cast<uint2 *>(p.rgba)[ray] = make_uint2(lo, hi);
        } else {
            reinterpret_cast<float4 *>(p.rgba)[ray] = make_float4(Cr, Cg, Cb, a);
        }
        if (p.nint) p.nint[ray] = n;
    }
}

// ------------------------------------------------------------------------------------------
// backward                                           reference: pipeline.cu:132-343
// Re-walks the ray with the same control flow as forward and scatters gradients.  All the
// reference's behaviours are kept: the accumulators of the last visited
// cell and of its exit neighbour are never flushed; the first segment's dt0/dP term is taken
// against the world origin; dL/dt0 receives depth_grad/s.
//
// MODE 1: one atomic per lane per value (the reference's scatter).
// MODE 2: lanes of the wave that sit in the same cell first sum their rows with a transposing
//         butterfly (rf_wave.hpp); one coalesced atomic row per distinct cell.

template <bool HALF>
__device__ __forceinline__ float load_attr_scalar(const void *base, size_t i) {
    if constexpr (HALF) {
        return (float)__builtin_bit_cast(_Float16, reinterpret_cast<const uint16_t *>(base)[i]);
    } else {
        return reinterpret_cast<const float *>(base)[i];
    }
}

// gradient accumulation: hardware fp32 atomic add (global_atomic_add_f32, no return)
__device__ __forceinline__ void grad_add(float *dst, float v) {
#ifdef RF_EXPERIMENT_NO_ATOMICS
    if (v == 123.456f) *dst = v;   // keeps the value computation alive, issues (almost) no memory op
#else
    unsafeAtomicAdd(dst, v);
#endif
}

constexpr int pow2_at_least(int v) { return v <= 8 ? 8 : (v <= 16 ? 16 : (v <= 32 ? 32 : 64)); }

template <int NB>
__device__ __forceinline__ void add_row_per_lane(float *dst, const float (&sh)[NB], float dLr,
                                                 float dLg, float dLb) {
#pragma unroll
    for (int i = 0; i < 3 * NB; ++i) {
        float gc = (i % 3 == 0) ? dLr : ((i % 3 == 1) ? dLg : dLb);
        grad_add(dst + i, sh[i / 3] * gc);
    }
}

// Adds this step's gradients to global memory.  Called by every lane (convergent).
//   has      lane composited a segment this step
//   row      its colour-gradient row is not all zero (lit cell, unclamped channel)
//   cur      the cell the row / density gradient belongs to
//   pg_on    (pgx,pgy,pgz) must be added to points_grad[prev]
template <int DEG, int MODE>
__device__ __forceinline__ void scatter_step(uint32_t lane, bool has, bool row, uint32_t cur,
                                             const float (&sh)[sh_dim(DEG)], float dLr, float dLg,
                                             float dLb, float dL_ds, bool pg_on, uint32_t prev,
                                             float pgx, float pgy, float pgz, float *attr_grad,
                                             float *points_grad, uint32_t pitch) {
    constexpr int NB = sh_dim(DEG);
    constexpr int A = 1 + 3 * NB;
    if constexpr (MODE == 1) {
        if (has) {
            if (pg_on) {
                float *pg = points_grad + 3 * (size_t)prev;
                grad_add(pg + 0, pgx);
                grad_add(pg + 1, pgy);
                grad_add(pg + 2, pgz);
            }
            float *dst = attr_grad + (size_t)cur * pitch;
            if (row) add_row_per_lane<NB>(dst, sh, dLr, dLg, dLb);
            grad_add(dst + (A - 1), dL_ds);
        }
    } else {
        constexpr int NV = pow2_at_least(A + 3);   // row (A values) + point gradient (3)
        bool pg_left = has && pg_on;        // point gradient not yet added
        bool ds_left = has;                 // density gradient not yet added
        // ---- full rows, one butterfly per distinct cell
        uint64_t todo = ballot(has && row);
        while (todo != 0ull) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t c = readlane(cur, leader);
            const bool mine = has && row && cur == c;
            const uint64_t same = ballot(mine);
            if (__builtin_popcountll(same) <= 2) {
                // (almost) nothing to share: the lanes add their own rows
                if (mine) add_row_per_lane<NB>(attr_grad + (size_t)cur * pitch, sh, dLr, dLg, dLb);
            } else {
                // the point gradient rides along when the whole group flushes to the same cell
                const uint32_t pl = readlane(prev, leader);
                const bool pg_uniform = ballot(mine && pg_on && prev != pl) == 0ull;
                const bool take_pg = mine && pg_on && pg_uniform;
                const bool any_pg = ballot(take_pg) != 0ull;
                float v[NV];
                const float mr = mine ? dLr : 0.0f, mg = mine ? dLg : 0.0f, mb = mine ? dLb : 0.0f;
#pragma unroll
                for (int i = 0; i < 3 * NB; ++i) {
                    float gc = (i % 3 == 0) ? mr : ((i % 3 == 1) ? mg : mb);
                    v[i] = sh[i / 3] * gc;
                }
                v[A - 1] = mine ? dL_ds : 0.0f;
                v[A + 0] = take_pg ? pgx : 0.0f;
                v[A + 1] = take_pg ? pgy : 0.0f;
                v[A + 2] = take_pg ? pgz : 0.0f;
#pragma unroll
                for (int i = A + 3; i < NV; ++i) v[i] = 0.0f;
                const float tot = transpose_reduce<NV>(v, lane);
                if (lane < (uint32_t)A) {
                    grad_add(attr_grad + (size_t)c * pitch + lane, tot);
                } else if (lane < (uint32_t)(A + 3) && any_pg) {
                    grad_add(points_grad + 3 * (size_t)pl + (lane - (uint32_t)A), tot);
                }
                if (mine) ds_left = false;
                if (take_pg) pg_left = false;
            }
            todo &= ~same;
        }
        // ---- density gradient of lanes whose row was not reduced above, per distinct cell
        todo = ballot(ds_left);
        while (todo != 0ull) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t c = readlane(cur, leader);
            const bool mine = ds_left && cur == c;
            const uint64_t same = ballot(mine);
            const float tot = wave_sum(mine ? dL_ds : 0.0f);
            if ((int)lane == leader) grad_add(attr_grad + (size_t)c * pitch + (A - 1), tot);
            todo &= ~same;
        }
        // ---- point gradients that did not ride along
        if (pg_left) {
            float *pg = points_grad + 3 * (size_t)prev;
            grad_add(pg + 0, pgx);
            grad_add(pg + 1, pgy);
            grad_add(pg + 2, pgz);
        }
    }
}

// Per-ray state of the backward pass (registers).
struct BwdRay {
    float Ox, Oy, Oz, dx, dy, dz;
    float gr, gg, gb, ga, outr, outg, outb, outa, err;
    float T, Cr, Cg, Cb, t0;
    uint32_t prev;
    float ppx, ppy, ppz;   // prev point
    float pgx, pgy, pgz;   // prev_point_grad
    float cgx, cgy, cgz;   // current_point_grad
    float ngx, ngy, ngz;   // next_point_grad
    uint32_t qi;
    float cq, cdg;
    const float *qp;
    const float *dgp;
};

// What one composited segment adds to the gradient buffers (scattered later by the whole wave).
struct StepGrad {
    bool has, row, pg_on;
    uint32_t cur, prev;
    float dLr, dLg, dLb, dL_ds, px, py, pz;
};

template <int DEG, bool HALF>
__device__ __forceinline__ void load_backward_ray(const BwdParams &p, uint32_t ray, BwdRay &R, uint32_t &cur) {
    const FoamView &fv = p.foam;
    const float *rp = p.rays + (size_t)ray * 6;
    R.Ox = rp[0];
    R.Oy = rp[1];
    R.Oz = rp[2];
    float dx = rp[3], dy = rp[4], dz = rp[5];
    float nrm = sqrtf(dot3(dx, dy, dz, dx, dy, dz));
    R.dx = dx / nrm;
    R.dy = dy / nrm;
    R.dz = dz / nrm;
    R.gr = load_attr_scalar<HALF>(p.rgba_grad, (size_t)ray * 4 + 0);
    R.gg = load_attr_scalar<HALF>(p.rgba_grad, (size_t)ray * 4 + 1);
    R.gb = load_attr_scalar<HALF>(p.rgba_grad, (size_t)ray * 4 + 2);
    R.ga = load_attr_scalar<HALF>(p.rgba_grad, (size_t)ray * 4 + 3);
    R.outr = load_attr_scalar<HALF>(p.rgba, (size_t)ray * 4 + 0);
    R.outg = load_attr_scalar<HALF>(p.rgba, (size_t)ray * 4 + 1);
    R.outb = load_attr_scalar<HALF>(p.rgba, (size_t)ray * 4 + 2);
    R.outa = load_attr_scalar<HALF>(p.rgba, (size_t)ray * 4 + 3);
    if (p.ray_error) R.err = load_attr_scalar<HALF>(p.ray_error, ray);
    cur = p.start[ray];
    const uint32_t nq = p.nq;
    if (nq) {
        R.qp = p.quantiles + (size_t)ray * nq;
        R.dgp = p.depth_grad + (size_t)ray * nq;
        R.cq = R.qp[0];
        for (uint32_t i = 0; i < nq; ++i) {
            uint32_t ci = p.qidx[(size_t)ray * nq + i];
            if (ci != kNone) {
                float s = fv.cells[ci].w;
                R.cdg += R.dgp[i] / s;
            }
        }
    }
}

__device__ __forceinline__ void init_backward_ray(BwdRay &R) {
    R.Ox = R.Oy = R.Oz = 0.0f;
    R.dx = R.dy = 0.0f;
    R.dz = 1.0f;
    R.gr = R.gg = R.gb = R.ga = R.outr = R.outg = R.outb = R.outa = R.err = 0.0f;
    R.T = 1.0f;
    R.Cr = R.Cg = R.Cb = 0.0f;
    R.t0 = 0.0f;
    R.prev = kNone;
    R.ppx = R.ppy = R.ppz = 0.0f;
    R.pgx = R.pgy = R.pgz = 0.0f;
    R.cgx = R.cgy = R.cgz = 0.0f;
    R.ngx = R.ngy = R.ngz = 0.0f;
    R.qi = 0;
    R.cq = R.cdg = 0.0f;
    R.qp = nullptr;
    R.dgp = nullptr;
}

// The backward functor for the segment [t0,t1] of cell `cur` (reference: pipeline.cu:180-331).
// Returns false when the ray terminates (transmittance below the threshold).
// PAIRS: the four bisector gradients as two shared evaluations (bisector_grad_pair) -- the same floats; worth it where
// registers are to spare (flat-batch replay: 4.47 -> 4.40 ms, every segment lit 11.98 -> 11.76), a loss where they are not
// (image replay at its 128-VGPR limit: 3.77 -> 3.99 ms, the shared values spill; profiles/r05/e_bisector_pairs_ab.log)
template <int DEG, bool HALF, bool QUANT = true, bool PAIRS = false>
__device__ __forceinline__ bool backward_segment(const BwdParams &p, BwdRay &R, const float (&sh)[sh_dim(DEG)],
                                                 uint32_t cur, float4 head, float4 nhead, float t1,
                                                 StepGrad &G) {
    const FoamView &fv = p.foam;
    const uint32_t nq = QUANT ? p.nq : 0u;
    const float t0 = R.t0;
    float s = head.w;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (s > 1e-6f) cell_rgb<DEG, HALF>(fv, cur, sh, r, g, b);
    float dt = __builtin_fmaxf(t1 - t0, 0.0f);
    float alpha = 1.0f - exp_(-s * dt);
    float w = R.T * alpha;
    float da_ds = dt * (1.0f - alpha);
    float da_ddt = (dt > 0.0f) ? s * (1.0f - alpha) : 0.0f;

    R.Cr = fma_(w, r, R.Cr);
    R.Cg = fma_(w, g, R.Cg);
    R.Cb = fma_(w, b, R.Cb);
    if (p.point_error) unsafeAtomicAdd(p.point_error + cur, w * R.err);

    float dLr = R.gr * w, dLg = R.gg * w, dLb = R.gb * w;
    float den = R.T * ((1.0f - alpha) + 1e-6f);
    float rr, rg, rb;
    div3_guarded(R.outr - R.Cr, R.outg - R.Cg, R.outb - R.Cb, den, rr, rg, rb);
    float dfr = r - rr;
    float dfg = g - rg;
    float dfb = b - rb;
    float dL_da = R.T * dot3(dfr, dfg, dfb, R.gr, R.gg, R.gb);
    dL_da = dL_da + ((1.0f - R.outa) * R.ga) / ((1.0f - alpha) + 1e-6f);

    float dL_ds = dL_da * da_ds;
    float dL_ddt = dL_da * da_ddt;
    float dL_dt0 = 0.0f;

    float Tn = R.T * (1.0f - alpha);
    if constexpr (QUANT) {
        while (R.qi < nq && Tn < R.cq) {
            float gi = R.dgp[R.qi] / s;
            dL_dt0 = dL_dt0 + gi;
            dL_ds = dL_ds + ((-gi) * log_(R.T / R.cq)) / s;
            R.cdg = R.cdg - gi;
            R.qi++;
            if (R.qi < nq) R.cq = R.qp[R.qi];
        }
        if (R.qi < nq) {
            dL_ds = fma_(-dt, R.cdg, dL_ds);
            dL_ddt = fma_(-s, R.cdg, dL_ddt);
        }
    }
    dL_dt0 = dL_dt0 + (-dL_ddt);
    float dL_dt1 = dL_ddt;

    // The four bisector gradients only enter through dL_dt0 / dL_dt1; both are exactly zero in a
    // cell of density 0 (83 % of the segments of the benchmark scene), where the block is skipped.
    // (Differs from the reference only if a bisector gradient is non-finite there: 0 * inf.)
    if (dL_dt0 != 0.0f || dL_dt1 != 0.0f) {
        float ax = 0.0f, ay = 0.0f, az = 0.0f;  // dt0_dprev
        float bx, by, bz;                        // dt1_dcurrent
        float ex, ey, ez;                        // dt0_dcurrent (vs prev, or vs the world origin on the first segment)
        float fx, fy, fz;                        // dt1_dnext
        if constexpr (PAIRS) {
            // the four gradients are two pairs over the same plane each -- (prev, cur) and (cur, next) --: one evaluation
            // per plane gives both, bit for bit what four separate evaluations return (rf_math.hpp)
            bisector_grad_pair(R.ppx, R.ppy, R.ppz, head.x, head.y, head.z, R.Ox, R.Oy, R.Oz, R.dx, R.dy, R.dz, ax, ay, az,
                               ex, ey, ez);
            if (R.prev == kNone) ax = ay = az = 0.0f;
            bisector_grad_pair(head.x, head.y, head.z, nhead.x, nhead.y, nhead.z, R.Ox, R.Oy, R.Oz, R.dx, R.dy, R.dz, bx, by,
                               bz, fx, fy, fz);
        } else {
            if (R.prev != kNone)
                bisector_grad(R.ppx, R.ppy, R.ppz, head.x, head.y, head.z, R.Ox, R.Oy, R.Oz, R.dx, R.dy, R.dz, ax, ay, az);
            bisector_grad(head.x, head.y, head.z, nhead.x, nhead.y, nhead.z, R.Ox, R.Oy, R.Oz, R.dx, R.dy, R.dz, bx, by, bz);
            bisector_grad(head.x, head.y, head.z, R.ppx, R.ppy, R.ppz, R.Ox, R.Oy, R.Oz, R.dx, R.dy, R.dz, ex, ey, ez);
            bisector_grad(nhead.x, nhead.y, nhead.z, head.x, head.y, head.z, R.Ox, R.Oy, R.Oz, R.dx, R.dy, R.dz, fx, fy, fz);
        }

        R.pgx = fma_(dL_dt0, ax, R.pgx);
        R.pgy = fma_(dL_dt0, ay, R.pgy);
        R.pgz = fma_(dL_dt0, az, R.pgz);
        R.cgx = R.cgx + fma_(dL_dt0, ex, dL_dt1 * bx);
        R.cgy = R.cgy + fma_(dL_dt0, ey, dL_dt1 * by);
        R.cgz = R.cgz + fma_(dL_dt0, ez, dL_dt1 * bz);
        R.ngx = fma_(dL_dt1, fx, R.ngx);
        R.ngy = fma_(dL_dt1, fy, R.ngy);
        R.ngz = fma_(dL_dt1, fz, R.ngz);
    }

    // what the reference adds with atomics at this point (pipeline.cu:305-328): prev_point_grad ->
    // points_grad[prev]; the SH row and dL/ds -> attr_grad[cur].  Exact zeros are not added.
    G.has = true;
    G.cur = cur;
    G.prev = R.prev;
    G.px = R.pgx;
    G.py = R.pgy;
    G.pz = R.pgz;
    G.pg_on = (R.prev != kNone) && (R.pgx != 0.0f || R.pgy != 0.0f || R.pgz != 0.0f);
    if (r == 0.0f) dLr = 0.0f;
    if (g == 0.0f) dLg = 0.0f;
    if (b == 0.0f) dLb = 0.0f;
    G.dLr = dLr;
    G.dLg = dLg;
    G.dLb = dLb;
    G.dL_ds = dL_ds;
    G.row = (dLr != 0.0f || dLg != 0.0f || dLb != 0.0f);

    R.ppx = head.x;
    R.ppy = head.y;
    R.ppz = head.z;
    R.prev = cur;
    R.pgx = R.cgx;
    R.pgy = R.cgy;
    R.pgz = R.cgz;
    R.cgx = R.ngx;
    R.cgy = R.ngy;
    R.cgz = R.ngz;
    R.ngx = R.ngy = R.ngz = 0.0f;
    R.T = Tn;
    return Tn > p.settings.weight_threshold;
}

__device__ __forceinline__ void clear_step(StepGrad &G) {
    G.has = G.row = G.pg_on = false;
    G.cur = 0;
    G.prev = kNone;
    G.dLr = G.dLg = G.dLb = G.dL_ds = G.px = G.py = G.pz = 0.0f;
}

template <int DEG, int MODE>
__device__ __forceinline__ void scatter_pending(const BwdParams &p, uint32_t lane, const float (&sh)[sh_dim(DEG)],
                                                StepGrad &G) {
#ifdef RF_EXPERIMENT_NO_SCATTER
    if (G.has && G.dL_ds == 123.456f) p.attr_grad[0] = G.dLr + G.dLg + G.dLb + G.px + G.py + G.pz;
    G.has = false;
    return;
#endif
    if (ballot(G.has) != 0ull) {
        scatter_step<DEG, MODE>(lane, G.has, G.row, G.cur, sh, G.dLr, G.dLg, G.dLb, G.dL_ds, G.pg_on, G.prev,
                                G.px, G.py, G.pz, p.attr_grad, p.points_grad, p.attr_pitch);
    }
    G.has = false;
    G.row = false;
    G.pg_on = false;
}

// Backward by re-walking (no trail available): same scan as forward.
template <int DEG, bool HALF, int MODE>
__global__ __launch_bounds__(kBlock) void backward_kernel(BwdParams p) {
    const uint32_t lane = threadIdx.x & 63u;

    uint32_t ray, slot;
    bool alive = map_ray(p.grid, ray, slot);
    const FoamView &fv = p.foam;
    constexpr int NB = sh_dim(DEG);
    // with a trail: this launch only handles the rays whose hops did not fit in it (the replay
    // kernel did the others)
    if (p.trail_hops) {
        if (alive && p.trail_hops[slot] <= p.trail_cap) alive = false;
        if (__syncthreads_or(alive ? 1 : 0) == 0) return;
    }

    BwdRay R;
    init_backward_ray(R);
    uint32_t cur = 0;
    if (alive) load_backward_ray<DEG, HALF>(p, ray, R, cur);
    float sh[NB];
    sh_basis<DEG>(R.dx, R.dy, R.dz, sh);
    const uint32_t max_steps = p.settings.max_intersections;

    uint32_t n = 0;
    uint32_t nb = 0, cnt = 0;
    float4 head = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (alive) {
        nb = fv.poff[cur];
        cnt = fv.poff[cur + 1] - nb;
        head = fv.cells[cur];
    }
    // Gradient contributions of the step just finished are scattered at the top of the NEXT
    // iteration: the atomics then complete under the long face scan (memory operations retire in
    // order, so a load issued after an atomic cannot complete before it).
    StepGrad G;
    clear_step(G);

    while (ballot(alive) != 0ull) {
        if (alive) {
            n++;
            if (n > max_steps) alive = false;
        }
        scatter_pending<DEG, MODE>(p, lane, sh, G);
        ScanResult sr;
        sr.t1 = __builtin_inff();
        sr.k = kNone;
        if (alive) {
            sr = scan_faces(fv.geo + (size_t)nb * 3u, cnt, head.x, head.y, head.z, R.Ox, R.Oy, R.Oz, R.dx, R.dy, R.dz);
            if (sr.k == kNone) alive = false;
        }
        uint32_t nxt = 0, nnb = 0, ncnt = 0;
        float4 nhead = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (alive) {
            const Link link = fv.link[nb + sr.k];
            nxt = link.nbr;
            nnb = link.first;
            ncnt = link.count;
            nhead = fv.cells[nxt];
        }
        if (alive) {
            const float t1 = sr.t1;
            if (t1 > R.t0) {
                if (!backward_segment<DEG, HALF>(p, R, sh, cur, head, nhead, t1, G)) alive = false;
            }
            R.t0 = __builtin_fmaxf(R.t0, t1);
            cur = nxt;
            head = nhead;
            nb = nnb;
            cnt = ncnt;
        }
    }
    scatter_pending<DEG, MODE>(p, lane, sh, G);
}

// Backward by replaying the trail trace_forward recorded for exactly these rays: hop i of a ray
// entered cell trail[i]; the t1 of the face crossed is recomputed with the same arithmetic (so it
// is the same float), and no face list is scanned nor any face table read.  Trail entries and the
// next cell's record are fetched two / one hops ahead.  A ray with more hops than the trail holds is skipped here and
// handled by a second launch of the re-walk kernel (backward_kernel), which takes only those.
template <int DEG, bool HALF, int MODE>
__global__ __launch_bounds__(kBlock) void backward_replay_kernel(BwdParams p) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t ray, slot;
    bool alive = map_ray(p.grid, ray, slot);
    const FoamView &fv = p.foam;
    constexpr int NB = sh_dim(DEG);
    const size_t slots = p.trail_slots;
    const uint32_t cap = p.trail_cap;

    BwdRay R;
    init_backward_ray(R);
    uint32_t cur = 0;
    uint32_t hops = 0;
    if (alive) {
        hops = p.trail_hops[slot];
        if (hops > cap) alive = false;   // did not fit in the trail: left to the re-walk launch
    }
    if (alive) load_backward_ray<DEG, HALF>(p, ray, R, cur);
    float sh[NB];
    sh_basis<DEG>(R.dx, R.dy, R.dz, sh);
    const uint32_t max_steps = p.settings.max_intersections;
    const uint32_t recorded = hops < cap ? hops : cap;   // hops present in the trail

    // pipeline registers: id1 = trail[i+1], id0 = trail[i], q0 = cells[id0]
    float4 head = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t id0 = 0, id1 = 0;
    if (alive) {
        head = fv.cells[cur];
        if (recorded > 0) id0 = p.trail[slot];
        if (recorded > 1) id1 = p.trail[slots + slot];
        if (recorded > 0) q0 = fv.cells[id0];
    }

    StepGrad G;
    clear_step(G);
    uint32_t i = 0;   // hop index
    uint32_t n = 0;
    // Order inside an iteration: (A) issue the prefetch loads, (B) compute the step from data
    // already in registers, (C) pin the prefetched values (the only wait), (D) scatter.  Memory
    // operations retire in order, so a load issued after an atomic cannot complete before it:
    // with this order the atomics of step i retire under the compute of step i+1 instead of
    // being waited for.
    while (ballot(alive) != 0ull) {
        if (alive) {
            n++;
            if (n > max_steps) alive = false;
        }
        if (alive && i >= hops) alive = false;   // forward stopped here (no exit face / step cap / opaque)
        // (A) prefetch: cell record of hop i+1, trail entry of hop i+2
        float4 q1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint32_t id2 = 0;
        if (alive) {
            if (i + 1 < recorded) q1 = fv.cells[id1];
            if (i + 2 < recorded) id2 = p.trail[(size_t)(i + 2) * slots + slot];
        }
        // (B) this hop: the face crossed is the bisector of (cur, id0); its fp16 offset is
        // recomputed from the two cell records exactly as rf_prepare_foam rounds it
        float4 nhead = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float t1 = 0.0f;
        if (alive) {
            nhead = q0;
            float ox, oy, oz;
            face_offset(head, nhead, ox, oy, oz);
            t1 = face_hit(ox, oy, oz, head.x, head.y, head.z, R.Ox, R.Oy, R.Oz, R.dx, R.dy, R.dz);
        }
        if (alive) {
            if (t1 > R.t0) {
                if (!backward_segment<DEG, HALF>(p, R, sh, cur, head, nhead, t1, G)) alive = false;
            }
            R.t0 = __builtin_fmaxf(R.t0, t1);
            cur = id0;
            head = nhead;
            i++;
        }
        // (C) rotate the pipeline; the asm pins make the loads complete HERE, before the atomics
        id0 = id1;
        id1 = id2;
        q0 = q1;
        asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(id1));
        // (D) scatter this hop's gradients
        scatter_pending<DEG, MODE>(p, lane, sh, G);
    }
}

// ------------------------------------------------------------------------------------------
// MODE 3: block-level gradient write-combining cache.
//
// Global fp32 atomics on this part execute at the memory side of the fabric (the XCD L2 drops the
// line), at a few 10^9 requests/s chip-wide -- far fewer than the (ray, step) contributions of a
// frame.  The four waves of a block walk one 16x16 pixel tile, i.e. the same few hundred cells
// within a few steps of each other, so their contributions are first summed in LDS:
//   * an open-addressing (kCacheProbes linear probes) table keyed by cell id; a row holds the cell's
//     3B colour gradients, its density gradient and its 3 point-gradient components, all as
//     DOUBLES, and every contribution is one ds_add_f64: no locks, no read-modify-write rounds;
//     lanes that hit the same address are serialised by the LDS itself (about 11 clocks per
//     lane).  Measured on gfx950 (scripts/probe/lds_atomics.hip, clocks per wave64 instruction):
//     ds_add_f64 9, ds_add_u32 5, but ds_add_f32 195 whatever the addresses -- the fp32 LDS atomic
//     is 20x slower than the fp64 one, which is what rules out rows of floats (the previous
//     design: float rows, colour updated by float4 read-modify-write under per-row locks, 8 lock
//     rounds per lit wave-step).  Sums are rounded to fp32 once, at the flush;
//   * lanes of a wave in the same cell are first merged in registers (DPP xor stages), which cuts
//     the lanes per address;
//   * rows of doubles are large (248 B at SH degree 2: 160 rows in the 40 KB a block may use at 4
//     blocks per CU; 424 B at degree 3: 124 rows at 3 blocks per CU), so the table is flushed often: every kEpoch steps the block synchronises and
//     evicts the rows that were not touched during the epoch (the walk has moved past those cells)
//     with one coalesced row of global atomics each, skipping zeros; rows still in use stay.
//     A lane whose key finds no free row adds straight to global memory instead (must stay rare:
//     scattered global atomics are the slow path this cache exists to avoid).
// Measured out in round 4 (profiles/r04/b_image_backward_refcount_cache_ab.log; the code is in commit 29902d7): the same
// cache WITHOUT block barriers -- a reference count per row, every wave sweeping a rotating quarter of the rows every
// four of its own steps -- runs the benchmark frame's backward in 4.18 ms against 3.75 (config 2: 3.42 against 3.06):
// two more LDS atomics and a key re-read per lane access cost more than the two barriers per epoch, and the waves of a
// tile, which share most of their cells, gain nothing from drifting apart.

#ifndef RF_CACHE_PROBES
#define RF_CACHE_PROBES 8
#endif
#ifndef RF_CACHE_EPOCH
#define RF_CACHE_EPOCH 4
#endif
#ifndef RF_ABSORB_STAGES
#define RF_ABSORB_STAGES 4
#endif
#ifndef RF_CACHE_ROWS_D2
#define RF_CACHE_ROWS_D2 160
#endif
#ifndef RF_BWD_WAVES
#define RF_BWD_WAVES 4
#endif
#ifndef RF_CACHE_ROWS_D3
#define RF_CACHE_ROWS_D3 124
#endif
#ifndef RF_BWD_WAVES_D3
#define RF_BWD_WAVES_D3 3
#endif
#ifndef RF_DTABLE_ROWS
#define RF_DTABLE_ROWS 768
#endif
#ifndef RF_STAGE_LANES_D3
#define RF_STAGE_LANES_D3 32
#endif
#ifndef RF_DIRECT_WAVES_D3
#define RF_DIRECT_WAVES_D3 3
#endif
#ifndef RF_MERGE_ROWS
#define RF_MERGE_ROWS 1
#endif
#ifndef RF_ROWS_FROM_BASIS
#define RF_ROWS_FROM_BASIS 1
#endif
constexpr int kCacheProbes = RF_CACHE_PROBES;
constexpr uint32_t kEpoch = RF_CACHE_EPOCH;

// One stage of the in-register pre-reduction that precedes the LDS adds: lanes l and l^BIT that
// both hold a contribution for the same key are merged into the lower lane; the upper lane drops
// out (`act` cleared).  Fewer lanes then hit the same LDS address.
template <int BIT, int NV>
__device__ __forceinline__ void absorb_stage(uint32_t lane, uint32_t key, bool &act, float (&v)[NV]) {
    const uint32_t kp = xor_lane_u<BIT>(key);
    const bool actp = xor_lane_u<BIT>(act ? 1u : 0u) != 0u;
    const bool same = act && actp && kp == key;
    if (ballot(same) == 0ull) return;
    const bool upper = (lane & (uint32_t)BIT) != 0u;
    const float m = (same && !upper) ? 1.0f : 0.0f;
    // only the lower lane of a pair receives (m = 1), so the value of lane + BIT is all that is needed: one DPP
    // read per value, written so that it folds into the multiply-add (v_fmac_f32_dpp) instead of a copy, an in-place
    // DPP move and a separate v_fmac -- three instructions per value per stage, the bulk of a lit wave-step
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = fma_(from_upper_lane<BIT>(v[i]), m, v[i]);
    if (same && upper) act = false;
}

template <int NB>
struct CacheLayout {
    static constexpr int NCOEF = 3 * NB;
    static constexpr int COL_DS = NCOEF;
    static constexpr int COL_PG = NCOEF + 1;
    static constexpr int NCOL = NCOEF + 4;
    static constexpr int STRIDE = NCOL | 1;   // odd number of doubles: columns spread over the banks
    static constexpr int ROWS = NB == 1 ? 256 : (NB == 4 ? 256 : (NB == 9 ? RF_CACHE_ROWS_D2 : RF_CACHE_ROWS_D3));
};

template <int ROWS>
__device__ __forceinline__ int cache_find(uint32_t *keys, uint32_t key) {
    const uint32_t h = (((key * 2654435761u) >> 16) * (uint32_t)ROWS) >> 16;
#pragma unroll
    for (int probe = 0; probe < kCacheProbes; ++probe) {
        uint32_t slot = h + (uint32_t)probe;
        slot = slot >= (uint32_t)ROWS ? slot - (uint32_t)ROWS : slot;
        const uint32_t old = atomicCAS(&keys[slot], kNone, key);
        if (old == kNone || old == key) return (int)slot;
    }
    return -1;
}

template <int NB>
__device__ __forceinline__ void cache_flush(double *rows, uint32_t *keys, uint8_t *touch, bool all,
                                              float *attr_grad, float *points_grad, uint32_t pitch) {
    using L = CacheLayout<NB>;
    constexpr int A = 1 + 3 * NB;
    const uint32_t lane = threadIdx.x & 63u, col0 = threadIdx.x & 31u, base = threadIdx.x & ~63u;
    const bool mine_exists = threadIdx.x < (uint32_t)L::ROWS;
    const uint32_t my_key = mine_exists ? keys[threadIdx.x] : kNone;
    const bool occupied = my_key != kNone;
    const bool evict = occupied && (all || touch[threadIdx.x] == (uint8_t)0);
    if (occupied && !evict) touch[threadIdx.x] = (uint8_t)0;
    if (evict) keys[threadIdx.x] = kNone;
    unsigned long long todo = ballot(evict);
    while (todo != 0ull) {
        const uint32_t b0 = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        uint32_t b1 = 64u;
        if (todo != 0ull) {
            b1 = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
        }
        const uint32_t mine = lane < 32u ? b0 : b1;
        const uint32_t key = __shfl(my_key, (int)(mine & 63u), 64);
        if (mine < 64u) {
            const uint32_t r = base + mine;
            for (uint32_t col = col0; col < (uint32_t)L::NCOL; col += 32u) {
                double *cell = rows + r * L::STRIDE + col;
                const float v = (float)*cell;
                if (v != 0.0f) {
                    *cell = 0.0;
                    float *dst;
                    if (col < (uint32_t)L::NCOEF) dst = attr_grad + (size_t)key * pitch + col;
                    else if (col == (uint32_t)L::COL_DS) dst = attr_grad + (size_t)key * pitch + (A - 1);
                    else dst = points_grad + 3 * (size_t)key + (col - (uint32_t)L::COL_PG);
                    grad_add(dst, v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The per-lane state of a trail replay and one hop of it, shared by the replay kernels below.
// Pipeline registers: id1 = trail[i+1], id0 = trail[i], q0 = cells[id0]; every call of step()
// issues the loads of hop i+1 / i+2 first and then computes hop i from registers.
template <int DEG, bool HALF, bool QUANT, bool PAIRS = false>
struct TrailWalker {
    static constexpr int NB = sh_dim(DEG);
    BwdRay R;
    float sh[NB];
    float4 head, q0;
    uint32_t cur, hops, recorded, i, n, id0, id1, slot;
    size_t slots;
    // (measured out in round 5: reading the trail through a wave-uniform row pointer + the lane's 32-bit offset, as the
    // forward writes it -- the image backward went from 3.76 to 3.92 ms, the flat one did not move)
    bool alive;
#ifdef RF_EXPERIMENT_SECTIONS
    unsigned long long sec_wait = 0, sec_segment = 0;   // wave clocks: until the hop's records are there / in backward_segment
#endif

    __device__ __forceinline__ void init(const BwdParams &p) {
        uint32_t ray;
        alive = map_ray(p.grid, ray, slot);
        const FoamView &fv = p.foam;
        slots = p.trail_slots;
        const uint32_t cap = p.trail_cap;
        init_backward_ray(R);
        cur = 0;
        hops = 0;
        if (alive) {
            hops = p.trail_hops[slot];
            if (hops > cap) alive = false;   // did not fit in the trail: left to the re-walk launch
        }
        if (alive) load_backward_ray<DEG, HALF>(p, ray, R, cur);
        sh_basis<DEG>(R.dx, R.dy, R.dz, sh);
        recorded = hops < cap ? hops : cap;
        head = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        id0 = id1 = 0;
        if (alive) {
            head = fv.cells[cur];
            if (recorded > 0) id0 = p.trail[slot];
            if (recorded > 1) id1 = p.trail[slots + slot];
            if (recorded > 0) q0 = fv.cells[id0];
        }
        i = 0;
        n = 0;
    }

    // one hop of every live lane; G receives the gradients of the segment just crossed (if any)
    __device__ __forceinline__ void step(const BwdParams &p, StepGrad &G) {
        const FoamView &fv = p.foam;
#ifdef RF_EXPERIMENT_SECTIONS
        const unsigned long long c0 = __builtin_readcyclecounter();
#endif
        if (alive) {
            n++;
            if (n > p.settings.max_intersections) alive = false;
        }
        if (alive && i >= hops) alive = false;   // forward stopped here (no exit face / step cap / opaque)
        float4 q1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint32_t id2 = 0;
        if (alive) {
            if (i + 1 < recorded) q1 = fv.cells[id1];
            if (i + 2 < recorded) id2 = p.trail[(size_t)(i + 2) * slots + slot];
        }
        // the face crossed is the bisector of (cur, id0); its fp16 offset is recomputed from the two
        // cell records exactly as rf_prepare_foam rounds it
        float4 nhead = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float t1 = 0.0f;
        if (alive) {
            nhead = q0;
            float ox, oy, oz;
            face_offset(head, nhead, ox, oy, oz);
            t1 = face_hit(ox, oy, oz, head.x, head.y, head.z, R.Ox, R.Oy, R.Oz, R.dx, R.dy, R.dz);
        }
#ifdef RF_EXPERIMENT_SECTIONS
        asm volatile("" : "+v"(t1));
        const unsigned long long c1 = __builtin_readcyclecounter();
        sec_wait += c1 - c0;
#endif
        if (alive) {
            if (t1 > R.t0) {
                if (!backward_segment<DEG, HALF, QUANT, PAIRS>(p, R, sh, cur, head, nhead, t1, G)) alive = false;
            }
#ifdef RF_EXPERIMENT_SECTIONS
            asm volatile("" : "+v"(G.dL_ds));
#endif
            R.t0 = __builtin_fmaxf(R.t0, t1);
            cur = id0;
            head = nhead;
            i++;
        }
#ifdef RF_EXPERIMENT_SECTIONS
        sec_segment += __builtin_readcyclecounter() - c1;
#endif
        id0 = id1;
        id1 = id2;
        q0 = q1;
    }
};

template <int DEG, bool HALF, bool QUANT>
__global__ __launch_bounds__(kBlock, (DEG <= 2 ? RF_BWD_WAVES : RF_BWD_WAVES_D3)) void backward_replay_cached_kernel(BwdParams p) {
    constexpr int NB = sh_dim(DEG);
    constexpr int A = 1 + 3 * NB;
    using L = CacheLayout<NB>;
    constexpr int STRIDE = L::STRIDE;
    constexpr int ROWS = L::ROWS;
    __shared__ __attribute__((aligned(16))) double s_rows[ROWS * STRIDE];
    __shared__ uint32_t s_keys[ROWS];
    __shared__ uint8_t s_touch[ROWS];
    for (uint32_t i = threadIdx.x; i < (uint32_t)(ROWS * STRIDE); i += kBlock) s_rows[i] = 0.0;
    for (uint32_t i = threadIdx.x; i < (uint32_t)ROWS; i += kBlock) {
        s_keys[i] = kNone;
        s_touch[i] = (uint8_t)0;
    }
    __syncthreads();

    TrailWalker<DEG, HALF, QUANT> W;
    W.init(p);
    const float (&sh)[NB] = W.sh;

    StepGrad G;
    clear_step(G);
    uint32_t it = 0;
    bool block_alive = true;
#ifdef RF_EXPERIMENT_SECTIONS
    unsigned long long sec_cache = 0, sec_flush = 0, sec_steps = 0, sec_total = __builtin_readcyclecounter();
#endif
    while (block_alive) {
        if (ballot(W.alive) != 0ull) {
            W.step(p, G);
#ifdef RF_EXPERIMENT_SECTIONS
            sec_steps++;
            const unsigned long long e0 = __builtin_readcyclecounter();
#endif

            const uint32_t lane = threadIdx.x & 63u;
            if (ballot(G.has) != 0ull) {
                bool act = G.has;
                if (ballot(G.has && G.row) != 0ull) {
                    // lit wave-step: colour + density of every lane with a contribution
                    float v[A];
#pragma unroll
                    for (int k = 0; k < 3 * NB; ++k) {
                        float gc = (k % 3 == 0) ? G.dLr : ((k % 3 == 1) ? G.dLg : G.dLb);
                        v[k] = G.row ? sh[k / 3] * gc : 0.0f;
                    }
                    v[A - 1] = G.dL_ds;
                    if constexpr (RF_ABSORB_STAGES > 0) absorb_stage<1, A>(lane, G.cur, act, v);
                    if constexpr (RF_ABSORB_STAGES > 1) absorb_stage<2, A>(lane, G.cur, act, v);
                    if constexpr (RF_ABSORB_STAGES > 2) absorb_stage<4, A>(lane, G.cur, act, v);
                    if constexpr (RF_ABSORB_STAGES > 3) absorb_stage<8, A>(lane, G.cur, act, v);
                    const int s_row = act ? cache_find<ROWS>(s_keys, G.cur) : -1;
                    if (act && s_row >= 0) {
                        s_touch[s_row] = (uint8_t)1;
                        double *row = s_rows + s_row * STRIDE;
                        atomicAdd(row + L::COL_DS, (double)v[A - 1]);
                        bool colour = false;
#pragma unroll
                        for (int k = 0; k < 3 * NB; ++k) colour = colour || (v[k] != 0.0f);
                        if (colour) {
#pragma unroll
                            for (int k = 0; k < 3 * NB; ++k) atomicAdd(row + k, (double)v[k]);
                        }
                    } else if (act) {
                        float *dst = p.attr_grad + (size_t)G.cur * p.attr_pitch;
#pragma unroll
                        for (int k = 0; k < A; ++k)
                            if (v[k] != 0.0f) grad_add(dst + k, v[k]);
                    }
                } else {
                    const int s_row = act ? cache_find<ROWS>(s_keys, G.cur) : -1;
                    if (act && s_row >= 0) {
                        s_touch[s_row] = (uint8_t)1;
                        atomicAdd(s_rows + s_row * STRIDE + L::COL_DS, (double)G.dL_ds);
                    } else if (act) {
                        grad_add(p.attr_grad + (size_t)G.cur * p.attr_pitch + (A - 1), G.dL_ds);
                    }
                }
                const bool pact = G.has && G.pg_on;
                if (ballot(pact) != 0ull) {
                    const int s_pg = pact ? cache_find<ROWS>(s_keys, G.prev) : -1;
                    if (pact && s_pg >= 0) {
                        s_touch[s_pg] = (uint8_t)1;
                        double *dst = s_rows + s_pg * STRIDE + L::COL_PG;
                        atomicAdd(dst + 0, (double)G.px);
                        atomicAdd(dst + 1, (double)G.py);
                        atomicAdd(dst + 2, (double)G.pz);
                    } else if (pact) {
                        float *dst = p.points_grad + 3 * (size_t)G.prev;
                        grad_add(dst + 0, G.px);
                        grad_add(dst + 1, G.py);
                        grad_add(dst + 2, G.pz);
                    }
                }
            }
            G.has = false;
            G.row = false;
            G.pg_on = false;
#ifdef RF_EXPERIMENT_SECTIONS
            sec_cache += __builtin_readcyclecounter() - e0;
#endif
        }
        it++;
        if ((it & (kEpoch - 1u)) == 0u) {
#ifdef RF_EXPERIMENT_SECTIONS
            const unsigned long long f0 = __builtin_readcyclecounter();
#endif
            block_alive = __syncthreads_or(W.alive ? 1 : 0) != 0;
            cache_flush<NB>(s_rows, s_keys, s_touch, !block_alive, p.attr_grad, p.points_grad, p.attr_pitch);
            __syncthreads();
#ifdef RF_EXPERIMENT_SECTIONS
            sec_flush += __builtin_readcyclecounter() - f0;
#endif
        }
    }
#ifdef RF_EXPERIMENT_SECTIONS
    // wave clocks per section: [8] waiting for the hop's records + face hit, [9] backward_segment, [10] merge + cache
    // updates of the step, [11] barriers + flush of the epochs, [12] the whole walk, [13] wave-steps
    if (p.stats && (threadIdx.x & 63u) == 0u) {
        atomicAdd(p.stats + 8, W.sec_wait);
        atomicAdd(p.stats + 9, W.sec_segment);
        atomicAdd(p.stats + 10, sec_cache);
        atomicAdd(p.stats + 11, sec_flush);
        atomicAdd(p.stats + 12, (unsigned long long)__builtin_readcyclecounter() - sec_total);
        atomicAdd(p.stats + 13, sec_steps);
    }
#endif
}

// ------------------------------------------------------------------------------------------
// Lock-free write-back cache entry {cell id, fp32 sum} in LDS, one 64-bit word (backward_replay_direct_kernel).
constexpr unsigned long long kEmptyEntry = (unsigned long long)kNone << 32;

// sum[key] += v in a direct-mapped table of ROWS entries; an entry that holds another cell is taken over and its sum
// sent to memory (dst[stride * cell]) as one atomic.  One compare-and-swap per attempt: whatever other lanes or waves
// do to the entry in between only makes the attempt repeat.
template <int ROWS>
__device__ __forceinline__ void table_add(unsigned long long *tab, uint32_t key, float v, float *dst, size_t stride) {
    const uint32_t slot = (((key * 2654435761u) >> 16) * (uint32_t)ROWS) >> 16;
    unsigned long long old = tab[slot];
    for (;;) {
        const uint32_t okey = (uint32_t)(old >> 32);
        const float sum = okey == key ? __builtin_bit_cast(float, (uint32_t)old) + v : v;
        const unsigned long long want = ((unsigned long long)key << 32) | (unsigned long long)__builtin_bit_cast(uint32_t, sum);
        const unsigned long long seen = atomicCAS(&tab[slot], old, want);
        if (seen == old) {
            if (okey != key && okey != kNone) {
                const float evicted = __builtin_bit_cast(float, (uint32_t)old);
                if (evicted != 0.0f) grad_add(dst + stride * (size_t)okey, evicted);
            }
            return;
        }
        old = seen;
    }
}

// table_add for up to N (key, value) pairs of a lane at once, each into its own table: the N reads go out together, then
// the N compare-and-swaps, and only the pairs whose swap lost are tried again -- one LDS round trip per attempt for all of
// a step's scalars (density gradient and the three point-gradient components) instead of one each.
template <int ROWS, int N>
__device__ __forceinline__ void table_add_n(unsigned long long (*tab)[ROWS], const uint32_t (&key)[N], const float (&v)[N],
                                            bool (&todo)[N], float *const (&dst)[N], const size_t (&stride)[N]) {
    uint32_t slot[N];
    unsigned long long old[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        slot[i] = (((key[i] * 2654435761u) >> 16) * (uint32_t)ROWS) >> 16;
        old[i] = todo[i] ? tab[i][slot[i]] : 0ull;
    }
    bool any = false;
#pragma unroll
    for (int i = 0; i < N; ++i) any |= todo[i];
    while (ballot(any) != 0ull) {
        unsigned long long seen[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            seen[i] = old[i];
            if (todo[i]) {
                const uint32_t okey = (uint32_t)(old[i] >> 32);
                const float sum = okey == key[i] ? __builtin_bit_cast(float, (uint32_t)old[i]) + v[i] : v[i];
                const unsigned long long want =
                    ((unsigned long long)key[i] << 32) | (unsigned long long)__builtin_bit_cast(uint32_t, sum);
                seen[i] = atomicCAS(&tab[i][slot[i]], old[i], want);
            }
        }
        any = false;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (todo[i]) {
                if (seen[i] == old[i]) {
                    const uint32_t okey = (uint32_t)(old[i] >> 32);
                    if (okey != key[i] && okey != kNone) {
                        const float evicted = __builtin_bit_cast(float, (uint32_t)old[i]);
                        if (evicted != 0.0f) grad_add(dst[i] + stride[i] * (size_t)okey, evicted);
                    }
                    todo[i] = false;
                } else {
                    old[i] = seen[i];
                    any = true;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// MODE 4: trail replay with direct, row-coalesced global atomics -- for flat (training-like) batches.
//
// A sparse batch leaves little to combine: even in the coherent order of rf_build_ray_order the 64
// rays of a wave sit in 40-50 different cells, a cell is visited by two or three rays of the whole
// batch, and the block cache turns into a queue of rows that are written once and flushed (or do
// not fit: measured 26 ms for 1M rays on the 2M-point foam against 4.2 ms for a dense 2M-ray frame).
// What the memory side is good at (scripts/probe/global_atomics.hip): 21 G scattered fp32 atomics
// per second when every request is its own cache line, 3.4 G when successive instructions hit the
// same lines (a lane walking along its row).  So here every contribution goes straight to memory,
// shaped for the atomic unit: a colour row as ONE instruction whose lanes are the row's columns -- the
// lanes' rows are transposed through a small LDS staging area of the wave (written lane-major, read
// column-major).  The scalars per cell -- the density gradient, one value per segment and 10 of the 13 ms
// when sent directly, and the three point-gradient components -- go through block-level write-back tables
// (table_add above; same-cell lanes pre-merged by DPP).
#ifndef RF_JOINT_TABLES
#define RF_JOINT_TABLES 0     // measured: 15.35 ms against 14.57 (every segment lit), 6.22 against 5.43: the retries of one table hold the other three
#endif
#ifndef RF_ROW_WINDOW
#define RF_ROW_WINDOW 4      // steps whose colour rows are merged before they are emitted (1, 2, 3, 4, 6 or 8)
#endif
template <int DEG, bool HALF, bool QUANT>
__global__ __launch_bounds__(kBlock, (DEG <= 2 ? 4 : RF_DIRECT_WAVES_D3)) void backward_replay_direct_kernel(BwdParams p) {
    constexpr int NB = sh_dim(DEG);
    constexpr int A = 1 + 3 * NB;
    constexpr int NC = 3 * NB;
    constexpr int SHP = (NC + 3) & ~3;             // staged floats per lane, float4-padded
    constexpr int PITCH = SHP;
    // Lanes of a wave staged at a time.  The 48-float rows of SH degree 3 staged for all 64 lanes are 48 KB per
    // block: with the density table two blocks fit a CU (2 waves/SIMD, and every cell gather of a sparse batch
    // is its own cache miss to hide).  Staging one half-wave after the other halves that (4 blocks per CU).
    constexpr int STAGE_LANES = DEG >= 2 ? RF_STAGE_LANES_D3 : 64;   // 64, 32 or 16
    // Rows wider than a half-wave (SH degree 3: 48 columns) are not staged at all: a gradient row is the outer product
    // basis(ray) x dL/drgb, and the ray's basis never changes -- every lane leaves its NB basis values in LDS ONCE, and
    // the lanes that emit a row (one per column) rebuild it from there and from three wave-broadcast scalars.  No
    // per-step staging writes, 16 KB instead of 24 KB per block, and the rows of ALL 64 lanes can be merged by cell.
    constexpr bool FROM_BASIS = (RF_ROWS_FROM_BASIS != 0) && NC > 32;
    constexpr int STAGE_FLOATS = FROM_BASIS ? (kBlock / 64) * 64 * NB : (kBlock / 64) * STAGE_LANES * PITCH;
    __shared__ __attribute__((aligned(16))) float s_stage[STAGE_FLOATS];
    const uint32_t lane = threadIdx.x & 63u;
    float *stage = s_stage + (threadIdx.x >> 6) * (STAGE_FLOATS / (kBlock / 64));   // this wave's slots
    // The scalar-per-cell contributions -- the density gradient (one per segment: every cell a ray crosses has one,
    // empty cells included) and the three point-gradient components (one triple per lit segment, for the cell before) --
    // are first summed per cell in block-level write-back caches: four direct-mapped tables (density, x, y, z) of
    // 64-bit entries {cell, fp32 sum}, updated by one 64-bit compare-and-swap -- add when the entry holds the lane's
    // cell, take the entry (and send what it held to memory as one atomic) when it holds another.  No locks, no epochs,
    // and above all NO BLOCK BARRIER in the walk: round 2's table was flushed by all four waves together every four
    // steps, which tied the waves of a block into lockstep -- twelve waves per CU hid latency like three (the launch
    // ran 6.1 ms of its 9.9 with every global atomic compiled out).  A block's rays meet a cell about ten times
    // (scripts/model_train_batch.py), so the tables remove nearly all of these requests.
    // 768 entries per table where four blocks share a CU (SH degree <= 2: 14 KB of staging + 24.6 KB of tables); 1024 at
    // degree 3, whose three blocks per CU have the room since the rows are rebuilt from the basis (16 KB + 32 KB): backward
    // of the training batch 5.04 -> 4.89 ms, every segment lit 14.97 -> 14.54; 512: 5.25 / 15.7; 1152: as 1024
    // (profiles/r04/i_table_rows_ab.log)
    constexpr int kRowWindow = RF_ROW_WINDOW;
    constexpr int DROWS = DEG >= 3 ? (RF_DTABLE_ROWS > 1024 ? RF_DTABLE_ROWS : 1024) : RF_DTABLE_ROWS;
    __shared__ unsigned long long s_tab[4][DROWS];
    for (uint32_t e = threadIdx.x; e < (uint32_t)(4 * DROWS); e += kBlock) (&s_tab[0][0])[e] = kEmptyEntry;
    __syncthreads();

    TrailWalker<DEG, HALF, QUANT, true> W;
    W.init(p);
    const float (&sh)[NB] = W.sh;
    if constexpr (FROM_BASIS) {
#pragma unroll
        for (int b = 0; b < NB; ++b) stage[lane * NB + b] = sh[b];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    StepGrad G;
    clear_step(G);
    // the steps whose colour rows are still to be emitted (FROM_BASIS with a window): per lane the cell and dL/drgb of each
    // step, per step the mask of its lit lanes
    uint32_t hc[kRowWindow];
    float hr[kRowWindow], hg[kRowWindow], hb[kRowWindow];
    unsigned long long hm[kRowWindow];
#pragma unroll
    for (int w = 0; w < kRowWindow; ++w) {
        hc[w] = 0u;
        hr[w] = hg[w] = hb[w] = 0.0f;
        hm[w] = 0ull;
    }
    uint32_t wi = 0;
#ifdef RF_EXPERIMENT_SECTIONS
    unsigned long long sec_tables = 0, sec_rows = 0, sec_steps = 0, sec_total = __builtin_readcyclecounter();
#endif
    while (ballot(W.alive) != 0ull) {
        W.step(p, G);
#ifdef RF_EXPERIMENT_SECTIONS
        sec_steps++;
        const unsigned long long e0 = __builtin_readcyclecounter();
        unsigned long long e1 = e0;
#endif

        if (ballot(G.has) != 0ull) {
            // density gradient and the point gradient of the previous cell: lanes of the wave in the same cell merged (DPP
            // xor stages), then all four scalars into the block's tables in one joint update
            {
                bool dact = G.has;
                float dv[1] = {G.dL_ds};
                absorb_stage<1, 1>(lane, G.cur, dact, dv);
                absorb_stage<2, 1>(lane, G.cur, dact, dv);
                absorb_stage<4, 1>(lane, G.cur, dact, dv);
                absorb_stage<8, 1>(lane, G.cur, dact, dv);
                bool pact = G.has && G.pg_on;
                float pv[3] = {G.px, G.py, G.pz};
                if (ballot(pact) != 0ull) {
                    absorb_stage<1, 3>(lane, G.prev, pact, pv);
                    absorb_stage<2, 3>(lane, G.prev, pact, pv);
                    absorb_stage<4, 3>(lane, G.prev, pact, pv);
                    absorb_stage<8, 3>(lane, G.prev, pact, pv);
                }
#if RF_JOINT_TABLES
                const uint32_t keys[4] = {G.cur, G.prev, G.prev, G.prev};
                const float vals[4] = {dv[0], pv[0], pv[1], pv[2]};
                bool todo[4] = {dact && dv[0] != 0.0f, pact && pv[0] != 0.0f, pact && pv[1] != 0.0f, pact && pv[2] != 0.0f};
                float *const dsts[4] = {p.attr_grad + (A - 1), p.points_grad + 0, p.points_grad + 1, p.points_grad + 2};
                const size_t strides[4] = {(size_t)p.attr_pitch, (size_t)3, (size_t)3, (size_t)3};
                table_add_n<DROWS, 4>(s_tab, keys, vals, todo, dsts, strides);
#else
                if (dact && dv[0] != 0.0f)
                    table_add<DROWS>(s_tab[0], G.cur, dv[0], p.attr_grad + (A - 1), (size_t)p.attr_pitch);
                if (pact) {
                    if (pv[0] != 0.0f) table_add<DROWS>(s_tab[1], G.prev, pv[0], p.points_grad + 0, (size_t)3);
                    if (pv[1] != 0.0f) table_add<DROWS>(s_tab[2], G.prev, pv[1], p.points_grad + 1, (size_t)3);
                    if (pv[2] != 0.0f) table_add<DROWS>(s_tab[3], G.prev, pv[2], p.points_grad + 2, (size_t)3);
                }
#endif
            }
#ifdef RF_EXPERIMENT_SECTIONS
            e1 = __builtin_readcyclecounter();
#endif
            // colour rows: stage lane-major, emit column-major (two rows per pass, a lane per column)
            const bool lit = G.has && G.row;
            if constexpr (FROM_BASIS && kRowWindow > 1) {
                // recorded below, emitted every kRowWindow steps
            } else if constexpr (FROM_BASIS) {
                // (Measured out in round 4, profiles/r04/e_row_emission_pipelined_ab.log: the members of all groups in ONE
                // loop with the next member's basis value requested before this one's is used -- 5.05 against 5.00 ms, the
                // wait counter at the loop's back edge makes the compiler wait for the early read anyway.)
                unsigned long long todo = ballot(lit);
                const uint32_t bcol = lane / 3u, ccol = lane - 3u * bcol;   // column `lane` = basis bcol, channel ccol
                while (todo != 0ull) {
                    const int src = __builtin_ctzll(todo);
                    const uint32_t cell = readlane(G.cur, src);
#if RF_MERGE_ROWS
                    unsigned long long group = ballot(lit && G.cur == cell);   // every lane of the wave in this cell
#else
                    unsigned long long group = 1ull << src;
#endif
                    todo &= ~group;
                    float v = 0.0f;
                    while (group != 0ull) {
                        const int m = __builtin_ctzll(group);
                        group &= group - 1ull;
                        const float gr = readlane_f(G.dLr, m), gg = readlane_f(G.dLg, m), gb = readlane_f(G.dLb, m);
                        const float gc = ccol == 0u ? gr : (ccol == 1u ? gg : gb);
                        if (lane < (uint32_t)NC) v += stage[(uint32_t)m * NB + bcol] * gc;
                    }
                    if (lane < (uint32_t)NC && v != 0.0f) grad_add(p.attr_grad + (size_t)cell * p.attr_pitch + lane, v);
                }
            } else if (ballot(lit) != 0ull) {
#pragma unroll
                for (int half = 0; half < 64 / STAGE_LANES; ++half) {
                    const bool mine = lit && (STAGE_LANES == 64 || (int)(lane / (uint32_t)STAGE_LANES) == half);
                    unsigned long long todo = ballot(mine);
                    if (todo == 0ull) continue;
                    const uint32_t sl = lane & (uint32_t)(STAGE_LANES - 1);   // staging slot of this lane
                    if (mine) {
                        float4 *dst4 = reinterpret_cast<float4 *>(stage + sl * PITCH);
#pragma unroll
                        for (int j = 0; j < SHP / 4; ++j) {
                            float x[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int k = 4 * j + c;
                                const float gc = (k % 3 == 0) ? G.dLr : ((k % 3 == 1) ? G.dLg : G.dLb);
                                x[c] = k < NC ? sh[(k < NC ? k : 0) / 3] * gc : 0.0f;
                            }
                            dst4[j] = make_float4(x[0], x[1], x[2], x[3]);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // Lanes of the wave (of this staged half) whose rows go to the SAME cell leave as one row: the emitting
                    // lanes sum the group's staged rows column by column (the same LDS reads as emitting them one by one)
                    // and issue one atomic per column.  What the memory side counts is (instruction, 64-byte line) pairs
                    // -- 21 G/s chip-wide, scripts/probe/global_atomics.hip -- and a batch whose every segment is lit
                    // (real training: the scene's softplus never returns exactly 0) is bound by exactly that.
                    if constexpr (NC > 32) {
                        // a row wider than a half-wave goes out as ONE instruction, a lane per column: split at column
                        // 32 the line that holds the boundary was requested twice (5 requests per 192-byte row, 4 now)
                        while (todo != 0ull) {
                            const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                            const uint32_t cell = readlane(G.cur, (int)src);
#if RF_MERGE_ROWS
                            unsigned long long group = ballot(mine && G.cur == cell);
#else
                            unsigned long long group = 1ull << src;
#endif
                            todo &= ~group;
                            if (lane < (uint32_t)NC) {
                                float v = 0.0f;
                                while (group != 0ull) {
                                    const uint32_t m = (uint32_t)__builtin_ctzll(group);
                                    group &= group - 1ull;
                                    v += stage[(m & (uint32_t)(STAGE_LANES - 1)) * PITCH + lane];
                                }
                                if (v != 0.0f) grad_add(p.attr_grad + (size_t)cell * p.attr_pitch + lane, v);
                            }
                        }
                    } else {
                        const uint32_t col0 = lane & 31u;
                        while (todo != 0ull) {
                            const uint32_t b0 = (uint32_t)__builtin_ctzll(todo);
                            const uint32_t cell0 = readlane(G.cur, (int)b0);
#if RF_MERGE_ROWS
                            const unsigned long long g0 = ballot(mine && G.cur == cell0);
#else
                            const unsigned long long g0 = 1ull << b0;
#endif
                            todo &= ~g0;
                            unsigned long long g1 = 0ull;
                            uint32_t cell1 = 0u;
                            if (todo != 0ull) {
                                const uint32_t b1 = (uint32_t)__builtin_ctzll(todo);
                                cell1 = readlane(G.cur, (int)b1);
#if RF_MERGE_ROWS
                                g1 = ballot(mine && G.cur == cell1);
#else
                                g1 = 1ull << b1;
#endif
                                todo &= ~g1;
                            }
                            unsigned long long group = lane < 32u ? g0 : g1;   // the group this half-wave emits
                            const uint32_t cell = lane < 32u ? cell0 : cell1;
                            if (col0 < (uint32_t)NC) {
                                float v = 0.0f;
                                while (group != 0ull) {
                                    const uint32_t m = (uint32_t)__builtin_ctzll(group);
                                    group &= group - 1ull;
                                    v += stage[(m & (uint32_t)(STAGE_LANES - 1)) * PITCH + col0];
                                }
                                if (v != 0.0f) grad_add(p.attr_grad + (size_t)cell * p.attr_pitch + col0, v);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();   // the slots are rewritten by the next half / next step
                }
            }
        }
        if constexpr (FROM_BASIS && kRowWindow > 1) {
            // Colour rows merged over TIME as well as over the wave: the lanes of a wave reach a cell a step or two after
            // one another, so the (lane, step) pairs of kRowWindow consecutive steps hold far fewer distinct cells than the
            // steps taken one by one -- on the training batch 0.56 rows per lit segment emitted step by step, 0.43 / 0.36 /
            // 0.32 over 2 / 3 / 4 steps (scripts/model_row_window.py) -- and with every segment lit the launch is bound by
            // the rows the memory side's atomic unit takes (section 4.3 of DESIGN.md).  A step leaves {cell, dL/drgb} in
            // four registers of its lane and its lit lanes as a wave mask; every kRowWindow steps (and when the wave's last
            // ray ends) the emitting lanes, one per column, rebuild every distinct cell's row from the rays' bases in LDS.
            // Measured (profiles/r05/b_row_window_ab.log; backward of the training batch, every segment lit / 16 % lit):
            // 14.75 / 4.93 ms step by step, 12.66 / 4.51 over 2 steps, 12.11 over 3, 11.83 / 4.44 over 4, 11.86 over 6.
            // Measured out beside it: the members of a row taken four at a time with their LDS reads in flight together
            // (14.57: a row's members are spread over the window's steps, whose registers differ, so most batches are
            // mostly padding); the window's dL/drgb in LDS with one member list per row (15.75: 12 KB of LDS per block at
            // the tables' expense and a scalar gather per member); the step's four table updates as one joint
            // compare-and-swap round (15.35: the retries of one table hold the other three); two / four rows in flight with
            // their member loops interleaved (12.12 / 14.02 against 11.61, r_rows_in_flight_ab.log) -- which showed that the
            // emission is bound by what it ISSUES, not by a member's LDS round trip.  Hence the member loop below: 13
            // instructions per member instead of 24 (s_lean_member_loop_ab.log: 11.62 -> 10.56 / 4.38 -> 4.05 ms).
            const bool litw = G.has && G.row;
            const unsigned long long lm = ballot(litw);
            switch (wi) {
#define RF_RECORD(k) case k: hc[k] = G.cur; hr[k] = G.dLr; hg[k] = G.dLg; hb[k] = G.dLb; hm[k] = lm; break;
                RF_RECORD(0)
#if RF_ROW_WINDOW > 1
                RF_RECORD(1)
#endif
#if RF_ROW_WINDOW > 2
                RF_RECORD(2)
#endif
#if RF_ROW_WINDOW > 3
                RF_RECORD(3)
#endif
#if RF_ROW_WINDOW > 4
                RF_RECORD(4)
                RF_RECORD(5)
#endif
#if RF_ROW_WINDOW > 6
                RF_RECORD(6)
                RF_RECORD(7)
#endif
#undef RF_RECORD
                default: break;
            }
            wi++;
            if (wi == (uint32_t)kRowWindow || ballot(W.alive) == 0ull) {
                wi = 0;
                const uint32_t bcol = lane / 3u, ccol = lane - 3u * bcol;   // column `lane` = basis bcol, channel ccol
                // The member loop is what the emission issues most of (3.2 members per row), so it is kept to one LDS read
                // and three FMAs whose gradient operand is the scalar register v_readlane wrote: every lane keeps all three
                // channels' sums and picks its own once per row (a per-member select costs three moves and two selects),
                // and lanes past the row read a clamped word instead of being masked off around the read.
                const float *srow = stage + (bcol < (uint32_t)NB ? bcol : (uint32_t)NB - 1u);
#pragma unroll
                for (int w = 0; w < kRowWindow; ++w) {
                    while (hm[w] != 0ull) {
                        const int src = __builtin_ctzll(hm[w]);
                        const uint32_t cell = readlane(hc[w], src);
                        float vr = 0.0f, vg = 0.0f, vb = 0.0f;
#pragma unroll
                        for (int u = w; u < kRowWindow; ++u) {
                            unsigned long long group = hm[u] & ballot(hc[u] == cell);   // step u's lit lanes in this cell
                            hm[u] &= ~group;
                            while (group != 0ull) {
                                const int m = __builtin_ctzll(group);
                                group &= ~(1ull << m);
                                const float x = srow[(uint32_t)m * NB];
                                vr = fma_(x, readlane_f(hr[u], m), vr);
                                vg = fma_(x, readlane_f(hg[u], m), vg);
                                vb = fma_(x, readlane_f(hb[u], m), vb);
                            }
                        }
                        const float v = ccol == 0u ? vr : (ccol == 1u ? vg : vb);
                        if (lane < (uint32_t)NC && v != 0.0f) grad_add(p.attr_grad + (size_t)cell * p.attr_pitch + lane, v);
                    }
                }
            }
        }
        G.has = false;
        G.row = false;
        G.pg_on = false;
#ifdef RF_EXPERIMENT_SECTIONS
        sec_tables += e1 - e0;
        sec_rows += __builtin_readcyclecounter() - e1;
#endif
    }
#ifdef RF_EXPERIMENT_SECTIONS
    // wave clocks per section: [8] waiting for the hop's records + face hit, [9] backward_segment (colour row gather and
    // the segment's arithmetic), [10] density / point-gradient tables, [11] colour rows staged and sent, [12] the whole
    // walk, [13] wave-steps
    if (p.stats && lane == 0u) {
        atomicAdd(p.stats + 8, W.sec_wait);
        atomicAdd(p.stats + 9, W.sec_segment);
        atomicAdd(p.stats + 10, sec_tables);
        atomicAdd(p.stats + 11, sec_rows);
        atomicAdd(p.stats + 12, (unsigned long long)__builtin_readcyclecounter() - sec_total);
        atomicAdd(p.stats + 13, sec_steps);
    }
#endif
    // every wave of the block is done: what the tables still hold goes to memory
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < (uint32_t)(4 * DROWS); e += kBlock) {
        const unsigned long long ent = (&s_tab[0][0])[e];
        const uint32_t key = (uint32_t)(ent >> 32);
        const float v = __builtin_bit_cast(float, (uint32_t)ent);
        if (key == kNone || v == 0.0f) continue;
        const uint32_t t = e / (uint32_t)DROWS;
        if (t == 0u) grad_add(p.attr_grad + (size_t)key * p.attr_pitch + (A - 1), v);
        else grad_add(p.points_grad + 3 * (size_t)key + (t - 1u), v);
    }
}

// ------------------------------------------------------------------------------------------
// foam packing

__device__ __forceinline__ uint2 pack_diff(float dx, float dy, float dz) {
    uint32_t lo = (uint32_t)float_to_half_bits(dx) | ((uint32_t)float_to_half_bits(dy) << 16);
    uint32_t hi = (uint32_t)float_to_half_bits(dz);
    return make_uint2(lo, hi);
}

// Padding entry at position j (1..3) of a list's last block, whose first entry (always a real face) has the fp16 offset
// (hx, hy, hz): the same offset times 2^j (exact in fp16) -- a bisector 2^j times as far along the same normal, which the
// scan can treat like any other face and which never wins (see "the face scan" above).  All-zero when the scaled offset
// leaves the fp16 range (offsets of 8190 and more: not a foam the fp16 table could hold anyway).
__device__ __forceinline__ void pad_offset(uint16_t hx, uint16_t hy, uint16_t hz, uint32_t j, uint16_t &px, uint16_t &py,
                                           uint16_t &pz) {
    const float k = (float)(1u << j);
    px = float_to_half_bits(half_lo(hx) * k);
    py = float_to_half_bits(half_lo(hy) * k);
    pz = float_to_half_bits(half_lo(hz) * k);
    const bool over = ((px & 0x7C00u) == 0x7C00u) | ((py & 0x7C00u) == 0x7C00u) | ((pz & 0x7C00u) == 0x7C00u);
    if (over) px = py = pz = (uint16_t)0;
}

// ---- padded offsets: poff[i] = sum_{j<i} round_up4(offsets[j+1] - offsets[j]) -------------------
// Three small launches: per-chunk sums (kScanChunk cells per 256-thread block, 4 cells per
// thread), one block scanning the chunk sums, then the per-cell exclusive scan inside each chunk.

__device__ __forceinline__ uint32_t padded_count(const uint32_t *__restrict__ offsets, uint32_t i, uint32_t n) {
    return i < n ? ((offsets[i + 1] - offsets[i] + 3u) & ~3u) : 0u;
}

// exclusive scan of one value per thread over a 256-thread block; returns the block total
__device__ __forceinline__ uint32_t block_scan256(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t up = __shfl_up(inc, d, 64);
        if ((int)lane >= d) inc += up;
    }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += s_wave[w];
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return base + inc - v;
}

__global__ __launch_bounds__(256) void padded_chunk_sums_kernel(const uint32_t *__restrict__ offsets,
                                                                uint32_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_wave[4];
    const uint32_t i0 = blockIdx.x * kScanChunk + threadIdx.x * 4u;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) v += padded_count(offsets, i0 + j, n);
    uint32_t total;
    (void)block_scan256(v, s_wave, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// in place: sums[b] <- sum of sums[0..b); one block, any number of chunks
__global__ __launch_bounds__(256) void padded_scan_sums_kernel(uint32_t *__restrict__ sums, uint32_t nchunks) {
    __shared__ uint32_t s_wave[4];
    const uint32_t per = (nchunks + 255u) / 256u;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t v = 0;
    for (uint32_t j = 0; j < per; ++j)
        if (b0 + j < nchunks) v += sums[b0 + j];
    uint32_t total;
    uint32_t run = block_scan256(v, s_wave, total);
    for (uint32_t j = 0; j < per; ++j)
        if (b0 + j < nchunks) {
            const uint32_t c = sums[b0 + j];
            sums[b0 + j] = run;
            run += c;
        }
}

__global__ __launch_bounds__(256) void padded_offsets_kernel(const uint32_t *__restrict__ offsets, uint32_t n,
                                                             const uint32_t *__restrict__ sums,
                                                             uint32_t *__restrict__ poff) {
    __shared__ uint32_t s_wave[4];
    const uint32_t i0 = blockIdx.x * kScanChunk + threadIdx.x * 4u;
    uint32_t c[4], v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        c[j] = padded_count(offsets, i0 + j, n);
        v += c[j];
    }
    uint32_t total;
    uint32_t run = sums[blockIdx.x] + block_scan256(v, s_wave, total);
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        if (i0 + j <= n) poff[i0 + j] = run;   // i0+j == n: the padded total
        run += c[j];
    }
}

// cells[i] = {x,y,z,density}; geo / link = the face tables (rf_foam.hpp).  A wave owns 64
// consecutive cells and streams their (contiguous, padded) entries with one lane per entry, so
// adjacency reads and table writes are coalesced; the owner of an entry is found by binary search
// in the wave's 65 padded offsets (LDS).  ext_diff != nullptr: take the half offsets from the
// caller's half4 table instead of recomputing them (trace_benchmark).
template <bool HALF>
__global__ __launch_bounds__(256) void prepare_foam_kernel(
    const float *__restrict__ points, const void *__restrict__ attributes, uint32_t attr_dim,
    uint32_t num_points, const uint32_t *__restrict__ adj, const uint32_t *__restrict__ offsets,
    const uint32_t *__restrict__ poff, const uint2 *__restrict__ ext_diff, float4 *__restrict__ cells,
    uint16_t *__restrict__ geo, Link *__restrict__ link, uint32_t *__restrict__ nbr) {
    __shared__ uint32_t s_off[4][66];
    __shared__ uint32_t s_csr[4][66];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t c0 = (blockIdx.x * 4u + wave) * 64u;
    if (c0 >= num_points) return;
    const uint32_t ncells = (num_points - c0 < 64u) ? num_points - c0 : 64u;
    uint32_t *off = s_off[wave];
    uint32_t *csr = s_csr[wave];
    if (lane <= ncells) {
        off[lane] = poff[c0 + lane];
        csr[lane] = offsets[c0 + lane];
    }
    if (lane == 0u) {
        off[ncells] = poff[c0 + ncells];
        csr[ncells] = offsets[c0 + ncells];
    }
    if (lane < ncells) {
        const uint32_t i = c0 + lane;
        float s = load_attr_scalar<HALF>(attributes, (size_t)i * attr_dim + attr_dim - 1);
        cells[i] = make_float4(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2], s);
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t f_begin = off[0], f_end = off[ncells];
    for (uint32_t f = f_begin + lane; f < f_end; f += 64u) {
        uint32_t lo = 0, hi = ncells;   // owner: last cell whose first entry is <= f
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (off[mid] <= f) lo = mid; else hi = mid;
        }
        const uint32_t j = f - off[lo];
        uint2 d = make_uint2(0u, 0u);
        Link lk = {0u, 0u, 0u};
        const bool real = j < csr[lo + 1] - csr[lo];
        {
            // a padding entry takes the offset of its block's first entry (a real face), scaled
            const uint32_t i = c0 + lo;
            const uint32_t src = csr[lo] + (real ? j : (j & ~3u));
            const uint32_t q = adj[src];
            if (ext_diff) {
                d = ext_diff[src];
            } else {
                const float px = points[3 * (size_t)i], py = points[3 * (size_t)i + 1], pz = points[3 * (size_t)i + 2];
                const float qx = points[3 * (size_t)q], qy = points[3 * (size_t)q + 1], qz = points[3 * (size_t)q + 2];
                d = pack_diff(qx - px, qy - py, qz - pz);
            }
            {
                // (a padding entry keeps the link of the face it copies: it never wins a scan -- see "the face scan" --
                // but should rounding ever let one, the ray crosses into the neighbour that face leads to instead of
                // following an all-zero link into cell 0)
                const uint32_t qb = poff[q];
                lk.nbr = q;
                lk.first = qb;
                lk.count = poff[q + 1] - qb;
            }
            if (!real) {
                uint16_t hx, hy, hz;
                pad_offset((uint16_t)(d.x & 0xFFFFu), (uint16_t)(d.x >> 16), (uint16_t)(d.y & 0xFFFFu), j & 3u, hx, hy, hz);
                d = make_uint2((uint32_t)hx | ((uint32_t)hy << 16), (uint32_t)hz);
            }
        }
        uint16_t *g = geo + (size_t)(f >> 2) * 12u + (f & 3u);
        g[0] = (uint16_t)(d.x & 0xFFFFu);
        g[4] = (uint16_t)(d.x >> 16);
        g[8] = (uint16_t)(d.y & 0xFFFFu);
        link[f] = lk;
        nbr[f] = real ? lk.nbr : kNone;     // (kNone is how the geometry-only repack recognises a padding entry)
    }
}

// Geometry-only repack: the padded offsets, the links and the neighbour list of a workspace depend on the
// adjacency alone, so while the triangulation is unchanged (every optimiser step between two rebuilds) only the
// cell records and the fp16 face offsets are rewritten.  A wave owns 64 consecutive cells; a lane takes one
// 4-entry block at a time (blocks never straddle cells: lists are padded to 4): one 16-byte read of the four
// neighbour indices, four point gathers (L2-resident: 24 MB), one 24-byte planar block written as three 8-byte
// stores; the owner cell comes from a binary search in the wave's 65 padded offsets (LDS), once per block.
// Streams 4 B in + 6 B out per entry (the first version walked the 12-byte links entry by entry with 2-byte
// stores: 0.26 ms for the 2M-point foam, 1.2 GB of traffic).
template <bool HALF>
__global__ __launch_bounds__(256) void prepare_geometry_kernel(
    const float *__restrict__ points, const void *__restrict__ attributes, uint32_t attr_dim,
    uint32_t num_points, const uint32_t *__restrict__ poff, const uint32_t *__restrict__ nbr,
    float4 *__restrict__ cells, uint16_t *__restrict__ geo) {
    __shared__ uint32_t s_off[4][66];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t c0 = (blockIdx.x * 4u + wave) * 64u;
    if (c0 >= num_points) return;
    const uint32_t ncells = (num_points - c0 < 64u) ? num_points - c0 : 64u;
    uint32_t *off = s_off[wave];
    if (lane <= ncells) off[lane] = poff[c0 + lane];
    if (lane == 0u) off[ncells] = poff[c0 + ncells];
    if (lane < ncells) {
        const uint32_t i = c0 + lane;
        float s = load_attr_scalar<HALF>(attributes, (size_t)i * attr_dim + attr_dim - 1);
        cells[i] = make_float4(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2], s);
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t b_begin = off[0] >> 2, b_end = off[ncells] >> 2;
    for (uint32_t b = b_begin + lane; b < b_end; b += 64u) {
        const uint32_t f = b << 2;
        uint32_t lo = 0, hi = ncells;   // owner: last cell whose first entry is <= f
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (off[mid] <= f) lo = mid; else hi = mid;
        }
        const uint32_t i = c0 + lo;
        const float px = points[3 * (size_t)i], py = points[3 * (size_t)i + 1], pz = points[3 * (size_t)i + 2];
        const uint4 q4 = *reinterpret_cast<const uint4 *>(nbr + f);
        const uint32_t q[4] = {q4.x, q4.y, q4.z, q4.w};
        uint16_t hx[4], hy[4], hz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hx[j] = hy[j] = hz[j] = (uint16_t)0;
            if (q[j] != kNone) {
                const float *qp = points + 3 * (size_t)q[j];
                hx[j] = float_to_half_bits(qp[0] - px);
                hy[j] = float_to_half_bits(qp[1] - py);
                hz[j] = float_to_half_bits(qp[2] - pz);
            } else {
                pad_offset(hx[0], hy[0], hz[0], (uint32_t)j, hx[j], hy[j], hz[j]);   // entry 0 of a block is a real face
            }
        }
        uint2 *g = reinterpret_cast<uint2 *>(geo + (size_t)b * 12u);
        g[0] = make_uint2((uint32_t)hx[0] | ((uint32_t)hx[1] << 16), (uint32_t)hx[2] | ((uint32_t)hx[3] << 16));
        g[1] = make_uint2((uint32_t)hy[0] | ((uint32_t)hy[1] << 16), (uint32_t)hy[2] | ((uint32_t)hy[3] << 16));
        g[2] = make_uint2((uint32_t)hz[0] | ((uint32_t)hz[1] << 16), (uint32_t)hz[2] | ((uint32_t)hz[3] << 16));
    }
}

// plain half4 table of the reference (pipeline.cu:546-568)
__global__ __launch_bounds__(256) void adjacent_diff_kernel(const float *__restrict__ points,
                                                            uint32_t num_points,
                                                            const uint32_t *__restrict__ adj,
                                                            const uint32_t *__restrict__ offsets,
                                                            uint2 *__restrict__ diff) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_points) return;
    float px = points[3 * (size_t)i], py = points[3 * (size_t)i + 1], pz = points[3 * (size_t)i + 2];
    for (uint32_t f = offsets[i]; f < offsets[i + 1]; ++f) {
        uint32_t q = adj[f];
        diff[f] = pack_diff(points[3 * (size_t)q] - px, points[3 * (size_t)q + 1] - py,
                            points[3 * (size_t)q + 2] - pz);
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

static bool valid_instance(int sh_degree, int attr_type) {
    return sh_degree >= 0 && sh_degree <= 3 && (attr_type == RF_ATTR_FLOAT32 || attr_type == RF_ATTR_FLOAT16);
}

static FoamView make_view(const FoamLayout &L, void *ws, const void *attributes, const uint32_t *offsets) {
    FoamView v;
    char *base = static_cast<char *>(ws);
    v.cells = reinterpret_cast<const float4 *>(base + L.cells_off);
    v.geo = reinterpret_cast<const uint16_t *>(base + L.geo_off);
    v.link = reinterpret_cast<const Link *>(base + L.link_off);
    v.poff = reinterpret_cast<const uint32_t *>(base + L.poff_off);
    v.offsets = offsets;
    v.sh = L.sh_repacked ? static_cast<const void *>(base + L.sh_off) : attributes;
    v.sh_stride = L.sh_stride;
    return v;
}

static int prepare_impl(int sh_degree, int attr_type, uint32_t num_points, const float *points,
                        const void *attributes, uint32_t adj_size, const uint32_t *adj,
                        const uint32_t *offsets, const void *ext_diff, void *ws, size_t ws_bytes,
                        hipStream_t stream, bool topology_valid = false) {
    const bool half = attr_type == RF_ATTR_FLOAT16;
    FoamLayout L = foam_layout(num_points, adj_size, sh_degree, half);
    if (!ws || ws_bytes < L.total) return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_workspace_bytes()");
    if (num_points == 0) return RF_OK;
    char *base = static_cast<char *>(ws);
    float4 *cells = reinterpret_cast<float4 *>(base + L.cells_off);
    uint16_t *geo = reinterpret_cast<uint16_t *>(base + L.geo_off);
    Link *link = reinterpret_cast<Link *>(base + L.link_off);
    uint32_t *nbr = reinterpret_cast<uint32_t *>(base + L.nbr_off);
    uint32_t *poff = reinterpret_cast<uint32_t *>(base + L.poff_off);
    uint32_t *sums = reinterpret_cast<uint32_t *>(base + L.scan_off);
    const uint32_t A = attribute_dim(sh_degree);
    dim3 block(256);
    dim3 grid((num_points + 255u) / 256u);   // 4 waves x 64 cells per block
    if (topology_valid && !ext_diff) {
        if (half)
            hipLaunchKernelGGL(prepare_geometry_kernel<true>, grid, block, 0, stream, points, attributes, A,
                               num_points, poff, nbr, cells, geo);
        else
            hipLaunchKernelGGL(prepare_geometry_kernel<false>, grid, block, 0, stream, points, attributes, A,
                               num_points, poff, nbr, cells, geo);
    } else {
        // padded offsets; chunks cover cells 0..num_points inclusive (the last entry is the total)
        const uint32_t nchunks = num_points / kScanChunk + 1u;
        hipLaunchKernelGGL(padded_chunk_sums_kernel, dim3(nchunks), block, 0, stream, offsets, num_points, sums);
        hipLaunchKernelGGL(padded_scan_sums_kernel, dim3(1), block, 0, stream, sums, nchunks);
        hipLaunchKernelGGL(padded_offsets_kernel, dim3(nchunks), block, 0, stream, offsets, num_points, sums, poff);
        if (half)
            hipLaunchKernelGGL(prepare_foam_kernel<true>, grid, block, 0, stream, points, attributes, A, num_points,
                               adj, offsets, poff, static_cast<const uint2 *>(ext_diff), cells, geo, link, nbr);
        else
            hipLaunchKernelGGL(prepare_foam_kernel<false>, grid, block, 0, stream, points, attributes, A, num_points,
                               adj, offsets, poff, static_cast<const uint2 *>(ext_diff), cells, geo, link, nbr);
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
    static int run(const FwdParams &p, bool bench, uint32_t forward_mode, hipStream_t stream) {
        uint32_t nb = launch_blocks(p.grid);
        if (nb == 0) return RF_OK;
        // Eager face blocks where a launch waits on memory rather than on issue slots: a flat batch (its lanes share
        // little, most face lists come from HBM) and any launch small enough to be resident at once with at most four
        // waves per SIMD (the launch is as long as its longest ray's chain of dependent loads).
        const bool eager = forward_mode ? forward_mode == 2u : (p.grid.img_w == 0 || nb <= kResidentBlocks);
        // ... behind a block-level LDS table of cell records and face blocks where a block's rays come back to the cells
        // they crossed: the 256-slot groups of a SORTED flat batch (rf_build_ray_order).  Training batch of bench.py
        // (profiles/r04/n_cell_table_ab.log): forward 4.36 -> 3.85 ms at SH 3, 4.29 -> 3.56 at SH 2, 6.36 -> 5.40 with every
        // segment lit; whole batch bitwise equal to the CPU checker.
        const bool cached = forward_mode ? forward_mode == 5u : (p.grid.img_w == 0 && p.grid.order != nullptr);
        const dim3 g(nb), b(kBlock);
        if (forward_mode == 4u && !bench && !p.nq && !p.stats && p.queue) {
            // persistent waves: as many blocks as are resident at once, refilled from the queue of the ordinary launch
            const uint32_t resident = 256u * (uint32_t)(DEG <= 2 ? RF_PERSISTENT_WAVES : 4);
            hipMemsetAsync(p.queue, 0, sizeof(uint32_t), stream);
            hipLaunchKernelGGL((forward_persistent_kernel<DEG, HALF>), dim3(nb < resident ? nb : resident), b, 0, stream, p,
                               p.queue, nb);
        } else if (forward_mode == 3u) {   // every face divided (scan_faces_strict); statistics stay on the filtered scan
            if (bench)
                hipLaunchKernelGGL((forward_kernel<DEG, HALF, true, false, false, kScanStrict>), g, b, 0, stream, p);
            else if (p.nq)
                hipLaunchKernelGGL((forward_kernel<DEG, HALF, false, true, false, kScanStrict>), g, b, 0, stream, p);
            else
                hipLaunchKernelGGL((forward_kernel<DEG, HALF, false, false, false, kScanStrict>), g, b, 0, stream, p);
        } else if (cached && !bench && !p.stats) {
            if (p.nq)
                hipLaunchKernelGGL((forward_kernel<DEG, HALF, false, true, false, kScanCached>), g, b, 0, stream, p);
            else
                hipLaunchKernelGGL((forward_kernel<DEG, HALF, false, false, false, kScanCached>), g, b, 0, stream, p);
        } else if (bench)
            hipLaunchKernelGGL((forward_kernel<DEG, HALF, true, false, false, kScanBlocks>), g, b, 0, stream, p);
        else if (p.stats)
            hipLaunchKernelGGL((forward_kernel<DEG, HALF, false, true, true, kScanBlocks>), g, b, 0, stream, p);
        else if (p.nq && eager)
            hipLaunchKernelGGL((forward_kernel<DEG, HALF, false, true, false, kScanEager>), g, b, 0, stream, p);
        else if (p.nq)
            hipLaunchKernelGGL((forward_kernel<DEG, HALF, false, true, false, kScanBlocks>), g, b, 0, stream, p);
        else if (eager)
            hipLaunchKernelGGL((forward_kernel<DEG, HALF, false, false, false, kScanEager>), g, b, 0, stream, p);
        else
            hipLaunchKernelGGL((forward_kernel<DEG, HALF, false, false, false, kScanBlocks>), g, b, 0, stream, p);
        return check_launch(bench ? "rf_trace_benchmark" : "rf_trace_forward");
    }
};

template <int DEG, bool HALF>
struct LaunchBackward {
    static int run(const BwdParams &p, int mode, hipStream_t stream) {
        uint32_t nb = launch_blocks(p.grid);
        if (nb == 0) return RF_OK;
        if (p.trail) {
            if (mode == 1)
                hipLaunchKernelGGL((backward_replay_kernel<DEG, HALF, 1>), dim3(nb), dim3(kBlock), 0, stream, p);
            else if (mode == 2)
                hipLaunchKernelGGL((backward_replay_kernel<DEG, HALF, 2>), dim3(nb), dim3(kBlock), 0, stream, p);
            else if (mode == 4) {
                if (p.nq)
                    hipLaunchKernelGGL((backward_replay_direct_kernel<DEG, HALF, true>), dim3(nb), dim3(kBlock), 0, stream, p);
                else
                    hipLaunchKernelGGL((backward_replay_direct_kernel<DEG, HALF, false>), dim3(nb), dim3(kBlock), 0, stream, p);
            } else if (p.nq)
                hipLaunchKernelGGL((backward_replay_cached_kernel<DEG, HALF, true>), dim3(nb), dim3(kBlock), 0, stream, p);
            else
                hipLaunchKernelGGL((backward_replay_cached_kernel<DEG, HALF, false>), dim3(nb), dim3(kBlock), 0, stream, p);
            // rays that did not fit in the trail (blocks without any exit at once)
            if (mode == 1)
                hipLaunchKernelGGL((backward_kernel<DEG, HALF, 1>), dim3(nb), dim3(kBlock), 0, stream, p);
            else
                hipLaunchKernelGGL((backward_kernel<DEG, HALF, 2>), dim3(nb), dim3(kBlock), 0, stream, p);
        } else {
            if (mode == 1)
                hipLaunchKernelGGL((backward_kernel<DEG, HALF, 1>), dim3(nb), dim3(kBlock), 0, stream, p);
            else
                hipLaunchKernelGGL((backward_kernel<DEG, HALF, 2>), dim3(nb), dim3(kBlock), 0, stream, p);
        }
        return check_launch("rf_trace_backward");
    }
};

static RayGrid make_grid(uint32_t num_rays, const rf_launch_opts *opts) {
    RayGrid g{num_rays, 0u, 0u, nullptr, nullptr};
    if (opts && opts->image_width && opts->image_height &&
        (uint64_t)opts->image_width * opts->image_height == num_rays) {
        g.img_w = opts->image_width;
        g.img_h = opts->image_height;
        g.tile_order = opts->tile_order;
    } else if (opts) {
        g.order = opts->ray_order;
        g.tile_order = opts->tile_order;
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

uint32_t rf_trail_slots(uint32_t num_rays, uint32_t image_width, uint32_t image_height) {
    rf_launch_opts o{};
    o.image_width = image_width;
    o.image_height = image_height;
    return num_tiles(make_grid(num_rays, &o)) * (uint32_t)kBlock;
}

uint32_t rf_launch_blocks(uint32_t num_rays, uint32_t image_width, uint32_t image_height, uint32_t *tiles) {
    rf_launch_opts o{};
    o.image_width = image_width;
    o.image_height = image_height;
    const RayGrid g = make_grid(num_rays, &o);
    const uint32_t nb = launch_blocks(g);
    if (tiles) {
        const uint32_t chunk = tile_chunk(g);
        for (uint32_t b = 0; b < nb; ++b) tiles[b] = dealt_tile(b, chunk, nb / (8u * chunk));
    }
    return nb;
}

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
    hipLaunchKernelGGL(adjacent_diff_kernel, dim3((num_points + 255u) / 256u), dim3(256), 0,
                       static_cast<hipStream_t>(stream), points, num_points, point_adjacency,
                       point_adjacency_offsets, static_cast<uint2 *>(adjacent_diff));
    return check_launch("rf_build_adjacent_diff");
}

int rf_prepare_foam(int sh_degree, int attr_type, uint32_t num_points, const float *points,
                    const void *attributes, uint32_t point_adjacency_size,
                    const uint32_t *point_adjacency, const uint32_t *point_adjacency_offsets,
                    const void *adjacent_diff, void *workspace, size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (!valid_instance(sh_degree, attr_type))
        return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported SH degree or attribute type");
    if (num_points && (!points || !attributes || !point_adjacency || !point_adjacency_offsets))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_prepare_foam: null pointer");
    return prepare_impl(sh_degree, attr_type, num_points, points, attributes, point_adjacency_size,
                        point_adjacency, point_adjacency_offsets, adjacent_diff, workspace,
                        workspace_bytes, static_cast<hipStream_t>(stream));
}

int rf_prepare_foam_geometry(int sh_degree, int attr_type, uint32_t num_points, const float *points,
                             const void *attributes, uint32_t point_adjacency_size, void *workspace,
                             size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (!valid_instance(sh_degree, attr_type))
        return fail(RF_ERR_INVALID_ARGUMENT, "Unsupported SH degree or attribute type");
    if (num_points && (!points || !attributes))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_prepare_foam_geometry: null pointer");
    return prepare_impl(sh_degree, attr_type, num_points, points, attributes, point_adjacency_size, nullptr,
                        nullptr, nullptr, workspace, workspace_bytes, static_cast<hipStream_t>(stream), true);
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
    // An empty batch launches no walk, but the workspace is still packed below: whatever num_rays is,
    // after a successful call the workspace describes these inputs (callers cache on that).
    if (!points || !attributes || !point_adjacency || !point_adjacency_offsets ||
        (num_rays != 0 && (!rays || !start_point_index || !ray_rgba)))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_forward: null pointer");
    if (num_depth_quantiles && depth_quantiles && (!quantile_depths || !quantile_point_indices))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_forward: depth quantile buffers missing");
    if (opts->forward_mode > 5u) return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_forward: forward_mode must be 0..5");
    const bool half = attr_type == RF_ATTR_FLOAT16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    FoamLayout L = foam_layout(num_points, point_adjacency_size, sh_degree, half);
    if (!opts->workspace || opts->workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_workspace_bytes()");
    if (opts->foam_prepared != 1u) {
        int rc = prepare_impl(sh_degree, attr_type, num_points, points, attributes, point_adjacency_size,
                              point_adjacency, point_adjacency_offsets, nullptr, opts->workspace,
                              opts->workspace_bytes, s, opts->foam_prepared == 2u);
        if (rc != RF_OK) return rc;
    }
    if (num_rays == 0) return RF_OK;
    FwdParams p{};
    p.foam = make_view(L, opts->workspace, attributes, point_adjacency_offsets);
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
    p.visit_marks = opts->stats ? opts->visit_marks : nullptr;
    p.tile_cost = opts->tile_cost;
    // the ray queue head of forward_mode 4: a word of its own in the workspace (FoamLayout::queue_off -- not the packing's
    // scan scratch any more; rf_workspace_bytes grew by 256 bytes for it in round 5)
    p.queue = reinterpret_cast<uint32_t *>(static_cast<char *>(opts->workspace) + L.queue_off);
    if (opts->trail && opts->trail_hops && opts->trail_cap) {
        if (opts->trail_slots < num_tiles(p.grid) * (uint32_t)kBlock)
            return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_forward: trail_slots smaller than rf_trail_slots()");
        if (opts->trail_slots >= (1u << 30))
            return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_forward: a trail row of 4 GB or more (2^30 slots) is not addressable");
        p.trail = opts->trail;
        p.trail_hops = opts->trail_hops;
        p.trail_cap = opts->trail_cap;
        p.trail_slots = opts->trail_slots;
    }
    return dispatch<LaunchForward>(sh_degree, half, p, false, opts->forward_mode, s);
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
    if (!points || !attributes || !point_adjacency || !point_adjacency_offsets ||
        (num_rays != 0 && (!rays || !start_point_index || !ray_rgba || !ray_rgba_grad || !points_grad ||
                           !attribute_grad)))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_backward: null pointer");
    if (num_depth_quantiles && depth_quantiles && (!quantile_point_indices || !depth_grad))
        return fail(RF_ERR_INVALID_ARGUMENT, "depth_grad must be provided if depth_quantiles is provided");
    if (opts->backward_mode > 4u)
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_backward: backward_mode must be 0..4");
    if (opts->forward_mode > 5u) return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_backward: forward_mode must be 0..5");
    const bool half = attr_type == RF_ATTR_FLOAT16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    FoamLayout L = foam_layout(num_points, point_adjacency_size, sh_degree, half);
    if (!opts->workspace || opts->workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_workspace_bytes()");
    if (opts->foam_prepared != 1u) {
        int rc = prepare_impl(sh_degree, attr_type, num_points, points, attributes, point_adjacency_size,
                              point_adjacency, point_adjacency_offsets, nullptr, opts->workspace,
                              opts->workspace_bytes, s, opts->foam_prepared == 2u);
        if (rc != RF_OK) return rc;
    }
    if (num_rays == 0) return RF_OK;   // workspace packed above, nothing to walk
    BwdParams p{};
    p.foam = make_view(L, opts->workspace, attributes, point_adjacency_offsets);
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
    p.stats = reinterpret_cast<unsigned long long *>(opts->stats);
    p.attr_pitch = opts->attr_grad_pitch ? opts->attr_grad_pitch : attribute_dim(sh_degree);
    if (p.attr_pitch < attribute_dim(sh_degree))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_backward: attr_grad_pitch smaller than the attribute dimension");
    if (opts->trail && opts->trail_hops && opts->trail_cap) {
        if (opts->trail_slots < num_tiles(p.grid) * (uint32_t)kBlock)
            return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_backward: trail_slots smaller than rf_trail_slots()");
        if (opts->trail_slots >= (1u << 30))
            return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_backward: a trail row of 4 GB or more (2^30 slots) is not addressable");
        p.trail = opts->trail;
        p.trail_hops = opts->trail_hops;
        p.trail_cap = opts->trail_cap;
        p.trail_slots = opts->trail_slots;
    }
    // 0 = auto: with a trail to replay, the block cache for image-shaped batches (dense: many rays per
    // cell and tile) and direct row atomics for flat ones; wave-reduced scatter without a trail
    int mode = (int)opts->backward_mode;
    if (mode == 0) mode = p.trail ? (p.grid.img_w ? 3 : 4) : 2;
    if ((mode == 3 || mode == 4) && !p.trail) mode = 2;
    return dispatch<LaunchBackward>(sh_degree, half, p, mode, s);
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
    const bool no_pixels = camera->width == 0 || camera->height == 0;
    if (!points || !attributes || !point_adjacency || !point_adjacency_offsets || !adjacent_diff ||
        (!no_pixels && (!start_point_index || !ray_rgba)))
        return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_benchmark: null pointer");
    const bool half = attr_type == RF_ATTR_FLOAT16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    FoamLayout L = foam_layout(num_points, point_adjacency_size, sh_degree, half);
    if (!opts->workspace || opts->workspace_bytes < L.total)
        return fail(RF_ERR_WORKSPACE, "workspace missing or smaller than rf_workspace_bytes()");
    if (!opts->foam_prepared) {
        // the half offsets are the CALLER's (benchmark.py:44-54): packed into the fat table as given
        int rc = prepare_impl(sh_degree, attr_type, num_points, points, attributes, point_adjacency_size,
                              point_adjacency, point_adjacency_offsets, adjacent_diff, opts->workspace,
                              opts->workspace_bytes, s);
        if (rc != RF_OK) return rc;
    }
    if (no_pixels) return RF_OK;   // workspace packed above (as for an empty ray batch)
    FwdParams p{};
    p.foam = make_view(L, opts->workspace, attributes, point_adjacency_offsets);
    p.grid = RayGrid{camera->width * camera->height, camera->width, camera->height, nullptr, opts->tile_order};
    p.tile_cost = opts->tile_cost;
    p.settings = *settings;
    p.start = start_point_index;
    p.cam = *camera;
    p.inv_tan_half_fov = 1.0f / tanf(camera->fov * 0.5f);
    p.rgba8 = ray_rgba;
    if (opts->forward_mode > 5u) return fail(RF_ERR_INVALID_ARGUMENT, "rf_trace_benchmark: forward_mode must be 0..5");
    return dispatch<LaunchForward>(sh_degree, half, p, true, opts->forward_mode == 3u ? 3u : 1u, s);
}

}  // extern "C"
