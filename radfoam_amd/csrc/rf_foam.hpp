// rf_foam.hpp -- HBM layout of the packed foam the walk kernels read.
//
// The caller's tensors are AoS as the reference API dictates (points[N][3], attributes[N][A],
// CSR offsets[N+1], adjacency[E]).  A hop of the reference walk chases four dependent pointers
// (offsets[i], offsets[i+1] -> face table -> adjacency[e] -> points[j]); rf_prepare_foam re-lays
// the foam so that a hop costs ONE dependent round trip and the face scan no unpacking:
//
//   workspace = [ float4 cells[N] | half geo[3 * EB] | Link link[EB] | uint32 nbr[EB] | uint32 poff[N+1] |
//                 uint32 scan scratch | uint32 ray queue head | SH rows[N][sh_stride] (optional) ]
//
//   Every cell's face list is padded to a multiple of 4 entries; poff[i] is the first (padded)
//   entry of cell i and poff[N] = E' <= EB = E + 3N the padded total.  Entries past a cell's real
//   faces are all-zero (a zero offset is never an exit candidate: o.d = 0).
//
//   cells[i]   {x, y, z, density}                                           16 B, 16-B aligned
//   geo        blocks of 4 consecutive entries, 24 B per block, components planar inside a block:
//                half x[4], y[4], z[4]   with (x,y,z)[e] = points[adj[e]] - points[owner(e)]
//              rounded to fp16 (RNE): exactly the values of the reference's half4 table
//              (pipeline.cu:546-568), 6 B per face instead of 8.  An iteration of the scan reads
//              a block with one 16-B and one 8-B load; each dword unpacks into an adjacent register
//              pair, the operand of one packed-fp32 instruction (two faces per VALU slot).
//   link[e]    {adj[e], poff[adj[e]], padded face count of adj[e]}, 12 B: the neighbour, where its
//              faces start and how many -- read once per hop, for the winning face only; the next
//              cell's face list, cell record and SH row can then all be requested at once.
//   nbr[e]     adj[e] again (kNone for padding entries), 4 B: what the geometry-only repack streams instead of
//              the 12-B links when only the points moved (an optimiser step between two triangulation rebuilds)
//   SH rows    the 3B colour coefficients of a cell, aligned rows of sh_stride scalars; present only for fp16
//              attributes whose row pitch (A scalars) is odd (d=1,3); otherwise the kernels read the caller's
//              attribute rows in place (fp32 rows of odd pitch through unaligned 16-byte loads).
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace rf {

constexpr uint32_t kFacePad = 32;     // entries of zero slack behind the last list (scan prefetch)
constexpr uint32_t kScanChunk = 1024;  // cells per block of the padded-offset prefix sum

struct Link {
    uint32_t nbr;    // neighbour cell
    uint32_t first;  // its first (padded) face entry
    uint32_t count;  // its padded face count
};

struct FoamLayout {
    size_t cells_off;
    size_t geo_off;
    size_t link_off;
    size_t nbr_off;
    size_t poff_off;
    size_t scan_off;     // per-chunk sums of the prefix sum
    size_t queue_off;    // one uint32: the ray queue head of the persistent-wave forward (forward_mode 4)
    size_t max_entries;  // EB: upper bound of the padded entry count
    size_t sh_off;       // 0 when rows are read in place
    uint32_t sh_stride;  // scalars per SH row as the kernels see it
    bool sh_repacked;
    size_t total;
};

inline uint32_t attribute_dim(int sh_degree) {
    return (sh_degree < 0 || sh_degree > 3) ? 0u : 1u + 3u * (uint32_t)((sh_degree + 1) * (sh_degree + 1));
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline FoamLayout foam_layout(uint32_t num_points, uint32_t adj_size, int sh_degree, int attr_half) {
    FoamLayout L{};
    const uint32_t A = attribute_dim(sh_degree);
    const uint32_t ncoef = A - 1;
    // fp32 rows are read in place whatever their pitch: as float4 (16 B) from addresses that are only 4-byte aligned
    // when A is odd (d = 1, 3), which the hardware serves in its unaligned access mode at no measurable cost
    // (profiles/r03/p_sh_rows_in_place_*: the 4 M-point SH-3 frame 20.5 -> 20.3 ms) -- round 2 repacked those rows
    // into aligned ones in every step (0.24 ms for 2 M points, 0.45 ms for 4 M).  fp16 rows are read as 4 halves
    // (8 B); with an odd pitch they are only 2-byte aligned, which dword loads do not accept: repacked.
    const bool in_place = (A % 4u) == 0u || !attr_half;
    L.sh_repacked = !in_place;
    L.sh_stride = in_place ? A : (uint32_t)align_up(ncoef, 4);
    L.cells_off = 0;
    size_t off = align_up((size_t)num_points * 16, 256);
    L.max_entries = align_up((size_t)adj_size + 3 * (size_t)num_points, 4);
    L.geo_off = off;
    off = align_up(off + (L.max_entries + kFacePad) * 6, 256);
    L.link_off = off;
    off = align_up(off + L.max_entries * 12, 256);
    L.nbr_off = off;
    off = align_up(off + L.max_entries * 4, 256);
    L.poff_off = off;
    off = align_up(off + ((size_t)num_points + 1) * 4, 256);
    L.scan_off = off;
    off = align_up(off + ((size_t)num_points / kScanChunk + 2) * 4, 256);
    L.queue_off = off;
    off += 256;
    if (L.sh_repacked) {
        L.sh_off = off;
        off = align_up(off + (size_t)num_points * L.sh_stride * (attr_half ? 2 : 4), 256);
    }
    L.total = off;
    return L;
}

}  // namespace rf
