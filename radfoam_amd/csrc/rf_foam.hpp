// rf_foam.hpp -- HBM layout of the packed foam the walk kernels read.
//
// The caller's tensors are AoS as the reference API dictates (points[N][3], attributes[N][A],
// CSR offsets[N+1], adjacency[E]).  rf_prepare_foam re-lays the per-cell scalars the walk touches
// on every step into one 32-byte record so that a hop costs two adjacent 16-byte gathers instead
// of four scattered ones (offsets[i], offsets[i+1], points[i] (12 B, unaligned), density at the
// end of the attribute row):
//
//   workspace = [ RfCell cells[N] | half4 face_diff[E + 32] | SH rows[N][sh_stride] (optional) ]
//
//   RfCell      {x, y, z, density, face_begin, face_end, 0, 0}            32 B, 32-B aligned
//   face_diff   half4(points[adj[e]] - points[owner(e)], 0), RNE          8 B per CSR entry; the
//               faces of one cell are one contiguous 8*F-byte run (reference layout,
//               pipeline.cu:546-568, incl. its +32 entries of padding)
//   SH rows     the 3B colour coefficients of a cell, 16-B aligned rows of sh_stride scalars;
//               present only when the caller's row pitch (A scalars) is not 16-B aligned
//               (fp32: d=1,3; fp16: all but d=3 use 8-B loads, d=1,3 are repacked); otherwise the
//               kernels read the caller's attribute rows in place.
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace rf {

struct alignas(32) RfCell {
    float x, y, z, s;
    uint32_t begin, end, pad0, pad1;
};

constexpr uint32_t kDiffPad = 32;  // entries; same slack the reference allocates (pipeline.cu:613)

struct FoamLayout {
    size_t cells_off;
    size_t diff_off;
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
    // rows readable in place iff the pitch keeps every row aligned for the vector loads used:
    // fp32 rows are read as float4 (16 B), fp16 rows as 4 halves (8 B)
    const bool in_place = (A % 4u) == 0u;
    L.sh_repacked = !in_place;
    L.sh_stride = in_place ? A : (uint32_t)align_up(ncoef, 4);
    L.cells_off = 0;
    size_t off = align_up((size_t)num_points * sizeof(RfCell), 256);
    L.diff_off = off;
    off = align_up(off + ((size_t)adj_size + kDiffPad) * 8, 256);
    if (L.sh_repacked) {
        L.sh_off = off;
        off = align_up(off + (size_t)num_points * L.sh_stride * (attr_half ? 2 : 4), 256);
    }
    L.total = off;
    return L;
}

}  // namespace rf
