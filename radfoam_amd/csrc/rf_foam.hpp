// rf_foam.hpp -- HBM layout of the packed foam the walk kernels read.
//
// The caller's tensors are AoS as the reference API dictates (points[N][3], attributes[N][A],
// CSR offsets[N+1], adjacency[E]).  A hop of the reference walk chases four dependent pointers
// (offsets[i], offsets[i+1] -> face table -> adjacency[e] -> points[j]); rf_prepare_foam re-lays
// the foam so that a hop costs ONE dependent round trip:
//
//   workspace = [ float4 cells[N] | uint2 geo[E + 32] | uint2 link[E] | SH rows[N][sh_stride] (optional) ]
//
//   cells[i]   {x, y, z, density}                                           16 B, 16-B aligned
//   geo[e]     8 B per CSR entry, a cell's faces contiguous (what the scan streams):
//                .x = half(dx) | half(dy) << 16      (dx,dy,dz) = points[adj[e]] - points[owner(e)],
//                .y = half(dz) | nbr_faces << 16       fp16 RNE == the reference's half4 table
//                                                      (pipeline.cu:546-568) with the neighbour's
//                                                      face count in the unused w slot
//   link[e]    {adj[e], offsets[adj[e]]}: the neighbour and its first face -- read once per hop,
//              for the winning face only.  Together with nbr_faces the winning face already names
//              the next cell AND where its faces are: the next cell's face list, cell record and
//              SH row can all be requested at once.
//   SH rows    the 3B colour coefficients of a cell, 16-B aligned rows of sh_stride scalars;
//              present only when the caller's row pitch (A scalars) is not 16-B aligned
//              (d=1,3); otherwise the kernels read the caller's attribute rows in place.
//
// nbr_faces is 16 bits: cells with more than 65535 Delaunay neighbours are not supported
// (rf_prepare_foam does not check; random and trained foams have < 100).
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace rf {

constexpr uint32_t kFacePad = 32;  // entries; same slack the reference allocates (pipeline.cu:613)

struct FoamLayout {
    size_t cells_off;
    size_t geo_off;
    size_t link_off;
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
    size_t off = align_up((size_t)num_points * 16, 256);
    L.geo_off = off;
    off = align_up(off + ((size_t)adj_size + kFacePad) * 8, 256);
    L.link_off = off;
    off = align_up(off + (size_t)adj_size * 8, 256);
    if (L.sh_repacked) {
        L.sh_off = off;
        off = align_up(off + (size_t)num_points * L.sh_stride * (attr_half ? 2 : 4), 256);
    }
    L.total = off;
    return L;
}

}  // namespace rf
