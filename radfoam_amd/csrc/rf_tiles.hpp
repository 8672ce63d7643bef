// rf_tiles.hpp -- which tile a block of a walk launch takes when nobody says otherwise (the static dealing), shared by the
// walk kernels (rf_kernels.hip) and the device-side builder of tile orders (rf_tile_prior.hip).
//
// A tile = a 16x16-pixel patch of an image-shaped batch (a wave owns an 8x8 quadrant) or 256 consecutive thread slots of a
// flat one.  The dispatcher places block b on XCD b % 8, and every XCD has a private L2: tiles are dealt to the XCDs in
// chunks of `chunk` consecutive tiles of the row-major order (a quarter of an image row), a round of 8 chunks at a time,
// the chunk an XCD takes rotating from round to round; the launch is padded to whole rounds (blocks whose tile falls past
// the end own no rays).  rf_kernels.hip has the measurements behind each of these choices.
#pragma once

#include <stdint.h>

#include <hip/hip_runtime.h>

namespace rf {

constexpr uint32_t kTileSlots = 256;     // thread slots per tile = threads per block of the walk kernels

inline __host__ __device__ uint32_t dealt_tile(uint32_t b, uint32_t chunk, uint32_t rounds) {
    const uint32_t x = b & 7u, i = b >> 3;
    uint32_t j = i / chunk;
    const uint32_t o = i - j * chunk;
    // rounds are visited from both ends of the image towards its middle (0, last, 1, last-1, ...):
    // the blocks still running when the launch drains are then neighbours in the image, of
    // similar length, instead of the longest walks of the frame
#ifdef RF_DEAL_MIDDLE_FIRST
    j = rounds - 1u - j;       // experiment: the same sequence backwards -- the middle of the image first, its ends last
#endif
    j = (j & 1u) ? rounds - 1u - (j >> 1) : (j >> 1);
    return (j * 8u + ((x + 3u * j) & 7u)) * chunk + o;
}

// tiles of a batch: image_width == 0 means a flat list of num_rays rays
inline __host__ __device__ uint32_t tile_count(uint32_t num_rays, uint32_t image_width, uint32_t image_height) {
    if (image_width) return ((image_width + 15u) >> 4) * ((image_height + 15u) >> 4);
    return (num_rays + kTileSlots - 1u) / kTileSlots;
}

inline __host__ __device__ uint32_t tile_chunk_of(uint32_t image_width) {
    if (image_width) {
        const uint32_t tiles_x = (image_width + 15u) >> 4;
        return tiles_x > 3u ? (tiles_x + 3u) >> 2 : 1u;
    }
    return 16u;
}

// blocks to launch: whole rounds of 8 chunks
inline __host__ __device__ uint32_t launch_block_count(uint32_t num_rays, uint32_t image_width, uint32_t image_height) {
    const uint32_t nt = tile_count(num_rays, image_width, image_height);
    if (nt == 0) return 0;
    const uint32_t round = 8u * tile_chunk_of(image_width);
    return (nt + round - 1u) / round * round;
}

}  // namespace rf
