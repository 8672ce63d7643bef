// rf_sort.hpp -- the device sort behind backward mode 5 (records of colour-row gradients ordered by cell), implemented
// in rf_adjacency.hip, where the library's other rocPRIM sorts live, and called by rf_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rf {

// temporary storage rocPRIM wants for sorting up to `capacity` (key, 16-byte record) pairs
size_t gather_sort_temp_bytes(uint32_t capacity);

// stable radix sort of `count` pairs by the low `key_bits` bits of the key; RF_OK or an error code (message set)
int gather_sort(const uint32_t *keys_in, uint32_t *keys_out, const uint4 *recs_in, uint4 *recs_out, uint32_t count,
                unsigned key_bits, void *temp, size_t temp_bytes, hipStream_t stream);

}  // namespace rf
