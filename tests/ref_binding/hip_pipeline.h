// hip_pipeline.h -- what the reference-side binding of Seam B exports next to the reference's own
// src/tracing/pipeline.h: the device-memory hooks the Pipeline subclass allocates through.
//
// In the reference these would be its TorchBuffer / CUDAArray helpers (torch_bindings.cpp:14-30, the torch
// caching allocator); the test harness of this repository plugs hipMalloc / hipFree / hipMemsetAsync.
#pragma once
#include <cstddef>
#include <cstdint>

namespace radfoam {

struct DeviceMemoryHooks {
    void *(*alloc)(size_t bytes);
    void (*free)(void *ptr);
    void (*zero)(void *ptr, size_t bytes, void *stream);   // stream-ordered memset to 0
    void *stream;                                          // hipStream_t the pipeline launches on (NULL = default)
};

// Must be called once before create_pipeline(); the hooks are copied.
void set_hip_pipeline_memory(const DeviceMemoryHooks &hooks);

}  // namespace radfoam
