// dev_mem.cpp -- the device-memory hooks of the Seam B test harness (hipMalloc / hipFree / hipMemsetAsync);
// a reference build would pass its torch caching-allocator helpers instead.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime_api.h>

extern "C" {

void *rb_dev_alloc(size_t bytes) {
    void *p = nullptr;
    return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr;
}

void rb_dev_free(void *ptr) {
    // hipFree waits for the device: the scratch of a call is released only after its kernels ran
    (void)hipFree(ptr);
}

void rb_dev_zero(void *ptr, size_t bytes, void *stream) { (void)hipMemsetAsync(ptr, 0, bytes, static_cast<hipStream_t>(stream)); }

}  // extern "C"
