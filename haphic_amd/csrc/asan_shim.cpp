// Only in the AddressSanitizer build (haphic_amd/build.py build_asan).  The host code is instrumented by ROCm's clang but runs on GCC's
// libasan (LD_PRELOAD): ROCm's own runtime intercepts hsa_amd_memory_pool_allocate for xnack+ device sanitizing, which a gfx950 /
// xnack- process cannot serve.  GCC 11's runtime speaks the same instrumentation ABI (v8) except for these three helpers.
#include <cstddef>
#include <cstring>
extern "C" {
__attribute__((no_sanitize("address"), visibility("default"))) void *__sanitizer_internal_memcpy(void *d, const void *s, size_t n) { return memcpy(d, s, n); }
__attribute__((no_sanitize("address"), visibility("default"))) void *__sanitizer_internal_memmove(void *d, const void *s, size_t n) { return memmove(d, s, n); }
__attribute__((no_sanitize("address"), visibility("default"))) void *__sanitizer_internal_memset(void *d, int c, size_t n) { return memset(d, c, n); }
}
