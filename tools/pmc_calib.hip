// Known-byte microkernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE / TCC_* counters on gfx950 for the
// lane access widths the MCL kernels use (2-byte, 4-byte, 8-byte and 16-byte lane loads), MI355X_MICROARCH.md §HBM:
// "calibrate on a known byte count in your own access pattern before trusting an absolute".  Each kernel streams a
// buffer exactly once per launch, fully coalesced; `big` (4 GiB) is far beyond the 256 MiB Infinity Cache, `small`
// (64 MiB) fits it and is re-read `reps` times inside one launch (first pass cold, the rest on-die).
// Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib;  run under rocprofv3 --pmc ...
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T> __device__ inline uint64_t fold(T v);
template <> __device__ inline uint64_t fold(unsigned short v) { return v; }
template <> __device__ inline uint64_t fold(unsigned int v) { return v; }
template <> __device__ inline uint64_t fold(uint2 v) { return (uint64_t)v.x + v.y; }
template <> __device__ inline uint64_t fold(uint4 v) { return (uint64_t)v.x + v.y + v.z + v.w; }

template <class T>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ p, size_t n, int reps, uint64_t *out) {
    uint64_t s = 0;
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += fold(p[i]);
    if (s == 0x123456789abcull) out[0] = s;
}
template <class T>
__global__ __launch_bounds__(256) void k_write(T *__restrict__ p, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

template <class T>
static void run_read(const char *name, const void *buf, size_t bytes, int reps, uint64_t *out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    k_read<T><<<256 * 16, 256>>>((const T *)buf, bytes / sizeof(T), reps, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("{\"kernel\": \"k_read<%s>\", \"buffer_bytes\": %zu, \"reps\": %d, \"bytes_read\": %.0f, \"ms\": %.3f, \"GBs\": %.1f}\n", name, bytes, reps,
           (double)bytes * reps, ms, (double)bytes * reps / ms / 1e6);
}
template <class T>
static void run_write(const char *name, void *buf, size_t bytes, T v) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    k_write<T><<<256 * 16, 256>>>((T *)buf, bytes / sizeof(T), v);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("{\"kernel\": \"k_write<%s>\", \"buffer_bytes\": %zu, \"bytes_written\": %.0f, \"ms\": %.3f, \"GBs\": %.1f}\n", name, bytes, (double)bytes, ms,
           (double)bytes / ms / 1e6);
}

int main() {
    const size_t big = (size_t)4 << 30, small = (size_t)64 << 20;
    void *buf = nullptr;
    uint64_t *out = nullptr;
    CK(hipMalloc(&buf, big));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(buf, 1, big));
    CK(hipDeviceSynchronize());
    // cold streaming reads (each launch evicts the Infinity Cache for the next: 4 GiB >> 256 MiB)
    run_read<unsigned short>("u16", buf, big, 1, out);
    run_read<unsigned int>("u32", buf, big, 1, out);
    run_read<uint2>("u64", buf, big, 1, out);
    run_read<uint4>("u128", buf, big, 1, out);
    // on-die re-reads: 64 MiB x 16 (one cold pass, fifteen from the Infinity Cache / L2)
    run_read<unsigned short>("u16", buf, small, 16, out);
    run_read<unsigned int>("u32", buf, small, 16, out);
    run_read<uint4>("u128", buf, small, 16, out);
    // stores
    run_write<unsigned short>("u16", buf, big, (unsigned short)7);
    run_write<unsigned int>("u32", buf, big, 7u);
    run_write<uint4>("u128", buf, big, make_uint4(1, 2, 3, 4));
    CK(hipDeviceSynchronize());
    return 0;
}
