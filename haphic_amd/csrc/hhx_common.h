// Shared host/device plumbing of libhaphic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/haphic_hip.h"

typedef int32_t i32;
typedef int64_t i64;
typedef uint32_t u32;
typedef uint64_t u64;

namespace hhx {

// ------------------------------------------------------------------ errors
extern thread_local std::string g_err;
extern thread_local hipStream_t g_stream;

inline int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HHX_HIP(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return hhx::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define HHX_TRY(expr)          \
    do {                       \
        int r_ = (expr);       \
        if (r_) return r_;     \
    } while (0)

#define HHX_LAUNCH_CHECK() HHX_HIP(hipGetLastError())

// ------------------------------------------------------------------ caching device allocator
// MCL allocates a handful of buffers per iteration; hipMalloc/hipFree cost ~100 us each and
// hipFree synchronises the device, so blocks are recycled by power-of-two size class.
void *pool_alloc(size_t bytes);   // nullptr on failure (g_err set)
void pool_free(void *p);
void pool_trim(bool everything = true, i64 keep_bytes = 0);   // false: a block taken ahead of its use (hhx_pool_prewarm) stays cached, once; keep_bytes: hhx_pool_trim_keep
i64 pool_cached_bytes();
// a private free list for a library thread that launches on its own stream (hhx_jobs.hip); nullptr = the callers' list
struct Arena { std::multimap<size_t, void *> free; };
extern thread_local Arena *g_arena;
void pool_free_synced(void *p);   // p is idle on every stream: straight to the callers' list
void pool_free_to_driver(void *p);  // p is idle on every stream: hipFree (waits for the device)
void arena_donate(Arena *a);      // the arena's thread has synchronised its stream: its cached blocks go to the callers' list

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    int alloc(size_t count) {
        release();
        n = count;
        p = (T *)pool_alloc((count ? count : 1) * sizeof(T));
        return p ? 0 : 1;
    }
    void release() {
        if (p) pool_free(p);
        p = nullptr;
        n = 0;
    }
};

// ------------------------------------------------------------------ optional per-kernel timing
// RAII: records a HIP event pair on g_stream around the launches in its scope when profiling is on.
struct KTimer {
    const char *name;
    int launches;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    explicit KTimer(const char *n, int launches = 1);
    ~KTimer();
};

// named event counters (only while profiling is enabled): products / entries streamed by a kernel class, so that
// bench.py can turn HIP-event times into achieved bytes per second
void prof_count(const char *name, i64 v);
bool prof_enabled();

// ------------------------------------------------------------------ tuning knobs (hhx_tune): measurement / experiment
// switches read at call time; unset -> the environment variable HHX_<NAME> (upper case) -> the default
i64 tune_get(const char *name, i64 dflt);

// ------------------------------------------------------------------ scans / reductions (hhx_scan.hip)
// exclusive scan of n int32 counts into int32 offsets out[0..n] (out[n] = total); total returned
// through *total_host after a stream sync.  in and out may alias only if out == in is NOT used.
int exclusive_scan_i32(const i32 *in, i32 *out, i64 n, i64 *total_host);
int exclusive_scan_i64(const i64 *in, i64 *out, i64 n, i64 *total_host);

// ------------------------------------------------------------------ the file-writer thread (hhx_jobs.hip)
// run() writes HT_links.pkl, paired_links.clm and full_links.pkl between the seams (:2879 :2888 :2929) and reads none of them: the *_async entry
// points queue the file on a library-owned host thread (two lanes, each with its own non-blocking stream and pool arena; the jobs of a lane run in
// submission order) and return; hhx_files_join waits and reports the first failure.  `handle`: the ingest handle a job reads (hhx_ingest_destroy waits for it).
int files_submit(const std::string &what, const void *handle, std::function<int()> job, int lane = 1);      // lane 0: the byte sinks, lane 1: everything else
void files_wait_handle(const void *handle);
// the writers behind the synchronous and the queued entry points: an open file descriptor in (closed by the callee, whatever happens)
int write_link_pickle_fd(int fd, const char *path, i64 n_keys, const i32 *name_i, const i32 *name_j, const i64 *count, i32 n_names,
                         const uint8_t *names_blob, const i64 *name_off, i64 *n_bytes);

}  // namespace hhx

// ------------------------------------------------------------------ matrix handle
struct hhx_csr {
    i32 n_rows = 0, n_cols = 0;
    i64 nnz = 0;
    hhx::DevBuf<i32> indptr;
    hhx::DevBuf<i32> indices;
    hhx::DevBuf<float> data;
};

// rows of the expanded matrix M^e as plain float32 (the inflation sweep; hhx_expand.hip): 0 = no entry
struct hhx_dense {
    i32 n_rows = 0, n_cols = 0;
    i64 ld = 0;                         // row pitch in floats: n_cols rounded up to a 128-byte line (rows and 64-column tiles start on line boundaries)
    i32 cap_win = 0, n_win = 0;         // the column-window plan of the expansion that filled it (summation order of the epilogue)
    mutable i64 last_cand = 0, last_out = 0;    // pool demand of the previous hhx_dense_inflate_prune (sizes the next call's pools) ...
    mutable double last_inflation = 0.0;        // ... and the inflation it was measured at
    bool integer = false;               // x holds y = float(S_ij) of the integer arithmetic; the entry of M^2 is float(y / row_div[i])
    bool tri = false;                   // all rows, symmetric: x holds the upper block triangle alone (hhx_expand.hip: tri_row_off), ldn = n_win * cap_win
    i64 ldn = 0;
    hhx::DevBuf<float> x;
    hhx::DevBuf<double> row_div;
};

// the right operand of iteration 0 described as a link matrix (hhx_mcl.hip normalise_links -> hhx_expand.hip)
struct hhx_links_operand {
    const unsigned short *n16 = nullptr;    // link count of every entry of the matrix
    const double *row_sum = nullptr;        // d_k: L1 row sums
    const u64 *W = nullptr;                 // rint(2^shift / d_k): non-null selects the integer arithmetic (symmetric matrix, d_max <= 2^18)
    int shift = 0;
    i32 a_row0 = 0;                         // the left operand is rows [a_row0, a_row0 + n_rows) of the same matrix ...
    i64 a_off = 0;                          // ... whose first entry is entry a_off of the matrix
    int sym = 0;                            // dense mode over all rows: compute the upper block triangle, transpose the rest
};

// ------------------------------------------------------------------ device helpers
#ifdef __HIPCC__
#define HHX_WAVE 64

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, HHX_WAVE);
    return __shfl(v, 0, HHX_WAVE);
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, HHX_WAVE));
    return __shfl(v, 0, HHX_WAVE);
}
__device__ __forceinline__ i32 wave_sum_i32(i32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, HHX_WAVE);
    return __shfl(v, 0, HHX_WAVE);
}
__device__ __forceinline__ i64 wave_sum_i64(i64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, HHX_WAVE);
    return __shfl(v, 0, HHX_WAVE);
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & (HHX_WAVE - 1); }
// x^r for the inflation step (`matrix.power(inflation)` :2038 on float32 data -> numpy powf), x >= 0.  exp2(r * log2(x)) in double:
// relative error ~ |r log2 x| * 2^-52 <= ~1e-14, i.e. the float32 result differs from the correctly rounded power in about one case
// in 10^6 (numpy's SIMD powf itself is within 1 ulp of libm's) — and it costs a third of the double-double pow() of the device
// library, which is what the 19 non-quadratic inflations of the sweep spend their time in.  One definition for every kernel.
__device__ __forceinline__ float hhx_powr(float x, double r) {
    return x > 0.0f ? (float)exp2(r * log2((double)x)) : 0.0f;
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release / acquire over ALL address
// spaces: on gfx950 it waits for every outstanding global load and store of the wave (s_waitcnt vmcnt(0)), so a prefetch
// issued before it is no prefetch and every phase of a kernel pays the drain of the stores of the phase before.  Use this
// where the waves of a workgroup exchange data through LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
