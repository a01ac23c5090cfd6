// Runtime plumbing: error state, stream, caching allocator, device scans, matrix handles.
#include <chrono>

#include "hhx_common.h"

namespace hhx {

thread_local std::string g_err;
thread_local hipStream_t g_stream = nullptr;

// ------------------------------------------------------------------ pool
// Blocks are recycled in stream order, so a free list belongs to ONE stream.  The caller's thread(s) share the global list
// (include/haphic_hip.h: one stream at a time); a library thread that launches on a stream of its own (the file-writer thread,
// hhx_jobs.hip) installs a private Arena: its blocks are taken from and returned to that arena alone, and are handed to the global
// list (arena_donate) only after the thread has synchronised its stream.
thread_local Arena *g_arena = nullptr;

namespace {
std::mutex g_pool_mu;
std::multimap<size_t, void *> g_free;           // size class -> block (the callers' stream)
struct Live { size_t size; Arena *owner; };
std::map<void *, Live> g_live;                  // block -> size class, arena it was taken from
std::vector<void *> g_quarantine;               // freed on another thread than the owner's: no stream order to rely on, released by pool_trim
std::map<void *, int> g_prewarmed;              // taken ahead of their use (hhx_pool_prewarm) and not handed out yet -> the hhx_pool_trim calls it has survived (one, at most)

size_t size_class(size_t bytes) {
    size_t c = 256;
    while (c < bytes) {
        // power-of-two classes up to 64 MiB, then 64 MiB granules (avoids 2x waste on multi-GB buffers)
        if (c >= (size_t(64) << 20)) return (bytes + (size_t(64) << 20) - 1) / (size_t(64) << 20) * (size_t(64) << 20);
        c <<= 1;
    }
    return c;
}
}  // namespace

void *pool_alloc(size_t bytes) {
    size_t c = size_class(bytes);
    Arena *const arena = g_arena;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        // best fit: the smallest cached block that is large enough and wastes at most 3/4 of itself.
        // hipMalloc costs ~30 ms per GB on this platform (0.48 s for a 17 GB table), so re-using a
        // somewhat larger block beats a fresh allocation by orders of magnitude.
        // (multi-GB blocks: at most 1/4 wasted — a cached 88 GB dense block must not be handed to a 25 GB request while the rest
        // of the device fills up behind it)
        std::multimap<size_t, void *> &fl = arena ? arena->free : g_free;
        auto it = fl.lower_bound(c);
        if (it != fl.end() && (it->first <= 4 * c && (it->first < (size_t(16) << 30) || it->first <= c + c / 4))) {
            void *p = it->second;
            g_live[p] = Live{it->first, arena};
            fl.erase(it);
            g_prewarmed.erase(p);
            return p;
        }
    }
    void *p = nullptr;
    const bool timed = prof_enabled() && !arena;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, c);
    if (e != hipSuccess) {
        (void)hipGetLastError();                 // the failed attempt must not surface at the next launch check
        pool_trim(true);
        if (timed) prof_count("pool_trims_on_failure", 1);
        e = hipMalloc(&p, c);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fail("hipMalloc(%zu bytes) failed: %s", c, hipGetErrorString(e));
            return nullptr;
        }
    }
    static const bool pool_log = getenv("HHX_POOL_LOG") != nullptr;       // measurement: every fresh block of the callers' threads, with what the driver took
    if (pool_log && !arena)
        fprintf(stderr, "[hhx pool] fresh %.3f GB in %.1f ms\n", (double)c / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (timed) {                                 // what fresh device memory costs the caller's thread (~30 ms per GB)
        prof_count("pool_fresh_bytes", (i64)c);
        prof_count("pool_fresh_calls", 1);
        prof_count("pool_fresh_us", (i64)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count());
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_live[p] = Live{c, arena};
    return p;
}

void pool_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) return;
    // Stream-ordered reuse: every consumer of a free list runs on one stream, so a recycled block is only ever touched by work
    // enqueued after its previous user.  A block that is freed from another thread than the one it was handed to has no such
    // order: it waits in quarantine for the next pool_trim (device-wide synchronisation).
    if (it->second.owner == g_arena) (g_arena ? g_arena->free : g_free).emplace(it->second.size, p);
    else g_quarantine.push_back(p);
    g_live.erase(it);
}

// the caller guarantees that NO work on any stream still touches p (it synchronised every stream that used it): reusable by the callers' stream at once
void pool_free_synced(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) return;
    g_free.emplace(it->second.size, p);
    g_live.erase(it);
}

// p is idle on every stream and will not be needed again soon: back to the driver (hipFree waits for the device: call it from a thread that can wait)
void pool_free_to_driver(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_live.erase(p);
    }
    (void)hipFree(p);
}

// after the arena's thread synchronised its stream: its cached blocks go to the callers' list
void arena_donate(Arena *a) {
    if (!a) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &kv : a->free) g_free.emplace(kv.first, kv.second);
    a->free.clear();
}

i64 pool_cached_bytes() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    i64 b = 0;
    for (auto &kv : g_free) b += (i64)kv.first;
    return b;
}

// everything: out of device memory (pool_alloc).  Otherwise (hhx_pool_trim: "what the last step cached is of no use to the next") a block that was
// taken ahead for a later step stays — once: the second trim that finds it unused releases it
void pool_trim(bool everything, i64 keep_bytes) {
    std::vector<void *> blocks;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        // only the list of the calling thread's stream and the quarantine: another thread's arena is in use by that thread
        std::multimap<size_t, void *> &fl = g_arena ? g_arena->free : g_free;
        std::multimap<size_t, void *> kept;
        i64 kept_mid = 0;
        for (auto it = fl.rbegin(); it != fl.rend(); ++it) {             // largest first
            auto pw = everything ? g_prewarmed.end() : g_prewarmed.find(it->second);
            const bool small = it->first <= (size_t(64) << 20), mid = !small && it->first <= (size_t(8) << 30);
            if (pw != g_prewarmed.end() && pw->second++ == 0) kept.emplace(it->first, it->second);
            else if (!everything && keep_bytes > 0 && (small || (mid && kept_mid + (i64)it->first <= keep_bytes))) {
                kept.emplace(it->first, it->second);                     // hhx_pool_trim_keep: what the next step can use whatever its sizes are
                if (mid) kept_mid += (i64)it->first;
            } else blocks.push_back(it->second);
        }
        fl.swap(kept);
        if (g_arena)                             // out of memory on the arena's thread: the callers' cache goes too
            { for (auto &kv : g_free) blocks.push_back(kv.second); g_free.clear(); }
        for (void *p : blocks) g_prewarmed.erase(p);
        blocks.insert(blocks.end(), g_quarantine.begin(), g_quarantine.end());
        g_quarantine.clear();
    }
    if (!blocks.empty()) (void)hipDeviceSynchronize();
    for (void *p : blocks) (void)hipFree(p);
}

// ------------------------------------------------------------------ tuning knobs
namespace {
std::mutex g_tune_mu;
std::map<std::string, i64> g_tune;
}  // namespace
i64 tune_get(const char *name, i64 dflt) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tune.find(name);
    if (it != g_tune.end()) return it->second;
    std::string env = "HHX_";
    for (const char *c = name; *c; ++c) env += (char)toupper((unsigned char)*c);
    const char *e = getenv(env.c_str());
    const i64 v = e ? atoll(e) : dflt;
    if (e) g_tune[name] = v;
    return v;
}

// ------------------------------------------------------------------ kernel timing
namespace {
bool g_prof_on = false;
struct ProfRec { std::string name; hipEvent_t e0, e1; int launches; };
std::vector<ProfRec> g_prof_pending;
std::map<std::string, std::pair<double, i64>> g_prof_total;
std::mutex g_prof_mu;
}  // namespace

KTimer::KTimer(const char *n, int l) : name(n), launches(l) {
    if (!g_prof_on) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
    (void)hipEventRecord(e0, g_stream);
}
KTimer::~KTimer() {
    if (!e0 || !e1) return;
    (void)hipEventRecord(e1, g_stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_pending.push_back({name, e0, e1, launches});
}
namespace { std::map<std::string, i64> g_prof_counters; }
void prof_count(const char *name, i64 v) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_counters[name] += v;
}
bool prof_enabled() { return g_prof_on; }

static void prof_collect() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto &t = g_prof_total[r.name];
            t.first += ms;
            t.second += r.launches;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof_pending.clear();
}

// ------------------------------------------------------------------ exclusive scan
// Tile = 256 threads x 8 items.  Level 0 scans tiles and emits tile sums; the sums are scanned
// recursively; a final pass adds the tile offsets.
constexpr int SCAN_T = 256, SCAN_I = 8, SCAN_TILE = SCAN_T * SCAN_I;

template <class T>
__global__ __launch_bounds__(SCAN_T) void k_scan_tiles(const T *in, T *out, T *tile_sums, i64 n) {
    __shared__ T wsum[SCAN_T / HHX_WAVE];
    i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_I;
    T v[SCAN_I];
    T local = 0;
#pragma unroll
    for (int k = 0; k < SCAN_I; ++k) {
        v[k] = (base + k < n) ? in[base + k] : T(0);
        local += v[k];
    }
    // inclusive wave scan of `local`
    T incl = local;
#pragma unroll
    for (int o = 1; o < HHX_WAVE; o <<= 1) {
        T t = __shfl_up(incl, o, HHX_WAVE);
        if (lane_id() >= o) incl += t;
    }
    int w = threadIdx.x / HHX_WAVE;
    if (lane_id() == HHX_WAVE - 1) wsum[w] = incl;
    __syncthreads();
    T woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    T run = woff + incl - local;
#pragma unroll
    for (int k = 0; k < SCAN_I; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == SCAN_T - 1 && tile_sums) tile_sums[blockIdx.x] = run;
}

template <class T>
__global__ __launch_bounds__(SCAN_T) void k_scan_add(T *out, const T *tile_off, i64 n) {
    i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_I;
    T off = tile_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_I; ++k)
        if (base + k < n) out[base + k] += off;
}

template <class T>
__global__ void k_set_last(T *out, i64 n, const T *in_last_src, const T *scan_last) {
    // out[n] = out[n-1] + in[n-1]
    out[n] = scan_last[0] + in_last_src[0];
}

template <class T>
static int scan_rec(const T *in, T *out, i64 n) {
    if (n <= 0) return 0;
    i64 tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tiles == 1) {
        k_scan_tiles<T><<<1, SCAN_T, 0, g_stream>>>(in, out, nullptr, n);
        HHX_LAUNCH_CHECK();
        return 0;
    }
    DevBuf<T> sums, offs;
    if (sums.alloc(tiles) || offs.alloc(tiles)) return 1;
    k_scan_tiles<T><<<(unsigned)tiles, SCAN_T, 0, g_stream>>>(in, out, sums.p, n);
    HHX_LAUNCH_CHECK();
    HHX_TRY(scan_rec<T>(sums.p, offs.p, tiles));
    k_scan_add<T><<<(unsigned)tiles, SCAN_T, 0, g_stream>>>(out, offs.p, n);
    HHX_LAUNCH_CHECK();
    return 0;
}

template <class T>
static int scan_total(const T *in, T *out, i64 n, i64 *total_host) {
    if (n == 0) {
        HHX_HIP(hipMemsetAsync(out, 0, sizeof(T), g_stream));
        if (total_host) *total_host = 0;
        return 0;
    }
    // `in` must stay intact until out[n] is formed: read in[n-1] first via a 1-thread kernel after the scan
    HHX_TRY(scan_rec<T>(in, out, n));
    k_set_last<T><<<1, 1, 0, g_stream>>>(out, n, in + (n - 1), out + (n - 1));
    HHX_LAUNCH_CHECK();
    if (total_host) {
        T t;
        HHX_HIP(hipMemcpyAsync(&t, out + n, sizeof(T), hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        *total_host = (i64)t;
    }
    return 0;
}

void u64_copy_async(const unsigned long long *src, unsigned long long *dst, i64 n) {
    (void)hipMemcpyAsync(dst, src, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToDevice, g_stream);
}

// The int32 scan wraps silently when the true total exceeds 2^31 - 1 (a nearly dense product at n >= 46k through the
// general S1 seams): the total is first reduced in 64 bits, and a total beyond the int32 index range of hhx_csr fails
// loudly instead of sizing an output from a wrapped number.
__global__ __launch_bounds__(256) void k_sum_i32_wide(const i32 *__restrict__ in, i64 n, unsigned long long *__restrict__ total) {
    i64 s = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) s += in[i];
    s = wave_sum_i64(s);
    if (lane_id() == 0 && s) atomicAdd(total, (unsigned long long)s);
}
int exclusive_scan_i32(const i32 *in, i32 *out, i64 n, i64 *total_host) {
    if (n > 0) {
        DevBuf<unsigned long long> wide;
        if (wide.alloc(1)) return 1;
        HHX_HIP(hipMemsetAsync(wide.p, 0, sizeof(unsigned long long), g_stream));
        k_sum_i32_wide<<<(unsigned)std::max<i64>(1, std::min<i64>((n + 255) / 256, 1024)), 256, 0, g_stream>>>(in, n, wide.p);
        HHX_LAUNCH_CHECK();
        unsigned long long t = 0;
        HHX_HIP(hipMemcpyAsync(&t, wide.p, sizeof t, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        if (t > (unsigned long long)INT32_MAX)
            return fail("a matrix of %llu entries exceeds the int32 index range of hhx_csr (scipy's too): split the operand into row blocks "
                        "(hhx_csr_row_block) or use the fused iteration (hhx_expand_inflate_prune)", t);
    }
    return scan_total<i32>(in, out, n, total_host);
}
int exclusive_scan_i64(const i64 *in, i64 *out, i64 n, i64 *total_host) { return scan_total<i64>(in, out, n, total_host); }

}  // namespace hhx

using namespace hhx;

// ------------------------------------------------------------------ C ABI: runtime
extern "C" const char *hhx_last_error(void) { return g_err.c_str(); }
extern "C" int hhx_version(void) { return 100; }
extern "C" int hhx_device_count(int *count) {
    HHX_HIP(hipGetDeviceCount(count));
    return 0;
}
extern "C" int hhx_set_device(int device) {
    HHX_HIP(hipSetDevice(device));
    return 0;
}
// The caching pool recycles a block as soon as it is freed, on the assumption that every kernel of the process runs on
// ONE stream (stream order then guarantees that the next user starts after the previous one finished).  Changing the
// stream therefore drains the device first; running two threads with different streams at the same time is outside the
// contract (include/haphic_hip.h).
extern "C" int hhx_set_stream(void *s) {
    if ((hipStream_t)s != g_stream) HHX_HIP(hipDeviceSynchronize());
    g_stream = (hipStream_t)s;
    return 0;
}
extern "C" int hhx_synchronize(void) {
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
// Blocks of the given sizes are taken from the driver NOW and left in the pool's cache: called from a helper thread of the caller while a long kernel runs on
// the caller's stream (the expansion of the inflation sweep, 300 ms), it hides what the next step's pools would cost in fresh device memory
// (12-30 ms per GB).  Nothing has touched the blocks: any stream may take them.
extern "C" int hhx_pool_prewarm(int32_t n, const int64_t *bytes) {
    if (n < 0 || (n && !bytes)) return fail("hhx_pool_prewarm: bad argument");
    for (i32 k = 0; k < n; ++k) {                        // in the order given (what is needed first comes first); every block is in the cache as soon as it exists
        if (bytes[k] <= 0) continue;
        void *p = nullptr;
        if (hipMalloc(&p, size_class((size_t)bytes[k])) != hipSuccess) { (void)hipGetLastError(); break; }     // no room: what is there is enough of a head start
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_free.emplace(size_class((size_t)bytes[k]), p);
        g_prewarmed[p] = 0;
    }
    return 0;
}

extern "C" int hhx_pool_trim(void) {
    pool_trim(false);
    return 0;
}
extern "C" int hhx_pool_trim_keep(int64_t keep_bytes) {
    if (keep_bytes < 0) return fail("hhx_pool_trim_keep: bad argument");
    pool_trim(false, keep_bytes);
    return 0;
}
// The knobs the kernels read (tune_get): which kernel class / arithmetic / layout a call takes.  Every setting of every knob gives
// the same results (the verification tests switch classes with them and compare bits); an unknown name is refused.
static const char *const k_tune_names[] = {"cls", "cls_nc", "cls_balance", "links_integer", "links_sym", "hash_max", "tile_u", "win_batch",
                                           "cache_slice_mb", "dense_tri", "reuse", "row_order", "dense_seed_hint", "block_tiles",
#ifdef HHX_PROBE_BUILD
                                           "probe",          // measurement build only: the LDS atomics switched off, results are garbage
#endif
                                           nullptr};
extern "C" int hhx_tune(const char *name, int64_t value) {
    if (!name) return fail("null name");
    bool known = false;
    for (const char *const *k = k_tune_names; *k; ++k) known |= strcmp(*k, name) == 0;
    if (!known) return fail("hhx_tune: unknown knob '%s'", name);
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (value == INT64_MIN) g_tune.erase(name);              // back to the default (or HHX_<NAME> of the environment)
    else g_tune[name] = value;
    return 0;
}
extern "C" int hhx_profile_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}
extern "C" int hhx_profile_reset(void) {
    prof_collect();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_total.clear();
    g_prof_counters.clear();
    return 0;
}
extern "C" int hhx_profile_counter(const char *name, i64 *value) {
    if (!name || !value) return fail("null pointer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = g_prof_counters.find(name);
    *value = it == g_prof_counters.end() ? 0 : it->second;
    return 0;
}
extern "C" int hhx_profile_get(const char *kernel, double *total_ms, i64 *launches) {
    if (!kernel) return fail("null kernel name");
    prof_collect();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = g_prof_total.find(kernel);
    if (total_ms) *total_ms = it == g_prof_total.end() ? 0.0 : it->second.first;
    if (launches) *launches = it == g_prof_total.end() ? 0 : it->second.second;
    return 0;
}

// ------------------------------------------------------------------ C ABI: matrices
static int csr_alloc(i32 n_rows, i32 n_cols, i64 nnz, hhx_csr **out) {
    if (n_rows < 0 || n_cols < 0 || nnz < 0) return fail("negative matrix dimension");
    if (nnz > INT32_MAX) return fail("nnz %lld exceeds the int32 index range of scipy CSC", (long long)nnz);
    hhx_csr *m = new hhx_csr();
    m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz;
    if (m->indptr.alloc((size_t)n_rows + 1) || m->indices.alloc((size_t)nnz) || m->data.alloc((size_t)nnz)) {
        delete m;
        return 1;
    }
    *out = m;
    return 0;
}
int hhx_csr_alloc_internal(i32 n_rows, i32 n_cols, i64 nnz, hhx_csr **out) { return csr_alloc(n_rows, n_cols, nnz, out); }

extern "C" int hhx_csr_from_host(i32 n_rows, i32 n_cols, const i32 *indptr, const i32 *indices, const float *data,
                                 hhx_csr **out) {
    if (!indptr || !out) return fail("hhx_csr_from_host: null pointer");
    i64 nnz = indptr[n_rows];
    hhx_csr *m = nullptr;
    HHX_TRY(csr_alloc(n_rows, n_cols, nnz, &m));
    hipError_t e = hipMemcpyAsync(m->indptr.p, indptr, sizeof(i32) * ((size_t)n_rows + 1), hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(m->indices.p, indices, sizeof(i32) * (size_t)nnz, hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(m->data.p, data, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { delete m; return fail("hhx_csr_from_host: copy failed: %s", hipGetErrorString(e)); }
    *out = m;
    return 0;
}

extern "C" int hhx_csr_from_device(i32 n_rows, i32 n_cols, i64 nnz, const i32 *indptr, const i32 *indices,
                                   const float *data, hhx_csr **out) {
    hhx_csr *m = nullptr;
    HHX_TRY(csr_alloc(n_rows, n_cols, nnz, &m));
    hipError_t e = hipMemcpyAsync(m->indptr.p, indptr, sizeof(i32) * ((size_t)n_rows + 1), hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(m->indices.p, indices, sizeof(i32) * (size_t)nnz, hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(m->data.p, data, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToDevice, g_stream);
    if (e != hipSuccess) { delete m; return fail("hhx_csr_from_device: copy failed: %s", hipGetErrorString(e)); }
    *out = m;
    return 0;
}

extern "C" int hhx_csr_shape(const hhx_csr *m, i32 *n_rows, i32 *n_cols, i64 *nnz) {
    if (!m) return fail("null matrix");
    if (n_rows) *n_rows = m->n_rows;
    if (n_cols) *n_cols = m->n_cols;
    if (nnz) *nnz = m->nnz;
    return 0;
}

extern "C" int hhx_csr_to_host(const hhx_csr *m, i32 *indptr, i32 *indices, float *data) {
    if (!m) return fail("null matrix");
    if (indptr) HHX_HIP(hipMemcpyAsync(indptr, m->indptr.p, sizeof(i32) * ((size_t)m->n_rows + 1), hipMemcpyDeviceToHost, g_stream));
    if (indices && m->nnz) HHX_HIP(hipMemcpyAsync(indices, m->indices.p, sizeof(i32) * (size_t)m->nnz, hipMemcpyDeviceToHost, g_stream));
    if (data && m->nnz) HHX_HIP(hipMemcpyAsync(data, m->data.p, sizeof(float) * (size_t)m->nnz, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int hhx_csr_device_ptrs(const hhx_csr *m, void **indptr, void **indices, void **data) {
    if (!m) return fail("null matrix");
    if (indptr) *indptr = m->indptr.p;
    if (indices) *indices = m->indices.p;
    if (data) *data = m->data.p;
    return 0;
}

extern "C" int hhx_csr_copy(const hhx_csr *m, hhx_csr **out) {
    if (!m) return fail("null matrix");
    return hhx_csr_from_device(m->n_rows, m->n_cols, m->nnz, m->indptr.p, m->indices.p, m->data.p, out);
}

__global__ void k_rebase_indptr(const i32 *src, i32 *dst, i32 n, i32 base) {
    i32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) dst[i] = src[i] - base;
}

extern "C" int hhx_csr_row_block(const hhx_csr *m, i32 r0, i32 r1, hhx_csr **out) {
    if (!m) return fail("null matrix");
    if (r0 < 0 || r1 < r0 || r1 > m->n_rows) return fail("row block [%d,%d) out of range", r0, r1);
    i32 ends[2];
    HHX_HIP(hipMemcpyAsync(&ends[0], m->indptr.p + r0, sizeof(i32), hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipMemcpyAsync(&ends[1], m->indptr.p + r1, sizeof(i32), hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    i64 nnz = ends[1] - ends[0];
    hhx_csr *b = nullptr;
    HHX_TRY(csr_alloc(r1 - r0, m->n_cols, nnz, &b));
    i32 n = r1 - r0;
    k_rebase_indptr<<<(n + 1 + 255) / 256, 256, 0, g_stream>>>(m->indptr.p + r0, b->indptr.p, n, ends[0]);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(b->indices.p, m->indices.p + ends[0], sizeof(i32) * (size_t)nnz, hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(b->data.p, m->data.p + ends[0], sizeof(float) * (size_t)nnz, hipMemcpyDeviceToDevice, g_stream);
    if (e != hipSuccess) { delete b; return fail("hhx_csr_row_block: %s", hipGetErrorString(e)); }
    *out = b;
    return 0;
}

// the row blocks stacked in order (equal column counts); the total number of entries must stay below 2^31
extern "C" int hhx_csr_vstack(i32 n_blocks, const hhx_csr *const *blocks, hhx_csr **out) {
    if (n_blocks < 1 || !blocks || !out) return fail("hhx_csr_vstack: bad argument");
    i64 rows = 0, nnz = 0;
    for (i32 k = 0; k < n_blocks; ++k) {
        if (!blocks[k] || blocks[k]->n_cols != blocks[0]->n_cols) return fail("hhx_csr_vstack: block %d is null or has another column count", k);
        rows += blocks[k]->n_rows;
        nnz += blocks[k]->nnz;
    }
    if (rows > INT32_MAX || nnz > INT32_MAX) return fail("hhx_csr_vstack: %lld rows / %lld entries exceed int32", (long long)rows, (long long)nnz);
    hhx_csr *m = nullptr;
    HHX_TRY(csr_alloc((i32)rows, blocks[0]->n_cols, nnz, &m));
    i64 r = 0, z = 0;
    hipError_t e = hipSuccess;
    for (i32 k = 0; k < n_blocks && e == hipSuccess; ++k) {
        const hhx_csr *b = blocks[k];
        // indptr of the block shifted by the entries above it; the last block also writes the closing entry
        k_rebase_indptr<<<(b->n_rows + 1 + 255) / 256, 256, 0, g_stream>>>(b->indptr.p, m->indptr.p + r, k + 1 == n_blocks ? b->n_rows : b->n_rows - 1, (i32)-z);
        e = hipGetLastError();
        if (e == hipSuccess && b->nnz) e = hipMemcpyAsync(m->indices.p + z, b->indices.p, sizeof(i32) * (size_t)b->nnz, hipMemcpyDeviceToDevice, g_stream);
        if (e == hipSuccess && b->nnz) e = hipMemcpyAsync(m->data.p + z, b->data.p, sizeof(float) * (size_t)b->nnz, hipMemcpyDeviceToDevice, g_stream);
        r += b->n_rows;
        z += b->nnz;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { delete m; return fail("hhx_csr_vstack: %s", hipGetErrorString(e)); }
    *out = m;
    return 0;
}

// ---- the per-iteration exchange of the row-block MCL (haphic_amd/sharded.py exchange_rows): a CSR row block as ONE int32 message
// [row lengths (n_rows) | column indices (nnz) | float32 value bits (nnz)], and the world's messages back into one CSR matrix
__global__ __launch_bounds__(256) void k_row_lengths(const i32 *__restrict__ indptr, i32 n, i32 *__restrict__ out) {
    for (i32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = indptr[i + 1] - indptr[i];
}
extern "C" int hhx_csr_pack_block(const hhx_csr *m, void *dst_dev, i64 capacity_words) {
    if (!m || !dst_dev) return fail("hhx_csr_pack_block: null pointer");
    const i64 need = (i64)m->n_rows + 2 * m->nnz;
    if (capacity_words < need) return fail("hhx_csr_pack_block: message of %lld words, buffer of %lld", (long long)need, (long long)capacity_words);
    i32 *dst = (i32 *)dst_dev;
    if (m->n_rows) k_row_lengths<<<(unsigned)std::min<i64>(((i64)m->n_rows + 255) / 256, 4096), 256, 0, g_stream>>>(m->indptr.p, m->n_rows, dst);
    HHX_LAUNCH_CHECK();
    if (m->nnz) {
        HHX_HIP(hipMemcpyAsync(dst + m->n_rows, m->indices.p, sizeof(i32) * (size_t)m->nnz, hipMemcpyDeviceToDevice, g_stream));
        HHX_HIP(hipMemcpyAsync(dst + m->n_rows + m->nnz, m->data.p, sizeof(float) * (size_t)m->nnz, hipMemcpyDeviceToDevice, g_stream));
    }
    return 0;
}
// message b starts at packed + b * stride_words and holds rows[b] rows / nnz[b] entries; the blocks are stacked in order.
// The row pointer is ONE device scan over the concatenated row lengths (no host-side cumsum, no per-block rebase).
extern "C" int hhx_csr_unpack_blocks(i32 n_blocks, const i64 *rows, const i64 *nnz, const void *packed_dev, i64 stride_words, i32 n_cols,
                                     hhx_csr **out) {
    if (n_blocks < 1 || !rows || !nnz || !packed_dev || !out) return fail("hhx_csr_unpack_blocks: bad argument");
    i64 R = 0, Z = 0;
    for (i32 b = 0; b < n_blocks; ++b) {
        if (rows[b] < 0 || nnz[b] < 0 || rows[b] + 2 * nnz[b] > stride_words) return fail("hhx_csr_unpack_blocks: message %d does not fit its stride", b);
        R += rows[b];
        Z += nnz[b];
    }
    if (R > INT32_MAX || Z > INT32_MAX) return fail("hhx_csr_unpack_blocks: %lld rows / %lld entries exceed int32", (long long)R, (long long)Z);
    hhx_csr *m = nullptr;
    HHX_TRY(csr_alloc((i32)R, n_cols, Z, &m));
    DevBuf<i32> lens;
    if (lens.alloc((size_t)R + 1)) { delete m; return 1; }
    const i32 *src = (const i32 *)packed_dev;
    i64 r = 0, z = 0;
    hipError_t e = hipSuccess;
    for (i32 b = 0; b < n_blocks && e == hipSuccess; ++b) {
        const i32 *msg = src + (size_t)b * (size_t)stride_words;
        if (rows[b]) e = hipMemcpyAsync(lens.p + r, msg, sizeof(i32) * (size_t)rows[b], hipMemcpyDeviceToDevice, g_stream);
        if (e == hipSuccess && nnz[b]) e = hipMemcpyAsync(m->indices.p + z, msg + rows[b], sizeof(i32) * (size_t)nnz[b], hipMemcpyDeviceToDevice, g_stream);
        if (e == hipSuccess && nnz[b]) e = hipMemcpyAsync(m->data.p + z, msg + rows[b] + nnz[b], sizeof(float) * (size_t)nnz[b], hipMemcpyDeviceToDevice, g_stream);
        r += rows[b];
        z += nnz[b];
    }
    if (e != hipSuccess) { delete m; return fail("hhx_csr_unpack_blocks: %s", hipGetErrorString(e)); }
    i64 total = 0;
    if (exclusive_scan_i32(lens.p, m->indptr.p, R, &total)) { delete m; return 1; }        // synchronises: `lens` dies with this frame
    if (total != Z) { delete m; return fail("hhx_csr_unpack_blocks: the row lengths add up to %lld, the headers to %lld", (long long)total, (long long)Z); }
    *out = m;
    return 0;
}

// bytes held by the library's caching pool (free for the next allocation of the library, not for the driver's hipMemGetInfo)
extern "C" int hhx_pool_cached_bytes(i64 *bytes) {
    if (!bytes) return fail("null pointer");
    *bytes = pool_cached_bytes();
    return 0;
}

extern "C" int hhx_mem_info(i64 *free_bytes, i64 *total_bytes) {
    size_t f = 0, t = 0;
    HHX_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (i64)f;
    if (total_bytes) *total_bytes = (i64)t;
    return 0;
}

extern "C" int hhx_csr_free(hhx_csr *m) {
    delete m;
    return 0;
}
