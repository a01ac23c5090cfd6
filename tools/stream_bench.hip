// Prototype of the class-stream inner loop (measurement only): 1024-thread workgroups that own (almost) all of a
// CU's LDS, every wave streaming pseudo-random SEGMENTS of 16-bit columns from a large buffer with a ring of R
// tiles in flight, each tile = one 16-byte lane load (8 columns per lane, 512 per wave), and adding a wave-uniform
// double to acc[col] in LDS for every valid entry (ds_add_f64) — or sinking it in a register (MODE 1), or skipping the
// loads (MODE 2: LDS atomics only).  Answers: how deep must the ring be, and what do the LDS atomics cost.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int CAP = 16704;          // accumulator slots (columns of one window at n = 100k)
constexpr int NDUMMY = 64;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

struct Tile { uint4 x; int lo, hi; double g; };

// MODE 0: loads + ds_add_f64; 1: loads + register sink; 2: no loads (columns from a hash), ds_add_f64
template <int R, int MODE>
__global__ __launch_bounds__(1024) void k_stream(const unsigned short *__restrict__ buf, uint32_t n_entries, int seg_len, int segs_per_wave,
                                                 double *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *acc = (double *)smem;
    for (int t = threadIdx.x; t < CAP + NDUMMY; t += 1024) acc[t] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wid = blockIdx.x * 16 + wave;
    const int dummy = CAP + lane;
    Tile ring[R];
    double sink = 0.0;
    auto fetch = [&](Tile &t, int i) {
        const bool valid = i < segs_per_wave;
        const uint32_t h = mix(wid * 7919u + (uint32_t)i);
        const uint32_t qb = valid ? h % (n_entries - 4096) : 0;
        const uint32_t q0 = qb & ~7u;
        t.lo = (int)(qb - q0);
        t.hi = valid ? t.lo + seg_len : 0;
        t.g = 1.0 + (double)(h & 255);
        if (MODE != 2) t.x = *reinterpret_cast<const uint4 *>(buf + q0 + lane * 8);
        else { const uint32_t a = mix(h + lane), b = mix(a); t.x = make_uint4(a & 0x3fff3fffu, (a >> 1) & 0x3fff3fffu, b & 0x3fff3fffu, (b >> 1) & 0x3fff3fffu); }
    };
    auto consume = [&](const Tile &t) {
        const uint32_t w[4] = {t.x.x, t.x.y, t.x.z, t.x.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = lane * 8 + j;
            const uint32_t col = (j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu);
            const bool ok = pos >= t.lo && pos < t.hi;
            if (MODE == 1) sink += ok ? t.g + (double)col : 0.0;
            else atomicAdd(&acc[ok ? (int)col : dummy], t.g);
        }
    };
#pragma unroll
    for (int r = 0; r < R - 1; ++r) fetch(ring[r], r);
    for (int i = 0; i < segs_per_wave; i += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            fetch(ring[(r + R - 1) % R], i + r + R - 1);
            consume(ring[r]);
        }
    }
    __syncthreads();
    double s = sink;
    for (int t = threadIdx.x; t < CAP; t += 1024) s += acc[t];
    if (s == 0.123456) out[0] = s;
}

template <int R, int MODE>
static void run(const char *name, const unsigned short *buf, uint32_t n_entries, int seg_len, int segs_per_wave, double *out) {
    const size_t lds = (size_t)(CAP + NDUMMY) * 8;
    CK(hipFuncSetAttribute((const void *)k_stream<R, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        k_stream<R, MODE><<<256, 1024, lds>>>(buf, n_entries, seg_len, segs_per_wave, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double products = 256.0 * 16 * segs_per_wave * seg_len;
    printf("{\"variant\": \"%s\", \"ring\": %d, \"mode\": %d, \"seg_len\": %d, \"ms\": %.3f, \"Gproducts_s\": %.1f, \"stream_GBs\": %.1f, \"products_per_clk_per_cu\": %.2f}\n",
           name, R, MODE, seg_len, best, products / best / 1e6, products * 2 / best / 1e6, products / (best * 1e-3) / 256 / 2.4e9);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)1 << 30;                 // 1 GiB of 16-bit columns: beyond the 256 MiB Infinity Cache
    const uint32_t n_entries = (uint32_t)(bytes / 2);
    unsigned short *buf = nullptr;
    double *out = nullptr;
    CK(hipMalloc(&buf, bytes + 65536));
    CK(hipMalloc(&out, 8));
    {
        std::vector<unsigned short> h(n_entries + 32768);
        uint32_t s = 12345;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)((s >> 8) % CAP); }
        CK(hipMemcpy(buf, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    const int spw = 20000;                                // segments per wave: 256 * 16 * 20000 * 415 = 3.4e10 products
    for (int seg : {415, 160}) {
        run<1, 0>("load+ds_add", buf, n_entries, seg, spw, out);
        run<2, 0>("load+ds_add", buf, n_entries, seg, spw, out);
        run<4, 0>("load+ds_add", buf, n_entries, seg, spw, out);
        run<8, 0>("load+ds_add", buf, n_entries, seg, spw, out);
        run<2, 1>("load+sink", buf, n_entries, seg, spw, out);
        run<4, 1>("load+sink", buf, n_entries, seg, spw, out);
        run<8, 1>("load+sink", buf, n_entries, seg, spw, out);
        run<2, 2>("ds_add only", buf, n_entries, seg, spw, out);
    }
    return 0;
}
