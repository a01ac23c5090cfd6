// LDS atomic throughput on gfx950 for the accumulator patterns of the fused MCL expansion (measurement only).
// 1024-thread workgroups (16 waves, one per CU), each lane issues 8 atomics per step to acc[idx]; the index of lane l
// in atomic j is drawn once from a pattern and then advanced by a multiple of 32 slots per step, which preserves its
// residue (bank) class.  Patterns: 0 own slot per lane (conflict-free), 1 random slots, 2 random with distinct
// residues mod 32 inside every 32-lane half, 3 distinct residues mod 16 inside every 16-lane group,
// 4 distinct residues mod 32 across lanes l and l+32 sharing a bank pair (2-way), 5 random inside a 2048-slot range.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int CAP = 16384;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <class T> __device__ __forceinline__ void add(T *p, T v) { atomicAdd(p, v); }

template <class T, int PATTERN>
__global__ __launch_bounds__(1024) void k_atomic(int steps, T *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *acc = (T *)smem;
    for (int t = threadIdx.x; t < CAP; t += 1024) acc[t] = (T)0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t idx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t h = mix(threadIdx.x * 131u + j * 7919u + blockIdx.x * 104729u);
        uint32_t v;
        if (PATTERN == 0) v = threadIdx.x + 1024 * j;
        else if (PATTERN == 1) v = h % CAP;
        else if (PATTERN == 2) v = ((h % (CAP / 32)) * 32) + (lane & 31);
        else if (PATTERN == 3) v = ((h % (CAP / 16)) * 16) + (lane & 15);
        else if (PATTERN == 4) v = ((h % (CAP / 64)) * 64) + lane;
        else v = h % 2048;
        idx[j] = v;
    }
    const T one = (T)1;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            add(&acc[idx[j]], one);
            idx[j] += 32 * 37;
            if (idx[j] >= CAP) idx[j] -= CAP;
        }
    }
    __syncthreads();
    T s = 0;
    for (int t = threadIdx.x; t < CAP; t += 1024) s += acc[t];
    if (s == (T)123456789) out[0] = s;
}

template <class T, int PATTERN>
static void run(const char *tname, const char *pname, T *out) {
    const size_t lds = (size_t)CAP * sizeof(T);
    CK(hipFuncSetAttribute((const void *)k_atomic<T, PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int steps = 20000;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        k_atomic<T, PATTERN><<<256, 1024, lds>>>(steps, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double wave_instr_per_cu = 16.0 * 8 * steps;
    const double clk = best * 1e-3 * 2.4e9;
    printf("{\"type\": \"%s\", \"pattern\": \"%s\", \"ms\": %.3f, \"clk_per_wave_instr\": %.2f, \"lanes_per_clk_per_cu\": %.2f, \"G_atomics_s\": %.0f}\n", tname, pname,
           best, clk / wave_instr_per_cu, 64.0 * wave_instr_per_cu / clk, 256.0 * 64 * wave_instr_per_cu / best / 1e6);
    fflush(stdout);
}

template <class T>
static void all(const char *tname, void *out) {
    run<T, 0>(tname, "own slot (conflict-free)", (T *)out);
    run<T, 1>(tname, "random", (T *)out);
    run<T, 2>(tname, "distinct mod 32 per 32-lane half", (T *)out);
    run<T, 3>(tname, "distinct mod 16 per 16-lane group", (T *)out);
    run<T, 4>(tname, "lane-aligned mod 64", (T *)out);
    run<T, 5>(tname, "random in 2048 slots", (T *)out);
}

int main() {
    void *out = nullptr;
    CK(hipMalloc(&out, 64));
    all<double>("f64", out);
    all<unsigned long long>("u64", out);
    all<float>("f32", out);
    all<unsigned int>("u32", out);
    return 0;
}
