// Issue rate of the VALU instructions the stream loops of hhx_expand.hip are made of (gfx950): cycles per wave instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip && ./tools/valu_rate        (one JSON line)
// Every kernel runs 8 independent dependency chains per lane, 4 waves per SIMD on every CU: the chains hide the latency, what is measured is issue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 4096;
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, float seed) {
    double d[8]; float f[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + seed * (threadIdx.x + i); f[i] = 1.0f + seed * (threadIdx.x + i); u[i] = threadIdx.x * 2654435761u + i; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) f[i] = __builtin_fmaf(f[i], 1.0000001f, 0.5f);                                   // v_fma_f32
            if (OP == 1) d[i] = __builtin_fma(d[i], 1.0000001, 0.5);                                      // v_fma_f64
            if (OP == 2) d[i] = d[i] * 1.0000001;                                                         // v_mul_f64
            if (OP == 3) d[i] = d[i] + 0.5;                                                               // v_add_f64
            if (OP == 4) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i])); f[i] = (float)u[i]; u[i] += 1; }   // + v_cvt_f32_u32 + v_add_u32 (subtract OP 6 twice)
            if (OP == 5) u[i] = __umul24(u[i], 0x9e3779u) + 1u;                                           // v_mad_u32_u24
            if (OP == 6) { asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(u[i])); u[i] += 1; }    // v_cvt_f32_u32 + v_add_u32
            if (OP == 7) u[i] = u[i] * 0x9e3779b1u + 1u;                                                  // v_mul_lo_u32 + add
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + f[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
double run(double *out, int cus, double ghz) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = cus * 4;                                    // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    k<OP><<<grid, 256>>>(out, 1e-9f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k<OP><<<grid, 256>>>(out, 1e-9f);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr_per_simd = 4.0 * ITER * 8;           // 4 waves x ITER x 8 instructions of the kind
    return ms * 1e-3 * ghz * 1e9 / wave_instr_per_simd;          // cycles per wave instruction and SIMD
}
int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * p.multiProcessorCount * 4 * 256));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"cycles_per_wave_instruction_per_simd\": {", p.gcnArchName, p.multiProcessorCount, ghz);
    printf("\"v_fma_f32\": %.2f, ", run<0>(out, p.multiProcessorCount, ghz));
    printf("\"v_fma_f64\": %.2f, ", run<1>(out, p.multiProcessorCount, ghz));
    printf("\"v_mul_f64\": %.2f, ", run<2>(out, p.multiProcessorCount, ghz));
    printf("\"v_add_f64\": %.2f, ", run<3>(out, p.multiProcessorCount, ghz));
    printf("\"v_cvt_f64_f32 + v_cvt_f32_u32 + v_add_u32\": %.2f, ", run<4>(out, p.multiProcessorCount, ghz));
    printf("\"v_mad_u32_u24\": %.2f, ", run<5>(out, p.multiProcessorCount, ghz));
    printf("\"v_cvt_f32_u32 + v_add_u32\": %.2f, ", run<6>(out, p.multiProcessorCount, ghz));
    printf("\"v_mul_lo_u32 + v_add_u32\": %.2f}}\n", run<7>(out, p.multiProcessorCount, ghz));
    return 0;
}
