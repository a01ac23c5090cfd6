// Expansion: C = A * B on CSR(T) — replaces sparse_dot_mkl.dot_product_mkl (MKL mkl_sparse_spmm,
// float32) at scripts/HapHiC_cluster.py:2017-2023.
//
// Row-wise Gustavson, one 256-thread workgroup per output row, everything staged in LDS:
//   * an n_cols-bit BITMAP of the row's output columns (ds_or_b32) — 12.5 KB at n = 100k.  A popcount
//     prefix over the bitmap words turns a column into its rank inside the sorted output row, so the
//     output comes out SORTED BY COLUMN with no sort and no hash probing, and the symbolic pass
//     (nnz per row) is the same bitmap + one popcount reduction.
//   * a compact array of 64-bit FIXED-POINT accumulators indexed by that rank (ds_add_u64).  Each
//     float32 product is exact in double (24x24-bit mantissas), scaled by 2^shift and truncated to an
//     integer; integer adds commute, so the sum is independent of the order in which lanes, waves or
//     GPUs deliver the products — bit-reproducible without serialising anything.
//   * each wave walks one row k of B at a time with lanes striding its entries: 256 contiguous bytes of
//     indices and of values per wave-instruction (coalesced; rows of B are re-read by many workgroups
//     and live in L2 / Infinity Cache — T after the first prune is tens to hundreds of MB).
// Rows whose output does not fit the LDS accumulator window are processed in several windows
// (products outside the current window are skipped), so any row length is handled.
#include "hhx_common.h"

using namespace hhx;

int hhx_csr_alloc_internal(i32 n_rows, i32 n_cols, i64 nnz, hhx_csr **out);
int hhx_expand_raw(const hhx_csr *a, const hhx_csr *b, int fx_shift, hhx_csr **out, i64 *n_products);   // hhx_expand.hip

namespace {

constexpr int SG_T = 256;
constexpr int SG_WAVES = SG_T / HHX_WAVE;

struct SgLds {
    u32 *bitmap;   // [W]
    u32 *prefix;   // [W] exclusive popcount prefix
    i64 *acc;      // [cap]
    u32 *scratch;  // [SG_T]
};

__device__ __forceinline__ SgLds carve(unsigned char *smem, i32 W, i32 cap) {
    SgLds l;
    l.acc = (i64 *)smem;                                   // 8-byte aligned first
    l.bitmap = (u32 *)(smem + (size_t)cap * 8);
    l.prefix = l.bitmap + W;
    l.scratch = l.prefix + W;
    return l;
}

// mark the columns of row `row` of A*B in the LDS bitmap; returns the wave-local product count
__device__ __forceinline__ i64 mark_row(const SgLds &l, i32 a_b, i32 a_e, const i32 *__restrict__ Aj,
                                        const i32 *__restrict__ Bp, const i32 *__restrict__ Bj) {
    const int lane = lane_id(), wave = threadIdx.x / HHX_WAVE;
    i64 prods = 0;
    for (i32 ai = a_b + wave; ai < a_e; ai += SG_WAVES) {
        const i32 k = Aj[ai];
        const i32 qb = Bp[k], qe = Bp[k + 1];
        prods += qe - qb;
        for (i32 q = qb + lane; q < qe; q += HHX_WAVE) {
            const i32 c = Bj[q];
            atomicOr(&l.bitmap[c >> 5], 1u << (c & 31));
        }
    }
    return prods;
}

// block-wide: prefix[w] = number of set bits in bitmap[0..w)
__device__ __forceinline__ void bitmap_prefix(const SgLds &l, i32 W) {
    const int tid = threadIdx.x;
    const i32 per = (W + SG_T - 1) / SG_T;
    const i32 w0 = tid * per, w1 = min(W, w0 + per);
    u32 local = 0;
    for (i32 w = w0; w < w1; ++w) local += __popc(l.bitmap[w]);
    l.scratch[tid] = local;
    __syncthreads();
    // 256-entry exclusive scan by the first wave (4 entries per lane)
    if (tid < HHX_WAVE) {
        u32 v0 = l.scratch[tid * 4], v1 = l.scratch[tid * 4 + 1], v2 = l.scratch[tid * 4 + 2], v3 = l.scratch[tid * 4 + 3];
        u32 s = v0 + v1 + v2 + v3, incl = s;
#pragma unroll
        for (int o = 1; o < HHX_WAVE; o <<= 1) {
            u32 t = __shfl_up(incl, o, HHX_WAVE);
            if (tid >= o) incl += t;
        }
        u32 ex = incl - s;
        l.scratch[tid * 4] = ex;
        l.scratch[tid * 4 + 1] = ex + v0;
        l.scratch[tid * 4 + 2] = ex + v0 + v1;
        l.scratch[tid * 4 + 3] = ex + v0 + v1 + v2;
    }
    __syncthreads();
    u32 run = l.scratch[tid];
    for (i32 w = w0; w < w1; ++w) {
        l.prefix[w] = run;
        run += __popc(l.bitmap[w]);
    }
    __syncthreads();
}

// ---- symbolic: nnz of every output row (+ total product count) -------------------------------
__global__ __launch_bounds__(SG_T) void k_spgemm_symbolic(i32 n_rows, const i32 *__restrict__ Ap,
                                                          const i32 *__restrict__ Aj, const i32 *__restrict__ Bp,
                                                          const i32 *__restrict__ Bj, i32 W, i32 cap,
                                                          i32 *__restrict__ row_nnz, u64 *__restrict__ n_products) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SgLds l = carve(smem, W, cap);
    const int tid = threadIdx.x;
    i64 prods = 0;
    for (i32 row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const i32 a_b = Ap[row], a_e = Ap[row + 1];
        if (a_e - a_b <= 1) {                       // empty, or a scaled copy of one row of B
            if (tid == 0) {
                i32 len = 0;
                if (a_e > a_b) { const i32 k = Aj[a_b]; len = Bp[k + 1] - Bp[k]; }
                row_nnz[row] = len;
                prods += len;
            }
            continue;
        }
        for (i32 w = tid; w < W; w += SG_T) l.bitmap[w] = 0;
        __syncthreads();
        const i64 f = mark_row(l, a_b, a_e, Aj, Bp, Bj);
        if (lane_id() == 0) prods += f;
        __syncthreads();
        u32 local = 0;
        for (i32 w = tid; w < W; w += SG_T) local += __popc(l.bitmap[w]);
        i32 s = wave_sum_i32((i32)local);
        if (lane_id() == 0) l.scratch[tid / HHX_WAVE] = (u32)s;
        __syncthreads();
        if (tid == 0) row_nnz[row] = (i32)(l.scratch[0] + l.scratch[1] + l.scratch[2] + l.scratch[3]);
        __syncthreads();
    }
    if (n_products && (lane_id() == 0 || tid == 0) && prods) atomicAdd((unsigned long long *)n_products, (unsigned long long)prods);
}

// ---- numeric ---------------------------------------------------------------------------------
__device__ __forceinline__ i64 to_fixed(float a, float b, double scale) {
    return __double2ll_rn((double)a * (double)b * scale);   // exact product, exact power-of-two scaling, round to nearest even
}

__global__ __launch_bounds__(SG_T) void k_spgemm_numeric(i32 n_rows, const i32 *__restrict__ Ap,
                                                         const i32 *__restrict__ Aj, const float *__restrict__ Ax,
                                                         const i32 *__restrict__ Bp, const i32 *__restrict__ Bj,
                                                         const float *__restrict__ Bx, const i32 *__restrict__ Cp,
                                                         i32 *__restrict__ Cj, float *__restrict__ Cx, i32 W, i32 cap,
                                                         double scale, double inv_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SgLds l = carve(smem, W, cap);
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / HHX_WAVE;
    for (i32 row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const i32 a_b = Ap[row], a_e = Ap[row + 1];
        const i32 base = Cp[row];
        const i32 nnz_row = Cp[row + 1] - base;
        if (a_e == a_b) continue;
        if (a_e - a_b == 1) {                       // C[row,:] = a * B[k,:], same fixed-point rounding
            const i32 k = Aj[a_b];
            const float a = Ax[a_b];
            const i32 qb = Bp[k];
            for (i32 t = tid; t < nnz_row; t += SG_T) {
                Cj[base + t] = Bj[qb + t];
                Cx[base + t] = (float)((double)to_fixed(a, Bx[qb + t], scale) * inv_scale);
            }
            continue;
        }
        for (i32 w = tid; w < W; w += SG_T) l.bitmap[w] = 0;
        __syncthreads();
        (void)mark_row(l, a_b, a_e, Aj, Bp, Bj);
        __syncthreads();
        bitmap_prefix(l, W);
        // sorted column indices straight from the bitmap
        for (i32 w = tid; w < W; w += SG_T) {
            u32 bits = l.bitmap[w];
            i32 r = base + (i32)l.prefix[w];
            while (bits) {
                const int b = __ffs(bits) - 1;
                Cj[r++] = (w << 5) + b;
                bits &= bits - 1;
            }
        }
        // accumulate, one window of `cap` ranks at a time
        for (i32 win0 = 0; win0 < nnz_row; win0 += cap) {
            const i32 wlen = min(cap, nnz_row - win0);
            for (i32 t = tid; t < wlen; t += SG_T) l.acc[t] = 0;
            __syncthreads();
            for (i32 ai = a_b + wave; ai < a_e; ai += SG_WAVES) {
                const i32 k = Aj[ai];
                const float a = Ax[ai];
                const i32 qb = Bp[k], qe = Bp[k + 1];
                for (i32 q = qb + lane; q < qe; q += HHX_WAVE) {
                    const i32 c = Bj[q];
                    const u32 word = l.bitmap[c >> 5];
                    const i32 r = (i32)(l.prefix[c >> 5] + __popc(word & ((1u << (c & 31)) - 1u))) - win0;
                    if ((u32)r < (u32)wlen)
                        atomicAdd((unsigned long long *)&l.acc[r], (unsigned long long)to_fixed(a, Bx[q], scale));
                }
            }
            __syncthreads();
            for (i32 t = tid; t < wlen; t += SG_T) Cx[base + win0 + t] = (float)((double)l.acc[t] * inv_scale);
            __syncthreads();
        }
    }
}

// ---- fixed-point scale: bound |C[i,j]| <= max_i sum_k |A[i,k]| * max|B| ------------------------
__global__ __launch_bounds__(256) void k_row_abs_sum_max(i32 n_rows, const i32 *__restrict__ indptr,
                                                         const float *__restrict__ data, u32 *__restrict__ out_bits) {
    const int lane = lane_id();
    float best = 0.f;
    for (i32 row = blockIdx.x * 4 + threadIdx.x / HHX_WAVE; row < n_rows; row += gridDim.x * 4) {
        double s = 0.0;
        for (i32 p = indptr[row] + lane; p < indptr[row + 1]; p += HHX_WAVE) s += fabs((double)data[p]);
        s = wave_sum_f64(s);
        best = fmaxf(best, (float)(s * (1.0 + 1e-6)));      // round up a little: this is only a bound
    }
    if (lane == 0 && best > 0.f) atomicMax(out_bits, __float_as_uint(best));
}

__global__ __launch_bounds__(256) void k_abs_max(i64 nnz, const float *__restrict__ data, u32 *__restrict__ out_bits) {
    float best = 0.f;
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (i64)gridDim.x * blockDim.x)
        best = fmaxf(best, fabsf(data[p]));
    best = wave_max_f32(best);
    if (lane_id() == 0 && best > 0.f) atomicMax(out_bits, __float_as_uint(best));
}

// [0] = 1 if some entry is negative or above 1 (or NaN)
__global__ __launch_bounds__(256) void k_out_of_unit(i64 nnz, const float *__restrict__ data, unsigned int *flag) {
    bool bad = false;
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (i64)gridDim.x * blockDim.x) bad |= !(data[p] >= 0.0f && data[p] <= 1.0f);
    if (__any(bad) && lane_id() == 0) atomicExch(flag, 1u);
}

// stochastic-like operands: entries in [0,1] and every row sum of A <= 1 (+ rounding slack) -> |C| <= 1
int is_stochastic_like(const hhx_csr *a, const hhx_csr *b, bool *yes) {
    DevBuf<u32> bits;
    DevBuf<unsigned int> flag;
    if (bits.alloc(1) || flag.alloc(1)) return 1;
    HHX_HIP(hipMemsetAsync(bits.p, 0, sizeof(u32), g_stream));
    HHX_HIP(hipMemsetAsync(flag.p, 0, sizeof(unsigned int), g_stream));
    k_row_abs_sum_max<<<(unsigned)std::min<i64>(((i64)a->n_rows + 3) / 4 + 1, 8192), 256, 0, g_stream>>>(a->n_rows, a->indptr.p, a->data.p, bits.p);
    if (a->nnz) k_out_of_unit<<<(unsigned)std::min<i64>((a->nnz + 255) / 256 + 1, 4096), 256, 0, g_stream>>>(a->nnz, a->data.p, flag.p);
    if (b->nnz) k_out_of_unit<<<(unsigned)std::min<i64>((b->nnz + 255) / 256 + 1, 4096), 256, 0, g_stream>>>(b->nnz, b->data.p, flag.p);
    HHX_LAUNCH_CHECK();
    u32 hb = 0; unsigned int hf = 0;
    HHX_HIP(hipMemcpyAsync(&hb, bits.p, sizeof hb, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipMemcpyAsync(&hf, flag.p, sizeof hf, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    float rs;
    memcpy(&rs, &hb, 4);
    *yes = !hf && rs <= 1.0001f;
    return 0;
}

int choose_shift(const hhx_csr *a, const hhx_csr *b, int *shift) {
    DevBuf<u32> bits;
    if (bits.alloc(2)) return 1;
    HHX_HIP(hipMemsetAsync(bits.p, 0, 2 * sizeof(u32), g_stream));
    unsigned ga = (unsigned)std::min<i64>(((i64)a->n_rows + 3) / 4 + 1, 8192);
    k_row_abs_sum_max<<<ga, 256, 0, g_stream>>>(a->n_rows, a->indptr.p, a->data.p, bits.p);
    HHX_LAUNCH_CHECK();
    unsigned gb = (unsigned)std::min<i64>((b->nnz + 255) / 256 + 1, 4096);
    k_abs_max<<<gb, 256, 0, g_stream>>>(b->nnz, b->data.p, bits.p + 1);
    HHX_LAUNCH_CHECK();
    u32 h[2];
    HHX_HIP(hipMemcpyAsync(h, bits.p, sizeof h, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    float fa, fb;
    memcpy(&fa, &h[0], 4);
    memcpy(&fb, &h[1], 4);
    double bound = (double)fa * (double)fb;
    if (!(bound > 0)) { *shift = 62; return 0; }
    int e;
    frexp(bound, &e);                 // bound < 2^e
    int s = 61 - e;                   // |sum| * 2^s < 2^61: headroom for truncation noise, sign bit free
    if (s > 1000) s = 1000;
    if (s < -1000) s = -1000;
    *shift = s;
    return 0;
}

}  // namespace

extern "C" int hhx_spgemm_ex(const hhx_csr *a, const hhx_csr *b, int fx_shift, hhx_csr **out, i64 *n_products) {
    if (!a || !b || !out) return fail("null pointer");
    if (a->n_cols != b->n_rows) return fail("spgemm shape mismatch: %d x %d times %d x %d", a->n_rows, a->n_cols, b->n_rows, b->n_cols);
    const i32 n_rows = a->n_rows, n_cols = b->n_cols;
    // Fast path (every product on the MCL path): stochastic-like operands and a shift the exact-double accumulation
    // can hold -> the fused window / bitmap kernels of hhx_expand.hip in plain-product mode (20x the throughput of the
    // generic kernels below at n = 10k-30k).  Same specification, same bits.
    if (fx_shift <= 52 && !getenv("HHX_SPGEMM_GENERIC")) {           // fx_shift < 0 (automatic): 52 if the operands qualify
        bool ok = false;
        HHX_TRY(is_stochastic_like(a, b, &ok));
        if (ok) return hhx_expand_raw(a, b, fx_shift < 0 ? 52 : fx_shift, out, n_products);
    }
    int shift = fx_shift;
    if (shift < 0 || shift > 1000) {
        // stochastic operands (every call on the MCL path): ||A||_inf = 1, max|B| <= 1 -> shift 60/61;
        // computed rather than assumed so that the S1 seam stays a general float32 SpGEMM.
        HHX_TRY(choose_shift(a, b, &shift));
    }
    const i32 W = (n_cols + 31) / 32;
    // LDS budget: bitmap + prefix (8 B per 32 columns) + scratch + accumulators.  Aim for 64 KB per
    // workgroup (2 per CU); grow to the full 160 KB for very wide matrices.
    const size_t fixed = (size_t)W * 8 + SG_T * 4;
    size_t budget = 64 * 1024;
    if (fixed + 2048 * 8 > budget) budget = 160 * 1024;
    if (fixed + 1024 * 8 > budget)
        return fail("spgemm: %d columns exceed the LDS bitmap capacity of this build (max ~4.7M)", n_cols);
    i32 cap = (i32)((budget - fixed) / 8);
    cap &= ~63;
    size_t lds = (size_t)cap * 8 + fixed;
    static int attr_dev = -1;           // the attribute is per device: keyed on the current ordinal
    int dev = 0;
    HHX_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_spgemm_symbolic, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HHX_HIP(hipFuncSetAttribute((const void *)k_spgemm_numeric, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_dev = dev;
    }
    DevBuf<i32> row_nnz;
    DevBuf<u64> prods;
    if (row_nnz.alloc((size_t)n_rows + 1) || prods.alloc(1)) return 1;
    HHX_HIP(hipMemsetAsync(prods.p, 0, sizeof(u64), g_stream));
    const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>(n_rows, 256 * 8));
    { KTimer kt("spgemm_symbolic");
    k_spgemm_symbolic<<<grid, SG_T, fixed, g_stream>>>(n_rows, a->indptr.p, a->indices.p, b->indptr.p, b->indices.p, W, 0,
                                                     row_nnz.p, prods.p); }
    HHX_LAUNCH_CHECK();
    DevBuf<i32> cp;
    if (cp.alloc((size_t)n_rows + 1)) return 1;
    i64 total = 0;
    HHX_TRY(exclusive_scan_i32(row_nnz.p, cp.p, n_rows, &total));
    if (n_products) {
        u64 f = 0;
        HHX_HIP(hipMemcpyAsync(&f, prods.p, sizeof f, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        *n_products = (i64)f;
    }
    // Wide output rows (the un-pruned pre-expansion is nearly dense): if the average row would need
    // several 64 KB accumulator windows, give the numeric kernel the whole 160 KB LDS (one workgroup per
    // CU, up to ~19k accumulators) so that the products are traversed once instead of once per window.
    if (n_rows > 0 && total / n_rows > cap / 2 && budget < 160 * 1024) {
        const i32 big = (i32)((160 * 1024 - fixed) / 8) & ~63;
        if (big > cap) { cap = big; lds = (size_t)cap * 8 + fixed; }
    }
    hhx_csr *c = nullptr;
    HHX_TRY(hhx_csr_alloc_internal(n_rows, n_cols, total, &c));
    HHX_HIP(hipMemcpyAsync(c->indptr.p, cp.p, sizeof(i32) * ((size_t)n_rows + 1), hipMemcpyDeviceToDevice, g_stream));
    { KTimer kt("spgemm_numeric");
    k_spgemm_numeric<<<grid, SG_T, lds, g_stream>>>(n_rows, a->indptr.p, a->indices.p, a->data.p, b->indptr.p, b->indices.p,
                                                    b->data.p, c->indptr.p, c->indices.p, c->data.p, W, cap,
                                                    ldexp(1.0, shift), ldexp(1.0, -shift)); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hhx_csr_free(c); return fail("spgemm numeric launch: %s", hipGetErrorString(e)); }
    *out = c;
    return 0;
}

extern "C" int hhx_spgemm(const hhx_csr *a, const hhx_csr *b, hhx_csr **out) { return hhx_spgemm_ex(a, b, -1, out, nullptr); }
