// Row-local MCL kernels on CSR(T): L1 normalise, inflate, prune, convergence, attractor read-out,
// and the mcl() driver.  Reference: scripts/HapHiC_cluster.py:1987-2095 (prune, mcl, interpret_result)
// and sklearn's _inplace_csr_row_normalize_l1 (double row sum, x = float(x / sum)).
//
// All of these are HBM-streaming kernels: one 64-lane wavefront owns one row, lanes stride the row
// so that a wave reads 256 contiguous bytes of indices / values per instruction; row sums are
// wave-level double reductions in a fixed tree (deterministic for a given row).
#include "hhx_common.h"

using namespace hhx;

int hhx_csr_alloc_internal(i32 n_rows, i32 n_cols, i64 nnz, hhx_csr **out);
int hhx_expand_class_stream(const hhx_csr *a, const hhx_csr *b, const hhx_links_operand *lk, int fx_shift,
                     double inflation, double pruning, hhx_csr **out, i64 *n_products, i64 *nnz_expanded);
int hhx_expand_dense_impl(const hhx_csr *a, const hhx_csr *b, const hhx_links_operand *lk, int fx_shift, hhx_dense **out, i64 *n_products,
                          i64 *nnz_expanded);
void hhx_expand_set_hint(i64 out, i64 cand);            // hhx_expand.hip: the pools of the next fused iteration sized from the one before
void hhx_expand_last_demand(i64 *out, i64 *cand);
int hhx_dense_layout(i32 n_rows, i32 n_cols, i64 nnz_b);
namespace hhx { i64 pool_cached_bytes(); }

namespace {

// Stochastic operands: every row of T sums to 1 and every entry is <= 1, so |C| <= 1 and products rounded on
// the 2^-52 grid add exactly in a double (hhx_expand.hip: acc_add); no per-call bound reduction is needed.
constexpr int HHX_MCL_FX_SHIFT = 52;

constexpr int ROW_T = 256;                       // 4 waves per workgroup, one row per wave
constexpr int ROW_WAVES = ROW_T / HHX_WAVE;

inline unsigned row_grid(i32 n_rows) {
    i64 blocks = ((i64)n_rows + ROW_WAVES - 1) / ROW_WAVES;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;    // grid-stride beyond 32 workgroups per CU
    return (unsigned)blocks;
}

__device__ __forceinline__ float inflate_one(float x, double r, bool square) {
    // numpy float32 `data ** r`: r == 2 -> x*x; otherwise powf with the exponent rounded to float32.
    // hhx_powr: exp2(r log2 x) in double, one rounding to float32 (hhx_common.h)
    return square ? x * x : hhx_powr(x, r);
}

// ---- L1 normalise in place ------------------------------------------------------------------
__global__ __launch_bounds__(ROW_T) void k_normalize_l1(i32 n_rows, const i32 *__restrict__ indptr,
                                                        float *__restrict__ data, double *__restrict__ row_sum) {
    const int lane = lane_id();
    for (i32 row = blockIdx.x * ROW_WAVES + threadIdx.x / HHX_WAVE; row < n_rows; row += gridDim.x * ROW_WAVES) {
        const i32 b = indptr[row], e = indptr[row + 1];
        double s = 0.0;
        for (i32 p = b + lane; p < e; p += HHX_WAVE) s += fabs((double)data[p]);
        s = wave_sum_f64(s);
        if (row_sum && lane == 0) row_sum[row] = s;
        if (s == 0.0) continue;
        for (i32 p = b + lane; p < e; p += HHX_WAVE) data[p] = (float)((double)data[p] / s);
    }
}

// ---- inflate (+normalise) and prune statistics ---------------------------------------------
// MODE 0: data already inflated+normalised (stand-alone prune()).
// MODE 1: data = power(data, r); normalise; then prune statistics (mcl() steps 3+4 fused).
// MODE 2: inflate + normalise only (no prune statistics).
// Per row: cnt = number of survivors (entries >= thr plus the first row maximum), amax = position
// of that maximum, s2 = double sum of the survivors (second normalisation, :2014).
template <int MODE>
__global__ __launch_bounds__(ROW_T) void k_inflate_stats(i32 n_rows, const i32 *__restrict__ indptr,
                                                         float *__restrict__ data, double r, int square,
                                                         float thr, i32 *__restrict__ cnt,
                                                         i32 *__restrict__ amax, double *__restrict__ s2) {
    const int lane = lane_id();
    for (i32 row = blockIdx.x * ROW_WAVES + threadIdx.x / HHX_WAVE; row < n_rows; row += gridDim.x * ROW_WAVES) {
        const i32 b = indptr[row], e = indptr[row + 1];
        double s1 = 1.0;
        if (MODE != 0) {
            double s = 0.0;
            for (i32 p = b + lane; p < e; p += HHX_WAVE) {
                float v = inflate_one(data[p], r, square);
                data[p] = v;
                s += fabs((double)v);
            }
            s1 = wave_sum_f64(s);
        }
        // second sweep: normalised value q, survivors, first maximum
        float best = -1.0f;
        i32 best_p = 0x7fffffff;
        i32 keep = 0;
        double ssum = 0.0;
        for (i32 p = b + lane; p < e; p += HHX_WAVE) {
            float q = data[p];
            if (MODE != 0 && s1 != 0.0) {
                q = (float)((double)q / s1);
                data[p] = q;
            }
            if (MODE != 2) {
                if (q > best) { best = q; best_p = p; }     // strict >: keeps the lowest index per lane
                if (q >= thr) { ++keep; ssum += fabs((double)q); }
            }
        }
        if (MODE == 2) continue;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_down(best, o, HHX_WAVE);
            i32 op = __shfl_down(best_p, o, HHX_WAVE);
            if (ob > best || (ob == best && op < best_p)) { best = ob; best_p = op; }
        }
        best = __shfl(best, 0, HHX_WAVE);
        best_p = __shfl(best_p, 0, HHX_WAVE);
        keep = wave_sum_i32(keep);
        ssum = wave_sum_f64(ssum);
        if (lane == 0) {
            if (e > b && !(best >= thr)) { ++keep; ssum += fabs((double)best); }   // restored maximum, :2010-2013
            cnt[row] = keep;
            amax[row] = (e > b) ? best_p : -1;
            s2[row] = ssum;
        }
    }
}

// ---- prune write: ordered in-wave compaction + second normalisation -------------------------
__global__ __launch_bounds__(ROW_T) void k_prune_write(i32 n_rows, const i32 *__restrict__ indptr,
                                                       const i32 *__restrict__ indices,
                                                       const float *__restrict__ data, float thr,
                                                       const i32 *__restrict__ amax,
                                                       const double *__restrict__ s2,
                                                       const i32 *__restrict__ out_indptr,
                                                       i32 *__restrict__ out_indices,
                                                       float *__restrict__ out_data) {
    const int lane = lane_id();
    for (i32 row = blockIdx.x * ROW_WAVES + threadIdx.x / HHX_WAVE; row < n_rows; row += gridDim.x * ROW_WAVES) {
        const i32 b = indptr[row], e = indptr[row + 1];
        const i32 am = amax[row];
        const double s = s2[row];
        i32 o = out_indptr[row];
        for (i32 p0 = b; p0 < e; p0 += HHX_WAVE) {
            const i32 p = p0 + lane;
            float q = 0.0f;
            i32 c = 0;
            bool k = false;
            if (p < e) {
                q = data[p];
                c = indices[p];
                k = (q >= thr) || (p == am);
            }
            const u64 mask = __ballot(k);
            if (k) {
                const i32 pos = o + __popcll(mask & ((1ull << lane) - 1ull));
                out_indices[pos] = c;
                out_data[pos] = (s != 0.0) ? (float)((double)q / s) : q;
            }
            o += __popcll(mask);
        }
    }
}

// ---- convergence statistic -------------------------------------------------------------------
__device__ __forceinline__ i32 find_col(const i32 *__restrict__ idx, i32 b, i32 e, i32 c) {
    while (b < e) {
        i32 m = b + ((e - b) >> 1);                     // b + e overflows int32 beyond 2^30 entries
        i32 v = idx[m];
        if (v < c) b = m + 1;
        else e = m;
    }
    return b;
}

__global__ __launch_bounds__(ROW_T) void k_convergence(i32 n_rows, const i32 *__restrict__ ap,
                                                       const i32 *__restrict__ aj, const float *__restrict__ ax,
                                                       const i32 *__restrict__ bp, const i32 *__restrict__ bj,
                                                       const float *__restrict__ bx, u32 *__restrict__ out_bits) {
    const int lane = lane_id();
    const float rtol = (float)1e-5;
    float best = 0.0f;
    for (i32 row = blockIdx.x * ROW_WAVES + threadIdx.x / HHX_WAVE; row < n_rows; row += gridDim.x * ROW_WAVES) {
        const i32 ab = ap[row], ae = ap[row + 1], bb = bp[row], be = bp[row + 1];
        for (i32 p = ab + lane; p < ae; p += HHX_WAVE) {       // entries of M (matched or M-only)
            const i32 c = aj[p];
            const i32 q = find_col(bj, bb, be, c);
            const float l = (q < be && bj[q] == c) ? bx[q] : 0.0f;
            const float d = fabsf(ax[p] - l) - rtol * fabsf(l);
            best = fmaxf(best, d);
        }
        for (i32 q = bb + lane; q < be; q += HHX_WAVE) {       // entries only in `last`
            const i32 c = bj[q];
            const i32 p = find_col(aj, ab, ae, c);
            if (!(p < ae && aj[p] == c)) {
                const float l = bx[q];
                const float d = fabsf(0.0f - l) - rtol * fabsf(l);
                best = fmaxf(best, d);
            }
        }
    }
    best = wave_max_f32(best);
    if (lane == 0 && best > 0.0f) atomicMax(out_bits, __float_as_uint(best));   // non-negative floats order as uints
}

// 16-bit link counts of a raw link matrix (:362-368: integer counts as float32, unit self loops); flags[0] is
// raised if some entry is not an integer in [0, 65535] (e.g. after --normalize_by_nlinks)
__global__ __launch_bounds__(256) void k_link_counts(i64 nnz, const float *__restrict__ data, unsigned short *__restrict__ n16,
                                                     unsigned int *flags) {
    bool bad = false;
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (i64)gridDim.x * blockDim.x) {
        const float v = data[p];
        const unsigned int c = (unsigned int)v;
        if (!(v >= 0.0f && v <= 65535.0f) || (float)c != v) bad = true;
        n16[p] = (unsigned short)c;
    }
    if (__any(bad) && lane_id() == 0) atomicExch(flags, 1u);
}
// Symmetry of the link matrix (dict_to_matrix mirrors every key, :350-356; a caller's own matrix need not be): the multiset of
// (row, column, count) must equal the multiset of (column, row, count).  Two independent 64-bit mixes of every entry are summed in
// both orientations; equal sums in both mixes <=> symmetric, up to a 2^-128 collision.  One streaming pass.
__device__ __forceinline__ u64 mix64(u64 x, u64 salt) {
    x += salt;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_symmetry_sums(i32 n_rows, const i32 *__restrict__ indptr, const i32 *__restrict__ indices,
                                                       const unsigned short *__restrict__ n16, unsigned long long *__restrict__ sums) {
    const int lane = lane_id();
    u64 s0 = 0, s1 = 0, t0 = 0, t1 = 0;
    for (i32 row = blockIdx.x * 4 + threadIdx.x / HHX_WAVE; row < n_rows; row += gridDim.x * 4)
        for (i32 p = indptr[row] + lane; p < indptr[row + 1]; p += HHX_WAVE) {
            const u64 j = (u64)(u32)indices[p], c = (u64)n16[p] << 48;
            const u64 fwd = ((u64)(u32)row << 24 | j) ^ c, rev = (j << 24 | (u64)(u32)row) ^ c;     // n < 2^24 (checked by the caller)
            s0 += mix64(fwd, 0x9e3779b97f4a7c15ull); s1 += mix64(fwd, 0xd1b54a32d192ed03ull);
            t0 += mix64(rev, 0x9e3779b97f4a7c15ull); t1 += mix64(rev, 0xd1b54a32d192ed03ull);
        }
    s0 = (u64)wave_sum_i64((i64)s0); s1 = (u64)wave_sum_i64((i64)s1); t0 = (u64)wave_sum_i64((i64)t0); t1 = (u64)wave_sum_i64((i64)t1);
    if (lane == 0) { atomicAdd(&sums[0], s0); atomicAdd(&sums[1], s1); atomicAdd(&sums[2], t0); atomicAdd(&sums[3], t1); }
}
// W_k = rint(2^shift / d_k) and the largest row sum (as the bits of a non-negative double: ordered like integers)
__global__ __launch_bounds__(256) void k_max_row_sum(i32 n, const double *__restrict__ d, unsigned long long *__restrict__ max_bits) {
    unsigned long long m = 0;
    for (i32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, (unsigned long long)__double_as_longlong(d[i]));
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned long long)__shfl_down((long long)m, o, HHX_WAVE));
    if (lane_id() == 0) atomicMax(max_bits, m);
}
__global__ __launch_bounds__(256) void k_fx_weights(i32 n, const double *__restrict__ d, double two_s, u64 *__restrict__ W) {
    for (i32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) W[i] = (u64)rint(two_s / d[i]);
}
__global__ __launch_bounds__(256) void k_any_zero(i32 n, const double *__restrict__ v, unsigned int *flags) {
    bool bad = false;
    for (i32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bad |= !(v[i] > 0.0);
    if (__any(bad) && lane_id() == 0) atomicExch(flags, 1u);
}

// keep != 0: c->data is left untouched (the inflated values go to a scratch copy): a row block of M^e that the
// inflation sweep revisits for every inflation
int inflate_prune_impl(hhx_csr *c, int mode, double inflation, double pruning, hhx_csr **out, int keep = 0) {
    const i32 n = c->n_rows;
    DevBuf<float> scratch;
    float *cdata = c->data.p;
    if (keep && mode == 1) {
        if (scratch.alloc((size_t)c->nnz + 1)) return 1;
        if (c->nnz) HHX_HIP(hipMemcpyAsync(scratch.p, c->data.p, sizeof(float) * (size_t)c->nnz, hipMemcpyDeviceToDevice, g_stream));
        cdata = scratch.p;
    }
    DevBuf<i32> cnt, amax, optr;
    DevBuf<double> s2;
    if (cnt.alloc((size_t)n + 1) || amax.alloc((size_t)n + 1) || s2.alloc((size_t)n + 1) || optr.alloc((size_t)n + 1)) return 1;
    const float thr = (float)pruning;
    const double r = (double)(float)inflation;
    const int square = (inflation == 2.0);
    { KTimer kt("inflate_stats");
    if (mode == 0)
        k_inflate_stats<0><<<row_grid(n), ROW_T, 0, g_stream>>>(n, c->indptr.p, cdata, r, square, thr, cnt.p, amax.p, s2.p);
    else
        k_inflate_stats<1><<<row_grid(n), ROW_T, 0, g_stream>>>(n, c->indptr.p, cdata, r, square, thr, cnt.p, amax.p, s2.p);
    }
    HHX_LAUNCH_CHECK();
    i64 total = 0;
    HHX_TRY(exclusive_scan_i32(cnt.p, optr.p, n, &total));
    hhx_csr *p = nullptr;
    HHX_TRY(hhx_csr_alloc_internal(n, c->n_cols, total, &p));
    HHX_HIP(hipMemcpyAsync(p->indptr.p, optr.p, sizeof(i32) * ((size_t)n + 1), hipMemcpyDeviceToDevice, g_stream));
    { KTimer kt("prune_write");
    k_prune_write<<<row_grid(n), ROW_T, 0, g_stream>>>(n, c->indptr.p, c->indices.p, cdata, thr, amax.p, s2.p,
                                                       p->indptr.p, p->indices.p, p->data.p); }
    HHX_LAUNCH_CHECK();
    if (scratch.p) HHX_HIP(hipStreamSynchronize(g_stream));       // the scratch copy dies with this frame
    *out = p;
    return 0;
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" int hhx_normalize_l1(hhx_csr *m) {
    if (!m) return fail("null matrix");
    k_normalize_l1<<<row_grid(m->n_rows), ROW_T, 0, g_stream>>>(m->n_rows, m->indptr.p, m->data.p, nullptr);
    HHX_LAUNCH_CHECK();
    return 0;
}

extern "C" int hhx_inflate(hhx_csr *m, double inflation) {
    if (!m) return fail("null matrix");
    if (!(inflation > 0)) return fail("inflation must be positive");
    k_inflate_stats<2><<<row_grid(m->n_rows), ROW_T, 0, g_stream>>>(m->n_rows, m->indptr.p, m->data.p,
                                                                   (double)(float)inflation, inflation == 2.0, 0.0f,
                                                                   nullptr, nullptr, nullptr);
    HHX_LAUNCH_CHECK();
    return 0;
}

extern "C" int hhx_prune(const hhx_csr *m, double pruning, hhx_csr **out) {
    if (!m || !out) return fail("null pointer");
    return inflate_prune_impl(const_cast<hhx_csr *>(m), 0, 2.0, pruning, out);   // mode 0 never writes m
}

extern "C" int hhx_inflate_prune(hhx_csr *c, double inflation, double pruning, hhx_csr **out) {
    if (!c || !out) return fail("null pointer");
    if (!(inflation > 0)) return fail("inflation must be positive");
    return inflate_prune_impl(c, 1, inflation, pruning, out);
}

extern "C" int hhx_inflate_prune_keep(const hhx_csr *c, double inflation, double pruning, hhx_csr **out) {
    if (!c || !out) return fail("null pointer");
    if (!(inflation > 0)) return fail("inflation must be positive");
    return inflate_prune_impl(const_cast<hhx_csr *>(c), 1, inflation, pruning, out, 1);
}

extern "C" int hhx_convergence_stat(const hhx_csr *m, const hhx_csr *last, float *stat) {
    if (!m || !last || !stat) return fail("null pointer");
    if (m->n_rows != last->n_rows) return fail("shape mismatch");
    DevBuf<u32> bits;
    if (bits.alloc(1)) return 1;
    HHX_HIP(hipMemsetAsync(bits.p, 0, sizeof(u32), g_stream));
    { KTimer kt("convergence");
    k_convergence<<<row_grid(m->n_rows), ROW_T, 0, g_stream>>>(m->n_rows, m->indptr.p, m->indices.p, m->data.p,
                                                              last->indptr.p, last->indices.p, last->data.p, bits.p); }
    HHX_LAUNCH_CHECK();
    u32 h = 0;
    HHX_HIP(hipMemcpyAsync(&h, bits.p, sizeof(u32), hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    memcpy(stat, &h, sizeof(float));
    return 0;
}

// mcl() driver.  pre_expanded != 0: `m` is the pre-expanded matrix (the reference seam, :2026);
// pre_expanded == 0: `m` is the L1-normalised link matrix and the pre-expansion of :2146-2147 is
// fused into iteration 0.  Every expansion that feeds an inflate+prune goes through the fused kernel.
struct NormalisedLinks;
static int expand_links_iteration0(const hhx_csr *norm, const NormalisedLinks &nl, double inflation, double pruning, hhx_csr **out, i64 *n_products,
                                   i64 *nnz_expanded);
static int mcl_impl(const hhx_csr *m, int pre_expanded, int expansion, double inflation, int max_iter, double pruning,
                    hhx_csr **out, int *n_iter, int *converged, i64 *stats, const NormalisedLinks *nl = nullptr, int first_it = 0) {
    if (!m || !out || !n_iter || !converged) return fail("null pointer");
    if (m->n_rows != m->n_cols) return fail("mcl needs a square matrix");
    if (expansion < 1) return fail("expansion must be >= 1");
    if (!(inflation > 0)) return fail("inflation must be positive");
    *n_iter = first_it;
    *converged = 0;
    hhx_csr *cur = nullptr;                  // matrix at the end of the previous iteration (== last_matrix)
    int rc = 0;
    i64 demand_out = 0, demand_cand = 0;     // candidate / survivor pool demand of the previous fused iteration
    bool have_demand = false;
    for (int it = first_it; it < max_iter && !rc; ++it) {
        const hhx_csr *src = cur ? cur : m;  // operand of this iteration's expansion
        const bool expand = (it > 0 || !pre_expanded) && expansion > 1;
        i64 st_a = src->nnz, st_f = 0, st_c = 0;
        hhx_csr *p = nullptr;
        if (!expand) {
            hhx_csr *c = nullptr;            // iteration 0 of the pre-expanded seam skips the expansion, :2030
            rc = hhx_csr_copy(src, &c);
            if (!rc) { st_c = c->nnz; rc = hhx_inflate_prune(c, inflation, pruning, &p); }
            if (c) hhx_csr_free(c);
        } else {
            // mkl_matrix_power(M, e) = M * M^(e-1)  ==  T^(e-1) * T on CSR(T), :2017-2023
            const hhx_csr *run = src;
            for (int e = 2; e < expansion && !rc; ++e) {
                hhx_csr *nx = nullptr;
                i64 f = 0;
                rc = hhx_spgemm_ex(run, src, HHX_MCL_FX_SHIFT, &nx, &f);
                st_f += f;
                if (run != src) hhx_csr_free(const_cast<hhx_csr *>(run));
                run = nx;
            }
            if (!rc) {
                i64 f = 0;
                if (it == 0 && nl && run == src)   // iteration 0, expansion 2: both operands are the link matrix itself
                    rc = expand_links_iteration0(src, *nl, inflation, pruning, &p, &f, &st_c);
                else {
                    // the pools of this iteration from the demand of the one before (first iteration of a resumed loop: the operand is what an
                    // iteration left, its survivors ARE the entries of `src`; the candidates of the low inflations run to ~2.5 x the survivors)
                    if (have_demand) hhx_expand_set_hint(demand_out, demand_cand);
                    else if (first_it > 0) hhx_expand_set_hint(src->nnz, 3 * src->nnz);
                    rc = hhx_expand_inflate_prune(run, src, HHX_MCL_FX_SHIFT, inflation, pruning, &p, &f, &st_c);   // :2030-2042
                    if (!rc) { hhx_expand_last_demand(&demand_out, &demand_cand); have_demand = true; }
                }
                st_f += f;
            }
            if (run != src && run) hhx_csr_free(const_cast<hhx_csr *>(run));
        }
        if (rc) break;
        if (stats) { stats[4 * it] = st_a; stats[4 * it + 1] = st_c; stats[4 * it + 2] = p->nnz; stats[4 * it + 3] = st_f; }
        *n_iter = it + 1;
        if (it > 1) {                                               // step 5), :2044-2050
            float d = 0.f;
            rc = hhx_convergence_stat(p, cur ? cur : m, &d);          // resumed at it >= 2: `m` is what the previous iteration left
            if (!rc && d <= (float)1e-8) {
                *converged = 1;
                if (cur) hhx_csr_free(cur);
                cur = p;
                break;
            }
        }
        if (cur) hhx_csr_free(cur);
        cur = p;                                                    // last_matrix = matrix.copy(), :2057
    }
    if (rc) { if (cur) hhx_csr_free(cur); return rc; }
    if (!cur) rc = hhx_csr_copy(m, &cur);                           // max_iter == 0
    *out = cur;
    return rc;
}

extern "C" int hhx_mcl(const hhx_csr *pre, int expansion, double inflation, int max_iter, double pruning,
                       hhx_csr **out, int *n_iter, int *converged, i64 *stats) {
    return mcl_impl(pre, 1, expansion, inflation, max_iter, pruning, out, n_iter, converged, stats);
}

// mcl() :2026-2062 picked up after its first `done` iterations: `m` is the matrix those iterations left (for the
// inflation sweep: iteration 0 = inflate + prune of the blocked M^e, hhx_inflate_prune_keep per row block; for the multi-GPU
// driver: the iterations left once the matrix is small enough to be replicated).  done >= 1; the convergence test needs
// two computed iterations: it runs from iteration max(done, 2) on, at iteration `done` against `m` itself.
extern "C" int hhx_mcl_resume(const hhx_csr *m, int done, int expansion, double inflation, int max_iter, double pruning,
                              hhx_csr **out, int *n_iter, int *converged, i64 *stats) {
    if (done < 1) return fail("hhx_mcl_resume: done must be >= 1");
    return mcl_impl(m, 1, expansion, inflation, max_iter, pruning, out, n_iter, converged, stats, nullptr, done);
}

extern "C" int hhx_mcl_normalized(const hhx_csr *norm, int expansion, double inflation, int max_iter, double pruning,
                                  hhx_csr **out, int *n_iter, int *converged, i64 *stats) {
    return mcl_impl(norm, 0, expansion, inflation, max_iter, pruning, out, n_iter, converged, stats);
}

// The raw link matrix normalised (:2144) together with what the class stream needs: the L1 row sums the
// normalisation divided by and the 16-bit link counts.  *usable = 0 if some value is not an integer in [0, 65535] or
// some row sum is zero (then only the normalised matrix is meaningful).
// integer: the matrix is symmetric and its row sums stay below 2^18 — iteration 0 can run in the integer arithmetic of
// DESIGN.md 4.1 (W, shift), whose accumulators are an exactly symmetric matrix.
struct NormalisedLinks {
    hhx_csr *norm = nullptr;
    DevBuf<double> row_sum;
    DevBuf<unsigned short> n16;
    DevBuf<u64> W;
    int shift = 0;
    bool usable = false, integer = false;
    ~NormalisedLinks() { if (norm) hhx_csr_free(norm); }
    hhx_links_operand operand(i32 a_row0 = 0, i64 a_off = 0, int sym = 0) const {
        hhx_links_operand lk;
        lk.n16 = n16.p; lk.row_sum = row_sum.p; lk.W = integer ? W.p : nullptr; lk.shift = shift; lk.a_row0 = a_row0; lk.a_off = a_off;
        lk.sym = integer ? sym : 0;
        return lk;
    }
};
static int normalise_links(const hhx_csr *links, NormalisedLinks *o) {
    HHX_TRY(hhx_csr_copy(links, &o->norm));
    DevBuf<unsigned int> flags;
    if (o->row_sum.alloc((size_t)links->n_rows + 1) || o->n16.alloc((size_t)links->nnz + 1) || flags.alloc(1)) return 1;
    unsigned int bad = 1;
    HHX_HIP(hipMemsetAsync(flags.p, 0, sizeof(unsigned int), g_stream));
    k_normalize_l1<<<row_grid(links->n_rows), ROW_T, 0, g_stream>>>(links->n_rows, o->norm->indptr.p, o->norm->data.p, o->row_sum.p);
    if (links->nnz)
        k_link_counts<<<(unsigned)std::max<i64>(1, std::min<i64>((links->nnz + 255) / 256, 65536)), 256, 0, g_stream>>>(
            links->nnz, links->data.p, o->n16.p, flags.p);
    if (links->n_rows)
        k_any_zero<<<(unsigned)std::max<i64>(1, std::min<i64>(((i64)links->n_rows + 255) / 256, 4096)), 256, 0, g_stream>>>(
            links->n_rows, o->row_sum.p, flags.p);
    HHX_LAUNCH_CHECK();
    HHX_HIP(hipMemcpyAsync(&bad, flags.p, sizeof bad, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    o->usable = !bad;
    o->integer = false;
    if (!o->usable || links->n_rows != links->n_cols || links->n_rows >= (1 << 24) || tune_get("links_integer", 1) == 0) return 0;
    // the integer arithmetic applies to a symmetric matrix whose weights keep 24 bits: shift = 61 - lg, lg = ceil(log2(d_max)) <= 18
    DevBuf<unsigned long long> red;
    if (red.alloc(5) || o->W.alloc((size_t)links->n_rows + 1)) return 1;
    HHX_HIP(hipMemsetAsync(red.p, 0, 5 * sizeof(unsigned long long), g_stream));
    k_symmetry_sums<<<row_grid(links->n_rows), ROW_T, 0, g_stream>>>(links->n_rows, links->indptr.p, links->indices.p, o->n16.p, red.p);
    k_max_row_sum<<<(unsigned)std::max<i64>(1, std::min<i64>(((i64)links->n_rows + 255) / 256, 1024)), 256, 0, g_stream>>>(links->n_rows, o->row_sum.p, red.p + 4);
    HHX_LAUNCH_CHECK();
    unsigned long long h[5];
    HHX_HIP(hipMemcpyAsync(h, red.p, sizeof h, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    if (h[0] != h[2] || h[1] != h[3]) return 0;                 // not symmetric
    double d_max;
    memcpy(&d_max, &h[4], sizeof d_max);
    int lg = 0;
    while (ldexp(1.0, lg) < d_max) ++lg;
    if (61 - 2 * lg < 24) return 0;                              // a weight would keep fewer than 24 bits
    o->shift = 61 - lg;
    k_fx_weights<<<(unsigned)std::max<i64>(1, std::min<i64>(((i64)links->n_rows + 255) / 256, 1024)), 256, 0, g_stream>>>(
        links->n_rows, o->row_sum.p, ldexp(1.0, o->shift), o->W.p);
    HHX_LAUNCH_CHECK();
    o->integer = true;
    return 0;
}

// Iteration 0 of mcl() on the link matrix itself (expansion 2).  In the integer arithmetic S = L D^-1 L is exactly symmetric: when
// n^2 floats fit the device the upper block triangle of Y = float(S) is computed (60 % of the products at 5 column windows), the
// rest transposed, and the rows are finished by the dense epilogue; otherwise every row walks all its products into the fused
// epilogue.  Same bits either way.
static int expand_links_iteration0(const hhx_csr *norm, const NormalisedLinks &nl, double inflation, double pruning, hhx_csr **out, i64 *n_products,
                                   i64 *nnz_expanded) {
    if (!nl.usable) return hhx_expand_inflate_prune(norm, norm, HHX_MCL_FX_SHIFT, inflation, pruning, out, n_products, nnz_expanded);
    if (nl.integer && tune_get("links_sym", 1) != 0) {
        if (hhx_dense_layout(norm->n_rows, norm->n_cols, norm->nnz) != 0) {          // the square block or its upper block triangle fits (+ operand stream, pools)
            const hhx_links_operand lk = nl.operand(0, 0, 1);
            hhx_dense *d = nullptr;
            const int rc_d = hhx_expand_dense_impl(norm, norm, &lk, HHX_MCL_FX_SHIFT, &d, n_products, nnz_expanded);
            if (rc_d == 0) {
                const int rc = hhx_dense_inflate_prune(d, inflation, pruning, out);
                hhx_dense_free(d);
                return rc;
            }
            if (rc_d != 2) return rc_d;              // 2: the dense block could not be allocated after all — every row walks all its products
        }
    }
    const hhx_links_operand lk = nl.operand();
    return hhx_expand_class_stream(norm, norm, &lk, HHX_MCL_FX_SHIFT, inflation, pruning, out, n_products, nnz_expanded);
}

// run_mcl_clustering :2144-2158 for one inflation straight from the RAW link matrix of dict_to_matrix
// (:362-368): L1 normalisation (:2144), pre-expansion (:2146-2147) fused into iteration 0, mcl().  When the
// matrix holds integer link counts <= 65535 (always, unless --normalize_by_nlinks / GFA weights were applied)
// iteration 0 streams its right operand as the class stream (hhx_expand.hip), in integer arithmetic when it is symmetric.
extern "C" int hhx_mcl_links(const hhx_csr *links, int expansion, double inflation, int max_iter, double pruning,
                             hhx_csr **out, int *n_iter, int *converged, i64 *stats) {
    if (!links || !out) return fail("null pointer");
    NormalisedLinks nl;
    HHX_TRY(normalise_links(links, &nl));
    return mcl_impl(nl.norm, 0, expansion, inflation, max_iter, pruning, out, n_iter, converged, stats, &nl);
}

// iteration 0 for the rows [r0, r1) of the link matrix (the row block of one rank): the bits of hhx_mcl_links' first iteration
extern "C" int hhx_expand_links(const hhx_csr *links, i32 r0, i32 r1, int fx_shift, double inflation, double pruning,
                                hhx_csr **out, i64 *n_products, i64 *nnz_expanded) {
    if (!links || !out) return fail("null pointer");
    if (r0 < 0 || r1 < r0 || r1 > links->n_rows) return fail("row block [%d,%d) out of range", r0, r1);
    NormalisedLinks nl;
    HHX_TRY(normalise_links(links, &nl));
    i32 off = 0;                                  // before the row block exists: an error here must not leak it
    HHX_HIP(hipMemcpyAsync(&off, links->indptr.p + r0, sizeof off, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    hhx_csr *a = nullptr;
    HHX_TRY(hhx_csr_row_block(nl.norm, r0, r1, &a));
    int rc;
    if (!nl.usable) rc = hhx_expand_inflate_prune(a, nl.norm, fx_shift, inflation, pruning, out, n_products, nnz_expanded);
    else {
        const hhx_links_operand lk = nl.operand(r0, off);
        rc = hhx_expand_class_stream(a, nl.norm, &lk, fx_shift, inflation, pruning, out, n_products, nnz_expanded);
    }
    hhx_csr_free(a);
    return rc;
}

// run_mcl_clustering :2144-2147 for rows [r0, r1) of the link matrix, kept for the whole inflation sweep: L1 normalisation, then the
// rows of M^2 = T[r0:r1, :] * T as a dense float32 block (hhx_dense).  hhx_dense_inflate_prune then gives iteration 0 of mcl()
// (:2037-2042) of these rows at any inflation without walking the products again.
extern "C" int hhx_expand_links_dense(const hhx_csr *links, i32 r0, i32 r1, int fx_shift, int upper_only, hhx_dense **out, i64 *n_products,
                                      i64 *nnz_expanded) {
    if (!links || !out) return fail("null pointer");
    if (links->n_rows != links->n_cols) return fail("hhx_expand_links_dense needs the square link matrix");
    if (r0 < 0 || r1 < r0 || r1 > links->n_rows) return fail("row block [%d,%d) out of range", r0, r1);
    NormalisedLinks nl;
    HHX_TRY(normalise_links(links, &nl));
    i32 off = 0;                                  // before the row block exists: an error here must not leak it
    HHX_HIP(hipMemcpyAsync(&off, links->indptr.p + r0, sizeof off, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    hhx_csr *a = nullptr;
    HHX_TRY(hhx_csr_row_block(nl.norm, r0, r1, &a));
    const bool whole = r0 == 0 && r1 == links->n_rows;
    if (upper_only && !nl.integer) { hhx_csr_free(a); return fail("hhx_expand_links_dense: upper_only needs the integer arithmetic (symmetric counts, row sums < 2^18)"); }
    // all rows: the symmetric half + transposition; upper_only on a row block: the half alone, the caller mirrors (multi-GPU)
    const hhx_links_operand lk = nl.operand(r0, off, (upper_only || (whole && tune_get("links_sym", 1) != 0)) ? 1 : 0);
    const int rc = hhx_expand_dense_impl(a, nl.norm, nl.usable ? &lk : nullptr, fx_shift, out, n_products, nnz_expanded);
    hhx_csr_free(a);
    return rc;
}

// can iteration 0 on this link matrix run in the integer arithmetic (symmetric counts, row sums below 2^18)?  *shift = s of DESIGN 4.1
extern "C" int hhx_links_integer_ok(const hhx_csr *links, int *ok, int *shift) {
    if (!links || !ok) return fail("null pointer");
    NormalisedLinks nl;
    HHX_TRY(normalise_links(links, &nl));
    *ok = nl.integer ? 1 : 0;
    if (shift) *shift = nl.integer ? nl.shift : -1;
    return 0;
}

// what hhx_mcl_links' iteration 0 will do with this matrix on this device right now: *integer = 1 the integer arithmetic (symmetric
// counts, row sums below 2^18); *layout = 1 the symmetric half into the square dense block (+ transposition), 2 into the upper
// block triangle alone, 0 every row walks all its products into the fused epilogue (no symmetry, or neither block fits)
extern "C" int hhx_links_plan(const hhx_csr *links, int *integer, int *layout) {
    if (!links || !integer || !layout) return fail("null pointer");
    NormalisedLinks nl;
    HHX_TRY(normalise_links(links, &nl));
    *integer = nl.integer ? 1 : 0;
    *layout = (nl.usable && nl.integer && tune_get("links_sym", 1) != 0) ? hhx_dense_layout(links->n_rows, links->n_cols, links->nnz) : 0;
    return 0;
}

// interpret_result(), array half.  The final matrix holds ~n entries, so this is a host pass over a
// D2H copy; the per-iteration work never leaves the device.
extern "C" int hhx_interpret(const hhx_csr *m, i32 *att, i32 *att_ptr, i32 *members, i32 *n_att) {
    if (!m || !att || !att_ptr || !members || !n_att) return fail("null pointer");
    const i32 n = m->n_rows;
    std::vector<i32> ip((size_t)n + 1), ix((size_t)m->nnz);
    std::vector<float> dx((size_t)m->nnz);
    HHX_TRY(hhx_csr_to_host(m, ip.data(), ix.data(), dx.data()));
    std::vector<i32> slot((size_t)n, -1);
    i32 na = 0;
    for (i32 r = 0; r < n; ++r)
        for (i32 p = ip[r]; p < ip[r + 1]; ++p)
            if (ix[p] == r && dx[p] != 0.0f) { slot[r] = na; att[na++] = r; break; }
    std::vector<i32> cnt((size_t)na + 1, 0);
    for (i32 r = 0; r < n; ++r)
        for (i32 p = ip[r]; p < ip[r + 1]; ++p)
            if (dx[p] != 0.0f && ix[p] >= 0 && ix[p] < n && slot[ix[p]] >= 0) cnt[slot[ix[p]] + 1]++;
    att_ptr[0] = 0;
    for (i32 a = 0; a < na; ++a) att_ptr[a + 1] = att_ptr[a] + cnt[a + 1];
    std::fill(cnt.begin(), cnt.end(), 0);
    for (i32 r = 0; r < n; ++r)
        for (i32 p = ip[r]; p < ip[r + 1]; ++p) {
            if (dx[p] == 0.0f || ix[p] < 0 || ix[p] >= n) continue;
            const i32 s = slot[ix[p]];
            if (s >= 0) members[att_ptr[s] + cnt[s]++] = r;
        }
    *n_att = na;
    return 0;
}
