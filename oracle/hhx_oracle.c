/*
 * hhx_oracle.c — CPU restatement of HapHiC's link-matrix + Markov-clustering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under haphic_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity pin: every function here is checked (tests/test_oracle_golden.py) against golden vectors
 * produced by running the reference's own Python functions (tests/golden/make_golden.py imports
 * /root/reference/scripts/HapHiC_cluster.py).  The reference has no tests of its own.
 *
 * Matrix convention: the reference keeps a column-stochastic matrix M in scipy CSC.  The CSC triple
 * (indptr, indices, data) of M is, byte for byte, the CSR triple of T = M^T.  Everything below is
 * written for CSR(T): "row j" here == "column j" in HapHiC_cluster.py.
 *
 * Each function cites the reference lines (HapHiC_cluster.py unless said otherwise) it restates.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t i32;
typedef int64_t i64;

/* ------------------------------------------------------------------------------------------------
 * L1 normalisation of every row of CSR(T)  ==  sklearn.preprocessing.normalize(M, 'l1', axis=0)
 * reference call sites :2014 :2038 :2144; arithmetic = sklearn utils/sparsefuncs_fast.pyx
 * _inplace_csr_row_normalize_l1: double accumulator, sequential storage order, x = float(x / sum).
 * ---------------------------------------------------------------------------------------------- */
static int g_threads;                            /* 0: OpenMP default (all cores); orc_set_threads */
int orc_get_threads(void);
/* rows are independent: the row loops below are spread over the host threads (a row's arithmetic is the serial loop's, so every
 * result bit is too); a call from inside a parallel region (orc_links_iteration0) runs its rows on the calling thread */
void orc_normalize_l1(i32 n, const i32 *indptr, float *data) {
#pragma omp parallel for schedule(static, 256) num_threads(orc_get_threads()) if (n > 4096)
    for (i32 r = 0; r < n; ++r) {
        double s = 0.0;
        for (i32 p = indptr[r]; p < indptr[r + 1]; ++p) s += fabs((double)data[p]);
        if (s == 0.0) continue;
        for (i32 p = indptr[r]; p < indptr[r + 1]; ++p) data[p] = (float)((double)data[p] / s);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Hadamard power (inflation), :2038 `matrix.power(inflation)` -> numpy float32 `data ** r`.
 * numpy special-cases r == 2 to x*x (fast_scalar_power); otherwise float32 powf with the exponent
 * rounded to float32 (weak python scalar).  numpy's SIMD powf is within 1 ulp of glibc's.
 * ---------------------------------------------------------------------------------------------- */
void orc_power(i64 nnz, float *data, double r) {
    if (r == 2.0) {
#pragma omp parallel for schedule(static) num_threads(orc_get_threads()) if (nnz > 1000000)
        for (i64 p = 0; p < nnz; ++p) data[p] = data[p] * data[p];
    } else {
        float rf = (float)r;
#pragma omp parallel for schedule(static) num_threads(orc_get_threads()) if (nnz > 100000)
        for (i64 p = 0; p < nnz; ++p) data[p] = powf(data[p], rf);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Expansion: C = A*B on CSR(T) (== (M_B * M_A) on the reference's CSC view; for the square both
 * operands are the same matrix, :2017-2023 mkl_matrix_power -> sparse_dot_mkl.dot_product_mkl,
 * pinned sparse_dot_mkl==0.9.4 / mkl==2024.2.0, float32 in/out).  MKL's accumulation order is
 * unspecified; this restatement uses Gustavson row-by-row with a float32 accumulator visiting the
 * k index in ascending order — bit-identical to scipy's csr_matmat (the SURVEY's MKL stand-in) on
 * canonical inputs.  Output rows are sorted by column.
 *   mode 0: float32 accumulation (reference-like).
 *   mode 1: the HIP kernel's specification — each product a*b formed exactly in double, rounded to the
 *           nearest multiple of 2^-fx_shift (ties to even), summed exactly (order independent: 64-bit
 *           integers here, exact double adds in the fused HIP kernel), rounded to float32 once.
 *           Used for bit-exact kernel checks.
 * Two-call pattern: pass Cj == NULL to get the row pointer only.
 * ---------------------------------------------------------------------------------------------- */
static int cmp_i32(const void *a, const void *b) {
    i32 x = *(const i32 *)a, y = *(const i32 *)b;
    return (x > y) - (x < y);
}

/* Rows are independent, so the sweep is spread over host threads (OpenMP, static per-thread scratch); the
 * arithmetic inside a row — and therefore every result bit, mode 0's sequential float32 sums included — is
 * what the serial loop produces.  orc_set_threads(1) (bench.py's cpu_baseline leg) gives the scalar port. */
void orc_set_threads(int n) { g_threads = n; }
int orc_get_threads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

i64 orc_spgemm(i32 n_rows, i32 n_cols, const i32 *Ap, const i32 *Aj, const float *Ax, const i32 *Bp,
               const i32 *Bj, const float *Bx, i32 *Cp, i32 *Cj, float *Cx, int mode, int fx_shift) {
    const double scale = ldexp(1.0, fx_shift), inv_scale = ldexp(1.0, -fx_shift);
    const int nt = orc_get_threads();
    i64 *row_nnz = (i64 *)calloc((size_t)n_rows + 1, sizeof(i64));
    for (int pass = 0; pass < (Cj ? 2 : 1); ++pass) {
#pragma omp parallel num_threads(nt)
        {
            /* the "touched" marks are a bitmap (n / 8 bytes: L1-resident) and only the accumulator of the mode in use exists, so that
             * a row's working set — 4 or 8 bytes per column — stays in the core's L2 at n = 100k (the tails of a 100k-contig mcl()
             * walk 10^11-10^12 products on the host: with an int32 mark + two accumulators per column every product missed) */
            uint64_t *mark = (uint64_t *)calloc(((size_t)n_cols + 63) / 64 + 1, sizeof(uint64_t));
            float *accf = mode == 0 ? (float *)calloc((size_t)n_cols + 1, sizeof(float)) : NULL;
            i64 *acci = mode != 0 ? (i64 *)calloc((size_t)n_cols + 1, sizeof(i64)) : NULL;
            i32 *cols = (i32 *)malloc(sizeof(i32) * ((size_t)n_cols + 1));
#pragma omp for schedule(dynamic, 16)
            for (i32 i = 0; i < n_rows; ++i) {
                i32 cnt = 0;
                for (i32 p = Ap[i]; p < Ap[i + 1]; ++p) {
                    i32 k = Aj[p];
                    float a = Ax[p];
                    for (i32 q = Bp[k]; q < Bp[k + 1]; ++q) {
                        i32 j = Bj[q];
                        const uint64_t bit = (uint64_t)1 << (j & 63);
                        if (!(mark[j >> 6] & bit)) {
                            mark[j >> 6] |= bit;
                            cols[cnt++] = j;
                            if (mode == 0) accf[j] = 0.0f; else acci[j] = 0;
                        }
                        if (pass == 1) {
                            if (mode == 0) accf[j] += a * Bx[q];
                            else acci[j] += (i64)llrint((double)a * (double)Bx[q] * scale);
                        }
                    }
                }
                if (pass == 0) row_nnz[i + 1] = cnt;
                else {
                    const i64 base = row_nnz[i];
                    qsort(cols, (size_t)cnt, sizeof(i32), cmp_i32);
                    for (i32 c = 0; c < cnt; ++c) {
                        Cj[base + c] = cols[c];
                        Cx[base + c] = mode == 0 ? accf[cols[c]] : (float)((double)acci[cols[c]] * inv_scale);
                    }
                }
                for (i32 c = 0; c < cnt; ++c) mark[cols[c] >> 6] = 0;
            }
            free(mark); free(accf); free(acci); free(cols);
        }
        if (pass == 0) {
            for (i32 i = 0; i < n_rows; ++i) row_nnz[i + 1] += row_nnz[i];
            for (i32 i = 0; i <= n_rows; ++i) Cp[i] = (i32)row_nnz[i];
        }
    }
    const i64 nnz = row_nnz[n_rows];
    free(row_nnz);
    return nnz;
}

/* The same product in ONE pass over the products, for orc_mcl_from (whose expansions at 100k contigs walk 10^10-10^12 products): every
 * thread appends its finished rows to its own growing buffer, a second (parallel) step copies them to their place.  Row for row the
 * arithmetic of orc_spgemm above — same visiting order of k and q, same accumulators, same rounding: the same bits.  The three
 * output arrays are malloc'ed here; returns nnz. */
static i64 spgemm_onepass(i32 n_rows, i32 n_cols, const i32 *Ap, const i32 *Aj, const float *Ax, const i32 *Bp, const i32 *Bj, const float *Bx,
                          int mode, int fx_shift, i32 **Cp_out, i32 **Cj_out, float **Cx_out) {
    const double scale = ldexp(1.0, fx_shift), inv_scale = ldexp(1.0, -fx_shift);
    const int nt = orc_get_threads();
    i32 *Cp = (i32 *)calloc((size_t)n_rows + 1, sizeof(i32));
    i64 *row_at = (i64 *)malloc(sizeof(i64) * ((size_t)n_rows + 1));       /* offset of the row inside its thread's buffer */
    i32 *row_th = (i32 *)malloc(sizeof(i32) * ((size_t)n_rows + 1));
    i32 **bj = (i32 **)calloc((size_t)nt, sizeof(i32 *));
    float **bx = (float **)calloc((size_t)nt, sizeof(float *));
#pragma omp parallel num_threads(nt)
    {
        const int th = omp_get_thread_num();
        uint64_t *mark = (uint64_t *)calloc(((size_t)n_cols + 63) / 64 + 1, sizeof(uint64_t));
        float *accf = mode == 0 ? (float *)calloc((size_t)n_cols + 1, sizeof(float)) : NULL;
        i64 *acci = mode != 0 ? (i64 *)calloc((size_t)n_cols + 1, sizeof(i64)) : NULL;
        i32 *cols = (i32 *)malloc(sizeof(i32) * ((size_t)n_cols + 1));
        size_t cap = (size_t)1 << 16, used = 0;
        i32 *oj = (i32 *)malloc(sizeof(i32) * cap);
        float *ox = (float *)malloc(sizeof(float) * cap);
#pragma omp for schedule(dynamic, 16)
        for (i32 i = 0; i < n_rows; ++i) {
            i32 cnt = 0;
            for (i32 p = Ap[i]; p < Ap[i + 1]; ++p) {
                const i32 k = Aj[p];
                const float a = Ax[p];
                for (i32 q = Bp[k]; q < Bp[k + 1]; ++q) {
                    const i32 j = Bj[q];
                    const uint64_t bit = (uint64_t)1 << (j & 63);
                    if (!(mark[j >> 6] & bit)) {
                        mark[j >> 6] |= bit;
                        cols[cnt++] = j;
                        if (mode == 0) accf[j] = 0.0f; else acci[j] = 0;
                    }
                    if (mode == 0) accf[j] += a * Bx[q];
                    else acci[j] += (i64)llrint((double)a * (double)Bx[q] * scale);
                }
            }
            qsort(cols, (size_t)cnt, sizeof(i32), cmp_i32);
            if (used + (size_t)cnt > cap) {
                while (used + (size_t)cnt > cap) cap *= 2;
                oj = (i32 *)realloc(oj, sizeof(i32) * cap);
                ox = (float *)realloc(ox, sizeof(float) * cap);
            }
            for (i32 c = 0; c < cnt; ++c) {
                oj[used + c] = cols[c];
                ox[used + c] = mode == 0 ? accf[cols[c]] : (float)((double)acci[cols[c]] * inv_scale);
                mark[cols[c] >> 6] = 0;
            }
            Cp[i + 1] = cnt;
            row_at[i] = (i64)used;
            row_th[i] = th;
            used += (size_t)cnt;
        }
        bj[th] = oj; bx[th] = ox;
        free(mark); free(accf); free(acci); free(cols);
    }
    i64 nnz = 0;
    for (i32 i = 0; i < n_rows; ++i) { nnz += Cp[i + 1]; if (nnz > 2147483647LL) nnz = 2147483647LL; Cp[i + 1] = (i32)nnz; }
    i32 *Cj = (i32 *)malloc(sizeof(i32) * (size_t)(nnz ? nnz : 1));
    float *Cx = (float *)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1));
#pragma omp parallel for schedule(static, 256) num_threads(nt)
    for (i32 i = 0; i < n_rows; ++i) {
        const i32 len = Cp[i + 1] - Cp[i];
        memcpy(Cj + Cp[i], bj[row_th[i]] + row_at[i], sizeof(i32) * (size_t)len);
        memcpy(Cx + Cp[i], bx[row_th[i]] + row_at[i], sizeof(float) * (size_t)len);
    }
    for (int t = 0; t < nt; ++t) { free(bj[t]); free(bx[t]); }
    free(bj); free(bx); free(row_at); free(row_th);
    *Cp_out = Cp; *Cj_out = Cj; *Cx_out = Cx;
    return nnz;
}

/* ------------------------------------------------------------------------------------------------
 * Pre-expansion of the link matrix, run_mcl_clustering :2144-2147, in the INTEGER specification of the HIP kernels (mode 2).
 * L is the raw symmetric link matrix of dict_to_matrix (integer counts c, :362-368), d_i its L1 row sums (:2144), and
 *     (M^2)_ij = sum_k (c_ik / d_i)(c_kj / d_k) = (1 / d_i) * S_ij,    S_ij = sum_k c_ik c_kj / d_k  — a SYMMETRIC matrix.
 * The kernels evaluate S in exact integer arithmetic, so that only its upper block triangle has to be computed:
 *     lg = ceil(log2(max_i d_i)), s = 61 - lg (applicable while s - lg >= 24, i.e. every weight keeps >= 24 bits)
 *     W_k = rint(2^s / d_k)                          (double division, round to nearest even, as an unsigned 64-bit integer)
 *     acc_ij = sum_k c_ik * c_kj * W_k               (exact; < 2^63 because c_kj W_k <= 2^s + c_kj / 2 and sum_k c_ik = d_i)
 *     y_ij = float(double(acc_ij) * 2^-s)            (= S_ij to float32; symmetric bit for bit)
 *     x_ij = float(double(y_ij) / d_i)               (the entry of the pre-expanded matrix)
 * Against the reference's float32 product of the float32-normalised matrix this differs by the roundings of the normalised
 * entries it never forms (<= 2^-23 relative), far inside the float32 accumulation noise of the reference's own SpGEMM.
 * rows: the output rows wanted (NULL = all).  Two-call pattern like orc_spgemm.  Returns nnz, -2 if the specification is not
 * applicable (row sums beyond 2^18, a zero row sum, non-integer or negative values).
 * ---------------------------------------------------------------------------------------------- */
int orc_links_shift(i32 n, const i32 *Lp, const float *Lx, double *d_out) {
    double dmax = 0.0;
    for (i32 r = 0; r < n; ++r) {
        double s = 0.0;
        for (i32 p = Lp[r]; p < Lp[r + 1]; ++p) {
            if (!(Lx[p] >= 0.0f) || Lx[p] != rintf(Lx[p]) || Lx[p] > 65535.0f) return -1;
            s += (double)Lx[p];
        }
        if (s == 0.0) return -1;
        if (d_out) d_out[r] = s;
        if (s > dmax) dmax = s;
    }
    int lg = 0;
    while (ldexp(1.0, lg) < dmax) ++lg;
    const int shift = 61 - lg;
    return shift - lg >= 24 ? shift : -1;
}

i64 orc_expand_links_ex(i32 n, const i32 *Lp, const i32 *Lj, const float *Lx, i32 n_rows, const i32 *rows, i32 *Cp, i32 *Cj, float *Cx, int divide);
i64 orc_expand_links(i32 n, const i32 *Lp, const i32 *Lj, const float *Lx, i32 n_rows, const i32 *rows, i32 *Cp, i32 *Cj, float *Cx) {
    return orc_expand_links_ex(n, Lp, Lj, Lx, n_rows, rows, Cp, Cj, Cx, 1);
}
/* divide == 0: the values are y_ij = float(S_ij), the symmetric matrix itself (what the kernels keep in their dense block and what
 * the ranks of the multi-GPU driver hand each other), without the final division by d_i */
i64 orc_expand_links_ex(i32 n, const i32 *Lp, const i32 *Lj, const float *Lx, i32 n_rows, const i32 *rows, i32 *Cp, i32 *Cj, float *Cx, int divide) {
    double *d = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    const int shift = orc_links_shift(n, Lp, Lx, d);
    if (shift < 0) { free(d); return -2; }
    uint64_t *W = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    for (i32 k = 0; k < n; ++k) W[k] = (uint64_t)rint(ldexp(1.0, shift) / d[k]);
    const double inv = ldexp(1.0, -shift);
    const int nt = orc_get_threads();
    i64 *row_nnz = (i64 *)calloc((size_t)n_rows + 1, sizeof(i64));
    for (int pass = 0; pass < (Cj ? 2 : 1); ++pass) {
#pragma omp parallel num_threads(nt)
        {
            uint64_t *mark = (uint64_t *)calloc(((size_t)n + 63) / 64 + 1, sizeof(uint64_t));      /* bitmap: see orc_spgemm */
            uint64_t *acc = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
            i32 *cols = (i32 *)malloc(sizeof(i32) * ((size_t)n + 1));
#pragma omp for schedule(dynamic, 16)
            for (i32 t = 0; t < n_rows; ++t) {
                const i32 i = rows ? rows[t] : t;
                i32 cnt = 0;
                for (i32 p = Lp[i]; p < Lp[i + 1]; ++p) {
                    const i32 k = Lj[p];
                    const uint64_t g = (uint64_t)Lx[p] * W[k];
                    for (i32 q = Lp[k]; q < Lp[k + 1]; ++q) {
                        const i32 j = Lj[q];
                        const uint64_t bit = (uint64_t)1 << (j & 63);
                        if (!(mark[j >> 6] & bit)) { mark[j >> 6] |= bit; cols[cnt++] = j; acc[j] = 0; }
                        if (pass == 1) acc[j] += g * (uint64_t)Lx[q];
                    }
                }
                if (pass == 0) row_nnz[t + 1] = cnt;
                else {
                    const i64 base = row_nnz[t];
                    qsort(cols, (size_t)cnt, sizeof(i32), cmp_i32);
                    for (i32 c = 0; c < cnt; ++c) {
                        const float y = (float)((double)acc[cols[c]] * inv);
                        Cj[base + c] = cols[c];
                        Cx[base + c] = divide ? (float)((double)y / d[i]) : y;
                    }
                }
                for (i32 c = 0; c < cnt; ++c) mark[cols[c] >> 6] = 0;
            }
            free(mark); free(acc); free(cols);
        }
        if (pass == 0) {
            for (i32 t = 0; t < n_rows; ++t) row_nnz[t + 1] += row_nnz[t];
            for (i32 t = 0; t <= n_rows; ++t) Cp[t] = (i32)row_nnz[t];
        }
    }
    const i64 nnz = row_nnz[n_rows];
    free(row_nnz); free(W); free(d);
    return nnz;
}

i64 orc_prune(i32 n, const i32 *indptr, const i32 *indices, const float *data, double pruning,
              i32 *out_indptr, i32 *out_indices, float *out_data);
/* Iteration 0 of mcl() (:2037-2042) for the rows `rows` of the pre-expanded matrix of the RAW link matrix L, in ONE pass over
 * the products: per row, the accumulators of orc_expand_links_ex (same integers, same two roundings), then orc_power,
 * orc_normalize_l1 and orc_prune applied to that single row — the composition tests/ use on sampled rows, here without ever
 * holding a row of M^2 longer than its own turn, so that ALL rows of a 100k-contig matrix (1.2e12 products, 1e10 entries of M^2)
 * can be checked.  Out: row t occupies [t * cap_row, t * cap_row + Ocnt[t]); returns the total, -2 if the integer specification
 * does not apply, -1 if a pruned row exceeds cap_row. */
i64 orc_links_iteration0(i32 n, const i32 *Lp, const i32 *Lj, const float *Lx, i32 n_rows, const i32 *rows, double inflation,
                         double pruning, i32 cap_row, i32 *Ocnt, i32 *Oj, float *Ox, i64 *n_expanded) {
    double *d = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    const int shift = orc_links_shift(n, Lp, Lx, d);
    if (shift < 0) { free(d); return -2; }
    uint64_t *W = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    for (i32 k = 0; k < n; ++k) W[k] = (uint64_t)rint(ldexp(1.0, shift) / d[k]);
    const double inv = ldexp(1.0, -shift);
    const int nt = orc_get_threads();
    i64 total = 0, expanded = 0;
    int overflow = 0;
#pragma omp parallel num_threads(nt) reduction(+ : total, expanded) reduction(| : overflow)
    {
        uint64_t *mark = (uint64_t *)calloc(((size_t)n + 63) / 64 + 1, sizeof(uint64_t));      /* bitmap marks + 8-byte sums: a row's working set stays in L2 */
        uint64_t *acc = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
        i32 *cols = (i32 *)malloc(sizeof(i32) * (size_t)n);
        float *val = (float *)malloc(sizeof(float) * (size_t)n);
        i32 *pj = (i32 *)malloc(sizeof(i32) * (size_t)n);
        float *px = (float *)malloc(sizeof(float) * (size_t)n);
#pragma omp for schedule(dynamic, 8)
        for (i32 t = 0; t < n_rows; ++t) {
            const i32 i = rows ? rows[t] : t;
            i32 cnt = 0;
            for (i32 p = Lp[i]; p < Lp[i + 1]; ++p) {
                const i32 k = Lj[p];
                const uint64_t g = (uint64_t)Lx[p] * W[k];
                for (i32 q = Lp[k]; q < Lp[k + 1]; ++q) {
                    const i32 j = Lj[q];
                    const uint64_t bit = (uint64_t)1 << (j & 63);
                    if (!(mark[j >> 6] & bit)) { mark[j >> 6] |= bit; cols[cnt++] = j; acc[j] = 0; }
                    acc[j] += g * (uint64_t)Lx[q];
                }
            }
            qsort(cols, (size_t)cnt, sizeof(i32), cmp_i32);
            for (i32 c = 0; c < cnt; ++c) {
                const float y = (float)((double)acc[cols[c]] * inv);
                val[c] = (float)((double)y / d[i]);
                mark[cols[c] >> 6] = 0;
            }
            expanded += cnt;
            i32 one[2] = {0, cnt}, op[2];
            orc_power(cnt, val, inflation);
            orc_normalize_l1(1, one, val);
            const i64 kept = orc_prune(1, one, cols, val, pruning, op, pj, px);
            if (kept > cap_row) { overflow = 1; Ocnt[t] = 0; continue; }
            Ocnt[t] = (i32)kept;
            memcpy(Oj + (size_t)t * cap_row, pj, sizeof(i32) * (size_t)kept);
            memcpy(Ox + (size_t)t * cap_row, px, sizeof(float) * (size_t)kept);
            total += kept;
        }
        free(mark); free(acc); free(cols); free(val); free(pj); free(px);
    }
    free(W); free(d);
    if (n_expanded) *n_expanded = expanded;
    return overflow ? -1 : total;
}

/* ------------------------------------------------------------------------------------------------
 * prune(), :1987-2014.  Input = inflated+normalised matrix (sorted rows).  Keep entries >= the
 * float32-rounded threshold (:1994 sparse branch / :2005 dense branch compare float32 data against
 * the python float under numpy scalar promotion -> float32 compare), restore each row's maximum from
 * the un-pruned matrix (:2010-2013; scipy argmax: first maximum in ascending index order), then L1
 * normalise (:2014).  Returns nnz; out arrays must hold nnz(in) entries.
 * ---------------------------------------------------------------------------------------------- */
i64 orc_prune(i32 n, const i32 *indptr, const i32 *indices, const float *data, double pruning,
              i32 *out_indptr, i32 *out_indices, float *out_data) {
    const float thr = (float)pruning;
    const int nt = orc_get_threads();
    /* pass 1: the first maximum and the number of survivors of every row; pass 2 (after the prefix sum): the ordered copy */
    i32 *amax = (i32 *)malloc(sizeof(i32) * (size_t)(n ? n : 1));
    out_indptr[0] = 0;
#pragma omp parallel for schedule(static, 256) num_threads(nt) if (n > 4096)
    for (i32 r = 0; r < n; ++r) {
        i32 b = indptr[r], e = indptr[r + 1];
        i32 am = -1, keep = 0;
        float m = 0.0f;
        for (i32 p = b; p < e; ++p)
            if (am < 0 || data[p] > m) { am = p; m = data[p]; }
        for (i32 p = b; p < e; ++p) keep += (data[p] >= thr || p == am);
        amax[r] = am;
        out_indptr[r + 1] = keep;
    }
    for (i32 r = 0; r < n; ++r) out_indptr[r + 1] += out_indptr[r];
#pragma omp parallel for schedule(static, 256) num_threads(nt) if (n > 4096)
    for (i32 r = 0; r < n; ++r) {
        i64 o = out_indptr[r];
        const i32 am = amax[r];
        for (i32 p = indptr[r]; p < indptr[r + 1]; ++p) {
            if (data[p] >= thr || p == am) {
                out_indices[o] = indices[p];
                out_data[o] = data[p];
                ++o;
            }
        }
    }
    free(amax);
    orc_normalize_l1(n, out_indptr, out_data);
    return out_indptr[n];
}

/* ------------------------------------------------------------------------------------------------
 * Convergence statistic, :2044-2050 (sparse branch):
 *     d = abs(M - last) - 1e-5 * abs(last);  converged <=> d.max() <= 1e-8
 * evaluated in float32 over the union of both sparsity patterns (implicit zeros contribute 0).
 * Returns max(0, max over stored entries).  Both inputs must have sorted rows.
 * ---------------------------------------------------------------------------------------------- */
float orc_convergence_stat(i32 n, const i32 *ap, const i32 *aj, const float *ax, const i32 *bp,
                           const i32 *bj, const float *bx) {
    float best = 0.0f;
    const float rtol = (float)1e-5;
#pragma omp parallel for schedule(static, 256) reduction(max : best) num_threads(orc_get_threads()) if (n > 4096)
    for (i32 r = 0; r < n; ++r) {
        i32 p = ap[r], pe = ap[r + 1], q = bp[r], qe = bp[r + 1];
        while (p < pe || q < qe) {
            float m = 0.0f, l = 0.0f;
            if (q >= qe || (p < pe && aj[p] < bj[q])) m = ax[p++];
            else if (p >= pe || bj[q] < aj[p]) l = bx[q++];
            else { m = ax[p++]; l = bx[q++]; }
            float d = fabsf(m - l) - rtol * fabsf(l);
            if (d > best) best = d;
        }
    }
    return best;
}

/* ------------------------------------------------------------------------------------------------
 * mcl(), :2026-2062, sparse mode.  `T` is the pre-expanded matrix (run_mcl_clustering :2144-2147).
 * The result is written into caller buffers of capacity `cap` entries; returns nnz, or -1 if cap is
 * too small.  stats (optional, 4 i64 per iteration): nnz_A, nnz_C, nnz_P, F (#products).
 * ---------------------------------------------------------------------------------------------- */
static i64 count_products(i32 n, const i32 *Ap, const i32 *Aj, const i32 *Bp) {
    i64 f = 0;
#pragma omp parallel for schedule(static, 256) reduction(+ : f) num_threads(orc_get_threads()) if (n > 4096)
    for (i32 i = 0; i < n; ++i)
        for (i32 p = Ap[i]; p < Ap[i + 1]; ++p) f += Bp[Aj[p] + 1] - Bp[Aj[p]];
    return f;
}

i64 orc_mcl_from(i32 n, const i32 *indptr, const i32 *indices, const float *data, int expansion,
                 double inflation, int iters, double pruning, int spgemm_mode, int fx_shift, i64 cap,
                 i32 *out_indptr, i32 *out_indices, float *out_data, int *n_iter, int *converged,
                 i64 *stats, int first_it) {
    i64 nnz = indptr[n];
    i32 *cp = (i32 *)malloc(sizeof(i32) * ((size_t)n + 1));
    i32 *cj = (i32 *)malloc(sizeof(i32) * (size_t)(nnz ? nnz : 1));
    float *cx = (float *)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1));
    memcpy(cp, indptr, sizeof(i32) * ((size_t)n + 1));
    memcpy(cj, indices, sizeof(i32) * (size_t)nnz);
    memcpy(cx, data, sizeof(float) * (size_t)nnz);
    i32 *lp = NULL, *lj = NULL;
    float *lx = NULL;
    *converged = 0;
    *n_iter = first_it;
    if (first_it >= 2) {         /* picked up after `first_it` iterations: the input is what the last of them left (last_matrix) */
        lp = (i32 *)malloc(sizeof(i32) * ((size_t)n + 1));
        lj = (i32 *)malloc(sizeof(i32) * (size_t)(nnz ? nnz : 1));
        lx = (float *)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1));
        memcpy(lp, cp, sizeof(i32) * ((size_t)n + 1));
        memcpy(lj, cj, sizeof(i32) * (size_t)nnz);
        memcpy(lx, cx, sizeof(float) * (size_t)nnz);
    }
    for (int it = first_it; it < iters; ++it) {
        i64 st_a = cp[n], st_f = 0;
        if (it != 0 && expansion > 1) { /* 2) expand, :2030-2035 */
            /* mkl_matrix_power(M, e) = M * M^(e-1) on CSC (:2017-2023)  ==  T^(e-1) * T on CSR(T) */
            i32 *bp = cp, *bj = cj; float *bx = cx;     /* the operand T */
            i32 *rp = cp, *rj = cj; float *rx = cx;     /* running power */
            for (int e = 2; e <= expansion; ++e) {
                i32 *np_ = NULL, *nj = NULL;
                float *nx = NULL;
                st_f += count_products(n, rp, rj, bp);
                spgemm_onepass(n, n, rp, rj, rx, bp, bj, bx, spgemm_mode, fx_shift, &np_, &nj, &nx);
                if (rp != bp) { free(rp); free(rj); free(rx); }
                rp = np_; rj = nj; rx = nx;
            }
            free(bp); free(bj); free(bx);
            cp = rp; cj = rj; cx = rx;
        }
        i64 st_c = cp[n];
        /* 3) inflate, :2037-2038 */
        orc_power(cp[n], cx, inflation);
        orc_normalize_l1(n, cp, cx);
        /* 4) prune, :2042 */
        i32 *pp = (i32 *)malloc(sizeof(i32) * ((size_t)n + 1));
        i32 *pj = (i32 *)malloc(sizeof(i32) * (size_t)(cp[n] ? cp[n] : 1));
        float *px = (float *)malloc(sizeof(float) * (size_t)(cp[n] ? cp[n] : 1));
        orc_prune(n, cp, cj, cx, pruning, pp, pj, px);
        free(cp); free(cj); free(cx);
        cp = pp; cj = pj; cx = px;
        if (stats) { stats[4 * it] = st_a; stats[4 * it + 1] = st_c; stats[4 * it + 2] = cp[n]; stats[4 * it + 3] = st_f; }
        *n_iter = it + 1;
        /* 5) convergence, :2044-2050 */
        if (it > 1) {
            float d = orc_convergence_stat(n, cp, cj, cx, lp, lj, lx);
            if (d <= (float)1e-8) { *converged = 1; break; }
        }
        /* last = M.copy(), :2057 */
        free(lp); free(lj); free(lx);
        lp = (i32 *)malloc(sizeof(i32) * ((size_t)n + 1));
        lj = (i32 *)malloc(sizeof(i32) * (size_t)(cp[n] ? cp[n] : 1));
        lx = (float *)malloc(sizeof(float) * (size_t)(cp[n] ? cp[n] : 1));
        memcpy(lp, cp, sizeof(i32) * ((size_t)n + 1));
        memcpy(lj, cj, sizeof(i32) * (size_t)cp[n]);
        memcpy(lx, cx, sizeof(float) * (size_t)cp[n]);
    }
    i64 out_nnz = cp[n];
    if (out_nnz <= cap) {
        memcpy(out_indptr, cp, sizeof(i32) * ((size_t)n + 1));
        memcpy(out_indices, cj, sizeof(i32) * (size_t)out_nnz);
        memcpy(out_data, cx, sizeof(float) * (size_t)out_nnz);
    } else out_nnz = -1;
    free(cp); free(cj); free(cx); free(lp); free(lj); free(lx);
    return out_nnz;
}

i64 orc_mcl(i32 n, const i32 *indptr, const i32 *indices, const float *data, int expansion,
            double inflation, int iters, double pruning, int spgemm_mode, int fx_shift, i64 cap,
            i32 *out_indptr, i32 *out_indices, float *out_data, int *n_iter, int *converged,
            i64 *stats) {
    return orc_mcl_from(n, indptr, indices, data, expansion, inflation, iters, pruning, spgemm_mode, fx_shift, cap,
                        out_indptr, out_indices, out_data, n_iter, converged, stats, 0);
}

/* ------------------------------------------------------------------------------------------------
 * interpret_result(), :2065-2095, array half: attractors = rows of M with a non-zero diagonal; the
 * cluster of attractor a = non-zero columns of ROW a of M == rows j of CSR(T) that hold column a.
 * Output: CSR-like (att_ptr, members) with attractors ascending and members ascending; returns the
 * number of attractors.  The set-of-tuples + partition validation (:2084-2095) stays in Python on
 * both sides (its iteration order is CPython's).
 * members must hold nnz entries, att/att_ptr n(+1).
 * ---------------------------------------------------------------------------------------------- */
i32 orc_interpret(i32 n, const i32 *indptr, const i32 *indices, const float *data, i32 *att,
                  i32 *att_ptr, i32 *members) {
    i32 *slot = (i32 *)malloc(sizeof(i32) * (size_t)n);
    i32 na = 0;
    for (i32 r = 0; r < n; ++r) {
        slot[r] = -1;
        for (i32 p = indptr[r]; p < indptr[r + 1]; ++p)
            if (indices[p] == r && data[p] != 0.0f) { slot[r] = na; att[na++] = r; break; }
    }
    i32 *cnt = (i32 *)calloc((size_t)na + 1, sizeof(i32));
    for (i32 r = 0; r < n; ++r)
        for (i32 p = indptr[r]; p < indptr[r + 1]; ++p)
            if (data[p] != 0.0f && slot[indices[p]] >= 0) cnt[slot[indices[p]] + 1]++;
    att_ptr[0] = 0;
    for (i32 a = 0; a < na; ++a) att_ptr[a + 1] = att_ptr[a] + cnt[a + 1];
    for (i32 a = 0; a <= na; ++a) cnt[a] = 0;
    for (i32 r = 0; r < n; ++r)
        for (i32 p = indptr[r]; p < indptr[r + 1]; ++p) {
            i32 s = slot[indices[p]];
            if (data[p] != 0.0f && s >= 0) members[att_ptr[s] + cnt[s]++] = r;
        }
    free(slot); free(cnt);
    return na;
}

/* ================================================================================================
 * Ingest: parse_alignments_for_ctgs (:1596-1655) and parse_alignments (:1658-1752) on integer ids.
 *
 * The Python host maps names to ids once: contig id c in [0, n_ctg) (any order) with
 *   ctg_rank[c]  = rank of the contig NAME in Python string order (key orientation, :1629),
 *   ctg_len[c]   = contig length,
 * and fragment ids f in [0, n_frag): an unsplit contig is one fragment, a split contig owns
 * nbins consecutive fragments starting at ctg_frag0[c] (bin k=1..nbins -> ctg_frag0[c]+k-1), with
 *   frag_rank[f] = rank of the fragment NAME (bins compare as strings: "_bin10" < "_bin2", :1720),
 *   frag_len[f], frag_nx[f] (membership in Nx_frag_set).
 * id < 0 means "name not in fa_dict" (:1625 / :1702).
 *
 * Outputs are insertion-ordered tables (Python dict order):
 *   full  : (ctg_i, ctg_j) -> count, HT counts [HH, HT, TH, TT]
 *   flank : (frag_i, frag_j) -> count
 *   frag_links[f] : per-fragment flank link total (ctg_link_dict / frag_link_dict)
 *   clm   : per full key, 4 distances per pair in stream order (update_clm_dict :395-401)
 *   coords: per full key, the first max_read_pairs (coord_i, coord_j) in stream order (:454-471)
 * ============================================================================================== */
typedef struct {
    i64 cap, n;           /* hash capacity (power of two), number of keys */
    i64 *slot_key;        /* -1 = empty */
    i64 *slot_idx;        /* insertion index */
    i64 *keys;            /* insertion-ordered keys */
    i64 keys_cap;
} omap_t;

static void omap_init(omap_t *m) {
    m->cap = 1024; m->n = 0; m->keys_cap = 512;
    m->slot_key = (i64 *)malloc(sizeof(i64) * (size_t)m->cap);
    m->slot_idx = (i64 *)malloc(sizeof(i64) * (size_t)m->cap);
    m->keys = (i64 *)malloc(sizeof(i64) * (size_t)m->keys_cap);
    for (i64 i = 0; i < m->cap; ++i) m->slot_key[i] = -1;
}
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static void omap_grow(omap_t *m) {
    i64 ncap = m->cap * 2;
    i64 *nk = (i64 *)malloc(sizeof(i64) * (size_t)ncap), *ni = (i64 *)malloc(sizeof(i64) * (size_t)ncap);
    for (i64 i = 0; i < ncap; ++i) nk[i] = -1;
    for (i64 i = 0; i < m->cap; ++i)
        if (m->slot_key[i] >= 0) {
            uint64_t h = mix64((uint64_t)m->slot_key[i]) & (uint64_t)(ncap - 1);
            while (nk[h] >= 0) h = (h + 1) & (uint64_t)(ncap - 1);
            nk[h] = m->slot_key[i]; ni[h] = m->slot_idx[i];
        }
    free(m->slot_key); free(m->slot_idx);
    m->slot_key = nk; m->slot_idx = ni; m->cap = ncap;
}
/* returns insertion index, *is_new set */
static i64 omap_get(omap_t *m, i64 key, int *is_new) {
    if ((m->n + 1) * 2 > m->cap) omap_grow(m);
    uint64_t h = mix64((uint64_t)key) & (uint64_t)(m->cap - 1);
    while (m->slot_key[h] >= 0) {
        if (m->slot_key[h] == key) { *is_new = 0; return m->slot_idx[h]; }
        h = (h + 1) & (uint64_t)(m->cap - 1);
    }
    m->slot_key[h] = key; m->slot_idx[h] = m->n;
    if (m->n == m->keys_cap) { m->keys_cap *= 2; m->keys = (i64 *)realloc(m->keys, sizeof(i64) * (size_t)m->keys_cap); }
    m->keys[m->n] = key;
    *is_new = 1;
    return m->n++;
}
static void omap_free(omap_t *m) { free(m->slot_key); free(m->slot_idx); free(m->keys); }

typedef struct {
    omap_t full, flank;
    i64 *full_cnt, *ht_cnt /* 4 per key */, *flank_cnt;
    i64 full_cap, flank_cap;
    i64 *frag_links; i32 n_frag;
    /* clm / coords: per-key growable arrays */
    i64 **clm; i64 *clm_len, *clm_cap;
    i64 **crd; i64 *crd_len;
    int want_clm, max_read_pairs;
} ingest_t;

static int is_flank(i64 coord, i64 length, i64 flank) { /* :299-307 */
    if (flank && (coord <= flank || coord > length - flank)) return 1;
    if (!flank) return 1;
    return 0;
}

void *orc_ingest_new(i32 n_frag, int want_clm, int max_read_pairs) {
    ingest_t *g = (ingest_t *)calloc(1, sizeof(ingest_t));
    omap_init(&g->full); omap_init(&g->flank);
    g->full_cap = 512; g->flank_cap = 512;
    g->full_cnt = (i64 *)calloc((size_t)g->full_cap, sizeof(i64));
    g->ht_cnt = (i64 *)calloc((size_t)g->full_cap * 4, sizeof(i64));
    g->flank_cnt = (i64 *)calloc((size_t)g->flank_cap, sizeof(i64));
    g->frag_links = (i64 *)calloc((size_t)(n_frag > 0 ? n_frag : 1), sizeof(i64));
    g->n_frag = n_frag;
    g->want_clm = want_clm; g->max_read_pairs = max_read_pairs;
    if (want_clm || max_read_pairs) {
        g->clm = (i64 **)calloc((size_t)g->full_cap, sizeof(i64 *));
        g->clm_len = (i64 *)calloc((size_t)g->full_cap, sizeof(i64));
        g->clm_cap = (i64 *)calloc((size_t)g->full_cap, sizeof(i64));
        g->crd = (i64 **)calloc((size_t)g->full_cap, sizeof(i64 *));
        g->crd_len = (i64 *)calloc((size_t)g->full_cap, sizeof(i64));
    }
    return g;
}

static void full_grow(ingest_t *g) {
    i64 oc = g->full_cap, nc = oc * 2;
    g->full_cnt = (i64 *)realloc(g->full_cnt, sizeof(i64) * (size_t)nc);
    g->ht_cnt = (i64 *)realloc(g->ht_cnt, sizeof(i64) * (size_t)nc * 4);
    memset(g->full_cnt + oc, 0, sizeof(i64) * (size_t)oc);
    memset(g->ht_cnt + oc * 4, 0, sizeof(i64) * (size_t)oc * 4);
    if (g->clm) {
        g->clm = (i64 **)realloc(g->clm, sizeof(i64 *) * (size_t)nc);
        g->clm_len = (i64 *)realloc(g->clm_len, sizeof(i64) * (size_t)nc);
        g->clm_cap = (i64 *)realloc(g->clm_cap, sizeof(i64) * (size_t)nc);
        g->crd = (i64 **)realloc(g->crd, sizeof(i64 *) * (size_t)nc);
        g->crd_len = (i64 *)realloc(g->crd_len, sizeof(i64) * (size_t)nc);
        memset(g->clm + oc, 0, sizeof(i64 *) * (size_t)oc);
        memset(g->clm_len + oc, 0, sizeof(i64) * (size_t)oc);
        memset(g->clm_cap + oc, 0, sizeof(i64) * (size_t)oc);
        memset(g->crd + oc, 0, sizeof(i64 *) * (size_t)oc);
        memset(g->crd_len + oc, 0, sizeof(i64) * (size_t)oc);
    }
    g->full_cap = nc;
}

/* One batch of alignments.  bins == 0: parse_alignments_for_ctgs; bins == 1: parse_alignments.
 * pos are 0-based as yielded by the generators (:1556, :1593).  bin_size only used when bins. */
void orc_ingest_push(void *h, i64 npairs, const i32 *id1, const i64 *pos1, const i32 *id2,
                     const i64 *pos2, int bins, i32 n_ctg, const i32 *ctg_rank, const i64 *ctg_len,
                     const i32 *ctg_frag0, const unsigned char *ctg_split, i64 bin_size,
                     const i32 *frag_rank, const i64 *frag_len, const unsigned char *frag_nx,
                     i64 flank) {
    ingest_t *g = (ingest_t *)h;
    for (i64 t = 0; t < npairs; ++t) {
        i32 r = id1[t], m = id2[t];
        if (bins) { /* :1699 skip intra-contig links of unsplit contigs (by NAME equality) */
            if (r == m && (r < 0 || !ctg_split[r])) continue;
        }
        if (r < 0 || m < 0 || r >= n_ctg || m >= n_ctg) continue; /* :1625 / :1702 */
        /* sorted(((ref,pos+1),(mref,mpos+1))) :1629 / :1706 — name first, then coordinate */
        i32 ci = r, cj = m;
        i64 xi = pos1[t] + 1, xj = pos2[t] + 1;
        if (ctg_rank[r] > ctg_rank[m] || (r == m && xi > xj)) { ci = m; cj = r; i64 tx = xi; xi = xj; xj = tx; }
        i32 fi = ctg_frag0[ci], fj = ctg_frag0[cj];
        i64 yi = xi, yj = xj;
        int i_bin = 0, j_bin = 0;
        if (bins) { /* convert_frags :1662-1670 */
            if (ctg_split[ci]) { i64 nb = (xi + bin_size - 1) / bin_size; fi += (i32)(nb - 1); yi = xi - (nb - 1) * bin_size; i_bin = 1; }
            if (ctg_split[cj]) { i64 nb = (xj + bin_size - 1) / bin_size; fj += (i32)(nb - 1); yj = xj - (nb - 1) * bin_size; j_bin = 1; }
            if (fi == fj) continue; /* :1715 */
            if (i_bin || j_bin) {     /* :1719-1720 re-sort by fragment name, then coordinate */
                if (frag_rank[fi] > frag_rank[fj]) { i32 tf = fi; fi = fj; fj = tf; i64 ty = yi; yi = yj; yj = ty; }
            }
        }
        /* flank links, :1636-1639 / :1726-1729 */
        if (frag_nx[fi] && frag_nx[fj] && is_flank(yi, frag_len[fi], flank) && is_flank(yj, frag_len[fj], flank)) {
            int isnew;
            i64 k = omap_get(&g->flank, ((i64)fi << 32) | (i64)(uint32_t)fj, &isnew);
            if (k >= g->flank_cap) {
                g->flank_cnt = (i64 *)realloc(g->flank_cnt, sizeof(i64) * (size_t)g->flank_cap * 2);
                memset(g->flank_cnt + g->flank_cap, 0, sizeof(i64) * (size_t)g->flank_cap);
                g->flank_cap *= 2;
            }
            g->flank_cnt[k]++;
            g->frag_links[fi]++; g->frag_links[fj]++;
        }
        if (bins && r == m) continue; /* :1736 intra-contig links feed only the flank dict */
        {
            int isnew;
            i64 k = omap_get(&g->full, ((i64)ci << 32) | (i64)(uint32_t)cj, &isnew);
            if (k >= g->full_cap) full_grow(g);
            i64 li = ctg_len[ci], lj = ctg_len[cj];
            if (g->want_clm) { /* update_clm_dict :395-401 with 0-based coords */
                i64 a = xi - 1, b = xj - 1;
                if (g->clm_len[k] + 4 > g->clm_cap[k]) {
                    g->clm_cap[k] = g->clm_cap[k] ? g->clm_cap[k] * 2 : 8;
                    g->clm[k] = (i64 *)realloc(g->clm[k], sizeof(i64) * (size_t)g->clm_cap[k]);
                }
                i64 *d = g->clm[k] + g->clm_len[k];
                d[0] = li - a + b; d[1] = li - a + lj - b; d[2] = a + b; d[3] = a + lj - b;
                g->clm_len[k] += 4;
            }
            /* update_HT_link_dict :404-416: suffix _T iff coord*2 > len */
            int ti = xi * 2 > li, tj = xj * 2 > lj;
            g->ht_cnt[k * 4 + ti * 2 + tj]++;
            g->full_cnt[k]++; /* :1649 */
            if (g->max_read_pairs && g->crd_len[k] < 2 * (i64)g->max_read_pairs) { /* :454-459 */
                if (!g->crd[k]) g->crd[k] = (i64 *)malloc(sizeof(i64) * 2 * (size_t)g->max_read_pairs);
                g->crd[k][g->crd_len[k]++] = xi;
                g->crd[k][g->crd_len[k]++] = xj;
            }
        }
    }
}

void orc_ingest_sizes(void *h, i64 *n_full, i64 *n_flank, i64 *clm_total, i64 *crd_total) {
    ingest_t *g = (ingest_t *)h;
    *n_full = g->full.n; *n_flank = g->flank.n;
    i64 c = 0, d = 0;
    if (g->clm) for (i64 k = 0; k < g->full.n; ++k) { c += g->clm_len[k]; d += g->crd_len[k]; }
    *clm_total = c; *crd_total = d;
}

void orc_ingest_fetch(void *h, i32 *full_i, i32 *full_j, i64 *full_cnt, i64 *ht_cnt, i32 *flank_i,
                      i32 *flank_j, i64 *flank_cnt, i64 *frag_links, i64 *clm_ptr, i64 *clm,
                      i64 *crd_ptr, i64 *crd) {
    ingest_t *g = (ingest_t *)h;
    for (i64 k = 0; k < g->full.n; ++k) {
        full_i[k] = (i32)(g->full.keys[k] >> 32);
        full_j[k] = (i32)(g->full.keys[k] & 0xffffffff);
        full_cnt[k] = g->full_cnt[k];
        memcpy(ht_cnt + 4 * k, g->ht_cnt + 4 * k, sizeof(i64) * 4);
    }
    for (i64 k = 0; k < g->flank.n; ++k) {
        flank_i[k] = (i32)(g->flank.keys[k] >> 32);
        flank_j[k] = (i32)(g->flank.keys[k] & 0xffffffff);
        flank_cnt[k] = g->flank_cnt[k];
    }
    memcpy(frag_links, g->frag_links, sizeof(i64) * (size_t)g->n_frag);
    if (g->clm && clm_ptr) {
        i64 c = 0, d = 0;
        for (i64 k = 0; k < g->full.n; ++k) {
            clm_ptr[k] = c; crd_ptr[k] = d;
            if (g->clm_len[k]) memcpy(clm + c, g->clm[k], sizeof(i64) * (size_t)g->clm_len[k]);
            if (g->crd_len[k]) memcpy(crd + d, g->crd[k], sizeof(i64) * (size_t)g->crd_len[k]);
            c += g->clm_len[k]; d += g->crd_len[k];
        }
        clm_ptr[g->full.n] = c; crd_ptr[g->full.n] = d;
    }
}

void orc_ingest_free(void *h) {
    ingest_t *g = (ingest_t *)h;
    if (g->clm) {
        for (i64 k = 0; k < g->full.n; ++k) { free(g->clm[k]); free(g->crd[k]); }
        free(g->clm); free(g->clm_len); free(g->clm_cap); free(g->crd); free(g->crd_len);
    }
    omap_free(&g->full); omap_free(&g->flank);
    free(g->full_cnt); free(g->ht_cnt); free(g->flank_cnt); free(g->frag_links);
    free(g);
}

/* ------------------------------------------------------------------------------------------------
 * dict_to_matrix(), :310-373, sparse branch, array half.  Input: the flank table in insertion order
 * (fi, fj, value) and a membership flag per fragment (frag_set).  Index assignment :337-349 = first
 * appearance scanning items in order, i before j; link-less fragments (:357-359, Python set order)
 * are appended by the Python caller, which passes their count as n_rest so that shape = n_linked +
 * n_rest.  Output CSR(T) (== CSC of the symmetric matrix) with sorted rows, self-loops of value 1
 * (:362-364), float32 values (:368).  frag_index[f] = matrix index or -1.  Returns nnz; pass
 * indices == NULL for the counting call.  n_linked returned through *n_linked.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { i32 col; float val; } ent_t;
static int cmp_ent(const void *a, const void *b) {
    i32 x = ((const ent_t *)a)->col, y = ((const ent_t *)b)->col;
    return (x > y) - (x < y);
}

i64 orc_dict_to_matrix(i64 n_keys, const i32 *fi, const i32 *fj, const double *val, i32 n_frag,
                       const unsigned char *in_set, i32 n_rest, int add_self_loops, i32 *frag_index,
                       i32 *n_linked, i32 *indptr, i32 *indices, float *data) {
    i32 idx = 0;
    for (i32 f = 0; f < n_frag; ++f) frag_index[f] = -1;
    for (i64 k = 0; k < n_keys; ++k) {
        if (!in_set[fi[k]] || !in_set[fj[k]]) continue;
        if (frag_index[fi[k]] < 0) frag_index[fi[k]] = idx++;
        if (frag_index[fj[k]] < 0) frag_index[fj[k]] = idx++;
    }
    *n_linked = idx;
    i32 shape = idx + n_rest;
    i64 *cnt = (i64 *)calloc((size_t)shape + 1, sizeof(i64));
    for (i64 k = 0; k < n_keys; ++k) {
        if (!in_set[fi[k]] || !in_set[fj[k]]) continue;
        cnt[frag_index[fi[k]] + 1]++; cnt[frag_index[fj[k]] + 1]++;
    }
    if (add_self_loops) for (i32 r = 0; r < shape; ++r) cnt[r + 1]++;
    for (i32 r = 0; r < shape; ++r) cnt[r + 1] += cnt[r];
    i64 nnz = cnt[shape];
    if (!indices) { for (i32 r = 0; r <= shape; ++r) indptr[r] = (i32)cnt[r]; free(cnt); return nnz; }
    ent_t *ent = (ent_t *)malloc(sizeof(ent_t) * (size_t)(nnz ? nnz : 1));
    i64 *cur = (i64 *)malloc(sizeof(i64) * ((size_t)shape + 1));
    memcpy(cur, cnt, sizeof(i64) * ((size_t)shape + 1));
    for (i64 k = 0; k < n_keys; ++k) {
        if (!in_set[fi[k]] || !in_set[fj[k]]) continue;
        i32 a = frag_index[fi[k]], b = frag_index[fj[k]];
        ent[cur[a]].col = b; ent[cur[a]++].val = (float)val[k];
        ent[cur[b]].col = a; ent[cur[b]++].val = (float)val[k];
    }
    if (add_self_loops) for (i32 r = 0; r < shape; ++r) { ent[cur[r]].col = r; ent[cur[r]++].val = 1.0f; }
    for (i32 r = 0; r < shape; ++r) qsort(ent + cnt[r], (size_t)(cnt[r + 1] - cnt[r]), sizeof(ent_t), cmp_ent);
    /* coo -> csc sums duplicates (scipy tocsc); duplicates cannot occur here (keys unique, i != j) */
    for (i32 r = 0; r <= shape; ++r) indptr[r] = (i32)cnt[r];
    for (i64 p = 0; p < nnz; ++p) { indices[p] = ent[p].col; data[p] = ent[p].val; }
    free(ent); free(cur); free(cnt);
    return nnz;
}

/* ------------------------------------------------------------------------------------------------
 * count_RE_sites(), :75-84, over segments: counts[s] = sum over sites of seq[off:off+len].count(site).
 * Python str.count counts NON-overlapping occurrences, scanning from the left and resuming after each
 * match.  sites are concatenated, each site_len[k] bytes (after parse_RE_sites' N expansion :56-72).
 * ---------------------------------------------------------------------------------------------- */
void orc_count_re_sites(const unsigned char *seq, i64 n_seg, const i64 *seg_off, const i64 *seg_len, i32 n_sites,
                        const unsigned char *sites, const i32 *site_len, i64 *counts) {
    for (i64 s = 0; s < n_seg; ++s) {
        i64 total = 0;
        const unsigned char *pat = sites;
        for (i32 k = 0; k < n_sites; ++k) {
            const i32 L = site_len[k];
            const i64 a = seg_off[s], e = a + seg_len[s];
            for (i64 p = a; p + L <= e;) {
                if (memcmp(seq + p, pat, (size_t)L) == 0) { ++total; p += L; } else ++p;
            }
            pat += L;
        }
        counts[s] = total;
    }
}

/* ------------------------------------------------------------------------------------------------
 * filter_fragments(), rank-sum statistic :866-892, on CSR rows of the symmetric link matrix without
 * self loops.  The reference sorts every DENSE row by links descending (stable: ties by index, :874-878),
 * takes the first topN fragments and sums min(rank_a(b), rank_b(a)) over combinations(top, 2) (:884-887).
 * Restated literally with a dense scratch row (n small in tests).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float v; i32 i; } rk_t;
static int cmp_rk(const void *a, const void *b) {
    const rk_t *x = (const rk_t *)a, *y = (const rk_t *)b;
    if (x->v != y->v) return x->v > y->v ? -1 : 1;
    return (x->i > y->i) - (x->i < y->i);
}
void orc_rank_sums(i32 n, const i32 *indptr, const i32 *indices, const float *data, int topN, i64 *out) {
    i32 *rank = (i32 *)malloc(sizeof(i32) * (size_t)n * (size_t)n);    /* rank[a*n + x] = position of x in the ranking of a */
    i32 *order = (i32 *)malloc(sizeof(i32) * (size_t)n * (size_t)n);
    rk_t *row = (rk_t *)malloc(sizeof(rk_t) * (size_t)n);
    for (i32 a = 0; a < n; ++a) {
        for (i32 i = 0; i < n; ++i) { row[i].v = 0.0f; row[i].i = i; }
        for (i32 p = indptr[a]; p < indptr[a + 1]; ++p) row[indices[p]].v = data[p];
        qsort(row, (size_t)n, sizeof(rk_t), cmp_rk);
        for (i32 r = 0; r < n; ++r) { order[(size_t)a * n + r] = row[r].i; rank[(size_t)a * n + row[r].i] = r; }
    }
    for (i32 f = 0; f < n; ++f) {
        const int t = topN < n ? topN : n;
        i64 s = 0;
        for (int i = 0; i < t; ++i)
            for (int j = i + 1; j < t; ++j) {
                const i32 a = order[(size_t)f * n + i], b = order[(size_t)f * n + j];
                const i32 r1 = rank[(size_t)a * n + b], r2 = rank[(size_t)b * n + a];
                s += r1 < r2 ? r1 : r2;
            }
        out[f] = s;
    }
    free(rank); free(order); free(row);
}
