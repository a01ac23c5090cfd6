// filter_fragments' rank-sum statistic, scripts/HapHiC_cluster.py:866-892 — the reference's scaling wall: it
// builds a DENSE n x n matrix (:868), sorts every dense row in Python (:874-878) and looks ranks up with
// list.index inside combinations(topN, 2) (:886-887): O(n^2 log n) time, n^2 floats of memory.
//
// On the sparse link matrix the same numbers need no dense row and no sort.  The reference's ranking of row f is
// "links descending, ties by ascending index" (Python's stable sort over enumerate(row)), link-less fragments
// (implicit zeros, the fragment itself included) therefore follow all linked ones in index order.  Hence
//   rank_f(x) = #{stored entries of row f that sort before x}  (+ x - #{stored columns < x}  if f and x share no link)
// and the top-N list of a row is N selections of "best entry that sorts after the previous pick", padded with the
// smallest absent indices.  One wavefront per fragment: N selections + 2 * C(N,2) rank queries, each a coalesced sweep
// of one sparse row (L2 resident: the queried rows are the fragment's strongest neighbours).
#include "hhx_common.h"

using namespace hhx;

namespace {

constexpr int RK_MAX_TOP = 64;

__device__ __forceinline__ i32 lb_cols(const i32 *__restrict__ cols, i32 b, i32 e, i32 x) {
    while (b < e) {
        const i32 m = b + ((e - b) >> 1);
        if (cols[m] < x) b = m + 1; else e = m;
    }
    return b;
}

// rank of fragment x in the link ranking of row a (0-based), wave-cooperative, result uniform
__device__ __forceinline__ i64 rank_in_row(i32 n, const i32 *__restrict__ ip, const i32 *__restrict__ ix, const float *__restrict__ dx,
                                           i32 a, i32 x) {
    const i32 b = ip[a], e = ip[a + 1];
    const i32 pos = lb_cols(ix, b, e, x);
    const bool found = pos < e && ix[pos] == x;
    const float v = found ? dx[pos] : 0.0f;
    i32 c = 0;
    for (i32 p = b + lane_id(); p < e; p += HHX_WAVE) {
        const float w = dx[p];
        c += (w > v) || (w == v && ix[p] < x);
    }
    i64 r = wave_sum_i32(c);
    if (v == 0.0f) r += (i64)x - (pos - b);            // link-less fragments with a smaller index (x itself may be stored with 0)
    else if (v < 0.0f) r += (i64)n - (e - b);          // every implicit zero beats a negative entry
    return r;
}

__global__ __launch_bounds__(256) void k_rank_sums(i32 n, const i32 *__restrict__ ip, const i32 *__restrict__ ix, const float *__restrict__ dx,
                                                   int topN, i64 *__restrict__ out) {
    __shared__ i32 s_top[4][RK_MAX_TOP];
    const int lane = lane_id(), wave = threadIdx.x / HHX_WAVE;
    i32 *top = s_top[wave];
    for (i32 f = blockIdx.x * 4 + wave; f < n; f += gridDim.x * 4) {
        const i32 b = ip[f], e = ip[f + 1];
        const int want = min(topN, (int)n);
        // ---- top list: stored entries by (value desc, column asc); then absent indices ascending.
        // A stored non-positive value would interleave with the implicit zeros; the link matrix holds counts (> 0).
        float pv = __int_as_float(0x7f800000);          // +inf
        i32 pc = -1;
        int got = 0;
        for (; got < want; ++got) {
            float bv = -1.0f; i32 bc = 0x7fffffff;
            for (i32 p = b + lane; p < e; p += HHX_WAVE) {
                const float w = dx[p]; const i32 c = ix[p];
                const bool after_prev = (w < pv) || (w == pv && c > pc);
                if (after_prev && w > 0.0f && (w > bv || (w == bv && c < bc))) { bv = w; bc = c; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_down(bv, o, HHX_WAVE);
                const i32 oc = __shfl_down(bc, o, HHX_WAVE);
                if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
            }
            bv = __shfl(bv, 0, HHX_WAVE); bc = __shfl(bc, 0, HHX_WAVE);
            if (bc == 0x7fffffff) break;                // stored positive entries exhausted
            top[got] = bc;                              // uniform value, written by every lane: each lane later reads its own write
            pv = bv; pc = bc;
        }
        if (got < want) {                               // pad with the smallest indices that hold no positive link
            i32 z = 0, pos = b;
            while (got < want && z < n) {
                while (pos < e && ix[pos] < z) ++pos;
                const bool linked = pos < e && ix[pos] == z && dx[pos] > 0.0f;
                if (!linked) { top[got] = z; ++got; }
                ++z;
            }
        }
        i64 sum = 0;
        for (int i = 0; i < got; ++i)
            for (int j = i + 1; j < got; ++j) {
                const i32 a = top[i], c = top[j];
                const i64 r1 = rank_in_row(n, ip, ix, dx, a, c);
                const i64 r2 = rank_in_row(n, ip, ix, dx, c, a);
                sum += r1 < r2 ? r1 : r2;
            }
        if (lane == 0) out[f] = sum;
    }
}

}  // namespace

extern "C" int hhx_rank_sums(const hhx_csr *m, int topN, i64 *rank_sum_host) {
    if (!m || !rank_sum_host) return fail("null pointer");
    if (m->n_rows != m->n_cols) return fail("hhx_rank_sums needs the square link matrix");
    if (topN < 0 || topN > RK_MAX_TOP) return fail("hhx_rank_sums: topN must be in [0, %d]", RK_MAX_TOP);
    const i32 n = m->n_rows;
    if (n == 0) return 0;
    DevBuf<i64> out;
    if (out.alloc((size_t)n)) return 1;
    const unsigned grid = (unsigned)std::min<i64>(((i64)n + 3) / 4, 256 * 32);
    { KTimer kt("rank_sums");
    k_rank_sums<<<grid, 256, 0, g_stream>>>(n, m->indptr.p, m->indices.p, m->data.p, topN, out.p); }
    HHX_LAUNCH_CHECK();
    HHX_HIP(hipMemcpyAsync(rank_sum_host, out.p, sizeof(i64) * (size_t)n, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
