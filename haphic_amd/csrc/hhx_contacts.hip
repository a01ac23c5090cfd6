// Scaffold-bin contact map of `haphic plot` (second half): HapHiC_plot.py parse_pairs :153-202 and
// parse_bam :205-245.  Per read pair the reference (a) drops it unless both contigs belong to a drawn scaffold
// (ctg_set, :184 / :228), (b) converts each end with convert_group_bin_id :155-168 — the contig's alignment bin
// (pos - 1) // bin_size selects a short list of closed ranges of the raw contig (the pieces of the scaffold bins that
// AGP line covers), the first range holding the position names (scaffold, scaffold bin), a scaffold that is not drawn
// ends the pair (None), a position whose alignment bin is not in the table is an error (KeyError -> Exception) — and
// (c) adds one to contact_matrix[bin(ref), bin(mref)] :200 / :243.
//
// The same binning shape as the cluster ingest, with a dense matrix instead of a dict as the accumulator.  One kernel:
//   * every lane converts PER read pairs (32 B of id / position loads per pair issued together; the range tables are a
//     few hundred KB and stay in L2),
//   * the cells of a 2048-pair tile are first counted in an LDS hash table (Hi-C puts a third of all pairs on the
//     diagonal cells, so the tile collapses the hot cells before they reach the fabric: ds_cmpst_b64 insert, ds_add count),
//   * one 64-bit global atomic add per distinct cell of the tile.  Integer counts: order free, bit reproducible.
// HBM-bound by the 16 B per pair of the four input streams once the matrix (8 B x n_bins^2) sits in the Infinity Cache.
#include "hhx_common.h"

using namespace hhx;

struct hhx_contact_map {
    i32 n_ctg = 0, bin_size = 0, n_bins = 0;
    DevBuf<unsigned char> in_set;
    DevBuf<i64> aln_ptr;
    DevBuf<i32> list_ptr, seg_lo, seg_hi, seg_bin;
    DevBuf<unsigned long long> cells, bad;
};

namespace {

constexpr int CM_T = 256, CM_PER = 8, CM_SLOTS = 4096;
constexpr unsigned long long CM_EMPTY = ~0ull;

struct CmTables {
    i32 n_ctg, bin_size;
    const unsigned char *in_set;
    const i64 *aln_ptr;
    const i32 *list_ptr, *seg_lo, *seg_hi, *seg_bin;
};

// convert_group_bin_id :155-168.  >= 0: total bin; -1: None (no range holds the position, or its scaffold is not drawn);
// -2: the KeyError of ctg_aln_dict[ctg][(pos - 1) // bin_size]
__device__ __forceinline__ i32 total_bin_of(const CmTables &t, i32 ctg, i64 pos) {
    const i64 q = pos - 1;
    if (q < 0) return -2;                                    // Python floor division: alignment bin -1, never a key
    const i64 a0 = t.aln_ptr[ctg], a1 = t.aln_ptr[ctg + 1];
    const i64 slot = a0 + q / t.bin_size;
    if (slot >= a1) return -2;
    const i32 b = t.list_ptr[slot], e = t.list_ptr[slot + 1];
    if (b == e) return -2;                                   // an alignment bin no AGP line touches is not a key either
    for (i32 s = b; s < e; ++s)
        if (pos >= t.seg_lo[s] && pos <= t.seg_hi[s]) return t.seg_bin[s];       // first range holding pos; -1: group not in group_list
    return -1;
}

__global__ __launch_bounds__(CM_T) void k_bin_contacts(CmTables t, i64 n, const i32 *__restrict__ id1, const i32 *__restrict__ pos1,
                                                       const i32 *__restrict__ id2, const i32 *__restrict__ pos2, i32 pos_offset, i64 n_bins,
                                                       unsigned long long *__restrict__ cells, unsigned long long *__restrict__ bad) {
    __shared__ unsigned long long s_key[CM_SLOTS];
    __shared__ u32 s_cnt[CM_SLOTS];
    const int tid = threadIdx.x;
    for (int s = tid; s < CM_SLOTS; s += CM_T) { s_key[s] = CM_EMPTY; s_cnt[s] = 0; }
    __syncthreads();
    const i64 tile = (i64)CM_T * CM_PER;
    for (i64 base = (i64)blockIdx.x * tile; base < n; base += (i64)gridDim.x * tile) {
        i32 r[CM_PER], m[CM_PER], p[CM_PER], q[CM_PER];
#pragma unroll
        for (int u = 0; u < CM_PER; ++u) {
            const i64 k = base + (i64)u * CM_T + tid;
            const bool in = k < n;
            r[u] = in ? id1[k] : -1; m[u] = in ? id2[k] : -1; p[u] = in ? pos1[k] : 0; q[u] = in ? pos2[k] : 0;
        }
#pragma unroll
        for (int u = 0; u < CM_PER; ++u) {
            const i32 a = r[u], b = m[u];
            if (a < 0 || b < 0 || a >= t.n_ctg || b >= t.n_ctg || !t.in_set[a] || !t.in_set[b]) continue;      // :184 / :228
            const unsigned long long at = 2ull * (unsigned long long)(base + (i64)u * CM_T + tid);
            const i32 x = total_bin_of(t, a, (i64)p[u] + pos_offset);
            if (x == -2) { atomicMin(bad, at); continue; }
            if (x < 0) continue;                              // the mate is not looked at (:187-189)
            const i32 y = total_bin_of(t, b, (i64)q[u] + pos_offset);
            if (y == -2) { atomicMin(bad, at + 1); continue; }
            if (y < 0) continue;
            const unsigned long long cell = (unsigned long long)x * (unsigned long long)n_bins + (unsigned long long)y;
            u32 h = (u32)((cell * 0x9e3779b97f4a7c15ull) >> 52);
            for (;;) {
                const unsigned long long seen = atomicCAS(&s_key[h], CM_EMPTY, cell);
                if (seen == CM_EMPTY || seen == cell) { atomicAdd(&s_cnt[h], 1u); break; }
                h = (h + 1) & (CM_SLOTS - 1);                // at most 2048 distinct cells per tile in 4096 slots
            }
        }
        __syncthreads();
        for (int s = tid; s < CM_SLOTS; s += CM_T) {
            const unsigned long long cell = s_key[s];
            if (cell != CM_EMPTY) {
                atomicAdd(&cells[cell], (unsigned long long)s_cnt[s]);
                s_key[s] = CM_EMPTY; s_cnt[s] = 0;
            }
        }
        __syncthreads();
    }
}

template <class T>
int to_device(DevBuf<T> &d, const T *h, size_t n) {
    if (d.alloc(n)) return 1;
    if (n) HHX_HIP(hipMemcpyAsync(d.p, h, n * sizeof(T), hipMemcpyHostToDevice, g_stream));
    return 0;
}

}  // namespace

extern "C" int hhx_contact_map_create(i32 n_ctg, const uint8_t *in_set, const i64 *aln_ptr, const i32 *list_ptr, i64 n_list, const i32 *seg_lo,
                                      const i32 *seg_hi, const i32 *seg_bin, i32 bin_size, i32 n_total_bins, hhx_contact_map **out) {
    if (!out) return fail("null pointer");
    if (n_ctg < 0 || n_list < 0 || n_list > INT32_MAX || bin_size <= 0 || n_total_bins < 0) return fail("hhx_contact_map_create: bad arguments");
    if (n_ctg && (!in_set || !aln_ptr || !list_ptr)) return fail("null pointer");
    if (n_list && (!seg_lo || !seg_hi || !seg_bin)) return fail("null pointer");
    const i64 n_slots = n_ctg ? aln_ptr[n_ctg] : 0;
    if (n_slots < 0 || (n_ctg && aln_ptr[0] != 0) || (n_slots && list_ptr[n_slots] != (i32)n_list)) return fail("hhx_contact_map_create: inconsistent range tables");
    for (i64 s = 0; s < n_list; ++s)
        if (seg_bin[s] >= n_total_bins) return fail("hhx_contact_map_create: range %lld names bin %d of %d", (long long)s, seg_bin[s], n_total_bins);
    hhx_contact_map *m = new hhx_contact_map();
    m->n_ctg = n_ctg; m->bin_size = bin_size; m->n_bins = n_total_bins;
    const size_t n_cells = (size_t)n_total_bins * (size_t)n_total_bins;
    static const i64 zero64 = 0;
    static const i32 zero32 = 0;
    int rc = to_device(m->in_set, in_set, (size_t)n_ctg) || to_device(m->aln_ptr, n_ctg ? aln_ptr : &zero64, (size_t)n_ctg + 1) ||
             to_device(m->list_ptr, n_slots || n_ctg ? list_ptr : &zero32, (size_t)n_slots + 1) || to_device(m->seg_lo, seg_lo, (size_t)n_list) ||
             to_device(m->seg_hi, seg_hi, (size_t)n_list) || to_device(m->seg_bin, seg_bin, (size_t)n_list) || m->cells.alloc(n_cells) || m->bad.alloc(1);
    if (!rc && hipMemsetAsync(m->cells.p, 0, sizeof(unsigned long long) * (n_cells ? n_cells : 1), g_stream) != hipSuccess) rc = fail("memset failed");
    if (!rc && hipStreamSynchronize(g_stream) != hipSuccess) rc = fail("hhx_contact_map_create: upload failed");
    if (rc) { delete m; return 1; }
    *out = m;
    return 0;
}

extern "C" int hhx_contact_map_push(hhx_contact_map *m, i64 n_pairs, const i32 *id1, const i32 *pos1, const i32 *id2, const i32 *pos2, int on_device,
                                    i32 pos_offset, i64 *bad) {
    if (bad) *bad = -1;
    if (!m) return fail("null handle");
    if (n_pairs < 0) return fail("hhx_contact_map_push: negative count");
    if (n_pairs == 0) return 0;
    if (!id1 || !pos1 || !id2 || !pos2) return fail("null pointer");
    DevBuf<i32> d[4];
    const i32 *src[4] = {id1, pos1, id2, pos2};
    if (!on_device)
        for (int k = 0; k < 4; ++k) {
            if (d[k].alloc((size_t)n_pairs)) return 1;
            HHX_HIP(hipMemcpyAsync(d[k].p, src[k], sizeof(i32) * (size_t)n_pairs, hipMemcpyHostToDevice, g_stream));
            src[k] = d[k].p;
        }
    HHX_HIP(hipMemsetAsync(m->bad.p, 0xff, sizeof(unsigned long long), g_stream));
    const CmTables t{m->n_ctg, m->bin_size, m->in_set.p, m->aln_ptr.p, m->list_ptr.p, m->seg_lo.p, m->seg_hi.p, m->seg_bin.p};
    const i64 tiles = (n_pairs + (i64)CM_T * CM_PER - 1) / ((i64)CM_T * CM_PER);
    {
        KTimer kt("bin_contacts");
        k_bin_contacts<<<(unsigned)std::max<i64>(1, std::min<i64>(tiles, 256 * 8)), CM_T, 0, g_stream>>>(t, n_pairs, src[0], src[1], src[2], src[3], pos_offset,
                                                                                                    (i64)m->n_bins, m->cells.p, m->bad.p);
    }
    HHX_LAUNCH_CHECK();
    prof_count("contact_pairs", n_pairs);
    unsigned long long b = 0;
    HHX_HIP(hipMemcpyAsync(&b, m->bad.p, sizeof b, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    if (bad) *bad = b == ~0ull ? -1 : (i64)b;
    return 0;
}

extern "C" int hhx_contact_map_fetch(hhx_contact_map *m, i64 *cells_host) {
    if (!m || !cells_host) return fail("null pointer");
    const size_t n_cells = (size_t)m->n_bins * (size_t)m->n_bins;
    if (n_cells) HHX_HIP(hipMemcpyAsync(cells_host, m->cells.p, sizeof(i64) * n_cells, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int hhx_contact_map_device(hhx_contact_map *m, void **cells_dev, i32 *n_bins) {
    if (!m || !cells_dev) return fail("null pointer");
    *cells_dev = m->cells.p;
    if (n_bins) *n_bins = m->n_bins;
    return 0;
}

extern "C" int hhx_contact_map_destroy(hhx_contact_map *m) {
    delete m;
    return 0;
}
