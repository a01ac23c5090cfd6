// Side products of the ingest loop that need the read pairs themselves, not only counts:
//   clm_dict      update_clm_dict :395-401 — four orientation distances per read pair, per contig pair, stream order
//   ctg_coord_dict record_coord_pairs :454-471 — the first max_read_pairs (coord_i, coord_j) per contig pair
// Both are "group the pairs by contig pair, keep stream order inside a group".  The pairs counted in
// full_link_dict are compacted (stably, so they stay in stream order) into (key, xi << 32 | xj) records at push
// time; at fetch time one STABLE radix sort by key (hhx_sort.h: hand-written LSD passes, ballot-ranked so that equal
// keys keep their stream order) groups them, the insertion-ordered key table is sorted the same way to pair every
// group with its dict position, and one wavefront per contig pair writes its distances.
#include <algorithm>
#include <cstring>

#include "hhx_ingest.h"
#include "hhx_filesink.h"
#include "hhx_sort.h"

using namespace hhx;

namespace {

constexpr int SD_T = 256, SD_ITEMS = 4, SD_TILE = SD_T * SD_ITEMS;

template <bool COMBINED>
__device__ __forceinline__ bool side_rec(const DevTables &t, i32 r, i32 m, i64 p1, i64 p2, u64 &key, u64 &xy) {
    u64 rec;
    if (!map_pair<COMBINED>(t, 0, r, m, p1, p2, rec, &xy)) return false;
    if (!(rec & FULL_BIT)) return false;
    key = rec & KEY_MASK;
    return true;
}

template <bool COMBINED, class POS>
__global__ __launch_bounds__(SD_T) void k_side_count(i64 n, const i32 *__restrict__ id1, const POS *__restrict__ pos1,
                                                     const i32 *__restrict__ id2, const POS *__restrict__ pos2, DevTables t, i64 *__restrict__ tile_cnt) {
    __shared__ i32 wsum[SD_T / HHX_WAVE];
    const i64 tile = blockIdx.x;
    const i64 base = tile * SD_TILE + (i64)threadIdx.x * SD_ITEMS;
    i32 c = 0;
#pragma unroll
    for (int k = 0; k < SD_ITEMS; ++k) {
        u64 key, xy;
        if (base + k < n && side_rec<COMBINED>(t, id1[base + k], id2[base + k], pos1[base + k], pos2[base + k], key, xy)) ++c;
    }
    c = wave_sum_i32(c);
    if (lane_id() == 0) wsum[threadIdx.x / HHX_WAVE] = c;
    __syncthreads();
    if (threadIdx.x == 0) { i64 s = 0; for (int w = 0; w < SD_T / HHX_WAVE; ++w) s += wsum[w]; tile_cnt[tile] = s; }
}

// stable: thread t owns SD_ITEMS consecutive pairs, positions = tile offset + exclusive scan of the per-thread counts
template <bool COMBINED, class POS>
__global__ __launch_bounds__(SD_T) void k_side_write(i64 n, const i32 *__restrict__ id1, const POS *__restrict__ pos1,
                                                     const i32 *__restrict__ id2, const POS *__restrict__ pos2, DevTables t,
                                                     const i64 *__restrict__ tile_off, u64 *__restrict__ okey, u64 *__restrict__ oxy) {
    __shared__ i32 wsum[SD_T / HHX_WAVE];
    const i64 tile = blockIdx.x;
    const i64 base = tile * SD_TILE + (i64)threadIdx.x * SD_ITEMS;
    u64 key[SD_ITEMS], xy[SD_ITEMS];
    bool ok[SD_ITEMS];
    i32 c = 0;
#pragma unroll
    for (int k = 0; k < SD_ITEMS; ++k) {
        ok[k] = base + k < n && side_rec<COMBINED>(t, id1[base + k], id2[base + k], pos1[base + k], pos2[base + k], key[k], xy[k]);
        c += ok[k];
    }
    i32 incl = c;
#pragma unroll
    for (int o = 1; o < HHX_WAVE; o <<= 1) {
        const i32 v = __shfl_up(incl, o, HHX_WAVE);
        if (lane_id() >= o) incl += v;
    }
    if (lane_id() == HHX_WAVE - 1) wsum[threadIdx.x / HHX_WAVE] = incl;
    __syncthreads();
    i64 o = tile_off[tile] + incl - c;
    for (int w = 0; w < (int)(threadIdx.x / HHX_WAVE); ++w) o += wsum[w];
#pragma unroll
    for (int k = 0; k < SD_ITEMS; ++k)
        if (ok[k]) { okey[o] = key[k]; oxy[o] = xy[k]; ++o; }
}

__global__ __launch_bounds__(256) void k_table_keys(i64 n, const i32 *__restrict__ fi, const i32 *__restrict__ fj, u64 *__restrict__ key, u64 *__restrict__ rnk) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
        key[r] = ((u64)(u32)fi[r] << ID_BITS) | (u64)(u32)fj[r];
        rnk[r] = (u64)r;
    }
}
// group boundaries of the sorted records
__global__ __launch_bounds__(256) void k_boundary_flags(i64 n, const u64 *__restrict__ sk, i64 *__restrict__ flag) {
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (i64)gridDim.x * blockDim.x) flag[p] = (p == 0 || sk[p] != sk[p - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_group_starts(i64 n, const u64 *__restrict__ sk, const i64 *__restrict__ gidx, i64 *__restrict__ gstart) {
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (i64)gridDim.x * blockDim.x)
        if (p == 0 || sk[p] != sk[p - 1]) gstart[gidx[p]] = p;
}
// group g (g-th smallest key) belongs to dict position r = srank[g]; its count and capped count in dict order
__global__ __launch_bounds__(256) void k_group_counts(i64 n_groups, const u64 *__restrict__ stk, const u64 *__restrict__ gkey_src, const i64 *__restrict__ gstart,
                                                      const u64 *__restrict__ srank, i64 max_pairs, i64 *__restrict__ cnt_by_r, i64 *__restrict__ cap_by_r,
                                                      unsigned int *__restrict__ mismatch) {
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (i64)gridDim.x * blockDim.x) {
        if (gkey_src[gstart[g]] != stk[g]) atomicExch(mismatch, 1u);
        const i64 c = gstart[g + 1] - gstart[g];
        const i64 r = (i64)srank[g];
        cnt_by_r[r] = c;
        cap_by_r[r] = c < max_pairs ? c : max_pairs;
    }
}
// one wavefront per contig pair: distances of update_clm_dict :395-401 (0-based coordinates) and the first coordinates
__global__ __launch_bounds__(256) void k_emit_pairs(i64 n_groups, const u64 *__restrict__ stk, const i64 *__restrict__ gstart, const u64 *__restrict__ srank,
                                                    const u64 *__restrict__ sxy, const UnitInfo *__restrict__ ctg, const i64 *__restrict__ clm_off,
                                                    const i64 *__restrict__ crd_off, i64 max_pairs, i64 *__restrict__ clm, i64 *__restrict__ crd) {
    const int lane = lane_id();
    for (i64 g = (i64)blockIdx.x * 4 + threadIdx.x / HHX_WAVE; g < n_groups; g += (i64)gridDim.x * 4) {
        const i64 b = gstart[g], e = gstart[g + 1], r = (i64)srank[g];
        const u64 key = stk[g];
        const i64 li = ctg[key >> ID_BITS].lenf & LEN_MASK, lj = ctg[key & ID_MASK].lenf & LEN_MASK;
        const i64 co = clm_off[r], ko = crd_off[r];
        for (i64 p = b + lane; p < e; p += HHX_WAVE) {
            const u64 xy = sxy[p];
            const i64 xi = (i64)(xy >> 32), xj = (i64)(xy & 0xffffffffu);
            const i64 a = xi - 1, c = xj - 1, t = p - b;
            i64 *d = clm + 4 * (co + t);
            d[0] = li - a + c; d[1] = li - a + lj - c; d[2] = a + c; d[3] = a + lj - c;
            if (t < max_pairs) { crd[2 * (ko + t)] = xi; crd[2 * (ko + t) + 1] = xj; }
        }
    }
}

inline unsigned grid_for(u64 n) {
    u64 b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)b;
}

__global__ __launch_bounds__(256) void k_iota_u64(i64 n, u64 *__restrict__ v) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) v[i] = (u64)i;
}
// one wavefront per contig pair: stream position (index among the kept pairs) of the first pair that fell into each
// head/tail quadrant — HT_link_dict's keys enter the dict in that order (update_HT_link_dict :404-416)
__global__ __launch_bounds__(256) void k_ht_first(i64 n_groups, const u64 *__restrict__ stk, const i64 *__restrict__ gstart, const u64 *__restrict__ srank,
                                                  const u64 *__restrict__ sidx, const u64 *__restrict__ xy, const UnitInfo *__restrict__ ctg,
                                                  i64 *__restrict__ first) {
    const int lane = lane_id();
    for (i64 g = (i64)blockIdx.x * 4 + threadIdx.x / HHX_WAVE; g < n_groups; g += (i64)gridDim.x * 4) {
        const i64 b = gstart[g], e = gstart[g + 1], r = (i64)srank[g];
        const u64 key = stk[g];
        const i64 li = ctg[key >> ID_BITS].lenf & LEN_MASK, lj = ctg[key & ID_MASK].lenf & LEN_MASK;
        long long m0 = INT64_MAX, m1 = INT64_MAX, m2 = INT64_MAX, m3 = INT64_MAX;
        for (i64 p = b + lane; p < e; p += HHX_WAVE) {
            const long long i = (long long)sidx[p];
            const u64 v = xy[i];
            const i64 xi = (i64)(v >> 32), xj = (i64)(v & 0xffffffffu);
            const int q = (xi * 2 > li ? 2 : 0) + (xj * 2 > lj ? 1 : 0);          // :408 coord * 2 > ctg_len -> '_T'
            if (q == 0) m0 = min(m0, i); else if (q == 1) m1 = min(m1, i); else if (q == 2) m2 = min(m2, i); else m3 = min(m3, i);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m0 = min(m0, __shfl_down(m0, o, HHX_WAVE)); m1 = min(m1, __shfl_down(m1, o, HHX_WAVE));
            m2 = min(m2, __shfl_down(m2, o, HHX_WAVE)); m3 = min(m3, __shfl_down(m3, o, HHX_WAVE));
        }
        if (lane == 0) { first[4 * r] = m0; first[4 * r + 1] = m1; first[4 * r + 2] = m2; first[4 * r + 3] = m3; }
    }
}

int sort_pairs_u64(const u64 *kin, u64 *kout, const u64 *vin, u64 *vout, i64 n) {
    return stable_sort_pairs_u64(kin, kout, vin, vout, n, 2 * ID_BITS);      // hhx_sort.h: hand-written stable LSD radix sort
}

}  // namespace

template <class POS>
int hhx_side_records_push(hhx_ingest *h, i64 n_pairs, const i32 *id1, const POS *pos1, const i32 *id2, const POS *pos2) {
    const i64 n_tiles = (n_pairs + SD_TILE - 1) / SD_TILE;
    DevBuf<i64> cnt, off;
    if (cnt.alloc((size_t)n_tiles + 1) || off.alloc((size_t)n_tiles + 2)) return 1;
    if (h->combined) k_side_count<true, POS><<<(unsigned)n_tiles, SD_T, 0, g_stream>>>(n_pairs, id1, pos1, id2, pos2, h->t, cnt.p);
    else k_side_count<false, POS><<<(unsigned)n_tiles, SD_T, 0, g_stream>>>(n_pairs, id1, pos1, id2, pos2, h->t, cnt.p);
    HHX_LAUNCH_CHECK();
    i64 total = 0;
    HHX_TRY(exclusive_scan_i64(cnt.p, off.p, n_tiles, &total));
    h->side_key.emplace_back();
    h->side_xy.emplace_back();
    if (h->side_key.back().alloc((size_t)total) || h->side_xy.back().alloc((size_t)total)) return 1;
    if (total) {
        if (h->combined) k_side_write<true, POS><<<(unsigned)n_tiles, SD_T, 0, g_stream>>>(n_pairs, id1, pos1, id2, pos2, h->t, off.p, h->side_key.back().p, h->side_xy.back().p);
        else k_side_write<false, POS><<<(unsigned)n_tiles, SD_T, 0, g_stream>>>(n_pairs, id1, pos1, id2, pos2, h->t, off.p, h->side_key.back().p, h->side_xy.back().p);
        HHX_LAUNCH_CHECK();
    }
    HHX_HIP(hipStreamSynchronize(g_stream));
    h->n_side += total;
    return 0;
}

template int hhx_side_records_push<i32>(hhx_ingest *, i64, const i32 *, const i32 *, const i32 *, const i32 *);
template int hhx_side_records_push<i64>(hhx_ingest *, i64, const i32 *, const i64 *, const i32 *, const i64 *);

int hhx_ingest_ordered_full_device(hhx_ingest *h, const i32 **fi, const i32 **fj);   // hhx_ingest.hip

// The kept read pairs grouped by contig pair: sorted (key, xy) records (stream order inside a group), the boundaries of the groups,
// and for the g-th smallest key the position r = srank[g] of that contig pair in full_link_dict (dict insertion order).
struct PairGroups {
    i64 K = 0, N = 0;
    DevBuf<u64> skey, sxy, stk, srank;
    DevBuf<i64> gstart;                    // [K + 1]
};

static int group_pairs(hhx_ingest *h, PairGroups &G, const char *who) {
    if (h->pairs_dropped) return fail("%s: the kept read pairs were released after paired_links.clm was written", who);
    const i32 *fi = nullptr, *fj = nullptr;
    HHX_TRY(hhx_ingest_ordered_full_device(h, &fi, &fj));
    const i64 K = h->n_full, N = h->n_side;
    G.K = K; G.N = N;
    if (K == 0) return 0;
    // concatenate the pushes (stream order), stable sort by key
    DevBuf<u64> key, xy, tkey, trnk;
    if (key.alloc((size_t)N) || xy.alloc((size_t)N) || G.skey.alloc((size_t)N) || G.sxy.alloc((size_t)N) || tkey.alloc((size_t)K) || trnk.alloc((size_t)K) ||
        G.stk.alloc((size_t)K) || G.srank.alloc((size_t)K)) return 1;
    i64 o = 0;
    for (size_t b = 0; b < h->side_key.size(); ++b) {
        const i64 nb = (i64)h->side_key[b].n;
        if (nb) {
            HHX_HIP(hipMemcpyAsync(key.p + o, h->side_key[b].p, 8 * (size_t)nb, hipMemcpyDeviceToDevice, g_stream));
            HHX_HIP(hipMemcpyAsync(xy.p + o, h->side_xy[b].p, 8 * (size_t)nb, hipMemcpyDeviceToDevice, g_stream));
        }
        o += nb;
    }
    HHX_TRY(sort_pairs_u64(key.p, G.skey.p, xy.p, G.sxy.p, N));
    key.release(); xy.release();
    k_table_keys<<<grid_for((u64)K), 256, 0, g_stream>>>(K, fi, fj, tkey.p, trnk.p);
    HHX_LAUNCH_CHECK();
    HHX_TRY(sort_pairs_u64(tkey.p, G.stk.p, trnk.p, G.srank.p, K));
    // groups of the sorted records <-> sorted table keys
    DevBuf<i64> flag, gidx;
    if (flag.alloc((size_t)N + 1) || gidx.alloc((size_t)N + 2) || G.gstart.alloc((size_t)K + 2)) return 1;
    k_boundary_flags<<<grid_for((u64)N), 256, 0, g_stream>>>(N, G.skey.p, flag.p);
    HHX_LAUNCH_CHECK();
    i64 n_groups = 0;
    HHX_TRY(exclusive_scan_i64(flag.p, gidx.p, N, &n_groups));
    if (n_groups != K) return fail("%s: %lld contig pairs in the records, %lld in the table", who, (long long)n_groups, (long long)K);
    k_group_starts<<<grid_for((u64)N), 256, 0, g_stream>>>(N, G.skey.p, gidx.p, G.gstart.p);
    HHX_HIP(hipMemcpyAsync(G.gstart.p + K, &N, sizeof(i64), hipMemcpyHostToDevice, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int hhx_ingest_fetch_pairs(hhx_ingest *h, i64 max_read_pairs, i64 *clm_ptr, i64 *clm, i64 *crd_ptr, i64 *crd) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (!h->keep_pairs) return fail("hhx_ingest_fetch_pairs: the handle was not created with hhx_ingest_keep_pairs");
    if (max_read_pairs < 0) max_read_pairs = 0;
    if (clm_ptr) clm_ptr[0] = 0;
    if (crd_ptr) crd_ptr[0] = 0;
    PairGroups G;
    HHX_TRY(group_pairs(h, G, "hhx_ingest_fetch_pairs"));
    const i64 K = G.K, N = G.N;
    if (K == 0) return 0;
    DevBuf<i64> cnt_r, cap_r, clm_off, crd_off;
    DevBuf<unsigned int> mismatch;
    if (cnt_r.alloc((size_t)K + 1) || cap_r.alloc((size_t)K + 1) || clm_off.alloc((size_t)K + 2) || crd_off.alloc((size_t)K + 2) || mismatch.alloc(1)) return 1;
    HHX_HIP(hipMemsetAsync(mismatch.p, 0, sizeof(unsigned int), g_stream));
    k_group_counts<<<grid_for((u64)K), 256, 0, g_stream>>>(K, G.stk.p, G.skey.p, G.gstart.p, G.srank.p, max_read_pairs, cnt_r.p, cap_r.p, mismatch.p);
    HHX_LAUNCH_CHECK();
    i64 clm_total = 0, crd_total = 0;
    HHX_TRY(exclusive_scan_i64(cnt_r.p, clm_off.p, K, &clm_total));
    HHX_TRY(exclusive_scan_i64(cap_r.p, crd_off.p, K, &crd_total));
    unsigned int mm = 0;
    HHX_HIP(hipMemcpyAsync(&mm, mismatch.p, sizeof mm, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    if (mm || clm_total != N) return fail("hhx_ingest_fetch_pairs: records and table disagree");
    DevBuf<i64> d_clm, d_crd;
    if (d_clm.alloc((size_t)N * 4) || d_crd.alloc((size_t)crd_total * 2 + 2)) return 1;
    k_emit_pairs<<<grid_for((u64)K * 64), 256, 0, g_stream>>>(K, G.stk.p, G.gstart.p, G.srank.p, G.sxy.p, h->t.ctg, clm_off.p, crd_off.p, max_read_pairs, d_clm.p, d_crd.p);
    HHX_LAUNCH_CHECK();
    if (clm_ptr) HHX_HIP(hipMemcpyAsync(clm_ptr, clm_off.p, sizeof(i64) * ((size_t)K + 1), hipMemcpyDeviceToHost, g_stream));
    if (crd_ptr) HHX_HIP(hipMemcpyAsync(crd_ptr, crd_off.p, sizeof(i64) * ((size_t)K + 1), hipMemcpyDeviceToHost, g_stream));
    if (clm && N) HHX_HIP(hipMemcpyAsync(clm, d_clm.p, sizeof(i64) * (size_t)N * 4, hipMemcpyDeviceToHost, g_stream));
    if (crd && crd_total) HHX_HIP(hipMemcpyAsync(crd, d_crd.p, sizeof(i64) * (size_t)crd_total * 2, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

// HT_link_dict's insertion order (update_HT_link_dict :404-416 inside the loops :1646 / :1746): for every contig pair of
// full_link_dict, in dict order, the stream position of the first read pair of each quadrant [HH, HT, TH, TT]
// (INT64_MAX: that quadrant never occurred).  Positions count the pairs that entered full_link_dict, in stream order, so
// sorting the non-empty (pair, quadrant) entries by them gives the dict order.  Same grouping as hhx_ingest_fetch_pairs
// (stable sort of the kept pairs by key), with the stream position carried as the sorted value.
static int ht_first_device(hhx_ingest *h, DevBuf<i64> &d_first, const char *who) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (!h->keep_pairs) return fail("%s: the handle was not created with hhx_ingest_keep_pairs", who);
    if (h->pairs_dropped) return fail("%s: the kept read pairs were released after paired_links.clm was written", who);
    const i32 *fi = nullptr, *fj = nullptr;
    HHX_TRY(hhx_ingest_ordered_full_device(h, &fi, &fj));
    const i64 K = h->n_full, N = h->n_side;
    if (K == 0) return 0;
    DevBuf<u64> key, xy, skey, idx, sidx, tkey, trnk, stk, srank;
    if (key.alloc((size_t)N) || xy.alloc((size_t)N) || skey.alloc((size_t)N) || idx.alloc((size_t)N) || sidx.alloc((size_t)N) || tkey.alloc((size_t)K) ||
        trnk.alloc((size_t)K) || stk.alloc((size_t)K) || srank.alloc((size_t)K)) return 1;
    i64 o = 0;
    for (size_t b = 0; b < h->side_key.size(); ++b) {
        const i64 nb = (i64)h->side_key[b].n;
        if (nb) {
            HHX_HIP(hipMemcpyAsync(key.p + o, h->side_key[b].p, 8 * (size_t)nb, hipMemcpyDeviceToDevice, g_stream));
            HHX_HIP(hipMemcpyAsync(xy.p + o, h->side_xy[b].p, 8 * (size_t)nb, hipMemcpyDeviceToDevice, g_stream));
        }
        o += nb;
    }
    k_iota_u64<<<grid_for((u64)N), 256, 0, g_stream>>>(N, idx.p);
    HHX_LAUNCH_CHECK();
    HHX_TRY(sort_pairs_u64(key.p, skey.p, idx.p, sidx.p, N));
    k_table_keys<<<grid_for((u64)K), 256, 0, g_stream>>>(K, fi, fj, tkey.p, trnk.p);
    HHX_LAUNCH_CHECK();
    HHX_TRY(sort_pairs_u64(tkey.p, stk.p, trnk.p, srank.p, K));
    DevBuf<i64> flag, gidx, gstart;
    if (flag.alloc((size_t)N + 1) || gidx.alloc((size_t)N + 2) || gstart.alloc((size_t)K + 2) || d_first.alloc((size_t)K * 4)) return 1;
    k_boundary_flags<<<grid_for((u64)N), 256, 0, g_stream>>>(N, skey.p, flag.p);
    HHX_LAUNCH_CHECK();
    i64 n_groups = 0;
    HHX_TRY(exclusive_scan_i64(flag.p, gidx.p, N, &n_groups));
    if (n_groups != K) return fail("%s: %lld contig pairs in the records, %lld in the table", who, (long long)n_groups, (long long)K);
    k_group_starts<<<grid_for((u64)N), 256, 0, g_stream>>>(N, skey.p, gidx.p, gstart.p);
    HHX_HIP(hipMemcpyAsync(gstart.p + K, &N, sizeof(i64), hipMemcpyHostToDevice, g_stream));
    k_ht_first<<<grid_for((u64)K * 64), 256, 0, g_stream>>>(K, stk.p, gstart.p, srank.p, sidx.p, xy.p, h->t.ctg, d_first.p);
    HHX_LAUNCH_CHECK();
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int hhx_ingest_fetch_ht_order(hhx_ingest *h, i64 *first) {
    DevBuf<i64> d_first;
    HHX_TRY(ht_first_device(h, d_first, "hhx_ingest_fetch_ht_order"));
    if (h->n_full == 0) return 0;
    if (!first) return fail("hhx_ingest_fetch_ht_order: null output");
    HHX_HIP(hipMemcpyAsync(first, d_first.p, sizeof(i64) * (size_t)h->n_full * 4, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

// HT_link_dict as items in insertion order: entry e = (contig pair k, quadrant q) with a non-zero count, sorted by the stream
// position of its first read pair; name ids are 2 * contig + (1 if that end is the tail '_T'), i.e. indices into
// [c0_H, c0_T, c1_H, c1_T, ...].  name_i == nullptr: only the number of items.
namespace {
__global__ __launch_bounds__(256) void k_ht_flag(i64 n, const i64 *__restrict__ ht, i64 *__restrict__ flag) {
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (i64)gridDim.x * blockDim.x) flag[e] = ht[e] != 0 ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_ht_compact(i64 n, const i64 *__restrict__ ht, const i64 *__restrict__ pos, const i64 *__restrict__ first,
                                                    u64 *__restrict__ key, u64 *__restrict__ val) {
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (i64)gridDim.x * blockDim.x)
        if (ht[e] != 0) { key[pos[e]] = (u64)first[e]; val[pos[e]] = (u64)e; }
}
__global__ __launch_bounds__(256) void k_ht_items(i64 n, const u64 *__restrict__ sval, const i32 *__restrict__ fi, const i32 *__restrict__ fj,
                                                  const i64 *__restrict__ ht, i32 *__restrict__ ni, i32 *__restrict__ nj, i64 *__restrict__ cnt) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i64 e = (i64)sval[t], k = e >> 2;
        const int q = (int)(e & 3);
        ni[t] = 2 * fi[k] + (q >> 1);
        nj[t] = 2 * fj[k] + (q & 1);
        cnt[t] = ht[e];
    }
}
}  // namespace

extern "C" int hhx_ingest_fetch_ht_items(hhx_ingest *h, int64_t *n_items, int32_t *name_i, int32_t *name_j, int64_t *count) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (!n_items) return fail("hhx_ingest_fetch_ht_items: null pointer");
    const i32 *fi = nullptr, *fj = nullptr;
    HHX_TRY(hhx_ingest_ordered_full_device(h, &fi, &fj));
    const i64 K = h->n_full, E = 4 * K;
    *n_items = 0;
    if (K == 0) return 0;
    const i64 *ht = h->ordered.ht.p;
    DevBuf<i64> flag, pos;
    if (flag.alloc((size_t)E + 1) || pos.alloc((size_t)E + 2)) return 1;
    k_ht_flag<<<grid_for((u64)E), 256, 0, g_stream>>>(E, ht, flag.p);
    HHX_LAUNCH_CHECK();
    i64 n = 0;
    HHX_TRY(exclusive_scan_i64(flag.p, pos.p, E, &n));
    *n_items = n;
    if (!name_i || !name_j || !count || n == 0) return 0;
    flag.release();
    DevBuf<i64> d_first;
    HHX_TRY(ht_first_device(h, d_first, "hhx_ingest_fetch_ht_items"));
    DevBuf<u64> key, val, skey, sval;
    if (key.alloc((size_t)n) || val.alloc((size_t)n) || skey.alloc((size_t)n) || sval.alloc((size_t)n)) return 1;
    k_ht_compact<<<grid_for((u64)E), 256, 0, g_stream>>>(E, ht, pos.p, d_first.p, key.p, val.p);
    HHX_LAUNCH_CHECK();
    int bits = 1;
    while (bits < 63 && ((u64)h->n_side >> bits)) ++bits;
    HHX_TRY(stable_sort_pairs_u64(key.p, skey.p, val.p, sval.p, n, bits));
    key.release(); val.release(); skey.release(); d_first.release(); pos.release();
    DevBuf<i32> ni, nj;
    DevBuf<i64> cnt;
    if (ni.alloc((size_t)n) || nj.alloc((size_t)n) || cnt.alloc((size_t)n)) return 1;
    k_ht_items<<<grid_for((u64)n), 256, 0, g_stream>>>(n, sval.p, fi, fj, ht, ni.p, nj.p, cnt.p);
    HHX_LAUNCH_CHECK();
    HHX_HIP(hipMemcpyAsync(name_i, ni.p, sizeof(i32) * (size_t)n, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipMemcpyAsync(name_j, nj.p, sizeof(i32) * (size_t)n, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipMemcpyAsync(count, cnt.p, sizeof(i64) * (size_t)n, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}


// ================================================================================================ paired_links.clm
// output_clm :376-392 on the device.  For every contig pair with at least two read pairs (`len(list_) < 8: continue`), in
// full_link_dict order, four lines — one per orientation n of update_clm_dict's distances :395-401 —
//     {ctg_i}{+|-} {ctg_j}{+|-}\t{2 * links}\t{d d d d ...}\n      with the distances ascending, each written twice.
// The read pairs are already grouped by contig pair (group_pairs).  The kept groups are numbered in dict order and cut into
// chunks of at most CLM_CHUNK distances; per chunk: one 64-bit word (line << dbits | distance) per distance, a keys-only stable
// radix sort of those words (= every line's distances ascending, lines in order), the byte length of every distance (+ the
// line header on the first of a line), a scan, and a formatting kernel that builds each block's bytes in LDS and stores them
// with 16-byte writes.  The text leaves through two pinned host buffers, written by a host thread while the next piece is copied.
#include <fcntl.h>
#include <unistd.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <thread>

namespace {

constexpr i64 CLM_CHUNK = (i64)1 << 27;            // distances per chunk (1 GB of sort keys, ~2 GB of text)
constexpr int CW_T = 256, CW_OUT_CAP = 24 * 1024;

__device__ __forceinline__ i32 dlen(u64 a) { i32 n = 0; do { ++n; a /= 10; } while (a); return n; }
__device__ __forceinline__ unsigned char *dput(unsigned char *o, u64 a) {
    const i32 n = dlen(a);
    for (i32 k = n - 1; k >= 0; --k) { o[k] = (unsigned char)('0' + a % 10); a /= 10; }
    return o + n;
}

__global__ __launch_bounds__(256) void k_clm_by_r(i64 K, const i64 *__restrict__ gstart, const u64 *__restrict__ srank, i64 *__restrict__ keep_r,
                                                  i64 *__restrict__ g_of_r) {
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < K; g += (i64)gridDim.x * blockDim.x) {
        const i64 r = (i64)srank[g];
        keep_r[r] = gstart[g + 1] - gstart[g] >= 2 ? 1 : 0;          // :385
        g_of_r[r] = g;
    }
}
__global__ __launch_bounds__(256) void k_clm_kept(i64 K, const i64 *__restrict__ keep_r, const i64 *__restrict__ kidx, const i64 *__restrict__ g_of_r,
                                                  const i64 *__restrict__ gstart, const u64 *__restrict__ stk, const i64 *__restrict__ name_off,
                                                  i64 *__restrict__ kept_g, i64 *__restrict__ kept_cnt, i32 *__restrict__ hdr_len) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < K; r += (i64)gridDim.x * blockDim.x) {
        if (!keep_r[r]) continue;
        const i64 kr = kidx[r], g = g_of_r[r], c = gstart[g + 1] - gstart[g];
        const u64 key = stk[g];
        const i64 ci = (i64)(key >> ID_BITS), cj = (i64)(key & ID_MASK);
        kept_g[kr] = g;
        kept_cnt[kr] = c;
        // "{ctg_i}{s} {ctg_j}{s}\t{2c}\t"
        hdr_len[kr] = (i32)((name_off[ci + 1] - name_off[ci]) + (name_off[cj + 1] - name_off[cj]) + 5 + dlen((u64)(2 * c)));
    }
}
// one wavefront per kept contig pair: its 4 * cnt sort keys, orientation-major
__global__ __launch_bounds__(256) void k_clm_keys(i64 kr0, i64 kr1, const i64 *__restrict__ kept_g, const i64 *__restrict__ eoff, const i64 *__restrict__ gstart,
                                                  const u64 *__restrict__ stk, const u64 *__restrict__ sxy, const UnitInfo *__restrict__ ctg, int dbits,
                                                  u64 *__restrict__ keys, unsigned int *__restrict__ bad) {
    const int lane = lane_id();
    const i64 e0 = eoff[kr0];
    for (i64 kr = kr0 + (i64)blockIdx.x * 4 + threadIdx.x / HHX_WAVE; kr < kr1; kr += (i64)gridDim.x * 4) {
        const i64 g = kept_g[kr], b = gstart[g], cnt = gstart[g + 1] - b;
        const u64 key = stk[g];
        const i64 li = ctg[key >> ID_BITS].lenf & LEN_MASK, lj = ctg[key & ID_MASK].lenf & LEN_MASK;
        u64 *o = keys + 4 * (eoff[kr] - e0);
        const u64 line = (u64)(kr - kr0) * 4;
        for (i64 t = lane; t < cnt; t += HHX_WAVE) {
            const u64 xy = sxy[b + t];
            const i64 a = (i64)(xy >> 32) - 1, c = (i64)(xy & 0xffffffffu) - 1;      // 0-based (:1643)
            if (a >= li || c >= lj) atomicExch(bad, 1u);                             // a coordinate beyond the contig's end: negative distances
            o[t] = (line << dbits) | (u64)(li - a + c);
            o[cnt + t] = ((line + 1) << dbits) | (u64)(li - a + lj - c);
            o[2 * cnt + t] = ((line + 2) << dbits) | (u64)(a + c);
            o[3 * cnt + t] = ((line + 3) << dbits) | (u64)(a + lj - c);
        }
    }
}
__global__ __launch_bounds__(256) void k_clm_len(i64 E, const u64 *__restrict__ sk, int dbits, i64 kr0, const i32 *__restrict__ hdr_len, i64 *__restrict__ len) {
    const u64 dmask = (1ull << dbits) - 1;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (i64)gridDim.x * blockDim.x) {
        const u64 k = sk[e];
        const u64 line = k >> dbits;
        const bool first = e == 0 || (sk[e - 1] >> dbits) != line;
        len[e] = 2 * dlen(k & dmask) + 2 + (first ? hdr_len[kr0 + (i64)(line >> 2)] : 0);
    }
}
// bytes of CW_T consecutive distances are built in LDS at (offset - out_bias), out_bias chosen so that LDS and HBM addresses
// agree mod 16, and leave with 16-byte stores (bytes at the two ragged ends); a block whose span exceeds the LDS window (long
// contig names) writes to HBM directly
__global__ __launch_bounds__(CW_T) void k_clm_write(i64 E, const u64 *__restrict__ sk, int dbits, i64 kr0, const i64 *__restrict__ kept_g,
                                                    const i64 *__restrict__ kept_cnt, const u64 *__restrict__ stk, const unsigned char *__restrict__ names,
                                                    const i64 *__restrict__ name_off, const i64 *__restrict__ off, unsigned char *__restrict__ text) {
    __shared__ __attribute__((aligned(16))) unsigned char s_out[CW_OUT_CAP];
    const u64 dmask = (1ull << dbits) - 1;
    for (i64 e0 = (i64)blockIdx.x * CW_T; e0 < E; e0 += (i64)gridDim.x * CW_T) {
        const i64 e1 = e0 + CW_T < E ? e0 + CW_T : E;
        const i64 b0 = off[e0], b1 = off[e1];
        const i64 out_bias = b0 & ~(i64)15;                      // text is 16-byte aligned (pool allocation)
        const bool staged = b1 - out_bias <= CW_OUT_CAP;
        const i64 e = e0 + threadIdx.x;
        if (e < e1) {
            const u64 k = sk[e];
            const u64 line = k >> dbits;
            const bool first = e == 0 || (sk[e - 1] >> dbits) != line;
            const bool last = e + 1 == E || (sk[e + 1] >> dbits) != line;
            unsigned char *o = staged ? s_out + (off[e] - out_bias) : text + off[e];
            if (first) {
                const i64 kr = kr0 + (i64)(line >> 2);
                const u64 key = stk[kept_g[kr]];
                const i64 ci = (i64)(key >> ID_BITS), cj = (i64)(key & ID_MASK);
                const int n = (int)(line & 3);
                for (i64 q = name_off[ci]; q < name_off[ci + 1]; ++q) *o++ = names[q];
                *o++ = (n & 2) ? '-' : '+'; *o++ = ' ';
                for (i64 q = name_off[cj]; q < name_off[cj + 1]; ++q) *o++ = names[q];
                *o++ = (n & 1) ? '-' : '+'; *o++ = '\t';
                o = dput(o, (u64)(2 * kept_cnt[kr]));
                *o++ = '\t';
            }
            const u64 d = k & dmask;
            o = dput(o, d); *o++ = ' ';
            o = dput(o, d); *o++ = last ? '\n' : ' ';
        }
        __syncthreads();
        if (staged) {
            for (i64 o = (i64)threadIdx.x * 16; out_bias + o < b1; o += (i64)CW_T * 16) {
                const i64 gpos = out_bias + o;
                if (gpos >= b0 && gpos + 16 <= b1) *reinterpret_cast<uint4 *>(text + gpos) = *reinterpret_cast<const uint4 *>(s_out + o);
                else for (int q = 0; q < 16; ++q) if (gpos + q >= b0 && gpos + q < b1) text[gpos + q] = s_out[o + q];
            }
            __syncthreads();
        }
    }
}

int bits_for(u64 v) { int b = 1; while (b < 64 && (v >> b)) ++b; return b; }

}  // namespace

// paired_links.clm into an open file descriptor (closed here, whatever happens)
int hhx::ingest_write_clm_fd(hhx_ingest *h, int fd, const uint8_t *names_blob, const i64 *name_off, i64 *n_lines, i64 *n_bytes) {
    if (n_lines) *n_lines = 0;
    if (n_bytes) *n_bytes = 0;
    FileSink sink;
    HHX_TRY(sink.open_fd(fd));
    PairGroups G;
    HHX_TRY(group_pairs(h, G, "hhx_ingest_write_clm"));
    const i64 K = G.K;
    if (K == 0) return sink.close();
    KTimer kt("clm_text");
    // contig names on the device
    const i32 n_ctg = h->t.n_ctg;
    DevBuf<unsigned char> d_names;
    DevBuf<i64> d_noff;
    const size_t blob_bytes = (size_t)name_off[n_ctg];
    if (d_names.alloc(blob_bytes + 1) || d_noff.alloc((size_t)n_ctg + 1)) return 1;
    if (blob_bytes) HHX_HIP(hipMemcpyAsync(d_names.p, names_blob, blob_bytes, hipMemcpyHostToDevice, g_stream));
    HHX_HIP(hipMemcpyAsync(d_noff.p, name_off, sizeof(i64) * ((size_t)n_ctg + 1), hipMemcpyHostToDevice, g_stream));
    // kept groups (two read pairs or more) in dict order
    DevBuf<i64> keep_r, kidx, g_of_r;
    if (keep_r.alloc((size_t)K + 1) || kidx.alloc((size_t)K + 2) || g_of_r.alloc((size_t)K)) return 1;
    k_clm_by_r<<<grid_for((u64)K), 256, 0, g_stream>>>(K, G.gstart.p, G.srank.p, keep_r.p, g_of_r.p);
    HHX_LAUNCH_CHECK();
    i64 nk = 0;
    HHX_TRY(exclusive_scan_i64(keep_r.p, kidx.p, K, &nk));
    if (nk == 0) return sink.close();
    DevBuf<i64> kept_g, kept_cnt, eoff;
    DevBuf<i32> hdr_len;
    if (kept_g.alloc((size_t)nk) || kept_cnt.alloc((size_t)nk + 1) || eoff.alloc((size_t)nk + 2) || hdr_len.alloc((size_t)nk)) return 1;
    k_clm_kept<<<grid_for((u64)K), 256, 0, g_stream>>>(K, keep_r.p, kidx.p, g_of_r.p, G.gstart.p, G.stk.p, d_noff.p, kept_g.p, kept_cnt.p, hdr_len.p);
    HHX_LAUNCH_CHECK();
    i64 n_kept_pairs = 0;
    HHX_TRY(exclusive_scan_i64(kept_cnt.p, eoff.p, nk, &n_kept_pairs));
    keep_r.release(); kidx.release(); g_of_r.release();
    std::vector<i64> h_eoff((size_t)nk + 1);
    HHX_HIP(hipMemcpyAsync(h_eoff.data(), eoff.p, sizeof(i64) * ((size_t)nk + 1), hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    const int dbits = bits_for((u64)(2 * h->max_ctg_len));           // every distance is at most len_i + len_j
    DevBuf<unsigned int> bad;
    if (bad.alloc(1)) return 1;
    HHX_HIP(hipMemsetAsync(bad.p, 0, sizeof(unsigned int), g_stream));
    i64 lines = 0;
    for (i64 kr0 = 0; kr0 < nk;) {
        // as many whole groups as fit CLM_CHUNK distances (at least one)
        const i64 limit = h_eoff[(size_t)kr0] + CLM_CHUNK / 4;
        i64 kr1 = (i64)(std::upper_bound(h_eoff.begin() + kr0 + 1, h_eoff.end(), limit) - h_eoff.begin()) - 1;
        if (kr1 <= kr0) kr1 = kr0 + 1;
        const i64 E = 4 * (h_eoff[(size_t)kr1] - h_eoff[(size_t)kr0]);
        const int lbits = bits_for((u64)(4 * (kr1 - kr0)));
        if (dbits + lbits > 64) return fail("hhx_ingest_write_clm: %d + %d key bits", dbits, lbits);
        DevBuf<u64> keys, skeys;
        DevBuf<i64> len, off;
        if (keys.alloc((size_t)E) || skeys.alloc((size_t)E)) return 1;
        k_clm_keys<<<grid_for((u64)(kr1 - kr0) * 64), 256, 0, g_stream>>>(kr0, kr1, kept_g.p, eoff.p, G.gstart.p, G.stk.p, G.sxy.p, h->t.ctg, dbits, keys.p, bad.p);
        HHX_LAUNCH_CHECK();
        unsigned int hb = 0;
        HHX_HIP(hipMemcpyAsync(&hb, bad.p, sizeof hb, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        if (hb) return fail("hhx_ingest_write_clm: a read position lies beyond the end of its contig");
        HHX_TRY(stable_sort_pairs_u64(keys.p, skeys.p, nullptr, nullptr, E, dbits + lbits));
        keys.release();
        if (len.alloc((size_t)E + 1) || off.alloc((size_t)E + 2)) return 1;
        k_clm_len<<<grid_for((u64)E), 256, 0, g_stream>>>(E, skeys.p, dbits, kr0, hdr_len.p, len.p);
        HHX_LAUNCH_CHECK();
        i64 B = 0;
        HHX_TRY(exclusive_scan_i64(len.p, off.p, E, &B));
        len.release();
        DevBuf<unsigned char> text;
        if (text.alloc((size_t)B + 16)) return 1;
        k_clm_write<<<(unsigned)std::min<i64>((E + CW_T - 1) / CW_T, 256 * 32), CW_T, 0, g_stream>>>(E, skeys.p, dbits, kr0, kept_g.p, kept_cnt.p, G.stk.p,
                                                                                                      d_names.p, d_noff.p, off.p, text.p);
        HHX_LAUNCH_CHECK();
        HHX_TRY(sink.write_device(text.p, (size_t)B));
        lines += 4 * (kr1 - kr0);
        kr0 = kr1;
    }
    if (n_lines) *n_lines = lines;
    if (n_bytes) *n_bytes = sink.pos;
    return sink.close();
}

static int clm_preconditions(hhx_ingest *h, const char *who) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (!h->keep_pairs) return fail("%s: the handle was not created with hhx_ingest_keep_pairs", who);
    if (h->pairs_dropped) return fail("%s: the kept read pairs were released after paired_links.clm was written", who);
    return 0;
}

extern "C" int hhx_ingest_write_clm(hhx_ingest *h, const char *path, const uint8_t *names_blob, const int64_t *name_off, int64_t *n_lines, int64_t *n_bytes) {
    HHX_TRY(clm_preconditions(h, "hhx_ingest_write_clm"));
    if (!path || !names_blob || !name_off) return fail("hhx_ingest_write_clm: null pointer");
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return fail("cannot open %s for writing: %s", path, strerror(errno));
    return ingest_write_clm_fd(h, fd, names_blob, name_off, n_lines, n_bytes);
}

// ---- the same on the library's file-writer thread (hhx_jobs.hip).  What can be refused is refused here, on the caller's thread: a read position
// beyond the end of its contig (the device writer does not print negative distances; the caller then takes the reference's loop).
namespace {
__global__ __launch_bounds__(256) void k_clm_range_check(i64 n, const u64 *__restrict__ key, const u64 *__restrict__ xy, const UnitInfo *__restrict__ ctg,
                                                         unsigned int *__restrict__ bad) {
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (i64)gridDim.x * blockDim.x) {
        const u64 k = key[p], v = xy[p];
        const i64 li = ctg[k >> ID_BITS].lenf & LEN_MASK, lj = ctg[k & ID_MASK].lenf & LEN_MASK;
        if ((i64)(v >> 32) - 1 >= li || (i64)(v & 0xffffffffu) - 1 >= lj) atomicExch(bad, 1u);
    }
}
}  // namespace

extern "C" int hhx_ingest_write_clm_async(hhx_ingest *h, const char *path, const uint8_t *names_blob, const int64_t *name_off, int drop_pairs_after) {
    HHX_TRY(clm_preconditions(h, "hhx_ingest_write_clm_async"));
    if (!path || !names_blob || !name_off) return fail("hhx_ingest_write_clm_async: null pointer");
    const i32 *fi = nullptr, *fj = nullptr;
    HHX_TRY(hhx_ingest_ordered_full_device(h, &fi, &fj));            // the ordered tables are made here, not on the writer thread
    {
        DevBuf<unsigned int> bad;
        if (bad.alloc(1)) return 1;
        HHX_HIP(hipMemsetAsync(bad.p, 0, sizeof(unsigned int), g_stream));
        for (size_t b = 0; b < h->side_key.size(); ++b) {
            const i64 nb = (i64)h->side_key[b].n;
            if (!nb) continue;
            k_clm_range_check<<<grid_for((u64)nb), 256, 0, g_stream>>>(nb, h->side_key[b].p, h->side_xy[b].p, h->t.ctg, bad.p);
            HHX_LAUNCH_CHECK();
        }
        unsigned int hb = 0;
        HHX_HIP(hipMemcpyAsync(&hb, bad.p, sizeof hb, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));                      // also: everything the writer thread will read is complete
        if (hb) return fail("hhx_ingest_write_clm: a read position lies beyond the end of its contig");
    }
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return fail("cannot open %s for writing: %s", path, strerror(errno));
    const i32 n_ctg = h->t.n_ctg;
    auto blob = std::make_shared<std::vector<uint8_t>>(names_blob, names_blob + (size_t)name_off[n_ctg] + 1);
    auto off = std::make_shared<std::vector<i64>>(name_off, name_off + n_ctg + 1);
    const bool drop = drop_pairs_after != 0;
    return files_submit(std::string("paired_links.clm -> ") + path, h, [h, fd, blob, off, drop]() -> int {
        const int rc = ingest_write_clm_fd(h, fd, blob->data(), off->data(), nullptr, nullptr);
        if (drop) {                                                   // 16 B per read pair of HBM that nothing reads after this file
            (void)hipStreamSynchronize(g_stream);
            for (auto *v : {&h->side_key, &h->side_xy})
                for (auto &buf : *v) { void *p = buf.p; buf.p = nullptr; buf.n = 0; pool_free_synced(p); }
            h->pairs_dropped = true;
        }
        return rc;
    });
}

extern "C" int hhx_ingest_fetch_flank_values(hhx_ingest *h, double *value) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    void *val = nullptr;
    HHX_TRY(hhx_ingest_flank_device(h, nullptr, nullptr, &val));
    if (value && h->n_flank) HHX_HIP(hipMemcpyAsync(value, val, sizeof(double) * (size_t)h->n_flank, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
