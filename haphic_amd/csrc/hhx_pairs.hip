// Side products of the ingest loop that need the read pairs themselves, not only counts:
//   clm_dict      update_clm_dict :395-401 — four orientation distances per read pair, per contig pair, stream order
//   ctg_coord_dict record_coord_pairs :454-471 — the first max_read_pairs (coord_i, coord_j) per contig pair
// Both are "group the pairs by contig pair, keep stream order inside a group".  The pairs counted in
// full_link_dict are compacted (stably, so they stay in stream order) into (key, xi << 32 | xj) records at push
// time; at fetch time one STABLE radix sort by key (hhx_sort.h: hand-written LSD passes, ballot-ranked so that equal
// keys keep their stream order) groups them, the insertion-ordered key table is sorted the same way to pair every
// group with its dict position, and one wavefront per contig pair writes its distances.
#include <cstring>

#include "hhx_ingest.h"
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

extern "C" int hhx_ingest_fetch_pairs(hhx_ingest *h, i64 max_read_pairs, i64 *clm_ptr, i64 *clm, i64 *crd_ptr, i64 *crd) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (!h->keep_pairs) return fail("hhx_ingest_fetch_pairs: the handle was not created with hhx_ingest_keep_pairs");
    if (max_read_pairs < 0) max_read_pairs = 0;
    const i32 *fi = nullptr, *fj = nullptr;
    HHX_TRY(hhx_ingest_ordered_full_device(h, &fi, &fj));
    const i64 K = h->n_full, N = h->n_side;
    if (clm_ptr) clm_ptr[0] = 0;
    if (crd_ptr) crd_ptr[0] = 0;
    if (K == 0) return 0;
    // concatenate the pushes (stream order), stable sort by key
    DevBuf<u64> key, xy, skey, sxy, tkey, trnk, stk, srank;
    if (key.alloc((size_t)N) || xy.alloc((size_t)N) || skey.alloc((size_t)N) || sxy.alloc((size_t)N) || tkey.alloc((size_t)K) || trnk.alloc((size_t)K) ||
        stk.alloc((size_t)K) || srank.alloc((size_t)K)) return 1;
    i64 o = 0;
    for (size_t b = 0; b < h->side_key.size(); ++b) {
        const i64 nb = (i64)h->side_key[b].n;
        if (nb) {
            HHX_HIP(hipMemcpyAsync(key.p + o, h->side_key[b].p, 8 * (size_t)nb, hipMemcpyDeviceToDevice, g_stream));
            HHX_HIP(hipMemcpyAsync(xy.p + o, h->side_xy[b].p, 8 * (size_t)nb, hipMemcpyDeviceToDevice, g_stream));
        }
        o += nb;
    }
    HHX_TRY(sort_pairs_u64(key.p, skey.p, xy.p, sxy.p, N));
    k_table_keys<<<grid_for((u64)K), 256, 0, g_stream>>>(K, fi, fj, tkey.p, trnk.p);
    HHX_LAUNCH_CHECK();
    HHX_TRY(sort_pairs_u64(tkey.p, stk.p, trnk.p, srank.p, K));
    // groups of the sorted records <-> sorted table keys
    DevBuf<i64> flag, gidx, gstart, cnt_r, cap_r, clm_off, crd_off;
    DevBuf<unsigned int> mismatch;
    if (flag.alloc((size_t)N + 1) || gidx.alloc((size_t)N + 2) || gstart.alloc((size_t)K + 2) || cnt_r.alloc((size_t)K + 1) || cap_r.alloc((size_t)K + 1) ||
        clm_off.alloc((size_t)K + 2) || crd_off.alloc((size_t)K + 2) || mismatch.alloc(1)) return 1;
    HHX_HIP(hipMemsetAsync(mismatch.p, 0, sizeof(unsigned int), g_stream));
    k_boundary_flags<<<grid_for((u64)N), 256, 0, g_stream>>>(N, skey.p, flag.p);
    HHX_LAUNCH_CHECK();
    i64 n_groups = 0;
    HHX_TRY(exclusive_scan_i64(flag.p, gidx.p, N, &n_groups));
    if (n_groups != K) return fail("hhx_ingest_fetch_pairs: %lld contig pairs in the records, %lld in the table", (long long)n_groups, (long long)K);
    k_group_starts<<<grid_for((u64)N), 256, 0, g_stream>>>(N, skey.p, gidx.p, gstart.p);
    HHX_HIP(hipMemcpyAsync(gstart.p + K, &N, sizeof(i64), hipMemcpyHostToDevice, g_stream));
    k_group_counts<<<grid_for((u64)K), 256, 0, g_stream>>>(K, stk.p, skey.p, gstart.p, srank.p, max_read_pairs, cnt_r.p, cap_r.p, mismatch.p);
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
    k_emit_pairs<<<grid_for((u64)K * 64), 256, 0, g_stream>>>(K, stk.p, gstart.p, srank.p, sxy.p, h->t.ctg, clm_off.p, crd_off.p, max_read_pairs, d_clm.p, d_crd.p);
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
extern "C" int hhx_ingest_fetch_ht_order(hhx_ingest *h, i64 *first) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (!h->keep_pairs) return fail("hhx_ingest_fetch_ht_order: the handle was not created with hhx_ingest_keep_pairs");
    const i32 *fi = nullptr, *fj = nullptr;
    HHX_TRY(hhx_ingest_ordered_full_device(h, &fi, &fj));
    const i64 K = h->n_full, N = h->n_side;
    if (K == 0) return 0;
    if (!first) return fail("hhx_ingest_fetch_ht_order: null output");
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
    DevBuf<i64> flag, gidx, gstart, d_first;
    if (flag.alloc((size_t)N + 1) || gidx.alloc((size_t)N + 2) || gstart.alloc((size_t)K + 2) || d_first.alloc((size_t)K * 4)) return 1;
    k_boundary_flags<<<grid_for((u64)N), 256, 0, g_stream>>>(N, skey.p, flag.p);
    HHX_LAUNCH_CHECK();
    i64 n_groups = 0;
    HHX_TRY(exclusive_scan_i64(flag.p, gidx.p, N, &n_groups));
    if (n_groups != K) return fail("hhx_ingest_fetch_ht_order: %lld contig pairs in the records, %lld in the table", (long long)n_groups, (long long)K);
    k_group_starts<<<grid_for((u64)N), 256, 0, g_stream>>>(N, skey.p, gidx.p, gstart.p);
    HHX_HIP(hipMemcpyAsync(gstart.p + K, &N, sizeof(i64), hipMemcpyHostToDevice, g_stream));
    k_ht_first<<<grid_for((u64)K * 64), 256, 0, g_stream>>>(K, stk.p, gstart.p, srank.p, sidx.p, xy.p, h->t.ctg, d_first.p);
    HHX_LAUNCH_CHECK();
    HHX_HIP(hipMemcpyAsync(first, d_first.p, sizeof(i64) * (size_t)K * 4, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
