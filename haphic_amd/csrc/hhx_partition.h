// Radix partition of (w0: u64, w1: u32 | u64 | nothing) records into 2^total_bits buckets, in levels of at most
// 8-9 bits.  Shared by the link-table group-by (hhx_ingest.hip: bucket = hash of the key) and by the link
// matrix build (hhx_matrix.hip: bucket = matrix row).
//
// Per level: a COUNT pass (LDS histogram per 4096-record tile -> global histogram), an exclusive scan, and
// a SCATTER pass.  The scatter stages the tile in LDS grouped by bucket (LDS histogram rank + tile-local
// exclusive scan) and then writes it out linearly, so every (tile, bucket) group is one contiguous,
// coalesced run of records; one global atomicAdd per (tile, bucket) reserves the run.  Scattering single
// 8/4-byte stores instead (first version) measured 4-5x write amplification in the HBM counters
// (profiles/r01_pmc_c3.txt).  Nothing here is stable or needs to be: records carry what they need.
#pragma once
#include "hhx_common.h"

namespace hhx {

constexpr int PT = 512, P_MAX_BINS = 512;
struct NoPayload {};       // w1_t of a source whose records are the bare 64-bit word
template <class W1> struct PartW1 { static constexpr bool HAS = true; static constexpr size_t BYTES = sizeof(W1); };
template <> struct PartW1<NoPayload> { static constexpr bool HAS = false; static constexpr size_t BYTES = 0; };
// records per thread and tile: 8 x 512 = 4096 records of 12 B, 7 x 512 of 16 B, 14 x 512 of 8 B — all stage in < 80 KB
// of LDS (two workgroups per CU).  The longer the tile, the longer the contiguous run a (tile, bucket) pair writes.
template <class W1> struct PartTile { static constexpr int ITEMS = !PartW1<W1>::HAS ? 14 : (sizeof(W1) == 4 ? 8 : 7), TILE = PT * ITEMS; };

struct PartLevel {
    int total_bits;     // buckets = 2^total_bits; bucket id comes from the Dig functor
    int shift;          // digit of this level = bucket >> shift
    int lds_bits;       // low lds_bits of the digit index the LDS histogram; the rest ("group") is constant
                        // within a tile except where a tile straddles two buckets of the previous level
};

template <class W1>
struct SrcRecs {
    typedef W1 w1_t;
    static constexpr bool MARK = false;
    const u64 *w0;
    const W1 *w1;
    __device__ __forceinline__ bool get(i64 idx, u64 &a, W1 &b) const {
        a = w0[idx];
        if constexpr (PartW1<W1>::HAS) b = w1[idx];
        return true;
    }
};

template <class Src, class Dig>
__device__ __forceinline__ u32 tile_group(const Src &src, const Dig &dig, i64 first, i64 n, const PartLevel &L) {
    if (L.shift + L.lds_bits >= L.total_bits) return 0;          // first level: the LDS histogram spans the whole digit
    u64 w0; typename Src::w1_t w1;
    (void)src.get(first < n ? first : n - 1, w0, w1);            // later levels read compact records: never invalid
    return (dig(w0) >> L.shift) >> L.lds_bits;
}

template <class Src, class Dig>
__global__ __launch_bounds__(PT) void k_part_count(Src src, Dig dig, i64 n, PartLevel L, unsigned long long *__restrict__ ghist) {
    constexpr int P_ITEMS = PartTile<typename Src::w1_t>::ITEMS, P_TILE = PartTile<typename Src::w1_t>::TILE;
    __shared__ u32 hist[P_MAX_BINS];
    __shared__ u32 s_grp;
    const int tid = threadIdx.x, nb = 1 << L.lds_bits;
    for (int t = tid; t < nb; t += PT) hist[t] = 0;
    u32 cur = 0xffffffffu;
    const i64 n_tiles = (n + P_TILE - 1) / P_TILE;
    for (i64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const i64 base = tile * P_TILE;
        if (tid == 0) s_grp = tile_group(src, dig, base, n, L);
        lds_barrier();
        const u32 tg = s_grp;
        if (tg != cur) {                                         // flush the histogram of the previous group
            if (cur != 0xffffffffu)
                for (int t = tid; t < nb; t += PT) {
                    if (hist[t]) atomicAdd(&ghist[((u64)cur << L.lds_bits) | (u64)t], (unsigned long long)hist[t]);
                    hist[t] = 0;
                }
            cur = tg;
            lds_barrier();
        }
        if constexpr (Src::MARK) {
            // a source with a side effect per record (get_marked / mark_load / mark_apply): the record loads of half a tile
            // first, then the loads the side effect needs, then its (rare) atomics — an atomic inside the per-record loop
            // would pin the loads of the following records behind it and serialise a dozen round trips per thread; halves,
            // because three 64-bit words per record in flight for the whole tile cost the kernel half its occupancy
            constexpr int HALF = (P_ITEMS + 1) / 2;
#pragma unroll
            for (int h = 0; h < P_ITEMS; h += HALF) {
                u64 w0[HALF], extra[HALF], seen[HALF];
                bool ok[HALF];
#pragma unroll
                for (int k = 0; k < HALF; ++k) {
                    const i64 idx = base + (i64)(h + k) * PT + tid;
                    ok[k] = h + k < P_ITEMS && idx < n && src.get_marked(idx, w0[k], extra[k]);
                }
#pragma unroll
                for (int k = 0; k < HALF; ++k) seen[k] = ok[k] ? src.mark_load(w0[k]) : 0;
#pragma unroll
                for (int k = 0; k < HALF; ++k)
                    if (ok[k]) {
                        src.mark_apply(w0[k], extra[k], seen[k]);
                        const u32 d = dig(w0[k]) >> L.shift;
                        if ((d >> L.lds_bits) == tg) atomicAdd(&hist[d & (u32)(nb - 1)], 1u);
                        else atomicAdd(&ghist[d], 1ull);
                    }
            }
        } else {
#pragma unroll
        for (int k = 0; k < P_ITEMS; ++k) {
            const i64 idx = base + (i64)k * PT + tid;
            u64 w0; typename Src::w1_t w1;
            if (idx < n && src.get(idx, w0, w1)) {
                const u32 d = dig(w0) >> L.shift;
                if ((d >> L.lds_bits) == tg) atomicAdd(&hist[d & (u32)(nb - 1)], 1u);
                else atomicAdd(&ghist[d], 1ull);
            }
        }
        }
        lds_barrier();
    }
    if (cur != 0xffffffffu)
        for (int t = tid; t < nb; t += PT)
            if (hist[t]) atomicAdd(&ghist[((u64)cur << L.lds_bits) | (u64)t], (unsigned long long)hist[t]);
}

template <class W1>
constexpr size_t part_scatter_lds() {
    return (size_t)PartTile<W1>::TILE * (8 + PartW1<W1>::BYTES + 2) + (size_t)P_MAX_BINS * (4 + 4 + 8) + 16;
}

template <class Src, class Dig>
__global__ __launch_bounds__(PT) void k_part_scatter(Src src, Dig dig, i64 n, PartLevel L, unsigned long long *__restrict__ cursor,
                                                     u64 *__restrict__ out_w0, typename Src::w1_t *__restrict__ out_w1) {
    typedef typename Src::w1_t W1;
    constexpr int P_ITEMS = PartTile<W1>::ITEMS, P_TILE = PartTile<W1>::TILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *s_w0 = (u64 *)smem;                                     // [P_TILE] tile grouped by bucket
    unsigned long long *gbase = (unsigned long long *)(s_w0 + P_TILE);   // [P_MAX_BINS] reserved global run of every bucket
    W1 *s_w1 = (W1 *)(gbase + P_MAX_BINS);                       // [P_TILE] (nothing for payload-free records)
    u32 *hist = (u32 *)((unsigned char *)s_w1 + (size_t)P_TILE * PartW1<W1>::BYTES);   // [P_MAX_BINS]
    u32 *lbase = hist + P_MAX_BINS;                              // [P_MAX_BINS] tile-local exclusive prefix
    u32 *s_misc = lbase + P_MAX_BINS;                            // [4]: group, wave sums scratch
    unsigned short *s_bin = (unsigned short *)(s_misc + 4);      // [P_TILE]
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / HHX_WAVE, nb = 1 << L.lds_bits;
    __shared__ u32 wsum[PT / HHX_WAVE];
    const i64 n_tiles = (n + P_TILE - 1) / P_TILE;
    for (i64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const i64 base = tile * P_TILE;
        for (int t = tid; t < nb; t += PT) hist[t] = 0;
        if (tid == 0) s_misc[0] = tile_group(src, dig, base, n, L);
        lds_barrier();
        const u32 tg = s_misc[0];
        u64 w0[P_ITEMS];
        W1 w1[P_ITEMS];
        u32 loc[P_ITEMS], rank[P_ITEMS];
#pragma unroll
        for (int k = 0; k < P_ITEMS; ++k) {
            const i64 idx = base + (i64)k * PT + tid;
            loc[k] = 0xffffffffu;
            if (idx < n && src.get(idx, w0[k], w1[k])) {
                const u32 d = dig(w0[k]) >> L.shift;
                if ((d >> L.lds_bits) == tg) {
                    loc[k] = d & (u32)(nb - 1);
                    rank[k] = atomicAdd(&hist[loc[k]], 1u);
                } else {                                         // straddling record: reserve its slot directly
                    const unsigned long long pos = atomicAdd(&cursor[d], 1ull);
                    out_w0[pos] = w0[k];
                    if constexpr (PartW1<W1>::HAS) out_w1[pos] = w1[k];
                }
            }
        }
        lds_barrier();
        // tile-local exclusive scan of the histogram (one bin per thread, nb <= PT) + global reservation
        {
            const u32 c = tid < nb ? hist[tid] : 0;
            u32 incl = c;
#pragma unroll
            for (int o = 1; o < HHX_WAVE; o <<= 1) {
                const u32 v = __shfl_up(incl, o, HHX_WAVE);
                if (lane >= o) incl += v;
            }
            if (lane == HHX_WAVE - 1) wsum[wave] = incl;
            lds_barrier();
            u32 woff = 0;
            for (int w = 0; w < wave; ++w) woff += wsum[w];
            if (tid < nb) {
                lbase[tid] = woff + incl - c;
                if (c) gbase[tid] = atomicAdd(&cursor[((u64)tg << L.lds_bits) | (u64)tid], (unsigned long long)c);
            }
            if (tid == PT - 1) s_misc[1] = woff + incl;          // records staged by this tile
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < P_ITEMS; ++k)
            if (loc[k] != 0xffffffffu) {
                const u32 s = lbase[loc[k]] + rank[k];
                s_w0[s] = w0[k];
                if constexpr (PartW1<W1>::HAS) s_w1[s] = w1[k];
                s_bin[s] = (unsigned short)loc[k];
            }
        lds_barrier();
        const u32 staged = s_misc[1];
        for (u32 s = tid; s < staged; s += PT) {                 // linear sweep: lanes write consecutive addresses inside a run
            const u32 b = s_bin[s];
            const unsigned long long pos = gbase[b] + (s - lbase[b]);
            out_w0[pos] = s_w0[s];
            if constexpr (PartW1<W1>::HAS) out_w1[pos] = s_w1[s];
        }
        lds_barrier();
    }
}

void u64_copy_async(const unsigned long long *src, unsigned long long *dst, i64 n);   // hhx_runtime.hip

// Host driver.  Partitions the records of `src` (n items, the source may drop some) into 2^total_bits buckets.
// Outputs: w0/w1 (bucket-grouped records), base[2^total_bits + 1] (device, record offsets), n_valid.
template <class W1>
struct Partitioned {
    DevBuf<u64> w0;
    DevBuf<W1> w1;
    DevBuf<i64> base;
    i64 n_valid = 0;
    u32 n_buckets = 1;
};

inline int part_levels(int total_bits, int max_bits, int *bits /* [4] */) {
    int n = total_bits <= 0 ? 1 : (total_bits + max_bits - 1) / max_bits;
    if (n > 4) n = 4;
    int left = total_bits;
    // the FIRST level gets the smaller share: its (tile, bucket) runs are scattered over the whole output, so they should be
    // long; a later level writes inside one bucket of the previous one, a region the L2 / Infinity Cache merges
    static const bool ascending = !getenv("HHX_PART_DESC");
    for (int l = 0; l < n; ++l) {
        bits[l] = ascending ? left / (n - l) : (left + (n - l) - 1) / (n - l);
        left -= bits[l];
    }
    return n;
}

// count_src (optional): the source object the level-1 COUNT pass reads through instead of src — same records, but its
// get() may carry a side effect that has to happen exactly once per record (hhx_matrix.hip: first positions).
// hist0 (optional): the level-1 histogram already counted by whoever produced the records (hhx_ingest.hip: k_map_records) —
// 2^bits[0] counts of the digit `bucket >> (total_bits - bits[0])` over the valid records; the level's COUNT pass is skipped.
template <class Src, class Dig>
int partition_records(const Src &src, const Dig &dig, i64 n_items, int total_bits, int max_bits_per_level,
                      Partitioned<typename Src::w1_t> *out, const char *timer_prefix, const Src *count_src = nullptr,
                      const unsigned long long *hist0 = nullptr) {
    typedef typename Src::w1_t W1;
    static int attr_dev = -1;           // the attribute is per device: keyed on the current ordinal (one static per instantiation)
    int dev = 0;
    HHX_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_part_scatter<Src, Dig>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_scatter_lds<W1>()));
        HHX_HIP(hipFuncSetAttribute((const void *)k_part_scatter<SrcRecs<W1>, Dig>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_scatter_lds<W1>()));
        attr_dev = dev;
    }
    if (max_bits_per_level > 9) max_bits_per_level = 9;          // P_MAX_BINS
    int bits[4];
    const int n_levels = part_levels(total_bits, max_bits_per_level, bits);
    int used = 0;
    for (int l = 0; l < n_levels; ++l) used += bits[l];
    if (used != total_bits) return fail("partition: %d bits do not fit %d levels", total_bits, n_levels);
    out->n_buckets = 1u << total_bits;
    DevBuf<u64> cur_w0, nxt_w0;
    DevBuf<W1> cur_w1, nxt_w1;
    DevBuf<i64> base;
    i64 n_cur = n_items;
    int done_bits = 0;
    char tname[64];
    for (int l = 0; l < n_levels; ++l) {
        done_bits += bits[l];
        const u32 nbk = 1u << done_bits;                         // buckets after this level
        const PartLevel L{total_bits, total_bits - done_bits, bits[l]};
        DevBuf<unsigned long long> hist, cursor;
        DevBuf<i64> nbase;
        if (hist.alloc((size_t)nbk + 1) || cursor.alloc((size_t)nbk + 1) || nbase.alloc((size_t)nbk + 2)) return 1;
        HHX_HIP(hipMemsetAsync(hist.p, 0, sizeof(unsigned long long) * ((size_t)nbk + 1), g_stream));
        const i64 tile = l == 0 ? PartTile<W1>::TILE : PartTile<W1>::TILE;
        const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>((n_cur + tile - 1) / tile, 256 * 4));
        const SrcRecs<W1> rs{cur_w0.p, cur_w1.p};
        snprintf(tname, sizeof tname, "%s_count%d", timer_prefix, l + 1);
        { KTimer kt(tname);
        if (l == 0 && hist0) u64_copy_async(hist0, hist.p, (i64)nbk);
        else if (l == 0) k_part_count<Src, Dig><<<grid, PT, 0, g_stream>>>(count_src ? *count_src : src, dig, n_cur, L, hist.p);
        else k_part_count<SrcRecs<W1>, Dig><<<grid, PT, 0, g_stream>>>(rs, dig, n_cur, L, hist.p); }
        HHX_LAUNCH_CHECK();
        i64 n_valid = 0;
        HHX_TRY(exclusive_scan_i64((const i64 *)hist.p, nbase.p, nbk, &n_valid));
        if (l == 0) out->n_valid = n_valid;
        if (n_valid == 0) {
            out->n_valid = 0;
            out->base = std::move(nbase);
            return 0;
        }
        if (nxt_w0.alloc((size_t)n_valid) || (PartW1<W1>::HAS && nxt_w1.alloc((size_t)n_valid))) return 1;
        u64_copy_async((const unsigned long long *)nbase.p, cursor.p, (i64)nbk + 1);
        snprintf(tname, sizeof tname, "%s_scatter%d", timer_prefix, l + 1);
        { KTimer kt(tname);
        if (l == 0) k_part_scatter<Src, Dig><<<grid, PT, part_scatter_lds<W1>(), g_stream>>>(src, dig, n_cur, L, cursor.p, nxt_w0.p, nxt_w1.p);
        else k_part_scatter<SrcRecs<W1>, Dig><<<grid, PT, part_scatter_lds<W1>(), g_stream>>>(rs, dig, n_cur, L, cursor.p, nxt_w0.p, nxt_w1.p); }
        HHX_LAUNCH_CHECK();
        HHX_HIP(hipStreamSynchronize(g_stream));                 // hist / cursor die here; the previous level's records too
        cur_w0 = std::move(nxt_w0);
        cur_w1 = std::move(nxt_w1);
        base = std::move(nbase);
        n_cur = n_valid;
    }
    out->w0 = std::move(cur_w0);
    out->w1 = std::move(cur_w1);
    out->base = std::move(base);
    return 0;
}

}  // namespace hhx
