// Stable LSD radix sort of (u64 key, u64 value) pairs, 8 bits per pass — the grouping step of the f2 side products
// (hhx_pairs.hip: read pairs grouped by contig pair, stream order kept inside a group, which is what makes
// update_clm_dict's lists :395-401 and record_coord_pairs' first-max_read_pairs rule :454-459 come out right).
//
// Per pass: k_rs_hist (tile histogram in LDS -> global table laid out [digit][tile]), one exclusive scan of that table
// (= the global position of the first item of every (digit, tile)), k_rs_scatter (re-reads the tile and places every
// item at table[digit][tile] + its rank among the tile's items of that digit).  Stability comes from the rank: a tile
// is walked in rounds of 256 consecutive items (item = round * 256 + thread, so loads are coalesced and the order
// (round, wave, lane) is the stream order); inside a wave, lanes with the same digit find each other with eight
// ballots (one per digit bit) and take their rank from a popcount; the waves' group sizes go through LDS.
#pragma once
#include "hhx_common.h"

namespace hhx {

constexpr int RS_T = 256, RS_ROUNDS = 8, RS_TILE = RS_T * RS_ROUNDS, RS_BINS = 256, RS_WAVES = RS_T / HHX_WAVE;

static __global__ __launch_bounds__(RS_T) void k_rs_hist(const u64 *__restrict__ key, i64 n, int shift, i64 n_tiles, i64 *__restrict__ table) {
    __shared__ u32 hist[RS_BINS];
    for (i64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const i64 base = tile * RS_TILE;
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; ++r) {
            const i64 i = base + r * RS_T + threadIdx.x;
            if (i < n) atomicAdd(&hist[(u32)(key[i] >> shift) & (RS_BINS - 1)], 1u);
        }
        __syncthreads();
        table[(i64)threadIdx.x * n_tiles + tile] = hist[threadIdx.x];
        __syncthreads();
    }
}

template <bool HAS_VAL>
static __global__ __launch_bounds__(RS_T) void k_rs_scatter(const u64 *__restrict__ key, const u64 *__restrict__ val, i64 n, int shift, i64 n_tiles,
                                                     const i64 *__restrict__ table, u64 *__restrict__ okey, u64 *__restrict__ oval) {
    __shared__ i64 base[RS_BINS];                 // global position of the next item of every digit of this tile
    __shared__ u32 wcnt[RS_WAVES][RS_BINS];       // items of every digit per wave, current round
    const int lane = lane_id(), wave = threadIdx.x / HHX_WAVE;
    const u64 lt = (1ull << lane) - 1ull;
    for (i64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        base[threadIdx.x] = table[(i64)threadIdx.x * n_tiles + tile];
        const i64 first = tile * RS_TILE;
        for (int r = 0; r < RS_ROUNDS; ++r) {
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) wcnt[w][threadIdx.x] = 0;
            __syncthreads();
            const i64 i = first + r * RS_T + threadIdx.x;
            const bool in = i < n;
            u64 k = 0, v = 0;
            if (in) { k = key[i]; if (HAS_VAL) v = val[i]; }
            const u32 d = (u32)(k >> shift) & (RS_BINS - 1);
            u64 peers = __ballot(in);             // lanes of this wave holding the same digit
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const u64 m = __ballot(in && ((d >> b) & 1u));
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            const u32 rank = (u32)__popcll(peers & lt);
            if (in && rank == 0) wcnt[wave][d] = (u32)__popcll(peers);      // one writer per (wave, digit)
            __syncthreads();
            if (in) {
                i64 pos = base[d] + rank;
                for (int w = 0; w < wave; ++w) pos += wcnt[w][d];
                okey[pos] = k;
                if (HAS_VAL) oval[pos] = v;
            }
            __syncthreads();
            {
                u32 add = 0;
#pragma unroll
                for (int w = 0; w < RS_WAVES; ++w) add += wcnt[w][threadIdx.x];
                base[threadIdx.x] += add;
            }
            __syncthreads();
        }
    }
}

// kout / vout receive the pairs sorted by the low `bits` bits of the key; equal keys keep their input order.
// kin / vin are left untouched.  vin == nullptr: keys only (vout is not written).
inline int stable_sort_pairs_u64(const u64 *kin, u64 *kout, const u64 *vin, u64 *vout, i64 n, int bits) {
    if (n <= 0) return 0;
    const bool has_val = vin != nullptr;
    const int passes = std::max(1, (bits + 7) / 8);
    const i64 n_tiles = (n + RS_TILE - 1) / RS_TILE;
    DevBuf<u64> tk, tv;
    DevBuf<i64> table, offs;
    if (table.alloc((size_t)n_tiles * RS_BINS + 1) || offs.alloc((size_t)n_tiles * RS_BINS + 2)) return 1;
    if (passes > 1 && (tk.alloc((size_t)n) || (has_val && tv.alloc((size_t)n)))) return 1;
    const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>(n_tiles, 256 * 8));
    // ping-pong so that the LAST pass writes kout / vout
    const u64 *sk = kin, *sv = vin;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        u64 *dk = to_out ? kout : tk.p, *dv = to_out ? vout : tv.p;
        k_rs_hist<<<grid, RS_T, 0, g_stream>>>(sk, n, 8 * p, n_tiles, table.p);
        HHX_LAUNCH_CHECK();
        i64 total = 0;
        HHX_TRY(exclusive_scan_i64(table.p, offs.p, n_tiles * RS_BINS, &total));
        if (total != n) return fail("radix sort: histogram total %lld != %lld", (long long)total, (long long)n);
        if (has_val) k_rs_scatter<true><<<grid, RS_T, 0, g_stream>>>(sk, sv, n, 8 * p, n_tiles, offs.p, dk, dv);
        else k_rs_scatter<false><<<grid, RS_T, 0, g_stream>>>(sk, nullptr, n, 8 * p, n_tiles, offs.p, dk, nullptr);
        HHX_LAUNCH_CHECK();
        sk = dk; sv = dv;
    }
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

}  // namespace hhx
