// Hi-C read-pair binning into the contig x contig (full / HT) and fragment x fragment (flank) link
// tables.  Reference: scripts/HapHiC_cluster.py:1596-1752 (parse_alignments_for_ctgs, parse_alignments),
// :299-307 (is_flank), :404-416 (update_HT_link_dict).
//
// The reference updates Python dicts pair by pair.  Here the same tables are built as a streaming
// group-by, with every pass reading and writing HBM sequentially (or in bucket-sized runs):
//   map      one thread per read pair: membership / orientation / bin conversion / flank test, exactly
//            the predicate chain of :1622-1653 (:1696-1750), producing ONE 64-bit record per surviving
//            pair: key (i << 29 | j) + HT quadrant + "counts in full_link_dict" + "counts in
//            flank_link_dict" bits, plus the pair's 32-bit ordinal in the batch;
//   partition  records are bucketed by the top bits of a 64-bit mix of the key in one or two radix
//            levels (LDS histogram per 4096-record tile, one global atomic per (tile, bucket) to
//            reserve the output range, grouped writes).  No ordering inside a bucket is needed — the
//            ordinal travels with the record — so there is no stable-sort machinery;
//   aggregate  one workgroup per bucket (~1k records): an LDS hash table keyed by the record key
//            accumulates the four HT counters, the flank counter and the MINIMUM ordinal seen by each
//            dict (ds_cmpst_b64 insert, ds_add / ds_min updates), then writes one row per distinct key.
// Counts and minima are order-independent integers, so the tables are bit-reproducible.  The Python
// dict INSERTION ORDER that dict_to_matrix's index assignment depends on (:337-349) is the order of the
// first-seen ordinals; it is recovered only where someone asks for it (hhx_ingest_fetch, the S5 seam)
// by ranking the ordinals with a bitmap + popcount prefix — the fused device path
// (hhx_ingest_link_matrix) uses the ordinals directly and never materialises the order.
// Several pushes produce several aggregated runs; hhx_ingest_finalize merges them with the same
// partition + aggregate pipeline (rows instead of pairs), which is also the multi-GPU exchange step.
#include "hhx_ingest.h"
#include "hhx_partition.h"

using namespace hhx;

namespace {

__device__ __host__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ u32 bucket_of(u64 key, int total_bits) {
    return total_bits ? (u32)(mix64(key) >> (64 - total_bits)) : 0u;
}

// ---- record sources (hhx_partition.h: get(idx, w0, w1)) --------------------------------------------------
template <bool COMBINED, class POS = i32>
struct SrcPairs {
    typedef u32 w1_t;
    static constexpr bool MARK = false;
    const i32 *id1; const POS *pos1; const i32 *id2; const POS *pos2;
    DevTables t;
    int stream;
    __device__ __forceinline__ bool get(i64 idx, u64 &rec, u32 &ord) const {
        ord = (u32)idx;
        return map_pair<COMBINED>(t, stream, id1[idx], id2[idx], (i64)pos1[idx], (i64)pos2[idx], rec);
    }
};
// The map is evaluated ONCE per pair (k_map_records: 16 B read, 8 B written); the count and scatter passes of the
// first radix level then stream the 8-byte records instead of re-running the predicate chain on the 16-byte pairs
// (measured: count 6.8 -> 0.8 ms, scatter 8.1 -> 3 ms per 500 M pairs, for 4 GB of scratch).
// Four consecutive pairs per lane: the four streams arrive as 16-byte loads, the eight table gathers (clamped ids,
// unconditional) are all issued before the first is consumed, and the records leave as two 16-byte stores — the
// kernel is bound by the latency of its dependent loads, not by bytes.  VEC = false: any alignment, any tail.
// The level-1 histogram of the group-by's radix partition is counted here too (hist: 2^bits counts of `bucket >> shift` over the
// admitted records, or null): the pass waits on its gathers, the hash and an LDS atomic per record ride along, and the
// partition's own COUNT pass over the 8-byte records (4 GB, 0.8 ms per 500 M pairs) is not launched.
struct MapHist {
    unsigned long long *hist;       // null: no histogram
    int total_bits, shift, bins;
};
__device__ __forceinline__ void map_hist_add(u32 *lh, const MapHist &H, u64 w) {
    if (w != EMPTY_KEY) atomicAdd(&lh[bucket_of(w & KEY_MASK, H.total_bits) >> H.shift], 1u);
}
template <bool COMBINED, bool VEC, class POS = i32>
__global__ __launch_bounds__(256) void k_map_records(SrcPairs<COMBINED, POS> src, i64 i0, i64 n, u64 *__restrict__ rec, MapHist H) {
    __shared__ u32 lh[hhx::P_MAX_BINS];
    if (H.hist) {
        for (int t = threadIdx.x; t < H.bins; t += blockDim.x) lh[t] = 0;
        __syncthreads();
    }
    if (!VEC || sizeof(POS) != 4) {
        for (i64 idx = i0 + (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (i64)gridDim.x * blockDim.x) {
            u64 w0; u32 w1;
            const u64 w = src.get(idx, w0, w1) ? w0 : EMPTY_KEY;
            rec[idx] = w;
            if (H.hist) map_hist_add(lh, H, w);
        }
    } else {
    const i64 groups = n >> 2;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (i64)gridDim.x * blockDim.x) {
        const int4 q1 = reinterpret_cast<const int4 *>(src.id1)[g], q2 = reinterpret_cast<const int4 *>(src.id2)[g];
        const int4 x1 = reinterpret_cast<const int4 *>(src.pos1)[g], x2 = reinterpret_cast<const int4 *>(src.pos2)[g];
        const i32 r[4] = {q1.x, q1.y, q1.z, q1.w}, m[4] = {q2.x, q2.y, q2.z, q2.w};
        const i32 p1[4] = {x1.x, x1.y, x1.z, x1.w}, p2[4] = {x2.x, x2.y, x2.z, x2.w};
        bool ok[4];
        UnitInfo a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ok[k] = pair_admitted(src.t, r[k], m[k]);
            a[k] = src.t.ctg[ok[k] ? r[k] : 0];
            b[k] = src.t.ctg[ok[k] ? m[k] : 0];
        }
        u64 w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u64 v;
            w[k] = (ok[k] && map_pair_with<COMBINED>(src.t, src.stream, r[k], m[k], p1[k], p2[k], a[k], b[k], v)) ? v : EMPTY_KEY;
        }
        reinterpret_cast<ulonglong2 *>(rec)[2 * g] = make_ulonglong2(w[0], w[1]);
        reinterpret_cast<ulonglong2 *>(rec)[2 * g + 1] = make_ulonglong2(w[2], w[3]);
        if (H.hist) {
#pragma unroll
            for (int k = 0; k < 4; ++k) map_hist_add(lh, H, w[k]);
        }
    }
    }
    if (H.hist) {
        __syncthreads();
        for (int t = threadIdx.x; t < H.bins; t += blockDim.x)
            if (lh[t]) atomicAdd(&H.hist[t], (unsigned long long)lh[t]);
    }
}
template <bool COMBINED, class POS>
void launch_map(const SrcPairs<COMBINED, POS> &src, i64 n, u64 *rec, const MapHist &H) {
    // (64-bit positions — contigs beyond 2^31 bp, rare — take the scalar kernel: one pair per lane)
    const bool aligned = sizeof(POS) == 4 && ((((uintptr_t)src.id1) | ((uintptr_t)src.id2) | ((uintptr_t)src.pos1) | ((uintptr_t)src.pos2) | ((uintptr_t)rec)) & 15) == 0;
    const i64 bulk = aligned && src.t.n_ctg > 0 ? (n & ~(i64)3) : 0;
    if (bulk) k_map_records<COMBINED, true, POS><<<(unsigned)std::max<i64>(1, std::min<i64>((bulk / 4 + 255) / 256, 256 * 16)), 256, 0, g_stream>>>(src, 0, bulk, rec, H);
    if (bulk < n) k_map_records<COMBINED, false, POS><<<(unsigned)std::max<i64>(1, std::min<i64>((n - bulk + 255) / 256, 256 * 16)), 256, 0, g_stream>>>(src, bulk, n, rec, H);
}
struct SrcMapped {
    typedef u32 w1_t;
    static constexpr bool MARK = false;
    const u64 *rec;
    __device__ __forceinline__ bool get(i64 idx, u64 &w0, u32 &ord) const {
        w0 = rec[idx];
        ord = (u32)idx;
        return w0 != EMPTY_KEY;
    }
};
struct SrcRows {            // table rows to be merged: the record is the bare key, the "ordinal" the row index
    typedef u32 w1_t;
    static constexpr bool MARK = false;
    const u64 *key;
    __device__ __forceinline__ bool get(i64 idx, u64 &rec, u32 &ord) const {
        rec = key[idx] & KEY_MASK;
        ord = (u32)idx;
        return true;
    }
};
// bucket of a record = top bits of the 64-bit mix of its key
struct DigKeyHash {
    int total_bits;
    __device__ __forceinline__ u32 operator()(u64 w0) const { return bucket_of(w0 & KEY_MASK, total_bits); }
};

// ---- aggregation: LDS hash table per bucket -----------------------------------------------------------
constexpr int AG_T = 512, AG_CAP = 2048;

struct AggParams {
    const u64 *rec;                       // partitioned records
    const u32 *ord;
    const unsigned long long *base;       // [n_buckets + 1]
    u32 n_buckets, buckets_per_wg;
    int total_bits;                       // bucket = top total_bits of mix64(key); the next bits pick the sub-pass
    u64 ord_base;                         // MODE 0: global ordinal of pair 0 of the batch
    const u64 *in_ord_full, *in_ord_flank;   // MODE 1: payload rows, indexed by ord
    const u32 *in_ht, *in_fl;
    u64 *o_key, *o_ord_full, *o_ord_flank;   // the run itself: every sub-pass reserves its rows with one atomic add on out_cursor
    u32 *o_ht, *o_fl;                        // (a run is unordered by definition, so the reservation order does not matter)
    unsigned long long *out_cursor;       // [1] rows written so far
    unsigned int *overflow;
};

// MODE 0: pair records (flags inside rec, 32-bit batch ordinals).  MODE 1: table rows (64-bit ordinals + counts gathered by row).
template <int MODE>
struct AggLds;
template <>
struct AggLds<0> { typedef u32 ord_t; };
template <>
struct AggLds<1> { typedef u64 ord_t; };

template <int MODE>
__global__ __launch_bounds__(AG_T) void k_aggregate(AggParams A) {
    typedef typename AggLds<MODE>::ord_t ord_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *s_key = (u64 *)smem;
    ord_t *s_of = (ord_t *)(s_key + AG_CAP);
    ord_t *s_ok = s_of + AG_CAP;
    u32 *s_cnt = (u32 *)(s_ok + AG_CAP);          // [5][AG_CAP]: HH, HT, TH, TT, flank
    u32 *s_scan = s_cnt + 5 * AG_CAP;              // [AG_T / 64 + 1]
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / HHX_WAVE;
    const ord_t ORD_NONE = (ord_t)~(ord_t)0;
    const u32 b0 = blockIdx.x * A.buckets_per_wg;
    const u32 b1 = min(A.n_buckets, b0 + A.buckets_per_wg);
    if (b0 >= A.n_buckets) return;
    unsigned long long &s_base = *reinterpret_cast<unsigned long long *>(s_scan + 16);   // dynamic LDS only: the 160 KB attribute needs it all
    // A bucket is ~1k records (two per thread): a chain of dependent round trips — bounds, records, the output
    // reservation, the stores — so the kernel is latency bound.  The first AG_R records per thread of the NEXT bucket are
    // loaded into registers before the current bucket's insert / scan / store phases begin.
    constexpr int AG_R = 4;
    u32 b = b0;
    unsigned long long rb = A.base[b], re = A.base[b + 1];
    unsigned long long rbn = b + 1 < b1 ? A.base[b + 1] : 0ull, ren = b + 1 < b1 ? A.base[b + 2] : 0ull;      // bounds run two buckets ahead
    u64 rec_r[AG_R];
    u32 ord_r[AG_R];
#pragma unroll
    for (int u = 0; u < AG_R; ++u) {
        const unsigned long long i = rb + tid + (unsigned long long)u * AG_T;
        rec_r[u] = i < re ? A.rec[i] : EMPTY_KEY;
        ord_r[u] = i < re ? A.ord[i] : 0u;
    }
    // The loop must be entered with these registers DEFINED, not with loads pending on them: the waitcnt pass is static,
    // so a load pending at the loop entry puts an s_waitcnt vmcnt(0) in front of every use inside the loop — and vmcnt(0)
    // also waits for the prefetch the iteration has just issued.
#pragma unroll
    for (int u = 0; u < AG_R; ++u) { asm volatile("" : "+v"(rec_r[u])); asm volatile("" : "+v"(ord_r[u])); }
    asm volatile("" : "+v"(rbn)); asm volatile("" : "+v"(ren));
    for (;;) {
        const u32 bn = b + 1;
        u64 rec_n[AG_R];
        u32 ord_n[AG_R];
#pragma unroll
        for (int u = 0; u < AG_R; ++u) {
            const unsigned long long i = rbn + tid + (unsigned long long)u * AG_T;
            rec_n[u] = i < ren ? A.rec[i] : EMPTY_KEY;
            ord_n[u] = i < ren ? A.ord[i] : 0u;
        }
        unsigned long long rbn2 = 0, ren2 = 0;
        if (bn + 1 < b1) { rbn2 = A.base[bn + 1]; ren2 = A.base[bn + 2]; }
        const u64 n = re - rb;
        if (n) {
        // A bucket holds ~8k records at most (read once from HBM, L2-resident afterwards); it is aggregated in 2^sbits
        // SUB-PASSES over the same records, sub-pass s taking the keys whose next sbits hash bits equal s, so
        // that each sub-pass sees ~1k records and fits the 2048-slot table.  (One more radix level would move
        // every record through HBM again instead.)
        int sbits = 0;
        while ((n >> sbits) > 1024 && sbits < 4) ++sbits;
        const u64 per_sub = (n >> sbits) + 1;
        u32 tsize = AG_CAP;
        if (2 * per_sub <= AG_CAP) { tsize = 64; while (tsize < 2 * per_sub) tsize <<= 1; }
        for (u32 sub = 0; sub < (1u << sbits); ++sub) {
        for (u32 s = tid; s < tsize; s += AG_T) {
            s_key[s] = EMPTY_KEY; s_of[s] = ORD_NONE; s_ok[s] = ORD_NONE;
#pragma unroll
            for (int c = 0; c < 5; ++c) s_cnt[c * AG_CAP + s] = 0;
        }
        lds_barrier();
        auto insert = [&](const u64 rec, const u32 ord) {
            const u64 key = rec & KEY_MASK;
            const u64 h = mix64(key);
            if (sbits && (u32)((h << A.total_bits) >> (64 - sbits)) != sub) return;
            u32 slot = (u32)h & (tsize - 1);
            u32 probe = 0;
            for (; probe < tsize; ++probe) {
                // (a volatile access would lose the LDS address space: a FLAT load, which waits for every outstanding global load)
                const u64 cur = __hip_atomic_load(&s_key[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == key) break;
                if (cur == EMPTY_KEY) {
                    const u64 old = atomicCAS((unsigned long long *)&s_key[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                    if (old == EMPTY_KEY || old == key) break;
                }
                slot = (slot + 1) & (tsize - 1);
            }
            if (probe == tsize) { atomicExch(A.overflow, 1u); return; }
            if constexpr (MODE == 0) {
                if (rec & FULL_BIT) {
                    atomicMin((u32 *)&s_of[slot], ord);
                    atomicAdd(&s_cnt[(u32)((rec >> HT_SHIFT) & 3) * AG_CAP + slot], 1u);
                }
                if (rec & FLANK_BIT) {
                    atomicMin((u32 *)&s_ok[slot], ord);
                    atomicAdd(&s_cnt[4 * AG_CAP + slot], 1u);
                }
            } else {
                atomicMin((unsigned long long *)&s_of[slot], (unsigned long long)A.in_ord_full[ord]);
                atomicMin((unsigned long long *)&s_ok[slot], (unsigned long long)A.in_ord_flank[ord]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32 v = A.in_ht[(u64)ord * 4 + c];
                    if (v) atomicAdd(&s_cnt[c * AG_CAP + slot], v);
                }
                const u32 f = A.in_fl[ord];
                if (f) atomicAdd(&s_cnt[4 * AG_CAP + slot], f);
            }
        };
#pragma unroll
        for (int u = 0; u < AG_R; ++u)
            if (rec_r[u] != EMPTY_KEY) insert(rec_r[u], ord_r[u]);
        for (unsigned long long i = rb + tid + (unsigned long long)AG_R * AG_T; i < re; i += AG_T) insert(A.rec[i], A.ord[i]);
        lds_barrier();
        // The prefetched registers are pinned HERE, before this bucket's reservation atomic and output stores are issued:
        // vmcnt completes in order, so a wait for the prefetch placed after the stores (at the loop latch, where the compiler
        // would put it) would drain the stores too, every bucket.  Here the loads were issued a whole insert phase ago.
#pragma unroll
        for (int u = 0; u < AG_R; ++u) { asm volatile("" : "+v"(rec_n[u])); asm volatile("" : "+v"(ord_n[u])); }
        asm volatile("" : "+v"(rbn2)); asm volatile("" : "+v"(ren2));
        // compaction: contiguous chunk of slots per thread, block exclusive scan of the occupied counts
        const u32 per = tsize >= AG_T ? tsize / AG_T : 1;
        const u32 s0 = min(tsize, (u32)tid * per), s1 = min(tsize, s0 + per);
        u32 mine = 0;
        for (u32 s = s0; s < s1; ++s) mine += s_key[s] != EMPTY_KEY;
        u32 incl = mine;
#pragma unroll
        for (int o = 1; o < HHX_WAVE; o <<= 1) {
            const u32 v = __shfl_up(incl, o, HHX_WAVE);
            if (lane >= o) incl += v;
        }
        if (lane == HHX_WAVE - 1) s_scan[wave] = incl;
        lds_barrier();
        u32 woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < AG_T / HHX_WAVE; ++w) { if (w < wave) woff += s_scan[w]; total += s_scan[w]; }
        if (tid == 0) s_base = total ? atomicAdd(A.out_cursor, (unsigned long long)total) : 0ull;
        lds_barrier();
        unsigned long long o = s_base + woff + incl - mine;
        for (u32 s = s0; s < s1; ++s) {
            const u64 key = s_key[s];
            if (key == EMPTY_KEY) continue;
            A.o_key[o] = key;
            if constexpr (MODE == 0) {
                A.o_ord_full[o] = s_of[s] == ORD_NONE ? NO_ORD : A.ord_base + (u64)s_of[s];
                A.o_ord_flank[o] = s_ok[s] == ORD_NONE ? NO_ORD : A.ord_base + (u64)s_ok[s];
            } else {
                A.o_ord_full[o] = (u64)s_of[s];
                A.o_ord_flank[o] = (u64)s_ok[s];
            }
            *reinterpret_cast<uint4 *>(A.o_ht + o * 4) = make_uint4(s_cnt[s], s_cnt[AG_CAP + s], s_cnt[2 * AG_CAP + s], s_cnt[3 * AG_CAP + s]);   // one 16-byte store
            A.o_fl[o] = s_cnt[4 * AG_CAP + s];
            ++o;
        }
        lds_barrier();
        }   // sub-passes
        } else {   // empty bucket: same pins, so that no path reaches the latch with loads pending (see above)
#pragma unroll
            for (int u = 0; u < AG_R; ++u) { asm volatile("" : "+v"(rec_n[u])); asm volatile("" : "+v"(ord_n[u])); }
            asm volatile("" : "+v"(rbn2)); asm volatile("" : "+v"(ren2));
        }
        if (bn >= b1) break;
        b = bn; rb = rbn; re = ren; rbn = rbn2; ren = ren2;
#pragma unroll
        for (int u = 0; u < AG_R; ++u) { rec_r[u] = rec_n[u]; ord_r[u] = ord_n[u]; }
    }
}

template <int MODE>
constexpr size_t agg_lds_bytes() {
    return (size_t)AG_CAP * (8 + 2 * sizeof(typename AggLds<MODE>::ord_t) + 5 * 4) + 16 * 4 + 8;      // + s_scan[16] + the reservation slot
}

inline unsigned grid_for(u64 n, unsigned per = 256) {
    u64 b = (n + per - 1) / per;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)b;
}

// ---- host driver: records from `src` (n items, some may be dropped by the map) -> one aggregated run ----
struct Payload {            // MODE 1 only
    const u64 *ord_full = nullptr, *ord_flank = nullptr;
    const u32 *ht = nullptr, *fl = nullptr;
};

// bucket bits of the group-by of n_items records (attempt: after an LDS overflow, more buckets) and the radix bits per level.
// Up to 2048 records a bucket (k_aggregate takes such a bucket in two sub-passes over the L2-resident records) and up to 9 bits a level: 500 M
// pairs are 2^18 buckets = TWO radix levels of 9 bits instead of 2^19 = three of 6 + 6 + 7 — one pass of the 6 GB of records through HBM less.
// Measured at C3 (tools/ingest_probe.py, profiles/r06_ingest_levels_probe.json): 1024 / 7 (rounds 1-6) 39.5 ms, 2048 / 9 37.1 ms, 4096 / 9 38.5 ms
// (four sub-passes: the aggregation 7.6 -> 9.1 ms), 1024 / 9 40.1 ms (still three levels).
inline i64 ingest_per_bucket() { static const i64 v = getenv("HHX_ING_BUCKET") ? atoll(getenv("HHX_ING_BUCKET")) : 2048; return v; }
inline int ingest_level_bits() { static const int v = getenv("HHX_ING_LBITS") ? atoi(getenv("HHX_ING_LBITS")) : 9; return v; }
inline int ingest_total_bits(i64 n_items, int attempt) {
    int total_bits = 0;
    while ((n_items >> total_bits) > ingest_per_bucket() && total_bits < 24) ++total_bits;
    return std::min(24, total_bits + 2 * attempt);
}

// hist0: the level-1 histogram of the records counted by their producer for attempt 0's bucket bits (or null)
template <class Src, int MODE>
int build_run(const Src &src, i64 n_items, const Payload &pl, u64 ord_base, LinkRun **out, const unsigned long long *hist0 = nullptr) {
    static int attr_set = -1;           // the attribute is per device: keyed on the current ordinal
    int attr_dev = 0;
    HHX_HIP(hipGetDevice(&attr_dev));
    if (attr_set != attr_dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_aggregate<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HHX_HIP(hipFuncSetAttribute((const void *)k_aggregate<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = attr_dev;
    }
    LinkRun *run = new LinkRun();
    if (n_items <= 0) { *out = run; return run->alloc(0) ? (delete run, 1) : 0; }
    if (n_items >= ((i64)1 << 32) - 1) { delete run; return fail("ingest: at most 2^32 - 2 pairs per push (got %lld)", (long long)n_items); }
    for (int attempt = 0; attempt < 4; ++attempt) {
        // buckets of ~<= 1k records (half of the 2048-slot LDS table even if every record is a new key; larger
        // buckets are aggregated in sub-passes), radix levels of <= 7 bits: >= 32 records = 256 B per (tile, bucket)
        // run.  Measured at 500 M pairs: 3 levels of 7+6+6 bits 34 ms, 2 levels of 8+8 bits + 8 sub-passes 40 ms.
        const int level_bits = ingest_level_bits(), total_bits = ingest_total_bits(n_items, attempt);
        const DigKeyHash dig{total_bits};
        Partitioned<u32> part;
        { int rc = partition_records(src, dig, n_items, total_bits, level_bits, &part, "part", (const Src *)nullptr, attempt == 0 ? hist0 : nullptr);
          if (rc) { delete run; return rc; } }
        const i64 n_valid = part.n_valid;
        if (n_valid == 0) { *out = run; return run->alloc(0) ? (delete run, 1) : 0; }
        if (MODE == 0 && attempt == 0) prof_count("ingest_records", n_valid);
        const u32 n_buckets = part.n_buckets;
        // ---- aggregate
        const u32 n_wg = std::min<u32>(n_buckets, 1024);
        AggParams A{};
        A.rec = part.w0.p; A.ord = part.w1.p; A.base = (const unsigned long long *)part.base.p;
        A.n_buckets = n_buckets; A.buckets_per_wg = (n_buckets + n_wg - 1) / n_wg;
        A.total_bits = total_bits;
        A.ord_base = ord_base;
        A.in_ord_full = pl.ord_full; A.in_ord_flank = pl.ord_flank; A.in_ht = pl.ht; A.in_fl = pl.fl;
        // The run is written once: its arrays are sized for the worst case (every record a new key) and n is set to the
        // rows actually reserved.  (Round 1 wrote gapped segments and compacted them: 3.9 ms and 15 GB of traffic per 500 M pairs.)
        DevBuf<unsigned long long> out_cursor;
        DevBuf<unsigned int> overflow;
        if (run->alloc(n_valid) || out_cursor.alloc(1) || overflow.alloc(1)) { delete run; return 1; }
        HHX_HIP(hipMemsetAsync(overflow.p, 0, sizeof(unsigned int), g_stream));
        HHX_HIP(hipMemsetAsync(out_cursor.p, 0, sizeof(unsigned long long), g_stream));
        A.o_key = run->key.p; A.o_ord_full = run->ord_full.p; A.o_ord_flank = run->ord_flank.p; A.o_ht = run->ht.p; A.o_fl = run->fl.p;
        A.out_cursor = out_cursor.p; A.overflow = overflow.p;
        { KTimer kt("aggregate");
        k_aggregate<MODE><<<n_wg, AG_T, agg_lds_bytes<MODE>(), g_stream>>>(A); }
        HHX_LAUNCH_CHECK();
        unsigned long long n_keys = 0;
        unsigned int ov = 0;
        HHX_HIP(hipMemcpyAsync(&n_keys, out_cursor.p, sizeof n_keys, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipMemcpyAsync(&ov, overflow.p, sizeof ov, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        if (ov) continue;                                       // a sub-pass held more distinct keys than LDS: more buckets
        run->n = (i64)n_keys;
        *out = run;
        return 0;
    }
    delete run;
    return fail("ingest: LDS aggregation kept overflowing (pathological key distribution)");
}

// ---- insertion order: bitmap over ordinals + popcount prefix -----------------------------------------
__global__ __launch_bounds__(256) void k_mark_ord(const u64 *__restrict__ ord, i64 n, u64 *bitmap) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        if (ord[i] != NO_ORD) atomicOr((unsigned long long *)&bitmap[ord[i] >> 6], 1ull << (ord[i] & 63));
}
__global__ __launch_bounds__(256) void k_popc_words(const u64 *bitmap, i64 n_words, i64 *out) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (i64)gridDim.x * blockDim.x) out[i] = __popcll(bitmap[i]);
}
__device__ __forceinline__ i64 ord_rank(const u64 *bitmap, const i64 *prefix, u64 ord) {
    return prefix[ord >> 6] + __popcll(bitmap[ord >> 6] & ((1ull << (ord & 63)) - 1ull));
}
__global__ __launch_bounds__(256) void k_emit_full(const u64 *__restrict__ key, const u64 *__restrict__ ord, const u32 *__restrict__ ht, i64 n,
                                                   const u64 *bitmap, const i64 *prefix, i32 *out_i, i32 *out_j, i64 *out_cnt, i64 *out_ht) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        if (ord[i] == NO_ORD) continue;
        const i64 r = ord_rank(bitmap, prefix, ord[i]);
        out_i[r] = (i32)(key[i] >> ID_BITS);
        out_j[r] = (i32)(key[i] & ID_MASK);
        i64 tot = 0;
        for (int k = 0; k < 4; ++k) { out_ht[4 * r + k] = ht[4 * i + k]; tot += ht[4 * i + k]; }
        out_cnt[r] = tot;                                           // full_link_dict :1649 == sum of the HT counts
    }
}
__global__ __launch_bounds__(256) void k_emit_flank(const u64 *__restrict__ key, const u64 *__restrict__ ord, const u32 *__restrict__ fl, i64 n,
                                                    const u64 *bitmap, const i64 *prefix, i32 *out_i, i32 *out_j, i64 *out_cnt,
                                                    double *out_val, unsigned long long *frag_links) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        if (ord[i] == NO_ORD) continue;
        const i64 r = ord_rank(bitmap, prefix, ord[i]);
        const i32 fi = (i32)(key[i] >> ID_BITS), fj = (i32)(key[i] & ID_MASK);
        out_i[r] = fi;
        out_j[r] = fj;
        out_cnt[r] = (i64)fl[i];
        out_val[r] = (double)fl[i];
        atomicAdd(&frag_links[fi], (unsigned long long)fl[i]);      // ctg_link_dict / frag_link_dict :1638-1639
        atomicAdd(&frag_links[fj], (unsigned long long)fl[i]);
    }
}
// [0] rows with a full ordinal, [1] rows with a flank ordinal, [2] 1 + largest ordinal
__global__ __launch_bounds__(256) void k_run_stats(const u64 *__restrict__ of, const u64 *__restrict__ ok, i64 n, unsigned long long *out) {
    unsigned long long a = 0, b = 0, mx = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        if (of[i] != NO_ORD) { ++a; mx = max(mx, (unsigned long long)of[i] + 1); }
        if (ok[i] != NO_ORD) { ++b; mx = max(mx, (unsigned long long)ok[i] + 1); }
    }
    a = (unsigned long long)wave_sum_i64((i64)a);
    b = (unsigned long long)wave_sum_i64((i64)b);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned long long)__shfl_down((long long)mx, o, HHX_WAVE));
    if (lane_id() == 0) {
        if (a) atomicAdd(&out[0], a);
        if (b) atomicAdd(&out[1], b);
        if (mx) atomicMax(&out[2], mx);
    }
}

template <class T>
int upload(DevBuf<T> &d, const T *h, size_t n) {
    if (d.alloc(n)) return 1;
    if (n) HHX_HIP(hipMemcpyAsync(d.p, h, n * sizeof(T), hipMemcpyHostToDevice, g_stream));
    return 0;
}

int run_stats(const LinkRun *r, i64 *n_full, i64 *n_flank, u64 *ord_limit) {
    unsigned long long h[3] = {0, 0, 0};
    if (r && r->n) {
        DevBuf<unsigned long long> d;
        if (d.alloc(3)) return 1;
        HHX_HIP(hipMemsetAsync(d.p, 0, sizeof h, g_stream));
        k_run_stats<<<grid_for((u64)r->n), 256, 0, g_stream>>>(r->ord_full.p, r->ord_flank.p, r->n, d.p);
        HHX_LAUNCH_CHECK();
        HHX_HIP(hipMemcpyAsync(h, d.p, sizeof h, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
    }
    if (n_full) *n_full = (i64)h[0];
    if (n_flank) *n_flank = (i64)h[1];
    if (ord_limit) *ord_limit = std::max<u64>(*ord_limit, h[2]);
    return 0;
}

// merge the runs of one table into a single run (rows re-aggregated by key: counts add, ordinals take the minimum)
int merge_runs(std::vector<LinkRun *> &runs) {
    if (runs.size() <= 1) return 0;
    i64 total = 0;
    for (auto *r : runs) total += r->n;
    LinkRun cat;
    if (cat.alloc(total)) return 1;
    i64 off = 0;
    for (auto *r : runs) {
        if (r->n) {
            HHX_HIP(hipMemcpyAsync(cat.key.p + off, r->key.p, 8 * (size_t)r->n, hipMemcpyDeviceToDevice, g_stream));
            HHX_HIP(hipMemcpyAsync(cat.ord_full.p + off, r->ord_full.p, 8 * (size_t)r->n, hipMemcpyDeviceToDevice, g_stream));
            HHX_HIP(hipMemcpyAsync(cat.ord_flank.p + off, r->ord_flank.p, 8 * (size_t)r->n, hipMemcpyDeviceToDevice, g_stream));
            HHX_HIP(hipMemcpyAsync(cat.ht.p + 4 * off, r->ht.p, 16 * (size_t)r->n, hipMemcpyDeviceToDevice, g_stream));
            HHX_HIP(hipMemcpyAsync(cat.fl.p + off, r->fl.p, 4 * (size_t)r->n, hipMemcpyDeviceToDevice, g_stream));
        }
        off += r->n;
    }
    HHX_HIP(hipStreamSynchronize(g_stream));
    for (auto *r : runs) delete r;
    runs.clear();
    LinkRun *merged = nullptr;
    Payload pl;
    pl.ord_full = cat.ord_full.p; pl.ord_flank = cat.ord_flank.p; pl.ht = cat.ht.p; pl.fl = cat.fl.p;
    const SrcRows src{cat.key.p};
    HHX_TRY((build_run<SrcRows, 1>(src, total, pl, 0, &merged)));
    runs.push_back(merged);
    return 0;
}

// insertion-ordered host-visible tables (the S5 seam): rank the first-seen ordinals
int materialize(hhx_ingest *h) {
    OrderedTables &o = h->ordered;
    if (o.ready) return 0;
    const LinkRun *rf = h->table(0), *rk = h->table(1);
    const i64 n_words = (i64)(h->ord_limit / 64) + 1;
    DevBuf<u64> bitmap;
    DevBuf<i64> wcnt, prefix;
    if (bitmap.alloc((size_t)n_words) || wcnt.alloc((size_t)n_words) || prefix.alloc((size_t)n_words + 1)) return 1;
    if (o.full_i.alloc((size_t)h->n_full) || o.full_j.alloc((size_t)h->n_full) || o.full_cnt.alloc((size_t)h->n_full) ||
        o.ht.alloc((size_t)h->n_full * 4) || o.flank_i.alloc((size_t)h->n_flank) || o.flank_j.alloc((size_t)h->n_flank) ||
        o.flank_cnt.alloc((size_t)h->n_flank) || o.flank_val.alloc((size_t)h->n_flank) || o.frag_links.alloc((size_t)h->t.n_frag))
        return 1;
    HHX_HIP(hipMemsetAsync(o.frag_links.p, 0, sizeof(unsigned long long) * (size_t)h->t.n_frag, g_stream));
    if (rf && rf->n) {
        HHX_HIP(hipMemsetAsync(bitmap.p, 0, sizeof(u64) * (size_t)n_words, g_stream));
        k_mark_ord<<<grid_for((u64)rf->n), 256, 0, g_stream>>>(rf->ord_full.p, rf->n, bitmap.p);
        HHX_LAUNCH_CHECK();
        k_popc_words<<<grid_for((u64)n_words), 256, 0, g_stream>>>(bitmap.p, n_words, wcnt.p);
        HHX_LAUNCH_CHECK();
        HHX_TRY(exclusive_scan_i64(wcnt.p, prefix.p, n_words, nullptr));
        k_emit_full<<<grid_for((u64)rf->n), 256, 0, g_stream>>>(rf->key.p, rf->ord_full.p, rf->ht.p, rf->n, bitmap.p, prefix.p, o.full_i.p,
                                                                 o.full_j.p, o.full_cnt.p, o.ht.p);
        HHX_LAUNCH_CHECK();
    }
    if (rk && rk->n) {
        HHX_HIP(hipMemsetAsync(bitmap.p, 0, sizeof(u64) * (size_t)n_words, g_stream));
        k_mark_ord<<<grid_for((u64)rk->n), 256, 0, g_stream>>>(rk->ord_flank.p, rk->n, bitmap.p);
        HHX_LAUNCH_CHECK();
        k_popc_words<<<grid_for((u64)n_words), 256, 0, g_stream>>>(bitmap.p, n_words, wcnt.p);
        HHX_LAUNCH_CHECK();
        HHX_TRY(exclusive_scan_i64(wcnt.p, prefix.p, n_words, nullptr));
        k_emit_flank<<<grid_for((u64)rk->n), 256, 0, g_stream>>>(rk->key.p, rk->ord_flank.p, rk->fl.p, rk->n, bitmap.p, prefix.p, o.flank_i.p,
                                                                  o.flank_j.p, o.flank_cnt.p, o.flank_val.p, o.frag_links.p);
        HHX_LAUNCH_CHECK();
    }
    HHX_HIP(hipStreamSynchronize(g_stream));
    o.n_full = h->n_full; o.n_flank = h->n_flank;
    o.ready = true;
    return 0;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" int hhx_ingest_create(const hhx_ingest_config *cfg, hhx_ingest **out) {
    if (!cfg || !out) return fail("null pointer");
    if (cfg->n_ctg <= 0 || cfg->n_frag <= 0) return fail("empty contig table");
    if (cfg->bins && cfg->bin_size <= 0) return fail("bins mode needs bin_size > 0");
    if ((i64)cfg->n_ctg >= ((i64)1 << ID_BITS) || (i64)cfg->n_frag >= ((i64)1 << ID_BITS))
        return fail("ingest: more than 2^29 contigs / fragments");
    for (i32 c = 0; c < cfg->n_ctg; ++c)
        if (cfg->ctg_len[c] < 0 || cfg->ctg_len[c] > LEN_MASK || cfg->ctg_frag0[c] < 0 || cfg->ctg_frag0[c] >= cfg->n_frag)
            return fail("ingest: bad length / first fragment for contig %d", c);
    // one table serves both dicts when every contig is its own fragment (parse_alignments_for_ctgs)
    bool identity = !cfg->bins && cfg->n_ctg == cfg->n_frag;
    for (i32 c = 0; identity && c < cfg->n_ctg; ++c)
        identity = cfg->ctg_frag0[c] == c && cfg->frag_len[c] == cfg->ctg_len[c];
    std::vector<UnitInfo> ci((size_t)cfg->n_ctg), fi((size_t)cfg->n_frag);
    for (i32 f = 0; f < cfg->n_frag; ++f) {
        fi[f].rank = cfg->frag_rank[f];
        fi[f].aux = 0;
        fi[f].lenf = (cfg->frag_len[f] & LEN_MASK) | (cfg->frag_nx[f] ? NX_BIT : 0);
    }
    for (i32 c = 0; c < cfg->n_ctg; ++c) {
        ci[c].rank = cfg->ctg_rank[c];
        ci[c].aux = cfg->ctg_frag0[c];
        ci[c].lenf = cfg->ctg_len[c] | (cfg->ctg_split[c] ? SPLIT_BIT : 0) | (identity && cfg->frag_nx[c] ? NX_BIT : 0);
    }
    hhx_ingest *h = new hhx_ingest();
    for (i32 c = 0; c < cfg->n_ctg; ++c) h->max_ctg_len = std::max<i64>(h->max_ctg_len, cfg->ctg_len[c]);
    int rc = upload(h->ctg_info, ci.data(), ci.size()) || upload(h->frag_info, fi.data(), fi.size());
    if (rc) { delete h; return 1; }
    hipError_t e = hipStreamSynchronize(g_stream);               // the host vectors die at return
    if (e != hipSuccess) { delete h; return fail("ingest_create: %s", hipGetErrorString(e)); }
    h->t.ctg = h->ctg_info.p; h->t.frag = h->frag_info.p;
    h->t.n_ctg = cfg->n_ctg; h->t.n_frag = cfg->n_frag; h->t.bin_size = cfg->bin_size; h->t.flank = cfg->flank;
    h->t.bins = cfg->bins; h->t.skip_intra = cfg->skip_intra;
    h->combined = identity;
    *out = h;
    return 0;
}

extern "C" int hhx_ingest_set_ordinal_base(hhx_ingest *h, int64_t base) {
    if (!h) return fail("null handle");
    if (h->n_pushed || base < 0) return fail("hhx_ingest_set_ordinal_base: call before the first push, with base >= 0");
    h->ord_base = (u64)base;
    return 0;
}

extern "C" int hhx_ingest_keep_pairs(hhx_ingest *h, int on) {
    if (!h) return fail("null handle");
    if (h->n_pushed) return fail("hhx_ingest_keep_pairs: call before the first push");
    h->keep_pairs = on != 0;
    return 0;
}

extern "C" int hhx_ingest_keep_frag_pairs(hhx_ingest *h, int on) {
    if (!h) return fail("null handle");
    if (h->n_pushed) return fail("hhx_ingest_keep_frag_pairs: call before the first push");
    h->keep_frag_pairs = on != 0;
    return 0;
}

// every distinct (frag_i, frag_j) the stream produced (ctg_pair_to_frag :1731-1733), in no particular order
extern "C" int hhx_ingest_fetch_frag_pairs(hhx_ingest *h, i64 *n_pairs, i32 *frag_i, i32 *frag_j) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (!h->keep_frag_pairs) return fail("hhx_ingest_fetch_frag_pairs: hhx_ingest_keep_frag_pairs was not requested");
    const LinkRun *r = h->runs[2].empty() ? nullptr : h->runs[2][0];
    const i64 n = r ? r->n : 0;
    if (n_pairs) *n_pairs = n;
    if (!frag_i || !frag_j || !n) return 0;
    std::vector<u64> keys((size_t)n);
    HHX_HIP(hipMemcpyAsync(keys.data(), r->key.p, 8 * (size_t)n, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    for (i64 k = 0; k < n; ++k) { frag_i[k] = (i32)(keys[(size_t)k] >> ID_BITS); frag_j[k] = (i32)(keys[(size_t)k] & ID_MASK); }
    return 0;
}

__global__ __launch_bounds__(256) void k_max_i64(i64 n, const i64 *__restrict__ a, const i64 *__restrict__ b, unsigned long long *__restrict__ out) {
    unsigned long long m = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        m = max(m, (unsigned long long)max(a[i], (i64)0));
        m = max(m, (unsigned long long)max(b[i], (i64)0));
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned long long)__shfl_down((long long)m, o, HHX_WAVE));
    if (lane_id() == 0) atomicMax(out, m);
}

// one batch of pairs (device arrays): map -> group-by run(s) (+ the side records); POS = i32 or i64 positions
template <class POS>
static int ingest_push_device(hhx_ingest *h, i64 n_pairs, const i32 *id1, const POS *pos1, const i32 *id2, const POS *pos2) {
    const u64 ord0 = h->ord_base + h->n_pushed;
    { KTimer kt("ingest");
    DevBuf<u64> mapped;
    DevBuf<unsigned long long> hist0;
    if (mapped.alloc((size_t)n_pairs) || hist0.alloc(P_MAX_BINS)) return 1;
    // the level-1 histogram of the group-by comes out of the map pass (build_run's first attempt; same bits as it derives)
    int lbits[4];
    const int total_bits = ingest_total_bits(n_pairs, 0);
    const int n_levels = part_levels(total_bits, std::min(ingest_level_bits(), 9), lbits);
    const bool fuse = total_bits > 0 && n_levels >= 1 && lbits[0] > 0 && !getenv("HHX_ING_NO_FUSED_COUNT");
    const MapHist H{fuse ? hist0.p : nullptr, total_bits, total_bits - lbits[0], 1 << lbits[0]};
    for (int stream = 0; stream < 3; ++stream) {
        if (stream == 1 && h->combined) continue;
        if (stream == 2 && !h->keep_frag_pairs) continue;
        if (fuse) HHX_HIP(hipMemsetAsync(hist0.p, 0, sizeof(unsigned long long) * P_MAX_BINS, g_stream));
        { KTimer kt2("map");
        if (h->combined && stream == 0) launch_map(SrcPairs<true, POS>{id1, pos1, id2, pos2, h->t, 0}, n_pairs, mapped.p, H);
        else launch_map(SrcPairs<false, POS>{id1, pos1, id2, pos2, h->t, stream}, n_pairs, mapped.p, H); }
        HHX_LAUNCH_CHECK();
        LinkRun *run = nullptr;
        HHX_TRY((build_run<SrcMapped, 0>(SrcMapped{mapped.p}, n_pairs, Payload(), ord0, &run, fuse ? hist0.p : nullptr)));
        h->runs[stream].push_back(run);
    } }
    if (h->keep_pairs) HHX_TRY(hhx_side_records_push<POS>(h, n_pairs, id1, pos1, id2, pos2));
    h->n_pushed += (u64)n_pairs;
    h->ord_limit = std::max<u64>(h->ord_limit, h->ord_base + h->n_pushed);
    return 0;
}

extern "C" int hhx_ingest_push(hhx_ingest *h, i64 n_pairs, const i32 *id1, const i32 *pos1, const i32 *id2, const i32 *pos2,
                               int on_device) {
    if (!h) return fail("null handle");
    if (h->finalized) return fail("ingest handle already finalized");
    if (n_pairs <= 0) return 0;
    const i32 *src[4] = {id1, pos1, id2, pos2};
    if (!on_device) {
        for (int k = 0; k < 4; ++k) {
            if (h->stage[k].n < (size_t)n_pairs && h->stage[k].alloc((size_t)n_pairs)) return 1;
            HHX_HIP(hipMemcpyAsync(h->stage[k].p, src[k], sizeof(i32) * (size_t)n_pairs, hipMemcpyHostToDevice, g_stream));
            src[k] = h->stage[k].p;
        }
    }
    return ingest_push_device<i32>(h, n_pairs, src[0], src[1], src[2], src[3]);
}

// the same with 64-bit positions: contigs of 2^31 bp and more, where the reference switches its coordinate arrays to int64
// (determine_int_type :116-147).  The CLM / coordinate side records (hhx_ingest_keep_pairs) hold 32-bit coordinates: with them
// switched on the positions must stay below 2^32 - 1 (checked here, loudly).
extern "C" int hhx_ingest_push64(hhx_ingest *h, i64 n_pairs, const i32 *id1, const i64 *pos1, const i32 *id2, const i64 *pos2, int on_device) {
    if (!h) return fail("null handle");
    if (h->finalized) return fail("ingest handle already finalized");
    if (n_pairs <= 0) return 0;
    const i32 *ids[2] = {id1, id2};
    const i64 *pos[2] = {pos1, pos2};
    if (!on_device) {
        for (int k = 0; k < 2; ++k) {
            if (h->stage[2 * k].n < (size_t)n_pairs && h->stage[2 * k].alloc((size_t)n_pairs)) return 1;
            if (h->stage64[k].n < (size_t)n_pairs && h->stage64[k].alloc((size_t)n_pairs)) return 1;
            HHX_HIP(hipMemcpyAsync(h->stage[2 * k].p, ids[k], sizeof(i32) * (size_t)n_pairs, hipMemcpyHostToDevice, g_stream));
            HHX_HIP(hipMemcpyAsync(h->stage64[k].p, pos[k], sizeof(i64) * (size_t)n_pairs, hipMemcpyHostToDevice, g_stream));
            ids[k] = h->stage[2 * k].p;
            pos[k] = h->stage64[k].p;
        }
    }
    if (h->keep_pairs) {
        DevBuf<unsigned long long> mx;
        if (mx.alloc(1)) return 1;
        HHX_HIP(hipMemsetAsync(mx.p, 0, sizeof(unsigned long long), g_stream));
        k_max_i64<<<(unsigned)std::max<i64>(1, std::min<i64>((n_pairs + 255) / 256, 4096)), 256, 0, g_stream>>>(n_pairs, pos[0], pos[1], mx.p);
        HHX_LAUNCH_CHECK();
        unsigned long long m = 0;
        HHX_HIP(hipMemcpyAsync(&m, mx.p, sizeof m, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        if (m >= 0xfffffffeull) return fail("hhx_ingest_push64: position %llu with the CLM / coordinate side records on (32-bit coordinates: contigs below 2^32 bp)", m);
    }
    return ingest_push_device<i64>(h, n_pairs, ids[0], pos[0], ids[1], pos[1]);
}

extern "C" int hhx_ingest_push_table(hhx_ingest *h, int which, i64 n_rows, const uint64_t *key, const uint64_t *ord_full,
                                     const uint64_t *ord_flank, const uint32_t *ht, const uint32_t *flank) {
    if (!h) return fail("null handle");
    if (h->finalized) return fail("ingest handle already finalized");
    if (n_rows < 0 || which < 0 || which > 1) return fail("hhx_ingest_push_table: bad argument");
    if (n_rows == 0) return 0;
    LinkRun *r = new LinkRun();
    if (r->alloc(n_rows)) { delete r; return 1; }
    hipError_t e = hipMemcpyAsync(r->key.p, key, 8 * (size_t)n_rows, hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r->ord_full.p, ord_full, 8 * (size_t)n_rows, hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r->ord_flank.p, ord_flank, 8 * (size_t)n_rows, hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r->ht.p, ht, 16 * (size_t)n_rows, hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r->fl.p, flank, 4 * (size_t)n_rows, hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { delete r; return fail("hhx_ingest_push_table: %s", hipGetErrorString(e)); }
    h->runs[(h->combined || which == 0) ? 0 : 1].push_back(r);
    return 0;
}

extern "C" int hhx_ingest_finalize(hhx_ingest *h, i64 *n_full_keys, i64 *n_flank_keys) {
    if (!h) return fail("null handle");
    if (!h->finalized) {
        { KTimer kt("ingest_merge");
        HHX_TRY(merge_runs(h->runs[0]));
        HHX_TRY(merge_runs(h->runs[1]));
        HHX_TRY(merge_runs(h->runs[2])); }
        i64 a = 0, b = 0, c = 0, d = 0;
        HHX_TRY(run_stats(h->table(0), &a, &b, &h->ord_limit));
        if (!h->combined) HHX_TRY(run_stats(h->table(1), &c, &d, &h->ord_limit));
        h->n_full = a;
        h->n_flank = h->combined ? b : d;
        for (auto &s : h->stage) s.release();
        for (auto &s : h->stage64) s.release();
        h->finalized = true;
    }
    if (n_full_keys) *n_full_keys = h->n_full;
    if (n_flank_keys) *n_flank_keys = h->n_flank;
    return 0;
}

extern "C" int hhx_ingest_table_device(hhx_ingest *h, int which, i64 *n_rows, void **key, void **ord_full, void **ord_flank,
                                       void **ht, void **flank) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    const LinkRun *r = h->table(which);
    if (n_rows) *n_rows = r ? r->n : 0;
    if (key) *key = r ? r->key.p : nullptr;
    if (ord_full) *ord_full = r ? r->ord_full.p : nullptr;
    if (ord_flank) *ord_flank = r ? r->ord_flank.p : nullptr;
    if (ht) *ht = r ? r->ht.p : nullptr;
    if (flank) *flank = r ? r->fl.p : nullptr;
    return 0;
}

extern "C" int hhx_ingest_fetch(hhx_ingest *h, i32 *full_i, i32 *full_j, i64 *full_cnt, i64 *ht_cnt, i32 *flank_i, i32 *flank_j,
                                i64 *flank_cnt, i64 *frag_links) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    HHX_TRY(materialize(h));
    const OrderedTables &o = h->ordered;
    const size_t nf = (size_t)o.n_full, nk = (size_t)o.n_flank;
    if (full_i && nf) HHX_HIP(hipMemcpyAsync(full_i, o.full_i.p, 4 * nf, hipMemcpyDeviceToHost, g_stream));
    if (full_j && nf) HHX_HIP(hipMemcpyAsync(full_j, o.full_j.p, 4 * nf, hipMemcpyDeviceToHost, g_stream));
    if (full_cnt && nf) HHX_HIP(hipMemcpyAsync(full_cnt, o.full_cnt.p, 8 * nf, hipMemcpyDeviceToHost, g_stream));
    if (ht_cnt && nf) HHX_HIP(hipMemcpyAsync(ht_cnt, o.ht.p, 32 * nf, hipMemcpyDeviceToHost, g_stream));
    if (flank_i && nk) HHX_HIP(hipMemcpyAsync(flank_i, o.flank_i.p, 4 * nk, hipMemcpyDeviceToHost, g_stream));
    if (flank_j && nk) HHX_HIP(hipMemcpyAsync(flank_j, o.flank_j.p, 4 * nk, hipMemcpyDeviceToHost, g_stream));
    if (flank_cnt && nk) HHX_HIP(hipMemcpyAsync(flank_cnt, o.flank_cnt.p, 8 * nk, hipMemcpyDeviceToHost, g_stream));
    if (frag_links) HHX_HIP(hipMemcpyAsync(frag_links, o.frag_links.p, 8 * (size_t)h->t.n_frag, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

int hhx_ingest_ordered_full_device(hhx_ingest *h, const i32 **fi, const i32 **fj) {
    HHX_TRY(materialize(h));
    *fi = h->ordered.full_i.p;
    *fj = h->ordered.full_j.p;
    return 0;
}

extern "C" int hhx_ingest_flank_device(hhx_ingest *h, void **fi, void **fj, void **val) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    HHX_TRY(materialize(h));
    if (fi) *fi = h->ordered.flank_i.p;
    if (fj) *fj = h->ordered.flank_j.p;
    if (val) *val = h->ordered.flank_val.p;
    return 0;
}

extern "C" int hhx_ingest_flank_count_device(hhx_ingest *h, void **cnt_i64) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    HHX_TRY(materialize(h));
    if (cnt_i64) *cnt_i64 = h->ordered.flank_cnt.p;
    return 0;
}

extern "C" int hhx_ingest_link_matrix(hhx_ingest *h, const uint8_t *in_set_host, int32_t n_rest, int add_self_loops,
                                      int32_t *frag_index_host, int32_t *n_linked, hhx_csr **out) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    KTimer kt("link_matrix");
    return hhx_link_matrix_from_run(h->table(1), h->t.n_frag, h->ord_limit, in_set_host, n_rest, add_self_loops, frag_index_host, n_linked, out);
}

extern "C" int hhx_ingest_destroy(hhx_ingest *h) {
    if (h) files_wait_handle(h);             // a queued file (hhx_jobs.hip) still reads the tables / the kept pairs
    delete h;
    return 0;
}
