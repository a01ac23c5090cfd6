// Hi-C read-pair binning into the contig x contig (full / HT) and fragment x fragment (flank) link
// tables, and dict_to_matrix.  Reference: scripts/HapHiC_cluster.py:1596-1752 (parse_alignments*),
// :299-307 (is_flank), :404-416 (update_HT_link_dict), :310-373 (dict_to_matrix).
//
// Design (HBM-bound integer work, no sort):
//   * one thread per read pair, 16 B of coalesced input (4 x int32 SoA streams);
//   * two open-addressing hash tables in HBM with 32-byte slots (key, first-seen ordinal, counters)
//     so that all atomics of one pair on one table land in ONE 32-byte sector; the tables are sized
//     for a load factor <= 0.5 out of 288 GB, never rehashed inside a batch;
//   * counts are integer atomics (order-free, deterministic); the Python dict INSERTION ORDER that
//     dict_to_matrix's index assignment depends on (:337-349) is recovered exactly from an atomicMin
//     of the pair's stream ordinal per key: a bitmap over ordinals + a popcount prefix scan gives each
//     key its rank in first-seen order (no sort of the keys).
#include "hhx_common.h"

using namespace hhx;

int hhx_csr_alloc_internal(i32 n_rows, i32 n_cols, i64 nnz, hhx_csr **out);

namespace {

constexpr u64 EMPTY_KEY = ~0ull;

struct __attribute__((aligned(32))) FullSlot {   // contig pair
    u64 key;        // (ctg_i << 32) | ctg_j
    u64 ord;        // min stream ordinal
    u32 ht[4];      // [HH, HT, TH, TT]
};
struct __attribute__((aligned(32))) FlankSlot {  // fragment pair
    u64 key;
    u64 ord;
    u64 cnt;
    u64 pad;
};

__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// returns slot index or ~0 on a full table; *n_keys counts fresh insertions
template <class Slot>
__device__ __forceinline__ u64 find_or_insert(Slot *tab, u64 mask, u64 key, unsigned long long *n_keys) {
    u64 h = mix64(key) & mask;
    for (u64 probe = 0; probe <= mask; ++probe) {
        // keys never change once set: a stale (non-coherent) read can only show EMPTY, and then the
        // device-scope CAS below returns the true occupant.
        u64 cur = *(volatile u64 *)&tab[h].key;
        if (cur == key) return h;
        if (cur == EMPTY_KEY) {
            u64 old = atomicCAS((unsigned long long *)&tab[h].key, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (old == EMPTY_KEY) { atomicAdd(n_keys, 1ull); return h; }
            if (old == key) return h;
        }
        h = (h + 1) & mask;
    }
    return ~0ull;
}

struct DevTables {           // device-resident copies of hhx_ingest_config arrays
    const i32 *ctg_rank;
    const i64 *ctg_len;
    const i32 *ctg_frag0;
    const unsigned char *ctg_split;
    const i32 *frag_rank;
    const i64 *frag_len;
    const unsigned char *frag_nx;
    i32 n_ctg, n_frag;
    i64 bin_size, flank;
    i32 bins, skip_intra;
};

__device__ __forceinline__ bool is_flank(i64 coord, i64 length, i64 flank) {   // :299-307
    return flank == 0 || coord <= flank || coord > length - flank;
}

__global__ __launch_bounds__(256) void k_ingest(i64 n_pairs, const i32 *__restrict__ id1, const i32 *__restrict__ pos1,
                                                const i32 *__restrict__ id2, const i32 *__restrict__ pos2, u64 ord0,
                                                DevTables t, FullSlot *full, u64 full_mask, FlankSlot *flank, u64 flank_mask,
                                                unsigned long long *counters /* [0]=full keys [1]=flank keys [2]=overflow */) {
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < n_pairs; idx += (i64)gridDim.x * blockDim.x) {
        const i32 r = id1[idx], m = id2[idx];
        if (t.skip_intra && r == m) continue;                                   // pairs_generator_inter_ctgs :1582
        if (t.bins && r == m && (r < 0 || r >= t.n_ctg || !t.ctg_split[r])) continue;   // :1699
        if (r < 0 || m < 0 || r >= t.n_ctg || m >= t.n_ctg) continue;          // :1625 / :1702
        i32 ci = r, cj = m;
        i64 xi = (i64)pos1[idx] + 1, xj = (i64)pos2[idx] + 1;                   // 1-based, :1629
        if (t.ctg_rank[r] > t.ctg_rank[m] || (r == m && xi > xj)) { ci = m; cj = r; const i64 tx = xi; xi = xj; xj = tx; }
        i32 fi = t.ctg_frag0[ci], fj = t.ctg_frag0[cj];
        i64 yi = xi, yj = xj;
        if (t.bins) {                                                           // convert_frags :1662-1670
            if (t.ctg_split[ci]) { const i64 nb = (xi + t.bin_size - 1) / t.bin_size; fi += (i32)(nb - 1); yi = xi - (nb - 1) * t.bin_size; }
            if (t.ctg_split[cj]) { const i64 nb = (xj + t.bin_size - 1) / t.bin_size; fj += (i32)(nb - 1); yj = xj - (nb - 1) * t.bin_size; }
            if (fi == fj) continue;                                             // :1715
            if (t.frag_rank[fi] > t.frag_rank[fj]) { const i32 tf = fi; fi = fj; fj = tf; const i64 ty = yi; yi = yj; yj = ty; }   // :1719-1720
        }
        const u64 ord = ord0 + (u64)idx;
        if (t.frag_nx[fi] && t.frag_nx[fj] && is_flank(yi, t.frag_len[fi], t.flank) && is_flank(yj, t.frag_len[fj], t.flank)) {
            const u64 s = find_or_insert(flank, flank_mask, ((u64)(u32)fi << 32) | (u64)(u32)fj, &counters[1]);
            if (s == ~0ull) { atomicAdd(&counters[2], 1ull); continue; }
            atomicMin((unsigned long long *)&flank[s].ord, (unsigned long long)ord);
            atomicAdd((unsigned long long *)&flank[s].cnt, 1ull);
        }
        if (t.bins && r == m) continue;                                         // :1736
        const u64 s = find_or_insert(full, full_mask, ((u64)(u32)ci << 32) | (u64)(u32)cj, &counters[0]);
        if (s == ~0ull) { atomicAdd(&counters[2], 1ull); continue; }
        atomicMin((unsigned long long *)&full[s].ord, (unsigned long long)ord);
        const int ti = xi * 2 > t.ctg_len[ci], tj = xj * 2 > t.ctg_len[cj];    // :404-416
        atomicAdd(&full[s].ht[ti * 2 + tj], 1u);
    }
}

template <class Slot>
__global__ __launch_bounds__(256) void k_clear(Slot *tab, u64 cap) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (u64)gridDim.x * blockDim.x) {
        Slot s;
        memset(&s, 0, sizeof s);
        s.key = EMPTY_KEY;
        s.ord = ~0ull;
        tab[i] = s;
    }
}

__global__ __launch_bounds__(256) void k_rehash_full(const FullSlot *old, u64 old_cap, FullSlot *nw, u64 mask, unsigned long long *dummy) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < old_cap; i += (u64)gridDim.x * blockDim.x) {
        if (old[i].key == EMPTY_KEY) continue;
        const u64 s = find_or_insert(nw, mask, old[i].key, dummy);
        nw[s].ord = old[i].ord;                                   // one writer per key
        for (int k = 0; k < 4; ++k) nw[s].ht[k] = old[i].ht[k];
    }
}
__global__ __launch_bounds__(256) void k_rehash_flank(const FlankSlot *old, u64 old_cap, FlankSlot *nw, u64 mask, unsigned long long *dummy) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < old_cap; i += (u64)gridDim.x * blockDim.x) {
        if (old[i].key == EMPTY_KEY) continue;
        const u64 s = find_or_insert(nw, mask, old[i].key, dummy);
        nw[s].ord = old[i].ord;
        nw[s].cnt = old[i].cnt;
    }
}

// ---- insertion order: bitmap over ordinals + popcount prefix -----------------------------------
template <class Slot>
__global__ __launch_bounds__(256) void k_mark_ord(const Slot *tab, u64 cap, u64 *bitmap) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (u64)gridDim.x * blockDim.x)
        if (tab[i].key != EMPTY_KEY) atomicOr((unsigned long long *)&bitmap[tab[i].ord >> 6], 1ull << (tab[i].ord & 63));
}
__global__ __launch_bounds__(256) void k_popc_words(const u64 *bitmap, i64 n_words, i64 *out) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (i64)gridDim.x * blockDim.x) out[i] = __popcll(bitmap[i]);
}
__device__ __forceinline__ i64 ord_rank(const u64 *bitmap, const i64 *prefix, u64 ord) {
    return prefix[ord >> 6] + __popcll(bitmap[ord >> 6] & ((1ull << (ord & 63)) - 1ull));
}
__global__ __launch_bounds__(256) void k_emit_full(const FullSlot *tab, u64 cap, const u64 *bitmap, const i64 *prefix,
                                                   i32 *out_i, i32 *out_j, i64 *out_cnt, i64 *out_ht) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (u64)gridDim.x * blockDim.x) {
        if (tab[i].key == EMPTY_KEY) continue;
        const i64 r = ord_rank(bitmap, prefix, tab[i].ord);
        out_i[r] = (i32)(tab[i].key >> 32);
        out_j[r] = (i32)(tab[i].key & 0xffffffffu);
        i64 tot = 0;
        for (int k = 0; k < 4; ++k) { out_ht[4 * r + k] = tab[i].ht[k]; tot += tab[i].ht[k]; }
        out_cnt[r] = tot;                                           // full_link_dict :1649 == sum of the HT counts
    }
}
__global__ __launch_bounds__(256) void k_emit_flank(const FlankSlot *tab, u64 cap, const u64 *bitmap, const i64 *prefix,
                                                    i32 *out_i, i32 *out_j, i64 *out_cnt, double *out_val,
                                                    unsigned long long *frag_links) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (u64)gridDim.x * blockDim.x) {
        if (tab[i].key == EMPTY_KEY) continue;
        const i64 r = ord_rank(bitmap, prefix, tab[i].ord);
        const i32 fi = (i32)(tab[i].key >> 32), fj = (i32)(tab[i].key & 0xffffffffu);
        out_i[r] = fi;
        out_j[r] = fj;
        out_cnt[r] = (i64)tab[i].cnt;
        out_val[r] = (double)tab[i].cnt;
        atomicAdd(&frag_links[fi], (unsigned long long)tab[i].cnt);  // ctg_link_dict / frag_link_dict :1638-1639
        atomicAdd(&frag_links[fj], (unsigned long long)tab[i].cnt);
    }
}

inline unsigned grid_for(u64 n) {
    u64 b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)b;
}

u64 next_pow2(u64 x) {
    u64 p = 1024;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace

struct hhx_ingest {
    DevTables t{};
    DevBuf<i32> ctg_rank, ctg_frag0, frag_rank;
    DevBuf<i64> ctg_len, frag_len;
    DevBuf<unsigned char> ctg_split, frag_nx;
    DevBuf<FullSlot> full;
    DevBuf<FlankSlot> flank;
    u64 full_cap = 0, flank_cap = 0;
    DevBuf<unsigned long long> counters;       // [0] full keys, [1] flank keys, [2] overflow, [3] scratch
    u64 n_pushed = 0;                          // stream ordinal of the next pair
    u64 max_keys_bound = 0;
    bool finalized = false;
    bool hint_given = false;
    i64 n_full = 0, n_flank = 0;
    DevBuf<i32> out_full_i, out_full_j, out_flank_i, out_flank_j;
    DevBuf<i64> out_full_cnt, out_ht, out_flank_cnt;
    DevBuf<double> out_flank_val;
    DevBuf<unsigned long long> frag_links;
    // staging for host-side inputs
    DevBuf<i32> stage[4];
};

template <class T>
static int upload(DevBuf<T> &d, const T *h, size_t n) {
    if (d.alloc(n)) return 1;
    if (n) HHX_HIP(hipMemcpyAsync(d.p, h, n * sizeof(T), hipMemcpyHostToDevice, g_stream));
    return 0;
}

static int ensure_capacity(hhx_ingest *h, u64 need_full, u64 need_flank) {
    // (re)allocate the tables so that `need` keys keep the load factor <= 0.5
    auto grow_full = [&](u64 cap) -> int {
        DevBuf<FullSlot> nw;
        if (nw.alloc(cap)) return 1;
        k_clear<FullSlot><<<grid_for(cap), 256, 0, g_stream>>>(nw.p, cap);
        HHX_LAUNCH_CHECK();
        if (h->full_cap) {
            k_rehash_full<<<grid_for(h->full_cap), 256, 0, g_stream>>>(h->full.p, h->full_cap, nw.p, cap - 1, h->counters.p + 3);
            HHX_LAUNCH_CHECK();
        }
        h->full = std::move(nw);
        h->full_cap = cap;
        return 0;
    };
    auto grow_flank = [&](u64 cap) -> int {
        DevBuf<FlankSlot> nw;
        if (nw.alloc(cap)) return 1;
        k_clear<FlankSlot><<<grid_for(cap), 256, 0, g_stream>>>(nw.p, cap);
        HHX_LAUNCH_CHECK();
        if (h->flank_cap) {
            k_rehash_flank<<<grid_for(h->flank_cap), 256, 0, g_stream>>>(h->flank.p, h->flank_cap, nw.p, cap - 1, h->counters.p + 3);
            HHX_LAUNCH_CHECK();
        }
        h->flank = std::move(nw);
        h->flank_cap = cap;
        return 0;
    };
    if (need_full * 2 > h->full_cap) HHX_TRY(grow_full(next_pow2(need_full * 2)));
    if (need_flank * 2 > h->flank_cap) HHX_TRY(grow_flank(next_pow2(need_flank * 2)));
    return 0;
}

extern "C" int hhx_ingest_create(const hhx_ingest_config *cfg, hhx_ingest **out) {
    if (!cfg || !out) return fail("null pointer");
    if (cfg->n_ctg <= 0 || cfg->n_frag <= 0) return fail("empty contig table");
    if (cfg->bins && cfg->bin_size <= 0) return fail("bins mode needs bin_size > 0");
    hhx_ingest *h = new hhx_ingest();
    int rc = upload(h->ctg_rank, cfg->ctg_rank, (size_t)cfg->n_ctg) || upload(h->ctg_len, cfg->ctg_len, (size_t)cfg->n_ctg) ||
             upload(h->ctg_frag0, cfg->ctg_frag0, (size_t)cfg->n_ctg) || upload(h->ctg_split, cfg->ctg_split, (size_t)cfg->n_ctg) ||
             upload(h->frag_rank, cfg->frag_rank, (size_t)cfg->n_frag) || upload(h->frag_len, cfg->frag_len, (size_t)cfg->n_frag) ||
             upload(h->frag_nx, cfg->frag_nx, (size_t)cfg->n_frag) || h->counters.alloc(4);
    if (rc) { delete h; return 1; }
    hipError_t e = hipMemsetAsync(h->counters.p, 0, 4 * sizeof(unsigned long long), g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { delete h; return fail("ingest_create: %s", hipGetErrorString(e)); }
    h->t.ctg_rank = h->ctg_rank.p; h->t.ctg_len = h->ctg_len.p; h->t.ctg_frag0 = h->ctg_frag0.p; h->t.ctg_split = h->ctg_split.p;
    h->t.frag_rank = h->frag_rank.p; h->t.frag_len = h->frag_len.p; h->t.frag_nx = h->frag_nx.p;
    h->t.n_ctg = cfg->n_ctg; h->t.n_frag = cfg->n_frag; h->t.bin_size = cfg->bin_size; h->t.flank = cfg->flank;
    h->t.bins = cfg->bins; h->t.skip_intra = cfg->skip_intra;
    // distinct unordered pairs (with the diagonal) bound the key count
    const double nb = (double)cfg->n_frag * ((double)cfg->n_frag + 1) / 2;
    h->max_keys_bound = nb > 9e18 ? ~0ull : (u64)nb;
    h->hint_given = cfg->expected_keys > 0;
    u64 hint = cfg->expected_keys > 0 ? (u64)cfg->expected_keys : (u64)1 << 16;
    if (hint > h->max_keys_bound) hint = h->max_keys_bound;
    rc = ensure_capacity(h, hint, hint);
    if (rc) { delete h; return rc; }
    *out = h;
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
    // capacity: with a caller hint the tables keep their size (an undersized hint overflows and
    // hhx_ingest_finalize reports it); without one every pair of the batch could be a new key,
    // bounded by the number of distinct fragment pairs.
    if (!h->hint_given) {
        unsigned long long c[4];
        HHX_HIP(hipMemcpyAsync(c, h->counters.p, sizeof c, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        const u64 need_full = std::min<u64>(c[0] + (u64)n_pairs, h->max_keys_bound);
        const u64 need_flank = std::min<u64>(c[1] + (u64)n_pairs, h->max_keys_bound);
        HHX_TRY(ensure_capacity(h, need_full, need_flank));
    }
    { KTimer kt("ingest");
    k_ingest<<<grid_for((u64)n_pairs), 256, 0, g_stream>>>(n_pairs, src[0], src[1], src[2], src[3], h->n_pushed, h->t, h->full.p,
                                                            h->full_cap - 1, h->flank.p, h->flank_cap - 1, h->counters.p); }
    HHX_LAUNCH_CHECK();
    h->n_pushed += (u64)n_pairs;
    if (!on_device) HHX_HIP(hipStreamSynchronize(g_stream));     // the staging buffers are reused by the next push
    return 0;
}

extern "C" int hhx_ingest_finalize(hhx_ingest *h, i64 *n_full_keys, i64 *n_flank_keys) {
    if (!h) return fail("null handle");
    if (!h->finalized) {
        unsigned long long c[4];
        HHX_HIP(hipMemcpyAsync(c, h->counters.p, sizeof c, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        if (c[2]) return fail("ingest: hash table overflow (%llu pairs dropped): raise expected_keys or push smaller batches", c[2]);
        h->n_full = (i64)c[0];
        h->n_flank = (i64)c[1];
        const i64 n_words = (i64)(h->n_pushed / 64) + 1;
        DevBuf<u64> bitmap;
        DevBuf<i64> wcnt, prefix;
        if (bitmap.alloc((size_t)n_words) || wcnt.alloc((size_t)n_words) || prefix.alloc((size_t)n_words + 1)) return 1;
        if (h->out_full_i.alloc((size_t)h->n_full) || h->out_full_j.alloc((size_t)h->n_full) || h->out_full_cnt.alloc((size_t)h->n_full) ||
            h->out_ht.alloc((size_t)h->n_full * 4) || h->out_flank_i.alloc((size_t)h->n_flank) || h->out_flank_j.alloc((size_t)h->n_flank) ||
            h->out_flank_cnt.alloc((size_t)h->n_flank) || h->out_flank_val.alloc((size_t)h->n_flank) || h->frag_links.alloc((size_t)h->t.n_frag))
            return 1;
        HHX_HIP(hipMemsetAsync(h->frag_links.p, 0, sizeof(unsigned long long) * (size_t)h->t.n_frag, g_stream));
        // full table
        HHX_HIP(hipMemsetAsync(bitmap.p, 0, sizeof(u64) * (size_t)n_words, g_stream));
        k_mark_ord<FullSlot><<<grid_for(h->full_cap), 256, 0, g_stream>>>(h->full.p, h->full_cap, bitmap.p);
        HHX_LAUNCH_CHECK();
        k_popc_words<<<grid_for((u64)n_words), 256, 0, g_stream>>>(bitmap.p, n_words, wcnt.p);
        HHX_LAUNCH_CHECK();
        HHX_TRY(exclusive_scan_i64(wcnt.p, prefix.p, n_words, nullptr));
        k_emit_full<<<grid_for(h->full_cap), 256, 0, g_stream>>>(h->full.p, h->full_cap, bitmap.p, prefix.p, h->out_full_i.p,
                                                                   h->out_full_j.p, h->out_full_cnt.p, h->out_ht.p);
        HHX_LAUNCH_CHECK();
        // flank table
        HHX_HIP(hipMemsetAsync(bitmap.p, 0, sizeof(u64) * (size_t)n_words, g_stream));
        k_mark_ord<FlankSlot><<<grid_for(h->flank_cap), 256, 0, g_stream>>>(h->flank.p, h->flank_cap, bitmap.p);
        HHX_LAUNCH_CHECK();
        k_popc_words<<<grid_for((u64)n_words), 256, 0, g_stream>>>(bitmap.p, n_words, wcnt.p);
        HHX_LAUNCH_CHECK();
        HHX_TRY(exclusive_scan_i64(wcnt.p, prefix.p, n_words, nullptr));
        k_emit_flank<<<grid_for(h->flank_cap), 256, 0, g_stream>>>(h->flank.p, h->flank_cap, bitmap.p, prefix.p, h->out_flank_i.p,
                                                                     h->out_flank_j.p, h->out_flank_cnt.p, h->out_flank_val.p,
                                                                     h->frag_links.p);
        HHX_LAUNCH_CHECK();
        HHX_HIP(hipStreamSynchronize(g_stream));
        // the tables are no longer needed
        h->full.release();
        h->flank.release();
        h->finalized = true;
    }
    if (n_full_keys) *n_full_keys = h->n_full;
    if (n_flank_keys) *n_flank_keys = h->n_flank;
    return 0;
}

extern "C" int hhx_ingest_fetch(hhx_ingest *h, i32 *full_i, i32 *full_j, i64 *full_cnt, i64 *ht_cnt, i32 *flank_i, i32 *flank_j,
                                i64 *flank_cnt, i64 *frag_links) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    const size_t nf = (size_t)h->n_full, nk = (size_t)h->n_flank;
    if (full_i && nf) HHX_HIP(hipMemcpyAsync(full_i, h->out_full_i.p, 4 * nf, hipMemcpyDeviceToHost, g_stream));
    if (full_j && nf) HHX_HIP(hipMemcpyAsync(full_j, h->out_full_j.p, 4 * nf, hipMemcpyDeviceToHost, g_stream));
    if (full_cnt && nf) HHX_HIP(hipMemcpyAsync(full_cnt, h->out_full_cnt.p, 8 * nf, hipMemcpyDeviceToHost, g_stream));
    if (ht_cnt && nf) HHX_HIP(hipMemcpyAsync(ht_cnt, h->out_ht.p, 32 * nf, hipMemcpyDeviceToHost, g_stream));
    if (flank_i && nk) HHX_HIP(hipMemcpyAsync(flank_i, h->out_flank_i.p, 4 * nk, hipMemcpyDeviceToHost, g_stream));
    if (flank_j && nk) HHX_HIP(hipMemcpyAsync(flank_j, h->out_flank_j.p, 4 * nk, hipMemcpyDeviceToHost, g_stream));
    if (flank_cnt && nk) HHX_HIP(hipMemcpyAsync(flank_cnt, h->out_flank_cnt.p, 8 * nk, hipMemcpyDeviceToHost, g_stream));
    if (frag_links) HHX_HIP(hipMemcpyAsync(frag_links, h->frag_links.p, 8 * (size_t)h->t.n_frag, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int hhx_ingest_flank_device(hhx_ingest *h, void **fi, void **fj, void **val) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (fi) *fi = h->out_flank_i.p;
    if (fj) *fj = h->out_flank_j.p;
    if (val) *val = h->out_flank_val.p;
    return 0;
}

extern "C" int hhx_ingest_flank_count_device(hhx_ingest *h, void **cnt_i64) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (cnt_i64) *cnt_i64 = h->out_flank_cnt.p;
    return 0;
}

extern "C" int hhx_ingest_destroy(hhx_ingest *h) {
    delete h;
    return 0;
}

// ---- merge of chunk-ordered tables (multi-GPU exchange) -------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_merge_insert(i64 n, const i32 *__restrict__ ki, const i32 *__restrict__ kj,
                                                      const i64 *__restrict__ w, FlankSlot *tab, u64 mask,
                                                      unsigned long long *counters) {
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (i64)gridDim.x * blockDim.x) {
        const u64 s = find_or_insert(tab, mask, ((u64)(u32)ki[idx] << 32) | (u64)(u32)kj[idx], &counters[0]);
        if (s == ~0ull) { atomicAdd(&counters[1], 1ull); continue; }
        atomicMin((unsigned long long *)&tab[s].ord, (unsigned long long)idx);
        atomicAdd((unsigned long long *)&tab[s].cnt, (unsigned long long)w[idx]);
    }
}
__global__ __launch_bounds__(256) void k_merge_emit(const FlankSlot *tab, u64 cap, const u64 *bitmap, const i64 *prefix,
                                                    i32 *out_i, i32 *out_j, i64 *out_cnt, double *out_val) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (u64)gridDim.x * blockDim.x) {
        if (tab[i].key == EMPTY_KEY) continue;
        const i64 r = ord_rank(bitmap, prefix, tab[i].ord);
        out_i[r] = (i32)(tab[i].key >> 32);
        out_j[r] = (i32)(tab[i].key & 0xffffffffu);
        out_cnt[r] = (i64)tab[i].cnt;
        out_val[r] = (double)tab[i].cnt;
    }
}
struct MergeOut {
    DevBuf<i32> i, j;
    DevBuf<i64> cnt;
    DevBuf<double> val;
};
thread_local MergeOut g_merge_out;
}  // namespace

extern "C" int hhx_table_merge(i64 n, const i32 *ki, const i32 *kj, const i64 *w, i64 *n_out, void **oi, void **oj,
                               void **ocnt, void **oval) {
    if (n < 0 || !n_out) return fail("hhx_table_merge: bad argument");
    const u64 cap = next_pow2((u64)(n > 0 ? n : 1) * 2);
    DevBuf<FlankSlot> tab;
    DevBuf<unsigned long long> counters;
    if (tab.alloc(cap) || counters.alloc(2)) return 1;
    HHX_HIP(hipMemsetAsync(counters.p, 0, 2 * sizeof(unsigned long long), g_stream));
    k_clear<FlankSlot><<<grid_for(cap), 256, 0, g_stream>>>(tab.p, cap);
    HHX_LAUNCH_CHECK();
    if (n) {
        k_merge_insert<<<grid_for((u64)n), 256, 0, g_stream>>>(n, ki, kj, w, tab.p, cap - 1, counters.p);
        HHX_LAUNCH_CHECK();
    }
    unsigned long long c[2];
    HHX_HIP(hipMemcpyAsync(c, counters.p, sizeof c, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    if (c[1]) return fail("hhx_table_merge: hash table overflow");
    const i64 k = (i64)c[0];
    const i64 n_words = n / 64 + 1;
    DevBuf<u64> bitmap;
    DevBuf<i64> wcnt, prefix;
    MergeOut &o = g_merge_out;
    if (bitmap.alloc((size_t)n_words) || wcnt.alloc((size_t)n_words) || prefix.alloc((size_t)n_words + 1) ||
        o.i.alloc((size_t)k) || o.j.alloc((size_t)k) || o.cnt.alloc((size_t)k) || o.val.alloc((size_t)k)) return 1;
    HHX_HIP(hipMemsetAsync(bitmap.p, 0, sizeof(u64) * (size_t)n_words, g_stream));
    k_mark_ord<FlankSlot><<<grid_for(cap), 256, 0, g_stream>>>(tab.p, cap, bitmap.p);
    HHX_LAUNCH_CHECK();
    k_popc_words<<<grid_for((u64)n_words), 256, 0, g_stream>>>(bitmap.p, n_words, wcnt.p);
    HHX_LAUNCH_CHECK();
    HHX_TRY(exclusive_scan_i64(wcnt.p, prefix.p, n_words, nullptr));
    k_merge_emit<<<grid_for(cap), 256, 0, g_stream>>>(tab.p, cap, bitmap.p, prefix.p, o.i.p, o.j.p, o.cnt.p, o.val.p);
    HHX_LAUNCH_CHECK();
    HHX_HIP(hipStreamSynchronize(g_stream));
    *n_out = k;
    if (oi) *oi = o.i.p;
    if (oj) *oj = o.j.p;
    if (ocnt) *ocnt = o.cnt.p;
    if (oval) *oval = o.val.p;
    return 0;
}

// ================================================================================================
// dict_to_matrix :310-373 on device
// ================================================================================================
namespace {

// first appearance of every fragment scanning the items in order, i before j (:337-349)
__global__ __launch_bounds__(256) void k_first_pos(i64 n_keys, const i32 *__restrict__ fi, const i32 *__restrict__ fj,
                                                   const unsigned char *__restrict__ in_set, unsigned long long *first_pos) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n_keys; k += (i64)gridDim.x * blockDim.x) {
        const i32 a = fi[k], b = fj[k];
        if (!in_set[a] || !in_set[b]) continue;
        atomicMin(&first_pos[a], (unsigned long long)(2 * k));
        atomicMin(&first_pos[b], (unsigned long long)(2 * k + 1));
    }
}
__global__ __launch_bounds__(256) void k_mark_first(i32 n_frag, const unsigned long long *first_pos, u64 *bitmap) {
    for (i32 f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frag; f += gridDim.x * blockDim.x)
        if (first_pos[f] != ~0ull) atomicOr((unsigned long long *)&bitmap[first_pos[f] >> 6], 1ull << (first_pos[f] & 63));
}
__global__ __launch_bounds__(256) void k_frag_index(i32 n_frag, const unsigned long long *first_pos, const u64 *bitmap,
                                                    const i64 *prefix, i32 *frag_index) {
    for (i32 f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frag; f += gridDim.x * blockDim.x)
        frag_index[f] = first_pos[f] == ~0ull ? -1 : (i32)ord_rank(bitmap, prefix, first_pos[f]);
}
__global__ __launch_bounds__(256) void k_row_counts(i64 n_keys, const i32 *__restrict__ fi, const i32 *__restrict__ fj,
                                                    const i32 *__restrict__ frag_index, i32 *cnt) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n_keys; k += (i64)gridDim.x * blockDim.x) {
        const i32 a = frag_index[fi[k]], b = frag_index[fj[k]];
        if (a < 0 || b < 0) continue;
        atomicAdd(&cnt[a], 1);
        atomicAdd(&cnt[b], 1);
    }
}
__global__ __launch_bounds__(256) void k_init_counts(i32 shape, i32 *cnt, i32 v) {
    for (i32 r = blockIdx.x * blockDim.x + threadIdx.x; r < shape; r += gridDim.x * blockDim.x) cnt[r] = v;
}
// unsorted fill (atomic cursors) ...
__global__ __launch_bounds__(256) void k_fill(i64 n_keys, const i32 *__restrict__ fi, const i32 *__restrict__ fj,
                                              const double *__restrict__ val, const i32 *__restrict__ frag_index,
                                              const i32 *__restrict__ indptr, i32 *cursor, i32 *tj, float *tx) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n_keys; k += (i64)gridDim.x * blockDim.x) {
        const i32 a = frag_index[fi[k]], b = frag_index[fj[k]];
        if (a < 0 || b < 0) continue;
        const float v = (float)val[k];                              // dtype=float32 at :368
        i32 p = indptr[a] + atomicAdd(&cursor[a], 1);
        tj[p] = b; tx[p] = v;
        p = indptr[b] + atomicAdd(&cursor[b], 1);
        tj[p] = a; tx[p] = v;
    }
}
__global__ __launch_bounds__(256) void k_fill_diag(i32 shape, const i32 *__restrict__ indptr, i32 *cursor, i32 *tj, float *tx) {
    for (i32 r = blockIdx.x * blockDim.x + threadIdx.x; r < shape; r += gridDim.x * blockDim.x) {
        const i32 p = indptr[r] + atomicAdd(&cursor[r], 1);
        tj[p] = r; tx[p] = 1.0f;                                    // self loops :362-364
    }
}
// ... then each row is put in column order with an LDS bitmap rank (columns of a row are unique):
// the same no-sort trick as the SpGEMM output.
__global__ __launch_bounds__(256) void k_sort_rows(i32 shape, i32 W, const i32 *__restrict__ indptr, const i32 *__restrict__ tj,
                                                   const float *__restrict__ tx, i32 *__restrict__ oj, float *__restrict__ ox) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *bitmap = (u32 *)smem, *prefix = bitmap + W, *scratch = prefix + W;
    const int tid = threadIdx.x;
    for (i32 row = blockIdx.x; row < shape; row += gridDim.x) {
        const i32 b = indptr[row], e = indptr[row + 1];
        if (e - b <= 1) {
            if (tid == 0 && e > b) { oj[b] = tj[b]; ox[b] = tx[b]; }
            continue;
        }
        for (i32 w = tid; w < W; w += 256) bitmap[w] = 0;
        __syncthreads();
        for (i32 p = b + tid; p < e; p += 256) atomicOr(&bitmap[tj[p] >> 5], 1u << (tj[p] & 31));
        __syncthreads();
        // exclusive popcount prefix over the words (serial chunk per thread + 256-entry scan)
        const i32 per = (W + 255) / 256, w0 = tid * per, w1 = min(W, w0 + per);
        u32 local = 0;
        for (i32 w = w0; w < w1; ++w) local += __popc(bitmap[w]);
        scratch[tid] = local;
        __syncthreads();
        if (tid < 64) {
            u32 v0 = scratch[tid * 4], v1 = scratch[tid * 4 + 1], v2 = scratch[tid * 4 + 2], v3 = scratch[tid * 4 + 3];
            u32 s = v0 + v1 + v2 + v3, incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                u32 t = __shfl_up(incl, o, 64);
                if (tid >= o) incl += t;
            }
            u32 ex = incl - s;
            scratch[tid * 4] = ex; scratch[tid * 4 + 1] = ex + v0; scratch[tid * 4 + 2] = ex + v0 + v1; scratch[tid * 4 + 3] = ex + v0 + v1 + v2;
        }
        __syncthreads();
        u32 run = scratch[tid];
        for (i32 w = w0; w < w1; ++w) { prefix[w] = run; run += __popc(bitmap[w]); }
        __syncthreads();
        for (i32 p = b + tid; p < e; p += 256) {
            const i32 c = tj[p];
            const i32 r = (i32)(prefix[c >> 5] + __popc(bitmap[c >> 5] & ((1u << (c & 31)) - 1u)));
            oj[b + r] = c;
            ox[b + r] = tx[p];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int hhx_dict_to_matrix(i64 n_keys, const i32 *frag_i, const i32 *frag_j, const double *value, int on_device,
                                  i32 n_frag, const uint8_t *in_set_host, i32 n_rest, int add_self_loops,
                                  i32 *frag_index_host, i32 *n_linked_out, hhx_csr **out) {
    if (!out || !in_set_host || n_frag <= 0 || n_keys < 0 || n_rest < 0) return fail("hhx_dict_to_matrix: bad argument");
    DevBuf<i32> sfi, sfj;
    DevBuf<double> sval;
    if (!on_device && n_keys) {
        if (upload(sfi, frag_i, (size_t)n_keys) || upload(sfj, frag_j, (size_t)n_keys) || upload(sval, value, (size_t)n_keys)) return 1;
        frag_i = sfi.p; frag_j = sfj.p; value = sval.p;
    }
    DevBuf<unsigned char> in_set;
    if (upload(in_set, (const unsigned char *)in_set_host, (size_t)n_frag)) return 1;
    DevBuf<unsigned long long> first_pos;
    DevBuf<i32> frag_index;
    if (first_pos.alloc((size_t)n_frag) || frag_index.alloc((size_t)n_frag)) return 1;
    HHX_HIP(hipMemsetAsync(first_pos.p, 0xff, sizeof(unsigned long long) * (size_t)n_frag, g_stream));
    const i64 n_words = (2 * n_keys) / 64 + 1;
    DevBuf<u64> bitmap;
    DevBuf<i64> wcnt, prefix;
    if (bitmap.alloc((size_t)n_words) || wcnt.alloc((size_t)n_words) || prefix.alloc((size_t)n_words + 1)) return 1;
    HHX_HIP(hipMemsetAsync(bitmap.p, 0, sizeof(u64) * (size_t)n_words, g_stream));
    if (n_keys) {
        k_first_pos<<<grid_for((u64)n_keys), 256, 0, g_stream>>>(n_keys, frag_i, frag_j, in_set.p, first_pos.p);
        HHX_LAUNCH_CHECK();
    }
    k_mark_first<<<grid_for((u64)n_frag), 256, 0, g_stream>>>(n_frag, first_pos.p, bitmap.p);
    HHX_LAUNCH_CHECK();
    k_popc_words<<<grid_for((u64)n_words), 256, 0, g_stream>>>(bitmap.p, n_words, wcnt.p);
    HHX_LAUNCH_CHECK();
    i64 n_linked = 0;
    HHX_TRY(exclusive_scan_i64(wcnt.p, prefix.p, n_words, &n_linked));
    k_frag_index<<<grid_for((u64)n_frag), 256, 0, g_stream>>>(n_frag, first_pos.p, bitmap.p, prefix.p, frag_index.p);
    HHX_LAUNCH_CHECK();
    const i64 shape64 = n_linked + n_rest;
    if (shape64 > INT32_MAX) return fail("matrix order exceeds int32");
    const i32 shape = (i32)shape64;
    DevBuf<i32> cnt, indptr, cursor;
    if (cnt.alloc((size_t)shape + 1) || indptr.alloc((size_t)shape + 1) || cursor.alloc((size_t)shape + 1)) return 1;
    k_init_counts<<<grid_for((u64)shape + 1), 256, 0, g_stream>>>(shape, cnt.p, add_self_loops ? 1 : 0);
    HHX_LAUNCH_CHECK();
    if (n_keys) {
        k_row_counts<<<grid_for((u64)n_keys), 256, 0, g_stream>>>(n_keys, frag_i, frag_j, frag_index.p, cnt.p);
        HHX_LAUNCH_CHECK();
    }
    i64 nnz = 0;
    HHX_TRY(exclusive_scan_i32(cnt.p, indptr.p, shape, &nnz));
    hhx_csr *m = nullptr;
    HHX_TRY(hhx_csr_alloc_internal(shape, shape, nnz, &m));
    DevBuf<i32> tj;
    DevBuf<float> tx;
    if (tj.alloc((size_t)nnz) || tx.alloc((size_t)nnz)) { hhx_csr_free(m); return 1; }
    hipError_t e = hipMemsetAsync(cursor.p, 0, sizeof(i32) * ((size_t)shape + 1), g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(m->indptr.p, indptr.p, sizeof(i32) * ((size_t)shape + 1), hipMemcpyDeviceToDevice, g_stream);
    if (e != hipSuccess) { hhx_csr_free(m); return fail("dict_to_matrix: %s", hipGetErrorString(e)); }
    if (n_keys) k_fill<<<grid_for((u64)n_keys), 256, 0, g_stream>>>(n_keys, frag_i, frag_j, value, frag_index.p, indptr.p, cursor.p, tj.p, tx.p);
    if (add_self_loops && shape) k_fill_diag<<<grid_for((u64)shape), 256, 0, g_stream>>>(shape, indptr.p, cursor.p, tj.p, tx.p);
    const i32 W = (shape + 31) / 32;
    const size_t lds = (size_t)W * 8 + 256 * 4;
    if (lds > 160 * 1024) { hhx_csr_free(m); return fail("dict_to_matrix: matrix order %d exceeds the LDS bitmap capacity", shape); }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_sort_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (shape) k_sort_rows<<<(unsigned)std::min<i64>(shape, 256 * 8), 256, lds, g_stream>>>(shape, W, indptr.p, tj.p, tx.p, m->indices.p, m->data.p);
    e = hipGetLastError();
    if (e == hipSuccess && frag_index_host)
        e = hipMemcpyAsync(frag_index_host, frag_index.p, sizeof(i32) * (size_t)n_frag, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { hhx_csr_free(m); return fail("dict_to_matrix: %s", hipGetErrorString(e)); }
    if (n_linked_out) *n_linked_out = (i32)n_linked;
    *out = m;
    return 0;
}
