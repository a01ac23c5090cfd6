// Internal layout of the link-table builder shared by hhx_ingest.hip (pairs -> aggregated tables)
// and hhx_matrix.hip (tables -> CSR link matrix).
#pragma once
#include "hhx_common.h"

namespace hhx {

// ---- record / key encoding ------------------------------------------------------------------------
// key  = (i << 29) | j, i and j contig or fragment ids (< 2^29), already oriented (:1629 / :1720)
// rec  = key | ht << 58 | FULL_BIT | FLANK_BIT        (one 64-bit word per surviving read pair)
constexpr int ID_BITS = 29;
constexpr u64 ID_MASK = (1ull << ID_BITS) - 1;
constexpr u64 KEY_MASK = (1ull << (2 * ID_BITS)) - 1;
constexpr int HT_SHIFT = 58;
constexpr u64 FULL_BIT = 1ull << 60, FLANK_BIT = 1ull << 61;
constexpr u64 EMPTY_KEY = ~0ull, NO_ORD = ~0ull;

// An aggregated link table ("run"): one row per distinct key, in no particular order.
//   ord_full / ord_flank : global stream ordinal of the first pair that touched the key in the
//                          full_link_dict (:1649) / flank_link_dict (:1637) sense, NO_ORD if never
//   ht[4]                : HT_link_dict counts [HH, HT, TH, TT] (:404-416); their sum is the full count
//   fl                   : flank_link_dict count
struct LinkRun {
    i64 n = 0;
    DevBuf<u64> key, ord_full, ord_flank;
    DevBuf<u32> ht;     // [n][4]
    DevBuf<u32> fl;     // [n]
    int alloc(i64 rows) {
        n = rows;
        return key.alloc((size_t)rows) || ord_full.alloc((size_t)rows) || ord_flank.alloc((size_t)rows) ||
               ht.alloc((size_t)rows * 4) || fl.alloc((size_t)rows);
    }
};

// Per-contig / per-fragment facts packed into one 16-byte record: a read pair needs two 16-byte
// gathers (four when contigs are split) instead of a dozen 1-8 byte ones.
//   rank : lexical rank of the NAME (key orientation :1629 / :1720)
//   aux  : contigs: id of the contig's first fragment; fragments: unused
//   lenf : length | NX_BIT (fragment is in Nx_frag_set; for a contig: its only fragment is) | SPLIT_BIT
constexpr i64 NX_BIT = (i64)1 << 61, SPLIT_BIT = (i64)1 << 62, LEN_MASK = ((i64)1 << 48) - 1;
struct __attribute__((aligned(16))) UnitInfo {
    i32 rank, aux;
    i64 lenf;
};

struct DevTables {           // device-resident, built from hhx_ingest_config
    const UnitInfo *ctg;     // [n_ctg]
    const UnitInfo *frag;    // [n_frag]
    i32 n_ctg, n_frag;
    i64 bin_size, flank;
    i32 bins, skip_intra;
};

#ifdef __HIPCC__
// ---- map: one read pair -> at most one record of stream `stream` (shared by the group-by and the side records)
__device__ __forceinline__ bool is_flank(i64 coord, i64 length, i64 flank) {   // :299-307
    return flank == 0 || coord <= flank || coord > length - flank;
}

// stream 0: the contig-pair table (full_link_dict + HT_link_dict; also flank_link_dict when no contig is
//           split, because fragment == contig then);  stream 1 (bins only): the fragment-pair flank table.
// COMBINED (no split contigs, parse_alignments_for_ctgs :1596-1655) is a compile-time variant: two
// 16-byte gathers and no divisions per pair.
// the part of the map that comes before the table look-ups
__device__ __forceinline__ bool pair_admitted(const DevTables &t, i32 r, i32 m) {
    if (t.skip_intra && r == m) return false;                                    // pairs_generator_inter_ctgs :1582
    return (u32)r < (u32)t.n_ctg && (u32)m < (u32)t.n_ctg;                       // :1625 / :1702 (name not in fa_dict)
}
// ... and the part after them: a = t.ctg[r], b = t.ctg[m] are handed in, so that a caller can issue the gathers of
// several pairs before any of them is consumed (k_map_records)
template <bool COMBINED>
__device__ __forceinline__ bool map_pair_with(const DevTables &t, int stream, i32 r, i32 m, i64 p1, i64 p2, UnitInfo a, UnitInfo b, u64 &rec,
                                              u64 *xy = nullptr) {
    if (!COMBINED && t.bins && r == m && !(a.lenf & SPLIT_BIT)) return false;    // :1699
    i32 ci = r, cj = m;
    i64 xi = p1 + 1, xj = p2 + 1;                                                // 1-based, :1629 (int32 or int64 positions, :116-147)
    if (a.rank > b.rank || (r == m && xi > xj)) {
        ci = m; cj = r;
        const i64 tx = xi; xi = xj; xj = tx;
        const UnitInfo tu = a; a = b; b = tu;
    }
    const i64 li = a.lenf & LEN_MASK, lj = b.lenf & LEN_MASK;
    if (xy) *xy = ((u64)(u32)xi << 32) | (u64)(u32)xj;                           // oriented 1-based contig coordinates
    const u64 ht = (u64)((xi * 2 > li) * 2 + (xj * 2 > lj));                     // :404-416
    if (COMBINED) {
        const bool flank_ok = (a.lenf & b.lenf & NX_BIT) && is_flank(xi, li, t.flank) && is_flank(xj, lj, t.flank);   // :1636
        rec = ((u64)(u32)ci << ID_BITS) | (u64)(u32)cj | (ht << HT_SHIFT) | FULL_BIT | (flank_ok ? FLANK_BIT : 0);
        return true;
    }
    i32 fi = a.aux, fj = b.aux;
    i64 yi = xi, yj = xj;
    if (t.bins) {                                                                // convert_frags :1662-1670
        if (a.lenf & SPLIT_BIT) { const i64 nb = (xi + t.bin_size - 1) / t.bin_size; fi += (i32)(nb - 1); yi = xi - (nb - 1) * t.bin_size; }
        if (b.lenf & SPLIT_BIT) { const i64 nb = (xj + t.bin_size - 1) / t.bin_size; fj += (i32)(nb - 1); yj = xj - (nb - 1) * t.bin_size; }
        if (fi == fj) return false;                                              // :1715
    }
    if (stream == 2) {                                                           // ctg_pair_to_frag :1731-1733: every pair that got here
        if (t.bins) {
            const UnitInfo fa = t.frag[fi], fb = t.frag[fj];
            if (fa.rank > fb.rank) { const i32 tf = fi; fi = fj; fj = tf; }
        }
        rec = ((u64)(u32)fi << ID_BITS) | (u64)(u32)fj | FLANK_BIT;
        return true;
    }
    if (stream == 0) {
        if (t.bins && r == m) return false;                                      // :1736
        rec = ((u64)(u32)ci << ID_BITS) | (u64)(u32)cj | (ht << HT_SHIFT) | FULL_BIT;
        return true;
    }
    UnitInfo fa = t.frag[fi], fb = t.frag[fj];
    if (t.bins && fa.rank > fb.rank) {                                           // :1719-1720
        const i32 tf = fi; fi = fj; fj = tf;
        const i64 ty = yi; yi = yj; yj = ty;
        const UnitInfo tu = fa; fa = fb; fb = tu;
    }
    if (!((fa.lenf & fb.lenf & NX_BIT) && is_flank(yi, fa.lenf & LEN_MASK, t.flank) && is_flank(yj, fb.lenf & LEN_MASK, t.flank)))
        return false;                                                            // :1726
    rec = ((u64)(u32)fi << ID_BITS) | (u64)(u32)fj | FLANK_BIT;
    return true;
}
template <bool COMBINED>
__device__ __forceinline__ bool map_pair(const DevTables &t, int stream, i32 r, i32 m, i64 p1, i64 p2, u64 &rec, u64 *xy = nullptr) {
    if (!pair_admitted(t, r, m)) return false;
    return map_pair_with<COMBINED>(t, stream, r, m, p1, p2, t.ctg[r], t.ctg[m], rec, xy);
}

#endif

// Insertion-ordered views (Python dict order), materialised on demand from a run.
struct OrderedTables {
    bool ready = false;
    i64 n_full = 0, n_flank = 0;
    DevBuf<i32> full_i, full_j, flank_i, flank_j;
    DevBuf<i64> full_cnt, ht, flank_cnt;
    DevBuf<double> flank_val;
    DevBuf<unsigned long long> frag_links;
};

}  // namespace hhx

struct hhx_ingest {
    hhx::DevTables t{};
    hhx::DevBuf<hhx::UnitInfo> ctg_info, frag_info;
    bool combined = true;                  // one table serves full and flank (no bins: fragment == contig)
    u64 ord_base = 0;                      // global ordinal of this handle's first pair (multi-GPU chunk offset)
    u64 n_pushed = 0;                      // pairs pushed so far
    u64 ord_limit = 0;                     // 1 + largest ordinal that can occur (pairs and pushed tables)
    std::vector<hhx::LinkRun *> runs[3];   // [0] full (or combined) table, [1] flank table (bins mode only),
                                           // [2] every fragment pair seen (ctg_pair_to_frag :1731, on request)
    bool keep_frag_pairs = false;
    bool finalized = false;
    i64 n_full = 0, n_flank = 0;
    hhx::OrderedTables ordered;
    // optional side product (hhx_ingest_keep_pairs): the oriented 1-based coordinates of every read pair counted in
    // full_link_dict, in stream order — what update_clm_dict :395-401 and record_coord_pairs :454-471 consume
    bool keep_pairs = false;
    std::vector<hhx::DevBuf<u64>> side_key, side_xy;     // one pair of arrays per push: key, (xi << 32 | xj)
    i64 n_side = 0;
    bool pairs_dropped = false;            // side_key / side_xy were released after paired_links.clm was written (hhx_ingest_write_clm_async)
    i64 max_ctg_len = 0;                   // longest contig (bounds the CLM distances: hhx_ingest_write_clm)
    hhx::DevBuf<i32> stage[4];             // staging for host-side inputs
    hhx::DevBuf<i64> stage64[2];           // ... of 64-bit positions (hhx_ingest_push64)
    ~hhx_ingest() {
        for (auto &v : runs)
            for (auto *r : v) delete r;
    }
    const hhx::LinkRun *table(int which) const {
        const auto &v = runs[(combined || which == 0) ? 0 : 1];
        return v.empty() ? nullptr : v[0];
    }
};

// hhx_pairs.hip: stable compaction of the side records of one push; CLM / coordinate lists from them
// (POS = i32 or i64 positions; the records hold 32-bit coordinates: a 64-bit stream must stay below 2^32, checked by the caller)
template <class POS>
int hhx_side_records_push(hhx_ingest *h, i64 n_pairs, const i32 *id1, const POS *pos1, const i32 *id2, const POS *pos2);

// hhx_matrix.hip: dict_to_matrix on a run (flank rows, first-seen order taken from ord_flank)
int hhx_link_matrix_from_run(const hhx::LinkRun *run, i32 n_frag, u64 ord_limit, const uint8_t *in_set_host, i32 n_rest, int add_self_loops,
                             i32 *frag_index_host, i32 *n_linked_out, hhx_csr **out);

// hhx_pairs.hip: paired_links.clm into an open file descriptor (closed there); hhx_ingest.hip: the dict-ordered tables (made on first use)
namespace hhx {
int ingest_write_clm_fd(hhx_ingest *h, int fd, const uint8_t *names_blob, const i64 *name_off, i64 *n_lines, i64 *n_bytes);
}
int hhx_ingest_ordered_full_device(hhx_ingest *h, const i32 **fi, const i32 **fj);
