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
    std::vector<hhx::LinkRun *> runs[2];   // [0] full (or combined) table, [1] flank table (bins mode only)
    bool finalized = false;
    i64 n_full = 0, n_flank = 0;
    hhx::OrderedTables ordered;
    hhx::DevBuf<i32> stage[4];             // staging for host-side inputs
    ~hhx_ingest() {
        for (auto &v : runs)
            for (auto *r : v) delete r;
    }
    const hhx::LinkRun *table(int which) const {
        const auto &v = runs[(combined || which == 0) ? 0 : 1];
        return v.empty() ? nullptr : v[0];
    }
};

// hhx_matrix.hip: dict_to_matrix on a run (flank rows, first-seen order taken from ord_flank)
int hhx_link_matrix_from_run(const hhx::LinkRun *run, i32 n_frag, u64 ord_limit, const uint8_t *in_set_host, i32 n_rest, int add_self_loops,
                             i32 *frag_index_host, i32 *n_linked_out, hhx_csr **out);
