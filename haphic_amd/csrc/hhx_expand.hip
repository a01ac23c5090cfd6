// Fused MCL iteration:  P = prune(normalize(power(A * B, r)))  without ever writing the expanded
// matrix C = A * B to HBM.  Reference: scripts/HapHiC_cluster.py mcl() :2030-2042 (expand, inflate,
// prune) and run_mcl_clustering :2144-2147 (the un-pruned pre-expansion, which at n = 100k would be a
// 10^10-entry matrix — it is consumed row by row here instead).
//
// One workgroup owns one output row.  Accumulators are 64-bit FIXED POINT in LDS (ds_add_u64: integer adds
// commute -> order-free, bit reproducible for any lane / wave / GPU count).  Measured on MI355X
// (tools/lds_atomic_bench.hip): a wave-wide ds_add_u64 on random slots costs 12.8 clk, ds_add_f64 25.4 clk, so the
// integer form it is; the float64 -> fixed conversion is one v_add_f64 and one 32-bit integer add (fx_bits below).
// Two row classes, chosen from the row's product count F_i:
//   WINDOW : F_i >> n_cols.  Dense accumulators indexed by (column - window start); several windows if
//            the row is wider than LDS, each window reading only its sub-range of every B row (rows
//            are sorted -> split points precomputed), so the products are traversed once.
//   COMPACT: light rows.  An n_cols-bit LDS bitmap + popcount prefix maps a column to its rank in the
//            sorted output row; accumulators are indexed by rank.
// Epilogue per window: x = float(acc), p = x^r, deterministic block sum S.  Entries that can still
// survive the threshold against the running sum (a lower bound of the row sum) go to a candidate list
// in HBM; when the row sum is final the candidates are tested exactly as the reference does
// (q = float(p / S) >= float32(pruning), first row maximum restored, second L1 normalisation) and the
// survivors are bump-allocated; a final pass packs the rows into CSR order.
#include <chrono>

#include "hhx_common.h"
#include "hhx_sort.h"

using namespace hhx;

// pool demand of the last fused iteration on this thread, and the hint for the next one (set by mcl()'s loop, consumed by hhx_expand_impl)
struct ExpandDemand { i64 out = 0, cand = 0; };
thread_local ExpandDemand g_expand_hint, g_expand_demand;
void hhx_expand_set_hint(i64 out, i64 cand) { g_expand_hint.out = out; g_expand_hint.cand = cand; }
void hhx_expand_last_demand(i64 *out, i64 *cand) { *out = g_expand_demand.out; *cand = g_expand_demand.cand; }

int hhx_csr_alloc_internal(i32 n_rows, i32 n_cols, i64 nnz, hhx_csr **out);
namespace hhx { i64 pool_cached_bytes(); }

namespace {

constexpr int EX_T_WIN = 1024;      // window kernel: 16 waves, one workgroup per CU (it owns all of LDS)
constexpr int EX_T_CMP = 256;       // compact kernel: light rows, several workgroups per CU
constexpr int EX_T_MAX = 1024;
constexpr int EX_WAVES_MAX = EX_T_MAX / HHX_WAVE;
#define EX_T ((int)blockDim.x)      // device code below is written for either width
#define EX_WAVES ((int)blockDim.x / HHX_WAVE)
constexpr int STAGE = 1024;         // compact kernel: staged A entries per chunk
constexpr int MAX_WIN = 512;
constexpr int N_DUMMY = HHX_WAVE;   // window kernel: one scratch accumulator per lane behind the window (masked entries)

struct ExParams {
    const i32 *Ap, *Aj; const float *Ax;
    const i32 *Bp, *Bj; const float *Bx;
    // window kernel operand stream.  Every (B row, column window) segment is described by one 32-byte record
    // {b0, b1, b2, b3 | b4, v1, v2, v3 (float bits)}: the entries of sub-segment c = 1, 2, 3, [b(c-1), b(c)), all hold
    // the SAME value v_c and are streamed as 16-bit window-local columns alone (2 B per product: their product with a_ik
    // is formed once per sub-segment); entries [b3, b4) are (16-bit column, float32 value) pairs (6 B per product).
    // For a general operand b0 = b1 = b2 = b3.  When B = D^-1 * L with integer link counts L (iteration 0 of
    // run_mcl_clustering: the normalised raw link matrix) the layout pass regroups every segment by link count —
    // [count 1][count 2][count 3][other], 75 % / 12 % / 5 % / 8 % of a Hi-C link matrix — and v_c = float(c / rowsum).
    // Sc16 / Sx are the (regrouped) columns / values, rec the records (two int4 per segment).
    const unsigned short *Sc16; const float *Sx; const int4 *rec;
    const int2 *Bjx;                // hash class: (column, float bits) pairs of B, one 8-byte word per entry (k_pack_jx)
    i32 narrow_classes;             // 1: some records have count-2 / count-3 sub-segments (the class stream); 0: count-1 only; -1: none (general operand)
    i32 wb;                         // A entries per wave batch
    i32 n_rows, n_cols;
    double scale, inv_scale;        // 2^(shift-52), 2^(52-shift): products are rounded on the 2^-52 grid of [0, 1]
    double r; int square; float thr;
    int raw;                        // 1: plain product C = A * B (hhx_spgemm): every non-zero entry is written, nothing else
    // candidate + survivor pools (col, value) and the per-row table
    i32 *cand_col; float *cand_val; i64 cand_cap;
    i32 *out_col; float *out_val; i64 out_cap;
    unsigned long long *cursors;    // [0] cand cursor [1] out cursor [2] overflow flag [3] nnz_C [4] products [5..6] window class [7] uniform products
    i64 *row_off; i32 *row_cnt;
    i32 n_win;
    // window class, one launch per column window: per-row state carried between the launches
    double *s_run;                  // [n_rows] running sum of p over the windows done so far
    i64 *g_win_off; i32 *g_win_cnt; // [n_rows][n_win] candidate segment of every (row, window)
    // dense mode (the inflation sweep, run_mcl_clustering :2155-2158): the window kernel stores x = float(acc) of every column
    // into row `row` of this n_rows x dense_ld float32 block instead of inflating / pruning (0 = no entry)
    float *dense; i64 dense_ld;
    // dense, all rows, symmetric mode: `dense` holds the UPPER BLOCK TRIANGLE alone (tri != 0) — block row I (rows [I cap, (I + 1) cap))
    // starts at tri_row_off(I) and keeps only its windows J >= I, rows tri_ldn - I cap floats apart (tri_ldn = n_win * cap):
    // cap^2 n_win (n_win + 1) / 2 floats instead of n^2 (88 GB instead of 160 GB at n = 200k)
    i32 tri; i64 tri_ldn;
    // INTEGER arithmetic of iteration 0 on the raw link matrix (specification: DESIGN.md 4.1, "integer arithmetic").  A = rows of L, B = L, integer link counts c:
    // the addend of product (i, k, j) is c_ik * W_k * c_kj with W_k = rint(2^s / d_k) — S = L D^-1 L, a symmetric matrix, in exact
    // 64-bit integers; y = float(acc * 2^-s), x = float(y / d_i).  W == nullptr: the float arithmetic above (fx_bits).
    const u64 *W; const unsigned short *A16; const double *row_div; double fx_inv;
    i32 sym;                        // dense: launch wv covers only the rows of the blocks I <= wv (blocks J >= I of S); the rest is mirrored afterwards
    i32 sym_row0;                   // ... global index of local row 0 (a row block of a multi-GPU rank; 0 for the whole matrix)
};

struct ExLds {
    u64 *acc;            // [cap (+ N_DUMMY)] exact fixed-point sums (see fx_bits); the epilogue reuses the slots for float p
    double *st_da;       // [STAGE]   (compact kernel)
    i64 *bcast;          // [1] block broadcast slot
    i32 *st_qb, *st_qe;  // [STAGE]   (compact kernel)
    double *red_d;       // [EX_WAVES_MAX]
    i32 *red_i;          // [EX_WAVES_MAX]
    float *red_f;        // [EX_WAVES_MAX]
    i64 *win_off;        // [MAX_WIN] (compact kernel)
    i32 *win_cnt;        // [MAX_WIN]
    u32 *bitmap, *prefix;  // [W] each (compact mode only)
    i32 *ctr;            // [2] window kernel: batch cursor of the row
};

// compact kernel.  nslots: entries of the LDS (offset, count) segment table
__host__ __device__ inline size_t ex_fixed_bytes(i32 W, int nslots) {
    return (size_t)STAGE * (8 + 4 + 4) + (size_t)EX_WAVES_MAX * (8 + 4 + 4) + 8 + (size_t)nslots * (8 + 4) +
           (size_t)W * 8;
}
__device__ __forceinline__ ExLds ex_carve(unsigned char *smem, i32 cap, i32 W, int nslots) {
    ExLds l;
    unsigned char *p = smem;
    l.acc = (u64 *)p; p += (size_t)cap * 8;
    l.st_da = (double *)p; p += STAGE * 8;
    l.red_d = (double *)p; p += EX_WAVES_MAX * 8;
    l.bcast = (i64 *)p; p += 8;
    l.win_off = (i64 *)p; p += (size_t)nslots * 8;
    l.st_qb = (i32 *)p; p += STAGE * 4;
    l.st_qe = (i32 *)p; p += STAGE * 4;
    l.red_i = (i32 *)p; p += EX_WAVES_MAX * 4;
    l.red_f = (float *)p; p += EX_WAVES_MAX * 4;
    l.win_cnt = (i32 *)p; p += (size_t)nslots * 4;
    l.bitmap = (u32 *)p; p += (size_t)W * 4;
    l.prefix = (u32 *)p;
    l.ctr = nullptr;
    return l;
}
// window kernel: the accumulators, one scratch slot per lane, the reduction scratch — no staging arrays (every wave
// fetches its own batches of A entries), so a window is (160 KiB - 1 KiB) / 8 = 20k columns wide
__host__ __device__ constexpr size_t win_fixed_bytes() {
    return (size_t)N_DUMMY * 8 + (size_t)EX_WAVES_MAX * (8 + 4 + 4) + 8 + 8;
}
__device__ __forceinline__ ExLds win_carve(unsigned char *smem, i32 cap) {
    ExLds l;
    unsigned char *p = smem;
    l.acc = (u64 *)p; p += ((size_t)cap + N_DUMMY) * 8;
    l.red_d = (double *)p; p += EX_WAVES_MAX * 8;
    l.bcast = (i64 *)p; p += 8;
    l.red_i = (i32 *)p; p += EX_WAVES_MAX * 4;
    l.red_f = (float *)p; p += EX_WAVES_MAX * 4;
    l.ctr = (i32 *)p;
    l.st_da = nullptr; l.st_qb = l.st_qe = nullptr; l.win_off = nullptr; l.win_cnt = nullptr; l.bitmap = l.prefix = nullptr;
    return l;
}

__device__ __forceinline__ float ex_inflate(float x, double r, int square) {
    return square ? x * x : hhx_powr(x, r);
}

__device__ __forceinline__ i32 lower_bound_i32(const i32 *__restrict__ a, i32 b, i32 e, i32 v) {
    while (b < e) {
        const i32 m = b + ((e - b) >> 1);              // b + e overflows int32 beyond 2^30 entries
        if (a[m] < v) b = m + 1; else e = m;
    }
    return b;
}

// ---- deterministic block reductions (fixed tree: wave shuffles, then the 4 wave results in order)
__device__ __forceinline__ double block_sum_f64(double v, double *red) {
    v = wave_sum_f64(v);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x / HHX_WAVE] = v;
    __syncthreads();
    double s = red[0];
    for (int k = 1; k < EX_WAVES; ++k) s += red[k];
    return s;
}
__device__ __forceinline__ i32 block_sum_i32(i32 v, i32 *red) {
    v = wave_sum_i32(v);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x / HHX_WAVE] = v;
    __syncthreads();
    i32 s = red[0];
    for (int k = 1; k < EX_WAVES; ++k) s += red[k];
    return s;
}
// exclusive scan of one i32 per thread; returns the thread's offset, *total = block total
__device__ __forceinline__ i32 block_excl_scan_i32(i32 v, i32 *red, i32 *total) {
    i32 incl = v;
#pragma unroll
    for (int o = 1; o < HHX_WAVE; o <<= 1) {
        const i32 t = __shfl_up(incl, o, HHX_WAVE);
        if (lane_id() >= o) incl += t;
    }
    __syncthreads();
    if (lane_id() == HHX_WAVE - 1) red[threadIdx.x / HHX_WAVE] = incl;
    __syncthreads();
    const int w = threadIdx.x / HHX_WAVE;
    i32 off = 0;
    i32 tot = 0;
    for (int k = 0; k < EX_WAVES; ++k) { if (k < w) off += red[k]; tot += red[k]; }
    *total = tot;
    return off + incl - v;
}
// argmax by (value desc, column asc)
__device__ __forceinline__ void block_argmax(float &q, i32 &c, float *redf, i32 *redi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float oq = __shfl_down(q, o, HHX_WAVE);
        const i32 oc = __shfl_down(c, o, HHX_WAVE);
        if (oq > q || (oq == q && oc < c)) { q = oq; c = oc; }
    }
    __syncthreads();
    if (lane_id() == 0) { redf[threadIdx.x / HHX_WAVE] = q; redi[threadIdx.x / HHX_WAVE] = c; }
    __syncthreads();
    q = redf[0]; c = redi[0];
    for (int k = 1; k < EX_WAVES; ++k)
        if (redf[k] > q || (redf[k] == q && redi[k] < c)) { q = redf[k]; c = redi[k]; }
}

// ---- stage one chunk of the A row (compact kernel): (a * scale, begin, end) of every referenced B row
__device__ __forceinline__ void stage_chunk(const ExParams &P, const ExLds &l, i32 a0, i32 len) {
    for (i32 t = threadIdx.x; t < len; t += EX_T) {
        const i32 k = P.Aj[a0 + t];
        l.st_da[t] = (double)P.Ax[a0 + t] * P.scale;
        l.st_qb[t] = P.Bp[k];
        l.st_qe[t] = P.Bp[k + 1];
    }
}

// ---- fixed point -----------------------------------------------------------------------------------
// The operands are stochastic (0 <= a, b <= 1, row sums <= 1), so a scaled product p = a * b * 2^(shift-52) lies in
// [0, 1]: p + 1.0 rounds it (to nearest, ties to even) to a multiple of 2^-52 inside [1, 2], where consecutive
// doubles are consecutive integers — the bit pattern of (p + 1.0) minus the bit pattern of 1.0 IS rint(p * 2^52).
// One v_add_f64 and one 32-bit integer add on the high word instead of a float64 -> int64 conversion (~9 VALU).
// Sums of at most 2^12 such values stay below 2^64 and add exactly in any order.  Specification: oracle mode 1,
// sum of rint(a * b * 2^shift), one rounding to float32 at the end.
__device__ __forceinline__ u64 fx_bits(double p) {
    return (u64)__double_as_longlong(p + 1.0) - 0x3ff0000000000000ull;
}
// fx_bits(a * b) with ONE instruction for the product and the rounding add: a and b are float32 values (a scaled by a power of two), so
// a * b is exact in a double and fma(a, b, 1.0) rounds the same exact number once — the same bits as (a * b) + 1.0
__device__ __forceinline__ u64 fx_bits_prod(double a, double b) {
    return (u64)__double_as_longlong(__builtin_fma(a, b, 1.0)) - 0x3ff0000000000000ull;
}
__device__ __forceinline__ void acc_add(u64 *slot, double p) {
    atomicAdd((unsigned long long *)slot, (unsigned long long)fx_bits(p));
}

// ---- window mode: acc[c - c0] += fixed(a * b) -----------------------------------------------------------------
// Every WAVE works on its own: it draws batches of P.wb consecutive A entries of the row from an LDS cursor (dynamic
// balance, no workgroup barrier inside a row), one entry per lane: the lane loads (k, a_ik) and the 16-byte record of
// (B row k, this window) and forms the segment's uniform product once.  The wave then walks the batch twice with
// wave-uniform cursors (v_readlane): first the value-uniform sub-segments as WIDE tiles — a lane loads 8 consecutive
// 16-bit columns with one 16-byte load, two such loads per tile = up to 1024 entries, 16 LDS atomics per lane — then
// the (column, value) sub-segments as tiles of UX entries per lane.  Both walks issue the loads of a GROUP of tiles
// back to back and consume the tiles in order while the later ones are still in flight.  Everything between a tile's
// loads and its atomics is branch-free per lane (masked entries go to a per-lane scratch slot behind the window),
// so that the compiler's s_waitcnt pass counts the loads exactly instead of falling back to vmcnt(0):
// tools/stream_bench.hip is the prototype of this loop.
constexpr int WB_MAX = 32;          // A entries per wave batch: P.wb <= WB_MAX (short rows take smaller batches so that every wave gets one)
constexpr int WIDE_UNIT = 512;      // entries per 16-byte lane load of a wave
constexpr int STREAM_PREFIX = 512;  // Sc16[8 l .. 8 l + 7] = cap + l: what lane l of a wide tile loads when its entries lie past the sub-segment (scratch
                                    // accumulator l), so that the consume loop needs no per-entry range test

__device__ __forceinline__ i32 ceil64(i32 x) { return (x + 63) & ~63; }
struct BatchRegs {                  // one A entry per lane (lanes >= cnt: empty segments)
    i32 b0, b1, b2, b3, b4;         // record boundaries
    u32 g_lo, g_hi;                 // fx_bits(a_ik * scale * v1): the count-1 sub-segment's fixed-point product
    u32 v2, v3;                     // float bits of the count-2 / count-3 values (their products are formed when the pass gets there)
    u32 da_lo, da_hi;               // a_ik * scale (double bits)
};
template <bool FX>
__device__ __forceinline__ void batch_load(const ExParams &P, i32 a_b, i32 a_e, i32 batch, i32 wv, BatchRegs &r) {
    const i32 e = a_b + batch * P.wb + lane_id();
    const bool ok = lane_id() < P.wb && e < a_e;
    const i32 ec = ok ? e : a_b;                      // unconditional loads (a_b < a_e whenever a batch exists)
    const i32 k = P.Aj[ec];
    const int4 *rp = P.rec + ((size_t)k * P.n_win + wv) * 2;
    const int4 r0 = rp[0], r1 = rp[1];
    r.b0 = ok ? r0.x : 0; r.b1 = ok ? r0.y : 0; r.b2 = ok ? r0.z : 0; r.b3 = ok ? r0.w : 0; r.b4 = ok ? r1.x : 0;
    r.v2 = (u32)r1.z; r.v3 = (u32)r1.w;
    if (FX) {                                         // integer arithmetic: c_ik * W_k, the addend of a count-1 entry of row k; the explicit
        const u64 g = (u64)P.A16[ec] * P.W[k];        // entries multiply it by their own count (xtile_consume)
        r.g_lo = (u32)g; r.g_hi = (u32)(g >> 32);
        r.da_lo = r.g_lo; r.da_hi = r.g_hi;
    } else {
        const double da = (double)P.Ax[ec] * P.scale;
        const u64 g = fx_bits(da * (double)__int_as_float(r1.y));
        r.g_lo = (u32)g; r.g_hi = (u32)(g >> 32);
        const u64 d = (u64)__double_as_longlong(da);
        r.da_lo = (u32)d; r.da_hi = (u32)(d >> 32);
    }
}

// wide tiles -------------------------------------------------------------------------------------------------
struct WTile { uint4 x0, x1; i32 hi; u32 g_lo, g_hi; bool valid; };       // valid positions lane * 8 + j (+ 512) < hi (a sub-segment starts on a tile boundary)
struct WCursor { i32 l, q, qb, qe; u32 g_lo, g_hi; };                     // all wave-uniform
__device__ __forceinline__ void wtile_fetch(const ExParams &P, const BatchRegs &r, i32 cnt, WCursor &c, WTile &t) {
    while (c.q >= c.qe && c.l + 1 < cnt) {            // next A entry of the batch with a uniform sub-segment
        ++c.l;
        c.qb = __builtin_amdgcn_readlane(r.b0, c.l);
        c.qe = __builtin_amdgcn_readlane(r.b1, c.l);
        c.g_lo = __builtin_amdgcn_readlane(r.g_lo, c.l);
        c.g_hi = __builtin_amdgcn_readlane(r.g_hi, c.l);
        c.q = c.qb & ~7;
    }
    t.valid = c.q < c.qe;
    t.hi = t.valid ? c.qe - c.q : 0;
    t.g_lo = c.g_lo; t.g_hi = c.g_hi;
    const i32 base = t.valid ? c.q : 0;
    const i32 e0 = base + lane_id() * 8, e1 = e0 + WIDE_UNIT;
    t.x0 = *reinterpret_cast<const uint4 *>(P.Sc16 + (lane_id() * 8 < t.hi ? e0 : lane_id() * 8));
    t.x1 = *reinterpret_cast<const uint4 *>(P.Sc16 + (lane_id() * 8 + WIDE_UNIT < t.hi ? e1 : lane_id() * 8));
    c.q += 2 * WIDE_UNIT;
}
// No range test per entry: a sub-segment starts on a tile boundary (64-slot aligned), the slots between its end and the next
// multiple of 8 hold the reading lane's scratch column (k_layout_write), and a lane whose 8 entries lie past the end loaded
// its block of the stream prefix instead — every column a lane holds is either real or its own scratch accumulator.
template <int PROBE>
__device__ __forceinline__ void wunit_consume(const ExLds &l, const uint4 &x, u64 g, u64 &sink) {
    const u32 w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u32 col = (j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu);
        if (PROBE == 1) sink += g + col;
        else atomicAdd((unsigned long long *)&l.acc[col], (unsigned long long)g);
    }
}
// (with count-2 / count-3 sub-segments behind it — hhx_tune("cls_nc") — the count-1 sub-segment is not padded: range test per entry)
template <int PROBE>
__device__ __forceinline__ void wunit_consume_masked(const ExLds &l, const uint4 &x, i32 pos0, i32 hi, u64 g, i32 dummy, u64 &sink) {
    const u32 w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u32 col = (j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu);
        const bool ok = pos0 + j < hi;
        if (PROBE == 1) sink += ok ? g + col : 0;
        else atomicAdd((unsigned long long *)&l.acc[ok ? (i32)col : dummy], (unsigned long long)g);
    }
}
template <int PROBE>
__device__ __forceinline__ void wtile_consume(const ExLds &l, const WTile &t, bool padded, i32 dummy, u64 &sink) {
    const u64 g = ((u64)t.g_hi << 32) | t.g_lo;
    if (padded) {
        wunit_consume<PROBE>(l, t.x0, g, sink);
        if (t.hi > WIDE_UNIT) wunit_consume<PROBE>(l, t.x1, g, sink);   // wave-uniform
    } else {
        wunit_consume_masked<PROBE>(l, t.x0, lane_id() * 8, t.hi, g, dummy, sink);
        if (t.hi > WIDE_UNIT) wunit_consume_masked<PROBE>(l, t.x1, lane_id() * 8 + WIDE_UNIT, t.hi, g, dummy, sink);
    }
}
// G tiles per group: the loads of the whole group are issued back to back, then the tiles are consumed in order while
// the later ones are still in flight (s_waitcnt vmcnt(2 (G - 1)), vmcnt(2 (G - 2)), ...).  Nothing that was loaded is
// carried across the loop's back edge: when a ring of tiles was kept in flight ACROSS iterations the register
// allocator copied loaded registers at the back edge, and every such copy is an s_waitcnt vmcnt(0).  The other waves
// of the CU cover the refill of a wave's group.
template <int K, int G, class Tile, class Fetch>
__device__ __forceinline__ void group_fetch(Tile (&t)[G], Fetch &fetch) {
    if constexpr (K < G) { fetch(t[K]); group_fetch<K + 1, G>(t, fetch); }
}
template <int K, int G, class Tile, class Consume>
__device__ __forceinline__ bool group_consume(const Tile (&t)[G], Consume &consume) {
    if constexpr (K == G) return true;
    else {
        if (!t[K].valid) return false;                // tiles are issued in order: the first invalid one ends the pass
        consume(t[K]);
        return group_consume<K + 1, G>(t, consume);
    }
}
template <int PROBE, int G>
__device__ __forceinline__ void pass_wide(const ExParams &P, const ExLds &l, const BatchRegs &r, i32 cnt, i32 dummy, u64 &sink) {
    WCursor c = {-1, 0, 0, 0, 0u, 0u};
    auto fetch = [&](WTile &t) { wtile_fetch(P, r, cnt, c, t); };
    const bool padded = P.narrow_classes == 0;          // count 1 is the only uniform class: its sub-segment is padded to whole lanes
    auto consume = [&](const WTile &t) { wtile_consume<PROBE>(l, t, padded, dummy, sink); };
    for (;;) {
        WTile t[G];
        group_fetch<0, G>(t, fetch);
        if (!group_consume<0, G>(t, consume)) return;
    }
}

// explicit tiles -----------------------------------------------------------------------------------------------
template <int UX>
struct XTile { u32 j[UX], v[UX]; i32 n; u32 da_lo, da_hi; bool valid; };      // entry u of the lane: position lane + 64 u < n
struct XCursor { i32 l, q, qe; u32 da_lo, da_hi; };
// FX: an explicit entry is ONE 32-bit word of Sx, link count << 16 | window-local column (4 B per product, one load)
template <int UX, bool FX>
__device__ __forceinline__ void xtile_fetch(const ExParams &P, const BatchRegs &r, i32 cnt, XCursor &c, XTile<UX> &t) {
    while (c.q >= c.qe && c.l + 1 < cnt) {
        ++c.l;
        c.q = (__builtin_amdgcn_readlane(r.b3, c.l) + 63) & ~63;      // the explicit part starts on the next multiple of 64 slots
        c.qe = __builtin_amdgcn_readlane(r.b4, c.l);
        c.da_lo = __builtin_amdgcn_readlane(r.da_lo, c.l);
        c.da_hi = __builtin_amdgcn_readlane(r.da_hi, c.l);
    }
    t.valid = c.q < c.qe;
    t.n = t.valid ? c.qe - c.q : 0;
    t.da_lo = c.da_lo; t.da_hi = c.da_hi;
    const i32 base = t.valid ? c.q : 0;
    if constexpr (FX) {                                            // iteration 0: one 32-bit word per entry; a masked position reads entry 0 of the array (one cached line)
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const i32 pos = lane_id() + u * HHX_WAVE;
            const u32 w = __float_as_uint(P.Sx[pos < t.n ? base + pos : 0]);
            t.j[u] = w & 0xffffu;
            t.v[u] = w >> 16;
        }
    } else {
        // the generic stream of the iterations >= 1: scalar base + 32-bit lane offset + immediate (global_load ... v_off, s[base] offset:u * 128) — a load costs a compare,
        // a select and (for the values) a shift instead of an add, a compare, a select, a sign extension and two 64-bit address adds.  A position past the end of the
        // segment reads the first entry of its block (one address for all such lanes, inside the arrays: they end with a tile of slack) and is masked where it is consumed.
        // (Not for iteration 0: there the blocks past the end pull lines of the next segment through a fabric that is the limit, + 5 % of its bytes; and choosing the base
        // per block with a scalar select serialises the loads of a tile: measured, 5.8 -> 7.3 s over the tail at 1.1.)
        const char *const bc = reinterpret_cast<const char *>(P.Sc16 + base), *const bx = reinterpret_cast<const char *>(P.Sx + base);
        u32 lane2 = (u32)lane_id() * 2u;
        asm volatile("" : "+v"(lane2));                           // opaque here: u * 128 stays an immediate of the load instead of a hoisted register per block
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const u32 off2 = lane_id() + u * HHX_WAVE < t.n ? lane2 : 0u;
            t.j[u] = (u32)*reinterpret_cast<const unsigned short *>(bc + (size_t)off2 + u * 128);
            t.v[u] = *reinterpret_cast<const u32 *>(bx + (size_t)(off2 * 2u) + u * 256);
        }
    }
    c.q += UX * HHX_WAVE;
}
template <int PROBE, int UX, bool FX>
__device__ __forceinline__ void xtile_consume(const ExLds &l, const XTile<UX> &t, i32 dummy, u64 &sink) {
    const u64 ga = ((u64)t.da_hi << 32) | t.da_lo;               // FX: c_ik * W_k; else the bits of double(a_ik * scale)
    const double da = __longlong_as_double((long long)ga);
#pragma unroll
    for (int u = 0; u < UX; ++u) {
        const bool ok = lane_id() + u * HHX_WAVE < t.n;
        const u64 g = FX ? ga * (u64)t.v[u] : fx_bits_prod(da, (double)__uint_as_float(t.v[u]));     // FX: the slot holds the link count c_kj
        if (PROBE == 1) sink += ok ? g + t.j[u] : 0;
        else atomicAdd((unsigned long long *)&l.acc[ok ? (i32)t.j[u] : dummy], (unsigned long long)g);
    }
}
template <int PROBE, int UX, int G, bool FX>
__device__ __forceinline__ void pass_explicit(const ExParams &P, const ExLds &l, const BatchRegs &r, i32 cnt, i32 dummy, u64 &sink) {
    XCursor c = {-1, 0, 0, 0u, 0u};
    auto fetch = [&](XTile<UX> &t) { xtile_fetch<UX, FX>(P, r, cnt, c, t); };
    auto consume = [&](const XTile<UX> &t) { xtile_consume<PROBE, UX, FX>(l, t, dummy, sink); };
    for (;;) {
        XTile<UX> t[G];
        group_fetch<0, G>(t, fetch);
        if (!group_consume<0, G>(t, consume)) return;
    }
}

// block tiles (the generic stream of the iterations >= 1; hhx_tune("block_tiles")) ------------------------------------------------
// An explicit tile above belongs to ONE (B row, window) segment: a segment of ~400 entries fills 6.3 of the 8 blocks of its tile, and over a low-inflation tail a third of
// all issued slots — loads, fixed-point products, LDS atomics — are masked positions (profiles/r06_tail_pmc.json).  Here the unit is the 64-entry BLOCK: the segments of
// the stream start on multiples of 64 slots and their last block is padded with (scratch column, 0.0f) entries (k_layout_write), so a block needs no range test, and a tile
// is any UX consecutive blocks of the batch's block list, whatever segments they belong to — the multiplier double(a_ik * scale) is taken per block (a scalar), not per tile.
// The block list of a batch (<= 32 segments, one per lane): an inclusive scan of the segments' block counts; then, CH blocks at a time, lane j finds the segment of block
// c0 + j by bisection over the scan (6 ds_bpermute) and keeps the block's descriptor (stream position, multiplier); the tiles read descriptors f .. f + UX - 1 with
// v_readlane.  Blocks past the end of the list are the EMPTY block: position 0 (the stream prefix: scratch columns), multiplier 0.  Every load of a group is issued
// unconditionally (exact wait counts), nothing between a tile's loads and its atomics depends on a lane.
template <int UX>
struct BTile { u32 j[UX], v[UX]; };
// tile = the blocks of descriptors f .. f + UX - 1 (wave-uniform f): scalar base per block + one 32-bit lane offset (global_load ... v_off, s[base:base+1])
template <int UX, int AM>
__device__ __forceinline__ void btile_fetch(const ExParams &P, i32 d_pos, i32 f, u32 lane2, u32 lane4, BTile<UX> &t) {
#pragma unroll
    for (int u = 0; u < UX; ++u) {
        const i32 pos = __builtin_amdgcn_readlane(d_pos, f + u);
        if constexpr (AM == 1) {                                      // one 32-bit offset from the start of the arrays (the host: the stream has fewer than 2^30 slots)
            const u32 o2 = lane2 + (u32)pos * 2u;
            t.j[u] = (u32)*reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(P.Sc16) + (size_t)o2);
            t.v[u] = *reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(P.Sx) + (size_t)(o2 * 2u));
            continue;
        }
        if constexpr (AM == 3) {                                      // the loads written by hand: scalar base per block, the two lane offsets of the pass as they are —
            // no VALU instruction for an address.  The compiler does not know these are loads: the consumer waits itself (btile_wait), and the values are first
            // touched THROUGH that wait.
            const char *const sc = reinterpret_cast<const char *>(P.Sc16) + ((size_t)(u32)pos << 1), *const sx = reinterpret_cast<const char *>(P.Sx) + ((size_t)(u32)pos << 2);
            asm volatile("global_load_ushort %0, %1, %2" : "=v"(t.j[u]) : "v"(lane2), "s"(sc));
            asm volatile("global_load_dword %0, %1, %2" : "=v"(t.v[u]) : "v"(lane4), "s"(sx));
            continue;
        }
        const char *const bc = reinterpret_cast<const char *>(P.Sc16 + pos), *const bx = reinterpret_cast<const char *>(P.Sx + pos);
        u32 off = lane2;
        asm volatile("" : "+v"(off));                                 // opaque per block: or (Sc16 + lane offset) is hoisted as a 64-bit VGPR pair and every load pays a 64-bit VALU add
        t.j[u] = (u32)*reinterpret_cast<const unsigned short *>(bc + (size_t)off);
        t.v[u] = *reinterpret_cast<const u32 *>(bx + (size_t)(off * 2u));
    }
}
// AFTER >= 0 (hand-written loads): the number of loads issued behind this tile's last one; block u is touched when at most AFTER + 2 (UX - 1 - u) loads are outstanding
template <int PROBE, int UX, int AFTER>
__device__ __forceinline__ void btile_consume(const ExLds &l, u32 d_lo, u32 d_hi, i32 f, BTile<UX> &t, u64 &sink) {
#pragma unroll
    for (int u = 0; u < UX; ++u) {
        if constexpr (AFTER >= 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(t.j[u]), "+v"(t.v[u]) : "n"(AFTER + 2 * (UX - 1 - u)));
        const u64 ga = ((u64)(u32)__builtin_amdgcn_readlane((int)d_hi, f + u) << 32) | (u32)__builtin_amdgcn_readlane((int)d_lo, f + u);
        const u64 g = fx_bits_prod(__longlong_as_double((long long)ga), (double)__uint_as_float(t.v[u]));
        if (PROBE == 1) sink += g + t.j[u];
        else atomicAdd((unsigned long long *)&l.acc[t.j[u]], (unsigned long long)g);
    }
}
template <int K, int G, int PROBE, int UX, int AM>
__device__ __forceinline__ void group_consume_blocks(const ExLds &l, u32 d_lo, u32 d_hi, i32 f, BTile<UX> (&t)[G], u64 &sink) {
    if constexpr (K < G) {
        btile_consume<PROBE, UX, AM == 3 ? 2 * UX * (G - 1 - K) : -1>(l, d_lo, d_hi, f + K * UX, t[K], sink);
        group_consume_blocks<K + 1, G, PROBE, UX, AM>(l, d_lo, d_hi, f, t, sink);
    }
}
template <int PROBE, int UX, int G, int AM>
__device__ __forceinline__ void pass_blocks(const ExParams &P, const ExLds &l, const BatchRegs &r, i32 cnt, u64 &sink) {
    constexpr int CH = 64 / (UX * G) * (UX * G);                       // blocks described at a time: whole groups of tiles
    const i32 lane = lane_id();
    const i32 qb = ceil64(r.b3);
    const i32 nblk = lane < cnt && r.b4 > qb ? (r.b4 - qb + 63) >> 6 : 0;
    i32 incl = nblk;
#pragma unroll
    for (int o = 1; o < HHX_WAVE; o <<= 1) {
        const i32 t = __shfl_up(incl, o, HHX_WAVE);
        if (lane >= o) incl += t;
    }
    const i32 excl = incl - nblk;
    const i32 total = __builtin_amdgcn_readlane(incl, HHX_WAVE - 1);
    u32 lane2 = (u32)lane * 2u;
    asm volatile("" : "+v"(lane2));                                   // opaque: one offset register for every load of the pass
    const u32 lane4 = lane2 * 2u;
    for (i32 c0 = 0; c0 < total; c0 += CH) {
        const i32 B = c0 + lane;
        i32 sg = 0;                                                   // the first segment whose inclusive count exceeds B (exists while B < total)
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
            if (__shfl(incl, sg + step - 1, HHX_WAVE) <= B) sg += step;
        const bool real = B < total;
        const i32 p_real = __shfl(qb, sg, HHX_WAVE) + ((B - __shfl(excl, sg, HHX_WAVE)) << 6);
        const u32 s_lo = (u32)__shfl((int)r.da_lo, sg, HHX_WAVE), s_hi = (u32)__shfl((int)r.da_hi, sg, HHX_WAVE);
        const i32 d_pos = real ? p_real : 0;                          // past the list: the empty block
        const u32 d_lo = real ? s_lo : 0u, d_hi = real ? s_hi : 0u;
        const i32 nb = min(CH, total - c0);
        i32 f = 0;
        // whole groups: the loads of G tiles back to back, then the tiles in order — one basic block, no test of any kind between a load and its atomic
        for (; f + UX * G <= nb; f += UX * G) {
            BTile<UX> t[G];
#pragma unroll
            for (int k = 0; k < G; ++k) btile_fetch<UX, AM>(P, d_pos, f + k * UX, lane2, lane4, t[k]);
            group_consume_blocks<0, G, PROBE, UX, AM>(l, d_lo, d_hi, f, t, sink);
        }
        // what is left of the list (fewer than G tiles, once per batch): a tile at a time; its blocks past the list are the empty block
        for (; f < nb; f += UX) {
            BTile<UX> t;
            btile_fetch<UX, AM>(P, d_pos, f, lane2, lane4, t);
            btile_consume<PROBE, UX, AM == 3 ? 0 : -1>(l, d_lo, d_hi, f, t, sink);
        }
    }
}

// narrow uniform tiles: the count-2 and count-3 sub-segments (a few dozen entries each) ------------------------------
// 16-bit columns only, UN entries per lane; the value of the sub-segment is wave-uniform, its product is formed in the fetch
template <int UN>
struct NTile { u32 j[UN]; i32 n; u32 g_lo, g_hi; bool valid; };
struct NCursor { i32 l, c, q, qe; u32 g_lo, g_hi; };                      // c: 0 = count-2 part next, 1 = count-3 part next
template <int UN>
__device__ __forceinline__ void ntile_fetch(const ExParams &P, const BatchRegs &r, i32 cnt, NCursor &c, NTile<UN> &t) {
    while (c.q >= c.qe && (c.c < 2 || c.l + 1 < cnt)) {
        if (c.c >= 2) { ++c.l; c.c = 0; }
        const i32 lo = __builtin_amdgcn_readlane(c.c ? r.b2 : r.b1, c.l), hi = __builtin_amdgcn_readlane(c.c ? r.b3 : r.b2, c.l);
        if (hi > lo) {
            const double da = __longlong_as_double((long long)(((u64)(u32)__builtin_amdgcn_readlane(r.da_hi, c.l) << 32) |
                                                                 (u32)__builtin_amdgcn_readlane(r.da_lo, c.l)));
            const u64 g = fx_bits(da * (double)__uint_as_float((u32)__builtin_amdgcn_readlane(c.c ? r.v3 : r.v2, c.l)));
            c.g_lo = (u32)g; c.g_hi = (u32)(g >> 32);
        }
        c.q = lo; c.qe = hi;
        ++c.c;
    }
    t.valid = c.q < c.qe;
    t.n = t.valid ? c.qe - c.q : 0;
    t.g_lo = c.g_lo; t.g_hi = c.g_hi;
    const i32 base = t.valid ? c.q : 0;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const i32 pos = lane_id() + u * HHX_WAVE;
        t.j[u] = (u32)P.Sc16[pos < t.n ? base + pos : 0];
    }
    c.q += UN * HHX_WAVE;
}
template <int PROBE, int UN>
__device__ __forceinline__ void ntile_consume(const ExLds &l, const NTile<UN> &t, i32 dummy, u64 &sink) {
    const u64 g = ((u64)t.g_hi << 32) | t.g_lo;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        if (u && u * HHX_WAVE >= t.n) break;                    // wave-uniform: the rest of the tile is empty
        const bool ok = lane_id() + u * HHX_WAVE < t.n;
        if (PROBE == 1) sink += ok ? g + t.j[u] : 0;
        else atomicAdd((unsigned long long *)&l.acc[ok ? (i32)t.j[u] : dummy], (unsigned long long)g);
    }
}
template <int PROBE, int UN, int G>
__device__ __forceinline__ void pass_narrow(const ExParams &P, const ExLds &l, const BatchRegs &r, i32 cnt, i32 dummy, u64 &sink) {
    NCursor c = {0, 0, 0, 0, 0u, 0u};
    if (cnt <= 0) return;
    auto fetch = [&](NTile<UN> &t) { ntile_fetch<UN>(P, r, cnt, c, t); };
    auto consume = [&](const NTile<UN> &t) { ntile_consume<PROBE, UN>(l, t, dummy, sink); };
    for (;;) {
        NTile<UN> t[G];
        group_fetch<0, G>(t, fetch);
        if (!group_consume<0, G>(t, consume)) return;
    }
}

// compact mode: mark, then acc[rank(c)] += fixed(a * b)
__device__ __forceinline__ void mark_compact(const ExParams &P, const ExLds &l, i32 len) {
    const int lane = lane_id(), wave = threadIdx.x / HHX_WAVE;
    for (i32 e = wave; e < len; e += EX_WAVES) {
        const i32 qb = l.st_qb[e], qe = l.st_qe[e];
        i32 q = qb + lane;
        for (; q + 3 * HHX_WAVE < qe; q += 4 * HHX_WAVE) {
            const i32 j0 = P.Bj[q], j1 = P.Bj[q + HHX_WAVE], j2 = P.Bj[q + 2 * HHX_WAVE], j3 = P.Bj[q + 3 * HHX_WAVE];
            atomicOr(&l.bitmap[j0 >> 5], 1u << (j0 & 31));
            atomicOr(&l.bitmap[j1 >> 5], 1u << (j1 & 31));
            atomicOr(&l.bitmap[j2 >> 5], 1u << (j2 & 31));
            atomicOr(&l.bitmap[j3 >> 5], 1u << (j3 & 31));
        }
        for (; q < qe; q += HHX_WAVE) {
            const i32 j = P.Bj[q];
            atomicOr(&l.bitmap[j >> 5], 1u << (j & 31));
        }
    }
}
__device__ __forceinline__ i32 rank_of(const ExLds &l, i32 c) {
    return (i32)(l.prefix[c >> 5] + __popc(l.bitmap[c >> 5] & ((1u << (c & 31)) - 1u)));
}
__device__ __forceinline__ void accumulate_compact(const ExParams &P, const ExLds &l, i32 len, i32 r0, i32 rlen) {
    const int lane = lane_id(), wave = threadIdx.x / HHX_WAVE;
    for (i32 e = wave; e < len; e += EX_WAVES) {
        const double da = l.st_da[e];
        const i32 qb = l.st_qb[e], qe = l.st_qe[e];
        i32 q = qb + lane;
        for (; q + 3 * HHX_WAVE < qe; q += 4 * HHX_WAVE) {
            const i32 j0 = P.Bj[q], j1 = P.Bj[q + HHX_WAVE], j2 = P.Bj[q + 2 * HHX_WAVE], j3 = P.Bj[q + 3 * HHX_WAVE];
            const float b0 = P.Bx[q], b1 = P.Bx[q + HHX_WAVE], b2 = P.Bx[q + 2 * HHX_WAVE], b3 = P.Bx[q + 3 * HHX_WAVE];
            const i32 s0 = rank_of(l, j0) - r0, s1 = rank_of(l, j1) - r0, s2 = rank_of(l, j2) - r0, s3 = rank_of(l, j3) - r0;
            if ((u32)s0 < (u32)rlen) acc_add(&l.acc[s0], da * (double)b0);
            if ((u32)s1 < (u32)rlen) acc_add(&l.acc[s1], da * (double)b1);
            if ((u32)s2 < (u32)rlen) acc_add(&l.acc[s2], da * (double)b2);
            if ((u32)s3 < (u32)rlen) acc_add(&l.acc[s3], da * (double)b3);
        }
        for (; q < qe; q += HHX_WAVE) {
            const i32 s = rank_of(l, P.Bj[q]) - r0;
            if ((u32)s < (u32)rlen) acc_add(&l.acc[s], da * (double)P.Bx[q]);
        }
    }
}

// popcount prefix over the LDS bitmap (256 threads, contiguous chunk of words per thread)
__device__ __forceinline__ i32 bitmap_prefix_total(const ExLds &l, i32 W) {
    const int tid = threadIdx.x;
    const i32 per = (W + EX_T - 1) / EX_T;
    const i32 w0 = min(W, tid * per), w1 = min(W, w0 + per);
    i32 local = 0;
    for (i32 w = w0; w < w1; ++w) local += __popc(l.bitmap[w]);
    i32 total;
    i32 run = block_excl_scan_i32(local, l.red_i, &total);
    for (i32 w = w0; w < w1; ++w) { l.prefix[w] = (u32)run; run += __popc(l.bitmap[w]); }
    __syncthreads();
    return total;
}

// ---- window epilogue: slots [0, wlen) hold the accumulators of columns col_of(slot) (ascending).
// Turns them into p = x^r (stored back as float bits, -1 = absent), returns the window sum.
// FROM_X: the slot's low word already holds x as float32 bits (k_dense_epilogue: the expanded row comes back from HBM)
// row_div != 0: the integer arithmetic of the link matrix — the slot holds acc (or, FROM_X, y = float(acc * 2^-s)) and x = float(y / d_i)
// SQUARE: inflation 2 known at compile time (no pow call in the instantiation: half the registers, two workgroups per CU)
template <bool COMPACT, bool FROM_X = false, bool SQUARE = false>
__device__ __forceinline__ double window_power_sum(const ExParams &P, const ExLds &l, i32 wlen, i32 *nnz_local, double row_div = 0.0) {
    const int tid = threadIdx.x;
    const i32 per = (wlen + EX_T - 1) / EX_T;
    const i32 s0 = min(wlen, tid * per), s1 = min(wlen, s0 + per);
    double s = 0.0;
    i32 nz = 0;
    for (i32 t = s0; t < s1; ++t) {
        const u64 ai = FROM_X ? (u64)((const u32 *)l.acc)[t] : l.acc[t];      // exact: the sum is below 2^53 (FROM_X: 4-byte slots of float bits)
        float p = -1.0f;
        if (COMPACT || ai != 0) {
            float x;
            if (row_div != 0.0) {
                const float y = FROM_X ? __uint_as_float((u32)ai) : (float)((double)(long long)ai * P.fx_inv);
                x = (float)((double)y / row_div);
            } else x = FROM_X ? __uint_as_float((u32)ai) : (float)((double)(long long)ai * 0x1p-52 * P.inv_scale);
            p = SQUARE ? x * x : (P.raw ? x : ex_inflate(x, P.r, P.square));
            s += (double)p;
            ++nz;
        }
        ((float *)l.acc)[FROM_X ? t : 2 * t] = p;
    }
    *nnz_local = nz;
    return block_sum_f64(s, l.red_d);
}

// column of slot t: window mode c0 + t; compact mode: enumerate the bitmap (thread-local walk)
struct BitWalk {
    i32 w, base; u32 bits;
};

// emits the candidates of one window (entries that may still survive given the running row sum, plus
// the window maximum) in column order; returns their count (uniform) and records (offset,count)
// SS: floats per slot (2: the 8-byte accumulator slots with p in the low word; 1: the 4-byte slots of k_dense_epilogue)
template <bool COMPACT, int SS = 2>
__device__ __forceinline__ void window_emit_candidates(const ExParams &P, const ExLds &l, i32 wlen, i32 c0, i32 r0,
                                                       double s_run, i64 *seg_off, i32 *seg_cnt) {
    const int tid = threadIdx.x;
    const i32 per = (wlen + EX_T - 1) / EX_T;
    const i32 s0 = min(wlen, tid * per), s1 = min(wlen, s0 + per);
    // window maximum (first by column == lowest slot)
    float bq = -1.0f; i32 bs = 0x7fffffff;
    for (i32 t = s0; t < s1; ++t) {
        const float p = ((const float *)l.acc)[SS * t];
        if (p > bq) { bq = p; bs = t; }
    }
    block_argmax(bq, bs, l.red_f, l.red_i);
    i32 cnt = 0;
    for (i32 t = s0; t < s1; ++t) {
        const float p = ((const float *)l.acc)[SS * t];
        if (p >= 0.0f && (P.raw || t == bs || (float)((double)p / s_run) >= P.thr)) ++cnt;
    }
    i32 total;
    i32 off = block_excl_scan_i32(cnt, l.red_i, &total);
    __syncthreads();
    if (tid == 0) {
        i64 base = 0;
        if (total) {
            base = (i64)atomicAdd(&P.cursors[0], (unsigned long long)total);
            if (base + total > P.cand_cap) { atomicExch(&P.cursors[2], 1ull); base = -1; }
        }
        *seg_off = base;                       // LDS (compact kernel) or HBM (window passes)
        *seg_cnt = base < 0 ? 0 : total;
        *l.bcast = base;
    }
    __syncthreads();
    const i64 base = *l.bcast;
    if (base < 0 || total == 0) return;
    // column of slot s0 in compact mode: find the (s0 + r0)-th set bit
    i32 w = 0; u32 bits = 0;
    if (COMPACT && s0 < s1) {
        const i32 target = s0 + r0;
        i32 lo = 0, hi = (P.n_cols + 31) / 32;        // last word with prefix <= target
        while (hi - lo > 1) { const i32 m = (lo + hi) >> 1; if ((i32)l.prefix[m] <= target) lo = m; else hi = m; }
        w = lo; bits = l.bitmap[w];
        for (i32 skip = target - (i32)l.prefix[w]; skip > 0; --skip) bits &= bits - 1;
    }
    i64 o = base + off;
    for (i32 t = s0; t < s1; ++t) {
        i32 col;
        if (COMPACT) {
            while (!bits) { ++w; bits = l.bitmap[w]; }
            col = (w << 5) + (__ffs(bits) - 1);
            bits &= bits - 1;
        } else col = c0 + t;
        const float p = ((const float *)l.acc)[SS * t];
        if (p >= 0.0f && (P.raw || t == bs || (float)((double)p / s_run) >= P.thr)) {
            P.cand_col[o] = col;
            P.cand_val[o] = p;
            ++o;
        }
    }
}

// ---- row finalisation over the candidate segments -------------------------------------------------
__device__ __forceinline__ void finalize_row(const ExParams &P, const ExLds &l, i32 row, i32 n_win, double s1,
                                             const i64 *win_off, const i32 *win_cnt) {
    const int tid = threadIdx.x;
    if (P.raw) s1 = 0.0;                              // plain product: q = p, every candidate is kept, no normalisation
    __threadfence_block();
    __syncthreads();
    // pass A: first row maximum of q = float(p / S)
    float bq = -1.0f; i32 bc = 0x7fffffff;
    for (i32 wv = 0; wv < n_win; ++wv) {
        const i64 base = win_off[wv];
        const i32 cnt = win_cnt[wv];
        for (i32 t = tid; t < cnt; t += EX_T) {
            const float p = P.cand_val[base + t];
            const float q = s1 != 0.0 ? (float)((double)p / s1) : p;
            const i32 c = P.cand_col[base + t];
            if (q > bq || (q == bq && c < bc)) { bq = q; bc = c; }
        }
    }
    block_argmax(bq, bc, l.red_f, l.red_i);
    // pass B: survivor count and second-normalisation sum (candidate order == column order)
    i32 keep = 0;
    double s2 = 0.0;
    for (i32 wv = 0; wv < n_win; ++wv) {
        const i64 base = win_off[wv];
        const i32 cnt = win_cnt[wv];
        for (i32 t0 = 0; t0 < cnt; t0 += EX_T) {
            const i32 t = t0 + tid;
            double v = 0.0;
            if (t < cnt) {
                const float p = P.cand_val[base + t];
                const float q = s1 != 0.0 ? (float)((double)p / s1) : p;
                if (P.raw || q >= P.thr || P.cand_col[base + t] == bc) { ++keep; v = (double)q; }
            }
            s2 += block_sum_f64(v, l.red_d);        // chunk sums added in order: deterministic
        }
    }
    const i32 total = block_sum_i32(keep, l.red_i);
    __syncthreads();
    if (tid == 0) {
        i64 base = 0;
        if (total) {
            base = (i64)atomicAdd(&P.cursors[1], (unsigned long long)total);
            if (base + total > P.out_cap) { atomicExch(&P.cursors[2], 1ull); base = -1; }
        }
        P.row_off[row] = base;
        P.row_cnt[row] = base < 0 ? 0 : total;
        *l.bcast = base;
    }
    __syncthreads();
    i64 o = *l.bcast;
    __syncthreads();
    if (o < 0 || total == 0) return;
    // pass C: ordered write of (column, float(q / S2))
    for (i32 wv = 0; wv < n_win; ++wv) {
        const i64 base = win_off[wv];
        const i32 cnt = win_cnt[wv];
        for (i32 t0 = 0; t0 < cnt; t0 += EX_T) {
            const i32 t = t0 + tid;
            bool k = false;
            float q = 0.f;
            i32 c = 0;
            if (t < cnt) {
                const float p = P.cand_val[base + t];
                q = s1 != 0.0 ? (float)((double)p / s1) : p;
                c = P.cand_col[base + t];
                k = P.raw || (q >= P.thr) || (c == bc);
            }
            i32 tot;
            const i32 pos = block_excl_scan_i32(k ? 1 : 0, l.red_i, &tot);
            if (k) {
                P.out_col[o + pos] = c;
                P.out_val[o + pos] = (s2 != 0.0 && !P.raw) ? (float)((double)q / s2) : q;
            }
            o += tot;
            __syncthreads();
        }
    }
}

// ---- the dense block of the expanded rows: square (every row n floats, `ld` apart) or upper block triangle ----------------------
__host__ __device__ inline i64 tri_row_off(i32 I, i32 cap, i64 ldn) { return (i64)cap * ((i64)I * ldn - (i64)cap * I * (I - 1) / 2); }
__host__ __device__ inline i64 tri_floats(i32 n_win, i32 cap) { return (i64)cap * cap * n_win * (n_win + 1) / 2; }
// where the epilogue finds (row, window): windows >= up_win0 in `up` (rows up_ld apart, its first window is up_win0), the windows
// before in `lo` (rows lo_ld apart) — the square block is up_win0 = 0; a block row I of the triangle is up_win0 = I with `lo` the
// transposed blocks (J < I, I) laid side by side.  Rows [row0, row1); row `row0` is row 0 of both.
struct DenseSrc {
    const float *up; i64 up_ld; i32 up_win0;
    const float *lo; i64 lo_ld;
    i32 row0, row1;
    __device__ __forceinline__ const float *at(i32 row, i32 wv, i32 cap) const {
        return wv >= up_win0 ? up + (size_t)(row - row0) * (size_t)up_ld + (size_t)(wv - up_win0) * cap
                             : lo + (size_t)(row - row0) * (size_t)lo_ld + (size_t)wv * cap;
    }
};

// ---- the kernels ----------------------------------------------------------------------------------
// Window class.  The loop nest is WINDOW-OUTER: one launch per column window, every launch sweeping all
// the rows.  While window w is being processed the only part of B that is read is its column slice
// B[:, w] (nnz_B / n_win entries: at n = 100k the class stream of one slice fits the 256 MiB Infinity Cache),
// and the per-row epilogue (finalize) becomes its own uniform launch.
// UX / RX: entries per lane of an explicit tile and explicit tiles per group; RW: wide tiles per group.  Measured at
// n = 100k, iteration 0 (tools/expand_probe.py): groups of 3 wide / 8 explicit tiles 590 ms, 3 / 5 597, 2 / 3 636; 512-thread
// workgroups with twice the group sizes 806 ms — the sixteen waves per CU, not the depth of a wave's group, hide the latency.
template <int PROBE, int UX, int RX, int RW, bool FX = false, int T = EX_T_WIN, bool BLK = false>
__global__ __launch_bounds__(T) void k_expand_window(ExParams P, const i32 *__restrict__ rows, i32 n_list, i32 cap, i32 wv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ExLds l = win_carve(smem, cap);
    const int tid = threadIdx.x;
    i64 nnzc = 0, n_uni = 0;                    // n_uni: products streamed as 16-bit columns only (cursors[7])
    i64 n_prod = 0, n_a = 0;                    // products walked / A entries staged by this launch (cursors[8], [9])
    u64 sink = 0;
    const i32 n_win = P.n_win;
    const i32 c0 = wv * cap, c1 = min(P.n_cols, c0 + cap), wlen = c1 - c0;
    const i32 dummy = cap + lane_id();
    for (i32 li = blockIdx.x; li < n_list; li += gridDim.x) {
        const i32 row = rows ? rows[li] : li;
        const i32 a_b = P.Ap[row], a_e = P.Ap[row + 1];
        const i32 n_batches = (a_e - a_b + P.wb - 1) / P.wb;
        if (tid == 0) l.ctr[0] = 0;
        for (i32 t = tid; t < wlen; t += EX_T) l.acc[t] = 0;
        __syncthreads();
        // every wave: draw a batch, prefetch the next one, walk the current one
        i32 batch = 0;
        if (lane_id() == 0) batch = atomicAdd(&l.ctr[0], 1);
        batch = __builtin_amdgcn_readfirstlane(batch);
        BatchRegs nxt;
        if (batch < n_batches) batch_load<FX>(P, a_b, a_e, batch, wv, nxt);
        while (batch < n_batches) {
            const BatchRegs cur = nxt;
            const i32 cnt = min(P.wb, a_e - (a_b + batch * P.wb));
            i32 nb = 0;
            if (lane_id() == 0) nb = atomicAdd(&l.ctr[0], 1);
            nb = __builtin_amdgcn_readfirstlane(nb);
            if (nb < n_batches) batch_load<FX>(P, a_b, a_e, nb, wv, nxt);
            n_uni += cur.b3 - cur.b0;
            n_prod += (cur.b3 - cur.b0) + max(0, cur.b4 - ceil64(cur.b3));
            n_a += lane_id() == 0 ? cnt : 0;
            if constexpr (BLK) pass_blocks<PROBE, UX, RX, RW>(P, l, cur, cnt, sink);              // a general operand (narrow_classes == -1) in 64-entry blocks
            else {
            if (P.narrow_classes >= 0) pass_wide<PROBE, RW>(P, l, cur, cnt, dummy, sink);      // -1: a general operand, every segment is explicit
            if (!FX && P.narrow_classes > 0) pass_narrow<PROBE, 2, 8>(P, l, cur, cnt, dummy, sink);
            pass_explicit<PROBE, UX, RX, FX>(P, l, cur, cnt, dummy, sink);
            }
            batch = nb;
        }
        __syncthreads();
        if (P.dense) {                              // the sweep: the expanded row leaves as float32, inflation-independent
            float *dst;
            if (P.tri) {                            // upper block triangle: this launch (window wv) only sees rows of the blocks I <= wv
                const i32 I = row / cap;
                dst = P.dense + tri_row_off(I, cap, P.tri_ldn) + (size_t)(row - I * cap) * (size_t)(P.tri_ldn - (i64)I * cap) + (size_t)(wv - I) * cap;
            } else dst = P.dense + (size_t)row * (size_t)P.dense_ld + c0;
            for (i32 t = tid; t < wlen; t += EX_T) {
                const u64 ai = l.acc[t];
                // float arithmetic: the x of window_power_sum, bit for bit.  Integer arithmetic: y = float(S_ij), symmetric — the
                // division by d_i is the epilogue's (k_dense_epilogue)
                dst[t] = FX ? (float)((double)(long long)ai * P.fx_inv) : (float)((double)(long long)ai * 0x1p-52 * P.inv_scale);
                nnzc += ai != 0 ? (P.sym && (P.sym_row0 + row) / cap != wv ? 2 : 1) : 0;       // symmetric mode: an off-diagonal block stands for its mirror image too
            }
            __syncthreads();
            continue;
        }
        i32 nz;
        const double sw = window_power_sum<false>(P, l, wlen, &nz, FX ? P.row_div[row] : 0.0);
        nnzc += nz;
        const double s_run = (wv == 0 ? 0.0 : P.s_run[row]) + sw;        // windows are launched in order
        window_emit_candidates<false>(P, l, wlen, c0, 0, s_run, &P.g_win_off[(size_t)row * n_win + wv],
                                      &P.g_win_cnt[(size_t)row * n_win + wv]);
        __syncthreads();
        if (tid == 0) P.s_run[row] = s_run;
    }
    nnzc = wave_sum_i64(nnzc);          // products are counted by the classification pass (cursors[4])
    if (lane_id() == 0 && nnzc) atomicAdd(&P.cursors[3], (unsigned long long)nnzc);
    n_uni = wave_sum_i64(n_uni);
    if (lane_id() == 0 && n_uni) atomicAdd(&P.cursors[7], (unsigned long long)n_uni);
    n_prod = wave_sum_i64(n_prod);
    if (lane_id() == 0 && n_prod) atomicAdd(&P.cursors[8], (unsigned long long)n_prod);
    n_a = wave_sum_i64(n_a);
    if (lane_id() == 0 && n_a) atomicAdd(&P.cursors[9], (unsigned long long)n_a);
    if (PROBE == 1 && sink == 0x123456789abcdefull) l.acc[0] = sink;
}

// ---- RE-USE of B rows across output rows (iterations >= 1, generic stream) ------------------------------------------------------
// The generic-stream window kernel above streams 6 B per product at the fabric's ceiling (1.1e12 products/s, §4.4 of DESIGN.md): the
// low-inflation tails of run_mcl_clustering's sweep are bound by the bytes of B they walk.  From iteration 2 on the rows of T that
// share an attractor (the column of their maximum) have nearly the same pattern (tools/lowtails.py --reuse: the union of four such
// rows' patterns holds 0.27-0.37 of the entries the four hold one by one), and output row i = sum_k a_ik B_k — so a workgroup that
// accumulates R output rows at once walks every B row of the UNION of their patterns once instead of once per row.
//   * the rows of the window class are sorted by attractor and taken R at a time (same attractor only; hhx_expand_impl);
//   * k_group_count / k_group_fill merge the R patterns of a group (LDS bitmap + rank) into a union row with R values per entry
//     (0 where a member has no entry);
//   * k_expand_group<R>: R accumulator windows of cap columns side by side in LDS (the column windows are R times narrower), one
//     (column, value) load of B per R products, R exact fixed-point adds (a member whose weight is 0 is skipped: the weights are
//     wave-uniform), then the window epilogue of k_expand_window once per member over its own accumulators.
// The sums are the same exact integers; the epilogue is the same code per (row, window).
struct GroupOp {
    const i32 *rows;        // [n_groups][R] member rows (-1: none)
    const i32 *Gp;          // [n_groups + 1] union rows
    const i32 *Gj;          // union columns k
    const float *Gx;        // [entries][R] a_ik of every member (0: no entry)
    i32 n_groups;
};
// one wave per row: key = column of the row maximum (first by column), value = the row
__global__ __launch_bounds__(256) void k_row_attractor(i32 n_list, const i32 *__restrict__ rows, const i32 *__restrict__ Ap, const i32 *__restrict__ Aj,
                                                       const float *__restrict__ Ax, u64 *__restrict__ key, u64 *__restrict__ val) {
    const int lane = lane_id();
    for (i32 li = blockIdx.x * 4 + threadIdx.x / HHX_WAVE; li < n_list; li += gridDim.x * 4) {
        const i32 row = rows[li];
        float bq = -1.0f; i32 bc = 0x7fffffff;
        for (i32 p = Ap[row] + lane; p < Ap[row + 1]; p += HHX_WAVE) {
            const float x = Ax[p];
            const i32 c = Aj[p];
            if (x > bq || (x == bq && c < bc)) { bq = x; bc = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float oq = __shfl_down(bq, o, HHX_WAVE);
            const i32 oc = __shfl_down(bc, o, HHX_WAVE);
            if (oq > bq || (oq == bq && oc < bc)) { bq = oq; bc = oc; }
        }
        if (lane == 0) { key[li] = (u64)(u32)bc; val[li] = (u64)(u32)row; }
    }
}
// min-hash of a row's column pattern: two rows share it with the probability of the Jaccard index of their patterns, so the rows of one cluster — of one chromosome,
// before the clusters have formed — sort next to each other (hhx_expand_impl: order_rows)
__global__ __launch_bounds__(256) void k_row_minhash(i32 n_list, const i32 *__restrict__ rows, const i32 *__restrict__ Ap, const i32 *__restrict__ Aj,
                                                     u64 *__restrict__ key, u64 *__restrict__ val) {
    const int lane = lane_id();
    for (i32 li = blockIdx.x * 4 + threadIdx.x / HHX_WAVE; li < n_list; li += gridDim.x * 4) {
        const i32 row = rows[li];
        u32 h = 0xffffffffu;
        for (i32 p = Ap[row] + lane; p < Ap[row + 1]; p += HHX_WAVE) h = min(h, (u32)Aj[p] * 0x9e3779b1u);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) h = min(h, (u32)__shfl_down((int)h, o, HHX_WAVE));
        if (lane == 0) { key[li] = (u64)h; val[li] = (u64)(u32)row; }
    }
}
__global__ __launch_bounds__(256) void k_rows_from_sorted(i32 n, const u64 *__restrict__ sorted_rows, i32 *__restrict__ list) {
    for (i32 li = blockIdx.x * blockDim.x + threadIdx.x; li < n; li += gridDim.x * blockDim.x) list[li] = (i32)sorted_rows[li];
}
// union pattern of a group: bitmap in LDS (dynamic: W words of bits + W words of prefix + scan scratch)
template <bool FILL>
__global__ __launch_bounds__(256) void k_group_union(i32 n_groups, i32 R, const i32 *__restrict__ grp_rows, const i32 *__restrict__ Ap,
                                                     const i32 *__restrict__ Aj, const float *__restrict__ Ax, i32 W, i32 *__restrict__ cnt,
                                                     const i32 *__restrict__ Gp, i32 *__restrict__ Gj, float *__restrict__ Gx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *bitmap = (u32 *)smem, *prefix = bitmap + W;
    i32 *red = (i32 *)(prefix + W);
    const int tid = threadIdx.x;
    for (i32 g = blockIdx.x; g < n_groups; g += gridDim.x) {
        for (i32 w = tid; w < W; w += 256) bitmap[w] = 0;
        __syncthreads();
        for (int r = 0; r < R; ++r) {
            const i32 row = grp_rows[(size_t)g * R + r];
            if (row < 0) continue;
            for (i32 p = Ap[row] + tid; p < Ap[row + 1]; p += 256) { const i32 c = Aj[p]; atomicOr(&bitmap[c >> 5], 1u << (c & 31)); }
        }
        __syncthreads();
        const i32 per = (W + 255) / 256;
        const i32 w0 = min(W, tid * per), w1 = min(W, w0 + per);
        i32 local = 0;
        for (i32 w = w0; w < w1; ++w) local += __popc(bitmap[w]);
        // exclusive scan over the 256 threads
        i32 incl = local;
#pragma unroll
        for (int o = 1; o < HHX_WAVE; o <<= 1) { const i32 t = __shfl_up(incl, o, HHX_WAVE); if (lane_id() >= o) incl += t; }
        if (lane_id() == HHX_WAVE - 1) red[tid / HHX_WAVE] = incl;
        __syncthreads();
        i32 off = 0, total = 0;
        for (int k = 0; k < 4; ++k) { if (k < tid / HHX_WAVE) off += red[k]; total += red[k]; }
        i32 run = off + incl - local;
        if (!FILL) {
            if (tid == 0) cnt[g] = total;
            __syncthreads();
            continue;
        }
        const i32 base = Gp[g];
        for (i32 w = w0; w < w1; ++w) {
            prefix[w] = (u32)run;
            u32 bits = bitmap[w];
            while (bits) { const int b = __ffs(bits) - 1; bits &= bits - 1; Gj[base + run] = (w << 5) + b; ++run; }
        }
        for (i32 t = tid; t < total * R; t += 256) Gx[(size_t)base * R + t] = 0.0f;
        __syncthreads();
        for (int r = 0; r < R; ++r) {
            const i32 row = grp_rows[(size_t)g * R + r];
            if (row < 0) continue;
            for (i32 p = Ap[row] + tid; p < Ap[row + 1]; p += 256) {
                const i32 c = Aj[p];
                const i32 rk = (i32)(prefix[c >> 5] + __popc(bitmap[c >> 5] & ((1u << (c & 31)) - 1u)));
                Gx[((size_t)base + rk) * R + r] = Ax[p];
            }
        }
        __syncthreads();
    }
}

template <int R>
struct GBatch { i32 b3, b4; u32 da_lo[R], da_hi[R]; };             // one union entry per lane
template <int R>
__device__ __forceinline__ void gbatch_load(const ExParams &P, const GroupOp &op, i32 a_b, i32 a_e, i32 batch, i32 wv, GBatch<R> &r) {
    const i32 e = a_b + batch * P.wb + lane_id();
    const bool ok = lane_id() < P.wb && e < a_e;
    const i32 ec = ok ? e : a_b;
    const i32 k = op.Gj[ec];
    const int4 *rp = P.rec + ((size_t)k * P.n_win + wv) * 2;
    const int4 r0 = rp[0], r1 = rp[1];
    r.b3 = ok ? r0.w : 0; r.b4 = ok ? r1.x : 0;
#pragma unroll
    for (int m = 0; m < R; ++m) {
        const u64 d = (u64)__double_as_longlong((double)op.Gx[(size_t)ec * R + m] * P.scale);
        r.da_lo[m] = (u32)d; r.da_hi[m] = (u32)(d >> 32);
    }
}
template <int UX>
struct GTile { u32 j[UX], v[UX]; i32 n, l; bool valid; };        // l: the union entry of the batch this tile belongs to (wave-uniform)
struct GCursor { i32 l, q, qe; };
template <int UX, int R>
__device__ __forceinline__ void gtile_fetch(const ExParams &P, const GBatch<R> &r, i32 cnt, GCursor &c, GTile<UX> &t) {
    while (c.q >= c.qe && c.l + 1 < cnt) {
        ++c.l;
        c.q = (__builtin_amdgcn_readlane(r.b3, c.l) + 63) & ~63;
        c.qe = __builtin_amdgcn_readlane(r.b4, c.l);
    }
    t.valid = c.q < c.qe;
    t.n = t.valid ? c.qe - c.q : 0;
    t.l = c.l;
    const i32 base = t.valid ? c.q : 0;
#pragma unroll
    for (int u = 0; u < UX; ++u) {
        const i32 pos = lane_id() + u * HHX_WAVE;
        const i32 qs = pos < t.n ? base + pos : 0;
        t.j[u] = (u32)P.Sc16[qs];
        t.v[u] = __float_as_uint(P.Sx[qs]);
    }
    c.q += UX * HHX_WAVE;
}
// the weights of the tile's union entry are read from the batch registers when the tile is consumed (two v_readlane per member):
// carrying them inside every tile of a group of G tiles in flight cost 2 R G scalar registers and spilled them.
// Member m's accumulators start at slot m * STRIDE with STRIDE a compile-time constant, so that the member offset is the immediate
// offset field of ds_add_u64 (or one add for offsets beyond 64 KB) instead of an index computation per product; the last N_DUMMY
// slots of every member's region are the scratch slots of masked lanes.  Per product: v_fma_f64, the high-word fix, ds_add_u64.
template <int R>
struct GroupStride { static constexpr int value = (((160 * 1024 - 784) / 8 / R) & ~63); };       // 784 = win_fixed_bytes()
template <int UX, int R>
__device__ __forceinline__ void gtile_consume(const ExLds &l, const GBatch<R> &r, const GTile<UX> &t, int probe, u64 &sink) {
    constexpr int STRIDE = GroupStride<R>::value;
    i32 idx[UX];
    double x[UX];
#pragma unroll
    for (int u = 0; u < UX; ++u) {
        idx[u] = lane_id() + u * HHX_WAVE < t.n ? (i32)t.j[u] : STRIDE - N_DUMMY + lane_id();
        x[u] = (double)__uint_as_float(t.v[u]);
    }
#pragma unroll
    for (int m = 0; m < R; ++m) {
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)r.da_lo[m], t.l), hi = (u32)__builtin_amdgcn_readlane((int)r.da_hi[m], t.l);
        if ((lo | hi) == 0u) continue;                            // wave-uniform: this member has no entry in column k
        const double da = __longlong_as_double((long long)(((u64)hi << 32) | lo));
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const u64 g = fx_bits_prod(da, x[u]);
            if (probe & 1) sink += g + (u64)idx[u];              // measurement only (HHX_GROUP_PROBE): the LDS atomics switched off
            else atomicAdd((unsigned long long *)&l.acc[m * STRIDE + idx[u]], (unsigned long long)g);
        }
    }
}
template <int R, int UX, int G>
__global__ __launch_bounds__(EX_T_WIN) void k_expand_group(ExParams P, GroupOp op, i32 cap, i32 wv, int probe) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // member m owns the slots [m * STRIDE, (m + 1) * STRIDE): its window of cap <= STRIDE - N_DUMMY columns, then scratch slots
    constexpr int stride = GroupStride<R>::value;
    const ExLds l = win_carve(smem, R * stride);
    const int tid = threadIdx.x;
    i64 nnzc = 0, n_prod = 0, n_a = 0;
    u64 sink = 0;
    const i32 n_win = P.n_win;
    const i32 c0 = wv * cap, c1 = min(P.n_cols, c0 + cap), wlen = max(0, c1 - c0);
    for (i32 g = blockIdx.x; g < op.n_groups; g += gridDim.x) {
        const i32 a_b = op.Gp[g], a_e = op.Gp[g + 1];
        const i32 n_batches = (a_e - a_b + P.wb - 1) / P.wb;
        if (tid == 0) l.ctr[0] = 0;
        for (i32 t = tid; t < R * stride; t += EX_T) l.acc[t] = 0;
        __syncthreads();
        i32 batch = 0;
        if (lane_id() == 0) batch = atomicAdd(&l.ctr[0], 1);
        batch = __builtin_amdgcn_readfirstlane(batch);
        GBatch<R> nxt;
        if (batch < n_batches) gbatch_load<R>(P, op, a_b, a_e, batch, wv, nxt);
        while (batch < n_batches) {
            const GBatch<R> cur = nxt;
            const i32 cnt = min(P.wb, a_e - (a_b + batch * P.wb));
            i32 nb = 0;
            if (lane_id() == 0) nb = atomicAdd(&l.ctr[0], 1);
            nb = __builtin_amdgcn_readfirstlane(nb);
            if (nb < n_batches) gbatch_load<R>(P, op, a_b, a_e, nb, wv, nxt);
            n_prod += max(0, cur.b4 - ceil64(cur.b3));
            n_a += lane_id() == 0 ? cnt : 0;
            GCursor c = {-1, 0, 0};
            auto fetch = [&](GTile<UX> &t) { gtile_fetch<UX, R>(P, cur, cnt, c, t); };
            auto consume = [&](const GTile<UX> &t) { gtile_consume<UX, R>(l, cur, t, probe, sink); };
            for (;;) {
                GTile<UX> t[G];
                group_fetch<0, G>(t, fetch);
                if (!group_consume<0, G>(t, consume)) break;
            }
            batch = nb;
        }
        __syncthreads();
        for (int m = 0; m < R; ++m) {                              // the epilogue of k_expand_window, once per member over its own accumulators
            const i32 row = op.rows[(size_t)g * R + m];
            if (row < 0) continue;
            ExLds lm = l;
            lm.acc = l.acc + (size_t)m * stride;
            i32 nz;
            const double sw = window_power_sum<false>(P, lm, wlen, &nz, 0.0);
            nnzc += nz;
            const double s_run = (wv == 0 ? 0.0 : P.s_run[row]) + sw;
            window_emit_candidates<false>(P, lm, wlen, c0, 0, s_run, &P.g_win_off[(size_t)row * n_win + wv], &P.g_win_cnt[(size_t)row * n_win + wv]);
            __syncthreads();
            if (tid == 0) P.s_run[row] = s_run;
        }
        __syncthreads();
    }
    nnzc = wave_sum_i64(nnzc);
    if (lane_id() == 0 && nnzc) atomicAdd(&P.cursors[3], (unsigned long long)nnzc);
    n_prod = wave_sum_i64(n_prod);
    if (lane_id() == 0 && n_prod) atomicAdd(&P.cursors[8], (unsigned long long)n_prod);
    n_a = wave_sum_i64(n_a);
    if (lane_id() == 0 && n_a) atomicAdd(&P.cursors[9], (unsigned long long)n_a);
    if (probe && sink == 0x123456789abcdefull) l.acc[0] = sink;
}

// Layout of the window kernel's operand stream, two passes with a scan between them.  One wave per (B row, column
// window) segment.  With link counts (n16 != nullptr: the class stream) the entries are regrouped by count — [count 1]
// [count 2][count 3][other], stable: columns stay ascending inside a class — otherwise the whole segment is "other".
// In the stream every segment starts on a multiple of 64 slots, and so does its explicit part: the 16-bit columns of a
// sub-segment then start on a 128-byte line and its float32 values on a 256-byte boundary, so a sub-segment touches
// ceil(bytes / 128) lines instead of one more (measured before the alignment: 1.22 x the algorithmic bytes on the
// fabric, profiles/r02_pmc_c3.txt).  The padding slots are never read.
//   record (two int4): {b0, b1, b2, b3 | b4, v1, v2, v3}: count-c entries in [b(c-1), b(c)), explicit entries in
//   [ceil64(b3), b4); v_c = float(double(c) / row_sum) is bit for bit what the normalised matrix holds for a count-c entry.
__device__ __forceinline__ int class_of(const unsigned short *__restrict__ n16, i32 q, i32 nc) {
    if (!n16) return 3;
    const i32 c = n16[q];
    return (c >= 1 && c <= nc) ? c - 1 : 3;
}
__global__ __launch_bounds__(256) void k_layout_sizes(i32 n_rows, i32 n_win, i32 cap, i32 nc, const i32 *__restrict__ Bp, const i32 *__restrict__ Bj,
                                                      const unsigned short *__restrict__ n16, int4 *__restrict__ cnt4, i64 *__restrict__ sizes,
                                                      unsigned long long *__restrict__ n_uniform) {
    const int lane = lane_id();
    const i64 total = (i64)n_rows * n_win;
    i64 uni = 0;
    for (i64 sg = (i64)blockIdx.x * 4 + threadIdx.x / HHX_WAVE; sg < total; sg += (i64)gridDim.x * 4) {
        const i32 k = (i32)(sg / n_win), w = (i32)(sg % n_win);
        const i32 rb = Bp[k], re = Bp[k + 1];
        const i32 qb = w == 0 ? rb : lower_bound_i32(Bj, rb, re, w * cap);
        const i32 qe = w == n_win - 1 ? re : lower_bound_i32(Bj, rb, re, (w + 1) * cap);
        i32 cnt[4] = {0, 0, 0, qe - qb};
        if (n16) {
            cnt[3] = 0;
            for (i32 q0 = qb; q0 < qe; q0 += HHX_WAVE) {
                const i32 q = q0 + lane;
                const int cls = q < qe ? class_of(n16, q, nc) : -1;
#pragma unroll
                for (int c = 0; c < 4; ++c) cnt[c] += __popcll(__ballot(cls == c));
            }
        }
        if (lane == 0) {
            cnt4[sg] = make_int4(cnt[0], cnt[1], cnt[2], cnt[3]);
            sizes[sg] = (i64)ceil64(cnt[0] + cnt[1] + cnt[2]) + (i64)ceil64(cnt[3]);
        }
        uni += cnt[0] + cnt[1] + cnt[2];
    }
    if (lane == 0 && uni) atomicAdd(n_uniform, (unsigned long long)uni);
}
// BALANCE: the count-1 sub-segment (consumed by the wide tiles: lane l of a 512-entry unit holds entries 8 l .. 8 l + 7 and
// LDS-atomic instruction j of the unit adds entry 8 l + j of every lane) is stored, unit by unit, in the order of
// (column mod 32) = the pair of LDS banks the 8-byte accumulator of the column occupies.  Instruction j then takes every
// eighth entry of that order: two lanes per bank pair, the conflict-free pattern (6.9 clk per ds_add_u64 against 12.8 on
// random slots, profiles/r02_lds_atomic_bench.jsonl).  The order of the adds is free — exact integer sums.
constexpr int LAY_UNIT = 512;
struct LayoutLds {                  // per wave
    unsigned short stage[LAY_UNIT + HHX_WAVE], sorted[LAY_UNIT];
    u32 hist[32];
    u32 nb[32], start[32], spill_pre[33], hole_pre[33];      // full units: exact dealing (see layout_flush_unit)
};
__device__ __forceinline__ void layout_flush_unit(LayoutLds &L, i32 n, unsigned short *__restrict__ out) {
    const int lane = lane_id();
    if (lane < 32) L.hist[lane] = 0;
    i32 key[LAY_UNIT / HHX_WAVE];
#pragma unroll
    for (int i = 0; i < LAY_UNIT / HHX_WAVE; ++i) {
        const i32 p = lane + i * HHX_WAVE;
        key[i] = p < n ? (i32)L.stage[p] : -1;
        if (key[i] >= 0) atomicAdd(&L.hist[key[i] & 31], 1u);
    }
    // exclusive scan of the 32 counts (lanes 0..31), in place: hist becomes the cursor of every bank pair
    u32 c = lane < 32 ? L.hist[lane] : 0u, incl = c;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const u32 v = __shfl_up(incl, o, HHX_WAVE);
        if (lane >= o) incl += v;
    }
    if (lane < 32) L.hist[lane] = incl - c;
#pragma unroll
    for (int i = 0; i < LAY_UNIT / HHX_WAVE; ++i)
        if (key[i] >= 0) L.sorted[atomicAdd(&L.hist[key[i] & 31], 1u)] = (unsigned short)key[i];
    if (n == LAY_UNIT) {
        // A full unit is DEALT: lane b and lane b + 32 take the first and the second eight entries of bank pair b, so no two
        // lanes of a half-wave share a bank pair; what a bank holds beyond sixteen entries fills the slots the short banks leave.
        const u32 over = c > 16u ? c - 16u : 0u, under = c < 16u ? 16u - c : 0u;
        u32 so = over, su = under;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const u32 v1 = __shfl_up(so, o, HHX_WAVE), v2 = __shfl_up(su, o, HHX_WAVE);
            if (lane >= o) { so += v1; su += v2; }
        }
        if (lane < 32) { L.nb[lane] = c; L.start[lane] = incl - c; L.spill_pre[lane + 1] = so; L.hole_pre[lane + 1] = su; }
        if (lane == 0) { L.spill_pre[0] = 0; L.hole_pre[0] = 0; }
        const u32 b = (u32)lane & 31u, r0 = 8u * ((u32)lane >> 5);
        const u32 nbb = L.nb[b], st = L.start[b], hp = L.hole_pre[b];
#pragma unroll
        for (u32 j = 0; j < 8; ++j) {
            const u32 r = r0 + j;
            u32 src;
            if (r < nbb) src = st + r;
            else {                                           // a hole: the (hp + r - nbb)-th entry of the spill, in bank order
                const u32 h = hp + (r - nbb);
                u32 bb = 0;
                while (L.spill_pre[bb + 1] <= h) ++bb;
                src = L.start[bb] + 16u + (h - L.spill_pre[bb]);
            }
            out[8 * lane + j] = L.sorted[src];
        }
        return;
    }
    // a partial unit: chunk c of the order (8 consecutive entries = one lane's share) goes to lane (c / 2) + ceil(C / 2) * (c % 2):
    // the two chunks of a bank pair land half a wave apart, so the lanes an LDS pass serves together hold different bank pairs
    const i32 full = n >> 3, half = (full + 1) >> 1;
    for (i32 p = lane; p < n; p += HHX_WAVE) {
        const i32 c = p >> 3;
        out[c < full ? 8 * ((c >> 1) + half * (c & 1)) + (p & 7) : p] = L.sorted[p];
    }
}
template <bool BALANCE>
__global__ __launch_bounds__(256) void k_layout_write(i32 n_rows, i32 n_win, i32 cap, i32 nc, const i32 *__restrict__ Bp, const i32 *__restrict__ Bj,
                                                      const float *__restrict__ Bx, const unsigned short *__restrict__ n16,
                                                      const double *__restrict__ row_sum, const int4 *__restrict__ cnt4, const i64 *__restrict__ off,
                                                      unsigned short *__restrict__ oc, float *__restrict__ ox, int4 *__restrict__ rec, int fx) {
    __shared__ LayoutLds s_lay[BALANCE ? 4 : 1];
    LayoutLds &L = s_lay[BALANCE ? threadIdx.x / HHX_WAVE : 0];
    const int lane = lane_id();
    const u64 lt = (1ull << lane) - 1ull;
    const i64 total = (i64)n_rows * n_win;
    for (i64 sg = (i64)blockIdx.x * 4 + threadIdx.x / HHX_WAVE; sg < total; sg += (i64)gridDim.x * 4) {
        const i32 k = (i32)(sg / n_win), w = (i32)(sg % n_win);
        const i32 rb = Bp[k], re = Bp[k + 1];
        const i32 qb = w == 0 ? rb : lower_bound_i32(Bj, rb, re, w * cap);
        const i32 qe = w == n_win - 1 ? re : lower_bound_i32(Bj, rb, re, (w + 1) * cap);
        const int4 c4 = cnt4[sg];
        const i32 s0 = (i32)off[sg] + STREAM_PREFIX;
        i32 base[4] = {s0, s0 + c4.x, s0 + c4.x + c4.y, s0 + ceil64(c4.x + c4.y + c4.z)};
        if (lane == 0) {
            const double s = row_sum ? row_sum[k] : 1.0;
            rec[2 * sg] = make_int4(base[0], base[1], base[2], base[2] + c4.z);
            rec[2 * sg + 1] = make_int4(base[3] + c4.w, __float_as_int((float)(1.0 / s)), __float_as_int((float)(2.0 / s)), __float_as_int((float)(3.0 / s)));
        }
        i32 fill = 0;                                        // count-1 entries waiting in the staging buffer (BALANCE)
        for (i32 q0 = qb; q0 < qe; q0 += HHX_WAVE) {
            const i32 q = q0 + lane;
            int cls = -1;
            i32 col = 0; float x = 0.f;
            // fx: the explicit entry is one word, link count << 16 | window-local column
            if (q < qe) { cls = class_of(n16, q, nc); col = Bj[q]; x = fx ? __uint_as_float((u32)n16[q] << 16 | (u32)(col - w * cap)) : Bx[q]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u64 m = __ballot(cls == c);
                if (BALANCE && c == 0) {                     // through the wave's staging buffer, a unit at a time
                    if (cls == 0) L.stage[fill + __popcll(m & lt)] = (unsigned short)(col - w * cap);
                    fill += __popcll(m);
                    if (fill >= LAY_UNIT) {
                        layout_flush_unit(L, LAY_UNIT, oc + base[0]);
                        base[0] += LAY_UNIT;
                        fill -= LAY_UNIT;
                        if (lane < fill) L.stage[lane] = L.stage[LAY_UNIT + lane];     // the spill-over (< 64 entries) moves to the front
                    }
                    continue;
                }
                if (cls == c) {
                    const i32 o = base[c] + __popcll(m & lt);
                    oc[o] = (unsigned short)(col - w * cap);
                    ox[o] = x;
                }
                base[c] += __popcll(m);
            }
        }
        if (BALANCE && fill) layout_flush_unit(L, fill, oc + base[0]);
        // the slots between the end of the count-1 sub-segment and the next multiple of 8 belong to the last lane that reads it
        if (n16 && nc == 1 && (c4.x & 7) && lane < 8 - (c4.x & 7)) oc[s0 + c4.x + lane] = (unsigned short)(cap + ((c4.x >> 3) & 63));
        // float stream: the slots between the end of the explicit part and the next multiple of 64 hold (a scratch column, 0.0f) — the block tiles
        // (pass_blocks) consume whole 64-entry blocks without a range test: a padding entry adds an exact 0 to a scratch accumulator
        if (!fx && lane < ((-c4.w) & 63)) { oc[base[3] + lane] = (unsigned short)(cap + lane); ox[base[3] + lane] = 0.0f; }
    }
    if (blockIdx.x == 0 && threadIdx.x < HHX_WAVE)           // the stream prefix: eight copies of lane l's scratch column (values: 0 — block 0 of the stream is the block tiles' empty block)
        for (int j = 0; j < 8; ++j) { oc[threadIdx.x * 8 + j] = (unsigned short)(cap + threadIdx.x); if (!fx) ox[threadIdx.x * 8 + j] = 0.0f; }
}

__global__ __launch_bounds__(EX_T_CMP) void k_expand_window_finalize(ExParams P, const i32 *__restrict__ rows, i32 n_list) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ExLds l = ex_carve(smem, 0, 0, 0);
    for (i32 li = blockIdx.x; li < n_list; li += gridDim.x) {
        const i32 row = rows ? rows[li] : li;
        finalize_row(P, l, row, P.n_win, P.s_run[row], &P.g_win_off[(size_t)row * P.n_win], &P.g_win_cnt[(size_t)row * P.n_win]);
        __syncthreads();
    }
}

// The epilogue of k_expand_window alone, fed from a dense float32 row block (hhx_dense: rows of M^2 stored once by the window
// kernel's dense mode): per (row, column window) the same x -> p = x^r, block sum, candidate emission against the running row
// sum, in the same slot order with the same workgroup shape — so that one expansion serves every inflation of the sweep
// (run_mcl_clustering :2155-2158 restarts each inflation from the same pre-expanded matrix) and gives the bits the fused
// single-inflation iteration gives.  k_expand_window_finalize then finishes the rows.
__host__ __device__ inline size_t dense_epi_lds_bytes(i32 cap) { return (size_t)cap * 4 + (size_t)EX_WAVES_MAX * (8 + 4 + 4) + 8 + 8; }
template <bool SQUARE>
__global__ __launch_bounds__(EX_T_WIN, SQUARE ? 8 : 4) void k_dense_epilogue(ExParams P, const DenseSrc S, i32 cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // 4-byte slots (the expanded row is float32 already): two workgroups share a CU, one loading its window while the other reduces
    ExLds l;
    {
        unsigned char *p = smem;
        l.red_d = (double *)p; p += EX_WAVES_MAX * 8;
        l.bcast = (i64 *)p; p += 8;
        l.red_i = (i32 *)p; p += EX_WAVES_MAX * 4;
        l.red_f = (float *)p; p += EX_WAVES_MAX * 4;
        l.ctr = (i32 *)p; p += 8;
        l.acc = (u64 *)p;                                   // [cap] floats
        l.st_da = nullptr; l.st_qb = l.st_qe = nullptr; l.win_off = nullptr; l.win_cnt = nullptr; l.bitmap = l.prefix = nullptr;
    }
    float *slot = (float *)l.acc;
    const int tid = threadIdx.x;
    i64 nnzc = 0;
    for (i32 row = S.row0 + blockIdx.x; row < S.row1; row += gridDim.x) {
        double s_run = 0.0;
        const double div = P.row_div ? P.row_div[row] : 0.0;
        for (i32 wv = 0; wv < P.n_win; ++wv) {
            const i32 c0 = wv * cap, wlen = min(P.n_cols, c0 + cap) - c0;
            const float *src = S.at(row, wv, cap);
            for (i32 t = tid; t < wlen; t += EX_T) slot[t] = src[t];
            __syncthreads();
            i32 nz;
            const double sw = window_power_sum<false, true, SQUARE>(P, l, wlen, &nz, div);
            nnzc += nz;
            s_run = (wv == 0 ? 0.0 : s_run) + sw;
            window_emit_candidates<false, 1>(P, l, wlen, c0, 0, s_run, &P.g_win_off[(size_t)row * P.n_win + wv], &P.g_win_cnt[(size_t)row * P.n_win + wv]);
            __syncthreads();
        }
        if (tid == 0) P.s_run[row] = s_run;
    }
    nnzc = wave_sum_i64(nnzc);
    if (lane_id() == 0 && nnzc) atomicAdd(&P.cursors[3], (unsigned long long)nnzc);
}

// k_dense_epilogue with the window's slots kept conflict-free and the per-slot divisions replaced — the same values, slot for slot:
//   * a thread owns `per` CONSECUTIVE slots (the order of the row sum and of the candidates depends on it), so with an even `per`
//     the 32 lanes of a ds_read group fall on 32 / g banks (g = the power of two in `per`: 4-way at per = 20).  Every 32-slot
//     bank row is therefore stored with its bank index XORed by (row mod g): the g lanes that met on one bank sit in rows an odd
//     number apart, so their keys differ, and they land on the g banks of their own aligned group (the other lanes' banks are g
//     apart).  The fill (consecutive slots per lane) stays a permutation of a row.
//   * float(a / d) with d fixed per row (the row sum d_i, then the running sum S) is taken as float(a * (1 / d)) unless that
//     product lies within 8 double ulps of the midpoint of two floats (or in the float subnormal range), where the true
//     quotient is computed: a * RN(1 / d) is within 2 ulps of a / d, RN(a / d) within half of one, so away from a midpoint
//     both round to the same float.  One multiplication instead of the ~64-cycle fp64 division sequence, 3e10 times at C3.
//   * the survival test of a slot is evaluated once (a bit per owned slot), not in the counting and in the writing pass.
//     (Deciding it by two float comparisons outside a band around thr * S, and reading the slots four at a time, measured
//     4 % slower in the same box: the pass is not bound by its instruction count.)
// Needs per <= DE_PER (the owned slots of the emission pass live in registers): any window the window kernel can hold in LDS.
constexpr int DE_PER = 20;
__device__ __forceinline__ i32 de_swz(i32 t, i32 gm) { return t ^ ((t >> 5) & gm); }
__device__ __forceinline__ float quot_fast(double a, double rd, bool *exact) {      // float(a * rd); *exact: it is float(a / d) for certain
    const double q = a * rd;
    const u32 lo = (u32)(u64)__double_as_longlong(q) & 0x1fffffffu;           // the 29 bits below a float32 mantissa
    *exact = (u32)(lo - 0x0ffffff8u) > 16u && q >= 0x1p-120;
    return (float)q;
}
__device__ __forceinline__ float quot_f32(double a, double d, double rd) {
    bool exact;
    const float q = quot_fast(a, rd, &exact);
    return __builtin_expect(exact, 1) ? q : (float)(a / d);
}
// LDS: the reduction scratch of ExLds plus a second i32 row (the scan's wave totals are written while the argmax's are still read)
__device__ __forceinline__ float de_load(const float *base, i32 idx) {        // uniform base + 32-bit byte offset (no 64-bit address arithmetic per load)
    return *(const float *)((const char *)base + (u32)(idx << 2));
}
__host__ __device__ inline size_t dense_epi_sw_lds_bytes(i32 cap) { return (size_t)((cap + 31) & ~31) * 4 + (size_t)EX_WAVES_MAX * (8 + 4 + 4 + 4) + 8 + 8; }
// PF: one workgroup per CU (128 registers a lane) that loads the window of step k + 1 into registers while step k is reduced — the
// general inflation, whose pow() needs the registers anyway; !PF: two workgroups per CU (64 registers), each loading its window at
// the start of the step — inflation 2 (measured: 28 ms against 32.5 ms with the prefetch and one workgroup).
template <bool SQUARE, bool PF>
__global__ __launch_bounds__(EX_T_WIN, PF ? 4 : 8) void k_dense_epilogue_sw(ExParams P, const DenseSrc S, i32 cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red_d; i64 *bcast; i32 *red_i, *red_s; float *red_f, *slot;
    {
        unsigned char *p = smem;
        red_d = (double *)p; p += EX_WAVES_MAX * 8;
        bcast = (i64 *)p; p += 8;
        red_i = (i32 *)p; p += EX_WAVES_MAX * 4;
        red_f = (float *)p; p += EX_WAVES_MAX * 4;
        red_s = (i32 *)p; p += EX_WAVES_MAX * 4;
        p += 8;
        slot = (float *)p;                                  // [cap rounded up to 32] floats
    }
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / HHX_WAVE;
    i32 nnzc = 0;                                           // at most rows per workgroup x n_cols slots: below 2^31
    // The (row, window) steps of this workgroup in one sequence (every barrier below orders LDS only: a __syncthreads() would wait
    // for the loads of the next window).
    i32 row = S.row0 + (i32)blockIdx.x, wv = 0;
    if (row >= S.row1) return;
    float v[DE_PER];
    double div_n = P.row_div ? P.row_div[row] : 0.0;
    if (PF) {
        const i32 wlen = min(P.n_cols, cap);
        const float *src = S.at(row, 0, cap);
#pragma unroll
        for (int i = 0; i < DE_PER; ++i) v[i] = de_load(src, min(tid + i * EX_T_WIN, wlen - 1));
    }
    double s_run = 0.0, div = 0.0;
    for (;;) {
        const i32 c0 = wv * cap, wlen = min(P.n_cols, c0 + cap) - c0;
        const i32 per = (wlen + EX_T_WIN - 1) / EX_T_WIN, gm = min(32, per & -per) - 1;
        i32 lt = tid;
        asm volatile("" : "+v"(lt));                        // opaque: or the 20 offsets lt + i * 1024 are hoisted out of the loop and spilled
        const i32 s0 = min(wlen, lt * per), s1 = min(wlen, s0 + per);
        if (!PF) {
            const float *src = S.at(row, wv, cap);
#pragma unroll
            for (int i = 0; i < DE_PER; ++i) v[i] = de_load(src, min(lt + i * EX_T_WIN, wlen - 1));      // clamped, not predicated: no branch per load
        }
        if (wv == 0) div = div_n;
        const double rdiv = div != 0.0 ? 1.0 / div : 0.0;          // per step, not held across it: the two registers spilled
#pragma unroll
        for (int i = 0; i < DE_PER; ++i)                    // (t >> 5) & gm does not depend on i: one swizzled base, constant offsets
            if (lt + i * EX_T_WIN < wlen) slot[de_swz(lt, gm) + i * EX_T_WIN] = v[i];
        lds_barrier();
        i32 row_n = row, wv_n = wv + 1;
        if (wv_n == P.n_win) { wv_n = 0; row_n = row + (i32)gridDim.x; }
        const bool more = row_n < S.row1;
        // x -> p = x^r, the window sum and the window maximum: the owned slots in ascending order (window_power_sum<false, true, SQUARE>
        // and the first pass of window_emit_candidates<false, 1>)
        // A slot survives the window if float(p / S) >= thr, S = the running row sum INCLUDING this window — known only after the
        // exchange below.  But S is at least the sum S' of the windows before, so p < thr * S' * (1 - 2^-18) already decides "pruned"
        // (float(p / S) is monotone in p and in 1 / S): the pass over the slots notes the others (a bit per owned slot), and the
        // exact test after the exchange visits those alone — every present slot in window 0, a handful of slots in the later ones.
        const double ts = (double)P.thr * s_run;            // s_run: still the sum of the windows before (unused at wv == 0)
        const float lo_p = (wv > 0 && !P.raw && ts >= 0x1p-100 && ts <= 0x1p100) ? (float)(ts * (1.0 - 0x1p-18)) : 0.0f;
        u32 maybe = 0;
        double s = 0.0;
        i32 nz = 0;
        float bq = -1.0f; i32 bs = 0x7fffffff;
        for (i32 t0 = s0; t0 < s1; t0 += 4) {
            u32 b[4];
            i32 at[4];                                      // the slot's place in LDS: the same for its read and its write
#pragma unroll
            for (int u = 0; u < 4; ++u) { at[u] = de_swz(min(t0 + u, s1 - 1), gm); b[u] = ((const u32 *)slot)[at[u]]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u < s1) {
                    float p = -1.0f;
                    if (b[u] != 0) {
                        const float y = __uint_as_float(b[u]);
                        const float x = div != 0.0 ? quot_f32((double)y, div, rdiv) : y;
                        p = SQUARE ? x * x : (P.raw ? x : ex_inflate(x, P.r, P.square));
                        s += (double)p;
                        ++nz;
                    }
                    if (p > bq) { bq = p; bs = t0 + u; }
                    if (p >= lo_p) maybe |= 1u << (t0 + u - s0);
                    slot[at[u]] = p;
                }
        }
        nnzc += nz;
        if (more && wv_n == 0 && P.row_div) div_n = P.row_div[row_n];
        if (PF && more) {                                   // the next step's window: 20 values in flight through the rest of this step
            const i32 c0n = wv_n * cap, wlen_n = min(P.n_cols, c0n + cap) - c0n;
            const float *src = S.at(row_n, wv_n, cap);
#pragma unroll
            for (int i = 0; i < DE_PER; ++i) v[i] = de_load(src, min(lt + i * EX_T_WIN, wlen_n - 1));
        }
        // block_sum_f64 and block_argmax in one exchange: the same wave trees, the same order over the waves
        s = wave_sum_f64(s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float oq = __shfl_down(bq, o, HHX_WAVE);
            const i32 oc = __shfl_down(bs, o, HHX_WAVE);
            if (oq > bq || (oq == bq && oc < bs)) { bq = oq; bs = oc; }
        }
        if (lane == 0) { red_d[wave] = s; red_f[wave] = bq; red_i[wave] = bs; }
        lds_barrier();
        double sw = red_d[0];
        bq = red_f[0]; bs = red_i[0];
#pragma unroll 3
        for (int k = 1; k < EX_T_WIN / HHX_WAVE; ++k) {       // (a full unroll holds 16 x 4 registers at once)
            sw += red_d[k];
            if (red_f[k] > bq || (red_f[k] == bq && red_i[k] < bs)) { bq = red_f[k]; bs = red_i[k]; }
        }
        s_run = (wv == 0 ? 0.0 : s_run) + sw;
        // survivors against the running sum: one bit per owned slot
        const double rs = 1.0 / s_run;
        u32 keep = (bs >= s0 && bs < s1) ? 1u << (bs - s0) : 0u;          // the window maximum stays, whatever its value
        while (maybe) {
            const int k = __ffs(maybe) - 1;
            maybe &= maybe - 1;
            if (P.raw || quot_f32((double)slot[de_swz(s0 + k, gm)], s_run, rs) >= P.thr) keep |= 1u << k;
        }
        // block_excl_scan_i32 of the survivor counts
        const i32 cnt = __popc(keep);
        i32 incl = cnt;
#pragma unroll
        for (int o = 1; o < HHX_WAVE; o <<= 1) {
            const i32 t = __shfl_up(incl, o, HHX_WAVE);
            if (lane >= o) incl += t;
        }
        if (lane == HHX_WAVE - 1) red_s[wave] = incl;
        lds_barrier();
        i32 off = 0, total = 0;
#pragma unroll 4
        for (int k = 0; k < EX_T_WIN / HHX_WAVE; ++k) { if (k < wave) off += red_s[k]; total += red_s[k]; }
        off += incl - cnt;
        if (tid == 0) {
            i64 base = 0;
            if (total) {
                base = (i64)atomicAdd(&P.cursors[0], (unsigned long long)total);
                if (base + total > P.cand_cap) { atomicExch(&P.cursors[2], 1ull); base = -1; }
            }
            P.g_win_off[(size_t)row * P.n_win + wv] = base;
            P.g_win_cnt[(size_t)row * P.n_win + wv] = base < 0 ? 0 : total;
            *bcast = base;
        }
        lds_barrier();
        const i64 base = *bcast;
        if (base >= 0 && total) {
            i64 o = base + off;
            while (keep) {
                const int k = __ffs(keep) - 1;
                keep &= keep - 1;
                P.cand_col[o] = c0 + s0 + k;
                P.cand_val[o] = slot[de_swz(s0 + k, gm)];
                ++o;
            }
        }
        if (wv == P.n_win - 1 && tid == 0) P.s_run[row] = s_run;
        if (!more) break;
        lds_barrier();                                      // the slots, the reduction rows and the broadcast word are rewritten by the next step
        row = row_n; wv = wv_n;
    }
    const i64 nnz_wave = wave_sum_i64((i64)nnzc);
    if (lane_id() == 0 && nnz_wave) atomicAdd(&P.cursors[3], (unsigned long long)nnz_wave);
}

// ---- the dense epilogue for SEVERAL inflations in one pass (run_mcl_clustering :2155-2158: every inflation restarts from the same
// pre-expanded matrix).  Per entry of M^2 the work of iteration 0 at inflation r is x = float(y / d_i), p = x^r =
// float(exp2(r * log2(double(x)))) (hhx_powr), the row sum, the survivors.  x and log2(x) do not depend on r: they are formed ONCE
// per (row, window) step and kept in registers (20 owned slots a thread: 20 floats + 20 doubles), then the rest of
// k_dense_epilogue_sw's step runs once per inflation — exp2, block sum + argmax, survivors against the running sum, scan, emission
// into that inflation's own candidate pool.  The same operations on the same operands in the same order as the one-inflation
// kernel: the same bits.  The log2 is more than half of hhx_powr and the 40 GB block is read once per group instead of once per
// inflation.
constexpr int MULTI_MAX = 8;
struct MultiOut {                       // one inflation of the group: its parameters and where its results go
    double r; int square, pad;
    i32 *cand_col; float *cand_val; i64 cand_cap;
    unsigned long long *cursors;        // [0] candidate cursor [2] overflow flag
    i64 *g_win_off; i32 *g_win_cnt;     // [n_rows][n_win]
    double *s_run;                      // [n_rows]
};
constexpr int MULTI_LREG = 11;          // of a thread's DE_PER logarithms, this many stay in registers; the others wait in LDS (conflict-free: slot tid + i * 1024)
__host__ __device__ inline size_t dense_epi_multi_lds_bytes(i32 cap) {
    return ((dense_epi_sw_lds_bytes(cap) + MULTI_MAX * 8 + 15) & ~(size_t)15) + (size_t)(DE_PER - MULTI_LREG) * EX_T_WIN * 8;
}
__global__ __launch_bounds__(EX_T_WIN, 4) void k_dense_epilogue_multi(ExParams P, const DenseSrc S, i32 cap, i32 K, const MultiOut *__restrict__ mo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red_d, *s_run_sh; i64 *bcast; i32 *red_i, *red_s; float *red_f, *slot;
    {
        unsigned char *p = smem;
        red_d = (double *)p; p += EX_WAVES_MAX * 8;
        s_run_sh = (double *)p; p += MULTI_MAX * 8;
        bcast = (i64 *)p; p += 8;
        red_i = (i32 *)p; p += EX_WAVES_MAX * 4;
        red_f = (float *)p; p += EX_WAVES_MAX * 4;
        red_s = (i32 *)p; p += EX_WAVES_MAX * 4;
        p += 8;
        slot = (float *)p;                                  // [cap rounded up to 32] floats
    }
    double *l_sh = (double *)(smem + ((dense_epi_sw_lds_bytes(cap) + MULTI_MAX * 8 + 15) & ~(size_t)15));      // [DE_PER - MULTI_LREG][EX_T_WIN]
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / HHX_WAVE;
    i32 nnzc = 0;
    i32 row = S.row0 + (i32)blockIdx.x, wv = 0;
    if (row >= S.row1) return;
    double div = 0.0;
    for (;;) {
        const i32 c0 = wv * cap, wlen = min(P.n_cols, c0 + cap) - c0;
        const i32 per = (wlen + EX_T_WIN - 1) / EX_T_WIN, gm = min(32, per & -per) - 1;
        i32 lt = tid;
        asm volatile("" : "+v"(lt));
        const i32 s0 = min(wlen, lt * per), s1 = min(wlen, s0 + per);
        {
            float v[DE_PER];
            const float *src = S.at(row, wv, cap);
#pragma unroll
            for (int i = 0; i < DE_PER; ++i) v[i] = de_load(src, min(lt + i * EX_T_WIN, wlen - 1));
#pragma unroll
            for (int i = 0; i < DE_PER; ++i)
                if (lt + i * EX_T_WIN < wlen) slot[de_swz(lt, gm) + i * EX_T_WIN] = v[i];
        }
        if (wv == 0) div = P.row_div ? P.row_div[row] : 0.0;
        const double rdiv = div != 0.0 ? 1.0 / div : 0.0;
        lds_barrier();
        // the owned slots, once for all inflations: log2 of x = float(y / d_i); NaN: no entry; x = 0 gives -inf, whose exp2(r * .)
        // is the 0 hhx_powr returns for it
        double Ls[MULTI_LREG];
        i32 nz = 0;
#pragma unroll
        for (int u = 0; u < DE_PER; ++u) {
            double L = __longlong_as_double(0x7ff8000000000000ll);
            if (s0 + u < s1) {
                const u32 b = ((const u32 *)slot)[de_swz(s0 + u, gm)];
                if (b != 0) {
                    const float y = __uint_as_float(b);
                    const float x = div != 0.0 ? quot_f32((double)y, div, rdiv) : y;
                    L = log2((double)x);
                    ++nz;
                }
            }
            if (u < MULTI_LREG) Ls[u] = L; else l_sh[(u - MULTI_LREG) * EX_T_WIN + tid] = L;      // (a thread reads back only what it wrote: no barrier)
            __builtin_amdgcn_sched_barrier(0);
        }
        nnzc += nz;
        i32 row_n = row, wv_n = wv + 1;
        if (wv_n == P.n_win) { wv_n = 0; row_n = row + (i32)gridDim.x; }
        const bool more = row_n < S.row1;
        for (i32 k = 0; k < K; ++k) {
            const double r = mo[k].r;
            const double s_prev = wv == 0 ? 0.0 : s_run_sh[k];
            const double ts = (double)P.thr * s_prev;
            const float lo_p = (wv > 0 && ts >= 0x1p-100 && ts <= 0x1p100) ? (float)(ts * (1.0 - 0x1p-18)) : 0.0f;
            u32 maybe = 0;
            double s = 0.0;
            float bq = -1.0f; i32 bs = 0x7fffffff;
#pragma unroll
            for (int u = 0; u < DE_PER; ++u)
                if (s0 + u < s1) {
                    float p = -1.0f;
                    const double L = u < MULTI_LREG ? Ls[u < MULTI_LREG ? u : 0] : l_sh[(u - MULTI_LREG) * EX_T_WIN + tid];
                    if (L == L) {
                        p = (float)exp2(r * L);                     // hhx_powr with its log2 hoisted
                        s += (double)p;
                    }
                    if (p > bq) { bq = p; bs = s0 + u; }
                    if (p >= lo_p) maybe |= 1u << u;
                    slot[de_swz(s0 + u, gm)] = p;
                    __builtin_amdgcn_sched_barrier(0);          // one exp2 at a time: twenty interleaved ones do not fit the 128 registers next to Ls
                }
            s = wave_sum_f64(s);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float oq = __shfl_down(bq, o, HHX_WAVE);
                const i32 oc = __shfl_down(bs, o, HHX_WAVE);
                if (oq > bq || (oq == bq && oc < bs)) { bq = oq; bs = oc; }
            }
            if (lane == 0) { red_d[wave] = s; red_f[wave] = bq; red_i[wave] = bs; }
            lds_barrier();
            double sw = red_d[0];
            bq = red_f[0]; bs = red_i[0];
#pragma unroll 3
            for (int j = 1; j < EX_T_WIN / HHX_WAVE; ++j) {
                sw += red_d[j];
                if (red_f[j] > bq || (red_f[j] == bq && red_i[j] < bs)) { bq = red_f[j]; bs = red_i[j]; }
            }
            const double s_run = s_prev + sw;
            const double rs = 1.0 / s_run;
            u32 keep = (bs >= s0 && bs < s1) ? 1u << (bs - s0) : 0u;
            while (maybe) {
                const int j = __ffs(maybe) - 1;
                maybe &= maybe - 1;
                if (quot_f32((double)slot[de_swz(s0 + j, gm)], s_run, rs) >= P.thr) keep |= 1u << j;
            }
            const i32 cnt = __popc(keep);
            i32 incl = cnt;
#pragma unroll
            for (int o = 1; o < HHX_WAVE; o <<= 1) {
                const i32 t = __shfl_up(incl, o, HHX_WAVE);
                if (lane >= o) incl += t;
            }
            if (lane == HHX_WAVE - 1) red_s[wave] = incl;
            lds_barrier();
            i32 off = 0, total = 0;
#pragma unroll 4
            for (int j = 0; j < EX_T_WIN / HHX_WAVE; ++j) { if (j < wave) off += red_s[j]; total += red_s[j]; }
            off += incl - cnt;
            if (tid == 0) {
                i64 base = 0;
                if (total) {
                    base = (i64)atomicAdd(&mo[k].cursors[0], (unsigned long long)total);
                    if (base + total > mo[k].cand_cap) { atomicExch(&mo[k].cursors[2], 1ull); base = -1; }
                }
                mo[k].g_win_off[(size_t)row * P.n_win + wv] = base;
                mo[k].g_win_cnt[(size_t)row * P.n_win + wv] = base < 0 ? 0 : total;
                *bcast = base;
                s_run_sh[k] = s_run;                            // read again at the next window (every lane read s_prev before the first barrier above)
                if (wv == P.n_win - 1) mo[k].s_run[row] = s_run;
            }
            lds_barrier();
            const i64 base = *bcast;
            if (base >= 0 && total) {
                i32 *cc = mo[k].cand_col;
                float *cv = mo[k].cand_val;
                i64 o = base + off;
                while (keep) {
                    const int j = __ffs(keep) - 1;
                    keep &= keep - 1;
                    cc[o] = c0 + s0 + j;
                    cv[o] = slot[de_swz(s0 + j, gm)];
                    ++o;
                }
            }
            lds_barrier();                                      // the slots, the reduction rows and the broadcast word are rewritten by the next inflation / step
        }
        if (!more) break;
        row = row_n; wv = wv_n;
    }
    const i64 nnz_wave = wave_sum_i64((i64)nnzc);
    if (lane_id() == 0 && nnz_wave) atomicAdd(&P.cursors[3], (unsigned long long)nnz_wave);
}

__global__ __launch_bounds__(EX_T_CMP) void k_expand_compact(ExParams P, const i32 *__restrict__ rows, i32 n_list, i32 cap, i32 W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ExLds l = ex_carve(smem, cap, W, MAX_WIN);
    const int tid = threadIdx.x;
    i64 nnzc = 0;
    for (i32 li = blockIdx.x; li < n_list; li += gridDim.x) {
        const i32 row = rows[li];
        const i32 a_b = P.Ap[row], a_e = P.Ap[row + 1];
        for (i32 w = tid; w < W; w += EX_T) l.bitmap[w] = 0;
        __syncthreads();
        for (i32 a0 = a_b; a0 < a_e; a0 += STAGE) {
            const i32 len = min(STAGE, a_e - a0);
            stage_chunk(P, l, a0, len);
            __syncthreads();
            mark_compact(P, l, len);
            __syncthreads();
        }
        const i32 nnz_row = bitmap_prefix_total(l, W);
        nnzc += (tid == 0) ? nnz_row : 0;
        double s_run = 0.0;
        i32 n_win = 0;
        for (i32 r0 = 0; r0 < nnz_row; r0 += cap, ++n_win) {      // rank windows (one for almost every row)
            const i32 rlen = min(cap, nnz_row - r0);
            for (i32 t = tid; t < rlen; t += EX_T) l.acc[t] = 0;
            __syncthreads();
            for (i32 a0 = a_b; a0 < a_e; a0 += STAGE) {
                const i32 len = min(STAGE, a_e - a0);
                stage_chunk(P, l, a0, len);
                __syncthreads();
                accumulate_compact(P, l, len, r0, rlen);
                __syncthreads();
            }
            i32 nz;
            s_run += window_power_sum<true>(P, l, rlen, &nz);
            window_emit_candidates<true>(P, l, rlen, 0, r0, s_run, &l.win_off[n_win], &l.win_cnt[n_win]);
            __syncthreads();
        }
        if (nnz_row == 0) {
            if (tid == 0) { P.row_off[row] = 0; P.row_cnt[row] = 0; }
        } else finalize_row(P, l, row, n_win, s_run, l.win_off, l.win_cnt);
        __syncthreads();
    }
    nnzc = wave_sum_i64(nnzc);
    if (lane_id() == 0 && nnzc) atomicAdd(&P.cursors[3], (unsigned long long)nnzc);
}

// ---- hash class: rows of a few thousand distinct output columns ----------------------------------------------------
// From iteration 1 on a row of T^2 holds ~10^3 distinct columns out of n (1158 on average at n = 100k) reached by ~10^5
// products: a dense column window spends its sweeps on empty accumulators (and one launch per window), the compact kernel
// walks the products twice (bitmap mark, then rank lookup).  Here the products are walked ONCE into an LDS hash table
// keyed by column — ds_read + (first touch only) ds_cmpst + ds_add_u64, the same exact fixed-point sums, so the order of
// the adds does not matter — and only the table's few thousand keys go through the bitmap rank that puts the sums in
// column order; from there on the row is finished by the compact kernel's own epilogue (bit-identical by construction).
// A row that turns out to hold more than HASH_LIMIT distinct columns is handed to the window / compact class lists.
constexpr int HASH_T = 512, HASH_C = 4096, HASH_LIMIT = 3072, HASH_STAGE = 256, HASH_PER = HASH_C / HASH_T, HASH_PROBES = 128;
constexpr int HASH_U = 4;           // 16-byte lane loads (two entries each) per B row and step: 4 x 128 = 512 entries
constexpr u32 HASH_EMPTY = 0xffffffffu;
#ifdef HHX_HASH_STATS
__device__ unsigned long long g_hash_stats[4];
#endif
__host__ __device__ inline size_t hash_lds_bytes(i32 W) {
    const size_t tail = (size_t)W * 8 > (size_t)(HASH_T / HHX_WAVE) * HHX_WAVE * 8 ? (size_t)W * 8 : (size_t)(HASH_T / HHX_WAVE) * HHX_WAVE * 8;      // bitmap + prefix; while the products are walked: the waves' leftover queues
    return (size_t)(HASH_C + HHX_WAVE) * 8 + (size_t)HASH_C * 4 + (size_t)HASH_STAGE * (8 + 4 + 4) + (size_t)EX_WAVES_MAX * (8 + 4 + 4) + 8 + (8 + 4) + 16 + tail;
}
// 12 bits = HASH_C slots.  A 24-bit multiply (v_mul_u32_u24: full rate; v_mul_lo_u32 issues at a quarter of it): columns are matrix indices, far below 2^24
__device__ __forceinline__ u32 hash_slot(u32 col) { return __umul24(col, 0x9e3779u) >> 20; }
// Inserting the 2 * HASH_U entries a lane holds of one B row.  A column's home is a two-slot bucket (an even slot and its
// neighbour, one 8-byte LDS read); keys are never removed, so a column sits before the first empty slot of its probe
// sequence and a stale read can only make a lane try a compare-and-swap it loses.
//   fast round   all the bucket reads first (one LDS latency for the batch, not one per product), then a predicated
//                ds_add_u64 wherever the bucket already holds the entry's column: ~95 % of the products of a row whose
//                ~10^3 columns are each hit ~10^2 times, in ~15 branch-free lane-instructions per product;
//   slow loop    what is left (first touches, columns pushed out of their home bucket) is handled per LANE — each lane
//                walks its own pending entries with the general probe / claim sequence, so the wave pays for the lane with
//                the most leftovers (two or three), not for eight wave-wide rounds with a few lanes active in each.
__device__ __forceinline__ bool hash_try(u64 *acc, u32 *keys, i32 *ctr, u32 slot, u32 seen, u32 col, u64 g) {
    if (seen == HASH_EMPTY) {
        seen = atomicCAS(&keys[slot], HASH_EMPTY, col);
        if (seen == HASH_EMPTY) {
            if (atomicAdd(&ctr[0], 1) >= HASH_LIMIT) ctr[1] = 1;          // too many distinct columns: the row leaves the class
            seen = col;
        }
    }
    if (seen != col) return false;
    atomicAdd((unsigned long long *)&acc[slot], (unsigned long long)g);
    return true;
}
// Round 6: the fast round is branch-free — the sum of an entry whose column is not in its home bucket (and of the slots a lane holds beyond the row's
// end) goes to the lane's own scratch accumulator behind the table, like the masked entries of the window kernel, instead of two exec-mask regions per
// product — and the leftovers (first touches, columns pushed out of their home bucket: 3.3 % of the products of iteration 1 at C3, but some in EVERY wave
// step of 512) no longer stop the wave: their positions are appended to a per-wave queue in LDS (a DPP prefix sum of the lanes' counts: no LDS round trip)
// and inserted in bulk, sixty-four at a time with every lane busy, when the queue fills or the wave has finished its staged entries.  The slow insert is a
// chain of dependent LDS round trips (bucket read, compare-and-swap, counter): run per wave step it was ~1.5 such chains per 8 products of a lane.
__device__ __forceinline__ i32 wave_incl_scan_dpp(i32 v) {      // inclusive prefix sum over the 64 lanes, VALU only (row_shr 1/2/4/8, row_bcast 15/31)
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ void hash_insert(u64 *acc, u32 *keys, i32 *ctr, u32 col, u64 g) {
    u32 slot = hash_slot(col) & (HASH_C - 2);
    for (int probes = 0;; ++probes) {
        const uint2 c2 = *reinterpret_cast<const uint2 *>(&keys[slot]);
        if (hash_try(acc, keys, ctr, slot, c2.x, col, g) || hash_try(acc, keys, ctr, slot + 1, c2.y, col, g)) break;
        if (probes >= HASH_PROBES) { ctr[1] = 1; break; }        // a cluster this long means the table is filling up
        slot = (slot + 2) & (HASH_C - 2);
    }
}
// the queued leftovers of this wave: entry = position in Bjx | staged A entry << 32
__device__ __forceinline__ void hash_drain(u64 *acc, u32 *keys, i32 *ctr, const int2 *__restrict__ Bjx, const double *st_da, const u64 *wq, i32 n) {
    for (i32 i = lane_id(); i < n; i += HHX_WAVE) {
        const u64 w = wq[i];
        const int2 e = Bjx[(i32)(u32)w];
        hash_insert(acc, keys, ctr, (u32)e.x, fx_bits_prod(st_da[(i32)(w >> 32)], (double)__int_as_float(e.y)));
    }
}
__device__ __forceinline__ void hash_consume(u64 *acc, u32 *keys, i32 *ctr, const int2 *__restrict__ Bjx, const double *st_da, u64 *wq, i32 wq_cap, i32 &wq_n,
                                             const int4 (&t)[HASH_U], i32 q_first, i32 qb, i32 qe, double da, i32 e_staged) {
    constexpr int K = 2 * HASH_U;
    uint2 cur[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const u32 col = (u32)((k & 1) ? t[k >> 1].z : t[k >> 1].x);
        cur[k] = *reinterpret_cast<const uint2 *>(&keys[hash_slot(col) & (HASH_C - 2)]);
    }
    const u32 scratch = HASH_C + (u32)lane_id();
    u32 pend = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const i32 q = q_first + (k >> 1) * 2 * HHX_WAVE + (k & 1);
        const bool valid = (k & 1) ? q < qe : (q >= qb && q < qe);       // a row starts at any parity: the slot before an odd start is not its entry
        const u32 col = (u32)((k & 1) ? t[k >> 1].z : t[k >> 1].x);
        const bool m0 = cur[k].x == col, m1 = cur[k].y == col;
        const bool hit = valid && (m0 || m1);
        const u64 g = fx_bits_prod(da, (double)__int_as_float((k & 1) ? t[k >> 1].w : t[k >> 1].y));
        atomicAdd((unsigned long long *)&acc[hit ? (hash_slot(col) & (HASH_C - 2)) + (m0 ? 0u : 1u) : scratch], (unsigned long long)g);
        pend |= (valid && !hit) ? 1u << k : 0u;
    }
#ifdef HHX_HASH_STATS                                        // measurement build only: leftovers per wave step (sum over lanes, maximum over lanes)
    {
        const i32 c = __popc(pend);
        i32 mx = c;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_down(mx, o, HHX_WAVE));
        const i32 sm = wave_sum_i32(c);
        if (lane_id() == 0) { atomicAdd(&g_hash_stats[0], 1ull); atomicAdd(&g_hash_stats[1], (unsigned long long)sm); atomicAdd(&g_hash_stats[2], (unsigned long long)mx); }
    }
#endif
    const i32 c = __popc(pend);
    const i32 incl = wave_incl_scan_dpp(c);
    const i32 total = __builtin_amdgcn_readlane(incl, HHX_WAVE - 1);
    if (total == 0) return;                                  // wave-uniform
    if (wq_n + total > wq_cap) {                             // no room: what is queued goes in first; a step with more leftovers than the queue holds
        hash_drain(acc, keys, ctr, Bjx, st_da, wq, wq_n);    // (the first steps of a row, when the table is empty) is inserted lane by lane as before
        wq_n = 0;
        if (total > wq_cap) {
            while (pend) {
                const int k = __ffs((int)pend) - 1;
                pend &= pend - 1;
                const int2 e = Bjx[q_first + (k >> 1) * 2 * HHX_WAVE + (k & 1)];
                hash_insert(acc, keys, ctr, (u32)e.x, fx_bits_prod(da, (double)__int_as_float(e.y)));
            }
            return;
        }
    }
    i32 at = wq_n + incl - c;
    while (pend) {                                           // writes only: nothing to wait for
        const int k = __ffs((int)pend) - 1;
        pend &= pend - 1;
        wq[at++] = (u64)(u32)(q_first + (k >> 1) * 2 * HHX_WAVE + (k & 1)) | ((u64)(u32)e_staged << 32);
    }
    wq_n += total;
}
__global__ __launch_bounds__(256) void k_pack_jx(i64 n, const i32 *__restrict__ j, const float *__restrict__ x, int2 *__restrict__ out) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) out[k] = make_int2(j[k], __float_as_int(x[k]));
}
__global__ __launch_bounds__(HASH_T, 4) void k_expand_hash(ExParams P, const i32 *__restrict__ rows, i32 n_list, i32 W, const i64 *__restrict__ row_f,
                                                        i64 window_min, i32 *__restrict__ list_window, i32 *__restrict__ list_compact,
                                                        unsigned int *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ExLds l;
    unsigned char *p = smem;
    l.acc = (u64 *)p; p += (size_t)(HASH_C + HHX_WAVE) * 8;        // + one scratch accumulator per lane (hash_consume)
    l.st_da = (double *)p; p += HASH_STAGE * 8;
    l.red_d = (double *)p; p += EX_WAVES_MAX * 8;
    l.bcast = (i64 *)p; p += 8;
    l.win_off = (i64 *)p; p += 8;
    u32 *keys = (u32 *)p; p += (size_t)HASH_C * 4;
    l.st_qb = (i32 *)p; p += HASH_STAGE * 4;
    l.st_qe = (i32 *)p; p += HASH_STAGE * 4;
    l.red_i = (i32 *)p; p += EX_WAVES_MAX * 4;
    l.red_f = (float *)p; p += EX_WAVES_MAX * 4;
    l.win_cnt = (i32 *)p; p += 4;
    l.ctr = (i32 *)p; p += 12;                                // [0] distinct columns so far, [1] the row does not fit
    l.bitmap = (u32 *)p; p += (size_t)W * 4;
    l.prefix = (u32 *)p;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / HHX_WAVE;
    // per-wave leftover queue in the bitmap + prefix region (idle until the table is read out): 8-byte entries, a multiple of 64 per wave
    const i32 region = max(2 * W, (HASH_T / HHX_WAVE) * HHX_WAVE * 2);             // in 4-byte words
    const i32 wq_cap = ((region / 2) / (HASH_T / HHX_WAVE)) & ~(HHX_WAVE - 1);
    u64 *const wq = reinterpret_cast<u64 *>(l.bitmap) + (size_t)wave * wq_cap;
    i64 nnzc = 0;
    for (i32 li = blockIdx.x; li < n_list; li += gridDim.x) {
        const i32 row = rows[li];
        const i32 a_b = P.Ap[row], a_e = P.Ap[row + 1];
        for (i32 t = tid; t < HASH_C; t += HASH_T) { keys[t] = HASH_EMPTY; l.acc[t] = 0; }
        if (tid == 0) { l.ctr[0] = 0; l.ctr[1] = 0; }
        __syncthreads();
        bool fits = true;
        i32 wq_n = 0;                                        // this wave's queued leftovers (wave-uniform)
        for (i32 a0 = a_b; a0 < a_e && fits; a0 += HASH_STAGE) {
            const i32 len = min(HASH_STAGE, a_e - a0);
            stage_chunk(P, l, a0, len);
            __syncthreads();
            // A wave takes the staged entries two at a time.  A lane loads two consecutive entries of B per 16-byte load
            // (column, value, column, value), HASH_U loads per B row and step = 512 entries, a whole row of the later
            // iterations; the loads of both rows are issued back to back, then consumed — nothing loaded is live across a
            // loop back-edge.  Rows start at any parity: the first load is aligned down and the entry before the row masked.
            for (i32 e = 2 * wave; e < len; e += 2 * (HASH_T / HHX_WAVE)) {
                const bool two = e + 1 < len;
                const double da0 = l.st_da[e], da1 = two ? l.st_da[e + 1] : 0.0;
                const i32 qb0 = l.st_qb[e], qe0 = l.st_qe[e], qb1 = two ? l.st_qb[e + 1] : 0, qe1 = two ? l.st_qe[e + 1] : 0;
                const i32 s0 = qb0 & ~1, s1 = qb1 & ~1;
                const i32 span = max(qe0 - s0, qe1 - s1);
                for (i32 off = 0; off < span; off += HASH_U * 2 * HHX_WAVE) {
                    int4 t0[HASH_U], t1[HASH_U];
                    // unconditional loads: a lane beyond its row's end reads the (even) slot before the end instead — hash_consume masks by position
#pragma unroll
                    for (int u = 0; u < HASH_U; ++u) {
                        const i32 q = s0 + off + u * 2 * HHX_WAVE + 2 * lane;
                        t0[u] = *reinterpret_cast<const int4 *>(P.Bjx + (q < qe0 ? q : s0));
                    }
#pragma unroll
                    for (int u = 0; u < HASH_U; ++u) {
                        const i32 q = s1 + off + u * 2 * HHX_WAVE + 2 * lane;
                        t1[u] = *reinterpret_cast<const int4 *>(P.Bjx + (q < qe1 ? q : s1));
                    }
                    if (__hip_atomic_load(&l.ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    hash_consume(l.acc, keys, l.ctr, P.Bjx, l.st_da, wq, wq_cap, wq_n, t0, s0 + off + 2 * lane, qb0, qe0, da0, e);
                    hash_consume(l.acc, keys, l.ctr, P.Bjx, l.st_da, wq, wq_cap, wq_n, t1, s1 + off + 2 * lane, qb1, qe1, da1, e + 1);
                }
                if (__hip_atomic_load(&l.ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            }
            hash_drain(l.acc, keys, l.ctr, P.Bjx, l.st_da, wq, wq_n);      // before the staged entries they refer to are replaced
            wq_n = 0;
            __syncthreads();
            fits = l.ctr[1] == 0;
            __syncthreads();
        }
        if (!fits) {                                         // uniform: the row goes to the class its product count names
            if (tid == 0) {
                const i64 f = row_f[row];
                if (f >= window_min) {
                    list_window[atomicAdd(&counts[0], 1u)] = row;
                    atomicAdd(&P.cursors[5], (unsigned long long)f);
                    atomicAdd(&P.cursors[6], (unsigned long long)(a_e - a_b));
                } else list_compact[atomicAdd(&counts[1], 1u)] = row;
            }
            continue;
        }
        // the table's keys -> bitmap; (key, sum) pairs wait in registers while the accumulator array changes its meaning
        for (i32 w = tid; w < W; w += HASH_T) l.bitmap[w] = 0;      // (the region held the leftover queues until here)
        __syncthreads();
        u32 kk[HASH_PER];
        u64 aa[HASH_PER];
#pragma unroll
        for (int u = 0; u < HASH_PER; ++u) {
            const i32 t = tid + u * HASH_T;
            kk[u] = keys[t];
            aa[u] = l.acc[t];
            if (kk[u] != HASH_EMPTY) atomicOr(&l.bitmap[kk[u] >> 5], 1u << (kk[u] & 31));
        }
        __syncthreads();
        const i32 nnz_row = bitmap_prefix_total(l, W);       // ends with a barrier
#pragma unroll
        for (int u = 0; u < HASH_PER; ++u)
            if (kk[u] != HASH_EMPTY) l.acc[rank_of(l, (i32)kk[u])] = aa[u];
        __syncthreads();
        nnzc += (tid == 0) ? nnz_row : 0;
        if (nnz_row == 0) {
            if (tid == 0) { P.row_off[row] = 0; P.row_cnt[row] = 0; }
        } else {
            i32 nz;
            const double s_run = window_power_sum<true>(P, l, nnz_row, &nz);
            window_emit_candidates<true>(P, l, nnz_row, 0, 0, s_run, &l.win_off[0], &l.win_cnt[0]);
            __syncthreads();
            finalize_row(P, l, row, 1, s_run, l.win_off, l.win_cnt);
        }
        __syncthreads();
    }
    nnzc = wave_sum_i64(nnzc);
    if (lane_id() == 0 && nnzc) atomicAdd(&P.cursors[3], (unsigned long long)nnzc);
}

// The symmetric pre-expansion computes the blocks (I, J >= I) of Y = float(S) only; block (J, I), J > I, is the transpose of (I, J)
// bit for bit (S is an exact integer matrix).  One workgroup per 64 x 64 tile of the strictly lower block triangle: the source
// tile is read row-wise (coalesced), turned in LDS and written row-wise.  cap is a multiple of 64, so a tile never straddles blocks.
// grid: (cap / 64, cap / 64, block pairs J > I) — only the tiles that are copied are launched.
__global__ __launch_bounds__(256) void k_transpose_lower(float *__restrict__ Y, i64 ld, i32 n, i32 cap) {
    __shared__ float tile[64][65];
    i32 bj = 1, bi = (i32)blockIdx.z;                               // pair index z = J (J - 1) / 2 + I, I < J
    while (bi >= bj) { bi -= bj; ++bj; }
    const i32 r0 = bj * cap + blockIdx.y * 64, c0 = bi * cap + blockIdx.x * 64;          // destination tile: rows r0.., columns c0..
    if (r0 >= n) return;                                            // the last block of rows is short
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) {                              // source tile: rows c0.., columns r0..
        const i32 sr = c0 + k, sc = r0 + tx;
        tile[k][tx] = (sr < n && sc < n) ? Y[(size_t)sr * (size_t)ld + sc] : 0.0f;
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        const i32 dr = r0 + k, dc = c0 + tx;
        if (dr < n && dc < n) Y[(size_t)dr * (size_t)ld + dc] = tile[tx][k];
    }
}

// Upper-block-triangle storage: before the rows of block row I are finished, the blocks (J, I), J < I — held as rows of block row J —
// are turned into rows of block row I, side by side in a scratch block (cap rows x I cap floats): dst[r][J cap + c] = src_J[c][r].
// grid: (cap / 64 tiles of destination columns, tiles of destination rows, J).  src block (J, I) starts at
// tri + tri_row_off(J) + (I - J) cap with pitch ldn - J cap; rows_I = rows of block row I (the last one may be short).
__global__ __launch_bounds__(256) void k_transpose_tri(const float *__restrict__ tri, float *__restrict__ dst, i64 dst_ld, i32 I, i32 rows_I, i32 cap, i64 ldn) {
    __shared__ float tile[64][65];
    const i32 J = (i32)blockIdx.z;
    const float *src = tri + tri_row_off(J, cap, ldn) + (size_t)(I - J) * cap;
    const i64 src_ld = ldn - (i64)J * cap;
    const i32 r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;             // destination tile: rows r0.. (of block row I), columns J cap + c0..
    if (r0 >= rows_I) return;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4)                                  // source tile: rows c0 + k of block row J, columns r0 + tx of window I
        tile[k][tx] = (r0 + tx < rows_I) ? src[(size_t)(c0 + k) * (size_t)src_ld + r0 + tx] : 0.0f;
    __syncthreads();
    for (int k = ty; k < 64; k += 4)
        if (r0 + k < rows_I) dst[(size_t)(r0 + k) * (size_t)dst_ld + (size_t)J * cap + c0 + tx] = tile[tx][k];
}

// ---- classification: product count per row, three row lists -----------------------------------------
// A block takes 64 consecutive rows at a time: its waves count the products of one row each (coalesced),
// then the first wave sorts the 64 rows into the class lists with ONE atomic per class and chunk (a
// per-row atomicAdd on a single counter costs ~11 ns each: 1.1 ms for 100k rows, per iteration).
constexpr int TINY_MAX = 32;        // rows with at most this many products go to the thread-per-row kernel
__global__ __launch_bounds__(256) void k_classify(i32 n_rows, const i32 *__restrict__ Ap, const i32 *__restrict__ Aj,
                                                  const i32 *__restrict__ Bp, i64 window_min_products, i64 hash_max_products, i64 tiny_max,
                                                  i32 *__restrict__ list_window, i32 *__restrict__ list_compact,
                                                  i32 *__restrict__ list_tiny, i32 *__restrict__ list_hash, i64 *__restrict__ row_f,
                                                  unsigned int *__restrict__ counts, unsigned long long *__restrict__ cursors) {
    __shared__ i64 fs[64];
    const int lane = lane_id(), wave = threadIdx.x / HHX_WAVE;
    i64 total = 0, total_w = 0, entries_w = 0;
    for (i32 chunk = blockIdx.x * 64; chunk < n_rows; chunk += gridDim.x * 64) {
        for (i32 r = wave; r < 64; r += 4) {
            const i32 row = chunk + r;
            i64 f = -1;
            if (row < n_rows) {
                f = 0;
                for (i32 p = Ap[row] + lane; p < Ap[row + 1]; p += HHX_WAVE) { const i32 k = Aj[p]; f += Bp[k + 1] - Bp[k]; }
                f = wave_sum_i64(f);
            }
            if (lane == 0) fs[r] = f;
        }
        __syncthreads();
        if (wave == 0) {
            const i64 f = fs[lane];
            const i32 row = chunk + lane;
            // hash class (3): everything between the tiny rows and hash_max_products, whatever window_min says — the hash kernel
            // hands back the rows with too many distinct columns (hash_max_products = 0 switches the class off)
            // (dense mode: tiny_max = hash_max = -1, window_min = 0 — every row, empty ones included, takes the window class)
            const int cls = f < 0 ? -1 : (f <= tiny_max ? 2 : (f <= hash_max_products ? 3 : (f >= window_min_products ? 0 : 1)));
            if (row < n_rows) row_f[row] = f;
            i32 *const lists[4] = {list_window, list_compact, list_tiny, list_hash};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u64 mask = __ballot(cls == c);
                if (mask) {
                    unsigned int base = 0;
                    const int leader = __ffsll((unsigned long long)mask) - 1;
                    if (lane == leader) base = atomicAdd(&counts[c], (unsigned int)__popcll(mask));
                    base = __shfl(base, leader, HHX_WAVE);
                    if (cls == c) lists[c][base + __popcll(mask & ((1ull << lane) - 1ull))] = row;
                }
            }
            if (f > 0) total += f;
            if (cls == 0) { total_w += f; entries_w += Ap[row + 1] - Ap[row]; }
        }
        __syncthreads();
    }
    if (wave == 0) {
        total = wave_sum_i64(total); total_w = wave_sum_i64(total_w); entries_w = wave_sum_i64(entries_w);
        if (lane == 0 && total) atomicAdd(&cursors[4], (unsigned long long)total);
        if (lane == 0 && total_w) { atomicAdd(&cursors[5], (unsigned long long)total_w); atomicAdd(&cursors[6], (unsigned long long)entries_w); }
    }
}

// products per row of a * b (the cost of a row of the expansion): one wave per row
__global__ __launch_bounds__(256) void k_row_products(i32 n_rows, const i32 *__restrict__ Ap, const i32 *__restrict__ Aj,
                                                      const i32 *__restrict__ Bp, i64 *__restrict__ out) {
    const int lane = lane_id();
    for (i32 row = blockIdx.x * 4 + threadIdx.x / HHX_WAVE; row < n_rows; row += gridDim.x * 4) {
        i64 f = 0;
        for (i32 p = Ap[row] + lane; p < Ap[row + 1]; p += HHX_WAVE) { const i32 k = Aj[p]; f += Bp[k + 1] - Bp[k]; }
        f = wave_sum_i64(f);
        if (lane == 0) out[row] = f;
    }
}

// ---- tiny rows (<= TINY_MAX products): one THREAD per row ------------------------------------------------
// After a few iterations almost every row of T has one to three entries; a workgroup per row then spends
// its time clearing an n-bit bitmap.  Here a thread merges the row's products into a sorted local list
// (exact grid-rounded double adds, same value as acc_add) and applies inflate / normalise / prune /
// normalise sequentially, i.e. in the reference's own summation order (:2037-2042, :1987-2014).
__global__ __launch_bounds__(256) void k_expand_tiny(ExParams P, const i32 *__restrict__ rows, i32 n_list) {
    const int lane = lane_id();
    i64 nnzc = 0;
    for (i32 base_i = blockIdx.x * blockDim.x; base_i < n_list; base_i += gridDim.x * blockDim.x) {
        const i32 li = base_i + threadIdx.x;
        i32 cols[TINY_MAX];
        double vals[TINY_MAX];
        i32 cnt = 0, row = -1;
        if (li < n_list) {
            row = rows[li];
            for (i32 p = P.Ap[row]; p < P.Ap[row + 1]; ++p) {
                const i32 k = P.Aj[p];
                const double da = (double)P.Ax[p] * P.scale;
                for (i32 q = P.Bp[k]; q < P.Bp[k + 1]; ++q) {
                    const i32 c = P.Bj[q];
                    const double g = (da * (double)P.Bx[q] + 1.0) - 1.0;
                    i32 pos = 0;
                    while (pos < cnt && cols[pos] < c) ++pos;
                    if (pos < cnt && cols[pos] == c) vals[pos] += g;
                    else {
                        for (i32 t = cnt; t > pos; --t) { cols[t] = cols[t - 1]; vals[t] = vals[t - 1]; }
                        cols[pos] = c; vals[pos] = g;
                        ++cnt;
                    }
                }
            }
        }
        // inflate + first normalisation, first maximum, survivors
        float pw[TINY_MAX];
        double s1 = 0.0;
        for (i32 t = 0; t < cnt; ++t) {
            const float x = (float)(vals[t] * P.inv_scale);
            pw[t] = P.raw ? x : ex_inflate(x, P.r, P.square);
            s1 += fabs((double)pw[t]);
        }
        if (P.raw) s1 = 0.0;
        i32 am = -1, keep = 0;
        float best = 0.f;
        double s2 = 0.0;
        for (i32 t = 0; t < cnt; ++t) {
            if (s1 != 0.0) pw[t] = (float)((double)pw[t] / s1);
            if (am < 0 || pw[t] > best) { am = t; best = pw[t]; }
        }
        for (i32 t = 0; t < cnt; ++t)
            if (P.raw || pw[t] >= P.thr || t == am) { ++keep; s2 += fabs((double)pw[t]); }
        nnzc += cnt;
        // one atomic per wave reserves the output rows of its 64 rows
        i32 incl = keep;
#pragma unroll
        for (int o = 1; o < HHX_WAVE; o <<= 1) {
            const i32 v = __shfl_up(incl, o, HHX_WAVE);
            if (lane >= o) incl += v;
        }
        const i32 wave_total = __shfl(incl, HHX_WAVE - 1, HHX_WAVE);
        unsigned long long wbase = 0;
        if (lane == HHX_WAVE - 1 && wave_total) wbase = atomicAdd(&P.cursors[1], (unsigned long long)wave_total);
        wbase = (unsigned long long)__shfl((long long)wbase, HHX_WAVE - 1, HHX_WAVE);
        if (row >= 0) {
            i64 o = (i64)wbase + incl - keep;
            if ((i64)wbase + wave_total > P.out_cap) { atomicExch(&P.cursors[2], 1ull); P.row_off[row] = 0; P.row_cnt[row] = 0; }
            else {
                P.row_off[row] = o;
                P.row_cnt[row] = keep;
                for (i32 t = 0; t < cnt; ++t)
                    if (P.raw || pw[t] >= P.thr || t == am) {
                        P.out_col[o] = cols[t];
                        P.out_val[o] = (s2 != 0.0 && !P.raw) ? (float)((double)pw[t] / s2) : pw[t];
                        ++o;
                    }
            }
        }
    }
    nnzc = wave_sum_i64(nnzc);
    if (lane == 0 && nnzc) atomicAdd(&P.cursors[3], (unsigned long long)nnzc);
}

// ---- pack the bump-allocated rows into CSR order ---------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_rows(i32 n_rows, const i64 *__restrict__ row_off, const i32 *__restrict__ indptr,
                                                   const i32 *__restrict__ pool_col, const float *__restrict__ pool_val,
                                                   i32 *__restrict__ out_j, float *__restrict__ out_x) {
    const int lane = lane_id();
    for (i32 row = blockIdx.x * 4 + threadIdx.x / HHX_WAVE; row < n_rows; row += gridDim.x * 4) {
        const i32 b = indptr[row], e = indptr[row + 1];
        const i64 src = row_off[row];
        for (i32 t = lane; t < e - b; t += HHX_WAVE) { out_j[b + t] = pool_col[src + t]; out_x[b + t] = pool_val[src + t]; }
    }
}

}  // namespace

// Optional description of the right operand.  n16 / row_sum: b is the L1-normalised link matrix, entry p of row k
// equals float(n16[p] / row_sum[k]) (checked by the caller): the window class then streams b regrouped by link
// count (the class stream above).  b itself still holds the float32 values (used by the light-row kernels).
struct CodedOperand {
    const unsigned short *n16 = nullptr;
    const double *row_sum = nullptr;
    int raw = 0;                        // plain product (no inflation / pruning): hhx_spgemm's fast path
    float *dense_out = nullptr;         // dense mode: n_rows x n_cols float32 block that receives the expanded rows; no CSR result
    i64 dense_ld = 0;                   // ... and its row pitch in floats (>= n_cols)
    i32 *plan_out = nullptr;            // dense mode: [cap_win, n_win] of the column-window plan, for k_dense_epilogue
    // integer arithmetic (ExParams::W): a = rows [a_row0, a_row0 + a->n_rows) of the link matrix whose normalised form is b
    const u64 *W = nullptr;
    const unsigned short *a16 = nullptr;     // link counts of a's entries
    const double *a_row_sum = nullptr;       // d_i of a's rows
    int shift = 0;
    int sym = 0;                        // dense mode: compute the blocks J >= I only; a == all rows: mirrored here, a row block: by the caller
    i32 sym_row0 = 0;
    int tri = 0;                        // sym over all rows, dense_out = the upper block triangle alone (tri_floats(n_win, cap_win) floats): nothing is mirrored
};

// the column-window plan of the window class for an operand of n_cols columns / nnz_b entries: the fewest, widest windows whose
// 8-byte accumulators fit LDS (hhx_expand_impl; hhx_expand_dense_impl sizes the triangle with it before the expansion runs)
static void window_plan(i32 n_cols, i64 nnz_b, i32 *cap_win_out, i32 *n_win_out) {
    const size_t fixed_win = win_fixed_bytes();
    const i64 slice_mb = tune_get("cache_slice_mb", 0);
    const i64 slice_bytes = slice_mb > 0 ? slice_mb << 20 : (i64)1 << 60;
    const i32 cap_max = (i32)((160 * 1024 - fixed_win) / 8) & ~63;
    i64 n_win64 = ((i64)n_cols + cap_max - 1) / cap_max;
    const i64 by_cache = (nnz_b * 8 + slice_bytes - 1) / slice_bytes;
    const i64 widest = std::max<i64>(1, (i64)n_cols / 2048);           // never narrower than 2048 columns
    n_win64 = std::max(n_win64, std::min(by_cache, widest));
    i32 cap_win = (i32)((((i64)n_cols + n_win64 - 1) / n_win64 + 63) & ~63);
    if (cap_win > cap_max) cap_win = cap_max;
    *cap_win_out = cap_win;
    *n_win_out = (n_cols + cap_win - 1) / cap_win;
}

template <int PROBE, int UX, int RX, int RW, bool FX, int T = EX_T_WIN, bool BLK = false>
static int launch_window_fx(const ExParams &P, const i32 *rows, i32 n_list, i32 cap, size_t lds, unsigned grid) {
    static int attr_dev = -1;           // the attribute is per device: keyed on the current ordinal
    int dev = 0;
    HHX_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_expand_window<PROBE, UX, RX, RW, FX, T, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_dev = dev;
    }
    for (i32 wv = 0; wv < P.n_win; ++wv) {
        // symmetric mode: the rows are in identity order and launch wv takes the row blocks I <= wv (blocks J >= I of S)
        const i32 n_w = P.sym ? (i32)std::max<i64>(0, std::min<i64>(n_list, (i64)(wv + 1) * cap - P.sym_row0)) : n_list;
        if (n_w == 0) continue;
        k_expand_window<PROBE, UX, RX, RW, FX, T, BLK><<<std::min<unsigned>(grid, (unsigned)std::max(n_w, 1)), T, lds, g_stream>>>(P, P.sym ? nullptr : rows, n_w, cap, wv);
    }
    return 0;
}
template <int PROBE, int UX, int RX, int RW>
static int launch_window(const ExParams &P, const i32 *rows, i32 n_list, i32 cap, size_t lds, unsigned grid) {
    if (P.W) return launch_window_fx<PROBE, UX, RX, RW, true>(P, rows, n_list, cap, lds, grid);
    return launch_window_fx<PROBE, UX, RX, RW, false>(P, rows, n_list, cap, lds, grid);
}

template <int R, int UX, int G>
static int launch_group(const ExParams &P, const GroupOp &op, i32 cap, size_t lds, unsigned grid) {
    static int attr_dev = -1;
    int dev = 0;
    HHX_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_expand_group<R, UX, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_dev = dev;
    }
    static const int probe = getenv("HHX_GROUP_PROBE") ? (atoi(getenv("HHX_GROUP_PROBE")) & 1) : 0;     // 1, measurement only: the LDS atomics off, garbage results
    for (i32 wv = 0; wv < P.n_win; ++wv) k_expand_group<R, UX, G><<<grid, EX_T_WIN, lds, g_stream>>>(P, op, cap, wv, probe);
    return 0;
}

// the bump-allocated rows -> CSR: scan of the row counts, ordered copy
static int pack_rows_to_csr(i32 n_rows, i32 n_cols, const i32 *row_cnt, i32 *indptr, const i64 *row_off, const i32 *pool_col, const float *pool_val,
                            hhx_csr **out) {
    i64 total = 0;
    HHX_TRY(exclusive_scan_i32(row_cnt, indptr, n_rows, &total));
    hhx_csr *p = nullptr;
    HHX_TRY(hhx_csr_alloc_internal(n_rows, n_cols, total, &p));
    HHX_HIP(hipMemcpyAsync(p->indptr.p, indptr, sizeof(i32) * ((size_t)n_rows + 1), hipMemcpyDeviceToDevice, g_stream));
    k_pack_rows<<<(unsigned)std::max<i64>(1, std::min<i64>(((i64)n_rows + 3) / 4, 8192)), 256, 0, g_stream>>>(
        n_rows, row_off, p->indptr.p, pool_col, pool_val, p->indices.p, p->data.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);       // the caller's pools are released on return
    if (e != hipSuccess) { hhx_csr_free(p); return fail("expand pack: %s", hipGetErrorString(e)); }
    *out = p;
    return 0;
}

int hhx_expand_impl(const hhx_csr *a, const hhx_csr *b, const CodedOperand &coded, int fx_shift, double inflation, double pruning,
                    hhx_csr **out, i64 *n_products, i64 *nnz_expanded) {
    const bool dense = coded.dense_out != nullptr;
    if (!a || !b || (!out && !dense)) return fail("null pointer");
    if (a->n_cols != b->n_rows) return fail("expand shape mismatch");
    if (!(inflation > 0)) return fail("inflation must be positive");
    if (b->nnz > (i64)INT32_MAX - 4096) return fail("expand: right operand has %lld entries; the tile cursors need 4096 below 2^31", (long long)b->nnz);
    if (fx_shift < 0 || fx_shift > 52) fx_shift = 52;        // the exact-double accumulation holds 52 fractional bits
    const auto t_enter = std::chrono::steady_clock::now();
    const i32 n_rows = a->n_rows, n_cols = b->n_cols;
    const i32 W = (n_cols + 31) / 32;
    // ---- plans
    const bool use_cls = coded.n16 != nullptr && !coded.raw && tune_get("cls", 1) != 0;
    const bool fx = use_cls && coded.W != nullptr;           // integer arithmetic of the link matrix: every row through the window class
    if (coded.W && !fx) return fail("expand: the integer arithmetic needs the class stream");
    if (coded.sym && !(dense && fx)) return fail("expand: the symmetric mode needs the dense integer mode");
    const bool sym_whole = coded.sym && a->n_rows == b->n_rows;          // all rows here: the mirror image is written here too
    // link counts 1..n_classes are streamed as columns only.  Measured at n = 100k (profiles/r02_expand_probe_c3.jsonl): counts 1-3
    // move 17 % fewer bytes than count 1 alone but run 20 % longer (the count-2 / count-3 sub-segments are a few dozen entries:
    // tiles of 128 that are mostly empty), so the default is 1
    const i32 n_classes = fx ? 1 : (i32)std::min<i64>(3, std::max<i64>(1, tune_get("cls_nc", 1)));
    const size_t fixed_win = win_fixed_bytes(), fixed_cmp = ex_fixed_bytes(W, MAX_WIN);
    // window class: the column window must fit LDS (8 B per column).  Measured on MI355X (n = 100k, 330M
    // entries): the fewest, widest windows win — 6 windows 1.41 s, 13 windows 1.54 s, 27 windows 2.24 s —
    // because the per-segment cost grows faster than the Infinity Cache hit rate of a narrower column slice
    // B[:, w].  tune "cache_slice_mb" (MB of B per slice) forces more windows for experiments.
    i32 cap_win = 0, n_win = 0;
    window_plan(n_cols, b->nnz, &cap_win, &n_win);
    if (coded.tri && !(coded.sym && a->n_rows == b->n_rows && coded.sym_row0 == 0)) return fail("expand: the triangle storage needs the symmetric mode over all rows");
    size_t budget_cmp = 64 * 1024;
    if (fixed_cmp + 2048 * 8 > budget_cmp) budget_cmp = 160 * 1024;
    if (fixed_cmp + 1024 * 8 > budget_cmp) return fail("expand: %d columns exceed the LDS bitmap capacity", n_cols);
    const i32 cap_cmp = (i32)((budget_cmp - fixed_cmp) / 8) & ~63;
    if (n_cols / 2 / cap_cmp + 2 > MAX_WIN - 1) return fail("expand: %d columns need too many LDS rank windows", n_cols);
    const size_t lds_win = (size_t)cap_win * 8 + fixed_win, lds_cmp = (size_t)cap_cmp * 8 + fixed_cmp;
    static int attr_set = -1;           // the attribute is per device: keyed on the current ordinal
    int attr_dev = 0;
    HHX_HIP(hipGetDevice(&attr_dev));
    if (attr_set != attr_dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_expand_compact, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = attr_dev;
    }
    static const bool debug = getenv("HHX_DEBUG") != nullptr;
    const int probe = (int)tune_get("probe", 0);
    (void)probe;
    // rows whose product count is well above the number of accumulator slots a dense sweep touches
    static const double wfac = getenv("HHX_WINDOW_FACTOR") ? atof(getenv("HHX_WINDOW_FACTOR")) : 0.5;
    const bool all_window = dense || fx;
    const i64 window_min = all_window ? 0 : std::max<i64>(4096, (i64)((double)n_cols * wfac));
    DevBuf<i32> list_w, list_c, list_t, list_h, row_cnt, indptr, g_win_cnt;
    DevBuf<i64> row_off, g_win_off, row_f;
    // hash class: rows of at most hash_max products (tune "hash_max", 0 = off); needs the bitmap next to a 64 KB table
    const size_t lds_hash = hash_lds_bytes(W);
    // off when B's rows are longer than the table can hold distinct columns anyway (iteration 0: the link matrix itself)
    const bool hash_fits = lds_hash <= 160 * 1024 && b->n_rows > 0 && b->nnz / b->n_rows <= HASH_LIMIT / 2;
    // ... and only for rows the window class would not serve faster.  Measured on the tails of the low inflations at n = 100k
    // (profiles/r04_tail_probe.jsonl): the hash kernel walks 3.4-5.3e11 products/s whatever the row, the window kernel 1.1e12 /s plus
    // ~0.6 us per row of sweeping its n accumulators — the hash class wins below ~4.5 n products per row, not up to 4 M
    // (at inflation 1.3, iterations 3-6: 312 ms through the hash class for 1.1e11 products that the window class walks in ~160)
    const i64 hash_dflt = std::min<i64>(4000000, std::max<i64>(65536, (i64)(4.5 * (double)n_cols)));
    const i64 hash_max = all_window ? -1 : (hash_fits ? std::max<i64>(0, tune_get("hash_max", hash_dflt)) : 0);
    DevBuf<int2> bjx;
    DevBuf<double> s_run;
    DevBuf<unsigned int> counts;
    DevBuf<unsigned long long> cursors;
    // re-use of B rows across output rows (k_expand_group): R rows of one attractor per workgroup, column windows R times narrower.
    // Generic stream only (iterations >= 1 of mcl(), hhx_spgemm excluded); tune "reuse": 0 off (the default), 2 or 4 rows per group.
    // OFF by default: measured (tools/lowtails.py --reuse-ab, profiles/r05_lowtails_reuse.jsonl) it streams 3.5-3.9 x fewer bytes, gives
    // the same bits, and is no faster — the per-product path of distinct 64-bit addends caps at ~1.1e12 products/s, the rate the
    // fabric-bound one-row kernel already reaches
    int reuse_R = (!use_cls && !dense && !coded.raw && a == b) ? (int)tune_get("reuse", 0) : 0;
    if (reuse_R != 2 && reuse_R != 4) reuse_R = 0;
    i32 cap_g = 0, n_win_g = 0;
    if (reuse_R) {
        static_assert(win_fixed_bytes() == 784, "GroupStride assumes the window kernel's fixed LDS bytes");
        const i32 cols_max = (reuse_R == 4 ? GroupStride<4>::value : GroupStride<2>::value) - N_DUMMY;      // columns of a member's window
        for (i64 wn = std::max<i64>(1, ((i64)n_cols + cols_max - 1) / cols_max);; ++wn) {
            const i32 c = (i32)((((i64)n_cols + wn - 1) / wn + 63) & ~63);
            if (c <= cols_max) { cap_g = c; n_win_g = (n_cols + c - 1) / c; break; }
        }
        if (cap_g < 1024) reuse_R = 0;                      // windows too narrow to be worth a launch each
    }
    const i32 n_win_tab = std::max(n_win, n_win_g);
    if (list_w.alloc((size_t)n_rows + 1) || list_c.alloc((size_t)n_rows + 1) || list_t.alloc((size_t)n_rows + 1) ||
        list_h.alloc((size_t)n_rows + 1) || row_f.alloc((size_t)n_rows + 1) || row_cnt.alloc((size_t)n_rows + 1) ||
        indptr.alloc((size_t)n_rows + 1) || row_off.alloc((size_t)n_rows + 1) || counts.alloc(4) || cursors.alloc(12) ||
        s_run.alloc((size_t)n_rows + 1) || g_win_off.alloc((size_t)n_rows * n_win_tab + 1) || g_win_cnt.alloc((size_t)n_rows * n_win_tab + 1))
        return 1;
    DevBuf<i32> grp_rows, Gp, Gj;
    DevBuf<float> Gx;
    i32 n_groups = 0;
    i64 group_entries = 0;              // entries of the union rows
    // Rows of the window class in the order of (min-hash of the pattern, row) — tune "row_order", 1 unless switched off.  The rows a launch has in flight together (one per
    // workgroup, 256 at a time) then walk largely the SAME rows of B, which they find in the Infinity Cache instead of HBM: the heavy iterations of a low-inflation tail
    // run at 1.31-1.37e12 products/s instead of 1.15-1.18e12 (C3, inflation 1.1: 7.39 -> 6.74 s).  The order of the rows changes no bit of the result.
    auto order_rows = [&](DevBuf<i32> &list, i64 nl) -> int {
        if (dense || coded.raw || a != b || !tune_get("row_order", 1)) return 0;
        KTimer kt("row_order");
        DevBuf<u64> key, val, skey, sval;
        if (key.alloc((size_t)nl) || val.alloc((size_t)nl) || skey.alloc((size_t)nl) || sval.alloc((size_t)nl)) return 1;
        k_row_minhash<<<(unsigned)std::min<i64>((nl + 3) / 4, 4096), 256, 0, g_stream>>>((i32)nl, list.p, a->indptr.p, a->indices.p, key.p, val.p);
        HHX_LAUNCH_CHECK();
        int rbits = 1;
        while (rbits < 31 && (n_rows >> rbits)) ++rbits;
        HHX_TRY(stable_sort_pairs_u64(val.p, sval.p, key.p, skey.p, nl, rbits));              // by row ...
        HHX_TRY(stable_sort_pairs_u64(skey.p, key.p, sval.p, val.p, nl, 32));                 // ... then, stably, by min-hash: the order does not depend on the list's
        k_rows_from_sorted<<<(unsigned)std::min<i64>((nl + 255) / 256, 4096), 256, 0, g_stream>>>((i32)nl, val.p, list.p);
        HHX_LAUNCH_CHECK();
        HHX_HIP(hipStreamSynchronize(g_stream));                                              // the sort buffers die with this scope
        return 0;
    };
    int group_mode = -1;                // -1: not decided yet (first attempt), 0: one row per workgroup, 1: groups
    DevBuf<int4> rec;                   // window kernel: records + stream of the right operand, built once per call
    DevBuf<unsigned short> c16;
    DevBuf<float> cls_x;
    double explicit_frac = 1.0;         // share of B's entries outside the value-uniform sub-segments
    i64 stream_slots = 0;               // slots of the operand stream (prefix included)
    if (cap_win + N_DUMMY > 65536) return fail("expand: column window wider than 16 bits");
    // candidate pool: early windows test against a partial row sum and admit more than finally survive
    i64 pool_cap = dense ? 64 : std::max<i64>(4 * a->nnz + 16 * (i64)n_rows, (i64)1 << 22);
    i64 cand_cap = dense ? 64 : (n_win > 1 ? 2 * pool_cap : pool_cap);
    // Inside mcl()'s loop (hhx_mcl.hip) the iteration before says what this one will need: its demand + 50 % instead of 4 and 8 entries per entry of
    // A.  The guess above made the tail at inflation 1.1 of a 100k-contig matrix ask for 22 GB of pools per iteration, of sizes that change every
    // iteration — 67-97 GB of FRESH device memory per tail, 1.2-1.8 s of its 7.4 s on the caller's thread (tools/tail_alloc_probe.py,
    // profiles/r06_tail_alloc_probe.json).  An overflow costs one retry of the iteration with the exact demand, as before.
    if (!dense && !coded.raw) {
        if (g_expand_hint.out > 0) {
            pool_cap = std::max<i64>(g_expand_hint.out + g_expand_hint.out / 2 + 16 * (i64)n_rows, (i64)1 << 22);
            cand_cap = std::max<i64>(g_expand_hint.cand + g_expand_hint.cand / 2 + 16 * (i64)n_rows, (i64)1 << 22);
        }
        g_expand_hint = ExpandDemand();
    }
    if (coded.plan_out) { coded.plan_out[0] = cap_win; coded.plan_out[1] = n_win; }
    if (coded.raw) {                    // every entry is kept: at most dense, and at most one entry per product
        HHX_HIP(hipMemsetAsync(counts.p, 0, 4 * sizeof(unsigned int), g_stream));
        HHX_HIP(hipMemsetAsync(cursors.p, 0, 12 * sizeof(unsigned long long), g_stream));
        k_classify<<<(unsigned)std::max<i64>(1, std::min<i64>(((i64)n_rows + 63) / 64, 4096)), 256, 0, g_stream>>>(
            n_rows, a->indptr.p, a->indices.p, b->indptr.p, window_min, (i64)0, (i64)TINY_MAX, list_w.p, list_c.p, list_t.p, list_h.p, row_f.p, counts.p, cursors.p);
        HHX_LAUNCH_CHECK();
        unsigned long long products = 0;
        HHX_HIP(hipMemcpyAsync(&products, cursors.p + 4, sizeof products, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        const i64 dense = (i64)n_rows * n_cols;
        pool_cap = cand_cap = std::min<i64>(dense, (i64)products) + n_rows;
    }
    for (int attempt = 0; attempt < 6; ++attempt) {
        DevBuf<i32> cand_col, out_col;
        DevBuf<float> cand_val, out_val;
        if (cand_col.alloc((size_t)cand_cap) || cand_val.alloc((size_t)cand_cap) || out_col.alloc((size_t)pool_cap) ||
            out_val.alloc((size_t)pool_cap)) return 1;
        HHX_HIP(hipMemsetAsync(counts.p, 0, 4 * sizeof(unsigned int), g_stream));
        HHX_HIP(hipMemsetAsync(cursors.p, 0, 12 * sizeof(unsigned long long), g_stream));
        k_classify<<<(unsigned)std::max<i64>(1, std::min<i64>(((i64)n_rows + 63) / 64, 4096)), 256, 0, g_stream>>>(
            n_rows, a->indptr.p, a->indices.p, b->indptr.p, window_min, hash_max, all_window ? (i64)-1 : (i64)TINY_MAX, list_w.p, list_c.p, list_t.p, list_h.p, row_f.p,
            counts.p, cursors.p);
        HHX_LAUNCH_CHECK();
        unsigned int hc[4];
        unsigned long long hw[2];                            // products / A entries of the window class
        HHX_HIP(hipMemcpyAsync(hc, counts.p, sizeof hc, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipMemcpyAsync(hw, cursors.p + 5, sizeof hw, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        ExParams P;
        P.Ap = a->indptr.p; P.Aj = a->indices.p; P.Ax = a->data.p;
        P.Bp = b->indptr.p; P.Bj = b->indices.p; P.Bx = b->data.p;
        P.n_rows = n_rows; P.n_cols = n_cols;
        P.scale = ldexp(1.0, fx_shift - 52); P.inv_scale = ldexp(1.0, 52 - fx_shift);
        P.r = (double)(float)inflation; P.square = inflation == 2.0; P.thr = (float)pruning;
        P.raw = coded.raw;
        P.cand_col = cand_col.p; P.cand_val = cand_val.p; P.cand_cap = cand_cap;
        P.out_col = out_col.p; P.out_val = out_val.p; P.out_cap = pool_cap;
        P.cursors = cursors.p; P.row_off = row_off.p; P.row_cnt = row_cnt.p;
        P.n_win = n_win;
        P.s_run = s_run.p; P.g_win_off = g_win_off.p; P.g_win_cnt = g_win_cnt.p;
        P.dense = coded.dense_out; P.dense_ld = coded.dense_ld ? coded.dense_ld : (i64)n_cols;
        P.W = fx ? coded.W : nullptr; P.A16 = coded.a16; P.row_div = coded.a_row_sum; P.fx_inv = ldexp(1.0, -coded.shift);
        P.sym = coded.sym; P.sym_row0 = coded.sym_row0;
        P.tri = coded.tri; P.tri_ldn = (i64)n_win * cap_win;
        P.Sc16 = nullptr; P.Sx = nullptr; P.rec = nullptr; P.Bjx = nullptr; P.narrow_classes = 0; P.wb = WB_MAX;
        if (hc[3]) {                                      // hash class first: it may add rows to the window / compact lists
            if (!bjx.p) {                                 // B as 8-byte (column, value) words, built once per call
                if (bjx.alloc((size_t)b->nnz + 4)) return 1;
                k_pack_jx<<<(unsigned)std::max<i64>(1, std::min<i64>((b->nnz + 255) / 256, 256 * 16)), 256, 0, g_stream>>>(b->nnz, b->indices.p, b->data.p, bjx.p);
                HHX_LAUNCH_CHECK();
            }
            P.Bjx = bjx.p;
            static int hash_attr = -1;
            int hash_dev = 0;
            HHX_HIP(hipGetDevice(&hash_dev));
            if (hash_attr != hash_dev) {
                HHX_HIP(hipFuncSetAttribute((const void *)k_expand_hash, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hash_attr = hash_dev;
            }
            {
                KTimer kt("expand_hash");
                const unsigned per_cu = lds_hash > 80 * 1024 ? 1 : 2;
                k_expand_hash<<<std::min<unsigned>(hc[3], 256 * per_cu * 4), HASH_T, lds_hash, g_stream>>>(P, list_h.p, (i32)hc[3], W, row_f.p, window_min,
                                                                                                      list_w.p, list_c.p, counts.p);
#ifdef HHX_HASH_STATS
                {
                    unsigned long long hs[4] = {0, 0, 0, 0};
                    (void)hipStreamSynchronize(g_stream);
                    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_hash_stats), sizeof hs);
                    fprintf(stderr, "[hash stats] rows %u: consumes %llu, leftovers %llu (%.2f per consume of 512), sum of per-consume maxima %llu (%.2f per consume)\n", hc[3], hs[0], hs[1],
                            hs[0] ? (double)hs[1] / hs[0] : 0.0, hs[2], hs[0] ? (double)hs[2] / hs[0] : 0.0);
                    unsigned long long z[4] = {0, 0, 0, 0};
                    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hash_stats), z, sizeof z);
                }
#endif
            }
            HHX_LAUNCH_CHECK();
            HHX_HIP(hipMemcpyAsync(hc, counts.p, sizeof hc, hipMemcpyDeviceToHost, g_stream));
            HHX_HIP(hipMemcpyAsync(hw, cursors.p + 5, sizeof hw, hipMemcpyDeviceToHost, g_stream));
            HHX_HIP(hipStreamSynchronize(g_stream));
        }
        const unsigned n_hash_rows = hc[3];
        // ---- rows of the window class grouped by attractor (k_expand_group): decided once per call
        if (group_mode < 0 || (group_mode == 1 && attempt > 0)) {
            group_mode = 0;
            n_groups = 0;
            const double a_len0 = hc[0] ? (double)hw[1] / (double)hc[0] : 0.0;
            if (reuse_R && hc[0] >= 1024 && a_len0 >= 128.0) {
                KTimer kt("group_build");
                const i64 nl = (i64)hc[0];
                DevBuf<u64> key, val, skey, sval;
                if (key.alloc((size_t)nl) || val.alloc((size_t)nl) || skey.alloc((size_t)nl) || sval.alloc((size_t)nl)) return 1;
                k_row_attractor<<<(unsigned)std::min<i64>((nl + 3) / 4, 4096), 256, 0, g_stream>>>((i32)nl, list_w.p, a->indptr.p, a->indices.p, a->data.p, key.p, val.p);
                HHX_LAUNCH_CHECK();
                // rows in list order are not ascending (the classification appends by wave): sort by (attractor, row)
                int rbits = 1;
                while (rbits < 31 && (n_rows >> rbits)) ++rbits;
                int cbits = 1;
                while (cbits < 31 && (n_cols >> cbits)) ++cbits;
                HHX_TRY(stable_sort_pairs_u64(val.p, sval.p, key.p, skey.p, nl, rbits));              // by row ...
                HHX_TRY(stable_sort_pairs_u64(skey.p, key.p, sval.p, val.p, nl, cbits));              // ... then, stably, by attractor
                std::vector<u64> h_att((size_t)nl), h_row((size_t)nl);
                HHX_HIP(hipMemcpyAsync(h_att.data(), key.p, 8 * (size_t)nl, hipMemcpyDeviceToHost, g_stream));
                HHX_HIP(hipMemcpyAsync(h_row.data(), val.p, 8 * (size_t)nl, hipMemcpyDeviceToHost, g_stream));
                HHX_HIP(hipStreamSynchronize(g_stream));
                std::vector<i32> h_grp;
                h_grp.reserve((size_t)nl + (size_t)reuse_R);
                for (i64 p0 = 0; p0 < nl;) {
                    i64 p1 = p0 + 1;
                    while (p1 < nl && p1 - p0 < reuse_R && h_att[(size_t)p1] == h_att[(size_t)p0]) ++p1;
                    for (i64 q = p0; q < p0 + reuse_R; ++q) h_grp.push_back(q < p1 ? (i32)h_row[(size_t)q] : -1);
                    p0 = p1;
                }
                n_groups = (i32)(h_grp.size() / (size_t)reuse_R);
                if ((i64)n_groups * 2 <= nl) {                // on average two rows or more per group: the walk is shared
                    const i32 Wb = (n_cols + 31) / 32;
                    const size_t lds_u = (size_t)Wb * 8 + 64;
                    DevBuf<i32> gcnt;
                    if (grp_rows.alloc(h_grp.size()) || gcnt.alloc((size_t)n_groups + 1) || Gp.alloc((size_t)n_groups + 2)) return 1;
                    HHX_HIP(hipMemcpyAsync(grp_rows.p, h_grp.data(), sizeof(i32) * h_grp.size(), hipMemcpyHostToDevice, g_stream));
                    static int gu_attr = -1;
                    int gu_dev = 0;
                    HHX_HIP(hipGetDevice(&gu_dev));
                    if (gu_attr != gu_dev) {
                        HHX_HIP(hipFuncSetAttribute((const void *)k_group_union<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        HHX_HIP(hipFuncSetAttribute((const void *)k_group_union<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        gu_attr = gu_dev;
                    }
                    if (lds_u <= 160 * 1024) {
                        const unsigned ggrid = (unsigned)std::min<i32>(n_groups, 256 * 8);
                        k_group_union<false><<<ggrid, 256, lds_u, g_stream>>>(n_groups, reuse_R, grp_rows.p, a->indptr.p, a->indices.p, a->data.p, Wb, gcnt.p, nullptr, nullptr, nullptr);
                        HHX_LAUNCH_CHECK();
                        i64 union_nnz = 0;
                        HHX_TRY(exclusive_scan_i32(gcnt.p, Gp.p, n_groups, &union_nnz));
                        if (union_nnz < (i64)INT32_MAX / 8) {
                            if (Gj.alloc((size_t)union_nnz + 1) || Gx.alloc((size_t)union_nnz * (size_t)reuse_R + 1)) return 1;
                            k_group_union<true><<<ggrid, 256, lds_u, g_stream>>>(n_groups, reuse_R, grp_rows.p, a->indptr.p, a->indices.p, a->data.p, Wb, nullptr, Gp.p, Gj.p, Gx.p);
                            HHX_LAUNCH_CHECK();
                            HHX_HIP(hipStreamSynchronize(g_stream));      // h_grp dies with this scope
                            group_mode = 1;
                            group_entries = union_nnz;
                            if (debug) fprintf(stderr, "[hhx expand] re-use: %lld rows in %d groups of <= %d, union rows hold %lld entries (the rows: %llu)\n",
                                               (long long)nl, n_groups, reuse_R, (long long)union_nnz, hw[1]);
                        }
                    }
                }
            }
        }
        if (group_mode == 0 && attempt == 0 && hc[0] >= 1024) HHX_TRY(order_rows(list_w, (i64)hc[0]));
        const i32 cap_use = group_mode == 1 ? cap_g : cap_win;
        const i32 n_win_use = group_mode == 1 ? n_win_g : n_win;
        P.n_win = n_win_use;
        // mean length of a B-row segment inside one column window: tiles of 2 / 4 / 8 entries per lane
        const int tile_env = (int)tune_get("tile_u", 0);
        const double seg_len = hw[1] ? (double)hw[0] / (double)hw[1] / (double)n_win_use : 0.0;
        int tile_u = 0;
        const bool long_segments = seg_len >= 192.0;          // iteration 0 (the link matrix is the operand) vs the pruned iterations: timed apart
        if (hc[0]) {
            const unsigned grid = std::min<unsigned>(hc[0], 256);
            if (!rec.p) {                                 // the operand stream of b, built once per call
                const i64 segs = (i64)b->n_rows * n_win_use;
                DevBuf<int4> cnt4;
                DevBuf<i64> sizes, offs;
                DevBuf<unsigned long long> n_uni;
                if (rec.alloc(2 * (size_t)segs + 2) || cnt4.alloc((size_t)segs + 1) || sizes.alloc((size_t)segs + 1) || offs.alloc((size_t)segs + 2) ||
                    n_uni.alloc(1)) return 1;
                HHX_HIP(hipMemsetAsync(n_uni.p, 0, sizeof(unsigned long long), g_stream));
                const unsigned lgrid = (unsigned)std::max<i64>(1, std::min<i64>((segs + 3) / 4, 65536));
                KTimer kt("class_layout");
                k_layout_sizes<<<lgrid, 256, 0, g_stream>>>(b->n_rows, n_win_use, cap_use, n_classes, b->indptr.p, b->indices.p, use_cls ? coded.n16 : nullptr,
                                                            cnt4.p, sizes.p, n_uni.p);
                HHX_LAUNCH_CHECK();
                i64 slots = 0;
                HHX_TRY(exclusive_scan_i64(sizes.p, offs.p, segs, &slots));
                if (slots > (i64)INT32_MAX - 4096) return fail("expand: the padded operand stream needs %lld slots (int32 cursors)", (long long)slots);
                unsigned long long h = 0;
                HHX_HIP(hipMemcpyAsync(&h, n_uni.p, sizeof h, hipMemcpyDeviceToHost, g_stream));
                HHX_HIP(hipStreamSynchronize(g_stream));
                explicit_frac = b->nnz ? 1.0 - (double)h / (double)b->nnz : 1.0;
                // slack: a wide tile reads up to 8 entries past a sub-segment end, an exhausted cursor entry 0
                if (slots + STREAM_PREFIX > (i64)INT32_MAX - 4096) return fail("expand: the padded operand stream needs %lld slots (int32 cursors)", (long long)slots);
                // + a tile of slack: a masked lane of an explicit tile reads the first entry of its block, whatever lies there (xtile_fetch)
                if (c16.alloc((size_t)slots + STREAM_PREFIX + 8 * HHX_WAVE + 64) || cls_x.alloc((size_t)slots + STREAM_PREFIX + 8 * HHX_WAVE + 64)) return 1;
                stream_slots = slots + STREAM_PREFIX;
                if (use_cls && tune_get("cls_balance", 1))
                    k_layout_write<true><<<lgrid, 256, 0, g_stream>>>(b->n_rows, n_win_use, cap_use, n_classes, b->indptr.p, b->indices.p, b->data.p, coded.n16,
                                                                      coded.row_sum, cnt4.p, offs.p, c16.p, cls_x.p, rec.p, fx ? 1 : 0);
                else
                    k_layout_write<false><<<lgrid, 256, 0, g_stream>>>(b->n_rows, n_win_use, cap_use, n_classes, b->indptr.p, b->indices.p, b->data.p,
                                                                       use_cls ? coded.n16 : nullptr, use_cls ? coded.row_sum : nullptr, cnt4.p, offs.p, c16.p, cls_x.p, rec.p, fx ? 1 : 0);
                HHX_LAUNCH_CHECK();
                HHX_HIP(hipStreamSynchronize(g_stream));         // cnt4 / sizes / offs die here
            }
            P.Sc16 = c16.p; P.Sx = cls_x.p; P.rec = rec.p;
            P.narrow_classes = use_cls ? (n_classes > 1 ? 1 : 0) : -1;
            // batches: 32 A entries per wave draw when the rows are long; for the few-hundred-entry rows of the later iterations
            // 8, so that all sixteen waves of the workgroup get work out of one row
            const double a_len = group_mode == 1 ? (double)group_entries / (double)std::max(n_groups, 1) : (hc[0] ? (double)hw[1] / (double)hc[0] : 0.0);
            P.wb = (i32)tune_get("win_batch", a_len >= 1536.0 ? 32 : (a_len >= 512.0 ? 16 : 8));
            if (P.wb < 1 || P.wb > WB_MAX) P.wb = WB_MAX;
            // explicit tiles: UX x 64 entries, sized to the mean explicit sub-segment (a tile costs its 2 UX loads and UX
            // LDS atomics per lane whether filled or not); tune "tile_u" overrides
            // block tiles (pass_blocks): the general operand only — no uniform sub-segments, float values, the padding written by k_layout_write
            int block_tiles = (!use_cls && !fx && P.narrow_classes == -1 && group_mode != 1) ? (int)tune_get("block_tiles", 15) : 0;
            if (block_tiles >= 10 && block_tiles < 20 && stream_slots >= ((i64)1 << 30) - 4096) block_tiles -= 10;      // 32-bit byte offsets of the float32 values
            const double xlen = seg_len * explicit_frac;
            const int ux = tile_env ? tile_env : (xlen > 288.0 ? 8 : (xlen > 200.0 ? 4 : (xlen > 136.0 ? 3 : (xlen > 68.0 ? 2 : 1))));
            tile_u = ux;
            if (group_mode == 1) {
                KTimer kt("expand_group", n_win_use);
                GroupOp op{grp_rows.p, Gp.p, Gj.p, Gx.p, n_groups};
                const size_t lds_g = (size_t)reuse_R * (reuse_R == 4 ? GroupStride<4>::value : GroupStride<2>::value) * 8 + fixed_win;
                const unsigned ggrid = (unsigned)std::min<i32>(n_groups, 256);
                if (reuse_R == 4) {
                    if (ux >= 4) HHX_TRY((launch_group<4, 4, 4>(P, op, cap_use, lds_g, ggrid)));
                    else if (ux >= 2) HHX_TRY((launch_group<4, 2, 8>(P, op, cap_use, lds_g, ggrid)));
                    else HHX_TRY((launch_group<4, 1, 8>(P, op, cap_use, lds_g, ggrid)));
                } else {
                    if (ux >= 4) HHX_TRY((launch_group<2, 4, 4>(P, op, cap_use, lds_g, ggrid)));
                    else if (ux >= 2) HHX_TRY((launch_group<2, 2, 8>(P, op, cap_use, lds_g, ggrid)));
                    else HHX_TRY((launch_group<2, 1, 8>(P, op, cap_use, lds_g, ggrid)));
                }
            } else {
                KTimer kt(long_segments ? "expand_window" : "expand_window_short", n_win);
#ifdef HHX_PROBE_BUILD             // measurement build only (-DHHX_PROBE_BUILD): the atomics switched off, results are garbage
                if (probe == 1 && ux >= 4) HHX_TRY((launch_window<1, 8, 3, 3>(P, list_w.p, (i32)hc[0], cap_win, lds_win, grid)));
                else if (probe == 1) HHX_TRY((launch_window<1, 3, 8, 3>(P, list_w.p, (i32)hc[0], cap_win, lds_win, grid)));
                else
#endif
#define HHX_BLK(UXV, GV, AMV) HHX_TRY((launch_window_fx<0, UXV, GV, AMV, false, EX_T_WIN, true>(P, list_w.p, (i32)hc[0], cap_win, lds_win, grid)))
                if (block_tiles && prof_enabled()) prof_count("expand_block_tile_launches", n_win);
                if (block_tiles == 15) HHX_BLK(4, 5, 1);          // groups of 20 blocks, one 32-bit offset per block
                else if (block_tiles == 5) HHX_BLK(4, 5, 0);      // ... a scalar base per block (streams of 2^30 slots and more)
                else if (block_tiles == 11) HHX_BLK(4, 4, 1);     // measured beside them (C3 tail at 1.1: 5.06 / 5.15 / 5.23 s against 5.01 / 5.17 s; tiles per segment: 5.81 s)
                else if (block_tiles == 1) HHX_BLK(4, 4, 0);
                else if (block_tiles == 31) HHX_BLK(4, 4, 3);     // the loads written by hand (no VALU instruction for an address): 4.92 s — 1.7 % for loads the compiler cannot see: not the default
                else if (block_tiles == 2) HHX_BLK(8, 3, 0);
                else if (block_tiles) return fail("expand: block_tiles %d is not a shape", block_tiles);
#undef HHX_BLK
                else if (ux >= 8) HHX_TRY((launch_window<0, 8, 3, 3>(P, list_w.p, (i32)hc[0], cap_win, lds_win, grid)));
                else if (ux == 4) HHX_TRY((launch_window<0, 4, 4, 3>(P, list_w.p, (i32)hc[0], cap_win, lds_win, grid)));
                else if (ux == 3) HHX_TRY((launch_window<0, 3, 8, 3>(P, list_w.p, (i32)hc[0], cap_win, lds_win, grid)));
                else if (ux == 2) HHX_TRY((launch_window<0, 2, 8, 3>(P, list_w.p, (i32)hc[0], cap_win, lds_win, grid)));
                else HHX_TRY((launch_window<0, 1, 8, 3>(P, list_w.p, (i32)hc[0], cap_win, lds_win, grid)));
            }
            HHX_LAUNCH_CHECK();
            if (sym_whole && n_win > 1 && !coded.tri) {
                KTimer kt("dense_transpose");
                const unsigned tiles = (unsigned)(cap_win / 64), pairs = (unsigned)(n_win * (n_win - 1) / 2);
                k_transpose_lower<<<dim3(tiles, tiles, pairs), 256, 0, g_stream>>>(coded.dense_out, coded.dense_ld ? coded.dense_ld : (i64)n_cols, n_cols, cap_win);
                HHX_LAUNCH_CHECK();
            }
            if (!dense) {
                KTimer kt("expand_finalize");
                k_expand_window_finalize<<<std::min<unsigned>(hc[0], 256 * 8), EX_T_CMP, ex_fixed_bytes(0, 0), g_stream>>>(P, list_w.p, (i32)hc[0]);
            }
        }
        HHX_LAUNCH_CHECK();
        if (hc[2]) {
            KTimer kt("expand_tiny");
            k_expand_tiny<<<(unsigned)std::max<i64>(1, std::min<i64>(((i64)hc[2] + 255) / 256, 4096)), 256, 0, g_stream>>>(P, list_t.p, (i32)hc[2]);
        }
        HHX_LAUNCH_CHECK();
        if (hc[1]) {
            KTimer kt("expand_compact");
            const unsigned per_cu = lds_cmp > 80 * 1024 ? 1 : 2;
            k_expand_compact<<<std::min<unsigned>(hc[1], 256 * per_cu * 4), EX_T_CMP, lds_cmp, g_stream>>>(P, list_c.p, (i32)hc[1], cap_cmp, W);
        }
        HHX_LAUNCH_CHECK();
        unsigned long long cur[12];
        HHX_HIP(hipMemcpyAsync(cur, cursors.p, sizeof cur, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        if (debug)
            fprintf(stderr, "[hhx expand] %d x %d, nnzA %lld nnzB %lld: window rows %u (n_win %d x %d cols, lds %zu), hash rows %u, compact rows %u; "
                    "candidates %llu / %lld, survivors %llu / %lld, tile %d (segments of %.0f), %.1f ms since entry%s\n", n_rows, n_cols, (long long)a->nnz, (long long)b->nnz, hc[0], n_win,
                    cap_win, lds_win, n_hash_rows, hc[1], cur[0], (long long)cand_cap, cur[1], (long long)pool_cap, tile_u, seg_len,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count(), cur[2] ? "  OVERFLOW -> retry" : "");
        if (dense) {                                   // the rows are in coded.dense_out; nothing to pack
            if (hc[0] != (unsigned)n_rows) return fail("expand (dense): %u of %d rows took the window class", hc[0], n_rows);
            if (prof_enabled()) {                      // what the launches actually walked (the symmetric mode skips the blocks J < I)
                prof_count("expand_window_products", (i64)cur[8]);
                prof_count("expand_window_a_reads", (i64)cur[9]);
                if (use_cls) prof_count("expand_window_uniform_products", (i64)cur[7]);
                prof_count("expand_window_explicit_bytes", (i64)(fx ? 4 : 6) * ((i64)cur[8] - (use_cls ? (i64)cur[7] : 0)));
            }
            if (n_products) *n_products = (i64)cur[4];
            if (nnz_expanded) *nnz_expanded = (i64)cur[3];
            return 0;
        }
        if (cur[2]) {                                  // a pool overflowed: grow and redo the launches
            if ((i64)cur[0] > cand_cap) cand_cap = std::max<i64>(cand_cap * 2, (i64)cur[0] + (i64)n_rows);
            if ((i64)cur[1] > pool_cap) pool_cap = std::max<i64>(pool_cap * 2, (i64)cur[1] + (i64)n_rows);
            continue;
        }
        if (prof_enabled() && hc[0] && group_mode == 1) {
            prof_count("expand_group_loaded_entries", (i64)cur[8]);           // entries of B streamed (each serves up to R products)
            prof_count("expand_group_products", (i64)hw[0]);                  // products of the rows of the class
            prof_count("expand_group_a_reads", (i64)cur[9]);
        } else if (prof_enabled() && hc[0]) {
            const bool longseg = long_segments;
            prof_count(longseg ? "expand_window_products" : "expand_window_short_products", (i64)cur[8]);
            prof_count(longseg ? "expand_window_a_reads" : "expand_window_short_a_reads", (i64)cur[9]);
            if (longseg) prof_count("expand_window_explicit_bytes", (i64)(fx ? 4 : 6) * ((i64)cur[8] - (use_cls ? (i64)cur[7] : 0)));
            if (use_cls) prof_count("expand_window_uniform_products", (i64)cur[7]);
        }
        if (n_products) *n_products = (i64)cur[4];
        if (nnz_expanded) *nnz_expanded = (i64)cur[3];
        g_expand_demand.out = (i64)cur[1];
        g_expand_demand.cand = (i64)cur[0];
        if (attempt) prof_count("expand_pool_retries", attempt);
        return pack_rows_to_csr(n_rows, n_cols, row_cnt.p, indptr.p, row_off.p, out_col.p, out_val.p, out);
    }
    return fail("expand: survivor pool kept overflowing");
}

// ---- the inflation sweep: one expansion, every inflation (run_mcl_clustering :2146-2158) ------------------------------
// The reference pre-expands once (:2146-2147) and restarts mcl() at every inflation from that matrix.  M^2 of a Hi-C link matrix is
// (nearly) dense — n^2 entries at n = 100k — so here a ROW BLOCK of it is stored as plain float32 (4 B per entry instead of the 8 of a
// CSR entry, no index traffic) by the window kernel's dense mode, and iteration 0 of every inflation is the epilogue alone over that
// block (k_dense_epilogue + k_expand_window_finalize): the 1.15e12 products of the expansion are walked once for the whole sweep.
// How the dense block of ALL rows of an order-n link matrix in the symmetric integer mode is stored — 1: the square (n rows of n floats:
// the upper block triangle is computed, the rest mirrored by k_transpose_lower), 2: the upper block triangle alone (+ a scratch block row
// while the rows are finished), 0: neither fits next to the operand stream and the pools (the caller walks all products into the
// fused epilogue instead).  tune "dense_tri": 1 forces the triangle, 0 forbids it.
int hhx_dense_layout(i32 n_rows, i32 n_cols, i64 nnz_b) {
    i32 cap = 0, n_win = 0;
    window_plan(n_cols, nnz_b, &cap, &n_win);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    const double avail = (double)free_b + (double)pool_cached_bytes();
    const double square = 4.0 * (double)n_rows * (double)(((i64)n_cols + 31) & ~(i64)31);
    const double tri = 4.0 * (double)tri_floats(n_win, cap) + 4.0 * (double)cap * (double)(n_win - 1) * cap;
    const double rest = 12.0 * (double)nnz_b + 8e9;             // operand stream (records, 16-bit columns, values: padded), candidate pools, the result
    const i64 mode = tune_get("dense_tri", -1);
    if (mode == 1 && n_win > 1) return tri + rest <= avail ? 2 : 0;
    // the square while it is a modest share of the device (n = 100k: 40 GB of 288); beyond that the triangle, even where the square
    // would still fit: a block of that size, cached between calls, starves everything else (n = 200k: 160 GB next to the resident
    // pairs made the ingest of the next step trim the pool and re-allocate — 5 s — and hipMalloc itself costs ~30 ms per GB)
    const bool square_fits = square * 1.25 + rest <= avail, tri_fits = mode != 0 && n_win > 1 && tri + rest <= avail;
    if (square_fits && (square <= 0.25 * (double)total_b || !tri_fits)) return 1;
    if (tri_fits) return 2;
    return 0;
}

int hhx_expand_dense_impl(const hhx_csr *a, const hhx_csr *b, const hhx_links_operand *lk, int fx_shift, hhx_dense **out, i64 *n_products,
                          i64 *nnz_expanded) {
    if (!a || !b || !out) return fail("null pointer");
    hhx_dense *d = new hhx_dense();
    d->n_rows = a->n_rows; d->n_cols = b->n_cols;
    d->ld = ((i64)b->n_cols + 31) & ~(i64)31;
    // Layout.  All rows in the symmetric integer mode: the upper block triangle alone when the square does not fit next to the
    // operand stream and the pools (n = 200k: 160 GB against 88) — the lower blocks are then never stored, the epilogue turns them
    // one block row at a time (hhx_dense_inflate_prune).  tune "dense_tri": 1 forces the triangle, 0 forbids it.
    const bool tri_ok = lk && lk->W && lk->sym && a->n_rows == b->n_rows && lk->a_row0 == 0;
    i32 plan_cap = 0, plan_win = 0;
    window_plan(b->n_cols, b->nnz, &plan_cap, &plan_win);
    const bool tri = tri_ok && hhx_dense_layout(a->n_rows, b->n_cols, b->nnz) == 2;
    d->tri = tri;
    d->ldn = (i64)plan_win * plan_cap;
    const size_t floats = tri ? (size_t)tri_floats(plan_win, plan_cap) : (size_t)a->n_rows * (size_t)d->ld;
    if (d->x.alloc(floats + 1)) { delete d; return 2; }      // 2: the block itself does not fit (callers may fall back)
    CodedOperand c;
    c.dense_out = d->x.p; c.dense_ld = d->ld; c.tri = tri ? 1 : 0;
    if (lk) {
        c.n16 = lk->n16; c.row_sum = lk->row_sum;
        if (lk->W) {                                   // integer arithmetic: the block holds y = float(S); its epilogue divides by d_i
            c.W = lk->W; c.shift = lk->shift; c.a16 = lk->n16 + lk->a_off; c.a_row_sum = lk->row_sum + lk->a_row0;
            c.sym = lk->sym; c.sym_row0 = lk->a_row0;
            if (d->row_div.alloc((size_t)a->n_rows + 1)) { delete d; return 1; }
            if (a->n_rows && hipMemcpyAsync(d->row_div.p, c.a_row_sum, sizeof(double) * (size_t)a->n_rows, hipMemcpyDeviceToDevice, g_stream) != hipSuccess) {
                delete d;
                return fail("hhx_expand_dense_impl: copy of the row sums failed");
            }
            d->integer = true;
        }
    }
    i32 plan[2] = {0, 0};
    c.plan_out = plan;
    const int rc = hhx_expand_impl(a, b, c, fx_shift, 2.0, 0.0, nullptr, n_products, nnz_expanded);
    if (rc) { delete d; return rc; }
    d->cap_win = plan[0]; d->n_win = plan[1];
    if (tri && (plan[0] != plan_cap || plan[1] != plan_win)) { delete d; return fail("hhx_expand_dense_impl: the window plan changed under the triangle"); }
    // A first guess of what the lowest inflation of a sweep keeps of these rows — 0.6 survivors and 1.0 candidates per entry of the operand's share of
    // the link matrix (measured at 100k contigs / inflation 1.1: 0.52 and 0.80) — so that the FIRST pass over the block can already take a group of
    // inflations with pools of the right order (without it the lowest inflation ran alone, twice: once to learn its demand).  Marked as
    // coming from below every inflation (last_inflation = 1): a wrong guess costs the retry of the inflations whose pools overflowed, as any hint does.
    if (tune_get("dense_seed_hint", 1) && b->n_rows > 0) {
        const double share = (double)a->n_rows / (double)b->n_rows;
        d->last_out = std::max<i64>((i64)(0.6 * share * (double)b->nnz), (i64)a->n_rows * 8);
        d->last_cand = std::max<i64>((i64)(1.0 * share * (double)b->nnz), (i64)a->n_rows * 16);
        d->last_inflation = 1.0;
    }
    *out = d;
    return 0;
}

extern "C" int hhx_dense_inflate_prune(const hhx_dense *d, double inflation, double pruning, hhx_csr **out) {
    if (!d || !out) return fail("null pointer");
    if (!(inflation > 0)) return fail("inflation must be positive");
    const i32 n_rows = d->n_rows, n_cols = d->n_cols, n_win = d->n_win, cap = d->cap_win;
    DevBuf<i32> row_cnt, indptr, g_win_cnt;
    DevBuf<i64> row_off, g_win_off;
    DevBuf<double> s_run;
    DevBuf<unsigned long long> cursors;
    if (row_cnt.alloc((size_t)n_rows + 1) || indptr.alloc((size_t)n_rows + 1) || row_off.alloc((size_t)n_rows + 1) || cursors.alloc(12) ||
        s_run.alloc((size_t)n_rows + 1) || g_win_off.alloc((size_t)n_rows * n_win + 1) || g_win_cnt.alloc((size_t)n_rows * n_win + 1)) return 1;
    static int attr_dev = -1;
    int dev = 0;
    HHX_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_dense_epilogue<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HHX_HIP(hipFuncSetAttribute((const void *)k_dense_epilogue<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HHX_HIP(hipFuncSetAttribute((const void *)k_dense_epilogue_sw<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HHX_HIP(hipFuncSetAttribute((const void *)k_dense_epilogue_sw<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_dev = dev;
    }
    // a pruned row holds at most 1 / pruning entries; an early window tests against a partial row sum and admits more
    i64 pool_cap = std::max<i64>((i64)n_rows * 512, (i64)1 << 22), cand_cap = 2 * pool_cap;
    if (d->last_out) {                                 // a sweep: the demand of the previous inflation, with room for a neighbouring one
        const bool above = inflation >= d->last_inflation;     // (demands shrink as the inflation grows: the same size re-uses the cached block)
        pool_cap = d->last_out + (above ? 0 : d->last_out / 2) + n_rows;
        cand_cap = d->last_cand + (above ? 0 : d->last_cand / 2) + n_rows;
    }
    for (int attempt = 0; attempt < 4; ++attempt) {
        DevBuf<i32> cand_col, out_col;
        DevBuf<float> cand_val, out_val;
        if (cand_col.alloc((size_t)cand_cap) || cand_val.alloc((size_t)cand_cap) || out_col.alloc((size_t)pool_cap) || out_val.alloc((size_t)pool_cap)) return 1;
        HHX_HIP(hipMemsetAsync(cursors.p, 0, 12 * sizeof(unsigned long long), g_stream));
        ExParams P;
        memset(&P, 0, sizeof P);
        P.n_rows = n_rows; P.n_cols = n_cols;
        P.scale = 1.0; P.inv_scale = 1.0;
        P.r = (double)(float)inflation; P.square = inflation == 2.0; P.thr = (float)pruning;
        P.cand_col = cand_col.p; P.cand_val = cand_val.p; P.cand_cap = cand_cap;
        P.out_col = out_col.p; P.out_val = out_val.p; P.out_cap = pool_cap;
        P.cursors = cursors.p; P.row_off = row_off.p; P.row_cnt = row_cnt.p;
        P.n_win = n_win; P.s_run = s_run.p; P.g_win_off = g_win_off.p; P.g_win_cnt = g_win_cnt.p;
        P.row_div = d->integer ? d->row_div.p : nullptr;
        if (n_rows) {
            // square block: one launch over all rows.  Upper block triangle: block row by block row — the blocks (J < I, I) are turned
            // into rows of block row I in a scratch block first (k_transpose_tri), then the rows of block row I are finished
            DevBuf<float> lower;
            const i64 lo_ld = (i64)(n_win - 1) * cap;
            if (d->tri && n_win > 1 && lower.alloc((size_t)cap * (size_t)lo_ld + 1)) return 1;
            static const int use_sw = getenv("HHX_DENSE_EPI_SW") ? atoi(getenv("HHX_DENSE_EPI_SW")) : 1;
            for (i32 I = 0; I < (d->tri ? n_win : 1); ++I) {
                DenseSrc S;
                if (d->tri) {
                    S.row0 = I * cap; S.row1 = std::min<i32>(n_rows, (I + 1) * cap);
                    S.up = d->x.p + tri_row_off(I, cap, d->ldn); S.up_ld = d->ldn - (i64)I * cap; S.up_win0 = I;
                    S.lo = lower.p; S.lo_ld = lo_ld;
                    if (I > 0) {
                        KTimer kt("dense_transpose");
                        const unsigned tiles = (unsigned)(cap / 64);
                        k_transpose_tri<<<dim3(tiles, (unsigned)((S.row1 - S.row0 + 63) / 64), (unsigned)I), 256, 0, g_stream>>>(d->x.p, lower.p, lo_ld, I, S.row1 - S.row0, cap, d->ldn);
                    }
                } else { S.row0 = 0; S.row1 = n_rows; S.up = d->x.p; S.up_ld = d->ld; S.up_win0 = 0; S.lo = nullptr; S.lo_ld = 0; }
                const i32 rows_here = S.row1 - S.row0;
                if (rows_here <= 0) continue;
                KTimer kt("dense_epilogue");
                const unsigned grid = std::min<unsigned>((unsigned)rows_here, 512);
                if (use_sw && (cap + EX_T_WIN - 1) / EX_T_WIN <= DE_PER) {
                    const unsigned grid1 = std::min<unsigned>((unsigned)rows_here, 256);
                    if (!P.square) k_dense_epilogue_sw<false, true><<<grid1, EX_T_WIN, dense_epi_sw_lds_bytes(cap), g_stream>>>(P, S, cap);
                    else k_dense_epilogue_sw<true, false><<<grid, EX_T_WIN, dense_epi_sw_lds_bytes(cap), g_stream>>>(P, S, cap);
                } else if (P.square) k_dense_epilogue<true><<<grid, EX_T_WIN, dense_epi_lds_bytes(cap), g_stream>>>(P, S, cap);
                else k_dense_epilogue<false><<<grid, EX_T_WIN, dense_epi_lds_bytes(cap), g_stream>>>(P, S, cap);
            }
            HHX_LAUNCH_CHECK();
            if (lower.p) HHX_HIP(hipStreamSynchronize(g_stream));          // the scratch block dies with this scope
            KTimer kt("expand_finalize");
            k_expand_window_finalize<<<std::min<unsigned>((unsigned)n_rows, 256 * 8), EX_T_CMP, ex_fixed_bytes(0, 0), g_stream>>>(P, nullptr, n_rows);
        }
        HHX_LAUNCH_CHECK();
        unsigned long long cur[8];
        HHX_HIP(hipMemcpyAsync(cur, cursors.p, sizeof cur, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        if (cur[2]) {                                  // a pool overflowed: the cursors hold the demand
            if ((i64)cur[0] > cand_cap) cand_cap = (i64)cur[0] + (i64)n_rows;
            if ((i64)cur[1] > pool_cap) pool_cap = std::max<i64>(pool_cap * 2, (i64)cur[1] + (i64)n_rows);
            continue;
        }
        d->last_cand = (i64)cur[0];
        d->last_out = (i64)cur[1];
        d->last_inflation = inflation;
        return pack_rows_to_csr(n_rows, n_cols, row_cnt.p, indptr.p, row_off.p, out_col.p, out_val.p, out);
    }
    return fail("dense inflate / prune: survivor pool kept overflowing");
}

// iteration 0 of mcl() (:2037-2042) of the block's rows at K inflations in ONE pass over the block (k_dense_epilogue_multi): outs[k] is
// bit for bit what hhx_dense_inflate_prune(d, inflations[k]) returns
static int dense_inflate_prune_multi_impl(const hhx_dense *d, int K, const double *inflations, double pruning, hhx_csr **outs);

// all or nothing: whatever path fails (a pool allocation, a HIP call in the middle of a pass, a later part of a split), the matrices
// already packed are released and every outs[k] is null on a non-zero return
extern "C" int hhx_dense_inflate_prune_multi(const hhx_dense *d, int K, const double *inflations, double pruning, hhx_csr **outs) {
    if (!d || !inflations || !outs) return fail("null pointer");
    if (K < 1 || K > MULTI_MAX) return fail("hhx_dense_inflate_prune_multi: 1 to %d inflations per pass", MULTI_MAX);
    for (int k = 0; k < K; ++k) outs[k] = nullptr;
    for (int k = 0; k < K; ++k) if (!(inflations[k] > 0)) return fail("inflation must be positive");
    const int rc = dense_inflate_prune_multi_impl(d, K, inflations, pruning, outs);
    if (rc)
        for (int k = 0; k < K; ++k)
            if (outs[k]) { hhx_csr_free(outs[k]); outs[k] = nullptr; }
    return rc;
}

static int dense_inflate_prune_multi_impl(const hhx_dense *d, int K, const double *inflations, double pruning, hhx_csr **outs) {
    const i32 n_rows = d->n_rows, n_cols = d->n_cols, n_win = d->n_win, cap = d->cap_win;
    // inflation 2 is x * x, not exp2(2 log2 x) (numpy's `** 2`, hhx_powr's callers): it goes through the one-inflation kernel;
    // so does everything when the window is wider than the owned-slot registers of the fused kernel
    bool single = (cap + EX_T_WIN - 1) / EX_T_WIN > DE_PER || K == 1;
    for (int k = 0; k < K; ++k) single = single || inflations[k] == 2.0;
    if (single) {
        // the group pass FIRST: its hint (the demand of its lowest inflation) then covers the inflations that go alone — the other way round the lone
        // 2.0 left a hint that every lower inflation of the group overflowed (a retry of the pass: 70 ms of the 259 ms of the 1.6-2.0 group until round 6)
        int rc1 = 0;
        std::vector<double> rest;
        std::vector<int> at, alone;
        for (int k = 0; k < K; ++k) {
            if (inflations[k] == 2.0 || K == 1 || (cap + EX_T_WIN - 1) / EX_T_WIN > DE_PER) alone.push_back(k);
            else { rest.push_back(inflations[k]); at.push_back(k); }
        }
        if (!rest.empty()) {
            std::vector<hhx_csr *> o2(rest.size(), nullptr);
            rc1 = hhx_dense_inflate_prune_multi(d, (int)rest.size(), rest.data(), pruning, o2.data());
            for (size_t t = 0; t < at.size(); ++t) outs[at[t]] = o2[t];
        }
        for (size_t t = 0; t < alone.size() && !rc1; ++t) rc1 = hhx_dense_inflate_prune(d, inflations[alone[t]], pruning, &outs[alone[t]]);
        if (rc1) for (int k = 0; k < K; ++k) if (outs[k]) { hhx_csr_free(outs[k]); outs[k] = nullptr; }
        return rc1;
    }
    // No hint yet (first call on this block): the lowest inflation goes alone through the one-inflation kernel — its demand bounds
    // everybody else's, so the pools of the pass over the others are sized right at once (a retry of a K-inflation pass, or K pools
    // at a guess, cost more: device allocations run at ~30 ms per GB).  With a hint: as many inflations per pass as fit ~6 GB of pools.
    {
        int lowest = 0;
        for (int k = 1; k < K; ++k) if (inflations[k] < inflations[lowest]) lowest = k;
        i64 per_infl = d->last_out ? 8 * (d->last_out + d->last_cand) : 0;
        if (d->last_out && inflations[lowest] < d->last_inflation) per_infl *= 2;
        // pools of a pass: 6 GB when the device is tight, up to 32 GB when a quarter of what is free (driver + the pool's cache) allows it
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        const i64 pass_bytes = std::min<i64>((i64)32 << 30, std::max<i64>((i64)6 << 30, ((i64)free_b + pool_cached_bytes()) / 4));
        const int fit = !d->last_out ? 0 : (int)std::max<i64>(1, std::min<i64>(MULTI_MAX, pass_bytes / std::max<i64>(per_infl, 1)));
        if (!d->last_out || fit < K) {
            // split: [lowest alone | the rest] without a hint; [the first `fit` | the rest] with one — recursion ends at K == 1 or fit >= K
            std::vector<int> first, rest;
            if (!d->last_out) { for (int k = 0; k < K; ++k) (k == lowest ? first : rest).push_back(k); }
            else { for (int k = 0; k < K; ++k) (k < fit ? first : rest).push_back(k); }
            int rc2 = 0;
            for (const std::vector<int> *part : {&first, &rest}) {
                if (rc2 || part->empty()) continue;
                std::vector<double> r2;
                for (int k : *part) r2.push_back(inflations[k]);
                std::vector<hhx_csr *> o2(part->size(), nullptr);
                rc2 = hhx_dense_inflate_prune_multi(d, (int)r2.size(), r2.data(), pruning, o2.data());
                for (size_t t = 0; t < part->size(); ++t) outs[(*part)[t]] = o2[t];
            }
            if (rc2) for (int k = 0; k < K; ++k) if (outs[k]) { hhx_csr_free(outs[k]); outs[k] = nullptr; }
            return rc2;
        }
    }
    static int attr_dev = -1;
    int dev = 0;
    HHX_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_dense_epilogue_multi, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_dev = dev;
    }
    int hint_k = 0;
    for (int k = 1; k < K; ++k) if (inflations[k] < inflations[hint_k]) hint_k = k;
    struct PerK {
        DevBuf<i32> row_cnt, indptr, g_win_cnt, cand_col, out_col;
        DevBuf<i64> row_off, g_win_off;
        DevBuf<float> cand_val, out_val;
        DevBuf<double> s_run;
        DevBuf<unsigned long long> cursors;
        i64 pool_cap = 0, cand_cap = 0;
        bool done = false;
    };
    std::vector<PerK> pk((size_t)K);
    for (int k = 0; k < K; ++k) {
        PerK &q = pk[(size_t)k];
        if (q.row_cnt.alloc((size_t)n_rows + 1) || q.indptr.alloc((size_t)n_rows + 1) || q.row_off.alloc((size_t)n_rows + 1) || q.cursors.alloc(12) ||
            q.s_run.alloc((size_t)n_rows + 1) || q.g_win_off.alloc((size_t)n_rows * n_win + 1) || q.g_win_cnt.alloc((size_t)n_rows * n_win + 1)) return 1;
        q.pool_cap = std::max<i64>((i64)n_rows * 512, (i64)1 << 22);
        q.cand_cap = 2 * q.pool_cap;
        if (d->last_out) {
            // the demand of the call before.  Survivors and candidates shrink as the inflation grows: an inflation at or above the
            // one that left the hint needs no more than it did (+ 10 %: the candidate test runs against partial row sums)
            // (exactly the hint's size when above it — the block the previous call left in the pool is then re-used as it is; a
            // larger request would be a fresh device allocation at ~30 ms per GB, and an overflow only costs a retry of this pass)
            const bool above = inflations[k] >= d->last_inflation;
            q.pool_cap = d->last_out + (above ? 0 : d->last_out) + n_rows;
            q.cand_cap = d->last_cand + (above ? 0 : d->last_cand) + n_rows;
        }
    }
    DevBuf<MultiOut> mo_dev;
    DevBuf<unsigned long long> cursors0;                    // P.cursors of the kernel: only [3] (entries of the block) is written
    if (mo_dev.alloc((size_t)MULTI_MAX) || cursors0.alloc(12)) return 1;
    int rc = 0;
    for (int attempt = 0; attempt < 4 && !rc; ++attempt) {
        // the inflations still to do (first attempt: all; then those whose pools overflowed, with the pools at their demand)
        std::vector<int> todo;
        for (int k = 0; k < K; ++k) if (!pk[(size_t)k].done) todo.push_back(k);
        if (todo.empty()) break;
        MultiOut mo[MULTI_MAX];
        memset(mo, 0, sizeof mo);
        for (size_t t = 0; t < todo.size(); ++t) {
            PerK &q = pk[(size_t)todo[t]];
            if (q.cand_col.alloc((size_t)q.cand_cap) || q.cand_val.alloc((size_t)q.cand_cap) || q.out_col.alloc((size_t)q.pool_cap) || q.out_val.alloc((size_t)q.pool_cap)) return 1;
            HHX_HIP(hipMemsetAsync(q.cursors.p, 0, 12 * sizeof(unsigned long long), g_stream));
            mo[t].r = (double)(float)inflations[todo[t]]; mo[t].square = inflations[todo[t]] == 2.0;
            mo[t].cand_col = q.cand_col.p; mo[t].cand_val = q.cand_val.p; mo[t].cand_cap = q.cand_cap;
            mo[t].cursors = q.cursors.p; mo[t].g_win_off = q.g_win_off.p; mo[t].g_win_cnt = q.g_win_cnt.p; mo[t].s_run = q.s_run.p;
        }
        HHX_HIP(hipMemcpyAsync(mo_dev.p, mo, sizeof mo, hipMemcpyHostToDevice, g_stream));
        HHX_HIP(hipMemsetAsync(cursors0.p, 0, 12 * sizeof(unsigned long long), g_stream));
        ExParams P;
        memset(&P, 0, sizeof P);
        P.n_rows = n_rows; P.n_cols = n_cols;
        P.scale = 1.0; P.inv_scale = 1.0;
        P.thr = (float)pruning;
        P.n_win = n_win;
        P.cursors = cursors0.p;
        P.row_div = d->integer ? d->row_div.p : nullptr;
        if (n_rows) {
            DevBuf<float> lower;
            const i64 lo_ld = (i64)(n_win - 1) * cap;
            if (d->tri && n_win > 1 && lower.alloc((size_t)cap * (size_t)lo_ld + 1)) return 1;
            for (i32 I = 0; I < (d->tri ? n_win : 1); ++I) {
                DenseSrc S;
                if (d->tri) {
                    S.row0 = I * cap; S.row1 = std::min<i32>(n_rows, (I + 1) * cap);
                    S.up = d->x.p + tri_row_off(I, cap, d->ldn); S.up_ld = d->ldn - (i64)I * cap; S.up_win0 = I;
                    S.lo = lower.p; S.lo_ld = lo_ld;
                    if (I > 0) {
                        KTimer kt("dense_transpose");
                        k_transpose_tri<<<dim3((unsigned)(cap / 64), (unsigned)((S.row1 - S.row0 + 63) / 64), (unsigned)I), 256, 0, g_stream>>>(d->x.p, lower.p, lo_ld, I, S.row1 - S.row0, cap, d->ldn);
                    }
                } else { S.row0 = 0; S.row1 = n_rows; S.up = d->x.p; S.up_ld = d->ld; S.up_win0 = 0; S.lo = nullptr; S.lo_ld = 0; }
                if (S.row1 <= S.row0) continue;
                KTimer kt("dense_epilogue");
                k_dense_epilogue_multi<<<std::min<unsigned>((unsigned)(S.row1 - S.row0), 256), EX_T_WIN, dense_epi_multi_lds_bytes(cap), g_stream>>>(P, S, cap, (i32)todo.size(), mo_dev.p);
            }
            HHX_LAUNCH_CHECK();
            for (size_t t = 0; t < todo.size(); ++t) {         // the rows of every inflation finished from its own candidate segments
                PerK &q = pk[(size_t)todo[t]];
                ExParams Q = P;
                Q.r = mo[t].r; Q.square = mo[t].square;
                Q.cand_col = q.cand_col.p; Q.cand_val = q.cand_val.p; Q.cand_cap = q.cand_cap;
                Q.out_col = q.out_col.p; Q.out_val = q.out_val.p; Q.out_cap = q.pool_cap;
                Q.cursors = q.cursors.p; Q.row_off = q.row_off.p; Q.row_cnt = q.row_cnt.p;
                Q.s_run = q.s_run.p; Q.g_win_off = q.g_win_off.p; Q.g_win_cnt = q.g_win_cnt.p;
                KTimer kt("expand_finalize");
                k_expand_window_finalize<<<std::min<unsigned>((unsigned)n_rows, 256 * 8), EX_T_CMP, ex_fixed_bytes(0, 0), g_stream>>>(Q, nullptr, n_rows);
            }
            HHX_LAUNCH_CHECK();
            HHX_HIP(hipStreamSynchronize(g_stream));             // (the scratch block row of the triangle dies with this scope)
        }
        for (size_t t = 0; t < todo.size() && !rc; ++t) {
            PerK &q = pk[(size_t)todo[t]];
            unsigned long long cur[4];
            HHX_HIP(hipMemcpyAsync(cur, q.cursors.p, sizeof cur, hipMemcpyDeviceToHost, g_stream));
            HHX_HIP(hipStreamSynchronize(g_stream));
            if (cur[2]) {                                      // a pool overflowed: the cursors hold the demand
                if ((i64)cur[0] > q.cand_cap) q.cand_cap = (i64)cur[0] + (i64)n_rows;
                if ((i64)cur[1] > q.pool_cap) q.pool_cap = std::max<i64>(q.pool_cap * 2, (i64)cur[1] + (i64)n_rows);
                continue;
            }
            if (todo[t] == hint_k) {                           // the hint for the next call: the demand of THIS call's lowest inflation (its largest)
                d->last_cand = std::max<i64>((i64)cur[0], 1); d->last_out = std::max<i64>((i64)cur[1], 1); d->last_inflation = inflations[todo[t]];
            }
            rc = pack_rows_to_csr(n_rows, n_cols, q.row_cnt.p, q.indptr.p, q.row_off.p, q.out_col.p, q.out_val.p, &outs[todo[t]]);
            q.done = true;
            q.cand_col.release(); q.cand_val.release(); q.out_col.release(); q.out_val.release();
        }
    }
    bool all = true;
    for (int k = 0; k < K; ++k) all = all && pk[(size_t)k].done;
    if (rc || !all) {
        for (int k = 0; k < K; ++k) if (outs[k]) { hhx_csr_free(outs[k]); outs[k] = nullptr; }
        return rc ? rc : fail("dense inflate / prune (multi): survivor pool kept overflowing");
    }
    return 0;
}

extern "C" int hhx_dense_shape(const hhx_dense *d, i32 *n_rows, i32 *n_cols, i64 *bytes) {
    if (!d) return fail("null handle");
    if (n_rows) *n_rows = d->n_rows;
    if (n_cols) *n_cols = d->n_cols;
    if (bytes) *bytes = (i64)sizeof(float) * (d->tri ? tri_floats(d->n_win, d->cap_win) : (i64)d->n_rows * d->ld);
    return 0;
}

// the block as device memory: n_rows rows of n_cols float32, *ld floats apart (multi-GPU: the ranks mirror their upper block triangles through it)
extern "C" int hhx_dense_device(const hhx_dense *d, void **x, i64 *ld, i32 *cap_win, i32 *n_win) {
    if (!d || !x || !ld) return fail("null pointer");
    if (d->tri) return fail("hhx_dense_device: this block is stored as its upper block triangle, not as rows");
    *x = d->x.p;
    *ld = d->ld;
    if (cap_win) *cap_win = d->cap_win;
    if (n_win) *n_win = d->n_win;
    return 0;
}

extern "C" int hhx_dense_free(hhx_dense *d) {
    delete d;
    return 0;
}

// ---- rectangles of the dense block on their way between the ranks (sharded.expand_links_symmetric): rank s hands rank r > s the
// rectangle Y[rows of s][columns = rows of r]; r stores its transpose; what mirrors inside a rank's own rows is turned in place.
// dst[c][r] = src[r][c], 64 x 64 tiles through LDS (both sides coalesced); rows / cols / pitches in floats, any alignment.
namespace {
__global__ __launch_bounds__(256) void k_transpose_rect(const float *__restrict__ src, i64 src_ld, float *__restrict__ dst, i64 dst_ld, i64 rows, i64 cols) {
    __shared__ float tile[64][65];
    const i64 r0 = (i64)blockIdx.y * 64, c0 = (i64)blockIdx.x * 64;       // source tile: rows r0.., columns c0..
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4)
        tile[k][tx] = (r0 + k < rows && c0 + tx < cols) ? src[(size_t)(r0 + k) * (size_t)src_ld + c0 + tx] : 0.0f;
    __syncthreads();
    for (int k = ty; k < 64; k += 4)
        if (c0 + k < cols && r0 + tx < rows) dst[(size_t)(c0 + k) * (size_t)dst_ld + r0 + tx] = tile[tx][k];
}
}  // namespace
// dst (cols x rows, rows dst_ld floats apart) = transpose of src (rows x cols, rows src_ld floats apart); device pointers; the two
// must not overlap
extern "C" int hhx_transpose_f32(const void *src, i64 src_ld, void *dst, i64 dst_ld, i64 rows, i64 cols) {
    if (rows < 0 || cols < 0 || src_ld < cols || dst_ld < rows) return fail("hhx_transpose_f32: bad shape");
    if (rows == 0 || cols == 0) return 0;
    if (!src || !dst) return fail("null pointer");
    const i64 gx = (cols + 63) / 64, gy = (rows + 63) / 64;
    if (gy > 65535) return fail("hhx_transpose_f32: more than 4 M rows");
    KTimer kt("dense_transpose");
    k_transpose_rect<<<dim3((unsigned)gx, (unsigned)gy), 256, 0, g_stream>>>((const float *)src, src_ld, (float *)dst, dst_ld, rows, cols);
    HHX_LAUNCH_CHECK();
    return 0;
}
// dst (rows x cols, packed or with its own pitch) = the rectangle src (rows x cols, rows src_ld floats apart): one strided device copy
extern "C" int hhx_copy_rect_f32(const void *src, i64 src_ld, void *dst, i64 dst_ld, i64 rows, i64 cols) {
    if (rows < 0 || cols < 0 || src_ld < cols || dst_ld < cols) return fail("hhx_copy_rect_f32: bad shape");
    if (rows == 0 || cols == 0) return 0;
    if (!src || !dst) return fail("null pointer");
    HHX_HIP(hipMemcpy2DAsync(dst, sizeof(float) * (size_t)dst_ld, src, sizeof(float) * (size_t)src_ld, sizeof(float) * (size_t)cols, (size_t)rows,
                             hipMemcpyDeviceToDevice, g_stream));
    return 0;
}

extern "C" int hhx_row_products(const hhx_csr *a, const hhx_csr *b, i64 *products_host) {
    if (!a || !b || !products_host) return fail("null pointer");
    if (a->n_cols != b->n_rows) return fail("row_products shape mismatch");
    DevBuf<i64> f;
    if (f.alloc((size_t)a->n_rows + 1)) return 1;
    k_row_products<<<(unsigned)std::max<i64>(1, std::min<i64>(((i64)a->n_rows + 3) / 4, 8192)), 256, 0, g_stream>>>(a->n_rows, a->indptr.p,
                                                                                                                   a->indices.p, b->indptr.p, f.p);
    HHX_LAUNCH_CHECK();
    HHX_HIP(hipMemcpyAsync(products_host, f.p, sizeof(i64) * (size_t)a->n_rows, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int hhx_expand_inflate_prune(const hhx_csr *a, const hhx_csr *b, int fx_shift, double inflation, double pruning,
                                        hhx_csr **out, i64 *n_products, i64 *nnz_expanded) {
    return hhx_expand_impl(a, b, CodedOperand(), fx_shift, inflation, pruning, out, n_products, nnz_expanded);
}

// b is the L1-normalised link matrix whose entry p equals float(n16[p] / row_sum[row]) (checked by the caller); with lk->W the
// products are formed in the integer arithmetic of the link matrix (a = rows [a_row0, ...) of it)
int hhx_expand_class_stream(const hhx_csr *a, const hhx_csr *b, const hhx_links_operand *lk, int fx_shift,
                     double inflation, double pruning, hhx_csr **out, i64 *n_products, i64 *nnz_expanded) {
    CodedOperand c;
    c.n16 = lk->n16; c.row_sum = lk->row_sum;
    if (lk->W) { c.W = lk->W; c.shift = lk->shift; c.a16 = lk->n16 + lk->a_off; c.a_row_sum = lk->row_sum + lk->a_row0; }
    return hhx_expand_impl(a, b, c, fx_shift, inflation, pruning, out, n_products, nnz_expanded);
}

// C = A * B for stochastic-like operands (entries in [0, 1], row sums of A <= 1): the fused kernels in plain-product mode
int hhx_expand_raw(const hhx_csr *a, const hhx_csr *b, int fx_shift, hhx_csr **out, i64 *n_products) {
    CodedOperand c;
    c.raw = 1;
    i64 nnzc = 0;
    return hhx_expand_impl(a, b, c, fx_shift, 2.0, 0.0, out, n_products, &nnzc);
}
