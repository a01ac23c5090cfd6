// .pairs text -> (ref id, pos, mref id, mpos) arrays on the device, plus the alignments.bed bytes:
// pairs_generator / pairs_generator_inter_ctgs (scripts/HapHiC_cluster.py:1539-1583).
//
//   reference, per line of the text file (universal newlines: '\n', '\r\n' and a lone '\r' end a line):
//       if not line.strip() or line.startswith('#'): continue                        :1552 / :1575
//       cols = line.split()                                                          :1554
//       ref, pos, mref, mpos = cols[1], int(cols[2]) - 1, cols[3], int(cols[4]) - 1  :1556
//       fbed.write('{ref}\t{pos}\t{pos}\t{cols[0]}/1\t255\t.\n{mref}\t{mpos}\t{mpos}\t{cols[0]}/2\t255\t.\n')  :1557
//       yield ref, mref, pos, mpos                 (inter_ctgs: only if ref != mref  :1582 — the ingest's skip_intra)
//
// The chunk of text lies in HBM; three streaming passes:
//   1. line breaks are counted per 4 KB block, scanned, and the line starts written (k_count_breaks / k_write_starts);
//   2. one thread per line finds the first five whitespace-separated tokens, resolves the two names through an
//      open-addressing table of the FASTA names (word-wise 64-bit hash, byte-verified) and parses the two integers
//      (k_parse_lines); a skipped line (blank, '#') yields ids of -1, which the ingest drops like any name that
//      is not in the FASTA (:1702), so no compaction is needed and stream order is untouched;
//   3. when the BED is wanted, the per-line output sizes of pass 2 are scanned and every line formats its two
//      records at its offset (k_bed_write).
// Whitespace is the ASCII subset of str.split()'s (\t \n \v \f \r \x1c-\x1f and space); int() takes an optional
// sign, digits and single underscores between digits.  A line with fewer than five columns or a malformed integer
// fails the call (the reference raises IndexError / ValueError there); positions must fit int32.
#include "hhx_common.h"

using namespace hhx;

struct hhx_pairs_parser {
    i32 n_names = 0;
    u32 mask = 0;
    DevBuf<u64> names;
    DevBuf<i64> name_off;
    DevBuf<i32> name_len;
    DevBuf<u64> slot_hash;
    DevBuf<i32> slot_id;
    // results of the last parse
    DevBuf<unsigned char> text, bed;
    DevBuf<i64> starts, bed_off;
    DevBuf<i32> id1, pos1, id2, pos2;
    DevBuf<i64> pos1w, pos2w;               // wide mode (hhx_pairs_parser_set_wide): 64-bit positions, contigs beyond 2^31 bp (:116-147)
    bool wide = false;
    DevBuf<unsigned long long> err;
    i64 n_lines = 0, bed_bytes = 0, lines_before = 0;
    // alignments.bed leaves through two pinned host buffers used in turn (hhx_pairs_parser_bed_host): the device -> host copy
    // runs at PCIe rate, and the caller's writer threads empty buffer k while chunk k + 1 is tokenised
    unsigned char *pin[2] = {nullptr, nullptr};
    size_t pin_cap[2] = {0, 0};
    int pin_next = 0;
    hhx_byte_sink *sink = nullptr;          // hhx_pairs_parser_set_bed_sink
    ~hhx_pairs_parser() {
        for (int k = 0; k < 2; ++k)
            if (pin[k]) (void)hipHostFree(pin[k]);
    }
};

namespace {

constexpr int TX_BLOCK = 4096;                 // bytes per workgroup step of the line-break passes
constexpr int LN_BLOCK = 128;                  // lines per workgroup of the parse / BED passes
constexpr int IN_CAP = 24 * 1024;              // LDS bytes for the text of those lines (longer spans read HBM directly)
constexpr int OUT_CAP = 36 * 1024;             // LDS bytes for their BED records

// name hash over the 8-byte words of the name (last word zero-padded), so that a lane hashes a 35-byte contig
// name in 5 steps; the table is byte-verified, the hash only has to spread
__host__ __device__ __forceinline__ u64 hash_step(u64 h, u64 w) { h = (h ^ w) * 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }
constexpr u64 HASH_SEED = 1469598103934665603ull;
__device__ __forceinline__ bool is_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

// bit k set: byte base + k ends a line ('\n', or a '\r' that is not followed by '\n'); 16 bytes per thread
__device__ __forceinline__ u32 break_mask(const unsigned char *__restrict__ t, i64 base, i64 n, bool aligned) {
    u32 m = 0;
    if (aligned && base + 16 < n) {                              // one 16-byte load + the look-ahead byte
        const uint4 v = *reinterpret_cast<const uint4 *>(t + base);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
        unsigned char nxt = t[base + 16];
#pragma unroll
        for (int k = 15; k >= 0; --k) {
            const unsigned char ch = (unsigned char)(w[k >> 2] >> (8 * (k & 3)));
            m |= (u32)(ch == '\n' || (ch == '\r' && nxt != '\n')) << k;
            nxt = ch;
        }
    } else {
        for (int k = 0; k < 16 && base + k < n; ++k) {
            const unsigned char ch = t[base + k];
            m |= (u32)(ch == '\n' || (ch == '\r' && (base + k + 1 >= n || t[base + k + 1] != '\n'))) << k;
        }
    }
    return m;
}

__global__ __launch_bounds__(256) void k_count_breaks(const unsigned char *__restrict__ t, i64 n, i64 n_blocks, i64 *__restrict__ counts) {
    __shared__ i32 wsum[4];
    const bool aligned = ((uintptr_t)t & 15) == 0;
    for (i64 b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        i32 c = __popc(break_mask(t, b * TX_BLOCK + (i64)threadIdx.x * 16, n, aligned));
        c = wave_sum_i32(c);
        if (lane_id() == 0) wsum[threadIdx.x / HHX_WAVE] = c;
        __syncthreads();
        if (threadIdx.x == 0) counts[b] = (i64)wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

// starts[1 + r] = p + 1 for the r-th break (starts[0] = 0 is written by thread 0 of block 0)
__global__ __launch_bounds__(256) void k_write_starts(const unsigned char *__restrict__ t, i64 n, i64 n_blocks,
                                                      const i64 *__restrict__ prefix, i64 *__restrict__ starts) {
    __shared__ i32 wsum[4];
    const bool aligned = ((uintptr_t)t & 15) == 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) starts[0] = 0;
    for (i64 b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const i64 base = b * TX_BLOCK + (i64)threadIdx.x * 16;
        u32 m = break_mask(t, base, n, aligned);
        const i32 c = __popc(m);
        i32 incl = c;                                            // inclusive wave scan
#pragma unroll
        for (int o = 1; o < HHX_WAVE; o <<= 1) {
            const i32 v = __shfl_up(incl, o, HHX_WAVE);
            if (lane_id() >= o) incl += v;
        }
        if (lane_id() == HHX_WAVE - 1) wsum[threadIdx.x / HHX_WAVE] = incl;
        __syncthreads();
        i64 at = prefix[b] + incl - c;
        for (int w = 0; w < (int)(threadIdx.x / HHX_WAVE); ++w) at += wsum[w];
        while (m) {
            const int k = __ffs(m) - 1;
            m &= m - 1;
            starts[1 + at++] = base + k + 1;
        }
        __syncthreads();
    }
}

// ---- byte readers: the lines of a workgroup staged in LDS (position -> LDS offset), or HBM directly
// operator()(p): the byte at p; word(p, nb): the nb <= 8 bytes at p as a little-endian u64 (upper bytes zero).
// LdsText keeps the last aligned 8-byte LDS word in registers: a sequential scan costs one ds_read_b64 per 8 bytes.
struct LdsText {
    const u64 *l64;
    i64 bias;                                                    // position of LDS byte 0 in the text (16-aligned)
    mutable i64 ci = -1;
    mutable u64 cw = 0;
    __device__ __forceinline__ LdsText(const unsigned char *lds, i64 b) : l64(reinterpret_cast<const u64 *>(lds)), bias(b) {}
    __device__ __forceinline__ unsigned char operator()(i64 p) const {
        const i64 o = p - bias, i = o >> 3;
        if (i != ci) { ci = i; cw = l64[i]; }
        return (unsigned char)(cw >> ((o & 7) * 8));
    }
    __device__ __forceinline__ u64 word(i64 p, int nb) const {
        const i64 o = p - bias, i = o >> 3;
        const int sh = (int)(o & 7) * 8;
        u64 w = l64[i] >> sh;
        if (sh) w |= l64[i + 1] << (64 - sh);                    // s_in carries 16 spare bytes for this
        return nb < 8 ? w & ((1ull << (8 * nb)) - 1) : w;
    }
};
struct HbmText {
    const unsigned char *t;
    __device__ __forceinline__ unsigned char operator()(i64 p) const { return t[p]; }
    __device__ __forceinline__ u64 word(i64 p, int nb) const {
        u64 w = 0;
        for (int q = 0; q < nb; ++q) w |= (u64)t[p + q] << (8 * q);
        return w;
    }
};

struct Tok { i64 s; i32 len; };
// tok[n] = (s, len) with n known only at run time: statically indexed selects keep the five tokens in registers
// (a dynamically indexed private array lands in scratch memory)
__device__ __forceinline__ void tok_set(Tok tok[5], int n, i64 s, i32 len) {
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (q == n) { tok[q].s = s; tok[q].len = len; }
}

// the first five tokens of [a, e); returns how many were found
template <class RD>
__device__ __forceinline__ int first_tokens(const RD &rd, i64 a, i64 e, Tok tok[5]) {
    int n = 0;
    i64 p = a;
    while (n < 5) {
        while (p < e && is_ws(rd(p))) ++p;
        if (p >= e) break;
        const i64 s = p;
        while (p < e && !is_ws(rd(p))) ++p;
        tok_set(tok, n, s, (i32)(p - s));
        ++n;
    }
    return n;
}

// whitespace bytes of an 8-byte word as an 8-bit mask.  Per byte, with the high bit forced on, (x - n) keeps the
// high bit iff the low 7 bits are >= n (n <= 0x80: no borrow crosses a byte), so two range tests give is_ws()
// exactly; the flags are gathered with one multiply (distinct (byte, shift) pairs never collide).
__device__ __forceinline__ u32 ws_mask8(u64 w) {
    constexpr u64 H = 0x8080808080808080ull, L = 0x0101010101010101ull;
    const u64 x = w | H;
    const u64 m = (((x - 9 * L) & ~(x - 14 * L)) | ((x - 0x1c * L) & ~(x - 0x21 * L))) & H & ~w;
    return (u32)((m * 0x0002040810204081ull) >> 56);
}

// first_tokens on the LDS-staged text: token boundaries from the whitespace masks of the aligned 8-byte words
__device__ __forceinline__ int first_tokens(const LdsText &rd, i64 a, i64 e, Tok tok[5]) {
    const u32 ao = (u32)(a - rd.bias), eo = (u32)(e - rd.bias);
    int n = 0;
    u32 ts = 0, in_tok = 0;
    for (u32 wi = ao >> 3; (wi << 3) < eo && n < 5; ++wi) {
        const i32 lo = (i32)ao - (i32)(wi << 3), hi = (i32)eo - (i32)(wi << 3);
        u32 valid = 0xffu;                                       // bytes outside [a, e) count as whitespace
        if (lo > 0) valid &= 0xffu << lo;
        if (hi < 8) valid &= (1u << hi) - 1;
        const u32 tokb = ~ws_mask8(rd.l64[wi]) & valid;          // bytes that belong to a token
        const u32 prev = (tokb << 1) | in_tok;
        const u32 starts = tokb & ~prev;
        u32 ev = (starts | (~tokb & prev)) & 0xffu;              // a token starts at / has ended before byte k
        while (ev && n < 5) {
            const int k = __ffs(ev) - 1;
            ev &= ev - 1;
            if ((starts >> k) & 1) ts = (wi << 3) + k;
            else { tok_set(tok, n, rd.bias + ts, (i32)((wi << 3) + k - ts)); ++n; }
        }
        in_tok = (tokb >> 7) & 1;
    }
    if (n < 5 && in_tok) { tok_set(tok, n, rd.bias + ts, (i32)(eo - ts)); ++n; }
    return n;
}

struct NameTable {
    const u64 *names;                                            // every name padded with zeros to whole 8-byte words
    const i64 *name_off;                                         // in words; name_len in bytes
    const i32 *name_len;
    const u64 *slot_hash;
    const i32 *slot_id;
    u32 mask;
};

template <class RD>
__device__ __forceinline__ u64 tok_word(const RD &rd, const Tok &k, int q) {
    return rd.word(k.s + 8 * (i64)q, k.len - 8 * q < 8 ? k.len - 8 * q : 8);
}
template <class RD>
__device__ __forceinline__ i32 lookup(const NameTable &T, const RD &rd, const Tok &k) {
    constexpr int MAXW = 8;                                      // the words of names up to 64 bytes stay in registers
    const int nw = (k.len + 7) >> 3;
    u64 w[MAXW];
    u64 h = HASH_SEED;
#pragma unroll
    for (int q = 0; q < MAXW; ++q)
        if (q < nw) { w[q] = tok_word(rd, k, q); h = hash_step(h, w[q]); }
    for (int q = MAXW; q < nw; ++q) h = hash_step(h, tok_word(rd, k, q));
    for (u32 s = (u32)h & T.mask;; s = (s + 1) & T.mask) {
        const i32 id = T.slot_id[s];
        if (id < 0) return -1;
        if (T.slot_hash[s] != h || T.name_len[id] != k.len) continue;
        const u64 *nm = T.names + T.name_off[id];
        u64 diff = 0;
#pragma unroll
        for (int q = 0; q < MAXW; ++q)
            if (q < nw) diff |= nm[q] ^ w[q];
        for (int q = MAXW; q < nw; ++q) diff |= nm[q] ^ tok_word(rd, k, q);
        if (!diff) return id;
    }
}

// int(token): [+-]? digit (_? digit)*   -> false when malformed; *range = 1 when value - 1 leaves the int32 window, 2 beyond 2^40
template <class RD>
__device__ __forceinline__ bool parse_int(const RD &rd, const Tok &k, i64 *val, int *range) {
    i32 q = 0;
    bool neg = false;
    const unsigned char c0 = rd(k.s);
    if (c0 == '+' || c0 == '-') { neg = c0 == '-'; ++q; }
    if (q >= k.len) return false;
    i64 v = 0;
    bool prev_digit = false, big = false;
    for (; q < k.len; ++q) {
        const unsigned char c = rd(k.s + q);
        if (c >= '0' && c <= '9') {
            if (v < (1ll << 40)) v = v * 10 + (c - '0'); else big = true;
            prev_digit = true;
        } else if (c == '_' && prev_digit && q + 1 < k.len) {
            prev_digit = false;
        } else {
            return false;
        }
    }
    if (!prev_digit) return false;
    v = neg ? -v : v;
    *range = big ? 2 : ((v - 1 > 2147483647ll || v - 1 < -2147483648ll) ? 1 : 0);
    *val = v;
    return true;
}

__device__ __forceinline__ i32 dec_len(i64 v) {
    i32 n = v < 0 ? 1 : 0;
    u64 a = v < 0 ? (u64)(-v) : (u64)v;
    do { ++n; a /= 10; } while (a);
    return n;
}
__device__ __forceinline__ unsigned char *put_dec(unsigned char *o, i64 v) {
    const i32 n = dec_len(v);
    u64 a = v < 0 ? (u64)(-v) : (u64)v;
    for (i32 k = n - 1; k >= (v < 0 ? 1 : 0); --k) { o[k] = (unsigned char)('0' + a % 10); a /= 10; }
    if (v < 0) o[0] = '-';
    return o + n;
}
template <class RD>
__device__ __forceinline__ unsigned char *put_tok(unsigned char *o, const RD &rd, const Tok &k) {
    for (i32 q = 0; q < k.len; ++q) o[q] = rd(k.s + q);
    return o + k.len;
}

enum { ERR_COLUMNS = 1, ERR_INT = 2, ERR_RANGE = 3 };

struct LineOut {
    i32 *id1, *pos1, *id2, *pos2;
    i64 *bed_len;
    unsigned long long *err;
    i64 *pos1w, *pos2w;                     // non-null: wide mode, the positions go here as int64
};

template <class RD>
__device__ __forceinline__ void parse_one(const RD &rd, i64 k, i64 a, i64 e, const NameTable &T, const LineOut &O) {
    Tok tok[5];
    i32 o1 = -1, o2 = -1, q1 = 0, q2 = 0;
    i64 blen = 0, w1 = 0, w2 = 0;
    const int nt = rd(a) == '#' ? -1 : first_tokens(rd, a, e, tok);
    if (nt > 0) {                                                // nt == 0: blank, nt == -1: header
        i64 v1 = 0, v2 = 0;
        int r1 = 0, r2 = 0;
        const int limit = O.pos1w ? 1 : 0;                       // wide mode accepts what does not fit int32
        if (nt < 5) atomicMin(O.err, ((unsigned long long)k << 8) | ERR_COLUMNS);
        else if (!parse_int(rd, tok[2], &v1, &r1) || !parse_int(rd, tok[4], &v2, &r2)) atomicMin(O.err, ((unsigned long long)k << 8) | ERR_INT);
        else if (r1 > limit || r2 > limit) atomicMin(O.err, ((unsigned long long)k << 8) | ERR_RANGE);
        else {
            o1 = lookup(T, rd, tok[1]);
            o2 = lookup(T, rd, tok[3]);
            q1 = (i32)(v1 - 1);
            q2 = (i32)(v2 - 1);
            w1 = v1 - 1;
            w2 = v2 - 1;
            if (O.bed_len) blen = tok[1].len + tok[3].len + 2 * (i64)tok[0].len + 2 * dec_len(v1 - 1) + 2 * dec_len(v2 - 1) + 24;
        }
    }
    O.id1[k] = o1; O.id2[k] = o2;
    if (O.pos1w) { O.pos1w[k] = w1; O.pos2w[k] = w2; }
    else { O.pos1[k] = q1; O.pos2[k] = q2; }
    if (O.bed_len) O.bed_len[k] = blen;
}

template <class RD>
__device__ __forceinline__ void bed_one(const RD &rd, i64 a, i64 e, i64 p1, i64 p2, unsigned char *o) {
    Tok tok[5];
    first_tokens(rd, a, e, tok);
    for (int side = 0; side < 2; ++side) {
        const i64 p = side ? p2 : p1;
        o = put_tok(o, rd, tok[side ? 3 : 1]); *o++ = '\t';
        o = put_dec(o, p); *o++ = '\t';
        o = put_dec(o, p); *o++ = '\t';
        o = put_tok(o, rd, tok[0]);
        *o++ = '/'; *o++ = side ? '2' : '1'; *o++ = '\t'; *o++ = '2'; *o++ = '5'; *o++ = '5'; *o++ = '\t'; *o++ = '.'; *o++ = '\n';
    }
}

// the text of lines [k0, k1) -> LDS, 16 bytes per lane per step; LDS offset == position - bias with bias 16-aligned,
// so that every 16-byte step is aligned on both sides.  Returns false when the span does not fit (or the text is
// not 16-byte aligned): the caller reads HBM directly.
__device__ __forceinline__ bool stage_lines(const unsigned char *__restrict__ t, i64 n, i64 a0, i64 e0, unsigned char *lds, i64 *bias) {
    const i64 base = a0 & ~(i64)15;
    *bias = base;
    if (e0 - base > IN_CAP || ((uintptr_t)t & 15)) return false;
    for (i64 o = (i64)threadIdx.x * 16; base + o < e0; o += (i64)blockDim.x * 16) {
        if (base + o + 16 <= n) *reinterpret_cast<uint4 *>(lds + o) = *reinterpret_cast<const uint4 *>(t + base + o);
        else for (int q = 0; base + o + q < n; ++q) lds[o + q] = t[base + o + q];
    }
    return true;
}

// STAGED = true: the blocks of 128 lines whose text fits the LDS window; the others (lines of kilobytes, or an
// unaligned device buffer) only raise O.err[1] and are taken by a second launch with STAGED = false, which reads HBM
// directly — so that the common kernel carries one reader, not two (registers, instruction cache).
template <bool STAGED>
__global__ __launch_bounds__(LN_BLOCK) void k_parse_lines(const unsigned char *__restrict__ t, i64 n, const i64 *__restrict__ starts, i64 n_lines,
                                                          NameTable T, LineOut O) {
    __shared__ __attribute__((aligned(16))) unsigned char s_in[STAGED ? IN_CAP + 16 : 16];
    const bool aligned = ((uintptr_t)t & 15) == 0;
    for (i64 k0 = (i64)blockIdx.x * LN_BLOCK; k0 < n_lines; k0 += (i64)gridDim.x * LN_BLOCK) {
        const i64 k1 = k0 + LN_BLOCK < n_lines ? k0 + LN_BLOCK : n_lines;
        const i64 a0 = starts[k0], e0 = k1 < n_lines ? starts[k1] : n;
        const bool fits = aligned && e0 - (a0 & ~(i64)15) <= IN_CAP;
        const i64 k = k0 + threadIdx.x;
        if (STAGED) {
            if (!fits) { if (threadIdx.x == 0) atomicOr(&O.err[1], 1ull); continue; }
            i64 bias;
            (void)stage_lines(t, n, a0, e0, s_in, &bias);
            __syncthreads();
            if (k < k1) parse_one(LdsText{s_in, bias}, k, starts[k], k + 1 < n_lines ? starts[k + 1] : n, T, O);
            __syncthreads();
        } else {
            if (fits) continue;
            if (k < k1) parse_one(HbmText{t}, k, starts[k], k + 1 < n_lines ? starts[k + 1] : n, T, O);
        }
    }
}

// BED records of a block of lines are formatted into LDS at (offset - out_bias), out_bias chosen so that LDS and
// HBM addresses agree mod 16, then leave with 16-byte stores (bytes at the two ragged ends)
__global__ __launch_bounds__(LN_BLOCK) void k_bed_write(const unsigned char *__restrict__ t, i64 n, const i64 *__restrict__ starts, i64 n_lines,
                                                        const i32 *__restrict__ pos1, const i32 *__restrict__ pos2, const i64 *__restrict__ pos1w,
                                                        const i64 *__restrict__ pos2w, const i64 *__restrict__ bed_off, unsigned char *__restrict__ bed) {
    __shared__ __attribute__((aligned(16))) unsigned char s_in[IN_CAP + 16];
    __shared__ __attribute__((aligned(16))) unsigned char s_out[OUT_CAP];
    for (i64 k0 = (i64)blockIdx.x * LN_BLOCK; k0 < n_lines; k0 += (i64)gridDim.x * LN_BLOCK) {
        const i64 k1 = k0 + LN_BLOCK < n_lines ? k0 + LN_BLOCK : n_lines;
        const i64 b0 = bed_off[k0], b1 = bed_off[k1];
        if (b1 == b0) continue;                                  // uniform across the block
        const i64 a0 = starts[k0], e0 = k1 < n_lines ? starts[k1] : n;
        i64 bias;
        const bool staged = stage_lines(t, n, a0, e0, s_in, &bias);
        const i64 out_bias = b0 & ~(i64)15;                      // bed is 16-byte aligned (pool allocation)
        const bool out_staged = b1 - out_bias <= OUT_CAP;
        __syncthreads();
        const i64 k = k0 + threadIdx.x;
        if (k < k1 && bed_off[k + 1] > bed_off[k]) {
            const i64 a = starts[k], e = k + 1 < n_lines ? starts[k + 1] : n;
            unsigned char *o = out_staged ? s_out + (bed_off[k] - out_bias) : bed + bed_off[k];
            const i64 p1 = pos1w ? pos1w[k] : (i64)pos1[k], p2 = pos2w ? pos2w[k] : (i64)pos2[k];
            if (staged) bed_one(LdsText{s_in, bias}, a, e, p1, p2, o);
            else bed_one(HbmText{t}, a, e, p1, p2, o);
        }
        __syncthreads();
        if (out_staged) {
            for (i64 o = (i64)threadIdx.x * 16; out_bias + o < b1; o += (i64)blockDim.x * 16) {
                const i64 g = out_bias + o;
                if (g >= b0 && g + 16 <= b1) *reinterpret_cast<uint4 *>(bed + g) = *reinterpret_cast<const uint4 *>(s_out + o);
                else for (int q = 0; q < 16; ++q) if (g + q >= b0 && g + q < b1) bed[g + q] = s_out[o + q];
            }
            __syncthreads();
        }
    }
}

unsigned grid_for(i64 work, int per_block) { return (unsigned)std::max<i64>(1, std::min<i64>((work + per_block - 1) / per_block, 256 * 32)); }

}  // namespace

extern "C" int hhx_pairs_parser_create(i32 n_names, const uint8_t *names, const i64 *name_off, hhx_pairs_parser **out) {
    if (n_names < 0 || !out || (n_names && (!names || !name_off))) return fail("hhx_pairs_parser_create: bad argument");
    auto *p = new hhx_pairs_parser();
    p->n_names = n_names;
    u32 cap = 16;
    while (cap < 2u * (u32)n_names + 2) cap <<= 1;
    p->mask = cap - 1;
    std::vector<u64> sh(cap, 0);
    std::vector<i32> si(cap, -1), len((size_t)n_names + 1, 0);
    std::vector<i64> off((size_t)n_names + 1, 0);
    for (i32 k = 0; k < n_names; ++k) { len[k] = (i32)(name_off[k + 1] - name_off[k]); off[k + 1] = off[k] + (len[k] + 7) / 8; }
    std::vector<u64> words((size_t)off[n_names] + 1, 0);
    for (i32 k = 0; k < n_names; ++k) {
        if (len[k]) memcpy(&words[(size_t)off[k]], names + name_off[k], (size_t)len[k]);
        u64 h = HASH_SEED;
        for (i64 q = off[k]; q < off[k + 1]; ++q) h = hash_step(h, words[(size_t)q]);
        u32 s = (u32)h & p->mask;
        while (si[s] >= 0) s = (s + 1) & p->mask;
        si[s] = k;
        sh[s] = h;
    }
    if (p->names.alloc(words.size()) || p->name_off.alloc(off.size()) || p->name_len.alloc(len.size()) || p->slot_hash.alloc(cap) ||
        p->slot_id.alloc(cap) || p->err.alloc(2)) { delete p; return 1; }
    hipError_t e = hipMemcpyAsync(p->names.p, words.data(), sizeof(u64) * words.size(), hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->name_off.p, off.data(), sizeof(i64) * off.size(), hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->name_len.p, len.data(), sizeof(i32) * len.size(), hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->slot_hash.p, sh.data(), sizeof(u64) * cap, hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->slot_id.p, si.data(), sizeof(i32) * cap, hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { delete p; return fail("hhx_pairs_parser_create: %s", hipGetErrorString(e)); }
    *out = p;
    return 0;
}

extern "C" int hhx_pairs_parser_destroy(hhx_pairs_parser *p) {
    delete p;
    return 0;
}

extern "C" int hhx_pairs_parse(hhx_pairs_parser *p, const uint8_t *text, i64 n_bytes, int on_device, int want_bed, i64 *n_lines, i64 *bed_bytes) {
    if (!p || n_bytes < 0 || (n_bytes && !text)) return fail("hhx_pairs_parse: bad argument");
    p->lines_before += p->n_lines;
    p->n_lines = 0;
    p->bed_bytes = 0;
    if (n_lines) *n_lines = 0;
    if (bed_bytes) *bed_bytes = 0;
    if (n_bytes == 0) return 0;
    const unsigned char *t = text;
    if (!on_device) {
        if (p->text.n < (size_t)n_bytes + 16 && p->text.alloc((size_t)n_bytes + 16)) return 1;
        HHX_HIP(hipMemcpyAsync(p->text.p, text, (size_t)n_bytes, hipMemcpyHostToDevice, g_stream));
        t = p->text.p;
    }
    const i64 n_blocks = (n_bytes + TX_BLOCK - 1) / TX_BLOCK;
    DevBuf<i64> cnt, pre;
    if (cnt.alloc((size_t)n_blocks) || pre.alloc((size_t)n_blocks + 1)) return 1;
    { KTimer kt("text_breaks");
    k_count_breaks<<<grid_for(n_blocks, 1), 256, 0, g_stream>>>(t, n_bytes, n_blocks, cnt.p); }
    HHX_LAUNCH_CHECK();
    i64 n_breaks = 0;
    HHX_TRY(exclusive_scan_i64(cnt.p, pre.p, n_blocks, &n_breaks));
    unsigned char last = 0;
    HHX_HIP(hipMemcpyAsync(&last, t + n_bytes - 1, 1, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    const i64 nl = n_breaks + ((last == '\n' || last == '\r') ? 0 : 1);
    if (p->starts.n < (size_t)n_breaks + 2 && p->starts.alloc((size_t)n_breaks + 2)) return 1;
    { KTimer kt("text_starts");
    k_write_starts<<<grid_for(n_blocks, 1), 256, 0, g_stream>>>(t, n_bytes, n_blocks, pre.p, p->starts.p); }
    HHX_LAUNCH_CHECK();
    if (std::min(std::min(p->id1.n, p->pos1.n), std::min(p->id2.n, p->pos2.n)) < (size_t)nl)
        if (p->id1.alloc((size_t)nl) || p->pos1.alloc((size_t)nl) || p->id2.alloc((size_t)nl) || p->pos2.alloc((size_t)nl)) return 1;
    if (p->wide && std::min(p->pos1w.n, p->pos2w.n) < (size_t)nl && (p->pos1w.alloc((size_t)nl) || p->pos2w.alloc((size_t)nl))) return 1;
    DevBuf<i64> bed_len;
    if (want_bed && bed_len.alloc((size_t)nl)) return 1;
    HHX_HIP(hipMemsetAsync(p->err.p, 0xff, sizeof(unsigned long long), g_stream));      // [0] first error (min), [1] unstaged blocks seen
    HHX_HIP(hipMemsetAsync(p->err.p + 1, 0, sizeof(unsigned long long), g_stream));
    const NameTable T{p->names.p, p->name_off.p, p->name_len.p, p->slot_hash.p, p->slot_id.p, p->mask};
    const LineOut O{p->id1.p, p->pos1.p, p->id2.p, p->pos2.p, want_bed ? bed_len.p : nullptr, p->err.p, p->wide ? p->pos1w.p : nullptr,
                    p->wide ? p->pos2w.p : nullptr};
    { KTimer kt("text_parse");
    k_parse_lines<true><<<grid_for(nl, LN_BLOCK), LN_BLOCK, 0, g_stream>>>(t, n_bytes, p->starts.p, nl, T, O); }
    HHX_LAUNCH_CHECK();
    unsigned long long errw[2] = {0, 0};
    HHX_HIP(hipMemcpyAsync(errw, p->err.p, sizeof errw, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    if (errw[1]) {                                           // some block of lines did not fit LDS
        k_parse_lines<false><<<grid_for(nl, LN_BLOCK), LN_BLOCK, 0, g_stream>>>(t, n_bytes, p->starts.p, nl, T, O);
        HHX_LAUNCH_CHECK();
        HHX_HIP(hipMemcpyAsync(errw, p->err.p, sizeof errw, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
    }
    const unsigned long long err = errw[0];
    if (err != ~0ull) {
        const long long line = (long long)(p->lines_before + (i64)(err >> 8) + 1);
        switch ((int)(err & 0xff)) {
            case ERR_COLUMNS: return fail("IndexError: .pairs line %lld has fewer than 5 columns", line);
            case ERR_INT: return fail("ValueError: .pairs line %lld: invalid position literal", line);
            default: return fail(p->wide ? "ValueError: .pairs line %lld: position beyond 2^40" : "ValueError: .pairs line %lld: position outside the int32 range", line);
        }
    }
    p->n_lines = nl;
    if (n_lines) *n_lines = nl;
    if (want_bed) {
        if (p->bed_off.n < (size_t)nl + 1 && p->bed_off.alloc((size_t)nl + 1)) return 1;
        i64 total = 0;
        HHX_TRY(exclusive_scan_i64(bed_len.p, p->bed_off.p, nl, &total));
        unsigned char *dst = nullptr;
        if (p->sink) { void *room = nullptr; HHX_TRY(hhx_byte_sink_reserve(p->sink, total + 16, &room)); dst = (unsigned char *)room; }
        else { if (p->bed.n < (size_t)total + 16 && p->bed.alloc((size_t)total + 16)) return 1; dst = p->bed.p; }
        { KTimer kt("text_bed");
        k_bed_write<<<grid_for(nl, LN_BLOCK), LN_BLOCK, 0, g_stream>>>(t, n_bytes, p->starts.p, nl, p->pos1.p, p->pos2.p, p->wide ? p->pos1w.p : nullptr,
                                                                          p->wide ? p->pos2w.p : nullptr, p->bed_off.p, dst); }
        HHX_LAUNCH_CHECK();
        if (p->sink) HHX_TRY(hhx_byte_sink_commit(p->sink, dst, total));
        p->bed_bytes = total;
        if (bed_bytes) *bed_bytes = total;
    }
    return 0;
}

extern "C" int hhx_pairs_parser_arrays(hhx_pairs_parser *p, void **id1, void **pos1, void **id2, void **pos2, void **bed) {
    if (!p) return fail("null parser");
    if (id1) *id1 = p->id1.p;
    if (pos1) *pos1 = p->wide ? (void *)p->pos1w.p : (void *)p->pos1.p;        // int64 arrays in wide mode
    if (id2) *id2 = p->id2.p;
    if (pos2) *pos2 = p->wide ? (void *)p->pos2w.p : (void *)p->pos2.p;
    if (bed) *bed = p->bed.p;
    return 0;
}

extern "C" int hhx_pairs_parser_bed_host(hhx_pairs_parser *p, void **host, i64 *n_bytes) {
    if (!p || !host || !n_bytes) return fail("null pointer");
    const int k = p->pin_next;
    p->pin_next ^= 1;
    const size_t n = (size_t)p->bed_bytes;
    if (n > p->pin_cap[k]) {
        if (p->pin[k]) (void)hipHostFree(p->pin[k]);
        p->pin[k] = nullptr;
        p->pin_cap[k] = n + n / 4 + 4096;
        HHX_HIP(hipHostMalloc((void **)&p->pin[k], p->pin_cap[k], hipHostMallocDefault));
    }
    if (n) {
        HHX_HIP(hipMemcpyAsync(p->pin[k], p->bed.p, n, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
    }
    *host = p->pin[k];
    *n_bytes = (i64)n;
    return 0;
}

// alignments.bed into a byte sink (hhx_jobs.hip) from the next parse on: the records are formatted straight into the sink's ring in HBM and
// queued for the file-writer thread; nullptr: back to the parser's own buffer (hhx_pairs_parser_bed_host / _fetch)
extern "C" int hhx_pairs_parser_set_bed_sink(hhx_pairs_parser *p, hhx_byte_sink *sink) {
    if (!p) return fail("null parser");
    p->sink = sink;
    return 0;
}

// positions as 64-bit integers from the next parse on (contigs beyond 2^31 bp: determine_int_type :116-147 picks int64 there);
// hhx_pairs_parser_arrays then hands out int64 position arrays (for hhx_ingest_push64) and hhx_pairs_parser_fetch64 copies them
extern "C" int hhx_pairs_parser_set_wide(hhx_pairs_parser *p, int on) {
    if (!p) return fail("null parser");
    p->wide = on != 0;
    return 0;
}
extern "C" int hhx_pairs_parser_fetch64(hhx_pairs_parser *p, i32 *id1, i64 *pos1, i32 *id2, i64 *pos2, uint8_t *bed) {
    if (!p) return fail("null parser");
    if (!p->wide) return fail("hhx_pairs_parser_fetch64: the parser is not in wide mode");
    const size_t n = (size_t)p->n_lines;
    if (n) {
        if (id1) HHX_HIP(hipMemcpyAsync(id1, p->id1.p, sizeof(i32) * n, hipMemcpyDeviceToHost, g_stream));
        if (pos1) HHX_HIP(hipMemcpyAsync(pos1, p->pos1w.p, sizeof(i64) * n, hipMemcpyDeviceToHost, g_stream));
        if (id2) HHX_HIP(hipMemcpyAsync(id2, p->id2.p, sizeof(i32) * n, hipMemcpyDeviceToHost, g_stream));
        if (pos2) HHX_HIP(hipMemcpyAsync(pos2, p->pos2w.p, sizeof(i64) * n, hipMemcpyDeviceToHost, g_stream));
    }
    if (bed && p->bed_bytes) HHX_HIP(hipMemcpyAsync(bed, p->bed.p, (size_t)p->bed_bytes, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
extern "C" int hhx_pairs_parser_fetch(hhx_pairs_parser *p, i32 *id1, i32 *pos1, i32 *id2, i32 *pos2, uint8_t *bed) {
    if (!p) return fail("null parser");
    if (p->wide && (pos1 || pos2)) return fail("hhx_pairs_parser_fetch: wide mode holds int64 positions (hhx_pairs_parser_fetch64)");
    const size_t nb = sizeof(i32) * (size_t)p->n_lines;
    if (nb) {
        if (id1) HHX_HIP(hipMemcpyAsync(id1, p->id1.p, nb, hipMemcpyDeviceToHost, g_stream));
        if (pos1) HHX_HIP(hipMemcpyAsync(pos1, p->pos1.p, nb, hipMemcpyDeviceToHost, g_stream));
        if (id2) HHX_HIP(hipMemcpyAsync(id2, p->id2.p, nb, hipMemcpyDeviceToHost, g_stream));
        if (pos2) HHX_HIP(hipMemcpyAsync(pos2, p->pos2.p, nb, hipMemcpyDeviceToHost, g_stream));
    }
    if (bed && p->bed_bytes) HHX_HIP(hipMemcpyAsync(bed, p->bed.p, (size_t)p->bed_bytes, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

// ---------------------------------------------------------------- id / position arrays -> .pairs text (synthetic inputs)
// The writer counterpart of hhx_pairs_parse, for measurement only: SURVEY §8(d) quotes C3 / C5 on read pairs "written as .pairs text", and
// formatting 5e8 lines on the host is not an option.  Line k = "r{first_read + k}\t{names[id1]}\t{pos1 + 1}\t{names[id2]}\t{pos2 + 1}\t+\t-\n"
// (the format tools/c1_run.py and the reference's simulation write; positions 1-based in the file, :1556).  One thread per line, two passes.
namespace {
__device__ __forceinline__ int dec_len(u64 a) { int n = 0; do { ++n; a /= 10; } while (a); return n; }
__device__ __forceinline__ unsigned char *dec_put(unsigned char *o, u64 a) {
    const int n = dec_len(a);
    for (int k = n - 1; k >= 0; --k) { o[k] = (unsigned char)('0' + a % 10); a /= 10; }
    return o + n;
}
__global__ __launch_bounds__(256) void k_pairs_line_len(i64 n, const i32 *__restrict__ id1, const i32 *__restrict__ pos1, const i32 *__restrict__ id2,
                                                        const i32 *__restrict__ pos2, const i32 *__restrict__ name_len, i64 first_read, i64 *__restrict__ len) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x)
        len[k] = 1 + dec_len((u64)(first_read + k)) + 1 + name_len[id1[k]] + 1 + dec_len((u64)pos1[k] + 1) + 1 + name_len[id2[k]] + 1 + dec_len((u64)pos2[k] + 1) + 5;
}
__global__ __launch_bounds__(256) void k_pairs_line_write(i64 n, const i32 *__restrict__ id1, const i32 *__restrict__ pos1, const i32 *__restrict__ id2,
                                                          const i32 *__restrict__ pos2, const u64 *__restrict__ names, const i64 *__restrict__ name_off,
                                                          const i32 *__restrict__ name_len, i64 first_read, const i64 *__restrict__ off,
                                                          unsigned char *__restrict__ text) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) {
        unsigned char *o = text + off[k];
        *o++ = 'r';
        o = dec_put(o, (u64)(first_read + k));
        *o++ = '\t';
        const i32 ids[2] = {id1[k], id2[k]};
        const i32 ps[2] = {pos1[k], pos2[k]};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned char *nm = reinterpret_cast<const unsigned char *>(names + name_off[ids[s]]);
            for (i32 q = 0; q < name_len[ids[s]]; ++q) *o++ = nm[q];
            *o++ = '\t';
            o = dec_put(o, (u64)ps[s] + 1);
            *o++ = '\t';
        }
        *o++ = '+'; *o++ = '\t'; *o++ = '-'; *o++ = '\n';
    }
}
}  // namespace

extern "C" int hhx_pairs_format(hhx_pairs_parser *p, int64_t n, const int32_t *dev_id1, const int32_t *dev_pos1, const int32_t *dev_id2, const int32_t *dev_pos2,
                                int64_t first_read, uint8_t *dev_text, int64_t capacity, int64_t *n_bytes) {
    if (!p || !n_bytes || n < 0 || (n && (!dev_id1 || !dev_pos1 || !dev_id2 || !dev_pos2))) return fail("hhx_pairs_format: bad argument");
    *n_bytes = 0;
    if (n == 0) return 0;
    DevBuf<i64> len, off;
    if (len.alloc((size_t)n + 1) || off.alloc((size_t)n + 2)) return 1;
    k_pairs_line_len<<<grid_for(n, 256), 256, 0, g_stream>>>(n, dev_id1, dev_pos1, dev_id2, dev_pos2, p->name_len.p, first_read, len.p);
    HHX_LAUNCH_CHECK();
    i64 total = 0;
    HHX_TRY(exclusive_scan_i64(len.p, off.p, n, &total));
    *n_bytes = total;
    if (!dev_text) return 0;
    if (capacity < total) return fail("hhx_pairs_format: %lld bytes of text, buffer of %lld", (long long)total, (long long)capacity);
    k_pairs_line_write<<<grid_for(n, 256), 256, 0, g_stream>>>(n, dev_id1, dev_pos1, dev_id2, dev_pos2, p->names.p, p->name_off.p, p->name_len.p, first_read, off.p, dev_text);
    HHX_LAUNCH_CHECK();
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
