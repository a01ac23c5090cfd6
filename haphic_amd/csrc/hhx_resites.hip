// Restriction-site counting: count_RE_sites (scripts/HapHiC_cluster.py:75-84) for all the segments that
// parse_fasta (:87-113) and stat_fragments (:188-296) need — whole contigs, bins, and the flank prefix /
// suffix of either — in one pass over the genome bytes.
//
//   reference:  RE_sites = sum over sites of seq[a:b].count(site)      (Python str.count: NON-overlapping,
//               leftmost-first; sites after parse_RE_sites' N expansion :56-72)
//
// A site that has no border (no proper prefix equal to a suffix — GATC, AAGCTT, GAATC, ... every common
// enzyme) cannot overlap itself, so its count in [a, b) is the number of match starts p with
// a <= p <= b - len: a difference of two prefix counts.  The genome is swept once per distinct site length:
// match starts are counted per 4096-byte block (coalesced byte stream, HBM-bound), the block counts are
// scanned, and every segment end point is resolved with one partial-block recount by a wavefront.  Segments may
// overlap arbitrarily (flank prefix and suffix of the same contig) at no extra cost.
// Sites WITH a border (e.g. GCGC, AAAA) take the exact sequential greedy scan, one thread per (segment, site).
#include "hhx_common.h"

using namespace hhx;

namespace {

constexpr int RS_BLOCK = 4096, RS_MAX_SITES = 64, RS_MAX_LEN = 32;

struct SiteSet {
    int n;                                  // sites of one length (border-free)
    int len;
    unsigned char pat[RS_MAX_SITES][RS_MAX_LEN];
};

__device__ __forceinline__ int matches_at(const unsigned char *__restrict__ seq, i64 p, i64 n, const SiteSet &S) {
    if (p + S.len > n) return 0;
    int c = 0;
    for (int s = 0; s < S.n; ++s) {
        bool ok = true;
        for (int k = 0; k < S.len; ++k) ok &= seq[p + k] == S.pat[s][k];
        c += ok;
    }
    return c;
}

// match starts inside every 4096-byte block
__global__ __launch_bounds__(256) void k_block_counts(const unsigned char *__restrict__ seq, i64 n, SiteSet S, i64 n_blocks,
                                                      i64 *__restrict__ counts) {
    __shared__ i32 wsum[4];
    for (i64 b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const i64 base = b * RS_BLOCK;
        i32 c = 0;
        for (int k = threadIdx.x; k < RS_BLOCK; k += 256)
            if (base + k < n) c += matches_at(seq, base + k, n, S);
        c = wave_sum_i32(c);
        if (lane_id() == 0) wsum[threadIdx.x / HHX_WAVE] = c;
        __syncthreads();
        if (threadIdx.x == 0) counts[b] = (i64)wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

// prefix(x) = number of match starts p < x; one wave per query point, two points per segment
__global__ __launch_bounds__(256) void k_segment_counts(const unsigned char *__restrict__ seq, i64 n, SiteSet S,
                                                        const i64 *__restrict__ block_prefix, i64 n_seg,
                                                        const i64 *__restrict__ seg_off, const i64 *__restrict__ seg_len,
                                                        i64 *__restrict__ out) {
    const int lane = lane_id();
    for (i64 q = (i64)blockIdx.x * 4 + threadIdx.x / HHX_WAVE; q < 2 * n_seg; q += (i64)gridDim.x * 4) {
        const i64 s = q >> 1;
        const i64 a = seg_off[s], len = seg_len[s];
        if (len < S.len) continue;                              // the slice is shorter than the site
        // counted starts: a <= p <= a + len - S.len  ->  prefix(a + len - S.len + 1) - prefix(a)
        const i64 x = (q & 1) ? a + len - S.len + 1 : a;
        const i64 b = x / RS_BLOCK;
        i32 c = 0;
        for (i64 p = b * RS_BLOCK + lane; p < x; p += HHX_WAVE) c += matches_at(seq, p, n, S);
        c = wave_sum_i32(c);
        if (lane == 0) {
            const i64 v = block_prefix[b] + c;
            atomicAdd((unsigned long long *)&out[s], (unsigned long long)((q & 1) ? v : -v));
        }
    }
}

// exact greedy scan for sites that can overlap themselves: one thread per (segment, site)
__global__ __launch_bounds__(256) void k_greedy_counts(const unsigned char *__restrict__ seq, i64 n_seg, const i64 *__restrict__ seg_off,
                                                       const i64 *__restrict__ seg_len, int n_sites, const unsigned char *__restrict__ pats,
                                                       const i32 *__restrict__ pat_len, i64 *__restrict__ out) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n_seg * n_sites; t += (i64)gridDim.x * blockDim.x) {
        const i64 s = t / n_sites;
        const int k = (int)(t % n_sites);
        const i32 L = pat_len[k];
        const unsigned char *pat = pats + (size_t)k * RS_MAX_LEN;
        const i64 a = seg_off[s], e = a + seg_len[s];
        i64 c = 0;
        for (i64 p = a; p + L <= e;) {
            bool ok = true;
            for (int j = 0; j < L && ok; ++j) ok = seq[p + j] == pat[j];
            if (ok) { ++c; p += L; } else ++p;
        }
        if (c) atomicAdd((unsigned long long *)&out[s], (unsigned long long)c);
    }
}

bool has_border(const unsigned char *p, int len) {
    for (int b = 1; b < len; ++b)
        if (memcmp(p, p + len - b, (size_t)b) == 0) return true;
    return false;
}

}  // namespace

extern "C" int hhx_count_re_sites(const uint8_t *seq_host, i64 seq_len, i64 n_seg, const i64 *seg_off, const i64 *seg_len,
                                  i32 n_sites, const uint8_t *sites, const i32 *site_len, i64 *counts_host) {
    if (seq_len < 0 || n_seg < 0 || n_sites < 0 || (n_seg && (!seg_off || !seg_len || !counts_host))) return fail("hhx_count_re_sites: bad argument");
    if (n_seg == 0) return 0;
    for (i64 s = 0; s < n_seg; ++s)
        if (seg_off[s] < 0 || seg_len[s] < 0 || seg_off[s] + seg_len[s] > seq_len) return fail("hhx_count_re_sites: segment %lld out of range", (long long)s);
    // split the sites: border-free ones grouped by length, the rest to the greedy kernel
    std::vector<SiteSet> sets;
    std::vector<unsigned char> gpat;
    std::vector<i32> glen;
    size_t off = 0;
    for (i32 k = 0; k < n_sites; ++k) {
        const i32 L = site_len[k];
        if (L <= 0 || L > RS_MAX_LEN) return fail("hhx_count_re_sites: site length %d not in [1, %d]", L, RS_MAX_LEN);
        const unsigned char *p = sites + off;
        off += (size_t)L;
        if (has_border(p, L)) {
            gpat.resize(gpat.size() + RS_MAX_LEN, 0);
            memcpy(gpat.data() + gpat.size() - RS_MAX_LEN, p, (size_t)L);
            glen.push_back(L);
            continue;
        }
        SiteSet *dst = nullptr;
        for (auto &S : sets)
            if (S.len == L && S.n < RS_MAX_SITES) dst = &S;
        if (!dst) { sets.emplace_back(); dst = &sets.back(); dst->n = 0; dst->len = L; memset(dst->pat, 0, sizeof dst->pat); }
        memcpy(dst->pat[dst->n++], p, (size_t)L);
    }
    DevBuf<unsigned char> seq;
    DevBuf<i64> d_off, d_len, d_out;
    if (seq.alloc((size_t)seq_len + 1) || d_off.alloc((size_t)n_seg) || d_len.alloc((size_t)n_seg) || d_out.alloc((size_t)n_seg)) return 1;
    if (seq_len) HHX_HIP(hipMemcpyAsync(seq.p, seq_host, (size_t)seq_len, hipMemcpyHostToDevice, g_stream));
    HHX_HIP(hipMemcpyAsync(d_off.p, seg_off, sizeof(i64) * (size_t)n_seg, hipMemcpyHostToDevice, g_stream));
    HHX_HIP(hipMemcpyAsync(d_len.p, seg_len, sizeof(i64) * (size_t)n_seg, hipMemcpyHostToDevice, g_stream));
    HHX_HIP(hipMemsetAsync(d_out.p, 0, sizeof(i64) * (size_t)n_seg, g_stream));
    const i64 n_blocks = seq_len / RS_BLOCK + 1;
    for (const SiteSet &S : sets) {
        DevBuf<i64> cnt, pre;
        if (cnt.alloc((size_t)n_blocks + 1) || pre.alloc((size_t)n_blocks + 2)) return 1;
        { KTimer kt("re_block_counts");
        k_block_counts<<<(unsigned)std::min<i64>(n_blocks, 256 * 16), 256, 0, g_stream>>>(seq.p, seq_len, S, n_blocks, cnt.p); }
        HHX_LAUNCH_CHECK();
        HHX_TRY(exclusive_scan_i64(cnt.p, pre.p, n_blocks, nullptr));
        k_segment_counts<<<(unsigned)std::max<i64>(1, std::min<i64>((2 * n_seg + 3) / 4, 256 * 16)), 256, 0, g_stream>>>(
            seq.p, seq_len, S, pre.p, n_seg, d_off.p, d_len.p, d_out.p);
        HHX_LAUNCH_CHECK();
        HHX_HIP(hipStreamSynchronize(g_stream));                 // cnt / pre die here
    }
    if (!glen.empty()) {
        DevBuf<unsigned char> d_pat;
        DevBuf<i32> d_plen;
        if (d_pat.alloc(gpat.size()) || d_plen.alloc(glen.size())) return 1;
        HHX_HIP(hipMemcpyAsync(d_pat.p, gpat.data(), gpat.size(), hipMemcpyHostToDevice, g_stream));
        HHX_HIP(hipMemcpyAsync(d_plen.p, glen.data(), sizeof(i32) * glen.size(), hipMemcpyHostToDevice, g_stream));
        const i64 work = n_seg * (i64)glen.size();
        k_greedy_counts<<<(unsigned)std::max<i64>(1, std::min<i64>((work + 255) / 256, 256 * 16)), 256, 0, g_stream>>>(
            seq.p, n_seg, d_off.p, d_len.p, (int)glen.size(), d_pat.p, d_plen.p, d_out.p);
        HHX_LAUNCH_CHECK();
        HHX_HIP(hipStreamSynchronize(g_stream));
    }
    HHX_HIP(hipMemcpyAsync(counts_host, d_out.p, sizeof(i64) * (size_t)n_seg, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
