// BAM front end: BGZF container -> BAM records -> the (contig id, position) arrays hhx_ingest_push
// takes.  Reference: scripts/HapHiC_cluster.py bam_generator :1586-1593 — pysam.AlignmentFile(..., format_options =
// [b'filter=flag.read1'] or [b'filter=flag.read1 && refid != mrefid'] :2837 :2855 :2862) yielding
// (reference_name, next_reference_name, reference_start, next_reference_start) per record that passes the filter.
//
// Split of the work:
//   host   BGZF blocks are independent raw-DEFLATE members (<= 64 KiB each): they are inflated by a pool of threads
//          (zlib) straight into one pinned buffer at the prefix sums of their ISIZE fields; one sequential walk over the
//          inflated bytes reads every record's block_size and collects the record offsets (a record's start is only
//          known from its predecessor's length).  A record cut by the end of a batch is carried into the next batch.
//   device the inflated bytes and the offsets go to HBM; k_bam_decode gathers refID / pos / flag / next_refID /
//          next_pos of every record, applies the htslib filter expression, maps BAM reference ids to contig ids of
//          the FASTA (-1 = name not in fa_dict, or unmapped: dropped by the ingest exactly like `ref not in fa_dict`
//          :1610 / :1702) and writes the four int32 arrays; record k of the batch = element k (filtered records carry
//          id -2), so stream order — which decides dict insertion order downstream — is preserved.
// BAM layout (SAM spec §4.2), little endian, offsets from the record's block_size field: refID 4, pos 8, flag 18,
// next_refID 24, next_pos 28.
#include "hhx_bgzf.h"

using namespace hhx;

struct hhx_bam {
    FILE *f = nullptr;
    int threads = 1;
    bool eof = false;
    std::vector<unsigned char> comp;        // compressed bytes not yet consumed (whole blocks + a partial tail)
    std::vector<unsigned char> carry;       // inflated bytes of an incomplete record (or header) from the previous batch
    std::string header_text;
    std::vector<std::string> ref_names;
    std::vector<i32> ref_len;
    std::string names_cat;                  // reference names, concatenated; name_off[n_ref + 1]
    std::vector<i64> name_off;
    i64 records_total = 0, last_n = 0;
    // batch buffers
    unsigned char *pin = nullptr; size_t pin_cap = 0;
    DevBuf<unsigned char> d_bytes;
    DevBuf<i64> d_off;
    DevBuf<i32> d_map, d_id1, d_pos1, d_id2, d_pos2;
    i32 map_n = -1;
    ~hhx_bam() {
        if (f) fclose(f);
        if (pin) (void)hipHostFree(pin);
    }
};

namespace {

int refill(hhx_bam *b, size_t want) {
    while (!b->eof && b->comp.size() < want) {
        const size_t old = b->comp.size(), chunk = std::max<size_t>(want - old, (size_t)1 << 20);
        b->comp.resize(old + chunk);
        const size_t got = fread(b->comp.data() + old, 1, chunk, b->f);
        b->comp.resize(old + got);
        if (got < chunk) { if (ferror(b->f)) return fail("BAM: read error"); b->eof = true; }
    }
    return 0;
}

// more inflated bytes appended to b->carry (used while the header is being read)
int pull_into_carry(hhx_bam *b, size_t inflated_target) {
    HHX_TRY(refill(b, (size_t)4 << 20));
    std::vector<Block> blocks;
    size_t used = 0, inflated = 0;
    HHX_TRY(scan_blocks(b->comp, inflated_target, blocks, used, inflated));
    if (blocks.empty()) return b->eof && b->comp.empty() ? 0 : (b->eof ? fail("BAM: truncated BGZF block at the end of the file") : 0);
    const size_t old = b->carry.size();
    b->carry.resize(old + inflated);
    HHX_TRY(inflate_blocks(b->comp.data(), blocks, b->carry.data() + old, b->threads));
    b->comp.erase(b->comp.begin(), b->comp.begin() + (long)used);
    return 0;
}

__global__ __launch_bounds__(256) void k_bam_decode(i64 n, const unsigned char *__restrict__ bytes, const i64 *__restrict__ off, i32 n_ref,
                                                    const i32 *__restrict__ ref_to_ctg, int need_flags, int drop_same_ref,
                                                    i32 *__restrict__ id1, i32 *__restrict__ pos1, i32 *__restrict__ id2, i32 *__restrict__ pos2) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) {
        const unsigned char *r = bytes + off[k];
        auto le32 = [&](int o) { return (i32)((u32)r[o] | ((u32)r[o + 1] << 8) | ((u32)r[o + 2] << 16) | ((u32)r[o + 3] << 24)); };
        const i32 ref = le32(4), pos = le32(8), mref = le32(24), mpos = le32(28);
        const int flag = (int)r[18] | ((int)r[19] << 8);
        bool keep = (flag & need_flags) == need_flags;                     // filter=flag.read1
        if (drop_same_ref && ref == mref) keep = false;                    // && refid != mrefid
        // -2: the record fails the filter (htslib never yields it); -1: unmapped end (refID -1: reference_name is None) or a
        // reference that is not in fa_dict.  hhx_ingest_push drops every negative id.
        id1[k] = !keep ? -2 : ((u32)ref < (u32)n_ref ? ref_to_ctg[ref] : -1);
        id2[k] = !keep ? -2 : ((u32)mref < (u32)n_ref ? ref_to_ctg[mref] : -1);
        pos1[k] = pos;                                                     // reference_start / next_reference_start: 0-based
        pos2[k] = mpos;
    }
}

}  // namespace

extern "C" int hhx_bam_open(const char *path, int threads, hhx_bam **out) {
    if (!path || !out) return fail("null pointer");
    hhx_bam *b = new hhx_bam();
    b->f = fopen(path, "rb");
    if (!b->f) { delete b; return fail("BAM: cannot open %s", path); }
    b->threads = threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency());
    // header: magic "BAM\1", l_text, text, n_ref, then per reference l_name, name (NUL-terminated), l_ref
    auto need = [&](size_t bytes) -> int {
        while (b->carry.size() < bytes) {
            const size_t before = b->carry.size();
            int rc = pull_into_carry(b, (size_t)8 << 20);
            if (rc) return rc;
            if (b->carry.size() == before) return fail("BAM: truncated header");
        }
        return 0;
    };
    int rc = need(12);
    if (!rc && memcmp(b->carry.data(), "BAM\1", 4) != 0) rc = fail("BAM: bad magic (not a BAM file)");
    size_t at = 0;
    if (!rc) {
        const size_t l_text = bgzf_rd32(b->carry.data() + 4);
        rc = need(12 + l_text);
        if (!rc) {
            b->header_text.assign((const char *)b->carry.data() + 8, l_text);
            while (!b->header_text.empty() && b->header_text.back() == '\0') b->header_text.pop_back();
            const size_t n_ref = bgzf_rd32(b->carry.data() + 8 + l_text);
            at = 12 + l_text;
            b->name_off.push_back(0);
            for (size_t r = 0; r < n_ref && !rc; ++r) {
                rc = need(at + 4);
                if (rc) break;
                const size_t l_name = bgzf_rd32(b->carry.data() + at);
                rc = need(at + 4 + l_name + 4);
                if (rc) break;
                std::string nm((const char *)b->carry.data() + at + 4, l_name ? l_name - 1 : 0);
                b->ref_len.push_back((i32)bgzf_rd32(b->carry.data() + at + 4 + l_name));
                b->names_cat += nm;
                b->name_off.push_back((i64)b->names_cat.size());
                b->ref_names.push_back(std::move(nm));
                at += 4 + l_name + 4;
            }
        }
    }
    if (rc) { delete b; return rc; }
    b->carry.erase(b->carry.begin(), b->carry.begin() + (long)at);
    *out = b;
    return 0;
}

extern "C" int hhx_bam_header(hhx_bam *b, i32 *n_ref, const char **text, i64 *text_len, const char **names, const i64 **name_off) {
    if (!b) return fail("null handle");
    if (n_ref) *n_ref = (i32)b->ref_names.size();
    if (text) *text = b->header_text.c_str();
    if (text_len) *text_len = (i64)b->header_text.size();
    if (names) *names = b->names_cat.c_str();
    if (name_off) *name_off = b->name_off.data();
    return 0;
}

extern "C" int hhx_bam_next(hhx_bam *b, int need_flags, int drop_same_ref, i32 n_ref, const i32 *ref_to_ctg_host, i64 max_inflated_bytes,
                            i64 *n_records, void **id1, void **pos1, void **id2, void **pos2) {
    if (!b || !n_records) return fail("null pointer");
    *n_records = 0;
    if (n_ref != (i32)b->ref_names.size() || (n_ref && !ref_to_ctg_host)) return fail("BAM: the reference-id map must have %zu entries", b->ref_names.size());
    if (b->map_n < 0) {
        if (b->d_map.alloc((size_t)n_ref + 1)) return 1;
        if (n_ref) HHX_HIP(hipMemcpyAsync(b->d_map.p, ref_to_ctg_host, sizeof(i32) * (size_t)n_ref, hipMemcpyHostToDevice, g_stream));
        b->map_n = n_ref;
    }
    if (max_inflated_bytes < ((i64)1 << 16)) max_inflated_bytes = (i64)1 << 16;       // one BGZF block inflates to <= 64 KiB
    for (;;) {
        HHX_TRY(refill(b, (size_t)max_inflated_bytes / 3 + ((size_t)1 << 20)));
        std::vector<Block> blocks;
        size_t used = 0, inflated = 0;
        HHX_TRY(scan_blocks(b->comp, (size_t)max_inflated_bytes, blocks, used, inflated));
        if (blocks.empty()) {
            if (!b->eof) return fail("BAM: a BGZF block larger than the read window");
            if (!b->comp.empty()) return fail("BAM: truncated BGZF block at the end of the file");
            if (b->carry.empty()) return 0;                      // end of file: *n_records == 0
            // no compressed data left, but inflated bytes are (hhx_bam_open inflates ahead while it reads the header):
            // they are walked below like any batch
        }
        const size_t total = b->carry.size() + inflated;
        if (total + 64 > b->pin_cap) {
            if (b->pin) (void)hipHostFree(b->pin);
            b->pin = nullptr;
            b->pin_cap = total + total / 4 + 64;
            HHX_HIP(hipHostMalloc((void **)&b->pin, b->pin_cap, hipHostMallocDefault));
        }
        if (!b->carry.empty()) memcpy(b->pin, b->carry.data(), b->carry.size());
        HHX_TRY(inflate_blocks(b->comp.data(), blocks, b->pin + b->carry.size(), b->threads));
        b->comp.erase(b->comp.begin(), b->comp.begin() + (long)used);
        // the sequential part: record offsets
        std::vector<i64> off;
        off.reserve(total / 200 + 16);
        size_t at = 0;
        while (at + 4 <= total) {
            const size_t len = bgzf_rd32(b->pin + at);
            if (len < 32) return fail("BAM: record of %zu bytes (corrupt stream)", len);
            if (at + 4 + len > total) break;
            off.push_back((i64)at);
            at += 4 + len;
        }
        b->carry.assign(b->pin + at, b->pin + total);
        if (off.empty()) {                                       // one record longer than the batch so far: read on
            if (blocks.empty()) return fail("BAM: truncated record at the end of the file");
            continue;
        }
        const i64 n = (i64)off.size();
        if (b->d_bytes.n < at + 16 && b->d_bytes.alloc(at + at / 4 + 16)) return 1;
        if ((i64)b->d_off.n < n && (b->d_off.alloc((size_t)n + n / 4) || b->d_id1.alloc((size_t)n + n / 4) || b->d_pos1.alloc((size_t)n + n / 4) ||
                                   b->d_id2.alloc((size_t)n + n / 4) || b->d_pos2.alloc((size_t)n + n / 4))) return 1;
        HHX_HIP(hipMemcpyAsync(b->d_bytes.p, b->pin, at, hipMemcpyHostToDevice, g_stream));
        HHX_HIP(hipMemcpyAsync(b->d_off.p, off.data(), sizeof(i64) * (size_t)n, hipMemcpyHostToDevice, g_stream));
        {
            KTimer kt("bam_decode");
            k_bam_decode<<<(unsigned)std::max<i64>(1, std::min<i64>((n + 255) / 256, 256 * 16)), 256, 0, g_stream>>>(
                n, b->d_bytes.p, b->d_off.p, b->map_n, b->d_map.p, need_flags, drop_same_ref, b->d_id1.p, b->d_pos1.p, b->d_id2.p, b->d_pos2.p);
        }
        HHX_LAUNCH_CHECK();
        HHX_HIP(hipStreamSynchronize(g_stream));                 // `off` and the pinned buffer are reused by the next batch
        b->records_total += n;
        b->last_n = n;
        *n_records = n;
        if (id1) *id1 = b->d_id1.p;
        if (pos1) *pos1 = b->d_pos1.p;
        if (id2) *id2 = b->d_id2.p;
        if (pos2) *pos2 = b->d_pos2.p;
        return 0;
    }
}

// host copies of the arrays of the last batch (for consumers that want the reference's tuples)
extern "C" int hhx_bam_fetch(hhx_bam *b, i32 *id1, i32 *pos1, i32 *id2, i32 *pos2) {
    if (!b) return fail("null handle");
    const size_t bytes = sizeof(i32) * (size_t)b->last_n;
    if (!bytes) return 0;
    if (id1) HHX_HIP(hipMemcpyAsync(id1, b->d_id1.p, bytes, hipMemcpyDeviceToHost, g_stream));
    if (pos1) HHX_HIP(hipMemcpyAsync(pos1, b->d_pos1.p, bytes, hipMemcpyDeviceToHost, g_stream));
    if (id2) HHX_HIP(hipMemcpyAsync(id2, b->d_id2.p, bytes, hipMemcpyDeviceToHost, g_stream));
    if (pos2) HHX_HIP(hipMemcpyAsync(pos2, b->d_pos2.p, bytes, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int hhx_bam_close(hhx_bam *b) {
    delete b;
    return 0;
}
