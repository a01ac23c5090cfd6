// BGZF (SAM specification 4.1): independent raw-DEFLATE members of at most 64 KiB, each with its compressed size in a BC extra subfield and its
// inflated size in the trailer — split by scan_blocks, inflated by a pool of threads (zlib) straight to the prefix sums of their ISIZE fields.
// Shared by the BAM front end (hhx_bam.hip) and the bgzipped .pairs reader (hhx_reader.hip).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "hhx_common.h"

namespace hhx {

struct Block { size_t cdata, clen, isize, out; };

inline u32 bgzf_rd32(const unsigned char *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

// Splits b->comp into whole BGZF blocks (at most max_inflated bytes of output).  Returns the number of compressed bytes
// covered; blocks[] get offsets into b->comp.  rc != 0 on a malformed header.
inline int scan_blocks(const std::vector<unsigned char> &comp, size_t max_inflated, std::vector<Block> &blocks, size_t &used, size_t &inflated) {
    used = 0; inflated = 0;
    const size_t n = comp.size();
    while (used + 18 <= n) {
        const unsigned char *p = comp.data() + used;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return fail("not a BGZF block at compressed offset (+%zu)", used);
        const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
        if (used + 12 + xlen > n) break;
        size_t bsize = 0;
        for (size_t x = 0; x + 4 <= xlen;) {                     // extra subfields: SI1 SI2 SLEN(2) data
            const unsigned char *s = p + 12 + x;
            const size_t slen = (size_t)s[2] | ((size_t)s[3] << 8);
            if (s[0] == 'B' && s[1] == 'C' && slen == 2) bsize = ((size_t)s[4] | ((size_t)s[5] << 8)) + 1;
            x += 4 + slen;
        }
        if (!bsize || bsize < 12 + xlen + 8) return fail("not a BGZF block: no BC subfield");
        if (used + bsize > n) break;
        const size_t isize = bgzf_rd32(p + bsize - 4);
        if (inflated + isize > max_inflated && !blocks.empty()) break;
        blocks.push_back({used + 12 + xlen, bsize - 12 - xlen - 8, isize, inflated});
        inflated += isize;
        used += bsize;
    }
    return 0;
}

inline int inflate_blocks(const unsigned char *comp, const std::vector<Block> &blocks, unsigned char *out, int threads) {
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        z_stream zs;
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= blocks.size()) return;
            const Block &b = blocks[k];
            if (b.isize == 0) continue;                          // the empty EOF marker block
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
            zs.next_in = const_cast<unsigned char *>(comp + b.cdata);
            zs.avail_in = (uInt)b.clen;
            zs.next_out = out + b.out;
            zs.avail_out = (uInt)b.isize;
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.avail_out != 0) bad = 1;
            inflateEnd(&zs);
        }
    };
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, blocks.size()));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return bad ? fail("a BGZF block failed to inflate") : 0;
}


}  // namespace hhx
