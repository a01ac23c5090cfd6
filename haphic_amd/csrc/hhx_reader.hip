// A text file as chunks of whole lines in PINNED host memory (the front of a1: pairs_generator* :1539-1583 read the .pairs file line by line).
// The file is read with pread() by a few threads into one of two pinned buffers while the caller tokenises the other (hhx_pairs_parse copies a
// pinned chunk to the device at PCIe rate; from a memory map of the file the same copy ran at 14 GB/s and unmapping 50 GB cost another second).
// A chunk ends after its last line break; the cut-off tail is carried to the front of the next chunk.  Universal newlines as in hhx_text.hip:
// a chunk may end on '\n' or on a '\r' — a "\r\n" pair split across two chunks would count an empty line more, which the tokeniser skips (:1552).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <thread>

#include "hhx_bgzf.h"

using namespace hhx;

struct hhx_text_reader {
    int fd = -1;
    i64 size = 0, at = 0;                 // file size, next byte to read
    size_t chunk = 0, cap = 0;
    int n_threads = 4;
    unsigned char *buf[2] = {nullptr, nullptr};
    i64 len[2] = {0, 0};                  // bytes of whole lines ready in buf[k]
    int state[2] = {0, 0};                // 0 free for the reader, 1 filled, 2 held by the caller
    std::vector<unsigned char> tail;      // the bytes after the last line break of the chunk read last: the front of the next chunk
    bool bgzf = false;                    // the file is BGZF: `at` / `size` count COMPRESSED bytes, the buffers hold inflated text
    std::vector<unsigned char> comp;      // compressed bytes read and not yet inflated (whole blocks + a partial tail)
    std::string emsg;                     // set with err == EBADMSG
    int fill = 0, take = 0;               // next buffer to fill / to hand out
    bool eof = false, stop = false;
    int err = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::thread th;

    void read_range(unsigned char *dst, i64 off, size_t n) {
        const int T = n < ((size_t)8 << 20) ? 1 : n_threads;
        std::vector<std::thread> pool;
        const size_t span = ((n + T - 1) / T + 4095) & ~(size_t)4095;
        auto work = [&](size_t lo, size_t hi) {
            while (lo < hi) {
                const ssize_t k = ::pread(fd, dst + lo, hi - lo, (off_t)(off + (i64)lo));
                if (k <= 0) { std::lock_guard<std::mutex> lk(mu); err = k < 0 ? errno : EIO; return; }
                lo += (size_t)k;
            }
        };
        for (int t = 1; t < T; ++t) { const size_t lo = std::min(n, span * t), hi = std::min(n, span * (t + 1)); if (lo < hi) pool.emplace_back(work, lo, hi); }
        work(0, std::min(n, span));
        for (auto &p : pool) p.join();
    }
    void loop() {
        for (;;) {
            int b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop || state[fill] == 0; });
                if (stop) return;
                b = fill;
            }
            size_t have = tail.size();
            if (have) memcpy(buf[b], tail.data(), have);             // buf[b] is free: the caller has let go of it
            tail.clear();
            size_t cut = 0;
            bool last = false;
            for (;;) {
                if (bgzf) {
                    // compressed bytes for about a chunk of text (BGZF of text: 3-5 x), whole blocks inflated in place behind what the buffer holds
                    const size_t want_c = (size_t)std::min<i64>((i64)std::max<size_t>(chunk / 3, (size_t)1 << 20), size - at);
                    if (want_c) {
                        const size_t old = comp.size();
                        comp.resize(old + want_c);
                        read_range(comp.data() + old, at, want_c);
                        at += (i64)want_c;
                    }
                    std::vector<Block> blocks;
                    size_t used = 0, inflated = 0;
                    const size_t room = cap - have;
                    if (scan_blocks(comp, room, blocks, used, inflated)) { std::lock_guard<std::mutex> lk(mu); err = EBADMSG; emsg = g_err; eof = true; cv.notify_all(); return; }
                    if (!blocks.empty() && inflated > room) { std::lock_guard<std::mutex> lk(mu); err = EOVERFLOW; eof = true; cv.notify_all(); return; }
                    if (blocks.empty() && at >= size && !comp.empty()) { std::lock_guard<std::mutex> lk(mu); err = EBADMSG; emsg = "truncated BGZF block at the end of the file"; eof = true; cv.notify_all(); return; }
                    if (!blocks.empty()) {
                        if (inflate_blocks(comp.data(), blocks, buf[b] + have, n_threads)) { std::lock_guard<std::mutex> lk(mu); err = EBADMSG; emsg = g_err; eof = true; cv.notify_all(); return; }
                        comp.erase(comp.begin(), comp.begin() + (long)used);
                        have += inflated;
                    }
                    last = at >= size && comp.empty();
                    cut = have;
                    if (last) break;
                    if (have < chunk && have + ((size_t)64 << 10) <= cap) continue;      // less than a chunk of text so far
                } else {
                    const size_t want = (size_t)std::min<i64>((i64)chunk, size - at);
                    if (have + want > cap) { std::lock_guard<std::mutex> lk(mu); err = EOVERFLOW; eof = true; cv.notify_all(); return; }      // a line longer than a chunk
                    if (want) read_range(buf[b] + have, at, want);
                    at += (i64)want;
                    have += want;
                    last = at >= size;
                    cut = have;
                    if (last) break;
                }
                while (cut > 0 && buf[b][cut - 1] != '\n' && buf[b][cut - 1] != '\r') --cut;
                if (cut > 0) break;                                  // else: no line break in the whole chunk, keep reading into the same buffer
                if (bgzf && have + ((size_t)64 << 10) > cap) { std::lock_guard<std::mutex> lk(mu); err = EOVERFLOW; eof = true; cv.notify_all(); return; }
            }
            if (!last) tail.assign(buf[b] + cut, buf[b] + have);
            {
                std::lock_guard<std::mutex> lk(mu);
                len[b] = (i64)cut;
                state[b] = 1;
                fill = b ^ 1;
                if (last) eof = true;
            }
            cv.notify_all();
            if (last) return;
        }
    }
};

static int text_reader_open(const char *path, int64_t chunk_bytes, int n_threads, bool bgzf, hhx_text_reader **out);
extern "C" int hhx_text_reader_open(const char *path, int64_t chunk_bytes, int n_threads, hhx_text_reader **out) {
    return text_reader_open(path, chunk_bytes, n_threads, false, out);
}
// the same over a BGZF file (bgzip): the chunks are inflated text.  Fails with "not a BGZF file" on anything else (plain gzip included)
extern "C" int hhx_text_reader_open_bgzf(const char *path, int64_t chunk_bytes, int n_threads, hhx_text_reader **out) {
    return text_reader_open(path, chunk_bytes, n_threads, true, out);
}
static int text_reader_open(const char *path, int64_t chunk_bytes, int n_threads, bool bgzf, hhx_text_reader **out) {
    if (!path || !out || chunk_bytes <= 0) return fail("hhx_text_reader_open: bad argument");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail("cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); return fail("cannot stat %s: %s", path, strerror(errno)); }
    if (bgzf) {
        unsigned char h[18];
        const ssize_t k = ::pread(fd, h, sizeof h, 0);
        const bool ok = k == (ssize_t)sizeof h && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
        if (!ok && st.st_size != 0) { ::close(fd); return fail("%s is not a BGZF file (bgzip); a plain gzip stream cannot be inflated in parallel", path); }
    }
    auto *r = new hhx_text_reader();
    r->fd = fd;
    r->size = (i64)st.st_size;
    r->chunk = (size_t)chunk_bytes;
    r->cap = 2 * (size_t)chunk_bytes + 4096;                           // a carried tail is shorter than a chunk (or the file has a line longer than one)
    r->bgzf = bgzf;
    if (!bgzf && (size_t)r->size + 4096 < r->cap) r->cap = (size_t)r->size + 4096;      // a small file: no more pinned memory than it has bytes (pinning costs ~0.5 ms per MB)
    if (bgzf && (size_t)r->size * 12 + ((size_t)256 << 10) < r->cap) r->cap = (size_t)r->size * 12 + ((size_t)256 << 10);     // (text deflates 3-5 x; 12 x is the bound assumed)
    r->n_threads = n_threads > 0 ? std::min(n_threads, 16) : 4;
    for (int k = 0; k < 2; ++k)
        if (hipHostMalloc((void **)&r->buf[k], r->cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < k; ++q) (void)hipHostFree(r->buf[q]);
            ::close(fd);
            delete r;
            return fail("hhx_text_reader_open: no pinned memory for two buffers of %zu bytes", (size_t)(2 * chunk_bytes + 4096));
        }
    if (r->size == 0) r->eof = true;
    else r->th = std::thread([r] { r->loop(); });
    *out = r;
    return 0;
}

// the next chunk of whole lines: *host stays valid until the next call (the other buffer is being filled meanwhile); *n_bytes == 0: the end of the file
extern "C" int hhx_text_reader_next(hhx_text_reader *r, const uint8_t **host, int64_t *n_bytes) {
    if (!r || !host || !n_bytes) return fail("hhx_text_reader_next: null pointer");
    std::unique_lock<std::mutex> lk(r->mu);
    // the buffer handed out last time goes back to the reader: the caller is done with it (hhx_pairs_parse synchronises its stream after the
    // host -> device copy, before it returns)
    const int prev = r->take ^ 1;
    if (r->state[prev] == 2) { r->state[prev] = 0; r->cv.notify_all(); }
    r->cv.wait(lk, [r] { return r->state[r->take] == 1 || r->err || (r->eof && r->state[r->take] != 1); });
    if (r->err) return fail("reading the text file failed: %s", r->err == EOVERFLOW ? "a line is longer than a chunk" : r->err == EBADMSG ? r->emsg.c_str() : strerror(r->err));
    if (r->state[r->take] != 1) { *host = nullptr; *n_bytes = 0; return 0; }
    *host = r->buf[r->take];
    *n_bytes = r->len[r->take];
    r->state[r->take] = 2;
    r->take ^= 1;
    return 0;
}

extern "C" int hhx_text_reader_close(hhx_text_reader *r) {
    if (!r) return 0;
    { std::lock_guard<std::mutex> lk(r->mu); r->stop = true; }
    r->cv.notify_all();
    if (r->th.joinable()) r->th.join();
    for (int k = 0; k < 2; ++k) if (r->buf[k]) (void)hipHostFree(r->buf[k]);
    if (r->fd >= 0) ::close(r->fd);
    delete r;
    return 0;
}
