// Host-side writers of the files run() produces from the link tables: full_links.pkl / HT_links.pkl.
//
// output_pickle :710-715 is `pickle.dump(dict_, fpkl)` of a defaultdict(int) keyed by name tuples (:1605 :1615).  With the
// tables held as arrays (haphic_amd/containers.py) the same object is serialised here without creating it: a protocol-4
// pickle stream
//     PROTO 4 | collections.defaultdict (builtins.int,) REDUCE | MARK (key value)* SETITEMS ... | STOP
// whose keys are TUPLE2 of two memoised strings (SHORT_BINUNICODE, BINUNICODE for names of 256 bytes and more; every later use of a name is a BINGET / LONG_BINGET to its memo slot, as the pickler does for a str object it has
// seen) and whose values are BININT1 / BININT2 / BININT / LONG1 by size, in batches of 1000 items like pickle's own
// batch_dict.  pickle.load() of the file gives a defaultdict(int) equal to the reference's, in the same insertion order.
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <thread>

#include "hhx_common.h"

using namespace hhx;

namespace {

struct ByteFile {
    int fd = -1;
    std::vector<unsigned char> buf;
    size_t used = 0;
    i64 total = 0;
    int err = 0;
    explicit ByteFile(size_t cap) : buf(cap) {}
    void flush() {
        const unsigned char *p = buf.data();
        size_t left = used;
        while (left && !err) {
            const ssize_t w = ::write(fd, p, left);
            if (w <= 0) { err = errno ? errno : EIO; break; }
            p += w; left -= (size_t)w;
        }
        total += (i64)used;
        used = 0;
    }
    inline unsigned char *room(size_t n) {
        if (used + n > buf.size()) flush();
        unsigned char *p = buf.data() + used;
        used += n;
        return p;
    }
    inline void byte(unsigned char b) { *room(1) = b; }
    inline void u32le(u32 v) { memcpy(room(4), &v, 4); }
};

}  // namespace

// the pickle into an open file descriptor (closed here, whatever happens); `path` only names the file in messages
int hhx::write_link_pickle_fd(int fd, const char *path, i64 n_keys, const i32 *name_i, const i32 *name_j, const i64 *count, i32 n_names,
                              const uint8_t *names_blob, const i64 *name_off, i64 *n_bytes) {
    if ((n_keys && (!name_i || !name_j || !count)) || !name_off || (n_names && !names_blob)) { ::close(fd); return fail("hhx_write_link_pickle: null pointer"); }
    for (i64 k = 0; k < n_keys; ++k)
        if ((u32)name_i[k] >= (u32)n_names || (u32)name_j[k] >= (u32)n_names) { ::close(fd); return fail("hhx_write_link_pickle: key %lld names an unknown id", (long long)k); }
    ByteFile f((size_t)8 << 20);
    f.fd = fd;
    static const unsigned char head[] = "\x80\x04"                                    // PROTO 4
                                        "\x8c\x0b" "collections" "\x94" "\x8c\x0b" "defaultdict" "\x94" "\x93" "\x94"   // STACK_GLOBAL, memo 0-2
                                        "\x8c\x08" "builtins" "\x94" "\x8c\x03" "int" "\x94" "\x93" "\x94"              // memo 3-5
                                        "\x85\x94" "R\x94";                                                             // TUPLE1 (6), REDUCE (7)
    memcpy(f.room(sizeof head - 1), head, sizeof head - 1);
    // memo slot of every name = 8 + its rank by first use (i before j, key by key): fixed before any byte is written, so that
    // the keys can be encoded in independent slices by several threads and the slices concatenated
    const int n_thr = (int)std::max<i64>(1, std::min<i64>({(i64)std::thread::hardware_concurrency(), (i64)8, n_keys / 200000 + 1}));
    std::vector<i64> first((size_t)n_names, INT64_MAX);
    {
        std::vector<std::vector<i64>> local((size_t)n_thr);
        std::vector<std::thread> pool;
        for (int t = 0; t < n_thr; ++t)
            pool.emplace_back([&, t] {
                std::vector<i64> &m = local[(size_t)t];
                m.assign((size_t)n_names, INT64_MAX);
                const i64 k0 = n_keys * t / n_thr, k1 = n_keys * (t + 1) / n_thr;
                for (i64 k = k0; k < k1; ++k) {
                    i64 &a = m[name_i[k]];
                    if (a == INT64_MAX) a = 2 * k;
                    i64 &b = m[name_j[k]];
                    if (b == INT64_MAX) b = 2 * k + 1;
                }
            });
        for (auto &th : pool) th.join();
        for (int t = 0; t < n_thr; ++t)
            for (i32 id = 0; id < n_names; ++id) first[id] = std::min(first[id], local[(size_t)t][id]);
    }
    std::vector<u32> memo((size_t)n_names, 0u);
    {
        std::vector<std::pair<i64, i32>> order;
        for (i32 id = 0; id < n_names; ++id)
            if (first[id] != INT64_MAX) order.emplace_back(first[id], id);
        std::sort(order.begin(), order.end());
        for (size_t r = 0; r < order.size(); ++r) memo[order[r].second] = 8 + (u32)r;
    }
    // one slice of the keys -> bytes (whole SETITEMS batches of 1000 keys)
    auto encode = [&](i64 k0, i64 k1, std::vector<unsigned char> &out) {
        out.clear();
        out.reserve((size_t)(k1 - k0) * 14 + 4096);
        auto put_name = [&](i32 id, i64 at) {
            if (first[id] != at) {
                const u32 m = memo[id];
                if (m < 256) { out.push_back('h'); out.push_back((unsigned char)m); }                  // BINGET
                else { out.push_back('j'); const unsigned char *q = (const unsigned char *)&m; out.insert(out.end(), q, q + 4); }   // LONG_BINGET
                return;
            }
            const u32 len = (u32)(name_off[id + 1] - name_off[id]);
            if (len < 256) { out.push_back(0x8c); out.push_back((unsigned char)len); }                 // SHORT_BINUNICODE
            else { out.push_back('X'); const unsigned char *q = (const unsigned char *)&len; out.insert(out.end(), q, q + 4); }    // BINUNICODE
            out.insert(out.end(), names_blob + name_off[id], names_blob + name_off[id + 1]);
            out.push_back(0x94);                                                                       // MEMOIZE
        };
        for (i64 b0 = k0; b0 < k1; b0 += 1000) {
            const i64 b1 = std::min<i64>(k1, b0 + 1000);
            out.push_back('(');                                                                        // MARK
            for (i64 k = b0; k < b1; ++k) {
                put_name(name_i[k], 2 * k);
                put_name(name_j[k], 2 * k + 1);
                out.push_back(0x86);                                                                   // TUPLE2
                const i64 v = count[k];
                if (v >= 0 && v < 256) { out.push_back('K'); out.push_back((unsigned char)v); }
                else if (v >= 0 && v < 65536) { out.push_back('M'); out.push_back((unsigned char)(v & 255)); out.push_back((unsigned char)(v >> 8)); }
                else if (v >= INT32_MIN && v <= INT32_MAX) { const i32 w = (i32)v; out.push_back('J'); const unsigned char *q = (const unsigned char *)&w; out.insert(out.end(), q, q + 4); }
                else {                                                                                 // LONG1: little-endian two's complement, minimal length
                    unsigned char b[9];
                    int n = 0;
                    i64 t = v;
                    for (;;) {
                        b[n++] = (unsigned char)(t & 255);
                        const i64 rest = t >> 8;                                                       // arithmetic shift
                        if ((rest == 0 && !(b[n - 1] & 0x80)) || (rest == -1 && (b[n - 1] & 0x80))) break;
                        t = rest;
                    }
                    out.push_back(0x8a); out.push_back((unsigned char)n);
                    out.insert(out.end(), b, b + n);
                }
            }
            out.push_back('u');                                                                        // SETITEMS
        }
    };
    // rounds of n_thr slices of SLICE keys: encoded in parallel, written in order by one thread while the next round is encoded
    const i64 SLICE = 1000 * 500;
    std::vector<std::vector<unsigned char>> bufs[2];
    bufs[0].resize((size_t)n_thr);
    bufs[1].resize((size_t)n_thr);
    f.flush();
    std::thread writer;
    int side = 0;
    for (i64 r0 = 0; r0 < n_keys; r0 += SLICE * n_thr, side ^= 1) {
        std::vector<std::thread> pool;
        int used = 0;
        for (int t = 0; t < n_thr && r0 + SLICE * t < n_keys; ++t, ++used)
            pool.emplace_back([&, t] { encode(r0 + SLICE * t, std::min<i64>(n_keys, r0 + SLICE * (t + 1)), bufs[side][(size_t)t]); });
        for (auto &th : pool) th.join();
        if (writer.joinable()) writer.join();
        if (f.err) break;
        writer = std::thread([&f, &bufs, side, used] {
            for (int t = 0; t < used && !f.err; ++t) {
                const unsigned char *p = bufs[side][(size_t)t].data();
                size_t left = bufs[side][(size_t)t].size();
                f.total += (i64)left;
                while (left) {
                    const ssize_t w = ::write(f.fd, p, left);
                    if (w <= 0) { f.err = errno ? errno : EIO; break; }
                    p += w; left -= (size_t)w;
                }
            }
        });
    }
    if (writer.joinable()) writer.join();
    f.byte('.');                                                                                      // STOP
    f.flush();
    const int cerr = ::close(f.fd) != 0 ? errno : 0;
    if (f.err || cerr) return fail("writing %s failed: %s", path, strerror(f.err ? f.err : cerr));
    if (n_bytes) *n_bytes = f.total;
    return 0;
}

extern "C" int hhx_write_link_pickle(const char *path, int64_t n_keys, const int32_t *name_i, const int32_t *name_j, const int64_t *count, int32_t n_names,
                                     const uint8_t *names_blob, const int64_t *name_off, int64_t *n_bytes) {
    if (!path) return fail("hhx_write_link_pickle: null pointer");
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return fail("cannot open %s for writing: %s", path, strerror(errno));
    return write_link_pickle_fd(fd, path, n_keys, name_i, name_j, count, n_names, names_blob, name_off, n_bytes);
}
