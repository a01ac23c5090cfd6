// Device bytes -> one file (paired_links.clm, alignments.bed): shared by hhx_pairs.hip and hhx_jobs.hip.
#pragma once
#include <fcntl.h>
#include <unistd.h>

#include <condition_variable>
#include <deque>
#include <thread>

#include "hhx_common.h"

namespace hhx {

// device bytes -> file: pieces through NBUF pinned buffers, pwrite() at each piece's own offset on NWRITE host threads while the next
// pieces are copied.  (More writers do not help: ONE file takes 3.3-6.3 GB/s on the RAM disk of the MI355X boxes with 1, 4, 8 or 12
// threads, by pwrite() or by memcpy into a shared mapping alike — tools/fs_write_bench.c, profiles/r06_fs_write_bench.jsonl.)
struct FileSink {
    static constexpr size_t PIECE = (size_t)64 << 20;
    static constexpr int NBUF = 4, NWRITE = 2;       // one file takes ~5 GB/s on a RAM disk whatever the number of writers (measured: tools/fs_write_bench.c)
    int fd = -1;
    i64 pos = 0;
    void *pin[NBUF] = {};
    bool busy[NBUF] = {};
    int next = 0, err = 0;
    bool stop = false;
    struct Job { int buf; size_t n; i64 at; };
    std::deque<Job> q;
    std::mutex mu;
    std::condition_variable cv;
    std::thread worker[NWRITE];

    int open_fd(int fd_) {
        fd = fd_;
        for (int b = 0; b < NBUF; ++b) HHX_HIP(hipHostMalloc(&pin[b], PIECE, hipHostMallocDefault));
        for (auto &w : worker) w = std::thread([this] {
            for (;;) {
                Job j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [this] { return stop || !q.empty(); });
                    if (q.empty()) return;
                    j = q.front(); q.pop_front();
                }
                const char *p = (const char *)pin[j.buf];
                size_t left = j.n; i64 at = j.at;
                while (left) {
                    const ssize_t w = ::pwrite(fd, p, left, at);
                    if (w <= 0) { std::lock_guard<std::mutex> lk(mu); err = errno ? errno : EIO; break; }
                    p += w; left -= (size_t)w; at += w;
                }
                { std::lock_guard<std::mutex> lk(mu); busy[j.buf] = false; }
                cv.notify_all();
            }
        });
        return 0;
    }
    int open(const char *path) {
        const int f = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (f < 0) return fail("cannot open %s for writing: %s", path, strerror(errno));
        return open_fd(f);
    }
    int write_device(const unsigned char *dev, size_t n) {
        for (size_t o = 0; o < n; o += PIECE) {
            const size_t m = n - o < PIECE ? n - o : PIECE;
            const int b = next; next = (next + 1) % NBUF;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !busy[b]; }); if (err) return fail("write failed: %s", strerror(err)); }
            HHX_HIP(hipMemcpyAsync(pin[b], dev + o, m, hipMemcpyDeviceToHost, g_stream));
            HHX_HIP(hipStreamSynchronize(g_stream));
            { std::lock_guard<std::mutex> lk(mu); busy[b] = true; q.push_back(Job{b, m, pos}); }
            cv.notify_all();
            pos += (i64)m;
        }
        return 0;
    }
    int close() {
        int rc = 0;
        if (worker[0].joinable()) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { for (bool x : busy) if (x) return false; return true; }); stop = true; }
            cv.notify_all();
            for (auto &w : worker) if (w.joinable()) w.join();
        }
        if (err) rc = fail("write failed: %s", strerror(err));
        for (int b = 0; b < NBUF; ++b) if (pin[b]) { (void)hipHostFree(pin[b]); pin[b] = nullptr; }
        if (fd >= 0) { if (::close(fd) != 0 && !rc) rc = fail("close failed: %s", strerror(errno)); fd = -1; }
        return rc;
    }
    ~FileSink() { (void)close(); }
};


}  // namespace hhx
