// The library's file-writer thread: the three files run() writes between its seams — HT_links.pkl :2879,
// paired_links.clm :2888, full_links.pkl :2929 (output_pickle :710-715, output_clm :376-392) — are read by nothing later in run(), yet their
// encoding (7.3 s at 100k contigs / 500 M pairs) sat between parse_alignments* and filter_fragments / dict_to_matrix / the inflation sweep.
// The *_async entry points validate and open the file on the caller's thread, queue the work and return; ONE host thread owned by the
// library runs the queue in submission order on a non-blocking stream of its own with a pool arena of its own (hhx_runtime.hip: a free list
// belongs to one stream), so the device half of a job (grouping / sorting / formatting the CLM text, ordering the HT items) overlaps the
// caller's kernels.  hhx_files_join waits for the queue and returns the first failure; hhx_ingest_destroy waits for the jobs of its handle.
#include <fcntl.h>
#include <unistd.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <thread>

#include "hhx_filesink.h"
#include "hhx_ingest.h"

using namespace hhx;

namespace {

struct Job {
    std::string what;
    const void *handle;
    int device;
    std::function<int()> run;
};

struct Writer {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job> q;
    const void *running_handle = nullptr;
    bool running = false, stop = false, started = false;
    std::thread th;
    std::vector<std::string> errors;
    i64 n_done = 0;
    std::map<int, hipStream_t> streams;          // one per device the jobs came from
    Arena arena;

    void loop() {
        g_arena = &arena;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                j = std::move(q.front());
                q.pop_front();
                running = true;
                running_handle = j.handle;
            }
            std::string err;
            hipStream_t s = nullptr;
            auto it = streams.find(j.device);
            if (j.device < 0) s = nullptr;
            else if (hipSetDevice(j.device) != hipSuccess) err = "hipSetDevice failed on the file-writer thread";
            else if (it != streams.end()) s = it->second;
            else if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) err = "hipStreamCreateWithFlags failed on the file-writer thread";
            else streams[j.device] = s;
            if (err.empty()) {
                g_stream = s;
                g_err.clear();
                if (j.run()) err = g_err.empty() ? "failed" : g_err;
                if (j.device >= 0) (void)hipStreamSynchronize(s);
            }
            j.run = nullptr;                       // what the job held (name tables, host copies) goes before anybody is told
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!err.empty()) errors.push_back(j.what + ": " + err);
                ++n_done;
                running = false;
                running_handle = nullptr;
                if (q.empty()) arena_donate(&arena);          // the stream is idle: its cached blocks serve the callers from here on
            }
            cv_done.notify_all();
        }
    }
    ~Writer() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;                           // a queue that is not empty here is still worked off (loop() returns on an empty queue only)
        }
        cv_work.notify_all();
        if (th.joinable()) th.join();
    }
};

Writer &writer() {
    static Writer w;
    return w;
}

}  // namespace

int hhx::files_submit(const std::string &what, const void *handle, std::function<int()> job) {
    int dev = -1;                                // -1: no device (host-only jobs: the pickle of caller-owned arrays on a machine without a GPU)
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
    Writer &w = writer();
    {
        std::lock_guard<std::mutex> lk(w.mu);
        if (!w.started) {
            w.started = true;
            w.th = std::thread([&w] { w.loop(); });
        }
        w.q.push_back(Job{what, handle, dev, std::move(job)});
    }
    w.cv_work.notify_all();
    return 0;
}

void hhx::files_wait_handle(const void *handle) {
    Writer &w = writer();
    std::unique_lock<std::mutex> lk(w.mu);
    w.cv_done.wait(lk, [&] {
        if (w.running && w.running_handle == handle) return false;
        for (const Job &j : w.q) if (j.handle == handle) return false;
        return true;
    });
}

extern "C" int hhx_files_pending(int64_t *n_pending, int64_t *n_done) {
    Writer &w = writer();
    std::lock_guard<std::mutex> lk(w.mu);
    if (n_pending) *n_pending = (i64)w.q.size() + (w.running ? 1 : 0);
    if (n_done) *n_done = w.n_done;
    return 0;
}

extern "C" int hhx_files_join(int64_t *n_failed) {
    Writer &w = writer();
    std::vector<std::string> errs;
    {
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv_done.wait(lk, [&] { return w.q.empty() && !w.running; });
        errs.swap(w.errors);
    }
    if (n_failed) *n_failed = (i64)errs.size();
    if (errs.empty()) return 0;
    std::string msg = errs[0];
    if (errs.size() > 1) msg += " (and " + std::to_string(errs.size() - 1) + " more file(s) failed)";
    return fail("%s", msg.c_str());
}

// full_links.pkl / HT_links.pkl (which = 0 full_link_dict, 1 HT_link_dict, 2 flank_link_dict with its integer counts) of a finalized handle, queued:
// the writer thread fetches the items in dict order from the device tables (HT: hhx_ingest_fetch_ht_items, ordered on the device) into
// host memory of its own and encodes them (hhx_write_link_pickle).  names: the table the ids index — contigs for 0, [c0_H, c0_T, c1_H, ...] for 1,
// fragments for 2.
extern "C" int hhx_ingest_write_link_pickle_async(hhx_ingest *h, int which, const char *path, int32_t n_names, const uint8_t *names_blob, const int64_t *name_off) {
    if (!h || !h->finalized) return fail("ingest handle not finalized");
    if (which < 0 || which > 2) return fail("hhx_ingest_write_link_pickle_async: which = %d", which);
    if (!path || !name_off || (n_names && !names_blob)) return fail("hhx_ingest_write_link_pickle_async: null pointer");
    if (which == 1 && (!h->keep_pairs || h->pairs_dropped)) return fail("hhx_ingest_write_link_pickle_async: HT_link_dict needs the kept read pairs (hhx_ingest_keep_pairs)");
    const i32 need = which == 0 ? h->t.n_ctg : which == 1 ? 2 * h->t.n_ctg : h->t.n_frag;
    if (n_names < need) return fail("hhx_ingest_write_link_pickle_async: %d names for ids up to %d", n_names, need);
    const i32 *fi = nullptr, *fj = nullptr;
    HHX_TRY(hhx_ingest_ordered_full_device(h, &fi, &fj));            // made on the caller's thread; the writer thread only reads them
    HHX_HIP(hipStreamSynchronize(g_stream));
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return fail("cannot open %s for writing: %s", path, strerror(errno));
    auto blob = std::make_shared<std::vector<uint8_t>>(names_blob, names_blob + (size_t)name_off[n_names] + (n_names ? 0 : 1));
    auto off = std::make_shared<std::vector<i64>>(name_off, name_off + n_names + 1);
    const std::string p = path;
    return files_submit(std::string(which == 1 ? "HT_link_dict -> " : which == 0 ? "full_link_dict -> " : "flank_link_dict -> ") + p, h, [h, which, fd, p, n_names, blob, off]() -> int {
        std::vector<i32> ni, nj;
        std::vector<i64> cnt;
        i64 n = 0;
        int rc = 0;
        if (which == 1) {
            rc = hhx_ingest_fetch_ht_items(h, &n, nullptr, nullptr, nullptr);
            if (!rc && n) {
                ni.resize((size_t)n); nj.resize((size_t)n); cnt.resize((size_t)n);
                rc = hhx_ingest_fetch_ht_items(h, &n, ni.data(), nj.data(), cnt.data());
            }
        } else {
            n = which == 0 ? h->n_full : h->n_flank;
            ni.resize((size_t)n); nj.resize((size_t)n); cnt.resize((size_t)n);
            rc = which == 0 ? hhx_ingest_fetch(h, ni.data(), nj.data(), cnt.data(), nullptr, nullptr, nullptr, nullptr, nullptr)
                            : hhx_ingest_fetch(h, nullptr, nullptr, nullptr, nullptr, ni.data(), nj.data(), cnt.data(), nullptr);
        }
        if (rc) { ::close(fd); return rc; }
        return write_link_pickle_fd(fd, p.c_str(), n, ni.data(), nj.data(), cnt.data(), n_names, blob->data(), off->data(), nullptr);
    });
}

// the pickle of arrays the CALLER owns, queued: name_i / name_j / count must stay untouched until hhx_files_join (the Python binding keeps them)
extern "C" int hhx_write_link_pickle_async(const char *path, int64_t n_keys, const int32_t *name_i, const int32_t *name_j, const int64_t *count, int32_t n_names,
                                           const uint8_t *names_blob, const int64_t *name_off) {
    if (!path || (n_keys && (!name_i || !name_j || !count)) || !name_off || (n_names && !names_blob)) return fail("hhx_write_link_pickle_async: null pointer");
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return fail("cannot open %s for writing: %s", path, strerror(errno));
    auto blob = std::make_shared<std::vector<uint8_t>>(names_blob, names_blob + (size_t)name_off[n_names] + (n_names ? 0 : 1));
    auto off = std::make_shared<std::vector<i64>>(name_off, name_off + n_names + 1);
    const std::string p = path;
    return files_submit("link pickle -> " + p, nullptr, [fd, p, n_keys, name_i, name_j, count, n_names, blob, off]() -> int {
        return write_link_pickle_fd(fd, p.c_str(), n_keys, name_i, name_j, count, n_names, blob->data(), off->data(), nullptr);
    });
}


// ---------------------------------------------------------------- alignments.bed, deferred
// pairs_generator* :1549-1557 write two BED records per read pair inside their loop: 67 GB next to a 50 GB .pairs file at C3, and a RAM disk of the
// MI355X boxes takes ~4.5 GB/s into one file — 15 s on the critical path of a stage that tokenises the text in 3.5 s.  Nothing in run() reads the file.
// A byte sink keeps the chunks WHERE THEY ARE MADE, in HBM (288 GB: the 67 GB fit beside everything else), and hands them to the file-writer thread:
// hhx_byte_sink_push_device takes the device buffer over (no copy), returns at once while fewer than `hbm_budget` bytes are waiting, and blocks the
// producer at the writer's pace beyond that.  The chunk jobs run in the queue like every other file; hhx_byte_sink_close queues the close.
struct hhx_byte_sink {
    std::string path;
    FileSink out;                        // used by the writer thread only
    bool opened = false, failed = false;
    std::mutex mu;
    std::condition_variable cv;
    i64 waiting = 0, budget = 0, pushed = 0;
};

extern "C" int hhx_byte_sink_open(const char *path, int64_t hbm_budget_bytes, hhx_byte_sink **out) {
    if (!path || !out) return fail("hhx_byte_sink_open: null pointer");
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return fail("cannot open %s for writing: %s", path, strerror(errno));
    auto *s = new hhx_byte_sink();
    s->path = path;
    s->out.fd = fd;                      // the pinned buffers and the pwrite() threads are made by the first chunk's job, on the writer thread
    if (hbm_budget_bytes <= 0) {
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); t = (size_t)64 << 30; }
        hbm_budget_bytes = (i64)(t / 4);                             // a quarter of the device
        const i64 env = tune_get("bed_hbm_gb", 0);
        if (env > 0) hbm_budget_bytes = env << 30;
    }
    s->budget = hbm_budget_bytes;
    *out = s;
    return 0;
}

// dev (a block of the library's pool, `bytes` long; n_bytes of it are the payload) now belongs to the sink.  Used by hhx_pairs_parser_bed_to_sink.
int hhx::byte_sink_push_block(hhx_byte_sink *s, void *dev, i64 n_bytes) {
    if (!s || !dev) return fail("hhx_byte_sink: null pointer");
    hipEvent_t ev = nullptr;
    HHX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HHX_HIP(hipEventRecord(ev, g_stream));                           // the kernel that fills the block is in flight on the caller's stream
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->waiting == 0 || s->waiting + n_bytes <= s->budget; });
        s->waiting += n_bytes;
        s->pushed += n_bytes;
    }
    return files_submit("bytes -> " + s->path, s, [s, dev, n_bytes, ev]() -> int {
        int rc = 0;
        if (!s->failed) {
            if (!s->opened) { rc = s->out.open_fd(s->out.fd); s->opened = true; }
            if (!rc && hipStreamWaitEvent(g_stream, ev, 0) != hipSuccess) rc = fail("hipStreamWaitEvent failed");
            if (!rc) rc = s->out.write_device((const unsigned char *)dev, (size_t)n_bytes);      // synchronises the stream piece by piece
            if (rc) s->failed = true;                                // the first failure is the one reported; later chunks are dropped
        }
        (void)hipStreamSynchronize(g_stream);
        (void)hipEventDestroy(ev);
        pool_free_synced(dev);                                       // idle on both streams: back to the callers' list
        { std::lock_guard<std::mutex> lk(s->mu); s->waiting -= n_bytes; }
        s->cv.notify_all();
        return rc;
    });
}

extern "C" int hhx_byte_sink_close(hhx_byte_sink *s, int64_t *n_bytes_pushed) {
    if (!s) return 0;
    if (n_bytes_pushed) *n_bytes_pushed = s->pushed;
    return files_submit("close " + s->path, s, [s]() -> int {
        int rc = 0;
        if (s->opened) rc = s->out.close();
        else if (s->out.fd >= 0) { if (::close(s->out.fd) != 0) rc = fail("close failed: %s", strerror(errno)); s->out.fd = -1; }
        const bool failed = s->failed;
        delete s;
        return failed ? 0 : rc;                                       // a failed chunk has already been reported
    });
}
