// The library's file-writer threads: the files run() writes between its seams — alignments.bed inside the generators :1549-1557,
// HT_links.pkl :2879, paired_links.clm :2888, full_links.pkl :2929 (output_pickle :710-715, output_clm :376-392) — are read by nothing later in run(),
// yet writing them (98 GB at 100k contigs / 500 M pairs: 20+ s on a RAM disk that takes ~4.3 GB/s per file) sat between parse_alignments* and
// filter_fragments / dict_to_matrix / the inflation sweep.  The *_async entry points validate and open the file on the caller's thread, queue the work
// and return; a LANE is a host thread owned by the library that runs its queue in submission order on a low-priority non-blocking stream of its own with a
// pool arena of its own (hhx_runtime.hip: a free list belongs to one stream), so the device half of a job (grouping / sorting / formatting the CLM
// text, ordering the HT items, copying the BED slabs out) overlaps the caller's kernels.  Two lanes, because several files scale where one does not.
// hhx_files_join waits for both queues and returns the first failure; hhx_ingest_destroy waits for the jobs of its handle.
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
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

// the writers' kernels (sorts, text formatting) and copies must not take compute units from the caller's kernels: the lowest stream priority
hipError_t create_low_priority_stream(hipStream_t *s) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; }
    hipError_t e = hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
    return e;
}

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
            else if (create_low_priority_stream(&s) != hipSuccess) err = "creating the stream of the file-writer thread failed";
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

// Two lanes, each a thread with its own stream and arena: ONE file takes ~4.3 GB/s on the RAM disk of the MI355X boxes however many threads write it
// (the inode's write lock), two files take 7.8 and four 12.7 (tools/fs_write_bench2.c) — so alignments.bed (the byte sink: lane 0) and the files made
// from the ingest handle (HT_links.pkl, paired_links.clm, full_links.pkl: lane 1) are written side by side.  Jobs of one lane run in submission order.
constexpr int N_LANES = 2;
Writer &writer(int lane) {
    static Writer w[N_LANES];
    static const int lanes = (int)std::max<i64>(1, std::min<i64>(N_LANES, tune_get("file_lanes", N_LANES)));     // HHX_FILE_LANES=1: one writer thread for everything
    return w[lane < lanes ? lane : lanes - 1];
}

}  // namespace

int hhx::files_submit(const std::string &what, const void *handle, std::function<int()> job, int lane) {
    int dev = -1;                                // -1: no device (host-only jobs: the pickle of caller-owned arrays on a machine without a GPU)
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
    Writer &w = writer(lane < 0 || lane >= N_LANES ? 1 : lane);
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
    for (int lane = 0; lane < N_LANES; ++lane) {
        Writer &w = writer(lane);
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv_done.wait(lk, [&] {
            if (w.running && w.running_handle == handle) return false;
            for (const Job &j : w.q) if (j.handle == handle) return false;
            return true;
        });
    }
}

extern "C" int hhx_files_pending(int64_t *n_pending, int64_t *n_done) {
    i64 pend = 0, done = 0;
    for (int lane = 0; lane < N_LANES; ++lane) {
        Writer &w = writer(lane);
        std::lock_guard<std::mutex> lk(w.mu);
        pend += (i64)w.q.size() + (w.running ? 1 : 0);
        done += w.n_done;
    }
    if (n_pending) *n_pending = pend;
    if (n_done) *n_done = done;
    return 0;
}

extern "C" int hhx_files_join(int64_t *n_failed) {
    std::vector<std::string> errs;
    for (int lane = 0; lane < N_LANES; ++lane) {
        Writer &w = writer(lane);
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv_done.wait(lk, [&] { return w.q.empty() && !w.running; });
        errs.insert(errs.end(), w.errors.begin(), w.errors.end());
        w.errors.clear();
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
// MI355X boxes takes ~4.5 GB/s into one file — 15 s on the critical path of a stage that tokenises the text in 3 s.  Nothing in run() reads the file.
// A byte sink keeps the bytes WHERE THEY ARE MADE, in HBM (288 GB: the 67 GB fit beside everything else): the producer reserves room in a ring of
// device slabs (hhx_byte_sink_reserve: blocks only when `hbm_budget` bytes are waiting, i.e. at the writer's pace beyond that), its kernel
// formats into the slab, and hhx_byte_sink_commit queues the range on the file-writer thread, which copies it out through pinned buffers and writes
// it.  The slabs come from the library's pool — taken ahead of use by a helper thread (a fresh block costs ~30 ms per GB) — and go back to it when
// their last range is on disk.  hhx_byte_sink_close queues the close; the handle is freed by that job.
struct hhx_byte_sink {
    struct Slab { unsigned char *p = nullptr; size_t cap = 0, used = 0; int outstanding = 0; bool sealed = false; };
    std::string path;
    FileSink out;                        // used by the writer thread only
    bool opened = false, failed = false;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Slab *> all;
    std::deque<Slab *> free_;
    Slab *cur = nullptr;
    size_t slab_bytes = 0;
    i64 budget = 0, allocated = 0, target = 0, pushed = 0;
    bool stop_alloc = false, alloc_failed = false;
    int device = 0;
    std::thread allocator;
    std::string alloc_err;

    void alloc_loop() {
        (void)hipSetDevice(device);
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop_alloc || (free_.size() < 2 && allocated + (i64)slab_bytes <= target); });
                if (stop_alloc) return;
            }
            void *p = pool_alloc(slab_bytes);                        // the callers' list (g_arena is null on this thread): the first user of the block is the caller's kernel
            std::lock_guard<std::mutex> lk(mu);
            if (!p) { alloc_failed = true; alloc_err = g_err; cv.notify_all(); return; }
            Slab *s = new Slab();
            s->p = (unsigned char *)p; s->cap = slab_bytes;
            all.push_back(s); free_.push_back(s);
            allocated += (i64)slab_bytes;
            cv.notify_all();
        }
    }
};

extern "C" int hhx_byte_sink_open(const char *path, int64_t hbm_budget_bytes, int64_t expected_bytes, hhx_byte_sink **out) {
    if (!path || !out) return fail("hhx_byte_sink_open: null pointer");
    int dev = 0;
    HHX_HIP(hipGetDevice(&dev));
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return fail("cannot open %s for writing: %s", path, strerror(errno));
    auto *s = new hhx_byte_sink();
    s->path = path;
    s->device = dev;
    s->out.fd = fd;                      // the pinned buffers and the pwrite() threads are made by the first range's job, on the writer thread
    if (hbm_budget_bytes <= 0) {
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); t = (size_t)64 << 30; }
        hbm_budget_bytes = (i64)(t / 4);                             // a quarter of the device
        const i64 env = tune_get("bed_hbm_gb", 0);
        if (env > 0) hbm_budget_bytes = env << 30;
    }
    const size_t G64 = (size_t)64 << 20;
    size_t slab = expected_bytes > 0 ? ((size_t)expected_bytes + G64 - 1) / G64 * G64 : (size_t)1 << 30;
    slab = std::min<size_t>(std::max<size_t>(slab, G64), (size_t)4 << 30);
    s->slab_bytes = slab;
    s->budget = std::max<i64>(hbm_budget_bytes, 2 * (i64)slab);
    s->target = std::min<i64>(s->budget, std::max<i64>(expected_bytes > 0 ? expected_bytes + (i64)slab : s->budget, 2 * (i64)slab));
    s->allocator = std::thread([s] { s->alloc_loop(); });
    *out = s;
    return 0;
}

// room for n_bytes in the ring (16-byte aligned): waits for a slab when the current one is full
extern "C" int hhx_byte_sink_reserve(hhx_byte_sink *s, int64_t n_bytes, void **dev) {
    if (!s || !dev || n_bytes < 0) return fail("hhx_byte_sink_reserve: bad argument");
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->cur) s->cur->used = (s->cur->used + 255) & ~(size_t)255;
    if (!s->cur || s->cur->used + (size_t)n_bytes > s->cur->cap) {
        if (s->cur) {                                                // full: goes back to the ring when its last range is written
            s->cur->sealed = true;
            if (s->cur->outstanding == 0) { s->cur->used = 0; s->cur->sealed = false; s->free_.push_back(s->cur); }
            s->cur = nullptr;
        }
        if ((size_t)n_bytes > s->slab_bytes) return fail("hhx_byte_sink_reserve: %lld bytes in one piece, slabs of %zu", (long long)n_bytes, s->slab_bytes);
        if (s->allocated + (i64)s->slab_bytes <= s->budget) s->target = std::max<i64>(s->target, std::min<i64>(s->budget, s->allocated + 2 * (i64)s->slab_bytes));   // more than was expected
        s->cv.notify_all();
        s->cv.wait(lk, [&] { return !s->free_.empty() || (s->alloc_failed && s->all.empty()); });
        if (s->free_.empty()) return fail("hhx_byte_sink: no device memory for a slab: %s", s->alloc_err.c_str());
        s->cur = s->free_.front();
        s->free_.pop_front();
        s->cv.notify_all();                                           // the helper thread allocates the next one
    }
    *dev = s->cur->p + s->cur->used;
    return 0;
}

// the n_bytes at `dev` (the last reservation) are being written by a kernel in flight on the caller's stream: queue them for the file
extern "C" int hhx_byte_sink_commit(hhx_byte_sink *s, void *dev, int64_t n_bytes) {
    if (!s || !dev || n_bytes < 0) return fail("hhx_byte_sink_commit: bad argument");
    if (n_bytes == 0) return 0;
    hipEvent_t ev = nullptr;
    HHX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HHX_HIP(hipEventRecord(ev, g_stream));
    hhx_byte_sink::Slab *slab = nullptr;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        slab = s->cur;
        if (!slab || (unsigned char *)dev != slab->p + slab->used) { (void)hipEventDestroy(ev); return fail("hhx_byte_sink_commit: not the last reservation"); }
        slab->used += (size_t)n_bytes;
        ++slab->outstanding;
        s->pushed += n_bytes;
    }
    return files_submit("bytes -> " + s->path, s, [s, slab, dev, n_bytes, ev]() -> int {
        int rc = 0;
        if (!s->failed) {
            if (!s->opened) { rc = s->out.open_fd(s->out.fd); s->opened = true; }
            if (!rc && hipStreamWaitEvent(g_stream, ev, 0) != hipSuccess) rc = fail("hipStreamWaitEvent failed");
            if (!rc) rc = s->out.write_device((const unsigned char *)dev, (size_t)n_bytes);      // synchronises the stream piece by piece
            if (rc) s->failed = true;                                // the first failure is the one reported; later ranges are dropped
        }
        (void)hipStreamSynchronize(g_stream);
        (void)hipEventDestroy(ev);
        void *give_back = nullptr;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            if (--slab->outstanding == 0 && slab->sealed) {
                if (s->stop_alloc || s->free_.size() >= 2) {         // the producer has finished (or is slower than the file): the slab leaves HBM for good —
                    give_back = slab->p;                             // to the driver, not to the pool's cache, where 4 GB blocks would starve what runs next
                    s->all.erase(std::find(s->all.begin(), s->all.end(), slab));
                    s->allocated -= (i64)slab->cap;
                    delete slab;
                } else { slab->used = 0; slab->sealed = false; s->free_.push_back(slab); }
            }
        }
        if (give_back) pool_free_to_driver(give_back);
        s->cv.notify_all();
        return rc;
    }, 0);
}

extern "C" int hhx_byte_sink_close(hhx_byte_sink *s, int64_t *n_bytes_pushed) {
    if (!s) return 0;
    if (n_bytes_pushed) *n_bytes_pushed = s->pushed;
    { std::lock_guard<std::mutex> lk(s->mu); s->stop_alloc = true; if (s->cur) s->cur->sealed = true; }
    s->cv.notify_all();
    if (s->allocator.joinable()) s->allocator.join();
    return files_submit("close " + s->path, s, [s]() -> int {
        int rc = 0;
        if (s->opened) rc = s->out.close();
        else if (s->out.fd >= 0) { if (::close(s->out.fd) != 0) rc = fail("close failed: %s", strerror(errno)); s->out.fd = -1; }
        (void)hipStreamSynchronize(g_stream);
        for (auto *slab : s->all) { pool_free_to_driver(slab->p); delete slab; }      // every range is on disk: idle on both streams
        const bool failed = s->failed;
        delete s;
        return failed ? 0 : rc;                                       // a failed range has already been reported
    }, 0);
}
