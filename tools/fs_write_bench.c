/* How fast can ONE file be written on this box?  alignments.bed (67 GB at C3), paired_links.clm (26 GB) and the pickles are single files, and pwrite() from
 * several threads does not scale on tmpfs / ext4 (the inode's write lock).  Variants, each writing `gb` GB from a 64 MB source buffer with `thr` threads:
 *   pwrite      pwrite() at disjoint offsets
 *   mmap        ftruncate to the final size, mmap(MAP_SHARED), memcpy into disjoint regions (page faults take per-page locks, not the inode's)
 *   mmap_chunk  the same in chunks of 512 MB: ftruncate grows the file by a chunk, the chunk is mapped, copied by all threads, unmapped (the final size unknown)
 *   falloc      fallocate the whole file first (pages exist), then pwrite
 * usage: fs_write_bench <dir> <gb> <thr>          one JSON line per variant */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static char *src;
static const size_t SRC = (size_t)64 << 20;
static int fd;
static char *map;
static size_t total;
static int n_thr, mode;

typedef struct { int k; size_t lo, hi; } job_t;

static void *work(void *arg) {
    job_t *j = (job_t *)arg;
    for (size_t at = j->lo; at < j->hi; at += SRC) {
        size_t n = j->hi - at < SRC ? j->hi - at : SRC;
        if (mode == 0) { size_t done = 0; while (done < n) { ssize_t w = pwrite(fd, src + done, n - done, (off_t)(at + done)); if (w <= 0) { perror("pwrite"); exit(1); } done += (size_t)w; } }
        else memcpy(map + at, src, n);
    }
    return NULL;
}
static void run_threads(size_t lo, size_t hi) {
    pthread_t th[64]; job_t jobs[64];
    size_t span = (hi - lo + n_thr - 1) / n_thr;
    span = (span + 4095) & ~(size_t)4095;
    for (int k = 0; k < n_thr; ++k) { jobs[k].k = k; jobs[k].lo = lo + span * k < hi ? lo + span * k : hi; jobs[k].hi = lo + span * (k + 1) < hi ? lo + span * (k + 1) : hi; pthread_create(&th[k], NULL, work, &jobs[k]); }
    for (int k = 0; k < n_thr; ++k) pthread_join(th[k], NULL);
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s dir gb threads\n", argv[0]); return 2; }
    total = (size_t)atol(argv[2]) << 30; n_thr = atoi(argv[3]);
    src = malloc(SRC); for (size_t i = 0; i < SRC; ++i) src[i] = (char)(i * 131 + 7);
    char path[4096]; snprintf(path, sizeof path, "%s/fs_write_bench.tmp", argv[1]);
    const char *names[] = {"pwrite", "mmap", "mmap_chunk", "falloc"};
    for (int v = 0; v < 4; ++v) {
        fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); return 1; }
        double t0 = now(), t_prep = 0;
        if (v == 0) { mode = 0; run_threads(0, total); }
        else if (v == 1) {
            if (ftruncate(fd, (off_t)total)) { perror("ftruncate"); return 1; }
            map = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (map == MAP_FAILED) { perror("mmap"); return 1; }
            mode = 1; run_threads(0, total); munmap(map, total);
        } else if (v == 2) {
            const size_t CH = (size_t)512 << 20;
            mode = 1;
            for (size_t at = 0; at < total; at += CH) {
                size_t n = total - at < CH ? total - at : CH;
                if (ftruncate(fd, (off_t)(at + n))) { perror("ftruncate"); return 1; }
                char *m = mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)at);
                if (m == MAP_FAILED) { perror("mmap"); return 1; }
                map = m - at; run_threads(at, at + n); munmap(m, n);
            }
        } else {
            if (fallocate(fd, 0, 0, (off_t)total)) { perror("fallocate"); return 1; }
            t_prep = now() - t0;
            mode = 0; run_threads(0, total);
        }
        double t1 = now();
        close(fd);
        printf("{\"dir\": \"%s\", \"variant\": \"%s\", \"gb\": %zu, \"threads\": %d, \"seconds\": %.3f, \"GBs\": %.2f, \"prep_s\": %.3f}\n", argv[1], names[v], total >> 30, n_thr, t1 - t0,
               (double)total / (t1 - t0) / 1e9, t_prep);
        fflush(stdout);
        unlink(path);
    }
    return 0;
}
