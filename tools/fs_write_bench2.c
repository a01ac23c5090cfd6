/* Companion of fs_write_bench.c: do SEVERAL files scale where one does not?  F files of gb/F GB each, one pwrite() thread per file.
 * usage: fs_write_bench2 <dir> <gb_total> <files> */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static char *src; static const size_t SRC = (size_t)64 << 20; static size_t per_file; static const char *dir;
static void *work(void *arg) {
    long k = (long)arg; char path[4096]; snprintf(path, sizeof path, "%s/fs_write_bench2.%ld.tmp", dir, k);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644); if (fd < 0) { perror("open"); exit(1); }
    for (size_t at = 0; at < per_file; at += SRC) { size_t n = per_file - at < SRC ? per_file - at : SRC, done = 0;
        while (done < n) { ssize_t w = pwrite(fd, src + done, n - done, (off_t)(at + done)); if (w <= 0) { perror("pwrite"); exit(1); } done += (size_t)w; } }
    close(fd); unlink(path); return NULL;
}
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    dir = argv[1]; size_t total = (size_t)atol(argv[2]) << 30; int F = atoi(argv[3]); per_file = total / F;
    src = malloc(SRC); for (size_t i = 0; i < SRC; ++i) src[i] = (char)(i * 131 + 7);
    pthread_t th[64]; double t0 = now();
    for (long k = 0; k < F; ++k) pthread_create(&th[k], NULL, work, (void *)k);
    for (int k = 0; k < F; ++k) pthread_join(th[k], NULL);
    double t1 = now();
    printf("{\"dir\": \"%s\", \"variant\": \"files\", \"gb\": %zu, \"files\": %d, \"seconds\": %.3f, \"GBs\": %.2f}\n", dir, total >> 30, F, t1 - t0, (double)total / (t1 - t0) / 1e9);
    return 0;
}
