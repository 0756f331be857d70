/* stage_rates.c - how fast can ordinary host memory be copied into page-locked staging memory on this box?
 * (the question behind the copier pool of libeddsa_amd/csrc/host_pipe.c)
 *   gcc -O2 -pthread -Iinclude tools/microbench/stage_rates.c -Llibeddsa_amd -leddsa_amd -Wl,-rpath,$PWD/libeddsa_amd -o /tmp/stage_rates
 * Prints GB/s of: memcpy and a non-temporal AVX2 copy, 1..8 threads, malloc -> pinned and pinned -> malloc. */
#define _GNU_SOURCE
#include <immintrin.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "eddsa_amd.h"

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

__attribute__((target("avx2"))) static void copy_nt(uint8_t *d, const uint8_t *s, size_t n)
{
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        __m256i a = _mm256_loadu_si256((const __m256i *)(s + i)), b = _mm256_loadu_si256((const __m256i *)(s + i + 32));
        __m256i c = _mm256_loadu_si256((const __m256i *)(s + i + 64)), e = _mm256_loadu_si256((const __m256i *)(s + i + 96));
        _mm256_stream_si256((__m256i *)(d + i), a); _mm256_stream_si256((__m256i *)(d + i + 32), b);
        _mm256_stream_si256((__m256i *)(d + i + 64), c); _mm256_stream_si256((__m256i *)(d + i + 96), e);
    }
    _mm_sfence();
    memcpy(d + i, s + i, n - i);
}

struct job { uint8_t *d; const uint8_t *s; size_t n; int nt; };
static void *run(void *a) { struct job *j = a; if (j->nt) copy_nt(j->d, j->s, j->n); else memcpy(j->d, j->s, j->n); return NULL; }

static double rate(uint8_t *d, const uint8_t *s, size_t n, int threads, int nt)
{
    double best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        pthread_t th[16]; struct job j[16];
        const size_t sl = (n / threads + 4095) & ~(size_t)4095;
        const double t0 = now();
        for (int t = 0; t < threads; t++) {
            size_t off = sl * t; if (off > n) off = n;
            j[t] = (struct job){ d + off, s + off, n - off < sl ? n - off : sl, nt };
            if (t) pthread_create(&th[t], NULL, run, &j[t]);
        }
        run(&j[0]);
        for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
        const double dt = now() - t0;
        if (dt < best) best = dt;
    }
    return n / best / 1e9;
}

int main(void)
{
    const size_t n = (size_t)128 << 20;
    uint8_t *m = aligned_alloc(4096, n), *m2 = aligned_alloc(4096, n), *p = eddsa_amd_host_alloc(n);
    if (!m || !m2 || !p) { fprintf(stderr, "allocation failed\n"); return 1; }
    memset(m, 1, n); memset(m2, 2, n); memset(p, 3, n);
    printf("128 MiB, best of 5, GB/s (threads are created per copy here: the pool of host_pipe.c has them waiting)\n");
    printf("%-28s", "threads");
    for (int t = 1; t <= 8; t++) printf("%7d", t);
    printf("\n");
    const char *names[] = { "malloc -> pinned, memcpy", "malloc -> pinned, nt-store", "pinned -> malloc, memcpy", "pinned -> malloc, nt-store",
                            "malloc -> malloc, memcpy", "malloc -> malloc, nt-store" };
    for (int k = 0; k < 6; k++) {
        uint8_t *d = k < 2 ? p : k < 4 ? m : m2; const uint8_t *s = k < 2 ? m : k < 4 ? p : m;
        printf("%-28s", names[k]);
        for (int t = 1; t <= 8; t++) printf("%7.1f", rate(d, s, n, t, k & 1));
        printf("\n");
    }
    for (size_t piece = (size_t)2 << 20; piece <= ((size_t)64 << 20); piece *= 2) {
        double t0 = now();
        for (size_t off = 0; off < n; off += piece) rate(p + off, m + off, piece, 5, 0);
        printf("pieces of %3zu MiB, 5 threads, memcpy: (x5 reps) %.1f GB/s\n", piece >> 20, 5.0 * n / (now() - t0) / 1e9);
    }
    eddsa_amd_host_free(p);
    return 0;
}
