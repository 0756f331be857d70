/*
 * threaded_callers.c - a threaded application written against eddsa.h only: T threads, each looping over the
 * single-item functions of the reference's public header (reference lib/eddsa.h:44-80; the reference is
 * reentrant and scales with its caller's threads).  Behind libeddsa_amd.so the concurrent calls of one
 * operation are merged into one GPU launch (host_pipe.c: the combiner); every caller must still get exactly its
 * own result.  Checks every result against the golden table and prints the aggregate rates.
 *
 * usage: threaded_callers <ed25519_table.bin> <ed25519_msgs.bin> <x25519_table.bin> [threads [iterations [entries [trace]]]]
 *        (entries: only the first so many table entries are used, i.e. messages shorter than that; default all 1024;
 *         trace: also print the host-side time stamps of the last combined launch of the verify loop)
 * exit status 0 = every result was right.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "eddsa.h"
#include "eddsa_amd.h"
#include "eddsa_amd_debug.h"      /* combiner statistics, pipeline trace: the measurement surface */

static uint8_t *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    *len = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *p = malloc(*len ? *len : 1);
    if (fread(p, 1, *len, f) != *len) { perror("fread"); exit(2); }
    fclose(f);
    return p;
}

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static const uint8_t *g_et, *g_msgs, *g_xt;
static int g_iters, g_mode, g_entries = 1024;             /* mode 0: every thread verifies; 1: threads take different operations */
static pthread_barrier_t g_start;

struct worker { int id; long done, wrong; };

/* entry i of the ed25519 table: sk | pk | sig, message of i bytes at offset i (i - 1) / 2 */
static void *work(void *arg)
{
    struct worker *w = (struct worker *)arg;
    uint8_t out[64];
    pthread_barrier_wait(&g_start);
    for (int it = 0; it < g_iters; it++) {
        const size_t i = (size_t)((w->id * 131 + it * 7) % g_entries);
        const uint8_t *sk = g_et + 128 * i, *pk = sk + 32, *sig = sk + 64, *m = g_msgs + i * (i - 1) / 2;
        const int op = g_mode == 0 ? 0 : w->id % 4;
        if (op == 0) {                  /* a genuine signature, then the same with one bit flipped */
            if (!ed25519_verify(sig, pk, m, i)) w->wrong++;
            if (i) { memcpy(out, sig, 64); out[(it % 64)] ^= 1u << (it % 8); if (ed25519_verify(out, pk, m, i)) w->wrong++; w->done++; }
        } else if (op == 1) {
            ed25519_sign(out, sk, pk, m, i);
            if (memcmp(out, sig, 64) != 0) w->wrong++;
        } else if (op == 2) {
            const uint8_t *pt = g_xt + 96 * i, *sc = pt + 32, *res = pt + 64;
            x25519(out, sc, pt);
            if (memcmp(out, res, 32) != 0) w->wrong++;
        } else {
            ed25519_genpub(out, sk);
            if (memcmp(out, pk, 32) != 0) w->wrong++;
        }
        w->done++;
    }
    return NULL;
}

static int run(int threads, const char *what)
{
    pthread_t *th = malloc(sizeof(pthread_t) * (size_t)threads);
    struct worker *ws = calloc((size_t)threads, sizeof(*ws));
    long done = 0, wrong = 0;
    pthread_barrier_init(&g_start, NULL, (unsigned)threads + 1);
    for (int t = 0; t < threads; t++) { ws[t].id = t; pthread_create(&th[t], NULL, work, &ws[t]); }
    pthread_barrier_wait(&g_start);
    const double t0 = now();
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); done += ws[t].done; wrong += ws[t].wrong; }
    const double dt = now() - t0;
    uint64_t st[2] = { 0, 0 };
    eddsa_amd_combiner_stats(st);
    printf("threaded_callers: %s: %d threads, %ld calls in %.3f s = %.0f calls/s, %ld wrong; %llu launches carried %llu calls so far\n",
           what, threads, done, dt, (double)done / dt, wrong, (unsigned long long)st[0], (unsigned long long)st[1]);
    pthread_barrier_destroy(&g_start);
    free(th); free(ws);
    return wrong != 0;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s ed25519_table.bin ed25519_msgs.bin x25519_table.bin [threads [iterations [entries]]]\n", argv[0]); return 2; }
    size_t el, ml, xl;
    g_et = slurp(argv[1], &el); g_msgs = slurp(argv[2], &ml); g_xt = slurp(argv[3], &xl);
    if (el != 1024 * 128 || ml != 1024 * 1023 / 2 || xl != 1024 * 96) { fprintf(stderr, "threaded_callers: unexpected table sizes\n"); return 2; }
    const int threads = argc > 4 ? atoi(argv[4]) : 64;
    g_iters = argc > 5 ? atoi(argv[5]) : 200;
    if (argc > 6 && atoi(argv[6]) > 0 && atoi(argv[6]) <= 1024) g_entries = atoi(argv[6]);
    uint8_t warm[32];
    ed25519_genpub(warm, g_et);         /* builds the engine (tables) before the clock starts */
    int bad = 0;
    const int trace = argc > 7;
    int tags[64]; unsigned chunks[64]; double ms[64];
    g_mode = 0;
    bad |= run(1, "ed25519_verify, one caller");
    if (trace) eddsa_amd_debug_pipe_trace(2, tags, chunks, ms, 0);   /* the 20th launch that carries 32 calls or more */
    bad |= run(threads, "ed25519_verify");
    if (trace) {
        const int k = eddsa_amd_debug_pipe_trace(0, tags, chunks, ms, 64);
        for (int i = 0; i < k; i++) printf("  stamp %d  %8.3f ms\n", tags[i], ms[i]);
    }
    g_mode = 1;
    bad |= run(threads, "verify / sign / x25519 / genpub by thread");
    eddsa_amd_shutdown();
    free((void *)g_et); free((void *)g_msgs); free((void *)g_xt);
    if (bad) { fprintf(stderr, "threaded_callers: WRONG RESULTS\n"); return 1; }
    printf("threaded_callers: ok\n");
    return 0;
}
