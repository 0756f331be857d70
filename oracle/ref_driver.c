/*
 * ref_driver.c - TEST INFRASTRUCTURE ONLY: a pthread loop over the REFERENCE's own exported
 * functions (oracle/_ref/libeddsa_ref.so, compiled from the reference sources in place).  Used by
 * bench.py's cpu_baseline leg (kind "reference") and by the oracle-pinning tests; never by the
 * product.  Contains no arithmetic of its own.
 */
#include <pthread.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

/* reference lib/eddsa.h:44-80 */
extern void ed25519_genpub(uint8_t pub[32], const uint8_t sec[32]);
extern void ed25519_sign(uint8_t sig[64], const uint8_t sec[32], const uint8_t pub[32], const uint8_t *data, size_t len);
extern bool ed25519_verify(const uint8_t sig[64], const uint8_t pub[32], const uint8_t *data, size_t len);
extern void x25519(uint8_t out[32], const uint8_t scalar[32], const uint8_t point[32]);

struct job { int kind; size_t lo, hi, msg_len; uint8_t *o; const uint8_t *a, *b, *c; };

static void *run(void *arg)
{
    struct job *j = (struct job *)arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        switch (j->kind) {
        case 0: j->o[i] = ed25519_verify(j->a + 64 * i, j->b + 32 * i, j->c + j->msg_len * i, j->msg_len); break;
        case 1: x25519(j->o + 32 * i, j->a + 32 * i, j->b + 32 * i); break;
        case 2: ed25519_sign(j->o + 64 * i, j->a + 32 * i, j->b + 32 * i, j->c + j->msg_len * i, j->msg_len); break;
        case 3: ed25519_genpub(j->o + 32 * i, j->a + 32 * i); break;
        }
    }
    return NULL;
}

static void spread(struct job proto, size_t n, int threads)
{
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    struct job *jobs = (struct job *)malloc(sizeof(struct job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = proto;
        jobs[t].lo = n * (size_t)t / (size_t)threads;
        jobs[t].hi = n * (size_t)(t + 1) / (size_t)threads;
        if (t > 0) pthread_create(&th[t], NULL, run, &jobs[t]);
    }
    run(&jobs[0]);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

void refdrv_verify_batch(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs, const uint8_t *msgs,
                         size_t msg_len, size_t n, int threads)
{
    struct job j = { 0, 0, 0, msg_len, ok, sigs, pubs, msgs };
    spread(j, n, threads);
}
void refdrv_x25519_batch(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n, int threads)
{
    struct job j = { 1, 0, 0, 0, out, scalars, points, NULL };
    spread(j, n, threads);
}
void refdrv_sign_batch(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                       size_t msg_len, size_t n, int threads)
{
    struct job j = { 2, 0, 0, msg_len, sigs, secs, pubs, msgs };
    spread(j, n, threads);
}
void refdrv_genpub_batch(uint8_t *pubs, const uint8_t *secs, size_t n, int threads)
{
    struct job j = { 3, 0, 0, 0, pubs, secs, NULL, NULL };
    spread(j, n, threads);
}
