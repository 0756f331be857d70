/*
 * host_side_stress.c - the host side of libeddsa_amd (csrc/host_pipe.c, csrc/eddsa_amd.c) under load, written to be run
 * under the sanitizers against tests/fake_hip/ in the build container (it also runs against the real library):
 *
 *   1. multi-chunk pipelines on small batches (eddsa_amd_set_pipeline(48, 96): three lanes, drains, secrets wiped) for
 *      verify with RAGGED messages (the chunks' offset tables are rebased), the opt-in batch verification, sign and
 *      x25519 - same bytes as one chunk;
 *   2. the fault hooks: inert until armed; a failed host-pointer call and a failed HIP call inside a verify pass come
 *      back as negative values, the next call works, no secret is left in a staging buffer;
 *   3. threads that issue chunked batches and single-item calls (merged by the combiner) while another thread switches the
 *      pipeline trace on and off and reads it;
 *   4. two threads shutting the library down at once while callers are still at work, then a call that builds it anew;
 *   5. nothing failed on a clean-up path, and (fake runtime only) every allocation, stream and event was released.
 *
 *   host_side_stress <ed25519_table.bin> <ed25519_msgs.bin> <x25519_table.bin> [threads [rounds]]
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "eddsa.h"
#include "eddsa_amd.h"
#include "eddsa_amd_debug.h"

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

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "host_side_stress: " __VA_ARGS__); fputc('\n', stderr); return 1; } } while (0)
#define RC(call) do { int rc_ = (call); CHECK(rc_ == 0, "%s: %s", #call, eddsa_amd_strerror(rc_)); } while (0)

static const uint8_t *g_et, *g_msgs, *g_xt;
enum { NE = 200 };                               /* table entries used: messages of 0 .. 199 bytes */
static uint8_t g_sig[64 * NE], g_pub[32 * NE], g_sec[32 * NE], g_mm[NE * NE / 2 + NE], g_sc[32 * NE], g_pt[32 * NE], g_xr[32 * NE];
static uint64_t g_off[NE + 1];
static int g_rounds = 3;
static int g_stop;                               /* atomic */

struct worker { int id; long wrong, calls, failed_while_shut; };

static void *batch_worker(void *arg)
{
    struct worker *w = (struct worker *)arg;
    uint8_t ok[NE], sig[64 * NE], x[32 * NE], one[64];
    for (int r = 0; r < g_rounds; r++) {
        const int what = (w->id + r) % 4;
        int rc = 0;
        if (what == 0) {
            memset(ok, 7, sizeof(ok));
            rc = ed25519_verify_batch(ok, g_sig, g_pub, g_mm, g_off, 0, NE);
            for (int i = 0; !rc && i < NE; i++) w->wrong += ok[i] != 1;
        } else if (what == 1) {
            rc = ed25519_sign_batch(sig, g_sec, g_pub, g_mm, g_off, 0, NE);
            if (!rc) w->wrong += memcmp(sig, g_sig, sizeof(sig)) != 0;
        } else if (what == 2) {
            rc = x25519_batch(x, g_sc, g_pt, NE);
            if (!rc) w->wrong += memcmp(x, g_xr, sizeof(x)) != 0;
        } else {
            const size_t i = (size_t)((w->id * 37 + r * 11) % NE);      /* single-item calls: the combiner */
            if (!ed25519_verify(g_sig + 64 * i, g_pub + 32 * i, g_mm + g_off[i], i)) w->wrong++;
            memcpy(one, g_sig + 64 * i, 64);
            one[r % 64] ^= 0x10;
            if (i && ed25519_verify(one, g_pub + 32 * i, g_mm + g_off[i], i)) w->wrong++;
            ed25519_genpub(one, g_sec + 32 * i);
            w->wrong += memcmp(one, g_pub + 32 * i, 32) != 0;
        }
        if (rc) { w->failed_while_shut++; fprintf(stderr, "host_side_stress: thread %d, operation %d: %s\n", w->id, what, eddsa_amd_strerror(rc)); }          /* (never expected: a call either runs before a shutdown or builds the engine anew) */
        w->calls++;
    }
    return NULL;
}

static void *trace_toggler(void *arg)
{
    (void)arg;
    int tags[64]; unsigned chunks[64]; double ms[64];
    for (int k = 0; !__atomic_load_n(&g_stop, __ATOMIC_ACQUIRE); k++) {
        eddsa_amd_debug_pipe_trace(k % 3, tags, chunks, ms, 64);
        struct timespec ts = { 0, 200000 };
        nanosleep(&ts, NULL);
    }
    eddsa_amd_debug_pipe_trace(0, tags, chunks, ms, 64);
    return NULL;
}

static void *shutter(void *arg)
{
    (void)arg;
    eddsa_amd_shutdown();
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s ed25519_table.bin ed25519_msgs.bin x25519_table.bin [threads [rounds]]\n", argv[0]); return 2; }
    size_t el, ml, xl;
    g_et = slurp(argv[1], &el); g_msgs = slurp(argv[2], &ml); g_xt = slurp(argv[3], &xl);
    CHECK(el == 1024 * 128 && ml == 1024 * 1023 / 2 && xl == 1024 * 96, "unexpected table sizes");
    const int threads = argc > 4 ? atoi(argv[4]) : 8;
    if (argc > 5) g_rounds = atoi(argv[5]);
    size_t pos = 0;
    for (size_t i = 0; i < NE; i++) {
        memcpy(g_sec + 32 * i, g_et + 128 * i, 32); memcpy(g_pub + 32 * i, g_et + 128 * i + 32, 32); memcpy(g_sig + 64 * i, g_et + 128 * i + 64, 64);
        g_off[i] = pos;
        memcpy(g_mm + pos, g_msgs + i * (i - 1) / 2, i);
        pos += i;
        memcpy(g_pt + 32 * i, g_xt + 96 * i, 32); memcpy(g_sc + 32 * i, g_xt + 96 * i + 32, 32); memcpy(g_xr + 32 * i, g_xt + 96 * i + 64, 32);
    }
    g_off[NE] = pos;

    /* ---- 2a. the hooks are inert until armed ---- */
    RC(eddsa_amd_init(0));
    CHECK(eddsa_amd_debug_fail_next_host_call() == EDDSA_AMD_HOOKS_OFF && eddsa_amd_debug_fail_hip_call(1) == EDDSA_AMD_HOOKS_OFF, "unarmed hooks acted");
    uint8_t ok[NE], ok1[NE], sig[64 * NE], x[32 * NE];
    RC(ed25519_verify_batch(ok1, g_sig, g_pub, g_mm, g_off, 0, NE));          /* one chunk */
    for (int i = 0; i < NE; i++) CHECK(ok1[i] == 1, "golden signature %d rejected", i);

    /* ---- 1. multi-chunk pipelines on small batches ---- */
    RC(eddsa_amd_debug_init(0, EDDSA_AMD_TEST_HOOKS));
    eddsa_amd_set_pipeline(48, 96);
    uint8_t bad[64 * NE];
    memcpy(bad, g_sig, sizeof(bad));
    for (int i = 0; i < NE; i += 3) bad[64 * i + 40] ^= 2;
    /* an offset table that does not start at 0: the messages sit 5 bytes into their buffer */
    uint8_t *shifted = malloc(sizeof(g_mm) + 5);
    uint64_t off5[NE + 1];
    memcpy(shifted + 5, g_mm, sizeof(g_mm));
    for (int i = 0; i <= NE; i++) off5[i] = g_off[i] + 5;
    /* (the secret-bearing operations first, on a fresh engine: the residue counters look at whole staging buffers, and a
     * verify call leaves its - public - signatures in them) */
    eddsa_amd_shutdown();
    RC(eddsa_amd_debug_init(0, EDDSA_AMD_TEST_HOOKS));
    uint64_t res[4];
    RC(x25519_batch(x, g_sc, g_pt, NE));
    CHECK(memcmp(x, g_xr, sizeof(x)) == 0, "chunked x25519 differs from the reference's table");
    RC(eddsa_amd_secret_residue(res));
    CHECK(res[0] == 0 && res[1] == 0 && res[2] == 0 && res[3] == 0, "x25519 left secrets behind: %llu %llu %llu %llu",
          (unsigned long long)res[0], (unsigned long long)res[1], (unsigned long long)res[2], (unsigned long long)res[3]);
    RC(ed25519_sign_batch(sig, g_sec, g_pub, shifted, off5, 0, NE));
    CHECK(memcmp(sig, g_sig, sizeof(sig)) == 0, "chunked ragged sign differs from the golden signatures");
    RC(eddsa_amd_secret_residue(res));           /* (its output, the signatures, is public and may stay) */
    CHECK(res[0] == 0 && res[2] == 0, "sign left secrets behind: %llu %llu", (unsigned long long)res[0], (unsigned long long)res[2]);
    RC(ed25519_verify_batch(ok, bad, g_pub, shifted, off5, 0, NE));
    for (int i = 0; i < NE; i++) CHECK(ok[i] == (i % 3 != 0), "chunked ragged verify: item %d", i);
    {   /* the opt-in batch verification through the same lanes: its statistics are summed across the chunks */
        uint32_t stats[4] = { 9, 9, 9, 9 };
        eddsa_amd_set_rlc_min_items(1);
        memset(ok, 7, sizeof(ok));
        RC(ed25519_verify_batch_rlc(ok, stats, bad, g_pub, shifted, off5, 0, NE));
        for (int i = 0; i < NE; i++) CHECK(ok[i] == (i % 3 != 0), "chunked ragged batch verification: item %d", i);
        CHECK(stats[0] + stats[1] == NE, "batch verification statistics: %u + %u items", stats[0], stats[1]);
        eddsa_amd_set_rlc_min_items(EDDSA_AMD_RLC_MIN_ITEMS_DEFAULT);
    }
    {   /* an offset table that runs backwards, in the first chunk and in a later one, and one whose span no buffer holds: the
         * call is refused (hipErrorInvalidValue = 1) before the chunk's bytes are touched - the sanitizers would see the
         * read of msg_off[k + 1] - msg_off[k] ~ 2^64 bytes - and the next call works */
        uint64_t offb[NE + 1];
        const int spots[3] = { 7, NE / 2, NE - 2 };
        for (int sidx = 0; sidx < 3; sidx++) {
            memcpy(offb, off5, sizeof(offb));
            offb[spots[sidx]] = offb[spots[sidx] + 1] + 3;           /* item spots[sidx] would have a negative length */
            CHECK(ed25519_verify_batch(ok, bad, g_pub, shifted, offb, 0, NE) == -1, "a decreasing offset table was accepted (verify, entry %d)", spots[sidx]);
            CHECK(ed25519_sign_batch(sig, g_sec, g_pub, shifted, offb, 0, NE) == -1, "a decreasing offset table was accepted (sign, entry %d)", spots[sidx]);
            RC(eddsa_amd_secret_residue(res));   /* (the chunks before the refused one have run: their secrets are gone like any failed call's) */
            CHECK(res[0] == 0 && res[2] == 0, "a refused sign call left secrets behind: %llu %llu", (unsigned long long)res[0], (unsigned long long)res[2]);
            CHECK(ed25519_verify_batch_rlc(ok, NULL, bad, g_pub, shifted, offb, 0, NE) == -1, "a decreasing offset table was accepted (batch verification)");
        }
        memcpy(offb, off5, sizeof(offb));
        offb[NE] = (uint64_t)1 << 62;
        CHECK(ed25519_verify_batch(ok, bad, g_pub, shifted, offb, 0, NE) == -1, "an offset table spanning 2^62 bytes was accepted");
        RC(ed25519_verify_batch(ok, bad, g_pub, shifted, off5, 0, NE));
        for (int i = 0; i < NE; i++) CHECK(ok[i] == (i % 3 != 0), "verify after refused calls: item %d", i);
    }
    free(shifted);

    /* ---- 2b. armed: a failed call is an error return, the next one works, nothing secret stays ---- */
    eddsa_amd_shutdown();
    RC(eddsa_amd_debug_init(0, EDDSA_AMD_TEST_HOOKS));
    CHECK(eddsa_amd_debug_fail_next_host_call() == 0, "hook not armed");
    CHECK(x25519_batch(x, g_sc, g_pt, NE) < 0, "the forced failure went unnoticed");
    RC(eddsa_amd_secret_residue(res));
    CHECK(res[2] == 0 && res[3] == 0, "a failed call left secrets in its staging buffers");
    RC(x25519_batch(x, g_sc, g_pt, NE));
    CHECK(memcmp(x, g_xr, sizeof(x)) == 0, "x25519 after a failed call");
    CHECK(eddsa_amd_debug_fail_hip_call(1) == 0, "hook not armed");
    CHECK(ed25519_verify_batch(ok, g_sig, g_pub, g_mm, g_off, 0, NE) < 0, "a failed HIP call inside the pass went unnoticed");
    CHECK(eddsa_amd_debug_fail_hip_call(0) == 0, "disarm");
    RC(ed25519_verify_batch(ok, g_sig, g_pub, g_mm, g_off, 0, NE));
    CHECK(memcmp(ok, ok1, NE) == 0, "verify after a failed pass");

    /* ---- 3. threads: chunked batches, combined single-item calls, the trace switched on and off meanwhile ---- */
    pthread_t th[64], tog;
    struct worker ws[64];
    const int T = threads > 64 ? 64 : threads;
    memset(ws, 0, sizeof(ws));
    pthread_create(&tog, NULL, trace_toggler, NULL);
    for (int t = 0; t < T; t++) { ws[t].id = t; pthread_create(&th[t], NULL, batch_worker, &ws[t]); }
    long wrong = 0, calls = 0, failed = 0;
    for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); wrong += ws[t].wrong; calls += ws[t].calls; failed += ws[t].failed_while_shut; }
    __atomic_store_n(&g_stop, 1, __ATOMIC_RELEASE);
    pthread_join(tog, NULL);
    CHECK(wrong == 0 && failed == 0, "%ld wrong results, %ld failed calls under load", wrong, failed);

    /* ---- 4. two shutdowns at once while callers are at work ---- */
    __atomic_store_n(&g_stop, 0, __ATOMIC_RELEASE);
    memset(ws, 0, sizeof(ws));
    for (int t = 0; t < T; t++) { ws[t].id = t; pthread_create(&th[t], NULL, batch_worker, &ws[t]); }
    pthread_t s1, s2;
    pthread_create(&s1, NULL, shutter, NULL);
    pthread_create(&s2, NULL, shutter, NULL);
    pthread_join(s1, NULL); pthread_join(s2, NULL);
    for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); wrong += ws[t].wrong; calls += ws[t].calls; failed += ws[t].failed_while_shut; }
    CHECK(wrong == 0 && failed == 0, "%ld wrong results, %ld failed calls around the shutdowns", wrong, failed);
    RC(x25519_batch(x, g_sc, g_pt, NE));                         /* builds the engine anew */
    CHECK(memcmp(x, g_xr, sizeof(x)) == 0, "x25519 after shutdown");

    /* ---- 5. clean-up paths ---- */
    eddsa_amd_shutdown();
    int first = 0;
    CHECK(eddsa_amd_debug_teardown_errors(&first) == 0, "a HIP call failed on a clean-up path: %d", first);
    long (*live)(void) = (long (*)(void))dlsym(RTLD_DEFAULT, "fake_hip_live_allocations");
    if (live) CHECK(live() == 0, "%ld allocations / streams / events outlive eddsa_amd_shutdown", live());
    free((void *)g_et); free((void *)g_msgs); free((void *)g_xt);
    printf("host_side_stress: ok (%d threads, %ld calls%s)\n", T, calls, live ? ", nothing left allocated in the fake runtime" : "");
    return 0;
}
