/*
 * host_fault_walk.c - every fallible HIP runtime call the host side makes, failed once, one at a time.
 *
 * Runs against tests/fake_hip/ only (the fake runtime's injector: fake_hip_fail_call); against the real runtime it prints
 * "skipped".  For each operation below the walk arms ordinal k = 1, 2, 3 ... of the runtime calls that follow, makes
 * the call, for every k up to the number of runtime calls the operation makes (long uniform stretches thinned, see walk()).
 * After every armed call:
 *   - the return value is 0 (the failure was on a path the library can do without, or was retried) and the output is
 *     the reference's, or it is negative - never a crash, a hang, or success with wrong bytes;
 *   - the same call, unarmed, works at once and gives the reference's bytes (the engine is not left half-built, no
 *     lane of the pipeline is left taken, no lock is held);
 *   - for the secret-bearing operations, the staging buffers hold no secret;
 * and at the end of each walk eddsa_amd_shutdown() returns every allocation, stream and event (counted by the fake
 * runtime; AddressSanitizer's leak check covers the host heap).
 *
 * Walks: eddsa_amd_init on a fresh process state (engine construction, table generation); host-pointer verify (ragged
 * messages, three chunks through the lanes), the opt-in batch verification, sign, x25519, genpub on a warm engine and on a cold one (first use builds
 * the engine, the lanes and the workspaces inside the call); a one-item host call (the combiner's path); device-pointer
 * verify; eddsa_amd_init_devices + ed25519_verify_batch_multi + ed25519_verify_batch_multi_dev over every fake device,
 * the last two also with the fake RCCL's calls failed in turn (communicator set-up, group start / end, each collective).
 *
 *   host_fault_walk <ed25519_table.bin> <ed25519_msgs.bin> <x25519_table.bin>
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

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

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "host_fault_walk: " __VA_ARGS__); fputc('\n', stderr); return 1; } } while (0)

enum { NE = 72 };              /* table entries used: messages of 0 .. 71 bytes */
static uint8_t g_sig[64 * NE], g_bad[64 * NE], g_pub[32 * NE], g_sec[32 * NE], g_mm[NE * NE / 2 + NE], g_sc[32 * NE], g_pt[32 * NE], g_xr[32 * NE];
static uint64_t g_off[NE + 1];

/* the injector in use: the fake runtime's, or (walks of the collectives) the fake RCCL's */
static long (*f_calls)(void), (*f_live)(void), (*f_fired)(void);
static void (*f_fail)(long);
static long (*hip_calls)(void), (*hip_fired)(void), (*rccl_calls)(void), (*rccl_fired)(void);
static void (*hip_fail)(long), (*rccl_fail)(long);
static void use_hip(void) { f_calls = hip_calls; f_fired = hip_fired; f_fail = hip_fail; }
static void use_rccl(void) { f_calls = rccl_calls; f_fired = rccl_fired; f_fail = rccl_fail; }

/* the operations walked: each returns the library's value and says whether the output is the reference's */
typedef int (*op_fn)(int *right);
static int op_verify(int *right)
{
    uint8_t ok[NE];
    memset(ok, 7, sizeof(ok));
    const int rc = ed25519_verify_batch(ok, g_bad, g_pub, g_mm, g_off, 0, NE);
    *right = 1;
    for (int i = 0; i < NE; i++) *right &= ok[i] == (i % 3 != 0);
    return rc;
}
static int op_verify_rlc(int *right)             /* the opt-in batch verification's host path (its statistics travel too) */
{
    uint8_t ok[NE];
    uint32_t stats[4] = { 9, 9, 9, 9 };
    memset(ok, 7, sizeof(ok));
    const int rc = ed25519_verify_batch_rlc(ok, stats, g_bad, g_pub, g_mm, g_off, 0, NE);
    *right = stats[0] + stats[1] == NE;
    for (int i = 0; i < NE; i++) *right &= ok[i] == (i % 3 != 0);
    return rc;
}
static int op_sign(int *right)
{
    uint8_t sig[64 * NE];
    memset(sig, 7, sizeof(sig));
    const int rc = ed25519_sign_batch(sig, g_sec, g_pub, g_mm, g_off, 0, NE);
    *right = memcmp(sig, g_sig, sizeof(sig)) == 0;
    return rc;
}
static int op_x25519(int *right)
{
    uint8_t x[32 * NE];
    memset(x, 7, sizeof(x));
    const int rc = x25519_batch(x, g_sc, g_pt, NE);
    *right = memcmp(x, g_xr, sizeof(x)) == 0;
    return rc;
}
static int op_genpub(int *right)
{
    uint8_t pub[32 * NE];
    memset(pub, 7, sizeof(pub));
    const int rc = ed25519_genpub_batch(pub, g_sec, NE);
    *right = memcmp(pub, g_pub, sizeof(pub)) == 0;
    return rc;
}
static int op_verify_one(int *right)             /* one item from host memory: the combiner carries it */
{
    uint8_t ok = 7;
    const int rc = ed25519_verify_batch(&ok, g_sig + 64 * 17, g_pub + 32 * 17, g_mm + g_off[17], NULL, 17, 1);
    *right = ok == 1;
    return rc;
}
static int op_verify_dev(int *right)             /* device pointers: buffers of the test's own on device 0 */
{
    uint8_t *d_ok = NULL, *d_sig = NULL, *d_pub = NULL, *d_msg = NULL, ok[NE];
    int rc = -1;
    *right = 0;
    /* (the test's own runtime calls are counted too: a failure of one of them is the test's to report as "failed") */
    if (hipSetDevice(0) != hipSuccess) return -1;
    if (hipMalloc((void **)&d_ok, NE) == hipSuccess && hipMalloc((void **)&d_sig, 64 * NE) == hipSuccess &&
        hipMalloc((void **)&d_pub, 32 * NE) == hipSuccess && hipMalloc((void **)&d_msg, 32 * NE) == hipSuccess &&
        hipMemcpy(d_sig, g_bad, 64 * NE, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_pub, g_pub, 32 * NE, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemset(d_msg, 0, 32 * NE) == hipSuccess && hipMemset(d_ok, 7, NE) == hipSuccess) {
        /* fixed-length messages of 32 zero bytes: every signature is wrong for them, which is a verdict like any other */
        rc = ed25519_verify_batch_dev(d_ok, d_sig, d_pub, d_msg, NULL, 32, NE, NULL);
        if (rc == 0 && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(ok, d_ok, NE, hipMemcpyDeviceToHost) != hipSuccess)) rc = -1;
        if (rc == 0) { *right = 1; for (int i = 0; i < NE; i++) *right &= ok[i] == 0; }
    }
    (void)hipGetLastError();
    if (d_ok) hipFree(d_ok);
    if (d_sig) hipFree(d_sig);
    if (d_pub) hipFree(d_pub);
    if (d_msg) hipFree(d_msg);
    return rc;
}
static int g_devices;
static size_t g_multi_items = NE;                /* NE - 1: shards of unequal length, gathered by grouped broadcasts */
static int op_init_devices(int *right)
{
    int devs[16];
    for (int i = 0; i < g_devices; i++) devs[i] = i;
    const int rc = eddsa_amd_init_devices(devs, g_devices);
    *right = rc != 0 || eddsa_amd_device_count() == g_devices;
    return rc;
}
static int op_verify_multi(int *right)
{
    uint8_t ok[NE];
    memset(ok, 7, sizeof(ok));
    const int rc = ed25519_verify_batch_multi(ok, g_bad, g_pub, g_mm, g_off, 0, NE);
    *right = 1;
    for (int i = 0; i < NE; i++) *right &= ok[i] == (i % 3 != 0);
    return rc;
}
static int op_verify_multi_dev(int *right)       /* shards in device memory, verdicts gathered to every device (fake RCCL) */
{
    uint8_t *ok_full[16] = { 0 }, *sigs[16] = { 0 }, *pubs[16] = { 0 }, *msgs[16] = { 0 }, ok[NE];
    int rc = -1, built = 1;
    *right = 0;
    for (int g = 0; g < g_devices && built; g++) {
        size_t lo, hi;
        eddsa_amd_shard_bounds(g_multi_items, g, g_devices, &lo, &hi);
        built = hipSetDevice(g) == hipSuccess && hipMalloc((void **)&ok_full[g], NE) == hipSuccess && hipMalloc((void **)&sigs[g], 64 * (hi - lo) + 1) == hipSuccess &&
                hipMalloc((void **)&pubs[g], 32 * (hi - lo) + 1) == hipSuccess && hipMalloc((void **)&msgs[g], 32 * (hi - lo) + 1) == hipSuccess &&
                hipMemcpy(sigs[g], g_bad + 64 * lo, 64 * (hi - lo), hipMemcpyHostToDevice) == hipSuccess &&
                hipMemcpy(pubs[g], g_pub + 32 * lo, 32 * (hi - lo), hipMemcpyHostToDevice) == hipSuccess &&
                hipMemset(msgs[g], 0, 32 * (hi - lo)) == hipSuccess && hipMemset(ok_full[g], 7, NE) == hipSuccess;
    }
    if (built) {
        rc = ed25519_verify_batch_multi_dev(ok_full, (const uint8_t *const *)sigs, (const uint8_t *const *)pubs, (const uint8_t *const *)msgs, 32, g_multi_items, NULL);
        if (rc == 0) {
            *right = 1;
            for (int g = 0; g < g_devices; g++) {
                if (hipSetDevice(g) != hipSuccess || hipMemcpy(ok, ok_full[g], NE, hipMemcpyDeviceToHost) != hipSuccess) { rc = -1; *right = 0; break; }
                for (size_t i = 0; i < g_multi_items; i++) *right &= ok[i] == 0;
            }
        }
    }
    (void)hipGetLastError();
    for (int g = 0; g < g_devices; g++) {
        hipSetDevice(g);
        if (ok_full[g]) hipFree(ok_full[g]);
        if (sigs[g]) hipFree(sigs[g]);
        if (pubs[g]) hipFree(pubs[g]);
        if (msgs[g]) hipFree(msgs[g]);
    }
    hipSetDevice(0);
    return rc;
}

/* cold: the library is shut down before every armed call, so the call builds the engine (and whatever it needs) itself.
 * A call that makes more than DENSE_HEAD + DENSE_TAIL runtime calls (engine construction creates a thousand events in
 * one loop) is walked call by call at both ends and every STRIDE-th call in between; skip != 0: the first `skip` calls
 * are engine construction, walked on their own - here only their first and last eight, and everything after them. */
enum { DENSE_HEAD = 64, DENSE_TAIL = 32, STRIDE = 29 };
static int op_init(int *right);
static long g_init_calls;                        /* runtime calls of engine construction: walked on their own, not again inside every cold call */
static int walk(const char *name, op_fn op, int cold, int secret, op_fn before, long skip)
{
    long surfaced = 0, absorbed = 0, walked = 0;
    int right = 0;
    /* how many runtime calls the operation makes when nothing fails */
    if (cold) { eddsa_amd_shutdown(); if (before) CHECK(before(&right) == 0, "%s: the preparing call failed unarmed", name); }
    const long c0 = f_calls();
    CHECK(op(&right) == 0 && right, "%s: failed unarmed", name);
    const long total = f_calls() - c0;
    CHECK(total > 0 && total < 100000, "%s: %ld runtime calls", name, total);
    for (long k = 1; k <= total; k++) {
        if (skip ? k > 8 && k + 8 < skip : k > DENSE_HEAD && k + DENSE_TAIL <= total && k % STRIDE != 0) continue;
        if (cold) {
            eddsa_amd_shutdown();
            CHECK(f_live() == 0, "%s, before fault %ld: %ld allocations / streams / events outlive eddsa_amd_shutdown", name, k, f_live());
            if (before) CHECK(before(&right) == 0, "%s: the preparing call failed unarmed", name);
        }
        const long fired0 = f_fired();
        f_fail(f_calls() + k);
        const int rc = op(&right);
        f_fail(0);
        const int fired = f_fired() != fired0;
        CHECK(rc <= 0, "%s, fault %ld: positive return value %d", name, k, rc);
        if (rc == 0) CHECK(right, "%s, fault %ld: success reported with wrong output", name, k);
        /* (a warm call can need fewer runtime calls than the counting one did, e.g. no staging buffer left to grow) */
        if (!fired) { CHECK(rc == 0, "%s, fault %ld never fired, yet the call failed: %s", name, k, eddsa_amd_strerror(rc)); continue; }
        walked++;
        if (rc) surfaced++; else absorbed++;
        if (secret) {
            uint64_t res[4];
            if (eddsa_amd_secret_residue(res) == 0)
                CHECK(res[0] == 0 && res[2] == 0 && (secret < 2 || (res[1] == 0 && res[3] == 0)), "%s, fault %ld: secrets left in the staging buffers (%llu %llu %llu %llu)",
                      name, k, (unsigned long long)res[0], (unsigned long long)res[1], (unsigned long long)res[2], (unsigned long long)res[3]);
        }
        const int again = op(&right);
        CHECK(again == 0 && right, "%s, after fault %ld (which came back as %d): the next call %s", name, k, rc, again ? eddsa_amd_strerror(again) : "gave wrong output");
    }
    CHECK(surfaced > 0, "%s: no failure ever surfaced", name);
    printf("host_fault_walk: %-28s %5ld runtime calls, %4ld of them failed in turn: %ld came back as errors, %ld were absorbed\n", name, total, walked, surfaced, absorbed);
    if (op == op_init) g_init_calls = total;
    return 0;
}

static int op_init(int *right) { *right = 1; return eddsa_amd_debug_init(0, EDDSA_AMD_TEST_HOOKS); }

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s ed25519_table.bin ed25519_msgs.bin x25519_table.bin\n", argv[0]); return 2; }
    hip_calls = (long (*)(void))dlsym(RTLD_DEFAULT, "fake_hip_calls");
    f_live = (long (*)(void))dlsym(RTLD_DEFAULT, "fake_hip_live_allocations");
    hip_fired = (long (*)(void))dlsym(RTLD_DEFAULT, "fake_hip_faults_fired");
    hip_fail = (void (*)(long))dlsym(RTLD_DEFAULT, "fake_hip_fail_call");
    if (!hip_calls || !f_live || !hip_fired || !hip_fail) { printf("host_fault_walk: skipped (not the fake runtime)\n"); return 0; }
    void *rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);        /* the fake one of tests/fake_hip/_build (LD_LIBRARY_PATH): the product loads the same */
    rccl_calls = rccl ? (long (*)(void))dlsym(rccl, "fake_rccl_calls") : NULL;
    rccl_fired = rccl ? (long (*)(void))dlsym(rccl, "fake_rccl_faults_fired") : NULL;
    rccl_fail = rccl ? (void (*)(long))dlsym(rccl, "fake_rccl_fail_call") : NULL;
    CHECK(rccl_calls && rccl_fired && rccl_fail, "librccl.so.1 is not the fake one");
    use_hip();
    setvbuf(stdout, NULL, _IOLBF, 0);
    size_t el, ml, xl;
    const uint8_t *et = slurp(argv[1], &el), *mm = slurp(argv[2], &ml), *xt = slurp(argv[3], &xl);
    CHECK(el == 1024 * 128 && ml == 1024 * 1023 / 2 && xl == 1024 * 96, "unexpected table sizes");
    size_t pos = 0;
    for (size_t i = 0; i < NE; i++) {
        memcpy(g_sec + 32 * i, et + 128 * i, 32); memcpy(g_pub + 32 * i, et + 128 * i + 32, 32); memcpy(g_sig + 64 * i, et + 128 * i + 64, 64);
        g_off[i] = pos;
        memcpy(g_mm + pos, mm + i * (i - 1) / 2, i);
        pos += i;
        memcpy(g_pt + 32 * i, xt + 96 * i, 32); memcpy(g_sc + 32 * i, xt + 96 * i + 32, 32); memcpy(g_xr + 32 * i, xt + 96 * i + 64, 32);
    }
    g_off[NE] = pos;
    memcpy(g_bad, g_sig, sizeof(g_bad));
    for (int i = 0; i < NE; i += 3) g_bad[64 * i + 40] ^= 2;
    hipGetDeviceCount(&g_devices);
    if (g_devices > 16) g_devices = 16;
    eddsa_amd_set_pipeline(16, 32);              /* 72 items: chunks of 16, 32, 24 through the three lanes */

    int right;
    /* engine construction */
    if (walk("eddsa_amd_init", op_init, 1, 0, NULL, 0)) return 1;
    /* warm engine: the first calls have built lanes and workspaces */
    CHECK(op_verify(&right) == 0 && right && op_sign(&right) == 0 && right && op_x25519(&right) == 0 && right && op_genpub(&right) == 0 && right &&
          op_verify_one(&right) == 0 && right && op_verify_dev(&right) == 0 && right, "unarmed calls");
    if (walk("verify, warm", op_verify, 0, 0, NULL, 0)) return 1;
    eddsa_amd_set_rlc_min_items(1);              /* the combination's route, whatever the size */
    CHECK(op_verify_rlc(&right) == 0 && right, "unarmed batch verification");
    if (walk("batch verification, warm", op_verify_rlc, 0, 0, NULL, 0)) return 1;
    if (walk("sign, warm", op_sign, 0, 1, NULL, 0)) return 1;
    if (walk("x25519, warm", op_x25519, 0, 2, NULL, 0)) return 1;
    if (walk("genpub, warm", op_genpub, 0, 1, NULL, 0)) return 1;
    if (walk("verify of one item, warm", op_verify_one, 0, 0, NULL, 0)) return 1;
    if (walk("verify, device pointers", op_verify_dev, 0, 0, NULL, 0)) return 1;
    /* cold: every armed call starts from nothing */
    if (walk("verify, cold", op_verify, 1, 0, NULL, g_init_calls)) return 1;
    if (walk("sign, cold", op_sign, 1, 0, NULL, g_init_calls)) return 1;
    if (walk("verify of one item, cold", op_verify_one, 1, 0, NULL, g_init_calls)) return 1;
    /* the device set */
    if (walk("eddsa_amd_init_devices", op_init_devices, 1, 0, NULL, 0)) return 1;
    if (walk("verify_batch_multi, cold", op_verify_multi, 1, 0, op_init_devices, 0)) return 1;
    CHECK(op_init_devices(&right) == 0 && op_verify_multi(&right) == 0 && right && op_verify_multi_dev(&right) == 0 && right, "unarmed multi-device calls");
    if (walk("verify_batch_multi, warm", op_verify_multi, 0, 0, NULL, 0)) return 1;
    if (walk("verify_batch_multi_dev", op_verify_multi_dev, 0, 0, NULL, 0)) return 1;
    g_multi_items = NE - 1;
    if (walk("  the same, unequal shards", op_verify_multi_dev, 0, 0, NULL, 0)) return 1;
    /* the collectives' own calls (ncclCommInitAll; group start, all-gather or broadcasts, group end) */
    use_rccl();
    if (walk("  RCCL calls, unequal shards", op_verify_multi_dev, 0, 0, NULL, 0)) return 1;
    g_multi_items = NE;
    if (walk("  RCCL calls, equal shards", op_verify_multi_dev, 0, 0, NULL, 0)) return 1;
    if (walk("init_devices, RCCL calls", op_init_devices, 1, 0, NULL, 0)) return 1;
    use_hip();

    eddsa_amd_shutdown();
    CHECK(f_live() == 0, "%ld allocations / streams / events outlive the last eddsa_amd_shutdown", f_live());
    free((void *)et); free((void *)mm); free((void *)xt);
    printf("host_fault_walk: ok (%d devices)\n", g_devices);
    return 0;
}
