/*
 * multi_device.c - a plain C caller of the single-process multi-device entry points of eddsa_amd.h
 * (contiguous shards, host thread per device, one RCCL gather of the verdict bytes).
 * The loop being sharded is the reference's ed25519_verify, lib/ed25519-sha512.c:148-181: the
 * verdict of item i must not depend on which device verified it.
 *
 *   multi_device <ed25519_table.bin> <ed25519_msgs.bin> [entries [uneven_items]]
 *     entries: only the first so many table entries (messages shorter than that) are used, default all 1024;
 *     uneven_items: size of the device-pointer batch with unequal shards, default 2^12 - 3
 *
 * 1. binds every visible device (eddsa_amd_init_devices(NULL, 0));
 * 2. host-pointer form: the 1024 golden signatures (message i has i bytes: ragged) plus corrupted
 *    copies through ed25519_verify_batch_multi == ed25519_verify_batch; sign / x25519 forms likewise;
 * 3. device-pointer form: fixed-length items, shard d uploaded to device d with the HIP runtime API,
 *    ed25519_verify_batch_multi_dev, every device's gathered vector == the single-device verdicts - once with
 *    equal shards (the grouped in-place ncclAllGather) and once with 2^k - 3 items (unequal shards: the grouped
 *    ncclBroadcasts).
 * exit status 0 = all checks passed.
 * The same source is also built against tests/fake_hip/ (a fake HIP runtime with 2, 3 or 8 "devices" and a fake RCCL that
 * checks the single-process call pattern) and run under the sanitizers in the build container.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "eddsa.h"
#include "eddsa_amd.h"

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

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "multi_device: " __VA_ARGS__); fputc('\n', stderr); return 1; } } while (0)
#define RC(call) do { int rc_ = (call); CHECK(rc_ == 0, "%s: %s", #call, eddsa_amd_strerror(rc_)); } while (0)
#define HIP(call) do { hipError_t e_ = (call); CHECK(e_ == hipSuccess, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

int main(int argc, char **argv)
{
    if (argc < 3 || argc > 5) { fprintf(stderr, "usage: %s ed25519_table.bin ed25519_msgs.bin [entries [uneven_items]]\n", argv[0]); return 2; }
    size_t el, ml;
    uint8_t *et = slurp(argv[1], &el), *msgs = slurp(argv[2], &ml);
    CHECK(el / 128 == 1024 && ml == 1023 * 1024 / 2, "unexpected table sizes");
    const size_t ne = argc > 3 && atoi(argv[3]) > 0 && atoi(argv[3]) <= 1024 ? (size_t)atoi(argv[3]) : 1024;
    const size_t nt_uneven = argc > 4 && atoi(argv[4]) > 0 ? (size_t)atoi(argv[4]) : 4093;

    RC(eddsa_amd_init_devices(NULL, 0));
    const int g = eddsa_amd_device_count();
    CHECK(g >= 1, "no device bound");

    /* shard bounds: contiguous, balanced, covering */
    for (size_t n = 0; n < 70; n++) {
        size_t expect_lo = 0;
        for (int w = 1; w <= 9; w += 4) {
            expect_lo = 0;
            for (int r = 0; r < w; r++) {
                size_t lo, hi;
                eddsa_amd_shard_bounds(n, r, w, &lo, &hi);
                CHECK(lo == expect_lo && hi >= lo && hi - lo <= n / (size_t)w + 1, "shard_bounds(%zu, %d, %d)", n, r, w);
                expect_lo = hi;
            }
            CHECK(expect_lo == n, "shards do not cover");
        }
    }

    /* ---- host-pointer forms, ragged messages: 2 x 1024 items, the second half corrupted ---- */
    const size_t n = 2 * ne;
    uint8_t *sig = malloc(64 * n), *pub = malloc(32 * n), *sec = malloc(32 * n), *mm = malloc(2 * ml);
    uint64_t *off = malloc((n + 1) * sizeof(uint64_t));
    uint8_t *ok1 = malloc(n), *okm = malloc(n), *sig1 = malloc(64 * n), *sigm = malloc(64 * n), *x1 = malloc(32 * n), *xm = malloc(32 * n);
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t k = i % ne, len = k, src = k * (k - 1) / 2;
        memcpy(sec + 32 * i, et + 128 * k, 32);
        memcpy(pub + 32 * i, et + 128 * k + 32, 32);
        memcpy(sig + 64 * i, et + 128 * k + 64, 64);
        off[i] = pos;
        memcpy(mm + pos, msgs + src, len);
        pos += len;
        if (i >= ne) sig[64 * i + (i % 64)] ^= (uint8_t)(1u << (i % 8));
    }
    off[n] = pos;
    RC(ed25519_verify_batch(ok1, sig, pub, mm, off, 0, n));
    RC(ed25519_verify_batch_multi(okm, sig, pub, mm, off, 0, n));
    CHECK(memcmp(ok1, okm, n) == 0, "verify_batch_multi differs from verify_batch");
    size_t acc = 0;
    for (size_t i = 0; i < n; i++) acc += ok1[i];
    CHECK(acc == ne, "expected exactly the genuine half to verify, got %zu", acc);
    RC(ed25519_sign_batch(sig1, sec, pub, mm, off, 0, n));
    RC(ed25519_sign_batch_multi(sigm, sec, pub, mm, off, 0, n));
    CHECK(memcmp(sig1, sigm, 64 * n) == 0, "sign_batch_multi differs from sign_batch");
    CHECK(memcmp(sig1, et + 64, 64) == 0, "RFC 8032 TEST 1 signature");
    RC(x25519_batch(x1, sec, pub, n));
    RC(x25519_batch_multi(xm, sec, pub, n));
    CHECK(memcmp(x1, xm, 32 * n) == 0, "x25519_batch_multi differs from x25519_batch");

    /* ---- device-pointer form: fixed 24-byte messages, signatures made on the GPU; equal shards, then unequal ones ---- */
    size_t total_dev = 0;
    for (int round = 0; round < 2; round++) {
    const size_t mlen = 24, nt = round == 0 ? (size_t)g * 341 : nt_uneven;
    uint8_t *fs = malloc(32 * nt), *fp = malloc(32 * nt), *fm = malloc(mlen * nt), *fsig = malloc(64 * nt), *want = malloc(nt);
    for (size_t i = 0; i < nt; i++) {
        memcpy(fs + 32 * i, et + 128 * (i % ne), 32);
        fs[32 * i + 5] ^= (uint8_t)(i / ne + 1);
        for (size_t b = 0; b < mlen; b++) fm[mlen * i + b] = (uint8_t)(i * 7 + b * 13 + (size_t)round);
    }
    RC(ed25519_genpub_batch(fp, fs, nt));
    RC(ed25519_sign_batch(fsig, fs, fp, fm, NULL, mlen, nt));
    for (size_t i = 0; i < nt; i += 3) fsig[64 * i + 33] ^= 4;
    RC(ed25519_verify_batch(want, fsig, fp, fm, NULL, mlen, nt));
    uint8_t *d_ok[64], *d_sig[64], *d_pub[64], *d_msg[64];
    void *streams[64];
    for (int d = 0; d < g; d++) {
        size_t lo, hi;
        eddsa_amd_shard_bounds(nt, d, g, &lo, &hi);
        HIP(hipSetDevice(eddsa_amd_device_at(d)));
        HIP(hipMalloc((void **)&d_ok[d], nt));
        HIP(hipMemset(d_ok[d], 0xee, nt));
        HIP(hipMalloc((void **)&d_sig[d], 64 * (hi - lo) + 16));
        HIP(hipMalloc((void **)&d_pub[d], 32 * (hi - lo) + 16));
        HIP(hipMalloc((void **)&d_msg[d], mlen * (hi - lo) + 16));
        HIP(hipMemcpy(d_sig[d], fsig + 64 * lo, 64 * (hi - lo), hipMemcpyHostToDevice));
        HIP(hipMemcpy(d_pub[d], fp + 32 * lo, 32 * (hi - lo), hipMemcpyHostToDevice));
        HIP(hipMemcpy(d_msg[d], fm + mlen * lo, mlen * (hi - lo), hipMemcpyHostToDevice));
        HIP(hipStreamCreate((hipStream_t *)&streams[d]));
    }
    HIP(hipSetDevice(eddsa_amd_device_at(g - 1)));      /* the caller's current device is none of the call's business (and is restored) */
    RC(ed25519_verify_batch_multi_dev(d_ok, (const uint8_t *const *)d_sig, (const uint8_t *const *)d_pub,
                                      (const uint8_t *const *)d_msg, mlen, nt, streams));
    int now_dev = -1;
    HIP(hipGetDevice(&now_dev));
    CHECK(now_dev == eddsa_amd_device_at(g - 1), "the call left device %d current", now_dev);
    for (int d = 0; d < g; d++) {
        uint8_t *got = malloc(nt);
        HIP(hipSetDevice(eddsa_amd_device_at(d)));
        HIP(hipStreamSynchronize((hipStream_t)streams[d]));
        HIP(hipMemcpy(got, d_ok[d], nt, hipMemcpyDeviceToHost));
        CHECK(memcmp(got, want, nt) == 0, "device %d: gathered verdicts differ from the single-device ones (%zu items)", d, nt);
        free(got);
        HIP(hipStreamDestroy((hipStream_t)streams[d]));
        HIP(hipFree(d_ok[d])); HIP(hipFree(d_sig[d])); HIP(hipFree(d_pub[d])); HIP(hipFree(d_msg[d]));
    }
    size_t a2 = 0;
    for (size_t i = 0; i < nt; i++) a2 += want[i];
    CHECK(a2 == nt - (nt + 2) / 3, "expected every third item rejected");
    total_dev += nt;
    free(fs); free(fp); free(fm); free(fsig); free(want);
    }
    const size_t nt = total_dev;
    /* under the fake RCCL of tests/fake_hip/ (it exports fake_rccl_stats; the real library does not): both forms of the
     * result gather did run - one grouped all-gather, and one broadcast per shard for the unequal ones */
    {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        void (*stats)(long *) = h ? (void (*)(long *))dlsym(h, "fake_rccl_stats") : NULL;
        if (stats && g > 1) {
            long st[2] = { 0, 0 };
            stats(st);
            CHECK(st[0] == 1 && st[1] == g, "expected 1 all-gather and %d broadcasts, the fake RCCL ran %ld and %ld", g, st[0], st[1]);
            printf("multi_device: fake RCCL ran %ld all-gather and %ld broadcasts over %d ranks\n", st[0], st[1], g);
        }
        if (h) dlclose(h);
    }
    eddsa_amd_shutdown();
    free(sig); free(pub); free(sec); free(mm); free(off); free(ok1); free(okm); free(sig1); free(sigm); free(x1); free(xm);
    free(et); free(msgs);
    printf("multi_device: ok (%d device%s, %zu + %zu items)\n", g, g == 1 ? "" : "s", n, nt);
    return 0;
}
