/*
 * multi_passes.c - ed25519_verify_batch_multi_dev with MANY passes per device.
 *
 * The single-process multi-device form enqueues the devices' work from one host thread, device after device, then one
 * grouped gather.  A host-side wait anywhere inside a device's enqueue would serialise the devices: device d + 1 would
 * start only when device d had finished something.  The candidates are in csrc/eddsa_amd.c: verify_on() runs every pass
 * of a shard through ONE workspace slot (the slot a stream used last), in stream order - it waits on the host only to
 * GROW a slot (first use) or when every slot is held by a batch verification; the pool of four bounds how many STREAMS
 * overlap on a device, not how many passes one stream may queue.
 *
 * Built against tests/fake_hip/ with CHUNK_MAX = 64 items (a pass of the product is 2^20): shards of 6 x 64 + 5 items
 * are seven passes per device, beyond the pool's four.  Run with FAKE_HIP_DEFER=1: the fake runtime queues asynchronous
 * work and runs it only when something FORCES it (fake_hip_tasks_run counts those), so after a warm call has sized the
 * workspaces, the call under test must return with the count unchanged - nothing waited, on any device - and with every
 * device's queue non-empty; the verdicts are compared after the caller's own synchronisation.
 *
 *   multi_passes <ed25519_table.bin>       exit status 0 = all checks passed
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "eddsa.h"
#include "eddsa_amd.h"
#include "fake_hip.h"

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "multi_passes: " __VA_ARGS__); fputc('\n', stderr); return 1; } } while (0)
#define RC(call) do { int rc_ = (call); CHECK(rc_ == 0, "%s: %s", #call, eddsa_amd_strerror(rc_)); } while (0)
#define HIP(call) do { hipError_t e_ = (call); CHECK(e_ == hipSuccess, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: %s ed25519_table.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    static uint8_t et[128 * 1024];
    CHECK(fread(et, 1, sizeof(et), f) == sizeof(et), "short table");
    fclose(f);
    CHECK(fake_hip_deferred(), "run with FAKE_HIP_DEFER=1: the eager model cannot tell whether the host waited");

    RC(eddsa_amd_init_devices(NULL, 0));
    const int g = eddsa_amd_device_count();
    CHECK(g >= 2, "needs FAKE_HIP_DEVICES >= 2");
    const size_t per = 6 * 64 + 5, nt = (size_t)g * per, mlen = 24;     /* seven passes of <= 64 items per device */
    uint8_t *fs = malloc(32 * nt), *fp = malloc(32 * nt), *fm = malloc(mlen * nt), *fsig = malloc(64 * nt), *want = malloc(nt);
    for (size_t i = 0; i < nt; i++) {
        memcpy(fs + 32 * i, et + 128 * (i % 1024), 32);
        fs[32 * i + 9] ^= (uint8_t)(i / 1024 + 1);
        for (size_t b = 0; b < mlen; b++) fm[mlen * i + b] = (uint8_t)(i * 5 + b * 11);
    }
    RC(ed25519_genpub_batch(fp, fs, nt));
    RC(ed25519_sign_batch(fsig, fs, fp, fm, NULL, mlen, nt));
    for (size_t i = 0; i < nt; i += 5) fsig[64 * i + 7] ^= 0x10;
    RC(ed25519_verify_batch(want, fsig, fp, fm, NULL, mlen, nt));

    uint8_t *d_ok[64], *d_sig[64], *d_pub[64], *d_msg[64];
    void *streams[64];
    for (int d = 0; d < g; d++) {
        HIP(hipSetDevice(eddsa_amd_device_at(d)));
        HIP(hipMalloc((void **)&d_ok[d], nt));
        HIP(hipMalloc((void **)&d_sig[d], 64 * per));
        HIP(hipMalloc((void **)&d_pub[d], 32 * per));
        HIP(hipMalloc((void **)&d_msg[d], mlen * per));
        HIP(hipMemcpy(d_sig[d], fsig + 64 * per * (size_t)d, 64 * per, hipMemcpyHostToDevice));
        HIP(hipMemcpy(d_pub[d], fp + 32 * per * (size_t)d, 32 * per, hipMemcpyHostToDevice));
        HIP(hipMemcpy(d_msg[d], fm + mlen * per * (size_t)d, mlen * per, hipMemcpyHostToDevice));
        HIP(hipStreamCreateWithFlags((hipStream_t *)&streams[d], hipStreamNonBlocking));
    }
    for (int round = 0; round < 3; round++) {      /* round 0 sizes the workspaces (that does wait: hipMalloc, the tables); 1 and 2 are the test */
        for (int d = 0; d < g; d++) { HIP(hipSetDevice(eddsa_amd_device_at(d))); HIP(hipMemset(d_ok[d], 0xee, nt)); HIP(hipDeviceSynchronize()); }
        const long before = fake_hip_tasks_run();
        RC(ed25519_verify_batch_multi_dev(d_ok, (const uint8_t *const *)d_sig, (const uint8_t *const *)d_pub,
                                          (const uint8_t *const *)d_msg, mlen, nt, streams));
        const long forced = fake_hip_tasks_run() - before;
        if (round > 0)
            CHECK(forced == 0, "round %d: %ld queued tasks were forced to run inside the call: the host waited between the devices' passes", round, forced);
        for (int d = 0; d < g; d++) {
            uint8_t *got = malloc(nt);
            HIP(hipSetDevice(eddsa_amd_device_at(d)));
            const long t0 = fake_hip_tasks_run();
            HIP(hipStreamSynchronize((hipStream_t)streams[d]));
            if (round > 0 && d == 0) CHECK(fake_hip_tasks_run() - t0 >= 7, "device 0 had only %ld tasks queued after seven passes (the fake launcher queues one per pass)", fake_hip_tasks_run() - t0);
            HIP(hipMemcpy(got, d_ok[d], nt, hipMemcpyDeviceToHost));
            CHECK(memcmp(got, want, nt) == 0, "round %d, device %d: gathered verdicts differ from the single-device ones", round, d);
            free(got);
        }
    }
    for (int d = 0; d < g; d++) {
        HIP(hipSetDevice(eddsa_amd_device_at(d)));
        HIP(hipStreamDestroy((hipStream_t)streams[d]));
        HIP(hipFree(d_ok[d])); HIP(hipFree(d_sig[d])); HIP(hipFree(d_pub[d])); HIP(hipFree(d_msg[d]));
    }
    eddsa_amd_shutdown();
    CHECK(fake_hip_live_allocations() == 0, "%ld allocations left in the fake runtime", fake_hip_live_allocations());
    free(fs); free(fp); free(fm); free(fsig); free(want);
    printf("multi_passes: ok (%d devices, seven passes of <= 64 items each per call, no task forced inside the calls)\n", g);
    return 0;
}
