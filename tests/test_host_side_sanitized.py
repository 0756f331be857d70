"""The product's HOST side - libeddsa_amd/csrc/eddsa_amd.c + host_pipe.c, unchanged - on the CPU under sanitizers, with
MORE THAN ONE device.  tests/fake_hip/ is a fake HIP runtime on host memory (2, 3 or 8
"devices", every pointer tagged with the device that owns it), CPU launchers that call the -DED_HOST_CHECK build of the
device source, and a fake RCCL that executes the grouped collectives of one process and checks the call pattern (every
rank takes part, buffers and streams on the communicator's device, the in-place rule sendbuff == recvbuff + rank * count).
Test binaries only: nothing here is linked into, or loaded by, the product.

Run under -fsanitize=thread and -fsanitize=address,undefined:
  tests/c/multi_device.c      eddsa_amd_init_devices, the *_multi host-pointer forms (a host thread per device) and
                              ed25519_verify_batch_multi_dev over 2, 3 and 8 devices - equal shards (one grouped in-place
                              all-gather) and 2^k - 3 items (unequal shards: one broadcast per shard) - every device's
                              gathered vector equal to the single-device verdicts
  tests/c/multi_passes.c      ed25519_verify_batch_multi_dev with seven passes per device (a build with passes of 64 items): in the
                              deferred model no queued task may be forced to run inside the call - no device waits on another
  tests/c/threaded_callers.c  64 threads looping over the eddsa.h single-item functions: the flat combiner
  tests/c/host_side_stress.c  multi-chunk pipelines with ragged messages, the fault hooks, the trace switched on and off
                              under load, two concurrent shutdowns beside callers, nothing leaked
  tests/c/selftest_dropin.c   (address build) the reference's selftests over eddsa.h and every batched entry point
  tests/c/host_fault_walk.c   (address build) every fallible runtime call of engine construction, of the host-pointer and
                              device-pointer calls (warm and cold), of the device set and of the gather - and every call
                              into the fake RCCL - failed once, one at a time: an error return or a right result, the next
                              call works, no secret stays in a staging buffer, nothing is leaked
Each also runs with FAKE_HIP_DEFER=1: the fake runtime then QUEUES asynchronous work per stream and runs it as late as the API
allows (only what a synchronisation or an event wait forces), so a consumer that lacks its wait reads stale bytes and the
program's own comparison with the reference fails - ordering between streams, which the eager model cannot see
(test_the_deferred_model_catches_a_missing_wait shows the checker at work).
This is the only multi-device evidence obtainable without a multi-GPU node; the launchers' own stream protocol (kernels.hip:
edk_verify's side stream) is replaced by the fake's and is tested on hardware only."""
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_hip")
GOLD = os.path.join(ROOT, "tests", "golden")


def golden_msg(i):
    out, c = b"", 0
    while len(out) < i:
        out += hashlib.sha512(b"libeddsa-amd golden msg" + i.to_bytes(4, "little") + c.to_bytes(4, "little")).digest()
        c += 1
    return out[:i]


@pytest.fixture(scope="module")
def msgs(tmp_path_factory):
    p = tmp_path_factory.mktemp("fake") / "msgs.bin"
    p.write_bytes(b"".join(golden_msg(i) for i in range(1024)))
    return str(p)


@pytest.fixture(scope="module", params=["thread", "address"])
def build(request):
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("the HIP headers (types only) are needed to compile the host side")
    san = request.param
    r = subprocess.run(["make", "-C", FAKE, "-j4", "SAN=" + san], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = os.path.join(FAKE, "_build", san)
    for exe in ("multi_device", "threaded_callers", "host_side_stress", "host_fault_walk", "selftest_dropin", "multi_passes"):
        # test binaries against the fake runtime: they must not pull in the real one
        ldd = subprocess.check_output(["ldd", os.path.join(out, exe)], text=True)
        assert "libamdhip64" not in ldd and "libfakehip.so" in ldd, ldd
    return san, out


def run(out, exe, args, devices, timeout=900, defer=0):
    env = dict(os.environ, FAKE_HIP_DEVICES=str(devices), FAKE_HIP_DEFER=str(defer), LD_LIBRARY_PATH=out,
               TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(out, exe)] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=timeout)
    text = r.stdout + r.stderr
    assert r.returncode == 0, text[-4000:]
    assert "Sanitizer" not in text and "runtime error" not in text, text[-4000:]
    return text


@pytest.mark.parametrize("devices,defer", [(2, 1), (3, 0), (8, 1)])
def test_multi_device_entry_points_over_several_devices(build, msgs, devices, defer):
    san, out = build
    # 96 table entries (messages of 0 .. 95 bytes), unequal shards from 2^8 - 3 items
    text = run(out, "multi_device", [os.path.join(GOLD, "ed25519_table.bin"), msgs, 96, 253], devices, defer=defer)
    assert f"multi_device: ok ({devices} devices" in text
    assert f"fake RCCL ran 1 all-gather and {devices} broadcasts over {devices} ranks" in text   # both forms of the gather ran


@pytest.mark.parametrize("devices", [2, 8])
def test_no_device_waits_on_another_with_many_passes_per_device(build, devices):
    """ed25519_verify_batch_multi_dev enqueues the devices' passes from ONE host thread.  With the host side
    built for passes of 64 items, shards of 389 items are seven passes per device - more than the four workspaces of a
    device's pool - and in the deferred model any host-side wait inside the call would force queued tasks to run:
    tests/c/multi_passes.c requires that none does, that every device's queue is full when the call returns, and that
    every device ends with the single-device verdicts"""
    san, out = build
    if (san, devices) in (("thread", 8), ("address", 2)):
        pytest.skip("eight devices run under ASan / UBSan, two under TSan (48 s less per suite)")
    text = run(out, "multi_passes", [os.path.join(GOLD, "ed25519_table.bin")], devices, defer=1)
    assert f"multi_passes: ok ({devices} devices" in text


@pytest.mark.parametrize("defer", [0, 1])
def test_threaded_callers_through_the_combiner(build, msgs, defer):
    san, out = build
    text = run(out, "threaded_callers", [os.path.join(GOLD, "ed25519_table.bin"), msgs, os.path.join(GOLD, "x25519_table.bin"),
                                         64, 6, 48, "trace"], 2, defer=defer)
    assert "threaded_callers: ok" in text and " 0 wrong" in text
    import re
    launches, calls = map(int, re.findall(r"(\d+) launches carried (\d+) calls so far", text)[-1])
    assert calls > 2 * launches                                   # the calls did meet: the merging code ran


@pytest.mark.parametrize("threads,defer", [(16, 1)])
def test_host_pipeline_faults_trace_and_concurrent_shutdown(build, msgs, threads, defer):
    san, out = build
    text = run(out, "host_side_stress", [os.path.join(GOLD, "ed25519_table.bin"), msgs, os.path.join(GOLD, "x25519_table.bin"),
                                         threads, 4], 2, defer=defer)
    assert "host_side_stress: ok" in text and "nothing left allocated in the fake runtime" in text


def test_the_reference_selftests_and_every_batched_entry_point(build, msgs):
    """tests/c/selftest_dropin.c (the reference's four selftests over eddsa.h, records, conversions, x25519_base: the program
    the GPU suite runs against the real library) through the unchanged host side, deferred model"""
    san, out = build
    if san != "address":
        pytest.skip("one thread")
    text = run(out, "selftest_dropin", [os.path.join(GOLD, "x25519_table.bin"), os.path.join(GOLD, "ed25519_table.bin"), msgs], 2, defer=1)
    assert "selftest_dropin: ok (1024 x25519 vectors, 1024 ed25519 vectors)" in text


def test_every_runtime_call_failed_in_turn(build, msgs):
    """error paths of the host side: ~850 injected failures of HIP runtime and RCCL calls, none of which may crash, hang,
    leak, leave a secret behind, report success with wrong bytes, or break the call that follows"""
    san, out = build
    if san != "address":
        pytest.skip("one thread: the leak and bounds checks of the address build are the point")
    # (in the deferred model: the failed call's earlier work is still queued when the error path runs)
    text = run(out, "host_fault_walk", [os.path.join(GOLD, "ed25519_table.bin"), msgs, os.path.join(GOLD, "x25519_table.bin")], 2, defer=1)
    assert "host_fault_walk: ok (2 devices)" in text
    import re
    rows = re.findall(r"host_fault_walk: (.+?)\s+(\d+) runtime calls,\s+(\d+) of them failed in turn: (\d+) came back as errors, (\d+) were absorbed", text)
    assert len(rows) == 19 and sum(int(r[2]) for r in rows) > 800, text
    assert all(int(r[2]) == int(r[3]) + int(r[4]) and int(r[3]) > 0 for r in rows)


def test_the_deferred_model_catches_a_missing_wait(build, tmp_path):
    """the checker checks: a download on the default stream does not wait for a NON-BLOCKING stream's kernels - the deferred
    model hands back stale bytes (the eager one cannot), and the same code on a blocking stream, or with the wait, is right"""
    san, out = build
    if san != "address":
        pytest.skip("once is enough")
    src = tmp_path / "missing_wait.c"
    src.write_text(r'''
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "eddsa_amd.h"
int main(int argc, char **argv) {
    const int blocking = atoi(argv[2]), wait = atoi(argv[3]);
    uint8_t tab[96 * 4], got[32 * 4], *out = 0, *sc = 0, *pt = 0;
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(tab, 1, sizeof(tab), f) != sizeof(tab)) return 2;
    hipStream_t st;
    hipSetDevice(0); hipMalloc((void **)&out, 128); hipMalloc((void **)&sc, 128); hipMalloc((void **)&pt, 128);
    for (int i = 0; i < 4; i++) { hipMemcpy(pt + 32 * i, tab + 96 * i, 32, hipMemcpyHostToDevice); hipMemcpy(sc + 32 * i, tab + 96 * i + 32, 32, hipMemcpyHostToDevice); }
    if (blocking) hipStreamCreate(&st); else hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (x25519_batch_dev(out, sc, pt, 4, st) != 0) return 3;
    if (wait) hipStreamSynchronize(st);
    hipMemcpy(got, out, 128, hipMemcpyDeviceToHost);
    int right = 1;
    for (int i = 0; i < 4; i++) right &= memcmp(got + 32 * i, tab + 96 * i + 64, 32) == 0;
    hipStreamSynchronize(st); hipStreamDestroy(st); hipFree(out); hipFree(sc); hipFree(pt);
    eddsa_amd_shutdown();
    return right ? 0 : 5;
}
''')
    exe = tmp_path / "missing_wait"
    objs = [os.path.join(out, o) for o in ("eddsa_amd.o", "host_pipe.o", "fake_kernels.o")]
    subprocess.check_call(["g++", "-fsanitize=address,undefined", "-x", "c", "-std=c11", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", str(src),
                           "-x", "none"] + objs + ["-o", str(exe), "-L" + out, "-lfakehip", "-Wl,-rpath," + out, "-lpthread", "-ldl"])
    table = os.path.join(GOLD, "x25519_table.bin")

    def rc(defer, blocking, wait):
        return subprocess.run([str(exe), table, str(blocking), str(wait)], env=dict(os.environ, FAKE_HIP_DEVICES="2", FAKE_HIP_DEFER=str(defer), LD_LIBRARY_PATH=out),
                              capture_output=True, text=True, timeout=300).returncode
    assert rc(0, 0, 0) == 0            # eager model: the bug is invisible
    assert rc(1, 0, 0) == 5            # deferred: stale bytes
    assert rc(1, 0, 1) == 0            # ... the wait fixes it
    assert rc(1, 1, 0) == 0            # ... and a blocking stream is ordered with the default stream by definition


def test_the_fake_runtime_catches_a_buffer_on_the_wrong_device(build, tmp_path):
    """the checker checks: a device-pointer call whose input shard lives on another device than its output aborts"""
    san, out = build
    src = tmp_path / "wrong_device.c"
    src.write_text(r'''
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "eddsa_amd.h"
int main(void) {
    uint8_t *out = 0, *sc = 0, *pt = 0;
    hipSetDevice(0); hipMalloc((void **)&out, 32 * 4); hipMalloc((void **)&sc, 32 * 4);
    hipSetDevice(1); hipMalloc((void **)&pt, 32 * 4);          /* the points live on device 1, the call runs on device 0 */
    hipSetDevice(0);
    return x25519_batch_dev(out, sc, pt, 4, 0) == 0 ? 0 : 3;
}
''')
    exe = tmp_path / "wrong_device"
    flags = ["-fsanitize=thread"] if san == "thread" else ["-fsanitize=address,undefined"]
    objs = [os.path.join(out, o) for o in ("eddsa_amd.o", "host_pipe.o", "fake_kernels.o")]
    subprocess.check_call(["g++"] + flags + ["-x", "c", "-std=c11", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", str(src),
                                             "-x", "none"] + objs + ["-o", str(exe), "-L" + out, "-lfakehip", "-Wl,-rpath," + out, "-lpthread", "-ldl"])
    r = subprocess.run([str(exe)], env=dict(os.environ, FAKE_HIP_DEVICES="2", LD_LIBRARY_PATH=out), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "expected device 0" in r.stderr, r.stdout + r.stderr
