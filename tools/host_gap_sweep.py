#!/usr/bin/env python3
"""Where the host-to-host verify call (2^20 items of the config-2 mix, numpy arrays in, numpy array out) loses against the
device-resident pass: the rate by copier threads, by first chunk, and from page-locked caller memory (no staging copy).
tools/host_gap_sweep.py  ->  profiles/r06_host_gap.txt"""
import os, sys, time, ctypes
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import libeddsa_amd as ed, workload
ed.use_debug_library()
ed.init(0)
n = 1 << 20
sk, msg = workload.sign_inputs(n, seed=1, config=2)
pk = ed.ed25519_genpub_batch(sk); sig = ed.ed25519_sign_batch(sk, pk, msg)
expect = workload.corrupt_for_verify(sig, pk, msg)
lib = ed.library()


def timeit(fn, reps=9):
    fn(); fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2], out


d = lambda a: torch.from_numpy(a).cuda()
ds, dp, dm = d(sig), d(pk), d(msg)


def dev():
    o = ed.ed25519_verify_batch(ds, dp, dm); torch.cuda.synchronize(); return o


t, tm, o = timeit(dev); assert np.array_equal(o.cpu().numpy(), expect)
print(f"device-resident pass                                   {n / t / 1e6:6.1f} M/s  best {t * 1e3:5.2f} ms  median {tm * 1e3:5.2f}", flush=True)
t0 = time.perf_counter(); ds2, dp2, dm2 = d(sig), d(pk), d(msg); torch.cuda.synchronize(); up = time.perf_counter() - t0
print(f"(torch H2D of the three arrays from pageable memory, one after the other: {up * 1e3:.2f} ms)")
for threads in (6, 10, 14):
    ed.set_host_threads(threads)
    for first in (15, 16, 17):
        lib.eddsa_amd_set_pipeline(ctypes.c_size_t(1 << first), ctypes.c_size_t(0))
        t, tm, o = timeit(lambda: ed.ed25519_verify_batch(sig, pk, msg)); assert np.array_equal(o, expect)
        print(f"malloc memory, {threads:2d} copier threads, first chunk 2^{first}      {n / t / 1e6:6.1f} M/s  best {t * 1e3:5.2f} ms  median {tm * 1e3:5.2f}", flush=True)
lib.eddsa_amd_set_pipeline(ctypes.c_size_t(0), ctypes.c_size_t(0))
ed.set_host_threads(6)
hs, hp, hm = ed.host_array(sig.shape), ed.host_array(pk.shape), ed.host_array(msg.shape)
hs[:], hp[:], hm[:] = sig, pk, msg
# chunk schedules (first chunk, every later one twice its predecessor up to the cap) and kernel orderings of consecutive chunks
# (-1 the operation's own = 0 side by side for verify; 1 in chunk order; 2 the next chunk beside the previous main kernel)
for chain in (-1, 1, 2):
    lib.eddsa_amd_set_pipeline_chain(chain)
    for first, cap in ((15, 0), (16, 0), (17, 0), (16, 16), (16, 17), (16, 18), (16, 19), (17, 17), (17, 18), (17, 19), (18, 18), (18, 19)):
        lib.eddsa_amd_set_pipeline(ctypes.c_size_t(1 << first), ctypes.c_size_t((1 << cap) if cap else 0))
        t, tm, o = timeit(lambda: ed.ed25519_verify_batch(hs, hp, hm)); assert np.array_equal(o, expect)
        t2, tm2, o = timeit(lambda: ed.ed25519_verify_batch(sig, pk, msg)); assert np.array_equal(o, expect)
        print(f"order {chain:2d}, first chunk 2^{first}, cap 2^{cap if cap else 20}: page-locked {n / t / 1e6:6.1f} M/s {t * 1e3:5.2f} ms | malloc {n / t2 / 1e6:6.1f} M/s {t2 * 1e3:5.2f} ms", flush=True)
lib.eddsa_amd_set_pipeline_chain(-1)
lib.eddsa_amd_set_pipeline(ctypes.c_size_t(0), ctypes.c_size_t(0))
