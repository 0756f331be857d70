"""Host buffer to host buffer rates of verify / x25519 / sign for several chunk schedules and kernel orderings of the host pipeline (eddsa_amd_set_pipeline, eddsa_amd_set_pipeline_chain)."""
import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
import workload, ctypes
ed.init(0)
n = 1 << 20
sk, msg = workload.sign_inputs(n, seed=1, config=2)
pk = ed.ed25519_genpub_batch(sk); sig = ed.ed25519_sign_batch(sk, pk, msg)
expect = workload.corrupt_for_verify(sig, pk, msg)
sc, pt = workload.x25519_inputs(n)
def timeit(fn, reps=6):
    fn(); fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    return best, out
lib = ed.library()
for chain in (0, 1):
    lib.eddsa_amd_set_pipeline_chain(chain)
    for first, stage in ((16, 18), (16, 19), (16, 20), (17, 18), (17, 19), (17, 20), (18, 19), (18, 20)):
        lib.eddsa_amd_set_pipeline(ctypes.c_size_t(1 << first), ctypes.c_size_t(1 << stage))
        dv, ok = timeit(lambda: ed.ed25519_verify_batch(sig, pk, msg)); assert np.array_equal(ok, expect)
        dx, _ = timeit(lambda: ed.x25519_batch(sc, pt))
        ds, _ = timeit(lambda: ed.ed25519_sign_batch(sk, pk, msg))
        print(f"chain {chain} first 2^{first} stage 2^{stage}: verify {n/dv/1e6:6.1f} M/s ({dv*1e3:5.2f} ms)  x25519 {n/dx/1e6:6.1f} ({dx*1e3:5.2f})  sign {n/ds/1e6:6.1f} ({ds*1e3:5.2f})", flush=True)
