"""One verify pass (config-2 mix, device-resident) by size: SIZES=19,18,17 python tools/verify_sizes.py"""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
import workload
ed.init(0)
if os.environ.get("OFFCURVE_MODE"): ed.set_offcurve_mode(int(os.environ["OFFCURVE_MODE"]))
if os.environ.get("VERIFY_ALGO"): ed.set_verify_algo(int(os.environ["VERIFY_ALGO"]))   # 1 full-length, 2 half-length, whatever the size
for lg in [int(x) for x in os.environ.get("SIZES", "20,19,18,17,16,15,14,13,12").split(",")]:
    n = 1 << lg
    sk, msg = workload.sign_inputs(n, seed=1, config=2)
    pk = ed.ed25519_genpub_batch(sk); sig = ed.ed25519_sign_batch(sk, pk, msg)
    expect = workload.corrupt_for_verify(sig, pk, msg)
    def timeit(fn, reps=5):
        fn(); best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
        return best, out
    dt, ok = timeit(lambda: ed.ed25519_verify_batch(sig, pk, msg)); assert np.array_equal(ok, expect)
    d = lambda a: torch.from_numpy(a).cuda()
    ds, dp, dm = d(sig), d(pk), d(msg)
    def dev():
        o = ed.ed25519_verify_batch(ds, dp, dm); torch.cuda.synchronize(); return o
    dt2, _ = timeit(dev)
    print(f"n=2^{lg}: host {dt*1e3:7.2f} ms {n/dt/1e6:6.1f} M/s | device {dt2*1e3:7.2f} ms {n/dt2/1e6:6.1f} M/s")
