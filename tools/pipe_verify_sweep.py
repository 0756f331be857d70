"""Host buffer to host buffer rate of ed25519_verify_batch (2^20 items, numpy arrays) by chunk schedule, for the config-2 mix
(an exact chain beside every chunk's main kernel) and for valid signatures only (no exact path below 2^19 items): what the
chains cost the host pipeline.  tools/pipe_verify_sweep.py"""
import os, sys, time, ctypes
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << 20
sk, msg = workload.sign_inputs(n, seed=1, config=2)
pk = ed.ed25519_genpub_batch(sk); sig = ed.ed25519_sign_batch(sk, pk, msg)
lib = ed.library()
def timeit(fn, reps=7):
    fn(); fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
    return min(ts), out
for kind in ("valid", "mix"):
    s, p, m = sig.copy(), pk.copy(), msg.copy()
    expect = np.ones(n, np.uint8) if kind == "valid" else workload.corrupt_for_verify(s, p, m)
    d = lambda a: torch.from_numpy(a).cuda()
    ds, dp, dm = d(s), d(p), d(m)
    def dev():
        o = ed.ed25519_verify_batch(ds, dp, dm); torch.cuda.synchronize(); return o
    t, o = timeit(dev); assert np.array_equal(o.cpu().numpy(), expect)
    print(f"{kind:5s} device-resident, one pass           {n / t / 1e6:6.1f} M/s ({t * 1e3:5.2f} ms)", flush=True)
    for first, stage in [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or ((0, 0), (16, 19), (16, 20), (17, 19), (17, 20), (18, 20), (19, 20), (20, 20)):
        lib.eddsa_amd_set_pipeline(ctypes.c_size_t(1 << first if first else 0), ctypes.c_size_t(1 << stage if stage else 0))
        t, o = timeit(lambda: ed.ed25519_verify_batch(s, p, m)); assert np.array_equal(o, expect)
        label = "default schedule" if not first else f"first 2^{first} stage 2^{stage}"
        print(f"{kind:5s} host to host, {label:22s} {n / t / 1e6:6.1f} M/s ({t * 1e3:5.2f} ms)", flush=True)
    lib.eddsa_amd_set_pipeline(ctypes.c_size_t(0), ctypes.c_size_t(0))
