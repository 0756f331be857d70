#!/usr/bin/env python3
"""The one-lane exact path alone and beside the windowed kernels: (a) self-check mode 2 (every item of a 2^20-item pass
through k_verify_exact_lane_*: nothing else runs), (b) genuine signatures under random keys (47 % no curve points).
Run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations."""
import torch, time, numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sk, msg = workload.sign_inputs(n)
d = lambda a: torch.from_numpy(a).cuda()
pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)); dm = d(msg)
rng = np.random.default_rng(1)
garbage = d(rng.integers(0, 256, (n, 32), dtype=np.uint8))

def timed(label, keys):
    ok = ed.ed25519_verify_batch(sig, keys, dm); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ok = ed.ed25519_verify_batch(sig, keys, dm)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{label}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} M/s  accepted {int(ok.sum())}", flush=True)

ed.set_offcurve_mode(2)
timed("mode 2 (every item through the exact path)", pk)
ed.set_offcurve_mode(True)
if len(sys.argv) > 3 and sys.argv[3] == "alone":     # (counter passes: only the dispatches that have the chip to themselves)
    sys.exit(0)
timed("valid", pk)
timed("random keys", garbage)
half = pk.clone(); half[::2] = garbage[::2]
timed("every second key random", half)
