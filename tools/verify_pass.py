#!/usr/bin/env python3
"""One verify pass of 2^L items, repeated: tools/verify_pass.py L [reps] [valid|mix] [gap_ms]
(the program rocprofv3 runs for the counter passes of tools/pmc_by_size.sh; prints the wall ms per pass)."""
import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
L = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kind = sys.argv[3] if len(sys.argv) > 3 else "mix"
gap = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
ed.init(0)
n = 1 << L
sk, msg = workload.sign_inputs(n, seed=1, config=2)
d = lambda a: torch.from_numpy(a).cuda()
pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)).cpu().numpy(); pk = pk.cpu().numpy()
expect = np.ones(n, np.uint8)
if kind == "mix": expect = workload.corrupt_for_verify(sig, pk, msg)
ds, dp, dm = d(sig), d(pk), d(msg)
for _ in range(2): ok = ed.ed25519_verify_batch(ds, dp, dm)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    if gap: time.sleep(gap * 1e-3)
    t0 = time.perf_counter(); ok = ed.ed25519_verify_batch(ds, dp, dm); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
assert np.array_equal(ok.cpu().numpy(), expect)
print(f"2^{L} {kind}: wall ms per pass min {min(ts):.3f} median {sorted(ts)[len(ts) // 2]:.3f}")
