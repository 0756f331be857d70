#!/usr/bin/env python3
"""one verify configuration, for a rocprofv3 --kernel-trace timeline: exact_trace.py <1/share or 0>"""
import torch, numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << int(os.environ.get("LOG2N", "20"))
step = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sk, msg = workload.sign_inputs(n)
d = lambda a: torch.from_numpy(a).cuda()
pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)); dm = d(msg)
if step:
    rng = np.random.default_rng(1)
    pk[::step] = d(rng.integers(0, 256, (n, 32), dtype=np.uint8))[::step]
for _ in range(6):
    ok = ed.ed25519_verify_batch(sig, pk, dm)
torch.cuda.synchronize()
print(int(ok.sum()))
