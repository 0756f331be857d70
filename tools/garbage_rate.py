#!/usr/bin/env python3
"""ms per verify pass of 2^20 genuine signatures under keys of which every k-th is a random string (k = 1, 2, 16): the
worst case a sender can construct, device-resident, HIP events.  For A/B runs through tools/ab.sh."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << 20
sk, msg = workload.sign_inputs(n)
d = lambda a: torch.from_numpy(a).cuda()
pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)); dm = d(msg)
garbage = d(np.random.default_rng(1).integers(0, 256, (n, 32), dtype=np.uint8))
out = []
for step in (1, 2, 16):
    keys = pk.clone(); keys[::step] = garbage[::step]
    for _ in range(3): ed.ed25519_verify_batch(sig, keys, dm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(8): ok = ed.ed25519_verify_batch(sig, keys, dm)
    e1.record(); torch.cuda.synchronize()
    out.append("1/%d random: %.2f ms (%d accepted)" % (step, e0.elapsed_time(e1) / 8, int(ok.sum())))
print(" | ".join(out))
