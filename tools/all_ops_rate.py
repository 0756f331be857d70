#!/usr/bin/env python3
"""Device-resident rate of every batched entry point at 2^20 items (kernel-only view, torch stream)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << 20
d = lambda a: torch.from_numpy(a).cuda()
sk, msg = workload.sign_inputs(n)
sc, pt = workload.x25519_inputs(n)
sk, msg, sc, pt = d(sk), d(msg), d(sc), d(pt)
pk = ed.ed25519_genpub_batch(sk)
sig = ed.ed25519_sign_batch(sk, pk, msg)
ops = {
    "ed25519_genpub_batch": lambda: ed.ed25519_genpub_batch(sk),
    "ed25519_sign_batch": lambda: ed.ed25519_sign_batch(sk, pk, msg),
    "ed25519_verify_batch": lambda: ed.ed25519_verify_batch(sig, pk, msg),
    "x25519_batch": lambda: ed.x25519_batch(sc, pt),
    "x25519_base_batch": lambda: ed.x25519_base_batch(sc),
    "pk_ed25519_to_x25519_batch": lambda: ed.pk_ed25519_to_x25519_batch(pk),
    "sk_ed25519_to_x25519_batch": lambda: ed.sk_ed25519_to_x25519_batch(sk),
}
for name, f in ops.items():
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name:32s} {dt*1e3:8.3f} ms  {n/dt/1e6:9.1f} M/s")
