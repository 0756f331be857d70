#!/usr/bin/env python3
"""Time of one pass of the fixed-base operations for small and mid-size batches, device-resident, back to back"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
d = lambda a: torch.from_numpy(a).cuda()
for name in ("genpub", "sign", "x25519_base"):
    print(f"{name:12s}", end=" ")
    for l in (0, 6, 10, 12, 13, 14, 15, 16, 18):
        n = 1 << l
        sk, msg = workload.sign_inputs(n, seed=1, config=5)
        dsk, dmsg = d(sk), d(msg)
        pk = ed.ed25519_genpub_batch(dsk)
        f = {"genpub": lambda: ed.ed25519_genpub_batch(dsk), "sign": lambda: ed.ed25519_sign_batch(dsk, pk, dmsg),
             "x25519_base": lambda: ed.x25519_base_batch(dsk)}[name]
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): f()
        torch.cuda.synchronize(); print(f"2^{l}: {(time.perf_counter()-t0)/30*1e3:.3f}", end="  ")
    print("ms")
