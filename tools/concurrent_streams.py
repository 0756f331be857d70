#!/usr/bin/env python3
"""Do verify passes issued by several host threads, each on its own stream, overlap on the GPU?
Time T threads x R passes of n items against one thread doing T x R passes."""
import os, sys, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
R = 40
for lg in (12, 14, 16):
    n = 1 << lg
    sk, msg = workload.sign_inputs(n)
    d = lambda a: torch.from_numpy(a).cuda()
    pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)); dm = d(msg)
    def worker(reps, res, k):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(reps):
                ok = ed.ed25519_verify_batch(sig, pk, dm)
            st.synchronize()
        res[k] = int(ok.sum())
    for T in (1, 2, 4):
        res = [None] * T
        worker(3, res, 0); torch.cuda.synchronize()
        th = [threading.Thread(target=worker, args=(R, res, k)) for k in range(T)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        assert all(r == n for r in res)
        print(f"n=2^{lg} threads={T}: {T*R*n/dt/1e6:7.1f} M/s  ({dt/(T*R)*1e3:.3f} ms per pass)")
