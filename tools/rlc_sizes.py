#!/usr/bin/env python3
"""Batch verification against the per-item kernels on all-valid batches of several sizes (inputs in HBM)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0]); sys.path.insert(0, __file__.rsplit("/", 1)[0])
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
ed.set_rlc_min_items(0)          # (the routing threshold is what this table decides)
N = 1 << 21
sk, msg = workload.sign_inputs(N, seed=1, config=2)
d_sk, d_msg = torch.from_numpy(sk).cuda(), torch.from_numpy(msg).cuda()
pk = ed.ed25519_genpub_batch(d_sk); sig = ed.ed25519_sign_batch(d_sk, pk, d_msg)
for l in (13, 14, 15, 16, 17, 18, 19, 20, 21):
    n = 1 << l
    row = []
    for fn in (ed.ed25519_verify_batch, ed.ed25519_verify_batch_rlc):
        for _ in range(3): ok = fn(sig[:n], pk[:n], d_msg[:n], msg_len=32)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ok = fn(sig[:n], pk[:n], d_msg[:n], msg_len=32)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        assert bool(ok.all())
        row.append(dt)
    print(f"n=2^{l}: per item {row[0]*1e3:7.3f} ms {n/row[0]/1e6:6.1f} M/s | batch verification {row[1]*1e3:7.3f} ms {n/row[1]/1e6:6.1f} M/s | x{row[0]/row[1]:.2f}")
