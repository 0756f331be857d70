#!/usr/bin/env python3
"""verify / sign throughput against the message length (inputs in HBM, 2^18 items): what the SHA-512 blocks and the
per-lane message reads cost beyond BASELINE's 32-byte messages"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << 18
rng = np.random.default_rng(5)
sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
d = lambda a: torch.from_numpy(a).cuda()
dsk = d(sk)
dpk = ed.ed25519_genpub_batch(dsk)
for L in (0, 32, 111, 112, 128, 256, 512, 1024, 4096):
    msg = d(rng.integers(0, 256, (n, max(L, 1)), dtype=np.uint8))[:, :L].contiguous()
    sig = ed.ed25519_sign_batch(dsk, dpk, msg)
    ok = ed.ed25519_verify_batch(sig, dpk, msg)
    assert int(ok.sum()) == n
    out = []
    for fn in (lambda: ed.ed25519_verify_batch(sig, dpk, msg), lambda: ed.ed25519_sign_batch(dsk, dpk, msg)):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); out.append(n * 5 / (time.perf_counter() - t0) / 1e6)
    blocks_v = (64 + L + 17 + 127) // 128
    print(f"msg_len {L:5d}: verify {out[0]:7.1f} M/s  sign {out[1]:7.1f} M/s   (SHA-512 blocks per verify {blocks_v}, message bytes read {L * out[0] / 1e3:7.1f} GB/s verify)")
