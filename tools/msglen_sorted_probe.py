#!/usr/bin/env python3
"""Does the ORDER of ragged messages matter when the chip is full?  2^20 items, lengths uniform in 0 .. L, as drawn and sorted by
length (what a length-sorted hash pre-pass would see): verify, sign and the batch verification (rlc), device-resident."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
g = torch.Generator(device="cuda").manual_seed(7)
dsk = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
dpk = ed.ed25519_genpub_batch(dsk)

def rate(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for L in (1024, 4096, 8192):
    rng = np.random.default_rng(L)
    lens = rng.integers(0, L + 1, n)
    for label, ls in (("as drawn", lens), ("sorted by length", np.sort(lens))):
        off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum(ls)
        blob = torch.randint(0, 256, (int(off[-1]),), dtype=torch.uint8, device="cuda", generator=g)
        doff = torch.from_numpy(off).cuda()
        sig = ed.ed25519_sign_batch(dsk, dpk, blob, msg_off=doff)
        assert int(ed.ed25519_verify_batch(sig, dpk, blob, msg_off=doff).sum()) == n
        tv = rate(lambda: ed.ed25519_verify_batch(sig, dpk, blob, msg_off=doff))
        ts = rate(lambda: ed.ed25519_sign_batch(dsk, dpk, blob, msg_off=doff))
        ok, st = ed.ed25519_verify_batch_rlc(sig, dpk, blob, msg_off=doff, return_stats=True)
        assert int(ok.sum()) == n and st[0] == n
        tr = rate(lambda: ed.ed25519_verify_batch_rlc(sig, dpk, blob, msg_off=doff))
        print(f"uniform 0 .. {L:5d} B, {label:17s} verify {tv*1e3:8.2f} ms {n/tv/1e6:7.2f} M/s | sign {ts*1e3:8.2f} ms {n/ts/1e6:7.2f} M/s"
              f" | batch verification {tr*1e3:8.2f} ms {n/tr/1e6:7.2f} M/s", flush=True)
        del blob
