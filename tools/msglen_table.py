#!/usr/bin/env python3
"""Verify and sign against the message length, device-resident inputs: fixed lengths 32 B .. 64 KiB, ragged lengths drawn
uniformly from 0 .. 64 KiB, and the skewed case (32-byte messages, one in 1024 of 1 MiB).  A lane hashes its message
block by block (csrc/sha512.h); a wave ends with its longest message."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
g = torch.Generator(device="cuda").manual_seed(7)
dsk = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
dpk = ed.ed25519_genpub_batch(dsk)


def rate(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def row(label, blob, off, mlen, total_bytes, max_len):
    kw = dict(msg_off=off) if off is not None else dict(msg_len=mlen)
    sig = ed.ed25519_sign_batch(dsk, dpk, blob, **kw)
    ok = ed.ed25519_verify_batch(sig, dpk, blob, **kw)
    assert int(ok.sum()) == n, label
    tv = rate(lambda: ed.ed25519_verify_batch(sig, dpk, blob, **kw))
    ts = rate(lambda: ed.ed25519_sign_batch(dsk, dpk, blob, **kw))
    blocks = (64 + max_len + 17 + 127) // 128
    print(f"{label:34s} verify {tv*1e3:9.2f} ms {n/tv/1e6:8.2f} M/s {total_bytes/tv/1e9:7.1f} GB/s | sign {ts*1e3:9.2f} ms {n/ts/1e6:8.2f} M/s"
          f" | longest message {blocks} blocks: {tv*1e6/blocks:6.2f} us of the verify pass per block of it", flush=True)


for L in (32, 1024, 4096, 16384, 65536):
    blob = torch.randint(0, 256, (n * L,), dtype=torch.uint8, device="cuda", generator=g)
    row(f"fixed {L} B", blob, None, L, n * L, L)
    del blob
rng = np.random.default_rng(3)
lens = rng.integers(0, 65537, n)
off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum(lens)
blob = torch.randint(0, 256, (int(off[-1]),), dtype=torch.uint8, device="cuda", generator=g)
row("ragged, uniform 0 .. 64 KiB", blob, torch.from_numpy(off).cuda(), 0, int(off[-1]), int(lens.max()))
order = np.argsort(lens, kind="stable")
# the same messages handed over in order of length (what a length-sorted pre-pass would see): every wave's lanes end together
lens_s = lens[order]
off_s = np.zeros(n + 1, np.int64); off_s[1:] = np.cumsum(lens_s)
row("  the same, sorted by length", blob, torch.from_numpy(off_s).cuda(), 0, int(off_s[-1]), int(lens_s.max()))
del blob
lens = np.full(n, 32, np.int64); lens[511::1024] = 1 << 20
off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum(lens)
blob = torch.randint(0, 256, (int(off[-1]),), dtype=torch.uint8, device="cuda", generator=g)
row("skewed: 32 B, 1 in 1024 of 1 MiB", blob, torch.from_numpy(off).cuda(), 0, int(off[-1]), 1 << 20)
lens_s = np.sort(lens)
off_s = np.zeros(n + 1, np.int64); off_s[1:] = np.cumsum(lens_s)
row("  the same, long ones together", blob, torch.from_numpy(off_s).cuda(), 0, int(off_s[-1]), 1 << 20)
