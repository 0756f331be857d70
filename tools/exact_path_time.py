#!/usr/bin/env python3
"""Cost of the exact (reference-order) path for off-curve public keys: time of a verify pass over
n items as a function of the share of off-curve keys."""
import torch, time, numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << 20
sk, msg = workload.sign_inputs(n)
d = lambda a: torch.from_numpy(a).cuda()
pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)); dm = d(msg)
rng = np.random.default_rng(1)
garbage = d(rng.integers(0, 256, (n, 32), dtype=np.uint8))
for share in (0, 1 / 1024, 1 / 128, 1 / 16, 1 / 4, 1 / 2, 1):
    keys = pk.clone()
    if share:
        step = int(1 / share)
        keys[::step] = garbage[::step]
    ed.ed25519_verify_batch(sig, keys, dm); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ok = ed.ed25519_verify_batch(sig, keys, dm)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"garbage keys 1/{int(1/share) if share else 0}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} M/s  accepted {int(ok.sum())}")

# the former cliff: the four-lane chain took 65 536 work-list entries and a spilling one-lane kernel the rest (rounds 1-3);
# now every stretch of 65 536 entries is one more launch of the same kernel.  Exactly k keys off the curve, k around 65 536:
lib = ed.library()
import ctypes
ed.debug_init(0, True)
flags = ed.debug_layer("ed_import_export", [bytes(r) for r in garbage[:400000].cpu().numpy()], 33)
off = np.array([f[32] == 0 for f in flags])
idx_off = np.nonzero(off)[0]
for k in (2048, 4096, 8192, 16384, 32768, 65536, 131072):
    keys = pk.clone()
    sel = torch.from_numpy(idx_off[:k]).cuda()
    keys[sel] = garbage[sel]
    ed.ed25519_verify_batch(sig, keys, dm); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ok = ed.ed25519_verify_batch(sig, keys, dm)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"exactly {k} keys off the curve: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} M/s  accepted {int(ok.sum())}")
ed.debug_init(0, False)

ed.set_offcurve_mode(False)
keys = pk.clone(); keys[::128] = garbage[::128]
ed.ed25519_verify_batch(sig, keys, dm); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ok = ed.ed25519_verify_batch(sig, keys, dm)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"reject mode, garbage keys 1/128: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} M/s  accepted {int(ok.sum())}")
ed.set_offcurve_mode(True)
