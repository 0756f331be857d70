#!/usr/bin/env python3
"""Soak of the one-lane exact path's hand-off protocol (units drawn from a counter, accumulators crossing between waves through
memory under acquire / release): the same passes over and over - every key random, every second key random, exactly 9000 keys off
the curve, self-check mode 2 - alone and from two host threads on two streams at once; every verdict vector must equal the first
one (and, for mode 2, the genuine verdicts).  tools/exact_soak.py [rounds]"""
import os, sys, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = 1 << 20
sk, msg = workload.sign_inputs(n)
d = lambda a: torch.from_numpy(a).cuda()
pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)); dm = d(msg)
rng = np.random.default_rng(1)
garbage = d(rng.integers(0, 256, (n, 32), dtype=np.uint8))
half = pk.clone(); half[::2] = garbage[::2]
few = pk.clone(); few[: 9000 * 2 + 4000] = garbage[: 9000 * 2 + 4000]     # ~10 k off the curve: just above the threshold
cases = {"all random": garbage, "every second": half, "just above the threshold": few}
first = {k: ed.ed25519_verify_batch(sig, v, dm).clone() for k, v in cases.items()}
bad = 0
for r in range(rounds):
    for k, v in cases.items():
        ok = ed.ed25519_verify_batch(sig, v, dm)
        if not torch.equal(ok, first[k]):
            bad += 1; print("MISMATCH", k, r, int((ok != first[k]).sum()), flush=True)
ed.set_offcurve_mode(2)
for r in range(max(4, rounds // 10)):
    ok = ed.ed25519_verify_batch(sig, pk, dm)
    if int(ok.sum()) != n: bad += 1; print("MISMATCH mode 2", r, n - int(ok.sum()), flush=True)
ed.set_offcurve_mode(True)
print(f"sequential: {rounds} rounds x {len(cases)} cases, {bad} mismatches", flush=True)

def worker(tag, keys, want, out):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for r in range(rounds // 2):
            ok = ed.ed25519_verify_batch(sig, keys, dm)
            s.synchronize()
            if not torch.equal(ok, want): out.append((tag, r))
errs = []
ts = [threading.Thread(target=worker, args=("all random", garbage, first["all random"], errs)),
      threading.Thread(target=worker, args=("every second", half, first["every second"], errs))]
for t in ts: t.start()
for t in ts: t.join()
print(f"two streams at once: {rounds // 2} rounds each, {len(errs)} mismatches {errs[:5]}", flush=True)
sys.exit(1 if bad or errs else 0)
