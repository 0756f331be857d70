#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer entry points (the streaming pipeline of
eddsa_amd.c) with pageable and with pinned caller memory.  DESIGN.md quotes these; bench.py's
`value` never includes transfers."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import libeddsa_amd as ed
import workload

ed.init(0)
n = 1 << 20
sk, msg = workload.sign_inputs(n, seed=1, config=2)
pk = ed.ed25519_genpub_batch(sk)
sig = ed.ed25519_sign_batch(sk, pk, msg)
expect = workload.corrupt_for_verify(sig, pk, msg)
sc, pt = workload.x25519_inputs(n)

def timeit(fn, reps=5):
    fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    return best, out

def pin(a):
    t = torch.from_numpy(a).pin_memory()
    return t.numpy()

for label, conv in (("pageable", lambda a: a), ("pinned", pin)):
    s_, p_, m_ = conv(sig), conv(pk), conv(msg)
    dt, ok = timeit(lambda: ed.ed25519_verify_batch(s_, p_, m_))
    assert np.array_equal(ok, expect)
    print(f"verify  host->host {label:9s}: {n/dt/1e6:7.1f} M/s  ({dt*1e3:.1f} ms for 2^20)")
    a_, b_ = conv(sc), conv(pt)
    dt, out = timeit(lambda: ed.x25519_batch(a_, b_))
    print(f"x25519  host->host {label:9s}: {n/dt/1e6:7.1f} M/s  ({dt*1e3:.1f} ms)")
    k_, q_, mm_ = conv(sk), conv(pk), conv(msg)
    dt, out = timeit(lambda: ed.ed25519_sign_batch(k_, q_, mm_))
    print(f"sign    host->host {label:9s}: {n/dt/1e6:7.1f} M/s  ({dt*1e3:.1f} ms)")
