#!/usr/bin/env python3
"""one host-pointer call repeated, for a rocprofv3 --kernel-trace --memory-copy-trace timeline:
host_trace.py <verify|x25519|sign> [log2n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
import workload
ed.init(0)
op = sys.argv[1] if len(sys.argv) > 1 else "verify"
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else (10 if op.endswith("1") else 20))
sk, msg = workload.sign_inputs(n, seed=1, config=2)
pk = ed.ed25519_genpub_batch(sk)
sig = ed.ed25519_sign_batch(sk, pk, msg)
workload.corrupt_for_verify(sig, pk, msg)
sc, pt = workload.x25519_inputs(n)
fn = {"verify": lambda: ed.ed25519_verify_batch(sig, pk, msg), "x25519": lambda: ed.x25519_batch(sc, pt),
      "sign": lambda: ed.ed25519_sign_batch(sk, pk, msg),
      "verify1": lambda: ed.ed25519_verify(sig[0].tobytes(), pk[0].tobytes(), msg[0].tobytes()),
      "sign1": lambda: ed.ed25519_sign(sk[0].tobytes(), pk[0].tobytes(), msg[0].tobytes()),
      "genpub1": lambda: ed.ed25519_genpub(sk[0].tobytes()),
      "x255191": lambda: ed.x25519(sc[0].tobytes(), pt[0].tobytes())}[op]
import ctypes
lib = ed.library()
tags, chunks, ms = (ctypes.c_int * 512)(), (ctypes.c_uint * 512)(), (ctypes.c_double * 512)()
lib.eddsa_amd_debug_pipe_trace(1, tags, chunks, ms, 0)
for _ in range(3):
    fn()
time.sleep(0.05)
t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
print(f"{op}: {dt*1e3:.2f} ms host to host, {n/dt/1e6:.1f} M/s")
k = lib.eddsa_amd_debug_pipe_trace(0, tags, chunks, ms, 512)
names = ["call start", "lane drained", "inputs staged+queued", "kernels queued", "download queued", "all lanes drained", "call end", "leader elected", "callers gathered", "requests packed", "results handed back"]
for i in range(k):
    print(f"  host {ms[i]:8.3f} ms  chunk {chunks[i]}  {names[tags[i]]}")
