"""Latency of one call of each single-item function (batch of one, host buffers) and of small verify batches."""
import time, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
sk = bytes(range(32)); pk = ed.ed25519_genpub(sk); msg = b"x" * 32; sig = ed.ed25519_sign(sk, pk, msg)
for name, f in (("ed25519_verify", lambda: ed.ed25519_verify(sig, pk, msg)), ("ed25519_sign", lambda: ed.ed25519_sign(sk, pk, msg)),
                ("x25519", lambda: ed.x25519(sk, pk)), ("ed25519_genpub", lambda: ed.ed25519_genpub(sk))):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(200): f()
    print(f"{name:16s} {(time.perf_counter()-t0)/200*1e6:8.1f} us per call (batch of one, host buffers)")
for n in (256, 4096, 65536):
    s = np.tile(np.frombuffer(sig, np.uint8), (n, 1)); p = np.tile(np.frombuffer(pk, np.uint8), (n, 1)); m = np.tile(np.frombuffer(msg, np.uint8), (n, 1))
    for _ in range(3): ed.ed25519_verify_batch(s, p, m)
    t0 = time.perf_counter()
    for _ in range(10): ed.ed25519_verify_batch(s, p, m)
    dt = (time.perf_counter()-t0)/10
    print(f"verify_batch n={n:6d}: {dt*1e3:7.3f} ms  {n/dt/1e6:7.2f} M/s (host buffers)")
