#!/usr/bin/env python3
"""Single-call latency of the eddsa.h drop-in functions and its floor: sk_ed25519_to_x25519 is one SHA-512
block on the GPU (about 10 us), so its latency is the host path's fixed cost (copies, launches, syncs)."""
import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
sk = bytes(range(32)); pk = ed.ed25519_genpub(sk); msg = b"x" * 32; sig = ed.ed25519_sign(sk, pk, msg)
for name, f in (("sk_ed25519_to_x25519 (fixed cost)", lambda: ed.sk_ed25519_to_x25519(sk)),
                ("pk_ed25519_to_x25519", lambda: ed.pk_ed25519_to_x25519(pk)),
                ("ed25519_genpub", lambda: ed.ed25519_genpub(sk)), ("ed25519_sign", lambda: ed.ed25519_sign(sk, pk, msg)),
                ("ed25519_verify", lambda: ed.ed25519_verify(sig, pk, msg)), ("x25519", lambda: ed.x25519(sk, pk)),
                ("x25519_base", lambda: ed.x25519_base(sk))):
    for _ in range(30): f()
    t0 = time.perf_counter()
    for _ in range(300): f()
    print(f"{name:36s} {(time.perf_counter()-t0)/300*1e6:8.1f} us per call")
