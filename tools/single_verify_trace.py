import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
sk = bytes(range(32)); pk = ed.ed25519_genpub(sk); msg = b"x" * 32; sig = ed.ed25519_sign(sk, pk, msg)
for _ in range(50): ed.ed25519_verify(sig, pk, msg)
t0=time.perf_counter()
for _ in range(200): ed.ed25519_verify(sig, pk, msg)
print("verify us", (time.perf_counter()-t0)/200*1e6)
