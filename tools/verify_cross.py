"""where the four-lane route and the mid-size arrangement cross: one valid-only pass, ms, algo 0 (default) / 3 (mid-size arrangement forced)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
d = lambda a: torch.from_numpy(a).cuda()
n0 = 1 << 15
sk, msg = workload.sign_inputs(n0, seed=1, config=2)
pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg))
dm = d(msg)
for algo in (0, 3):
    ed.set_verify_algo(algo)
    print(f"algo {algo}", end="  ")
    for n in (16384, 18000, 20000, 22000, 24576, 26000, 28000, 30000, 32768):
        a, b, c = sig[:n].contiguous(), pk[:n].contiguous(), dm[:n].contiguous()
        for _ in range(5): ed.ed25519_verify_batch(a, b, c)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): ed.ed25519_verify_batch(a, b, c)
        torch.cuda.synchronize(); print(f"{n}: {(time.perf_counter()-t0)/30*1e3:.3f}", end="  ")
    print("ms")
ed.set_verify_algo(0)
