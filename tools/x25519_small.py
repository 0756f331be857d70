"""Time of one x25519 pass for small and mid-size batches, device-resident, back to back."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
d = lambda a: torch.from_numpy(a).cuda()
print("x25519      ", end=" ")
for l in (0, 10, 12, 13, 14, 15, 16):
    n = 1 << l
    sc, pt = workload.x25519_inputs(n)
    a, b = d(sc), d(pt)
    for _ in range(5): ed.x25519_batch(a, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ed.x25519_batch(a, b)
    torch.cuda.synchronize(); print(f"2^{l}: {(time.perf_counter()-t0)/30*1e3:.3f}", end="  ")
print("ms")
