import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import libeddsa_amd as ed, workload
ed.init(0)
for l in (16, 15, 14, 13):
    n = 1 << l
    sk, msg = workload.sign_inputs(n, seed=1, config=2)
    d = lambda a: torch.from_numpy(a).cuda()
    pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)).cpu().numpy(); pk = pk.cpu().numpy()
    workload.corrupt_for_verify(sig, pk, msg)
    ds, dp, dm = d(sig), d(pk), d(msg)
    for _ in range(5): ed.ed25519_verify_batch(ds, dp, dm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ed.ed25519_verify_batch(ds, dp, dm)
    torch.cuda.synchronize(); print(f"2^{l}: {(time.perf_counter()-t0)/30*1e3:.3f} ms", end="  ")
print()
