"""Mid-size verify passes: the default routes (algo 0), the mid-size arrangement forced (3) and one lane per item throughout (2) - verdicts against the constructed ones, then ms per pass."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
d = lambda a: torch.from_numpy(a).cuda()
for n in (1000, 1 << 15, (1 << 16) + 5, 1 << 17, (1 << 18) - 3):
    sk, msg = workload.sign_inputs(n, seed=3, config=2)
    pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)).cpu().numpy(); pk = pk.cpu().numpy()
    want = workload.corrupt_for_verify(sig, pk, msg)
    ds, dp, dm = d(sig), d(pk), d(msg)
    res = {}
    for algo in (0, 3, 2):
        ed.set_verify_algo(algo)
        res[algo] = ed.ed25519_verify_batch(ds, dp, dm).cpu().numpy()
    print(n, "algo3==want", np.array_equal(res[3], want), "algo0==want", np.array_equal(res[0], want), "algo2==want", np.array_equal(res[2], want))
for clean in (True, False):
    for algo in (0, 3, 2):
        ed.set_verify_algo(algo)
        print(f"algo {algo} {'valid only   ' if clean else 'config-2 mix '}", end=" ")
        for l in (17, 16, 15, 14, 13):
            n = 1 << l
            sk, msg = workload.sign_inputs(n, seed=1, config=2)
            pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)).cpu().numpy(); pk = pk.cpu().numpy()
            if not clean: workload.corrupt_for_verify(sig, pk, msg)
            ds, dp, dm = d(sig), d(pk), d(msg)
            for _ in range(5): ed.ed25519_verify_batch(ds, dp, dm)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): ed.ed25519_verify_batch(ds, dp, dm)
            torch.cuda.synchronize(); print(f"2^{l}: {(time.perf_counter()-t0)/30*1e3:.3f}", end="  ")
        print("ms")
