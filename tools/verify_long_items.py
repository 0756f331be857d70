"""Small verify passes over several different batches of genuine signatures: with pairs searched up to 2^134 a batch contains an
item without a short pair now and then (8.5 in 10^5), and the four-lane evaluation then runs 64 windows in that item's wave.
Mean and worst pass time over the batches, ms."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
d = lambda a: torch.from_numpy(a).cuda()
for n in (256, 1024, 2048, 8192, 16384, 24576):
    ts = []
    for seed in range(12):
        sk, msg = workload.sign_inputs(n, seed=100 + seed, config=2)
        pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)); dm = d(msg)
        for _ in range(3): ok = ed.ed25519_verify_batch(sig, pk, dm)
        assert int(ok.sum()) == n
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ed.ed25519_verify_batch(sig, pk, dm)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"n={n:6d}: mean {np.mean(ts):.3f}  min {min(ts):.3f}  max {max(ts):.3f}  ms over 12 batches")
