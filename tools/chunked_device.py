#!/usr/bin/env python3
"""What does chunking alone cost?  The config-2 batch, resident in HBM, verified (a) in one call, (b) in the host
pipeline's chunk schedule on three streams side by side, (c) the same in stream order - no copies anywhere."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
n = 1 << 20
sk, msg = workload.sign_inputs(n, seed=1, config=2)
d = lambda a: torch.from_numpy(a).cuda()
pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)).cpu().numpy(); pk = pk.cpu().numpy()
expect = workload.corrupt_for_verify(sig, pk, msg)
ds, dp, dm = d(sig), d(pk), d(msg)
streams = [torch.cuda.Stream() for _ in range(3)]

def one():
    ok = ed.ed25519_verify_batch(ds, dp, dm); torch.cuda.synchronize(); return ok

def chunked(sizes, side_by_side=True):
    outs, lo = [], 0
    for k, m in enumerate(sizes):
        st = streams[k % 3] if side_by_side else streams[0]
        with torch.cuda.stream(st):
            outs.append(ed.ed25519_verify_batch(ds[lo:lo + m], dp[lo:lo + m], dm[lo:lo + m]))
        lo += m
    torch.cuda.synchronize()
    return torch.cat(outs)

def timeit(fn, reps=7):
    fn(); fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    assert np.array_equal(out.cpu().numpy(), expect)
    return best * 1e3

print(f"one call                         {timeit(one):6.2f} ms")
for sizes in ([1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 16], [1 << 16, 1 << 17, 1 << 18, 589824], [1 << 17, 1 << 18, 655360],
              [1 << 18] * 4, [1 << 19] * 2):
    print(f"{str([s >> 10 for s in sizes]):32s} K items: side by side {timeit(lambda: chunked(sizes)):6.2f} ms   in order {timeit(lambda: chunked(sizes, False)):6.2f} ms")
