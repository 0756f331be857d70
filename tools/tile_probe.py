#!/usr/bin/env python3
"""Does tiling a 2^20-item verify pass pay?  The batch resident in HBM, verified in one call and as equal tiles of 2^L items
on 1, 2 or 3 streams, for valid signatures only (no exact path below 2^18 items: the pair search is the wide one) and for
the config-2 mix (an exact chain beside every tile):  tools/tile_probe.py [reps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ed.init(0)
n = 1 << 20
sk, msg = workload.sign_inputs(n, seed=1, config=2)
d = lambda a: torch.from_numpy(a).cuda()
pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)).cpu().numpy(); pk = pk.cpu().numpy()
streams = [torch.cuda.Stream() for _ in range(4)]
for kind in ("valid", "mix"):
    expect = np.ones(n, np.uint8)
    if kind == "mix": expect = workload.corrupt_for_verify(sig, pk, msg)
    ds, dp, dm = d(sig), d(pk), d(msg)

    def one():
        ok = ed.ed25519_verify_batch(ds, dp, dm); torch.cuda.synchronize(); return ok

    def tiled(L, S):
        outs, m = [], 1 << L
        for k in range(n >> L):
            with torch.cuda.stream(streams[k % S]):
                outs.append(ed.ed25519_verify_batch(ds[k * m:(k + 1) * m], dp[k * m:(k + 1) * m], dm[k * m:(k + 1) * m]))
        torch.cuda.synchronize()
        return torch.cat(outs)

    def timeit(fn):
        fn(); fn(); ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); out = fn(); ts.append((time.perf_counter() - t0) * 1e3)
        assert np.array_equal(out.cpu().numpy(), expect)
        return min(ts), sorted(ts)[len(ts) // 2]

    print(f"{kind:5s} one call            min {timeit(one)[0]:6.2f} ms")
    for L in (19, 18, 17, 16):
        line = f"{kind:5s} tiles of 2^{L}:"
        for S in (1, 2, 3, 4):
            line += f"   {S} stream{'s' if S > 1 else ' '} {timeit(lambda: tiled(L, S))[0]:6.2f}"
        print(line, flush=True)
