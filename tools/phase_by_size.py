"""HIP-event phase times of one verify pass by size, for valid signatures only and for the config-2 mix (whose
off-curve keys bring the exact path's kernels beside the main kernel)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import libeddsa_amd as ed, workload
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
ed.init(0)
d = lambda a: torch.from_numpy(a).cuda()
for l in ([int(a) for a in sys.argv[1:]] or (20, 19, 18, 17, 16)):
    n = 1 << l
    sk, msg = workload.sign_inputs(n, seed=1, config=2)
    pk = ed.ed25519_genpub_batch(d(sk)); sig = ed.ed25519_sign_batch(d(sk), pk, d(msg)).cpu().numpy(); pk = pk.cpu().numpy()
    for clean in (True, False):
        if not clean: workload.corrupt_for_verify(sig, pk, msg)
        ds, dp, dm = d(sig), d(pk), d(msg)
        for _ in range(3): ed.ed25519_verify_batch(ds, dp, dm)
        ed.set_profiling(True)
        for _ in range(10): ed.ed25519_verify_batch(ds, dp, dm)
        torch.cuda.synchronize()
        ph = ed.verify_phase_ms(); ed.set_profiling(False)
        print(f"2^{l} {'valid ' if clean else 'mix   '} prepare(+halve) {ph[0]:.3f}  main {ph[1]:.3f}  finish {ph[2]:.3f} ms   main per 2^18 items {ph[1] * (1 << 18) / n:.3f}")
