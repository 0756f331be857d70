#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer entry points (the lanes of host_pipe.c), host buffer to host
buffer, 2^20 items: from ordinary (malloc / numpy) memory, staged by the copier pool with 0..8 helper threads, and
from page-locked memory (eddsa_amd_host_alloc), which is used in place.  DESIGN.md quotes these; bench.py's `value`
never includes transfers.      python tools/host_path_bench.py [log2n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401  (one HIP runtime per process: torch's)
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
import workload

ed.init(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
sk, msg = workload.sign_inputs(n, seed=1, config=2)
pk = ed.ed25519_genpub_batch(sk)
sig = ed.ed25519_sign_batch(sk, pk, msg)
expect = workload.corrupt_for_verify(sig, pk, msg)
sc, pt = workload.x25519_inputs(n)
want_x = ed.x25519_batch(sc, pt)
want_sig = ed.ed25519_sign_batch(sk, pk, msg)


def timeit(fn, reps=7):
    fn(); fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    return best, out


def pin(a):
    b = ed.host_array(a.shape)
    b[...] = a
    return b


def report(label, conv):
    s_, p_, m_ = conv(sig), conv(pk), conv(msg)
    dt, ok = timeit(lambda: ed.ed25519_verify_batch(s_, p_, m_))
    assert np.array_equal(ok, expect)
    line = f"{label:22s} verify {n/dt/1e6:7.1f} M/s ({dt*1e3:6.2f} ms)"
    a_, b_ = conv(sc), conv(pt)
    dt, out = timeit(lambda: ed.x25519_batch(a_, b_))
    assert np.array_equal(out, want_x)
    line += f"   x25519 {n/dt/1e6:7.1f} M/s ({dt*1e3:6.2f} ms)"
    k_, q_, mm_ = conv(sk), conv(pk), conv(msg)
    dt, out = timeit(lambda: ed.ed25519_sign_batch(k_, q_, mm_))
    assert np.array_equal(out, want_sig)
    line += f"   sign {n/dt/1e6:7.1f} M/s ({dt*1e3:6.2f} ms)"
    print(line, flush=True)


print(f"host buffer to host buffer, {n} items, best of 7 (outputs land in ordinary numpy arrays)")
for t in (0, 2, 4, 6, 8):
    ed.set_host_threads(t)
    ed.shutdown(); ed.init(0)                      # the pool restarts with the new size
    report(f"malloc, {t} helpers", lambda a: a)
ed.set_host_threads(4)
ed.shutdown(); ed.init(0)
report("page-locked inputs", pin)
# kernel-only, for the ratio
d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
ds, dp, dm = d(sig), d(pk), d(msg)
def dev_verify():
    ok = ed.ed25519_verify_batch(ds, dp, dm); torch.cuda.synchronize(); return ok
dt, _ = timeit(dev_verify)
print(f"{'device-resident':22s} verify {n/dt/1e6:7.1f} M/s ({dt*1e3:6.2f} ms)")
