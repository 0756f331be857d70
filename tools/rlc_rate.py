#!/usr/bin/env python3
"""Rate of the opt-in batch verification against the per-item kernels on an all-valid 2^20 batch
(VERDICT r01 next-step 3), inputs resident in HBM, HIP events on the launch stream; one JSON line."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so
import workload

ed.init(0)
n = 1 << 20
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sk, msg = workload.sign_inputs(n, seed=1, config=2)
d_sk, d_msg = torch.from_numpy(sk).cuda(), torch.from_numpy(msg).cuda()
pk = ed.ed25519_genpub_batch(d_sk)
sig = ed.ed25519_sign_batch(d_sk, pk, d_msg)


def timed(fn):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out


ms_item, ok1 = timed(lambda: ed.ed25519_verify_batch(sig, pk, d_msg, msg_len=32))
ms_rlc, ok2 = timed(lambda: ed.ed25519_verify_batch_rlc(sig, pk, d_msg, msg_len=32))
_, st = ed.ed25519_verify_batch_rlc(sig, pk, d_msg, msg_len=32, return_stats=True)
# canonical work of the combination per item: 48 mixed additions (7 M) + 2 decompressions (255 S + 19 M each)
# + the 8 doublings per window and bucket, amortised: 48 * 8 * 128 / 8192 doublings (4 S + 4 M)
mul32 = 48 * 700 + 2 * (255 * 55 + 19 * 100) + int(48 * 8 * 128 / 8192 * (4 * 55 + 4 * 100))
print(json.dumps({"workload": "2^20 all-valid signatures (config 2 keys and messages, no corruption), 32-byte messages",
                  "per_item": {"ms": ms_item, "verifies_per_s": n / ms_item * 1e3},
                  "rlc": {"ms": ms_rlc, "verifies_per_s": n / ms_rlc * 1e3, "stats": st,
                          "canonical_mul32_per_item": mul32, "achieved_Tmul32_s": n * mul32 / ms_rlc * 1e3 / 1e12},
                  "speedup": ms_item / ms_rlc, "all_accepted": bool(ok1.all()) and bool(ok2.all())}))
