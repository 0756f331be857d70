#!/usr/bin/env python3
"""Constant-time probe: run the three secret-scalar point kernels (k_x25519_base_point,
k_genpub_point, k_sign_point; <1>: one lane per item, <4>: the four-lane form of small passes) on 2^16 and on 2^12 secrets of one CLASS - zero | ones | random | mixed (every lane of a
wave a different class) - so that tools/ct_counters.sh can compare the hardware counters of the
launches across classes.  x25519_base takes the scalar itself (clamped, x25519.c:163-166), so `zero`
and `ones` really are the extreme digit strings there; genpub and sign hash the key first."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import libeddsa_amd as ed
ed.use_debug_library()   # the hooks (route selection, phase timings, traces) live in libeddsa_amd_debug.so

cls = sys.argv[1] if len(sys.argv) > 1 else "random"
n = 1 << 16
rng = np.random.default_rng(1)
if cls == "zero":
    sec = np.zeros((n, 32), np.uint8)
elif cls == "ones":
    sec = np.full((n, 32), 255, np.uint8)
elif cls == "mixed":
    sec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sec[0::3] = 0
    sec[1::3] = 255
else:
    sec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
msg = np.zeros((n, 32), np.uint8)
ed.init(0)
d_sec, d_msg = torch.from_numpy(sec).cuda(), torch.from_numpy(msg).cuda()
for _ in range(3):
    for m in (n, 1 << 12):
        ed.x25519_base_batch(d_sec[:m])
        pk = ed.ed25519_genpub_batch(d_sec[:m])
        ed.ed25519_sign_batch(d_sec[:m], pk, d_msg[:m])
torch.cuda.synchronize()
print("ct_probe", cls, "done")
