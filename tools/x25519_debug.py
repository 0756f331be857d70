import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import libeddsa_amd as ed
from conftest import Oracle
ed.init(0)
import ctypes
o = Oracle(ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle", "liboracle.so")))
rng = np.random.default_rng(3)
n = 64
sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); pt = rng.integers(0, 256, (n, 32), dtype=np.uint8)
pt[:8] = 0; pt[:8, 0] = 9
sc[0] = 0; sc[1] = 0; sc[1, 0] = 8; sc[2] = 0; sc[2,31] = 0x40; sc[3] = 255
want = o.x25519_batch(sc, pt)
got = ed.x25519_batch(torch.from_numpy(sc).cuda(), torch.from_numpy(pt).cuda()).cpu().numpy()
bad = [i for i in range(n) if not np.array_equal(want[i], got[i])]
print("bad", len(bad), bad[:20])
for i in range(n):
    one = ed.x25519_batch(torch.from_numpy(sc[i:i+1].copy()).cuda(), torch.from_numpy(pt[i:i+1].copy()).cuda()).cpu().numpy()
    if not np.array_equal(one[0], want[i]): print("single bad", i, end="; ")
print()
