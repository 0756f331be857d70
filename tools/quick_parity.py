#!/usr/bin/env python3
"""Ad-hoc GPU parity probe (development aid; the real parity tests live in tests/).
Runs every batch entry point of libeddsa_amd.so against the oracle on seeded random inputs."""
import ctypes, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
amd = ctypes.CDLL(os.path.join(ROOT, "libeddsa_amd", "libeddsa_amd.so"))
orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
P = ctypes.c_void_p
def ptr(a): return a.ctypes.data_as(P)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mlen = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rng = np.random.default_rng(7)
rc = amd.eddsa_amd_init(0)
assert rc == 0, rc
thr = os.cpu_count()

def check(name, a, b):
    bad = np.nonzero(np.any(a.reshape(len(a), -1) != b.reshape(len(b), -1), axis=1))[0]
    print(f"{name:28s} n={len(a)} mismatches={len(bad)}" + (f" first={bad[:5]}" if len(bad) else ""))
    return len(bad) == 0

ok_all = True
# x25519
sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); pt = rng.integers(0, 256, (n, 32), dtype=np.uint8)
pt[0] = 0; pt[1] = 0; pt[1, 0] = 1; pt[2] = 255; pt[3] = 255; pt[3, 31] = 127; pt[4, :] = 0; pt[4, 0] = 9
g = np.zeros((n, 32), np.uint8); o = np.zeros((n, 32), np.uint8)
t0 = time.time(); rc = amd.x25519_batch(ptr(g), ptr(sc), ptr(pt), ctypes.c_size_t(n)); t1 = time.time(); assert rc == 0, rc
orc.orc_x25519_batch(ptr(o), ptr(sc), ptr(pt), ctypes.c_size_t(n), thr)
ok_all &= check("x25519", g, o)
# genpub
sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
pk_g = np.zeros((n, 32), np.uint8); pk_o = np.zeros((n, 32), np.uint8)
rc = amd.ed25519_genpub_batch(ptr(pk_g), ptr(sk), ctypes.c_size_t(n)); assert rc == 0, rc
orc.orc_ed25519_genpub_batch(ptr(pk_o), ptr(sk), ctypes.c_size_t(n), thr)
ok_all &= check("ed25519_genpub", pk_g, pk_o)
# sign
msg = rng.integers(0, 256, (n, max(mlen, 1)), dtype=np.uint8)[:, :mlen].copy()
sg_g = np.zeros((n, 64), np.uint8); sg_o = np.zeros((n, 64), np.uint8)
rc = amd.ed25519_sign_batch(ptr(sg_g), ptr(sk), ptr(pk_o), ptr(msg), None, ctypes.c_size_t(mlen), ctypes.c_size_t(n)); assert rc == 0, rc
orc.orc_ed25519_sign_batch(ptr(sg_o), ptr(sk), ptr(pk_o), ptr(msg), ctypes.c_size_t(mlen), ctypes.c_size_t(n), thr)
ok_all &= check("ed25519_sign", sg_g, sg_o)
# verify: valid, corrupted, garbage
sig = sg_o.copy(); pub = pk_o.copy(); m2 = msg.copy()
for i in range(n):
    k = i % 8
    if k == 1: sig[i, rng.integers(0, 32)] ^= 1 << rng.integers(0, 8)
    elif k == 2: sig[i, 32 + rng.integers(0, 32)] ^= 1 << rng.integers(0, 8)
    elif k == 3: pub[i, rng.integers(0, 32)] ^= 1 << rng.integers(0, 8)
    elif k == 4 and mlen: m2[i, rng.integers(0, mlen)] ^= 1 << rng.integers(0, 8)
    elif k == 5: sig[i] = rng.integers(0, 256, 64); pub[i] = rng.integers(0, 256, 32)
    elif k == 6:  # S + l still verifies in the reference
        L = 2**252 + 27742317777372353535851937790883648493
        s = int.from_bytes(sig[i, 32:].tobytes(), "little") + L
        if s < 2**256: sig[i, 32:] = np.frombuffer(s.to_bytes(32, "little"), np.uint8)
v_g = np.zeros(n, np.uint8); v_o = np.zeros(n, np.uint8)
t0 = time.time(); rc = amd.ed25519_verify_batch(ptr(v_g), ptr(sig), ptr(pub), ptr(m2), None, ctypes.c_size_t(mlen), ctypes.c_size_t(n)); t1 = time.time(); assert rc == 0, rc
orc.orc_ed25519_verify_batch(ptr(v_o), ptr(sig), ptr(pub), ptr(m2), ctypes.c_size_t(mlen), ctypes.c_size_t(n), thr)
ok_all &= check("ed25519_verify", v_g, v_o)
print("   accepted:", int(v_o.sum()), "of", n, f"(gpu host-call {t1-t0:.3f}s)")
# x25519_base, conversions
for name, gf, of in (("x25519_base", amd.x25519_base_batch, orc.orc_x25519_base),
                     ("pk_ed25519_to_x25519", amd.pk_ed25519_to_x25519_batch, orc.orc_pk_ed25519_to_x25519),
                     ("sk_ed25519_to_x25519", amd.sk_ed25519_to_x25519_batch, orc.orc_sk_ed25519_to_x25519)):
    inp = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    g = np.zeros((n, 32), np.uint8); o = np.zeros((n, 32), np.uint8)
    rc = gf(ptr(g), ptr(inp), ctypes.c_size_t(n)); assert rc == 0, rc
    m = min(n, 2048)
    for i in range(m): of(ptr(o[i]), ptr(inp[i]))
    ok_all &= check(name, g[:m], o[:m])
print("ALL OK" if ok_all else "MISMATCH")
sys.exit(0 if ok_all else 1)
