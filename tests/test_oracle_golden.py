"""The oracle (CPU restatement) against every committed golden vector: the reference's own x25519
table, the regenerated ed25519 table, the edge-case verdicts and the layer KATs (tests/golden/,
written by tools/gen_golden.py from the compiled reference)."""
import ctypes
import hashlib

import numpy as np

from gen_golden import golden_msg  # tools/ is on sys.path (conftest)

H = bytes.fromhex


def _call(oracle, name, out_len, *ins):
    out = ctypes.create_string_buffer(out_len)
    getattr(oracle.lib, name)(out, *ins)
    return out.raw


def test_x25519_reference_table(oracle, golden):
    raw = golden("x25519_table.bin")
    assert len(raw) == 1024 * 96
    hi = 0
    for i in range(1024):
        pt, sc, res = raw[96 * i:96 * i + 32], raw[96 * i + 32:96 * i + 64], raw[96 * i + 64:96 * i + 96]
        assert oracle.x25519(sc, pt) == res, i
        hi += pt[31] >> 7
    assert hi == 508          # 508 table points have bit 255 set, and it is NOT masked


def test_ed25519_table(oracle, golden):
    raw = golden("ed25519_table.bin")
    for i in range(1024):
        sk, pk, sig = raw[128 * i:128 * i + 32], raw[128 * i + 32:128 * i + 64], raw[128 * i + 64:128 * i + 128]
        msg = golden_msg(i)
        assert oracle.genpub(sk) == pk, i
        assert oracle.sign(sk, pk, msg) == sig, i
        assert oracle.verify(sig, pk, msg), i
    assert raw[32:64].hex().startswith("d75a9801")      # RFC 8032 7.1 TEST 1


def test_verify_edge_cases(oracle, golden):
    cases = golden("verify_edges.json")
    assert len(cases) > 250
    for c in cases:
        assert oracle.verify(H(c["sig"]), H(c["pub"]), H(c["msg"])) == c["accept"], c["name"]
    names = {c["name"]: c["accept"] for c in cases}
    assert names["S+1l (S not range-checked: accepted)"] is True
    assert names["S-1"] is False


def test_layer_kats(oracle, golden):
    k = golden("layer_kats.json")
    for a, b, r in k["fld_mul"]:
        assert _call(oracle, "orc_fld_mul", 32, H(a), H(b)).hex() == r
    for name, fn in (("fld_sq", "orc_fld_sq"), ("fld_inv", "orc_fld_inv"), ("fld_pow2523", "orc_fld_pow2523"),
                     ("ed_import_export", "orc_ed_import_export"), ("ed_scale_base", "orc_ed_scale_base"),
                     ("pk_to_x", "orc_pk_ed25519_to_x25519"), ("sk_to_x", "orc_sk_ed25519_to_x25519"),
                     ("x25519_base", "orc_x25519_base")):
        for a, r in k[name]:
            assert _call(oracle, fn, 32, H(a)).hex() == r, (name, a)
    for s, r in k["sc_import"]:
        assert _call(oracle, "orc_sc_reduce_bytes", 32, H(s), ctypes.c_size_t(len(s) // 2)).hex() == r
    for a, b, c, r in k["sc_muladd"]:
        assert _call(oracle, "orc_sc_muladd", 32, H(a), H(b), H(c)).hex() == r
    for s, t, q, r in k["ed_dual_scale"]:
        assert _call(oracle, "orc_ed_dual_scale", 32, H(s), H(t), H(q)).hex() == r
    for n, r in k["sha512"]:
        m = golden_msg(n)
        assert oracle.sha512(m).hex() == r == hashlib.sha512(m).hexdigest()


def test_comb_table_points(oracle, golden):
    """the oracle derives the comb table the reference ships as generated data; compare by value"""
    pts = golden("comb_points.bin")
    tab = ctypes.create_string_buffer(32 * 8 * 96)
    oracle.lib.orc_ed_lookup_bytes(tab)
    p = 2**255 - 19
    inv2 = pow(2, p - 2, p)
    for e in range(256):
        ymx = int.from_bytes(tab.raw[96 * e:96 * e + 32], "little")
        ypx = int.from_bytes(tab.raw[96 * e + 32:96 * e + 64], "little")
        y, x = (ypx + ymx) * inv2 % p, (ypx - ymx) * inv2 % p
        enc = (y | (x & 1) << 255).to_bytes(32, "little")
        assert enc == pts[32 * e:32 * e + 32], e


def test_batch_digest_small(oracle, golden):
    import workload
    d = golden("batch_digests.json")
    n = 1 << 14
    sc, pt = workload.x25519_inputs(n)
    assert hashlib.sha512(oracle.x25519_batch(sc, pt).tobytes()).hexdigest() == d["x25519_2^14"]
    sk, msg = workload.sign_inputs(n)
    pk = oracle.genpub_batch(sk)
    assert hashlib.sha512(pk.tobytes()).hexdigest() == d["genpub_2^14"]
    assert hashlib.sha512(oracle.sign_batch(sk, pk, msg, 32).tobytes()).hexdigest() == d["sign_2^14"]
    sk, msg = workload.sign_inputs(n, seed=1, config=2)
    pk = oracle.genpub_batch(sk)
    sig = oracle.sign_batch(sk, pk, msg, 32)
    expect = workload.corrupt_for_verify(sig, pk, msg)
    v = d["verify_2^14"]
    assert hashlib.sha512(sig.tobytes() + pk.tobytes() + msg.tobytes()).hexdigest() == v["inputs_sha512"]
    ok = oracle.verify_batch(sig, pk, msg, 32)
    assert np.array_equal(ok, expect) and int(ok.sum()) == v["accepted"]
    assert hashlib.sha512(ok.tobytes()).hexdigest() == v["verdicts_sha512"]
