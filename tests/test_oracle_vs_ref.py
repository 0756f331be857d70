"""Differential pinning of the oracle against the compiled reference itself (oracle/_ref, built
from the reference sources in place).  Skipped where neither the prebuilt library nor the
reference checkout exists."""
import ctypes

import numpy as np

L = 2**252 + 27742317777372353535851937790883648493
P = 2**255 - 19


class Ed(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64 * 5) for n in ("x", "y", "t", "z")]


def _ref(lib, name, out_len, *args):
    out = ctypes.create_string_buffer(out_len)
    getattr(lib, name)(out, *args)
    return out.raw


def _inputs(rng, n):
    special = [0, 1, 2, 8, 9, P - 1, P, P + 1, 2**255 - 1, 2**255, 2**255 + 18, 2**256 - 1, L - 1, L, L + 1]
    out = [int(s).to_bytes(32, "little") for s in special]
    while len(out) < n:
        out.append(bytes(rng.integers(0, 256, 32, dtype=np.uint8)))
    return out


def test_protocol_functions(oracle, reflib):
    rng = np.random.default_rng(11)
    ins = _inputs(rng, 120)
    for i, a in enumerate(ins):
        b = ins[(5 * i + 7) % len(ins)]
        assert oracle.x25519(a, b) == _ref(reflib, "x25519", 32, a, b)
        assert oracle.x25519_base(a) == _ref(reflib, "x25519_base", 32, a)
        assert oracle.pk_to_x(a) == _ref(reflib, "pk_ed25519_to_x25519", 32, a)
        assert oracle.sk_to_x(a) == _ref(reflib, "sk_ed25519_to_x25519", 32, a)
        pk = _ref(reflib, "ed25519_genpub", 32, a)
        assert oracle.genpub(a) == pk
        msg = bytes(rng.integers(0, 256, i * 3, dtype=np.uint8))
        sig = _ref(reflib, "ed25519_sign", 64, a, pk, msg, ctypes.c_size_t(len(msg)))
        assert oracle.sign(a, pk, msg) == sig
        # sign hashes the caller's pub without checking it
        assert oracle.sign(a, b, msg) == _ref(reflib, "ed25519_sign", 64, a, b, msg, ctypes.c_size_t(len(msg)))
        for s, p_, m in ((sig, pk, msg), (sig, b, msg), (a + b, pk, msg), (sig[:32] + a, pk, msg)):
            assert oracle.verify(s, p_, m) == bool(reflib.ed25519_verify(s, p_, m, ctypes.c_size_t(len(m))))


def test_dual_scale_including_off_curve_points(oracle, reflib):
    """ed_dual_scale on arbitrary 32-byte 'points' (about half decode to off-curve pairs): the oracle
    applies the reference's formulas in the reference's order, so even those agree byte for byte"""
    rng = np.random.default_rng(12)
    ins = _inputs(rng, 200)
    for i in range(len(ins)):
        s, t, q = ins[i], ins[(3 * i + 1) % len(ins)], ins[(7 * i + 2) % len(ins)]
        S = (ctypes.c_int64 * 5)(); T = (ctypes.c_int64 * 5)(); Q = Ed(); R = Ed()
        reflib.sc_import(S, s, ctypes.c_size_t(32)); reflib.sc_import(T, t, ctypes.c_size_t(32))
        reflib.ed_import(ctypes.byref(Q), q)
        reflib.ed_dual_scale(ctypes.byref(R), S, T, ctypes.byref(Q))
        want = _ref(reflib, "ed_export", 32, ctypes.byref(R))
        got = ctypes.create_string_buffer(32)
        oracle.lib.orc_ed_dual_scale(got, s, t, q)
        assert got.raw == want, i


def test_field_and_scalar_layers(oracle, reflib):
    rng = np.random.default_rng(13)
    ins = _inputs(rng, 150)
    for i, a in enumerate(ins):
        b = ins[(11 * i + 3) % len(ins)]
        fa = (ctypes.c_int64 * 5)(); fb = (ctypes.c_int64 * 5)(); fo = (ctypes.c_int64 * 5)()
        reflib.fld_import(fa, a); reflib.fld_import(fb, b)
        for name, orc, args in (("fld_mul", "orc_fld_mul", (fa, fb)), ("fld_sq", "orc_fld_sq", (fa,)),
                                ("fld_inv", "orc_fld_inv", (fa,)), ("fld_pow2523", "orc_fld_pow2523", (fa,))):
            getattr(reflib, name)(fo, *args)
            want = _ref(reflib, "fld_export", 32, fo)
            got = ctypes.create_string_buffer(32)
            getattr(oracle.lib, orc)(got, a, *([b] if len(args) == 2 else []))
            assert got.raw == want, (name, i)
        for blob in (a, a + b):
            x = (ctypes.c_int64 * 5)()
            reflib.sc_import(x, blob, ctypes.c_size_t(len(blob)))
            want = _ref(reflib, "sc_export", 32, x)
            got = ctypes.create_string_buffer(32)
            oracle.lib.orc_sc_reduce_bytes(got, blob, ctypes.c_size_t(len(blob)))
            assert got.raw == want
            assert int.from_bytes(want, "little") == int.from_bytes(blob, "little") % L


def test_both_limb_builds_agree(oracle):
    """the 32-bit-limb build of the reference gives the same bytes"""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libeddsa_ref32.so")
    if not os.path.exists(path):
        import pytest
        pytest.skip("32-bit-limb reference build absent")
    r32 = ctypes.CDLL(path)
    rng = np.random.default_rng(14)
    for a in _inputs(rng, 60):
        b = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        assert oracle.x25519(a, b) == _ref(r32, "x25519", 32, a, b)
        assert oracle.genpub(a) == _ref(r32, "ed25519_genpub", 32, a)
