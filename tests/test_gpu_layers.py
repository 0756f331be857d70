"""Device-side layer KATs: every group of tests/golden/layer_kats.json - vectors produced
by the reference's static library (fld_mul/sq/inv/pow2523 lib/fld.c:209-280,578-709, sc_import/sc_mul lib/sc.c:191-266,
ed_import/ed_export/ed_scale_base/ed_dual_scale lib/ed.c:100-169,397-507, SHA-512 lib/sha512.c:127-210) - runs ON THE GPU
as a layer, through the probe library's entry point eddsa_amd_probe_layer, in the one-lane form and, where it exists, in the
four-lane (DPP) form, including operands at the documented limb bounds.  What the device toolchain makes of the limb
arithmetic and of the lane exchanges is the one thing the host build of the same source cannot vouch for (the
v_subrev_u32_dpp trap, DESIGN.md 7, was caught by a whole-operation KAT that happened to cover one limb).  Also here:
the fault injector that makes a checked HIP call of a verify pass fail."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H = bytes.fromhex
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493


def le(x, n=32):
    return int(x).to_bytes(n, "little")


def golden_msg(i):
    out, c = b"", 0
    while len(out) < i:
        out += hashlib.sha512(b"libeddsa-amd golden msg" + i.to_bytes(4, "little") + c.to_bytes(4, "little")).digest()
        c += 1
    return out[:i]


@pytest.fixture(scope="module")
def probe(engine):
    engine.debug_init(0, True)
    yield engine
    engine.debug_init(0, False)


def test_the_hooks_are_inert_until_armed_and_the_probe_checks_its_arguments(engine):
    engine.debug_init(0, False)
    assert engine.debug_fail_hip_call(1) == engine.HOOKS_OFF and engine.debug_fail_next_host_call() == engine.HOOKS_OFF
    engine.debug_init(0, True)
    # the layer probes are a library of their own (libeddsa_amd_probe.so), loaded beside the product: nothing to arm
    assert engine.debug_layer("fe_sq", [le(3)], 32) == [le(9)]
    with pytest.raises(engine.EddsaAmdError):                    # widths are checked per op
        engine.debug_layer("fe_mul", [bytes(32)], 32)
    with pytest.raises(engine.EddsaAmdError):                    # a SHA-512 item whose length field exceeds its slot
        engine.debug_layer("sha512", [(100).to_bytes(8, "little") + bytes(40)], 64)
    engine.debug_init(0, False)


def test_field_layer_kats_on_the_device(probe, golden):
    k = golden("layer_kats.json")
    got = probe.debug_layer("fe_mul", [H(a) + H(b) for a, b, _ in k["fld_mul"]], 32)
    assert [g.hex() for g in got] == [r for _, _, r in k["fld_mul"]]
    # the same products with the operands at the documented limits (csrc/fe25519.h:10-16): f = 7a (< 8u), g = 3b (< 3.36u)
    loose = probe.debug_layer("fe_mul_loose", [H(a) + H(b) + bytes([7, 3]) + bytes(6) for a, b, _ in k["fld_mul"]], 32)
    fold = lambda h: int.from_bytes(H(h), "little") % 2**255 + 19 * (H(h)[31] >> 7)   # noqa: E731  (fld_import: bit 255 -> +19)
    assert loose == [le(7 * fold(a) * 3 * fold(b) % P) for a, b, _ in k["fld_mul"]]
    for name, op in (("fld_sq", "fe_sq"), ("fld_inv", "fe_inv"), ("fld_pow2523", "fe_pow2523")):
        got = probe.debug_layer(op, [H(a) for a, _ in k[name]], 32)
        assert [g.hex() for g in got] == [r for _, r in k[name]], name
    assert len(k["fld_mul"]) >= 70 and len(k["fld_inv"]) >= 70


def test_field_layer_random_and_extreme_operands_on_the_device(probe):
    rng = np.random.default_rng(404)
    rb = lambda: bytes(rng.integers(0, 256, 32, dtype=np.uint8))       # noqa: E731
    ones = b"\xff" * 32
    pairs = [(rb(), rb()) for _ in range(4000)] + [(ones, ones), (ones, le(P - 1)), (le(P), le(P + 18)), (le(2**255 - 1), le(2**255 - 1))]
    val = lambda b: int.from_bytes(b, "little") % 2**255 + 19 * (b[31] >> 7)   # noqa: E731
    got = probe.debug_layer("fe_mul", [a + b for a, b in pairs], 32)
    assert got == [le(val(a) * val(b) % P) for a, b in pairs]
    got = probe.debug_layer("fe_sq", [a for a, _ in pairs], 32)
    assert got == [le(val(a) ** 2 % P) for a, _ in pairs]
    for ka in (1, 4, 7):
        for kb in (1, 2, 3):
            got = probe.debug_layer("fe_mul_loose", [a + b + bytes([ka, kb]) + bytes(6) for a, b in pairs[-300:]], 32)
            assert got == [le(ka * val(a) * kb * val(b) % P) for a, b in pairs[-300:]], (ka, kb)
    sample = pairs[:256] + pairs[-4:]
    got = probe.debug_layer("fe_inv", [a for a, _ in sample], 32)
    assert got == [le(pow(val(a) % P, P - 2, P)) for a, _ in sample]
    got = probe.debug_layer("fe_pow2523", [a for a, _ in sample], 32)
    assert got == [le(pow(val(a) % P, (P - 5) // 8, P)) for a, _ in sample]


def test_scalar_and_hash_layer_kats_on_the_device(probe, golden):
    k = golden("layer_kats.json")
    short = [(s, r) for s, r in k["sc_import"] if len(s) == 64]
    wide = [(s, r) for s, r in k["sc_import"] if len(s) == 128]
    assert short and wide and len(short) + len(wide) == len(k["sc_import"])
    assert [g.hex() for g in probe.debug_layer("sc_reduce32", [H(s) for s, _ in short], 32)] == [r for _, r in short]
    assert [g.hex() for g in probe.debug_layer("sc_reduce64", [H(s) for s, _ in wide], 32)] == [r for _, r in wide]
    got = probe.debug_layer("sc_muladd", [H(a) + H(b) + H(c) for a, b, c, _ in k["sc_muladd"]], 32)
    assert [g.hex() for g in got] == [r for _, _, _, r in k["sc_muladd"]]
    rng = np.random.default_rng(405)
    rnd = [bytes(rng.integers(0, 256, 96, dtype=np.uint8)) for _ in range(2000)]
    iv = lambda b: int.from_bytes(b, "little")                          # noqa: E731
    assert probe.debug_layer("sc_muladd", rnd, 32) == [le((iv(x[:32]) * iv(x[32:64]) + iv(x[64:])) % L) for x in rnd]
    assert probe.debug_layer("sc_reduce64", [x[:64] for x in rnd], 32) == [le(iv(x[:64]) % L) for x in rnd]
    # SHA-512 of lengths 0 .. 299 (and the block boundaries 111 / 112 / 127 / 128 / 239 / 240 among them)
    width = 8 + (max(n for n, _ in k["sha512"]) + 15) // 16 * 16
    items = [le(n, 8) + golden_msg(n) + bytes(width - 8 - n) for n, _ in k["sha512"]]
    got = probe.debug_layer("sha512", items, 64)
    assert [g.hex() for g in got] == [r for _, r in k["sha512"]]
    assert all(g == hashlib.sha512(golden_msg(n)).digest() for g, (n, _) in zip(got, k["sha512"]))


def test_group_layer_kats_on_the_device(probe, golden, oracle):
    k = golden("layer_kats.json")
    got = probe.debug_layer("ed_import_export", [H(a) for a, _ in k["ed_import_export"]], 33)
    assert [g[:32].hex() for g in got] == [r for _, r in k["ed_import_export"]]
    assert {g[32] for g in got} == {0, 1}                         # the fixture holds keys on and off the curve
    got = probe.debug_layer("ed_scale_base", [H(a) for a, _ in k["ed_scale_base"]], 32)
    assert [g.hex() for g in got] == [r for _, r in k["ed_scale_base"]]
    # the comb through the LDS image and the wave shuffle on whole waves of different scalars, against the oracle's k B
    import ctypes
    rng = np.random.default_rng(406)
    xs = [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(1500)] + [le(0), le(1), le(L - 1), le(L), le(2**256 - 1)]
    out = ctypes.create_string_buffer(32)
    want = []
    for x in xs:
        oracle.lib.orc_ed_scale_base(out, le(int.from_bytes(x, "little") % L))
        want.append(out.raw)
    assert probe.debug_layer("ed_scale_base", xs, 32) == want


def test_dual_scale_in_the_references_order_on_the_device_all_five_forms(probe, golden, oracle):
    """ed_dual_scale (lib/ed.c:455-507) replayed formula by formula: the literal chain, the uniform one, and the four-lane
    chain of the exact path (set-up and chain, DPP exchanges) - on curve points and on 'points' that are not on the curve,
    where the bytes depend on the exact sequence of formulas"""
    import ctypes
    k = golden("layer_kats.json")
    items = [H(s) + H(t) + H(q) for s, t, q, _ in k["ed_dual_scale"]]
    want = [r for _, _, _, r in k["ed_dual_scale"]]
    rng = np.random.default_rng(407)
    out = ctypes.create_string_buffer(32)
    extra, extra_want, off = [], [], 0
    for i in range(200):
        s, t, q = (bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(3))
        if i % 3 == 0:
            q = oracle.genpub(q)                                  # a curve point now and then
        oracle.lib.orc_ed_dual_scale(out, s, t, q)
        extra.append(s + t + q); extra_want.append(out.raw.hex())
    flags = probe.debug_layer("ed_import_export", [x[64:] for x in extra], 33)
    off = sum(1 for f in flags if f[32] == 0)
    assert off > 50                                               # plenty of off-curve inputs among them
    for form in (0, 2, 1, 3, 4):                                  # 3: one lane per item over table entries, in stretches; 4: two items per lane (k_verify_exact_lane_chain)
        got = probe.debug_layer("ed_dual_scale", items + extra, 32, form=form)
        bad = [i for i, (g, w) in enumerate(zip(got, want + extra_want)) if g.hex() != w]
        assert not bad, (form, bad[:10])
    # whole waves and a ragged tail through the four-lane form: 16 items per wave
    for n in (1, 15, 16, 17, 33):
        assert [g.hex() for g in probe.debug_layer("ed_dual_scale", extra[:n], 32, form=1)] == extra_want[:n], n


def test_windowed_steps_one_lane_and_four_lanes_on_the_device(probe, oracle):
    """the two steps of the windowed evaluation - doubling, addition of a table entry - with one lane per item
    (ge25519.h) and with a coordinate per lane (quad_lanes.h: quad_dbl, quad_add_entry): enc(2 P + k B) for P = a B"""
    import ctypes
    rng = np.random.default_rng(408)
    out = ctypes.create_string_buffer(32)

    def sb(x):
        oracle.lib.orc_ed_scale_base(out, le(x % L))
        return out.raw

    items, want = [], []
    for i in range(300):
        a = int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "little") % L
        kk = int(rng.integers(0, 32769)) if i > 8 else (0, 1, 2, 32768, 32767, 255, 256, 4096, 8)[i]
        if i in (3, 4):
            a = (0, 1)[i - 3]                                     # P = neutral element, P = B
        items.append(sb(a) + le(kk, 2) + bytes(6)); want.append(sb(2 * a + kk))
    for form in (0, 1):
        got = probe.debug_layer("ge_dbl_add", items, 32, form=form)
        bad = [i for i, (g, w) in enumerate(zip(got, want)) if g != w]
        assert not bad, (form, bad[:10])


def test_a_failed_hip_call_inside_a_verify_pass_surfaces_as_an_error(probe, oracle):
    """every event record, stream wait, work-list reset and launch of edk_verify is checked; the fault
    hook makes the nth of them fail, for every n: the pass returns a negative value (no silent race on the verdict
    bytes), and the engine is usable afterwards"""
    import torch
    import workload
    for n in (300, 5000, 40000, (1 << 18) + 5):                   # one pass on four different routes
        sk, msg = workload.sign_inputs(n, seed=9, config=2)
        pk = oracle.genpub_batch(sk)
        sig = oracle.sign_batch(sk, pk, msg, 32) if n <= 5000 else probe.ed25519_sign_batch(sk, pk, msg)
        expect = workload.corrupt_for_verify(sig, pk, msg)
        d = lambda a: torch.from_numpy(a).cuda()                  # noqa: E731
        ds, dp, dm = d(sig), d(pk), d(msg)
        assert probe.debug_fail_hip_call(0) == 0
        assert np.array_equal(probe.ed25519_verify_batch(ds, dp, dm).cpu().numpy(), expect)
        calls = probe.debug_hip_calls()
        assert 8 <= calls <= 20, calls
        for nth in range(1, calls + 1):
            assert probe.debug_fail_hip_call(nth) == 0
            with pytest.raises(probe.EddsaAmdError):
                probe.ed25519_verify_batch(ds, dp, dm)
            torch.cuda.synchronize()
        assert probe.debug_fail_hip_call(calls + 1) == 0          # armed beyond the pass: nothing fails
        assert np.array_equal(probe.ed25519_verify_batch(ds, dp, dm).cpu().numpy(), expect)
        probe.debug_fail_hip_call(0)
        # the host-pointer path reports it too (and single-item callers would abort: no silent wrong verdict)
        probe.debug_fail_hip_call(3)
        with pytest.raises(probe.EddsaAmdError):
            probe.ed25519_verify_batch(sig, pk, msg)
        probe.debug_fail_hip_call(0)
        assert np.array_equal(probe.ed25519_verify_batch(sig, pk, msg), expect)
    # the opt-in batch verification checks its calls the same way (hash tree on the side stream, combination, copy of the
    # group verdicts): a failure anywhere in it is an error return as well
    probe.set_rlc_min_items(0)
    try:
        vs, vp, vm = d(sig), d(pk), d(msg)
        probe.debug_fail_hip_call(0)
        ok0 = probe.ed25519_verify_batch_rlc(vs, vp, vm).cpu().numpy()
        calls = probe.debug_hip_calls()
        assert np.array_equal(ok0, expect) and calls >= 15
        for nth in range(1, calls + 1, 3):
            probe.debug_fail_hip_call(nth)
            with pytest.raises(probe.EddsaAmdError):
                probe.ed25519_verify_batch_rlc(vs, vp, vm)
            torch.cuda.synchronize()
        probe.debug_fail_hip_call(0)
        assert np.array_equal(probe.ed25519_verify_batch_rlc(vs, vp, vm).cpu().numpy(), expect)
    finally:
        probe.set_rlc_min_items(probe.RLC_MIN_ITEMS_DEFAULT)
    # and nothing on the clean-up paths behind those failures (waiting for what was queued, releasing the slot) failed
    # in turn: HIP calls that have nobody to report to are counted, not ignored
    assert probe.debug_teardown_errors() == (0, 0)


def test_garbage_keys_fill_the_exact_path_beyond_its_slots(probe, oracle):
    """ed_import never fails (lib/ed.c:100-149): a caller can send nothing but garbage keys, and about half of all
    32-byte strings are no curve point.  140 000 genuine signatures under random keys: ~70 000 work-list entries, more
    than the 65 536 the exact path has in flight at once (the strided rounds of k_verify_exact_quad; until round 4 the
    rest fell to a spilling one-lane kernel) - every verdict against the oracle"""
    import torch
    import workload
    n = 140000
    sk, msg = workload.sign_inputs(n, seed=12, config=2)
    pk = probe.ed25519_genpub_batch(sk)
    sig = probe.ed25519_sign_batch(sk, pk, msg)
    rng = np.random.default_rng(409)
    bad = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    keep = np.arange(n) % 7 == 0                                  # one genuine key in seven stays
    bad[keep] = pk[keep]
    flags = probe.debug_layer("ed_import_export", [bytes(r) for r in bad[:4096]], 33)
    assert 0.4 < sum(1 for f in flags if f[32] == 0) / 4096 < 0.5
    want = oracle.verify_batch(sig, bad, msg, 32)
    assert want.sum() == keep.sum()
    d = lambda a: torch.from_numpy(a).cuda()                      # noqa: E731
    assert np.array_equal(probe.ed25519_verify_batch(d(sig), d(bad), d(msg)).cpu().numpy(), want)
    assert np.array_equal(probe.ed25519_verify_batch(sig, bad, msg), want)          # host pipeline: two chunks
