"""The DEVICE source (libeddsa_amd/csrc/lanes.h + field/scalar/hash/group headers), compiled for the
host CPU with -DED_HOST_CHECK, against the golden vectors and the oracle.  Two things are proven
without a GPU: (1) the kernels' algorithms (windowed double-scalar multiplication with signed
digits, comb with constant-time select, ladder, Barrett, SHA-512 padding, permissive import) give
the reference's bytes; (2) no limb precondition and no 64-bit column sum is violated on any input
driven through here, including the extreme ones (every assertion of fe25519.h is live)."""
import ctypes
import hashlib

import numpy as np

from gen_golden import golden_msg

H = bytes.fromhex
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
SZ = ctypes.c_size_t


def call(h, name, out_len, *args):
    out = ctypes.create_string_buffer(out_len)
    getattr(h, name)(out, *args)
    return out.raw


def no_violations(h):
    assert h.hc_violations() == 0, h.hc_first_violation()


def le(x, n=32):
    return int(x).to_bytes(n, "little")


EXTREME = [0, 1, 2, 9, 19, P - 1, P, P + 1, 2**255 - 19 - 1, 2**255 - 1, 2**255, 2**255 + 18, 2**256 - 1,
           (1 << 26) - 1, 1 << 26, ((1 << 255) - 1) ^ ((1 << 128) - 1), int("aa" * 32, 16), int("55" * 32, 16),
           L - 1, L, L + 1, 2**252, 2**253 - 1]


def test_x25519_table_and_extremes(hostcheck, oracle, golden):
    raw = golden("x25519_table.bin")
    for i in range(0, 1024, 3):
        pt, sc, res = raw[96 * i:96 * i + 32], raw[96 * i + 32:96 * i + 64], raw[96 * i + 64:96 * i + 96]
        assert call(hostcheck, "hc_x25519", 32, sc, pt) == res, i
    for a in EXTREME:
        for b in EXTREME[::2]:
            assert call(hostcheck, "hc_x25519", 32, le(a), le(b)) == oracle.x25519(le(a), le(b))
    no_violations(hostcheck)


def test_ed25519_table(hostcheck, golden):
    raw = golden("ed25519_table.bin")
    for i in list(range(0, 1024, 7)) + [47, 48, 111, 112, 1023]:
        sk, pk, sig = raw[128 * i:128 * i + 32], raw[128 * i + 32:128 * i + 64], raw[128 * i + 64:128 * i + 128]
        msg = golden_msg(i)
        assert call(hostcheck, "hc_genpub", 32, sk) == pk, i
        assert call(hostcheck, "hc_sign", 64, sk, pk, msg, SZ(len(msg))) == sig, i
        assert hostcheck.hc_verify(sig, pk, msg, SZ(len(msg))) == 1, i
    no_violations(hostcheck)


def test_verify_edge_cases(hostcheck, golden):
    for c in golden("verify_edges.json"):
        msg = H(c["msg"])
        assert hostcheck.hc_verify(H(c["sig"]), H(c["pub"]), msg, SZ(len(msg))) == int(c["accept"]), c["name"]
    no_violations(hostcheck)


def test_layer_kats_and_loose_operands(hostcheck, golden):
    k = golden("layer_kats.json")
    for a, b, r in k["fld_mul"]:
        assert call(hostcheck, "hc_fe_mul", 32, H(a), H(b)).hex() == r
        # the same product with operands at the documented limits: f = 7a (< 8u), g = 3b (< 3.36u)
        want = (7 * int.from_bytes(H(a), "little") * 3 * int.from_bytes(H(b), "little")) % P
        assert call(hostcheck, "hc_fe_mul_loose", 32, H(a), 7, H(b), 3) == le(want)
    for name, fn in (("fld_sq", "hc_fe_sq"), ("fld_inv", "hc_fe_inv"), ("fld_pow2523", "hc_fe_pow2523"),
                     ("ed_import_export", "hc_ed_import_export"), ("ed_scale_base", "hc_scale_base"),
                     ("pk_to_x", "hc_pk_to_x"), ("sk_to_x", "hc_sk_to_x"), ("x25519_base", "hc_x25519_base")):
        for a, r in k[name]:
            assert call(hostcheck, fn, 32, H(a)).hex() == r, (name, a)
    for s, r in k["sc_import"]:
        assert call(hostcheck, "hc_sc_reduce", 32, H(s), SZ(len(s) // 2)).hex() == r
    for a, b, c, r in k["sc_muladd"]:
        assert call(hostcheck, "hc_sc_muladd", 32, H(a), H(b), H(c)).hex() == r
    for n, r in k["sha512"]:
        m = golden_msg(n)
        assert call(hostcheck, "hc_sha512", 64, m, SZ(n)).hex() == r
    no_violations(hostcheck)


def test_sha512_length_field_beyond_32_bits(hostcheck):
    """lib/sha512.c:196-203 writes the bit length as 128 bits; csrc/sha512.h keeps it in one 64-bit word.  A message of
    2^29 + 5 bytes has a bit length of 2^32 + 40 - the first that needs the upper half of that word - and one lane of the
    device would take a minute over it, so the device SOURCE hashes it here, against hashlib; and the same around the
    sizes the GPU suite runs on the device (tests/test_gpu_messages.py)"""
    for n in (16384 - 17, 65536 + 112, (1 << 20) + 17, (1 << 29) + 5):
        m = (hashlib.sha512(n.to_bytes(8, "little")).digest() * (n // 64 + 1))[:n]
        assert call(hostcheck, "hc_sha512", 64, m, SZ(n)) == hashlib.sha512(m).digest(), n
    no_violations(hostcheck)


def test_all_ones_operands_stay_in_bounds(hostcheck):
    """worst-case limbs: every limb of both operands at its maximum, at the loose-operand limits"""
    ones = b"\xff" * 32
    for ka in (1, 4, 7):
        for kb in (1, 2, 3):
            a = int.from_bytes(ones, "little") % P + 0   # 2^256-1 folds to 2^255-1+19 -> limbs all ones
            want = (ka * ((2**255 - 1) + 19) * kb * ((2**255 - 1) + 19)) % P
            assert call(hostcheck, "hc_fe_mul_loose", 32, ones, ka, ones, kb) == le(want)
    no_violations(hostcheck)


def test_batch_inverse_is_the_inverse(hostcheck):
    rng = np.random.default_rng(8)
    for k in (1, 2, 5, 8):
        zs = [int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "little") for _ in range(k)]
        zs[0] = P - 1
        out = ctypes.create_string_buffer(32 * k)
        hostcheck.hc_batch_inverse(out, b"".join(le(z) for z in zs), k)
        for j, z in enumerate(zs):
            assert int.from_bytes(out.raw[32 * j:32 * j + 32], "little") == pow(z % P, P - 2, P), (k, j)
    no_violations(hostcheck)


def test_random_against_oracle(hostcheck, oracle):
    rng = np.random.default_rng(21)
    rb = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))  # noqa: E731
    for i in range(150):
        sk, msg = rb(32), rb(int(rng.integers(0, 200)))
        pk = oracle.genpub(sk)
        assert call(hostcheck, "hc_genpub", 32, sk) == pk
        sig = oracle.sign(sk, pk, msg)
        assert call(hostcheck, "hc_sign", 64, sk, pk, msg, SZ(len(msg))) == sig
        cases = [(sig, pk), (rb(64), rb(32)), (sig[:32] + le((int.from_bytes(sig[32:], "little") + L) % 2**256), pk),
                 (sig, rb(32)), (sig[:31] + bytes([sig[31] ^ 0x80]) + sig[32:], pk)]
        for s, p_ in cases:
            assert hostcheck.hc_verify(s, p_, msg, SZ(len(msg))) == int(oracle.verify(s, p_, msg))
        a, b = rb(32), rb(32)
        assert call(hostcheck, "hc_x25519", 32, a, b) == oracle.x25519(a, b)
        assert call(hostcheck, "hc_x25519_base", 32, a) == oracle.x25519_base(a)
        assert call(hostcheck, "hc_pk_to_x", 32, a) == oracle.pk_to_x(a)
        assert call(hostcheck, "hc_sk_to_x", 32, a) == oracle.sk_to_x(a)
    no_violations(hostcheck)


def test_exact_chain_for_off_curve_points(hostcheck, oracle, golden):
    """ed_dual_scale replayed in the reference's order (divergent form and the uniform form the GPU
    runs): the encoded result equals the reference's even when the 'point' is not on the curve"""
    k = golden("layer_kats.json")
    rng = np.random.default_rng(31)
    rb = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))  # noqa: E731
    off = 0
    for uniform in (0, 1):
        for s, t, q, r in k["ed_dual_scale"]:
            assert call(hostcheck, "hc_dual_scale_exact", 32, H(s), H(t), H(q), uniform).hex() == r
            assert call(hostcheck, "hc_dual_scale_exact_table", 32, H(s), H(t), H(q), uniform).hex() == r
        for i in range(120):
            s, t, q = rb(32), rb(32), rb(32)
            if i % 9 == 0: s = bytes(32)
            if i % 13 == 0: t = bytes(32)
            want = ctypes.create_string_buffer(32)
            oracle.lib.orc_ed_dual_scale(want, s, t, q)
            assert call(hostcheck, "hc_dual_scale_exact", 32, s, t, q, uniform) == want.raw
            # the table form of the one-lane kernels, in one go and stretch by stretch
            assert call(hostcheck, "hc_dual_scale_exact_table", 32, s, t, q, uniform) == want.raw
            off += call(hostcheck, "hc_ed_import_export", 32, q) is not None and hostcheck.hc_ed_import_export(ctypes.create_string_buffer(32), q) == 0
    assert off > 60                                   # plenty of genuinely off-curve inputs were exercised
    # two items per lane (lanes.h: exact_pair_iterations, what k_verify_exact_lane_chain runs): pairs of the triples above
    # - equal, different, a lone first item - in one go and in units of 75 / 40 / 1 iterations, both results as the oracle's
    trip = [(rb(32) if i % 7 else bytes(32), rb(32) if i % 5 else bytes(32), rb(32)) for i in range(40)] + \
           [(H(s), H(t), H(q)) for s, t, q, _ in k["ed_dual_scale"][:12]]
    def ref(s, t, q):
        want = ctypes.create_string_buffer(32)
        oracle.lib.orc_ed_dual_scale(want, s, t, q)
        return want.raw
    for n, (x, y) in enumerate(zip(trip, trip[1:] + trip[:1])):
        for unit in (0, 75, 40, 1)[: 4 if n < 6 else 2]:
            oa, ob = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
            have_b = n % 4 != 3
            assert hostcheck.hc_dual_scale_exact_pair(oa, ob, b"".join(x), b"".join(y), int(have_b), unit) == 0, (n, unit)
            assert oa.raw == ref(*x) and (not have_b or ob.raw == ref(*y)), (n, unit)
    for c in golden("verify_edges.json"):
        msg = H(c["msg"])
        assert hostcheck.hc_verify_exact(H(c["sig"]), H(c["pub"]), msg, SZ(len(msg))) == int(c["accept"]), c["name"]
        # the one-lane throughput form of the same chain (k_verify_exact_lane_*: packed cached entries in the item's table)
        assert hostcheck.hc_verify_exact_table(H(c["sig"]), H(c["pub"]), msg, SZ(len(msg))) == int(c["accept"]), c["name"]
    # genuine signatures under garbage keys (half are no curve points), R = 0 under garbage keys (the Z = 0 corner), and
    # genuine items: the table form decides every one as the oracle does
    accepted = 0
    for i in range(60):
        sk, msg = rb(32), rb(int(rng.integers(0, 90)))
        pk = oracle.genpub(sk)
        sig = oracle.sign(sk, pk, msg)
        for s_, p_ in ((sig, pk), (sig, rb(32)), (bytes(32) + rb(32), rb(32)), (rb(64), rb(32))):
            want = int(oracle.verify(s_, p_, msg))
            accepted += want
            assert hostcheck.hc_verify_exact_table(s_, p_, msg, SZ(len(msg))) == want
    assert accepted >= 60
    no_violations(hostcheck)


def test_tables_match_the_reference_points(hostcheck, oracle, golden):
    """comb[i][k] = (k+1) * 4096^i * B (row 0 = the first row of the reference's lib/ed_lookup64.h)
    and base16[k] = k * B, every comb entry and a sample of base16 against the oracle's k * B"""
    base16 = np.zeros((32769, 32), np.uint32); comb = np.zeros((704, 32), np.uint32)
    hostcheck.hc_tables(base16.ctypes.data_as(ctypes.c_void_p), comb.ctypes.data_as(ctypes.c_void_p))
    pos = [0, 26, 51, 77, 102, 128, 153, 179, 204, 230]
    val = lambda limbs: sum(int(v) << s for v, s in zip(limbs, pos))  # noqa: E731
    inv2 = pow(2, P - 2, P)
    ell = 2**252 + 27742317777372353535851937790883648493

    def enc(entry):
        ymx, ypx = val(entry[0:10]), val(entry[10:20])
        y, x = (ypx + ymx) * inv2 % P, (ypx - ymx) * inv2 % P
        return (y | (x & 1) << 255).to_bytes(32, "little")

    pts = golden("comb_points.bin")
    for k in range(8):
        assert enc(comb[k]) == pts[32 * k:32 * k + 32], k
    out = ctypes.create_string_buffer(32)
    for e in range(704):
        oracle.lib.orc_ed_scale_base(out, int((e % 32 + 1) * 4096 ** (e // 32) % ell).to_bytes(32, "little"))
        assert enc(comb[e]) == out.raw, e
    for k in list(range(1, 32769, 331)) + [32767, 32768]:
        oracle.lib.orc_ed_scale_base(out, int(k).to_bytes(32, "little"))
        assert enc(base16[k]) == out.raw, k
    no_violations(hostcheck)


def test_the_assertions_are_live(hostcheck):
    """an operand beyond the documented limit must trip the check (then reset the counter)"""
    hostcheck.hc_reset()
    call(hostcheck, "hc_fe_mul_loose", 32, b"\xff" * 32, 7, b"\xff" * 32, 4)      # g = 4u > 3.36u
    assert hostcheck.hc_violations() > 0 and b"fe25519.h" in hostcheck.hc_first_violation()
    hostcheck.hc_reset()


def test_fuzz_field_and_scalars_with_hypothesis(hostcheck, oracle):
    """property-based: products/squares/inverses of arbitrary 256-bit strings and reductions of
    arbitrary 32/64-byte strings mod l agree with big-integer arithmetic, with no bound violation"""
    from hypothesis import given, settings, strategies as st

    b32 = st.binary(min_size=32, max_size=32)

    @settings(max_examples=300, deadline=None)
    @given(b32, b32, st.integers(1, 7), st.integers(1, 3))
    def mul(a, b, ka, kb):
        x = int.from_bytes(a, "little") % 2**255 + 19 * (a[31] >> 7)
        y = int.from_bytes(b, "little") % 2**255 + 19 * (b[31] >> 7)
        assert call(hostcheck, "hc_fe_mul", 32, a, b) == le(x * y % P)
        assert call(hostcheck, "hc_fe_mul_loose", 32, a, ka, b, kb) == le(ka * x * kb * y % P)
        assert call(hostcheck, "hc_fe_sq", 32, a) == le(x * x % P)
        assert call(hostcheck, "hc_fe_inv", 32, a) == le(pow(x % P, P - 2, P))

    @settings(max_examples=300, deadline=None)
    @given(st.binary(min_size=64, max_size=64), b32, b32, b32)
    def scalars(w, a, b, c):
        assert call(hostcheck, "hc_sc_reduce", 32, w, SZ(64)) == le(int.from_bytes(w, "little") % L)
        assert call(hostcheck, "hc_sc_reduce", 32, a, SZ(32)) == le(int.from_bytes(a, "little") % L)
        r = (int.from_bytes(a, "little") * int.from_bytes(b, "little") + int.from_bytes(c, "little")) % L
        assert call(hostcheck, "hc_sc_muladd", 32, a, b, c) == le(r)

    mul()
    scalars()
    no_violations(hostcheck)


# ---------------------------------------------------------------------------------------------------------
# batch verification (libeddsa_amd/csrc/rlc_lanes.h): the per-lane steps of rlc.hip on the host, bounds asserted
# ---------------------------------------------------------------------------------------------------------
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SMALL_ORDER_Y = {1, P - 1, 0,
                 int.from_bytes(bytes.fromhex("26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05"), "little"),
                 int.from_bytes(bytes.fromhex("c7176a703d4dd84fba3c0b760d10670f2a2053fa2c39ccc64ec7fd7792ac037a"), "little")}


def _decode(enc):
    """(on_curve, x, y) of ed_import's result (lib/ed.c:100-149), big-integer arithmetic"""
    v = int.from_bytes(enc, "little")
    sign, y = v >> 255, (v & (2**255 - 1)) % P
    u, w = (y * y - 1) % P, (D * y * y + 1) % P
    x2 = u * pow(w, P - 2, P) % P
    x = pow(x2, (P + 3) // 8, P)
    if (x * x - x2) % P:
        x = x * pow(2, (P - 1) // 4, P) % P
    on = (x * x - x2) % P == 0
    if on and (x & 1) != sign:
        x = (P - x) % P
    return on, x, y


def test_rlc_decoding_and_routing_flags(hostcheck, oracle):
    import ctypes
    rng = np.random.default_rng(21)
    le = lambda x: int(x).to_bytes(32, "little")  # noqa: E731
    encs = [le(1), le(P - 1), le(0), le(1 << 255), le(1 | 1 << 255), le(P), le(P + 1), le(P + 3), le((P - 1) | 1 << 255),
            bytes.fromhex("26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05"),
            bytes.fromhex("c7176a703d4dd84fba3c0b760d10670f2a2053fa2c39ccc64ec7fd7792ac03fa")]
    encs += [le(y) for y in range(2, 30)] + [le(2**255 - 1), le(2**256 - 1)]
    encs += [oracle.genpub(bytes(rng.integers(0, 256, 32, dtype=np.uint8))) for _ in range(40)]
    encs += [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(40)]
    out = ctypes.create_string_buffer(96)
    seen = set()
    for enc in encs:
        v = int.from_bytes(enc, "little")
        on, x, y = _decode(enc)
        small = on and y in SMALL_ORDER_Y
        # a key: flagged for the per-item path when it is no curve point or has small order
        fl = hostcheck.hc_rlc_decode(out, enc, 0)
        assert fl == (2 if (not on or small) else 0), enc.hex()
        if on:                                          # the entry is the NEGATED point: (y + x, y - x, -2dxy)
            nx = (P - x) % P
            assert out.raw == le((y - nx) % P) + le((y + nx) % P) + le(2 * D * nx * y % P), enc.hex()
        # an R: only the canonical encoding of a curve point survives
        valid = on and (v & (2**255 - 1)) < P and not (x == 0 and v >> 255)
        fl = hostcheck.hc_rlc_decode(out, enc, 1)
        assert fl == ((1 if valid else 0) | (2 if valid and small else 0)), enc.hex()
        seen.add((on, small, valid))
    assert {(True, False, True), (True, True, True), (False, False, False), (True, True, False), (True, False, False)} <= seen
    assert hostcheck.hc_violations() == 0, hostcheck.hc_first_violation()


def test_rlc_coefficients_and_digits(hostcheck):
    import ctypes
    import hashlib
    rng = np.random.default_rng(22)
    da, dr = (ctypes.c_int8 * 32)(), (ctypes.c_int8 * 16)()
    zs = ctypes.create_string_buffer(32)
    for k in range(60):
        seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        i = int(rng.integers(0, 2**40)) if k % 2 else k
        t = int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "little") % L if k > 3 else (0, 1, L - 1, L - 2)[k]
        s = int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "little") % L if k > 3 else (L - 1, 0, 1, 7)[k]
        hostcheck.hc_rlc_scalars(da, dr, zs, seed, ctypes.c_uint64(i), t.to_bytes(32, "little"), s.to_bytes(32, "little"))
        h = hashlib.sha512(seed + i.to_bytes(8, "little") + b"rlc\0" + bytes(20)).digest()
        z = (int.from_bytes(h[:16], "little") & (2**126 - 1)) | 1
        assert all(-128 <= d <= 127 for d in list(da) + list(dr))
        assert sum(d << (8 * j) for j, d in enumerate(dr)) == z
        # the key's scalar: z t mod 8 l, centred (a torsion component of A must see z t itself, not z t mod l)
        av = sum(d << (8 * j) for j, d in enumerate(da))
        assert (av - z * t) % (8 * L) == 0 and abs(av) <= 4 * L and (av < 0) == ((z * t // L) % 8 >= 4)
        assert int.from_bytes(zs.raw, "little") == z * s % L
    assert hostcheck.hc_violations() == 0, hostcheck.hc_first_violation()


def _mixed_order_signatures(oracle, count, tors_index, seed):
    """signatures whose prime-order part is honest under keys A' = a B + T, T = small_order_points()[tors_index]: the
    reference's sign takes the public key as an argument without checking it (lib/ed25519-sha512.c:84-123), so
    sign(sk, A', M) is R = r B, S = r + H(R | A' | M) a, and S B - t A' - R = -t T: accepted iff ord(T) divides t"""
    from gen_golden import ed_add_affine, ed_dec, ed_enc, small_order_points
    rng = np.random.default_rng(seed)
    T = ed_dec(small_order_points()[tors_index])
    out = []
    for _ in range(count):
        sk = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        msg = bytes(rng.integers(0, 256, 24, dtype=np.uint8))
        a1 = ed_enc(ed_add_affine(ed_dec(oracle.genpub(sk)), T))
        out.append((oracle.sign(sk, a1, msg), a1, msg))
    return out


def _torsion_order(index):
    from gen_golden import ed_add_affine, ed_dec, small_order_points
    T = ed_dec(small_order_points()[index])
    acc, k = T, 1
    while acc != (0, 1):
        acc, k = ed_add_affine(acc, T), k + 1
    return k


def test_rlc_single_item_groups_are_exact_on_mixed_order_points(hostcheck, oracle, golden):
    """A ONE-item combination z (S B - t A - R) with z odd, z < l decides exactly what the reference's per-item check
    decides, whatever small-order components A and R carry - provided the key's scalar is z t mod 8 l and not
    z t mod l (with the latter 51 of 400 such signatures passed that the reference rejects).  Checked
    on the 192 reference-pinned vectors of verify_torsion.json and on 400 fresh signatures under A = a B + T8."""
    import ctypes
    valid = (ctypes.c_uint8 * 1)()
    for c in golden("verify_torsion.json"):
        msg = H(c["msg"])
        got = hostcheck.hc_rlc_group(valid, H(c["sig"]), H(c["pub"]), msg, SZ(len(msg)), 1)
        assert got == int(c["accept"]) and valid[0] == 1, c["name"]
    idx8 = [i for i in range(8) if _torsion_order(i) == 8][0]
    accepted = 0
    for sig, pub, msg in _mixed_order_signatures(oracle, 400, idx8, 31):
        want = int(oracle.verify(sig, pub, msg))
        t = int.from_bytes(hashlib.sha512(sig[:32] + pub + msg).digest(), "little") % L
        assert want == (t % 8 == 0)
        assert hostcheck.hc_rlc_group(valid, sig, pub, msg, SZ(len(msg)), 1) == want
        accepted += want
    assert 25 <= accepted <= 80                                    # about one in eight
    assert hostcheck.hc_violations() == 0, hostcheck.hc_first_violation()


def test_rlc_documented_caveat_two_crafted_items_can_cancel(hostcheck, oracle):
    """What the opt-in mode does NOT promise (include/eddsa_amd.h): two or more crafted items in ONE group.  Two items
    whose defects -t T are the same point of order 2 always cancel (the coefficients are odd), so the group passes
    although the reference rejects both; this pins the documented statement, it is not a goal."""
    import ctypes
    idx2 = [i for i in range(8) if _torsion_order(i) == 2][0]
    items = [x for x in _mixed_order_signatures(oracle, 40, idx2, 32) if not oracle.verify(*x)][:2]
    assert len(items) == 2                                         # t odd: the reference rejects each
    sig = b"".join(x[0] for x in items); pub = b"".join(x[1] for x in items); msg = b"".join(x[2] for x in items)
    valid = (ctypes.c_uint8 * 2)()
    assert hostcheck.hc_rlc_group(valid, sig, pub, msg, SZ(24), 2) == 1
    for k in range(2):                                             # alone, each is rejected
        assert hostcheck.hc_rlc_group(valid, items[k][0], items[k][1], items[k][2], SZ(24), 1) == 0


def test_rlc_whole_group_by_double_and_add(hostcheck, oracle):
    """the combination of rlc.hip for one small group, every step from rlc_lanes.h, evaluated without buckets:
    neutral exactly when every item the combination represents is valid"""
    import ctypes
    rng = np.random.default_rng(23)
    n, mlen = 24, 20
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msg = rng.integers(0, 256, (n, mlen), dtype=np.uint8)
    pk = oracle.genpub_batch(sk)
    sig = oracle.sign_batch(sk, pk, msg, mlen)
    valid = (ctypes.c_uint8 * n)()
    P_ = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    run = lambda s, p, m: hostcheck.hc_rlc_group(valid, P_(np.ascontiguousarray(s)), P_(np.ascontiguousarray(p)),  # noqa: E731
                                                 P_(np.ascontiguousarray(m)), ctypes.c_size_t(mlen), n)
    assert run(sig, pk, msg) == 1 and all(valid)
    s2 = sig.copy()                                             # S + l: reduced, not range-checked
    s2[3, 32:] = np.frombuffer((int.from_bytes(sig[3, 32:].tobytes(), "little") + L).to_bytes(32, "little"), np.uint8)
    assert run(s2, pk, msg) == 1
    s2 = sig.copy(); s2[5, 40] ^= 1
    assert run(s2, pk, msg) == 0                                # a wrong S
    m2 = msg.copy(); m2[7, 0] ^= 1
    assert run(sig, pk, m2) == 0                                # a wrong message
    s2 = sig.copy(); s2[9, :32] = np.frombuffer((P + 1).to_bytes(32, "little"), np.uint8)
    assert run(s2, pk, msg) == 1 and valid[9] == 0 and sum(valid) == n - 1    # non-canonical R: rejected at once, excluded
    p2 = pk.copy(); p2[11] = np.frombuffer((P - 1).to_bytes(32, "little"), np.uint8)
    assert run(sig, p2, msg) == 0                               # a key of order 2: flagged for the per-item path
    p2 = pk.copy(); p2[11] = np.frombuffer((2).to_bytes(32, "little"), np.uint8)
    assert run(sig, p2, msg) == 0                               # a key that is no curve point: flagged
    assert hostcheck.hc_violations() == 0, hostcheck.hc_first_violation()


# ---------------------------------------------------------------------------------------------
# half-length verification (libeddsa_amd/csrc/halve.h)
# ---------------------------------------------------------------------------------------------
N8L = 8 * L


def halve_model(t, th=134, retry_min=122):
    """the rule of halve_scalar_lane on Python integers: (found, u, v)"""
    r0, u0, r1, u1, tried = N8L, 0, t, 1, False
    while True:
        if r1 < (1 << th):
            if u1 & 1:
                return abs(u1) < (1 << th), u1, r1
            if tried or r1 < (1 << retry_min):
                return False, u1, r1
            tried = True
        q = r0 // r1
        if q >= 1 << 31:
            return False, u1, r1
        r0, r1, u0, u1 = r1, r0 - q * r1, u1, u0 - q * u1


def test_halved_scalar_pairs(hostcheck):
    """v = u t (mod 8 l), u odd, both below 2^134, and the same pair as Euclid's algorithm on integers gives -
    for random t, tiny and huge t, and t whose continued fraction has a giant quotient right at the half-way
    point (those are the ones handed to the exact path)"""
    rng = np.random.default_rng(77)
    ts = [0, 1, 2, 3, 5, 8, L - 1, L - 2, L // 2, L // 3, (1 << 134) - 1, 1 << 134, (1 << 134) + 1, (1 << 127) + 12345]
    ts += [N8L // k % L for k in (3, 5, 7, 9, 1000003, (1 << 61) - 1, (1 << 100) + 277, (1 << 125) + 1, (1 << 126) + 3, (1 << 128) + 51)]
    ts += [((1 << 127) * k + 1) % L for k in (2, 6, 10)]
    ts += [int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "little") % L for _ in range(3000)]
    found = 0
    # the wide form (passes below 2^18 items): bound 2^138, retry from 2^118
    for t in ts[:600]:
        v, u, neg = ctypes.create_string_buffer(20), ctypes.create_string_buffer(20), ctypes.c_int(0)
        good = hostcheck.hc_halve(v, u, ctypes.byref(neg), le(t), 1)
        want, mu, mv = halve_model(t, 138, 118)
        assert bool(good) == want, hex(t)
        if good:
            ui = int.from_bytes(u.raw, "little") * (-1 if neg.value else 1)
            assert (ui, int.from_bytes(v.raw, "little")) == (mu, mv) and ui & 1 and abs(ui) < 1 << 138 and (ui * t - mv) % N8L == 0
    counters = (ctypes.c_long * 2)()
    hostcheck.hc_halve_counters(counters, 1)
    for t in ts:
        v, u, neg = ctypes.create_string_buffer(20), ctypes.create_string_buffer(20), ctypes.c_int(0)
        good = hostcheck.hc_halve(v, u, ctypes.byref(neg), le(t), 0)
        want, mu, mv = halve_model(t)
        assert bool(good) == want, hex(t)
        if good:
            found += 1
            ui = int.from_bytes(u.raw, "little") * (-1 if neg.value else 1)
            vi = int.from_bytes(v.raw, "little")
            assert (ui, vi) == (mu, mv), hex(t)
            assert ui & 1 and abs(ui) < 1 << 134 and 0 <= vi < 1 << 134 and (ui * t - vi) % N8L == 0
    assert found >= len(ts) - 12
    # most of the way is covered by Lehmer rounds (one exact long update per ~23 bits), the plain loop does the rest
    hostcheck.hc_halve_counters(counters, 1)
    assert counters[0] >= 4 * 3000 and counters[1] <= 12 * len(ts), list(counters)
    no_violations(hostcheck)


def test_verify_half_length_edges_and_torsion(hostcheck, golden):
    """the half-length path on the reference-pinned edge cases and on the mixed-order keys / commitments of
    verify_torsion.json (where u t A != (u t mod l) A, and a plain mod-l split would accept or reject wrongly):
    same verdict as the reference whenever the path keeps the item (it hands off-curve keys to the exact path)"""
    kept = 0
    for c in golden("verify_edges.json") + golden("verify_torsion.json"):
        msg = H(c["msg"])
        got = hostcheck.hc_verify_half(H(c["sig"]), H(c["pub"]), msg, SZ(len(msg)))
        if got == 2:
            assert hostcheck.hc_verify_exact(H(c["sig"]), H(c["pub"]), msg, SZ(len(msg))) == int(c["accept"]), c["name"]
        else:
            kept += 1
            assert got & 1 == int(c["accept"]), c["name"]
            # the same item as a short lane of a wave that runs the long loop (neutral additions from window 34 on)
            assert hostcheck.hc_verify_half_in_long_wave(H(c["sig"]), H(c["pub"]), msg, SZ(len(msg))) == int(c["accept"]), c["name"]
            # and in the wide form of mid-size passes (35 windows)
            assert hostcheck.hc_verify_half_wide(H(c["sig"]), H(c["pub"]), msg, SZ(len(msg))) in (2, int(c["accept"])), c["name"]
    assert kept > 350
    no_violations(hostcheck)


def test_verify_half_length_random_against_oracle(hostcheck, oracle):
    rng = np.random.default_rng(23)
    rb = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))  # noqa: E731
    for i in range(120):
        sk, msg = rb(32), rb(int(rng.integers(0, 150)))
        pk = oracle.genpub(sk)
        sig = oracle.sign(sk, pk, msg)
        flip = bytearray(sig); flip[int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        cases = [(sig, pk), (rb(64), pk), (bytes(flip), pk), (sig[:32] + le((int.from_bytes(sig[32:], "little") + L) % 2**256), pk),
                 (sig, rb(32)), (sig[:31] + bytes([sig[31] ^ 0x80]) + sig[32:], pk), (le(P + 1) + sig[32:], pk)]
        for s, p_ in cases:
            got = hostcheck.hc_verify_half(s, p_, msg, SZ(len(msg)))
            want = int(oracle.verify(s, p_, msg))
            assert (got != 2 and got & 1 == want) or (got == 2 and hostcheck.hc_verify_exact(s, p_, msg, SZ(len(msg))) == want)
    no_violations(hostcheck)


def test_verify_half_length_rejects_what_export_never_writes(hostcheck, oracle):
    """x = 0 with the sign bit, y >= p: R strings only a permissive decoder takes (cf. tests/test_gpu_half.py)"""
    rs = [le(1), le(1 | 1 << 255), le(P + 1), le(P - 1), le((P - 1) | 1 << 255), le(P), le(2**255 - 1)]
    for r in rs:
        for a in (le(1), le(P - 1), le(1 | 1 << 255)):
            for m in (b"a", b"bb", b"ccc", b"dddd"):
                got = hostcheck.hc_verify_half(r + le(0), a, m, SZ(len(m)))
                assert got & 3 == int(oracle.verify(r + le(0), a, m)), (r.hex(), a.hex(), m)
    no_violations(hostcheck)


def test_items_without_a_short_pair_run_the_long_loop(hostcheck, oracle):
    """about 1 hash in 10^4 has no pair (u, v) the rule of halve.h accepts: such an item keeps (u, v) = (1, t) and its
    wave runs 64 windows.  Search signatures of one key for such t (SHA-512 of R | A | M mod l against the model),
    then: the genuine signature is accepted, a corrupted one rejected, both through the long loop (return value + 4)"""
    sk = bytes(range(32))
    pk = oracle.genpub(sk)
    hits = []
    for i in range(200000):
        msg = b"long-loop search %d" % i
        sig = oracle.sign(sk, pk, msg)
        t = int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % L
        if not halve_model(t)[0]:
            hits.append((sig, msg))
            if len(hits) == 3:
                break
    assert len(hits) == 3
    for sig, msg in hits:
        assert hostcheck.hc_verify_half(sig, pk, msg, SZ(len(msg))) == 5
        bad = sig[:40] + bytes([sig[40] ^ 1]) + sig[41:]
        assert hostcheck.hc_verify_half(bad, pk, msg, SZ(len(msg))) == 4
        bad = bytes([sig[0] ^ 2]) + sig[1:]
        assert hostcheck.hc_verify_half(bad, pk, msg, SZ(len(msg))) & 1 == 0
    no_violations(hostcheck)


def test_a_wrong_quotient_in_the_pair_search_is_caught_by_the_exact_check(hostcheck, oracle):
    """the pair search takes its quotients from doubles; verify_half_scalars_lane re-verifies
    u t = v (mod 8 l) with integers and drops a pair that fails (the item then keeps (u, v) = (1, t)).  Two injected
    faults (halve.h: HALVE_FAULT): a quotient one too large at the n-th half-step of the plain loop - the failure the
    margins exclude; the remainder wraps and the search usually ends in one of its own give-ups - and the same
    quotient reaching only the cofactor update, which leaves a plausible pair with a broken congruence.  Required:
    the verdict is the reference's every time, and every broken pair that the search returns is refused (+ 8)."""
    sk = bytes(range(1, 33))
    pk = oracle.genpub(sk)
    refused = {1: 0, -1: 0}
    for i in range(40):
        msg = b"fault injection %d" % i
        sig = oracle.sign(sk, pk, msg)
        bad = sig[:45] + bytes([sig[45] ^ 4]) + sig[46:]
        for s_, want in ((sig, 1), (bad, 0)):
            for sign in (1, -1):
                hostcheck.hc_halve_fault(sign * (1 + i % 7))
                got = hostcheck.hc_verify_half(s_, pk, msg, SZ(len(msg)))
                hostcheck.hc_halve_fault(0)
                assert got & 1 == want, (i, sign, got)
                if got & 8:
                    assert got & 4                          # a refused pair means the long loop
                    refused[sign] += 1
    assert refused[-1] >= 70, refused                       # a wrong cofactor always leaves a pair to refuse
    hostcheck.hc_reset()                                    # the injected faults tripped the no-borrow assertions, as they must
    # and without injection nothing is ever refused
    for i in range(200):
        msg = b"no fault %d" % i
        sig = oracle.sign(sk, pk, msg)
        assert hostcheck.hc_verify_half(sig, pk, msg, SZ(len(msg))) in (1, 5)
    no_violations(hostcheck)
