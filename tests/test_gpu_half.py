"""GPU tests of the half-length verification (libeddsa_amd/csrc/halve.h: u (S B - t A - R) = 0 with half-length
u, v = u t mod 8l; k_verify_halve + k_verify_main_half), the default route of verify passes above 2^14 items: the
same verdict bytes as the reference, as the full-length kernels and as the oracle, on the reference-pinned edge
cases, on mixed-order keys and commitments, and on seeded batches large enough to contain items the pair search
hands to the exact path (about 1 in 10^4)."""
import numpy as np
import pytest

import workload  # tools/ is on sys.path (conftest)

pytestmark = pytest.mark.gpu
H = bytes.fromhex


def arr(rows):
    return np.frombuffer(b"".join(rows), np.uint8).reshape(len(rows), -1).copy()


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(autouse=True)
def restore(engine):
    yield
    engine.set_verify_algo(0)
    engine.set_offcurve_mode(True)


@pytest.mark.parametrize("algo", [0, 1, 2, 3])
def test_edge_and_torsion_vectors_on_either_evaluation(engine, golden, algo):
    """verify_edges.json (S + k l, non-canonical / small-order / off-curve A, non-canonical R, flipped bits) and
    verify_torsion.json (A = a B + T, R = r B + T' for all 64 pairs of points of order dividing 8: accepted
    exactly when t T + T' = 0, which a pair (u, v) with v = u t mod l only - not mod 8l - would get wrong)"""
    engine.set_verify_algo(algo)
    cases = golden("verify_edges.json") + golden("verify_torsion.json")
    msgs = [H(c["msg"]) for c in cases]
    off = np.zeros(len(msgs) + 1, np.uint64)
    off[1:] = np.cumsum([len(m) for m in msgs])
    blob = np.frombuffer(b"".join(msgs), np.uint8).copy()
    sig, pub = arr([H(c["sig"]) for c in cases]), arr([H(c["pub"]) for c in cases])
    want = np.array([c["accept"] for c in cases], np.uint8)
    got = engine.ed25519_verify_batch(sig, pub, blob, msg_off=off)
    bad = [cases[i]["name"] for i in np.nonzero(got != want)[0]]
    assert not bad, bad
    got = engine.ed25519_verify_batch(dev(sig), dev(pub), dev(blob), msg_off=dev(off.astype(np.int64))).cpu().numpy()
    assert np.array_equal(got, want)
    tors = np.array([c["accept"] for c in golden("verify_torsion.json")])
    assert 20 <= tors.sum() <= 40 and len(tors) == 192


def test_half_full_and_oracle_agree_on_a_large_seeded_batch(engine, oracle):
    """2^17 + 333 items of the config-2 recipe (1/16 corrupted, edge vectors spliced in, some off-curve keys; about
    a dozen items whose t has no short pair): default route (half-length), forced full-length, forced half-length
    and the oracle return the same bytes"""
    n = (1 << 17) + 333
    sk, msg = workload.sign_inputs(n, seed=9, config=2)
    pk = engine.ed25519_genpub_batch(sk)
    sig = engine.ed25519_sign_batch(sk, pk, msg)
    expect = workload.corrupt_for_verify(sig, pk, msg, seed=9, config=2)
    want = oracle.verify_batch(sig, pk, msg, msg.shape[1])
    assert np.array_equal(want, expect)
    d = dev(sig), dev(pk), dev(msg)
    for algo in (0, 2, 3, 1):
        engine.set_verify_algo(algo)
        got = engine.ed25519_verify_batch(*d, msg_len=msg.shape[1]).cpu().numpy()
        assert np.array_equal(got, want), (algo, np.nonzero(got != want)[0][:10])
    engine.set_verify_algo(0)
    assert np.array_equal(engine.ed25519_verify_batch(sig, pk, msg, msg_len=msg.shape[1]), want)      # host path, chunked


@pytest.mark.parametrize("n", [1, 2, 255, 257, 4099])
def test_forced_half_length_on_small_and_ragged_passes(engine, oracle, n):
    rng = np.random.default_rng(4000 + n)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    lens = rng.integers(0, 150, n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    msgs = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)
    pk = oracle.genpub_batch(sk)
    sig = engine.ed25519_sign_batch(dev(sk), dev(pk), dev(msgs), msg_off=dev(off.astype(np.int64))).cpu().numpy()
    kind = rng.integers(0, 6, n)
    for i in np.nonzero(kind == 1)[0]: sig[i, rng.integers(0, 64)] ^= 1 << rng.integers(0, 8)
    for i in np.nonzero(kind == 2)[0]: pk[i, rng.integers(0, 32)] ^= 1 << rng.integers(0, 8)
    g = np.nonzero(kind == 3)[0]
    pk[g] = rng.integers(0, 256, (len(g), 32), dtype=np.uint8)
    want = np.array([oracle.verify(sig[i].tobytes(), pk[i].tobytes(), msgs[int(off[i]):int(off[i + 1])].tobytes())
                     for i in range(n)], np.uint8)
    for algo in (0, 2, 3, 1):
        engine.set_verify_algo(algo)
        assert np.array_equal(engine.ed25519_verify_batch(sig, pk, msgs, msg_off=off), want), algo


def test_reject_mode_keeps_every_on_curve_item(engine, oracle):
    """set_offcurve_mode(False) has no exact path, so it must not use the half-length route (whose pair search gives
    up on a few items): on a batch without off-curve keys it still returns the reference's verdicts"""
    n = 1 << 16
    sk, msg = workload.sign_inputs(n, seed=12, config=2)
    pk = engine.ed25519_genpub_batch(sk)
    sig = engine.ed25519_sign_batch(sk, pk, msg)
    sig[5::16, 3] ^= 4
    want = oracle.verify_batch(sig, pk, msg, msg.shape[1])
    engine.set_offcurve_mode(False)
    engine.set_verify_algo(2)
    assert np.array_equal(engine.ed25519_verify_batch(sig, pk, msg, msg_len=msg.shape[1]), want)


def test_commitments_that_only_decode_permissively(engine, oracle):
    """R strings that ed_import would accept but ed_export never produces - x = 0 with the sign bit set, y >= p -
    next to their canonical spellings, under keys and S for which the canonical one is accepted: the byte comparison
    of the reference rejects the former, and so must the point comparison of the half-length route"""
    P = 2**255 - 19
    le = lambda x: int(x).to_bytes(32, "little")   # noqa: E731
    ident, minus1 = le(1), le(P - 1)
    rs = [ident, le(1 | 1 << 255), le(P + 1), le((P + 1) | 1 << 255), minus1, le((P - 1) | 1 << 255), le(P), le(2**255 - 1)]
    keys = [ident, minus1, le(1 | 1 << 255)]
    sigs, pubs, msgs = [], [], []
    for r in rs:
        for a in keys:
            for m in range(4):
                sigs.append(r + le(0)); pubs.append(a); msgs.append(b"permissive R %d" % m + bytes(19))
    sig, pub, msg = arr(sigs), arr(pubs), arr(msgs)
    want = np.array([oracle.verify(s, p, m) for s, p, m in zip(sigs, pubs, msgs)], np.uint8)
    assert want[:12].sum() >= 8 and want[12:24].sum() == 0          # canonical identity accepted, its signed spelling never
    for algo in (0, 2, 3, 1):                         # 0: lane pairs and quads at this size
        engine.set_verify_algo(algo)
        assert np.array_equal(engine.ed25519_verify_batch(sig, pub, msg, msg_len=msg.shape[1]), want), algo


def test_device_pair_search_against_integers(engine):
    """the pair search as the DEVICE runs it (reciprocal by Newton steps instead of the host build's division) against
    Euclid on Python integers: random t, tiny t, t = (8l)/k with giant quotients at the start, in the middle and
    right at the threshold, quotients just below and above the 2^31 limit"""
    L = 2**252 + 27742317777372353535851937790883648493
    N = 8 * L

    def model(t, th=134, retry_min=122):
        """(found, u, v, largest quotient met) by the rule of halve.h without its 31-bit quotient limit"""
        r0, u0, r1, u1, tried, qmax = N, 0, t, 1, False, 0
        while True:
            if r1 < (1 << th):
                if u1 & 1:
                    return abs(u1) < (1 << th), u1, r1, qmax
                if tried or r1 < (1 << retry_min):
                    return False, u1, r1, qmax
                tried = True
            q = r0 // r1
            qmax = max(qmax, q)
            r0, r1, u0, u1 = r1, r0 - q * r1, u1, u0 - q * u1

    rng = np.random.default_rng(99)
    ts = [0, 1, 2, 3, 5, L - 1, L - 2, L // 2, L // 3, (1 << 134) - 1, 1 << 134, (1 << 134) + 1]
    ts += [N // k % L for k in (3, 5, 7, 9, 1000003, (1 << 31) - 2, (1 << 31) - 1, 1 << 31, (1 << 31) + 1, (1 << 61) - 1,
                                (1 << 100) + 277, (1 << 120) + 1, (1 << 121) + 7, (1 << 125) + 1, (1 << 126) + 3, (1 << 128) + 51)]
    # continued fractions with a chosen large quotient q after a random prefix: t = N * (convergent-like ratio)
    for q in (1 << 10, 1 << 20, (1 << 24) - 1, 1 << 24, (1 << 24) + 1, 1 << 30, (1 << 31) - 3):
        for depth in (1, 20, 40, 60, 70, 75):
            a, b = 1, 0
            for k in range(depth):
                a, b = int(rng.integers(1, 4)) * a + b, a
            a, b = q * a + b, a
            ts.append(N * b // a % L)
    ts += [int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "little") % L for _ in range(20000)]
    # the wide form of passes below 2^18 items
    for t, (ok, u, v) in zip(ts[:3000], engine.debug_halve(ts[:3000], wide=True)):
        want, mu, mv, qmax = model(t, 138, 118)
        if qmax < (1 << 31) - (1 << 28):
            assert ok == want, hex(t)
        if ok:
            assert want and (u, v) == (mu, mv) and u & 1 and abs(u) < 1 << 138 and (u * t - v) % N == 0, hex(t)
    got = engine.debug_halve(ts)
    found = 0
    for t, (ok, u, v) in zip(ts, got):
        want, mu, mv, qmax = model(t)
        # a quotient of 31 bits or more makes the search give up, unless the Lehmer rounds have already taken
        # enough of it away in partial steps: either outcome is fine near the limit, a found pair must be Euclid's
        if qmax < (1 << 31) - (1 << 28):
            assert ok == want, hex(t)
        elif qmax >= (1 << 32):
            assert not ok, hex(t)
        if ok:
            found += 1
            assert want and (u, v) == (mu, mv) and u & 1 and (u * t - v) % N == 0, hex(t)
    assert found > len(ts) - 40


def test_small_passes_with_items_that_have_no_short_pair(engine, oracle):
    """a pass of 2^15 signatures (the four-lane route) contains two or three items whose t has no short pair: their
    waves run 64 windows and the other items of those waves add neutral elements; same bytes as the oracle, also
    with corrupted items next to them"""
    import hashlib
    L = 2**252 + 27742317777372353535851937790883648493
    N = 8 * L

    def has_pair(t):
        r0, u0, r1, u1, tried = N, 0, t, 1, False
        while True:
            if r1 < (1 << 134):
                if u1 & 1:
                    return abs(u1) < (1 << 134)
                if tried or r1 < (1 << 122):
                    return False
                tried = True
            q = r0 // r1
            r0, r1, u0, u1 = r1, r0 - q * r1, u1, u0 - q * u1

    n = 1 << 15
    sk, msg = workload.sign_inputs(n, seed=21, config=2)
    pk = engine.ed25519_genpub_batch(sk)
    sig = engine.ed25519_sign_batch(sk, pk, msg)
    long_items = [i for i in range(n) if not has_pair(
        int.from_bytes(hashlib.sha512(sig[i, :32].tobytes() + pk[i].tobytes() + msg[i].tobytes()).digest(), "little") % L)]
    assert 1 <= len(long_items) <= 12, long_items
    want = np.ones(n, np.uint8)
    assert np.array_equal(engine.ed25519_verify_batch(sig, pk, msg, msg_len=32), want)
    for i in long_items[:2]:                       # a neighbour in the same wave, and the long item itself
        sig[i ^ 1, 40] ^= 1; want[i ^ 1] = 0
    sig[long_items[-1], 33] ^= 4; want[long_items[-1]] = 0
    got = engine.ed25519_verify_batch(sig, pk, msg, msg_len=32)
    assert np.array_equal(got, want) and np.array_equal(oracle.verify_batch(sig, pk, msg, 32), want)
    # the same items in the smallest passes (the windows' sums are added up first, 64 of them for an item without a
    # pair): a few hundred items around each, the item with one neighbour, the item alone
    for i in long_items[:3]:
        for lo, hi in ((max(0, i - 150), min(n, i + 200)), (i & ~1, (i & ~1) + 2), (i, i + 1)):
            got = engine.ed25519_verify_batch(sig[lo:hi].copy(), pk[lo:hi].copy(), msg[lo:hi].copy(), msg_len=32)
            assert np.array_equal(got, want[lo:hi]), (i, lo, hi)
