"""GPU tests of the opt-in batch verification (the reference's TODO lib/ed25519-sha512.c:13-14):
whatever route an item takes - accepted with its group by the random linear combination, rejected at once for
a non-canonical R, or decided by the per-item kernels after its group failed or was flagged - the verdict
bytes must be those of the reference's per-item loop."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = 8192
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(autouse=True)
def always_combine(engine):
    """these tests drive the combination itself with small batches: switch off the routing of calls below 5 x 2^15
    items to the per-item kernels (test_small_calls_go_to_the_per_item_kernels checks the default)"""
    engine.set_rlc_min_items(0)
    yield
    engine.set_rlc_min_items(engine.RLC_MIN_ITEMS_DEFAULT)


def _signed(engine, n, seed, mlen=32):
    rng = np.random.default_rng(seed)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msg = rng.integers(0, 256, (n, mlen), dtype=np.uint8)
    pk = engine.ed25519_genpub_batch(dev(sk))
    sig = engine.ed25519_sign_batch(dev(sk), pk, dev(msg)).cpu().numpy()
    return sig, pk.cpu().numpy(), msg


def test_all_valid_groups_pass_by_combination(engine):
    n = 3 * G + 1000                                   # three full groups and a partial one
    sig, pk, msg = _signed(engine, n, 1)
    ok, st = engine.ed25519_verify_batch_rlc(dev(sig), dev(pk), dev(msg), msg_len=32, return_stats=True)
    assert bool(ok.all()) and st == (n, 0, 0, 4)
    ok, st = engine.ed25519_verify_batch_rlc(sig, pk, msg, msg_len=32, return_stats=True)   # host-pointer pipeline
    assert bool(ok.all()) and st == (n, 0, 0, 4)
    # S + k l (S is reduced, not range-checked: sc.c:191-214) still passes with its group
    s2 = sig.copy()
    for i, k in ((5, 1), (G + 7, 7), (2 * G + 9, 14)):
        s2[i, 32:] = np.frombuffer((int.from_bytes(sig[i, 32:].tobytes(), "little") + k * L).to_bytes(32, "little"), np.uint8)
    ok, st = engine.ed25519_verify_batch_rlc(dev(s2), dev(pk), dev(msg), msg_len=32, return_stats=True)
    assert bool(ok.all()) and st == (n, 0, 0, 4)


def test_small_calls_go_to_the_per_item_kernels(engine):
    """default routing: below 5 x 2^15 items the combination's own latency (hash tree, Horner) costs more than it
    saves, so the call uses the per-item kernels; from 5 x 2^15 = 163 840 on it combines"""
    engine.set_rlc_min_items(engine.RLC_MIN_ITEMS_DEFAULT)
    assert engine.RLC_MIN_ITEMS_DEFAULT == 5 << 15
    sig, pk, msg = _signed(engine, 400000, 11)
    n = 3 * G
    ok, st = engine.ed25519_verify_batch_rlc(dev(sig[:n]), dev(pk[:n]), dev(msg[:n]), msg_len=32, return_stats=True)
    assert bool(ok.all()) and st == (0, n, 3, 0)
    ok, st = engine.ed25519_verify_batch_rlc(sig[:n], pk[:n], msg[:n], msg_len=32, return_stats=True)      # host-pointer pipeline
    assert bool(ok.all()) and st == (0, n, 3, 0)
    n = 400000
    ok, st = engine.ed25519_verify_batch_rlc(dev(sig), dev(pk), dev(msg), msg_len=32, return_stats=True)
    assert bool(ok.all()) and st == (n, 0, 0, 49)


def test_tiny_batches(engine):
    """one item, a handful of items: a single partial group, a one-level hash tree"""
    sig, pk, msg = _signed(engine, 5, 9)
    for n in (1, 2, 5):
        ok, st = engine.ed25519_verify_batch_rlc(sig[:n], pk[:n], msg[:n], msg_len=32, return_stats=True)
        assert bool(ok.all()) and st == (n, 0, 0, 1)
    bad = sig.copy(); bad[0, 33] ^= 1
    ok, st = engine.ed25519_verify_batch_rlc(bad[:1], pk[:1], msg[:1], msg_len=32, return_stats=True)
    assert ok.tolist() == [0] and st == (0, 1, 1, 0)
    ok = engine.ed25519_verify_batch_rlc(bad, pk, msg, msg_len=32)
    assert ok.tolist() == [0, 1, 1, 1, 1]
    assert engine.ed25519_verify_batch_rlc(sig[:0], pk[:0], msg[:0], msg_len=32).shape == (0,)


def test_a_failing_group_is_decided_per_item(engine, oracle):
    n = 4 * G
    sig, pk, msg = _signed(engine, n, 2)
    bad = sig.copy()
    bad[G + 17, 3] ^= 0x10                              # R of one item of group 1: another valid point encoding or none
    bad[3 * G + 100, 40] ^= 1                           # S of one item of group 3
    msg2 = msg.copy(); msg2[3 * G + 101, 0] ^= 1
    ok, st = engine.ed25519_verify_batch_rlc(dev(bad), dev(pk), dev(msg2), msg_len=32, return_stats=True)
    want = np.ones(n, np.uint8); want[[G + 17, 3 * G + 100, 3 * G + 101]] = 0
    assert np.array_equal(ok.cpu().numpy(), want)
    assert st[2] + st[3] == 4 and st[2] in (1, 2) and st[0] + st[1] == n    # group 1 fails only if its R still decodes
    assert np.array_equal(want[G:G + 64], oracle.verify_batch(bad[G:G + 64], pk[G:G + 64], msg2[G:G + 64], 32))


def test_a_window_flag_that_is_never_raised_costs_the_pass_a_wait_not_its_result(engine):
    """k_rlc_horner takes each group's window points as k_rlc_bucket's blocks raise their flags.  With the test hook keeping
    the flag of window point 40 of group 0 down, that group's Horner wave (groups 0..15: all four here) polls for its bound -
    14 ms, where the bucket launch takes 2 - marks its groups undecided and ends; k_rlc_final, queued behind both
    launches, evaluates them from memory.  Same verdicts, same statistics, nothing hangs, nothing is an error - and the hook
    reports the wave that gave up (none in the passes before and after)"""
    n = 4 * G
    sig, pk, msg = _signed(engine, n, 5)
    engine.debug_init(0, True)
    try:
        ok, st = engine.ed25519_verify_batch_rlc(dev(sig), dev(pk), dev(msg), msg_len=32, return_stats=True)
        assert bool(ok.all()) and st == (n, 0, 0, 4)
        assert engine.debug_withhold_handoff(41) == 0                    # (no wave gave up so far)
        bad = sig.copy(); bad[2 * G + 5, 40] ^= 1                        # ... and a group that fails, decided by the same route
        ok, st = engine.ed25519_verify_batch_rlc(dev(bad), dev(pk), dev(msg), msg_len=32, return_stats=True)
        want = np.ones(n, np.uint8); want[2 * G + 5] = 0
        assert np.array_equal(ok.cpu().numpy(), want) and st == (3 * G, G, 1, 3)
        assert engine.debug_withhold_handoff(0) == 1                     # one wave (groups 0..15) gave up, once
        ok, st = engine.ed25519_verify_batch_rlc(dev(sig), dev(pk), dev(msg), msg_len=32, return_stats=True)
        assert bool(ok.all()) and st == (n, 0, 0, 4)
        assert engine.debug_withhold_handoff(0) == 0
    finally:
        engine.debug_withhold_handoff(0)
        engine.debug_init(0, False)


def test_concurrent_combinations_on_three_streams(engine):
    """Three host threads, a stream each, six passes each, at once: every pass has its own workspace, bucket launch and Horner
    kernel, and a Horner kernel polls flags that only ITS pass's bucket blocks raise - which may still be waiting for another
    pass's blocks to leave the chip.  Every pass must return its own verdicts (one thread's batch is all valid, one has a
    failing group, one a group with a small-order key that goes per item), with the statistics of a pass run alone"""
    import threading
    import torch
    n = 6 * G + 300
    sig, pk, msg = _signed(engine, n, 77)
    cases = []
    for kind in range(3):
        s2, p2 = sig.copy(), pk.copy()
        want = np.ones(n, np.uint8)
        if kind == 1:
            s2[2 * G + 9, 35] ^= 4; want[2 * G + 9] = 0
        if kind == 2:
            p2[4 * G + 1] = np.frombuffer(bytes.fromhex("0100000000000000000000000000000000000000000000000000000000000000"), np.uint8)
            want[4 * G + 1] = 0                                   # the neutral element as a key: its group is decided per item
        ok, st = engine.ed25519_verify_batch_rlc(dev(s2), dev(p2), dev(msg), msg_len=32, return_stats=True)
        assert np.array_equal(ok.cpu().numpy(), want), kind
        cases.append((dev(s2), dev(p2), torch.from_numpy(want).cuda(), st))
    dm = dev(msg)
    errs = []

    def worker(kind):
        s2, p2, want, st0 = cases[kind]
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for r in range(6):
                ok, st = engine.ed25519_verify_batch_rlc(s2, p2, dm, msg_len=32, return_stats=True)
                stream.synchronize()
                if not torch.equal(ok, want) or st != st0:
                    errs.append((kind, r, int((ok != want).sum()), st, st0))
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs[:4]


def test_routing_of_items_the_combination_cannot_represent(engine, oracle):
    """non-canonical / off-curve R: rejected at once, the group still passes by combination; off-curve or
    small-order A, small-order R: the whole group goes to the per-item kernels"""
    n = 3 * G
    sig, pk, msg = _signed(engine, n, 3)
    s, p = sig.copy(), pk.copy()
    le = lambda x: np.frombuffer(int(x).to_bytes(32, "little"), np.uint8)  # noqa: E731
    s[10, :32] = le(P + 1)                              # group 0: R = identity written as y = p + 1
    s[11, :32] = le(2)                                  #          R with y = 2: not on the curve
    s[12, :32] = le(1 | 1 << 255)                       #          R = (0, 1) with the sign bit: never canonical
    p[G + 5] = le(2)                                    # group 1: off-curve key
    p[2 * G + 5] = le(P - 1)                            # group 2: key of order 2
    s[2 * G + 6, :32] = le(1); s[2 * G + 6, 32:] = 0; p[2 * G + 6] = le(1)   # R = A = identity, S = 0: accepted per item
    ok, st = engine.ed25519_verify_batch_rlc(dev(s), dev(p), dev(msg), msg_len=32, return_stats=True)
    ok = ok.cpu().numpy()
    assert np.array_equal(ok, oracle.verify_batch(s, p, msg, 32))
    assert ok[2 * G + 6] == 1 and not ok[[10, 11, 12, G + 5, 2 * G + 5]].any()
    assert st == (G, 2 * G, 2, 1)


def test_edge_fixture_and_ragged_messages(engine, golden):
    """the 273 reference-pinned edge cases (S + k l, non-canonical and small-order A, off-curve A, flipped bits,
    message lengths 0..1023), once alone and once embedded in genuine traffic"""
    cases = golden("verify_edges.json")
    sig = np.array([list(bytes.fromhex(c["sig"])) for c in cases], np.uint8)
    pub = np.array([list(bytes.fromhex(c["pub"])) for c in cases], np.uint8)
    msgs = [bytes.fromhex(c["msg"]) for c in cases]
    want = np.array([c["accept"] for c in cases], np.uint8)
    off = np.concatenate([[0], np.cumsum([len(m) for m in msgs])]).astype(np.uint64)
    flat = np.frombuffer(b"".join(msgs), np.uint8)
    assert np.array_equal(engine.ed25519_verify_batch_rlc(sig, pub, flat, msg_off=off), want)
    gs, gp, gm = _signed(engine, 2 * G, 4, mlen=20)
    sig2 = np.concatenate([gs[:G], sig, gs[G:]]); pub2 = np.concatenate([gp[:G], pub, gp[G:]])
    lens = np.concatenate([np.full(G, 20), [len(m) for m in msgs], np.full(G, 20)])
    off2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    flat2 = np.concatenate([gm[:G].reshape(-1), flat, gm[G:].reshape(-1)])
    ok, st = engine.ed25519_verify_batch_rlc(sig2, pub2, flat2, msg_off=off2, return_stats=True)
    assert np.array_equal(ok, np.concatenate([np.ones(G, np.uint8), want, np.ones(G, np.uint8)]))
    assert st[3] >= 1 and st[2] >= 1


def test_config2_batch_digest_through_the_rlc_entry_point(engine, golden):
    """the 2^20 config-2 batch (1/16 corrupted + edge vectors: every group fails or is flagged) comes back
    with the reference's verdict digest; an all-valid 2^20 + 5000 batch (two passes) is accepted whole"""
    import workload
    v = golden("batch_digests.json")["verify_2^20"]
    n = 1 << 20
    sk, msg = workload.sign_inputs(n, seed=1, config=2)
    pk = engine.ed25519_genpub_batch(dev(sk))
    sig = engine.ed25519_sign_batch(dev(sk), pk, dev(msg))
    ok, st = engine.ed25519_verify_batch_rlc(sig, pk, dev(msg), msg_len=32, return_stats=True)
    assert bool(ok.all()) and st == (n, 0, 0, n // G)
    sig, pk = sig.cpu().numpy(), pk.cpu().numpy()
    expect = workload.corrupt_for_verify(sig, pk, msg)
    ok, st = engine.ed25519_verify_batch_rlc(dev(sig), dev(pk), dev(msg), msg_len=32, return_stats=True)
    ok = ok.cpu().numpy()
    assert np.array_equal(ok, expect) and hashlib.sha512(ok.tobytes()).hexdigest() == v["verdicts_sha512"]
    assert st == (0, n, n // G, 0)
    n2 = n + 5000
    s2, p2, m2 = _signed(engine, n2, 6)
    ok, st = engine.ed25519_verify_batch_rlc(dev(s2), dev(p2), dev(m2), msg_len=32, return_stats=True)
    assert bool(ok.all()) and st == (n2, 0, 0, n // G + 1)


def test_coefficients_depend_on_the_whole_batch(engine):
    """the combination is deterministic (same batch, same group verdicts) and a change anywhere in the batch
    changes the seed: two invalid items whose errors cancel under one set of coefficients cannot be arranged,
    here checked the cheap way - repeated runs agree, and runs on permuted batches still reject the bad item"""
    n = 2 * G
    sig, pk, msg = _signed(engine, n, 7)
    sig[77, 35] ^= 2
    want = np.ones(n, np.uint8); want[77] = 0
    a = engine.ed25519_verify_batch_rlc(sig, pk, msg, msg_len=32)
    b = engine.ed25519_verify_batch_rlc(sig, pk, msg, msg_len=32)
    assert np.array_equal(a, want) and np.array_equal(b, want)
    perm = np.random.default_rng(0).permutation(n)
    c = engine.ed25519_verify_batch_rlc(sig[perm], pk[perm], msg[perm], msg_len=32)
    assert np.array_equal(c, want[perm])


# ---------------------------------------------------------------------------------------------------------
# mixed-order keys and commitments (VERDICT r02 / ADVICE r02): A = a B + T, R = r B + T' with T, T' of order dividing 8.
# The key's scalar of the combination is z t mod 8 l (rlc_lanes.h: rlc_key_scalar_mod_8l), so a ONE-item combination is
# exactly the per-item check; with z t mod l one such item in eight passed that the reference rejects.
# ---------------------------------------------------------------------------------------------------------
def _torsion_cases(golden):
    cases = golden("verify_torsion.json")
    sig = np.array([list(bytes.fromhex(c["sig"])) for c in cases], np.uint8)
    pub = np.array([list(bytes.fromhex(c["pub"])) for c in cases], np.uint8)
    msgs = [bytes.fromhex(c["msg"]) for c in cases]
    want = np.array([c["accept"] for c in cases], np.uint8)
    return sig, pub, msgs, want


def test_torsion_fixture_as_one_item_combinations(engine, golden):
    """each of the 192 reference-pinned vectors as its own call (a group of one): the fixture's verdict, and decided by the
    combination itself whenever that verdict is 'accept' (a rejected one falls to the per-item kernels, same verdict)"""
    sig, pub, msgs, want = _torsion_cases(golden)
    by_combination = 0
    for i in range(len(msgs)):
        m = np.frombuffer(msgs[i], np.uint8)
        ok, st = engine.ed25519_verify_batch_rlc(sig[i:i + 1], pub[i:i + 1], m, msg_len=len(msgs[i]), return_stats=True)
        assert ok.tolist() == [want[i]], i
        assert st == ((1, 0, 0, 1) if want[i] else (0, 1, 1, 0)), (i, st)
        by_combination += st[3]
    assert by_combination == int(want.sum()) == 27


def test_torsion_fixture_one_vector_per_group_in_valid_traffic(engine, golden):
    """the same vectors, one in each of 192 groups of otherwise genuine signatures (two passes): a group passes by
    combination exactly when its vector is one the reference accepts, and every verdict byte is the fixture's"""
    sig, pub, msgs, want = _torsion_cases(golden)
    k, mlen = len(msgs), len(msgs[0])
    assert all(len(m) == mlen for m in msgs)
    n = k * G
    gs, gp, gm = _signed(engine, n, 41, mlen=mlen)
    slots = np.arange(k) * G + (np.arange(k) * 37 + 11) % G
    gs[slots], gp[slots] = sig, pub
    gm[slots] = np.array([list(m) for m in msgs], np.uint8)
    expect = np.ones(n, np.uint8)
    expect[slots] = want
    ok, st = engine.ed25519_verify_batch_rlc(dev(gs), dev(gp), dev(gm), msg_len=mlen, return_stats=True)
    assert np.array_equal(ok.cpu().numpy(), expect)
    acc = int(want.sum())
    assert st == (acc * G, (k - acc) * G, k - acc, acc)


def test_mixed_order_keys_never_pass_a_combination_the_reference_rejects(engine, oracle):
    """400 signatures with an honest prime-order part under keys A' = a B + T8 (sign takes the key unchecked,
    lib/ed25519-sha512.c:84-123): S B - t A' - R = -t T8, accepted by the reference iff 8 | t.  One-item combinations
    and one-per-group in valid traffic: no accept the oracle rejects (it was 51 of 400 with the scalar mod l)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden import ed_add_affine, ed_dec, ed_enc, small_order_points
    t8 = ed_dec(small_order_points()[4])
    assert ed_add_affine(ed_add_affine(t8, t8), ed_add_affine(t8, t8)) == (0, P - 1)     # 4 T8 = (0, -1): order 8
    k, mlen = 400, 24
    rng = np.random.default_rng(31)
    sk = rng.integers(0, 256, (k, 32), dtype=np.uint8)
    msg = rng.integers(0, 256, (k, mlen), dtype=np.uint8)
    pk = engine.ed25519_genpub_batch(dev(sk)).cpu().numpy()
    pk1 = np.array([list(ed_enc(ed_add_affine(ed_dec(pk[i].tobytes()), t8))) for i in range(k)], np.uint8)
    sig = engine.ed25519_sign_batch(dev(sk), dev(pk1), dev(msg)).cpu().numpy()
    want = oracle.verify_batch(sig, pk1, msg, mlen)
    assert 25 <= int(want.sum()) <= 80                                   # about one in eight
    assert np.array_equal(engine.ed25519_verify_batch(sig, pk1, msg, msg_len=mlen), want)
    for i in range(k):
        ok, st = engine.ed25519_verify_batch_rlc(sig[i:i + 1], pk1[i:i + 1], msg[i], msg_len=mlen, return_stats=True)
        assert ok.tolist() == [want[i]] and st[3] == want[i], i
    groups = 64
    gs, gp, gm = _signed(engine, groups * G, 43, mlen=mlen)
    slots = np.arange(groups) * G + 5
    gs[slots], gp[slots], gm[slots] = sig[:groups], pk1[:groups], msg[:groups]
    expect = np.ones(groups * G, np.uint8)
    expect[slots] = want[:groups]
    ok, st = engine.ed25519_verify_batch_rlc(dev(gs), dev(gp), dev(gm), msg_len=mlen, return_stats=True)
    assert np.array_equal(ok.cpu().numpy(), expect)
    assert st[3] == int(want[:groups].sum()) and st[2] == groups - st[3]
