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
    """these tests drive the combination itself with small batches: switch off the routing of calls below 3 x 2^17
    items to the per-item kernels (test_small_calls_go_to_the_per_item_kernels checks the default)"""
    engine.set_rlc_min_items(0)
    yield
    engine.set_rlc_min_items(3 << 17)


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
    """default routing: below 3 x 2^17 items the combination's own latency (hash tree, Horner) costs more than it
    saves, so the call uses the per-item kernels; from 3 x 2^17 on it combines"""
    engine.set_rlc_min_items(3 << 17)
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
