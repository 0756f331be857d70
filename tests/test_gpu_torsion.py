"""Mixed-order inputs AT SCALE.  The reference checks neither subgroup membership nor small order
(/root/reference/lib/ed25519-sha512.c:148-181 over lib/ed.c:100-149's permissive ed_import): a key A' = a B + T_A and a commitment
R' = r B + T_R with T_A, T_R of order dividing 8 are accepted exactly when t T_A + T_R = 0.  Those are the inputs on which the
half-length route's v = u t (mod 8 l) (csrc/halve.h) and the batch verification's scalar mod 8 l (csrc/rlc.hip) decide the
verdict byte; until round 5 the device saw them only as the 192 fixture vectors, in passes of at most 465 items.  Here: a pass
of 2^17 + 333 items of the config-2 recipe with a random torsion component in a third of the items (tools/workload.py:
add_torsion), and the same items four times over as a pass of 2^19 + 333 - on every evaluation the library has, and through
ed25519_verify_batch_rlc; every expected byte is the oracle's."""
import numpy as np
import pytest

import workload

pytestmark = pytest.mark.gpu
G = 8192


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def batch(engine, oracle):
    n = (1 << 17) + 333
    sk, msg = workload.sign_inputs(n, seed=11, config=2)
    pk = engine.ed25519_genpub_batch(sk)
    sig = engine.ed25519_sign_batch(sk, pk, msg)
    s2, p2, touched = workload.add_torsion(sk, pk, sig, msg, seed=11)            # from the genuine items ...
    m0 = msg.copy()
    expect = workload.corrupt_for_verify(sig, pk, msg, seed=11, config=2)       # ... beside the config-2 mix: 1/16 corrupted, edge vectors
    sig[touched], pk[touched], msg[touched] = s2[touched], p2[touched], m0[touched]
    want = oracle.verify_batch(sig, pk, msg, 32)
    assert np.array_equal(want[~touched], expect[~touched])
    # t T_A + T_R = 0 for 5 of 16 keys with T_A alone (by the order of T_A: 1, 1/2, 1/4, 1/8), for 1 in 8 otherwise: 3/16 overall
    acc = int(want[touched].sum())
    assert abs(int(touched.sum()) - n // 3) <= 1 and 0.16 * touched.sum() < acc < 0.215 * touched.sum(), (touched.sum(), acc)
    return sig, pk, msg, want, touched


@pytest.fixture(autouse=True)
def restore(engine):
    yield
    engine.set_verify_algo(0)
    engine.set_rlc_min_items(engine.RLC_MIN_ITEMS_DEFAULT)


def test_every_evaluation_on_a_third_of_the_items_with_torsion(engine, batch):
    """2^17 + 333 items: the default arrangement of that size (three-lane preparation, one-lane half-length evaluation with the
    long loop in place), full-length windows, k_verify_halve + k_verify_main_half<35>, the mid-size arrangement; then the
    host-pointer pipeline (chunks of 2^16 + 2^17: the four-lane evaluation never sees these sizes, the fixture tests cover it)"""
    sig, pk, msg, want, touched = batch
    d = dev(sig), dev(pk), dev(msg)
    for algo in (0, 1, 2, 3):
        engine.set_verify_algo(algo)
        got = engine.ed25519_verify_batch(*d, msg_len=32).cpu().numpy()
        assert np.array_equal(got, want), (algo, np.nonzero(got != want)[0][:10], touched[np.nonzero(got != want)[0][:10]])
    engine.set_verify_algo(0)
    assert np.array_equal(engine.ed25519_verify_batch(sig, pk, msg, msg_len=32), want)


def test_the_large_pass_route_on_the_same_items_four_times_over(engine, batch):
    """2^19 + 333 items: k_verify_halve<134> + k_verify_main_half<34>, the route of the headline batch (pairs searched up to
    2^134 instead of 2^138: other pairs (u, v) for the same t, and the items without a short pair join the exact path)"""
    sig, pk, msg, want, _ = batch
    n = (1 << 19) + 333
    reps = -(-n // sig.shape[0])
    big = [np.tile(a, (reps, 1))[:n] for a in (sig, pk, msg)]
    got = engine.ed25519_verify_batch(*(dev(a) for a in big), msg_len=32).cpu().numpy()
    assert np.array_equal(got, np.tile(want, reps)[:n])


def test_batch_verification_with_one_such_item_per_group(engine, batch):
    """32 groups of genuine signatures with ONE rewritten item each: a combination of items of which one carries a torsion
    defect passes exactly when that defect is zero (z is odd and below l: csrc/rlc.hip), so the verdict bytes are the
    per-item ones, and the groups decided by the combination are those whose item the reference accepts"""
    sig, pk, msg, want, touched = batch
    idx = np.nonzero(touched)[0]
    pick = np.concatenate([idx[want[idx] == 1][:12], idx[want[idx] == 0][:20]])
    groups = len(pick)
    sk2, msg2 = workload.sign_inputs(groups * G, seed=12, config=2)
    pk2 = engine.ed25519_genpub_batch(sk2)
    sig2 = engine.ed25519_sign_batch(sk2, pk2, msg2)
    slots = np.arange(groups) * G + (np.arange(groups) * 53 + 7) % G
    sig2[slots], pk2[slots], msg2[slots] = sig[pick], pk[pick], msg[pick]
    expect = np.ones(groups * G, np.uint8)
    expect[slots] = want[pick]
    engine.set_rlc_min_items(0)
    ok, st = engine.ed25519_verify_batch_rlc(dev(sig2), dev(pk2), dev(msg2), msg_len=32, return_stats=True)
    assert np.array_equal(ok.cpu().numpy(), expect)
    assert st == (12 * G, 20 * G, 20, 12)


def test_batch_verification_with_thousands_of_them_per_group(engine, batch):
    """The whole pass through ed25519_verify_batch_rlc: 17 groups with some 2700 rewritten items each.  The documented caveat
    (csrc/rlc.hip, include/eddsa_amd.h) is that defects of small order can cancel inside a group, so a group may pass although
    the reference rejects some of its items; outside it nothing may differ: every item that was not rewritten, and every item of
    a group that contains a defect of another kind (the corrupted signatures of the config-2 mix: 1 in 16, so every group here),
    has the reference's verdict - with a corrupted item in every group, that is every byte."""
    sig, pk, msg, want, touched = batch
    engine.set_rlc_min_items(0)
    got, st = engine.ed25519_verify_batch_rlc(dev(sig), dev(pk), dev(msg), msg_len=32, return_stats=True)
    got = got.cpu().numpy()
    assert np.array_equal(got, want)
    assert st[3] == 0                                           # no group passed by combination
    # ... and where the caveat CAN apply - groups whose only rejected items are torsion defects: the valid signatures of the pass
    # plus the rewritten items, rejected ones included
    keep = (want == 1) | touched
    s2, p2, m2, w2, t2 = sig[keep], pk[keep], msg[keep], want[keep], touched[keep]
    got, st = engine.ed25519_verify_batch_rlc(dev(s2), dev(p2), dev(m2), msg_len=32, return_stats=True)
    got = got.cpu().numpy()
    diff = np.nonzero(got != w2)[0]
    assert np.array_equal(got[~t2], w2[~t2])                                   # nothing but rewritten items may differ,
    assert (got[diff] == 1).all() and (w2[diff] == 0).all()                    # only as accepts the per-item check rejects,
    for g in np.unique(diff // G):                                            # only whole groups at a time (the group passed) ...
        members = np.arange(g * G, min((g + 1) * G, len(w2)))
        rejected = members[w2[members] == 0]
        assert set(diff[diff // G == g]) == set(rejected) and len(rejected) >= 2 and t2[rejected].all()
    # ... and that does happen: with thousands of defects drawn from a group of eight elements about one group in eight cancels
    print("groups:", -(-len(w2) // G), "passed by combination although the reference rejects items of theirs:", len(np.unique(diff // G)), st)
