"""Full-size (BASELINE.json configs: batch 2^20) checks on the GPU: output digests precomputed by
the compiled reference in the build container (tests/golden/batch_digests.json), plus
size-independent properties."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 1 << 20


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_x25519_2_20_digest_and_commutativity(engine, golden):
    import workload
    sc, pt = workload.x25519_inputs(N)
    out = engine.x25519_batch(dev(sc), dev(pt)).cpu().numpy()
    assert hashlib.sha512(out.tobytes()).hexdigest() == golden("batch_digests.json")["x25519_2^20"]
    # Diffie-Hellman: a*(b*G) == b*(a*G)
    a, b = dev(sc), dev(pt)
    ga, gb = engine.x25519_base_batch(a), engine.x25519_base_batch(b)
    import torch
    assert torch.equal(engine.x25519_batch(a, gb), engine.x25519_batch(b, ga))


def test_sign_2_20_digest_and_round_trip(engine, golden):
    import torch
    import workload
    d = golden("batch_digests.json")
    sk, msg = workload.sign_inputs(N)
    pk = engine.ed25519_genpub_batch(dev(sk))
    assert hashlib.sha512(pk.cpu().numpy().tobytes()).hexdigest() == d["genpub_2^20"]
    sig = engine.ed25519_sign_batch(dev(sk), pk, dev(msg))
    assert hashlib.sha512(sig.cpu().numpy().tobytes()).hexdigest() == d["sign_2^20"]
    assert bool(engine.ed25519_verify_batch(sig, pk, dev(msg)).all())          # sign -> verify round trip
    # any single flipped message bit is rejected
    bad = dev(msg).clone(); bad[:, 7] ^= 0x20
    assert int(engine.ed25519_verify_batch(sig, pk, bad).sum()) == 0


def test_verify_2_20_config2(engine, golden):
    """config 2: the reference's verdicts on the seeded batch are the constructed 15/16 pattern"""
    import workload
    v = golden("batch_digests.json")["verify_2^20"]
    sk, msg = workload.sign_inputs(N, seed=1, config=2)
    pk = engine.ed25519_genpub_batch(dev(sk))
    sig = engine.ed25519_sign_batch(dev(sk), pk, dev(msg)).cpu().numpy()
    pk = pk.cpu().numpy()
    expect = workload.corrupt_for_verify(sig, pk, msg)
    assert hashlib.sha512(sig.tobytes() + pk.tobytes() + msg.tobytes()).hexdigest() == v["inputs_sha512"]
    ok = engine.ed25519_verify_batch(dev(sig), dev(pk), dev(msg)).cpu().numpy()
    assert np.array_equal(ok, expect) and int(ok.sum()) == v["accepted"]
    assert hashlib.sha512(ok.tobytes()).hexdigest() == v["verdicts_sha512"]


def test_verify_across_workspace_chunks(engine):
    """n > 2^20 goes through the workspace in two passes"""
    import workload
    n = N + 1000
    sk, msg = workload.sign_inputs(n, seed=9, config=4)
    pk = engine.ed25519_genpub_batch(dev(sk))
    sig = engine.ed25519_sign_batch(dev(sk), pk, dev(msg)).cpu().numpy()
    pk = pk.cpu().numpy()
    expect = workload.corrupt_for_verify(sig, pk, msg, seed=9, config=4)
    ok = engine.ed25519_verify_batch(dev(sig), dev(pk), dev(msg)).cpu().numpy()
    assert np.array_equal(ok, expect)


def test_reference_order_kernels_on_a_large_pass(engine):
    """self-check mode 2 on a pass of more than 2^19 items: the one-lane chain decides the first 65 536
    items and the strided k_verify_exact the rest; verdicts equal the windowed kernels' (the smaller
    pass of tests/test_gpu_parity.py covers the four-lane chain)"""
    import workload
    n = (1 << 19) + 5000
    sk, msg = workload.sign_inputs(n, seed=12, config=2)
    pk = engine.ed25519_genpub_batch(dev(sk))
    sig = engine.ed25519_sign_batch(dev(sk), pk, dev(msg)).cpu().numpy()
    pk = pk.cpu().numpy()
    expect = workload.corrupt_for_verify(sig, pk, msg)
    engine.set_offcurve_mode(2)
    try:
        replay = engine.ed25519_verify_batch(dev(sig), dev(pk), dev(msg), msg_len=32).cpu().numpy()
    finally:
        engine.set_offcurve_mode(True)
    assert np.array_equal(replay, expect) and 0 < expect.sum() < n


def test_random_keys_2_20_against_the_oracle_item_by_item(engine, oracle):
    """the worst case at BASELINE's size: 2^20 genuine signatures of config 2 under random 32-byte keys (47 % are no curve points and
    take the two-items-per-lane replay of the reference's chain, 493 k entries; every 16th key stays genuine; every 5th of the bad keys
    meets R = 0, the chain's Z = 0 corner) - all 2^20 verdicts against the oracle's, which pins the reference byte for byte on keys
    off the curve (tests/test_oracle_vs_ref.py), device pointers and, in chunks, host pointers"""
    import workload
    sk, msg = workload.sign_inputs(N, seed=3, config=2)
    pk = engine.ed25519_genpub_batch(dev(sk)).cpu().numpy()
    sig = engine.ed25519_sign_batch(dev(sk), dev(pk), dev(msg)).cpu().numpy()
    rng = np.random.default_rng(2020)
    keys = rng.integers(0, 256, (N, 32), dtype=np.uint8)
    keys[3::16] = pk[3::16]
    sig[5::80, :32] = 0
    want = oracle.verify_batch(sig, keys, msg, 32)
    assert want[3::16].sum() > N // 20 and want.sum() < N // 8
    assert np.array_equal(engine.ed25519_verify_batch(dev(sig), dev(keys), dev(msg), msg_len=32).cpu().numpy(), want)
    assert np.array_equal(engine.ed25519_verify_batch(sig, keys, msg, msg_len=32), want)
