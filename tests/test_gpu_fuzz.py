"""Seeded differential fuzz on the GPU: random batch sizes (on both sides of every route boundary),
message lengths (fixed and ragged), corruptions and garbage keys, every batched entry point against
the oracle, host-pointer and device-pointer paths.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


SIZES = [1, 2, 3, 63, 64, 65, 255, 256, 257, 511, 777, 1024, 2047, 2048, 2049, 4099, 24576, 24577, 16383, 16384, 16385, 20011, 32768, 32769, 33000]


@pytest.fixture(scope="module")
def device_set(engine):
    return engine.init_devices()


@pytest.fixture(autouse=True)
def always_combine(engine):
    """small batches through the combination itself (by default calls below 5 x 2^15 items use the per-item kernels)"""
    engine.set_rlc_min_items(0)
    yield
    engine.set_rlc_min_items(engine.RLC_MIN_ITEMS_DEFAULT)


# EDDSA_FUZZ_EXTRA=k adds k more random cases (a longer soak after kernel changes)
@pytest.mark.parametrize("case", range(len(SIZES) + 5 + int(__import__("os").environ.get("EDDSA_FUZZ_EXTRA", "0"))))
def test_fuzz_against_the_oracle(engine, oracle, device_set, case):
    rng = np.random.default_rng(1000 + case)
    n = int(SIZES[case % len(SIZES)]) if case < len(SIZES) else int(rng.integers(1, 9000 if case < len(SIZES) + 5 else 70000))
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ragged = case % 3 == 1
    if ragged:
        lens = rng.integers(0, 200, n)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        msgs = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)
        one = lambda i: msgs[int(off[i]):int(off[i + 1])].tobytes()   # noqa: E731
    else:
        mlen = int(rng.choice([0, 1, 31, 32, 47, 48, 64, 111, 112, 113, 129]))
        msgs = rng.integers(0, 256, (n, mlen), dtype=np.uint8)
    # genpub, sign: device results against the oracle
    pk = oracle.genpub_batch(sk)
    assert np.array_equal(engine.ed25519_genpub_batch(sk), pk)
    if ragged:
        sig = np.stack([np.frombuffer(oracle.sign(sk[i].tobytes(), pk[i].tobytes(), one(i)), np.uint8) for i in range(min(n, 300))])
        got = engine.ed25519_sign_batch(dev(sk), dev(pk), dev(msgs), msg_off=dev(off.astype(np.int64))).cpu().numpy()
        assert np.array_equal(got[:len(sig)], sig)
        sig = got.copy()
    else:
        sig = oracle.sign_batch(sk, pk, msgs, mlen)
        assert np.array_equal(engine.ed25519_sign_batch(dev(sk), dev(pk), dev(msgs), msg_len=mlen).cpu().numpy(), sig)
    # corrupt: R, S, A, garbage keys (about half of those are off the curve), S + l
    kind = rng.integers(0, 8, n)
    bit = rng.integers(0, 256, n)
    for i in np.nonzero(kind == 1)[0]: sig[i, bit[i] // 8 % 32] ^= 1 << (bit[i] % 8)
    for i in np.nonzero(kind == 2)[0]: sig[i, 32 + bit[i] // 8 % 32] ^= 1 << (bit[i] % 8)
    for i in np.nonzero(kind == 3)[0]: pk[i, bit[i] // 8 % 32] ^= 1 << (bit[i] % 8)
    g = np.nonzero(kind == 4)[0]
    pk[g] = rng.integers(0, 256, (len(g), 32), dtype=np.uint8)
    ell = 2**252 + 27742317777372353535851937790883648493
    for i in np.nonzero(kind == 5)[0]:
        s = int.from_bytes(sig[i, 32:].tobytes(), "little") + ell
        if s < 2**256: sig[i, 32:] = np.frombuffer(s.to_bytes(32, "little"), np.uint8)
    if ragged:
        want = np.array([oracle.verify(sig[i].tobytes(), pk[i].tobytes(), one(i)) for i in range(n)], np.uint8)
        got_h = engine.ed25519_verify_batch(sig, pk, msgs, msg_off=off)
        got_d = engine.ed25519_verify_batch(dev(sig), dev(pk), dev(msgs), msg_off=dev(off.astype(np.int64))).cpu().numpy()
    else:
        want = oracle.verify_batch(sig, pk, msgs, mlen)
        got_h = engine.ed25519_verify_batch(sig, pk, msgs, msg_len=mlen)
        got_d = engine.ed25519_verify_batch(dev(sig), dev(pk), dev(msgs), msg_len=mlen).cpu().numpy()
    assert np.array_equal(got_h, want) and np.array_equal(got_d, want)
    # both evaluations (full-length windows, half-length scalars) whatever the pass size
    for algo in (1, 2):
        engine.set_verify_algo(algo)
        try:
            got_a = (engine.ed25519_verify_batch(sig, pk, msgs, msg_off=off) if ragged else
                     engine.ed25519_verify_batch(dev(sig), dev(pk), dev(msgs), msg_len=mlen).cpu().numpy())
        finally:
            engine.set_verify_algo(0)
        assert np.array_equal(got_a, want), algo
    # the opt-in batch verification (random linear combination per group of 8192, per-item fallback) and the
    # single-process multi-device form return the same verdict bytes
    if ragged:
        got_r = engine.ed25519_verify_batch_rlc(sig, pk, msgs, msg_off=off)
        got_m = engine.ed25519_verify_batch_multi(sig, pk, msgs, msg_off=off)
    else:
        got_r = engine.ed25519_verify_batch_rlc(dev(sig), dev(pk), dev(msgs), msg_len=mlen).cpu().numpy()
        got_m = engine.ed25519_verify_batch_multi(sig, pk, msgs, msg_len=mlen)
    assert np.array_equal(got_r, want) and np.array_equal(got_m, want)
    # x25519 and x25519_base on the same random bytes
    assert np.array_equal(engine.x25519_batch(dev(sk), dev(pk)).cpu().numpy(), oracle.x25519_batch(sk, pk))
    m = min(n, 500)
    xb = np.stack([np.frombuffer(oracle.x25519_base(sk[i].tobytes()), np.uint8) for i in range(m)])
    assert np.array_equal(engine.x25519_base_batch(dev(sk)).cpu().numpy()[:m], xb)


@pytest.mark.parametrize("case", range(6))
def test_fuzz_batch_verification_on_mostly_valid_traffic(engine, oracle, case):
    """batches of several groups in which only a few items are bad (the regime batch verification is for): the
    verdicts are the per-item ones, most groups are decided by the combination"""
    rng = np.random.default_rng(5000 + case)
    n = int(rng.integers(2 * 8192, 5 * 8192))
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk = engine.ed25519_genpub_batch(dev(sk))
    sig = engine.ed25519_sign_batch(dev(sk), pk, dev(msgs)).cpu().numpy()
    pk = pk.cpu().numpy()
    nbad = int(rng.integers(0, 4))
    for i in rng.integers(0, n, nbad):
        which = int(rng.integers(0, 4))
        if which == 0: sig[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        elif which == 1: sig[i, 32 + int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        elif which == 2: pk[i] = rng.integers(0, 256, 32, dtype=np.uint8)
        else: msgs[i, 0] ^= 1
    want = oracle.verify_batch(sig, pk, msgs, 32)
    got, st = engine.ed25519_verify_batch_rlc(dev(sig), dev(pk), dev(msgs), msg_len=32, return_stats=True)
    assert np.array_equal(got.cpu().numpy(), want)
    groups = (n + 8191) // 8192
    assert st[2] + st[3] == groups and st[2] <= nbad and st[0] + st[1] == n


CHUNK_EDGE_SIZES = [65535, 65536, 65537, 98303, 98304, 98305, 131071, 131072, 131073, 196607, 196608, 196609, 393215, 393217,
                    458751, 458753, 600001]


@pytest.mark.parametrize("n", CHUNK_EDGE_SIZES)
def test_host_pipeline_at_chunk_boundaries(engine, oracle, n):
    """the host-pointer pipeline (host_pipe.c: chunks of 2^16 / 2^17 items doubling up to the stage size, three lanes, a short
    tail travelling with the last chunk) at sizes on both sides of every boundary of its schedule: same bytes as the
    device-pointer entry points on the same data (those are pinned against the reference elsewhere), the constructed
    verdict pattern, and the oracle on a sample that straddles the first chunk boundary"""
    rng = np.random.default_rng(n)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msg = rng.integers(0, 256, (n, 20), dtype=np.uint8)
    d_sk, d_msg = dev(sk), dev(msg)
    d_pk = engine.ed25519_genpub_batch(d_sk)
    d_sig = engine.ed25519_sign_batch(d_sk, d_pk, d_msg, msg_len=20)
    pk, sig = d_pk.cpu().numpy(), d_sig.cpu().numpy()
    assert np.array_equal(engine.ed25519_genpub_batch(sk), pk)                       # host path, chunked
    assert np.array_equal(engine.ed25519_sign_batch(sk, pk, msg, msg_len=20), sig)
    bad = sig.copy()
    flip = (np.arange(n) % 11 == 3) | (np.arange(n) == n - 1) | (np.arange(n) == 65536 % n)
    bad[flip, 41] ^= 8
    ok = engine.ed25519_verify_batch(bad, pk, msg, msg_len=20)
    assert np.array_equal(ok, (~flip).astype(np.uint8))
    assert np.array_equal(ok, engine.ed25519_verify_batch(dev(bad), d_pk, d_msg, msg_len=20).cpu().numpy())
    pt = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    xo = engine.x25519_batch(sk, pt)
    assert np.array_equal(xo, engine.x25519_batch(d_sk, dev(pt)).cpu().numpy())
    lo = max(0, min(n, 65536) - 40)
    hi = min(n, lo + 80)
    assert np.array_equal(xo[lo:hi], oracle.x25519_batch(sk[lo:hi], pt[lo:hi]))
    assert np.array_equal(ok[lo:hi], oracle.verify_batch(bad[lo:hi], pk[lo:hi], msg[lo:hi], 20))
    assert np.array_equal(sig[lo:hi], oracle.sign_batch(sk[lo:hi], pk[lo:hi], msg[lo:hi], 20))


LARGE_ROUTE_EDGES = [65535, 65536, 65537, 262143, 262144, 262145, 524287, 524288, 524289]


@pytest.mark.parametrize("n", LARGE_ROUTE_EDGES)
def test_large_route_boundaries_against_the_oracle(engine, oracle, n):
    """both sides of the thresholds above the small-pass routes (kernels.hip: edk_verify) - 2^16: the exact chain packs 16
    items to the wave; 2^18: three-lane preparation with the long loop in place -> prepare + halve + one-lane evaluation;
    2^19: pairs up to 2^138 / 35 windows -> 2^134 / 34 - on the config-2 mix with garbage keys on top (off-curve keys and
    items without a short pair on the exact path), EVERY verdict against the oracle, device- and host-pointer entry points"""
    import workload
    sk, msg = workload.sign_inputs(n, seed=77, config=2)
    pk = engine.ed25519_genpub_batch(dev(sk))
    sig = engine.ed25519_sign_batch(dev(sk), pk, dev(msg)).cpu().numpy()
    pk = pk.cpu().numpy()
    workload.corrupt_for_verify(sig, pk, msg, seed=77)
    rng = np.random.default_rng(n)
    g = rng.integers(0, n, n // 97)
    pk[g] = rng.integers(0, 256, (len(g), 32), dtype=np.uint8)            # garbage keys: about half are no curve point
    want = oracle.verify_batch(sig, pk, msg, 32)
    assert 0.8 < want.mean() < 0.95
    got_d = engine.ed25519_verify_batch(dev(sig), dev(pk), dev(msg), msg_len=32).cpu().numpy()
    assert np.array_equal(got_d, want)
    if n in (65536, 262145, 524288):
        assert np.array_equal(engine.ed25519_verify_batch(sig, pk, msg, msg_len=32), want)      # the host pipeline's chunks of the same batch


@pytest.mark.parametrize("n,garbage_every", [(200000, 50), (400000, 2), (1 << 20, 3)])
def test_host_calls_of_several_chunks_with_garbage_keys(engine, oracle, n, garbage_every):
    """host_pipe.c: multi-chunk verify calls (packed arrays, records, ragged messages with rebased offset tables) whose chunks
    each carry exact work - few garbage keys, and so many that a chunk's work list passes the 65 536 entries one launch of the
    chain takes: every verdict against the oracle and against the device-pointer entry point"""
    import workload
    sk, msg = workload.sign_inputs(n, seed=78, config=2)
    pk = engine.ed25519_genpub_batch(dev(sk))
    sig = engine.ed25519_sign_batch(dev(sk), pk, dev(msg)).cpu().numpy()
    pk = pk.cpu().numpy()
    rng = np.random.default_rng(n + garbage_every)
    g = np.arange(0, n, garbage_every)
    pk[g] = rng.integers(0, 256, (len(g), 32), dtype=np.uint8)
    sig[5::11, 3] ^= 2
    want = oracle.verify_batch(sig, pk, msg, 32)
    got_h = engine.ed25519_verify_batch(sig, pk, msg, msg_len=32)
    assert np.array_equal(got_h, want)
    assert np.array_equal(engine.ed25519_verify_batch(dev(sig), dev(pk), dev(msg), msg_len=32).cpu().numpy(), want)
    rec = np.concatenate([sig, pk, msg], axis=1)                          # the record form shares the pipeline
    assert np.array_equal(engine.ed25519_verify_records(rec, 0, 64, 96, 32), want)
    # ragged messages (chunk offset tables rebased) with the same keys
    lens = rng.integers(0, 40, n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)
    sig_r = engine.ed25519_sign_batch(dev(sk), dev(pk), dev(blob), msg_off=dev(off.astype(np.int64))).cpu().numpy()
    got_r = engine.ed25519_verify_batch(sig_r, pk, blob, msg_off=off)
    assert np.array_equal(got_r, engine.ed25519_verify_batch(dev(sig_r), dev(pk), dev(blob), msg_off=dev(off.astype(np.int64))).cpu().numpy())
    lo = 65536 - 50
    for i in range(lo, lo + 100):                                         # across the first chunk boundary, against the oracle
        assert got_r[i] == oracle.verify(sig_r[i].tobytes(), pk[i].tobytes(), blob[int(off[i]):int(off[i + 1])].tobytes()), i
