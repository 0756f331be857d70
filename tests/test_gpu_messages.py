"""Messages beyond the golden table's 1023 bytes, and offset tables that lie.

The reference streams any length through its compression function (lib/sha512.c:143-210; its own selftest hashes 16 KiB
buffers, test/selftest-sha512.c:11-41); the device hashes a message block by block in one lane (csrc/sha512.h).  Sign and
verify over 4 KiB, 16 KiB, 64 KiB and 1 MiB messages - fixed-length and ragged layouts, host-pointer and device-pointer
paths - against the oracle, bit-exact, with hashlib's SHA-512 as a second, independent check of what the oracle hashed.
Then the offset table: the host-pointer entry points refuse one that decreases; the device-pointer kernels clamp."""
import ctypes
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L = 2**252 + 27742317777372353535851937790883648493


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ragged(msgs):
    off = np.zeros(len(msgs) + 1, np.uint64)
    off[1:] = np.cumsum([len(m) for m in msgs])
    return np.frombuffer(b"".join(msgs), np.uint8).copy(), off


def challenge(sig, pk, msg):
    """t = SHA-512(R || A || M) mod l by hashlib: what ed25519-sha512.c:166-171 feeds the scalar multiplication"""
    return int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % L


@pytest.mark.parametrize("mlen,n", [(4096, 96), (16384, 80), (65536, 70), (1 << 20, 6)])
def test_long_messages_fixed_length(engine, oracle, mlen, n):
    rng = np.random.default_rng(mlen)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msg = rng.integers(0, 256, (n, mlen), dtype=np.uint8)
    pk = oracle.genpub_batch(sk)
    sig = oracle.sign_batch(sk, pk, msg, mlen)
    # the oracle's own hash against hashlib on the whole stream (an error in either would show as a wrong S below)
    assert oracle.sha512(sig[0, :32].tobytes() + pk[0].tobytes() + msg[0].tobytes()) == \
        hashlib.sha512(sig[0, :32].tobytes() + pk[0].tobytes() + msg[0].tobytes()).digest()
    assert np.array_equal(engine.ed25519_sign_batch(sk, pk, msg, msg_len=mlen), sig)                       # host pointers
    assert np.array_equal(engine.ed25519_sign_batch(dev(sk), dev(pk), dev(msg), msg_len=mlen).cpu().numpy(), sig)
    bad_msg, bad_sig = msg.copy(), sig.copy()
    bad_msg[1::3, -1] ^= 0x40                      # the LAST byte of a long message: only a hash of all of it notices
    bad_msg[2::3, mlen // 2 + 5] ^= 1
    want = oracle.verify_batch(sig, pk, bad_msg, mlen)
    assert want[0::3].all() and not want[1::3].any() and not want[2::3].any()
    assert np.array_equal(engine.ed25519_verify_batch(sig, pk, bad_msg, msg_len=mlen), want)
    assert np.array_equal(engine.ed25519_verify_batch(dev(sig), dev(pk), dev(bad_msg), msg_len=mlen).cpu().numpy(), want)
    assert engine.ed25519_verify_batch(dev(sig), dev(pk), dev(msg), msg_len=mlen).cpu().numpy().all()
    # S = r + t a with t from hashlib: R and S of the device's signature are consistent with an independent SHA-512
    # (S B = R + t A is what verify checks; here t itself is recomputed outside the oracle and the product)
    for i in (0, n - 1):
        t = challenge(sig[i].tobytes(), pk[i].tobytes(), msg[i].tobytes())
        h = hashlib.sha512(sk[i].tobytes()).digest()
        a = int.from_bytes(h[:32], "little") & ~7 & ~(1 << 255) | (1 << 254)
        r = int.from_bytes(hashlib.sha512(h[32:] + msg[i].tobytes()).digest(), "little") % L
        assert int.from_bytes(sig[i, 32:].tobytes(), "little") == (r + t * a) % L


def test_long_messages_ragged(engine, oracle):
    """lengths around every size of the list above in ONE ragged batch (so one wave carries 3-byte and 1 MiB messages side
    by side), offsets that leave the messages unaligned, sign and verify, host and device paths"""
    import torch
    rng = np.random.default_rng(77)
    lens = [0, 3, 1023, 1024, 1025, 4095, 4096, 4097, 16384 - 17, 16384, 16384 + 111, 65536 - 1, 65536, 65536 + 112, 3, 32,
            (1 << 20) - 129, 1 << 20, (1 << 20) + 1, 200000, 131072 + 64, 7, 128 * 9 - 17, 128 * 9 - 16] * 2
    n = len(lens)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = [bytes(rng.integers(0, 256, k, dtype=np.uint8)) for k in lens]
    pk = oracle.genpub_batch(sk)
    sig = np.frombuffer(b"".join(oracle.sign(sk[i].tobytes(), pk[i].tobytes(), msgs[i]) for i in range(n)), np.uint8).reshape(n, 64).copy()
    for i in (8, 17):
        assert oracle.sha512(msgs[i]) == hashlib.sha512(msgs[i]).digest()
    blob, off = ragged(msgs)
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    assert np.array_equal(engine.ed25519_sign_batch(sk, pk, blob, msg_off=off), sig)
    assert np.array_equal(engine.ed25519_sign_batch(dev(sk), dev(pk), dev(blob), msg_off=d_off).cpu().numpy(), sig)
    bad = blob.copy()
    flipped = [i for i in range(n) if i % 2 and lens[i]]
    for i in flipped:
        bad[int(off[i + 1]) - 1] ^= 0x80           # the last byte of the message
    want = np.array([oracle.verify(sig[i].tobytes(), pk[i].tobytes(), bytes(bad[int(off[i]):int(off[i + 1])])) for i in range(n)], np.uint8)
    assert not want[flipped].any() and want.sum() == n - len(flipped)
    assert np.array_equal(engine.ed25519_verify_batch(sig, pk, bad, msg_off=off), want)
    assert np.array_equal(engine.ed25519_verify_batch(dev(sig), dev(pk), dev(bad), msg_off=d_off).cpu().numpy(), want)
    assert np.array_equal(engine.ed25519_verify_batch_rlc(dev(sig), dev(pk), dev(bad), msg_off=d_off).cpu().numpy(), want)
    # the eddsa.h single-item functions (lib/eddsa.h:47-52) on a 1 MiB message
    i = lens.index(1 << 20)
    assert engine.ed25519_sign(sk[i].tobytes(), pk[i].tobytes(), msgs[i]) == sig[i].tobytes()
    assert engine.ed25519_verify(sig[i].tobytes(), pk[i].tobytes(), msgs[i]) is True
    assert engine.ed25519_verify(sig[i].tobytes(), pk[i].tobytes(), msgs[i][:-1] + bytes([msgs[i][-1] ^ 1])) is False


@pytest.mark.parametrize("n", [4095, 4096, 70000, 300000])
def test_ragged_batches_are_hashed_in_order_of_length(engine, oracle, n):
    """passes of 4096 items or more with an offset table take their items through a permutation sorted by message length
    (csrc/kernels.hip: msg_order - a wave ends with its longest message); the verdicts and signatures must not know: lengths from
    0 to 3000 bytes with a few of 40 000, one length repeated a thousand times, on the routes of 4095 (no permutation), 4096, 70 000
    and 300 000 items, verify and sign and the batch verification (per-item routing and the combination), device pointers and - through the chunked pipeline - host
    pointers"""
    import torch
    rng = np.random.default_rng(n)
    m = 1500
    lens = rng.integers(0, 3001, m)
    lens[::97] = 40000
    lens[5:m:3] = 77
    sk = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    msgs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in lens]
    pk = oracle.genpub_batch(sk)
    sig = np.frombuffer(b"".join(oracle.sign(sk[i].tobytes(), pk[i].tobytes(), msgs[i]) for i in range(m)), np.uint8).reshape(m, 64).copy()
    idx = rng.integers(0, m, n)                                  # the batch: n draws from the m signed items, in random order
    blob, off = ragged([msgs[i] for i in idx])
    sk_n, pk_n, sig_n = sk[idx], pk[idx], sig[idx].copy()
    want = np.ones(n, np.uint8)
    bad = rng.permutation(n)[: n // 7]
    sig_n[bad, 33] ^= 0x20
    want[bad] = 0
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    got = engine.ed25519_verify_batch(dev(sig_n), dev(pk_n), dev(blob), msg_off=d_off).cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(engine.ed25519_verify_batch_rlc(dev(sig_n), dev(pk_n), dev(blob), msg_off=d_off).cpu().numpy(), want)
    # the combination itself (csrc/rlc.hip: k_rlc_hash takes the same permutation; calls of this size go to the per-item
    # kernels unless told otherwise): the genuine batch is accepted whole by it, the damaged one falls back group by group
    engine.set_rlc_min_items(0)
    try:
        ok, st = engine.ed25519_verify_batch_rlc(dev(sig[idx]), dev(pk_n), dev(blob), msg_off=d_off, return_stats=True)
        assert ok.cpu().numpy().all() and st[0] == n and st[2] == 0
        assert np.array_equal(engine.ed25519_verify_batch_rlc(dev(sig_n), dev(pk_n), dev(blob), msg_off=d_off).cpu().numpy(), want)
    finally:
        engine.set_rlc_min_items(engine.RLC_MIN_ITEMS_DEFAULT)
    out = engine.ed25519_sign_batch(dev(sk_n), dev(pk_n), dev(blob), msg_off=d_off).cpu().numpy()
    assert np.array_equal(out, sig[idx])
    if n <= 70000:
        assert np.array_equal(engine.ed25519_verify_batch(sig_n, pk_n, blob, msg_off=off), want)
        assert np.array_equal(engine.ed25519_sign_batch(sk_n, pk_n, blob, msg_off=off), sig[idx])


def test_device_sha512_of_long_messages_against_hashlib(engine):
    """the SHA-512 layer alone (eddsa_amd_debug_layer) on 4 KiB .. 1 MiB + 17 bytes: every block boundary case of the
    padding (lib/sha512.c:176-210) at sizes the golden layer vectors (0..299 bytes) do not reach"""
    engine.debug_init(0, True)
    try:
        rng = np.random.default_rng(5)
        for base in (4096, 16384, 65536, 1 << 20):
            lens = [base - 17, base - 16, base - 1, base, base + 1, base + 111, base + 112, base + 17]
            width = max(lens)
            items = []
            for k in lens:
                m = bytes(rng.integers(0, 256, k, dtype=np.uint8))
                items.append((m, k.to_bytes(8, "little") + m + bytes(width - k)))
            got = engine.debug_layer("sha512", [it[1] for it in items], 64)
            for (m, _), g in zip(items, got):
                assert g == hashlib.sha512(m).digest(), len(m)
    finally:
        engine.debug_init(0, False)


def raw_verify(engine, ok, sig, pk, blob, off, n):
    """the C entry point itself: libeddsa_amd/api.py validates the table before it calls, the library must not depend on it"""
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    return engine.library().ed25519_verify_batch(P(ok), P(sig), P(pk), P(blob), P(off), ctypes.c_size_t(0), ctypes.c_size_t(n))


def make_batch(oracle, n, seed):
    rng = np.random.default_rng(seed)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(1, 90, n)]
    pk = oracle.genpub_batch(sk)
    sig = np.frombuffer(b"".join(oracle.sign(sk[i].tobytes(), pk[i].tobytes(), msgs[i]) for i in range(n)), np.uint8).reshape(n, 64).copy()
    blob, off = ragged(msgs)
    return sk, pk, sig, blob, off


@pytest.mark.parametrize("n", [300, 70000, 200000])
def test_host_calls_refuse_an_offset_table_that_decreases(engine, oracle, n):
    """lengths come from msg_off[i + 1] - msg_off[i]: a table that runs backwards would make that 2^64 - something.  The
    host-pointer entry points walk each chunk's table before they touch the chunk (csrc/host_pipe.c: offsets_ok):
    hipErrorInvalidValue (1), whichever chunk holds the bad entry (n = 70000 and 200000 are calls of several chunks), and
    the engine works afterwards"""
    m = min(n, 600)
    sk, pk, sig, blob, off = make_batch(oracle, m, n)
    reps = (n + m - 1) // m
    pk_n, sig_n, sk_n = (np.tile(a, (reps, 1))[:n].copy() for a in (pk, sig, sk))
    lens = np.tile(np.diff(off.astype(np.int64)), reps)[:n]
    off_n = np.zeros(n + 1, np.uint64); off_n[1:] = np.cumsum(lens)
    blob_n = np.tile(blob, reps)[:int(off_n[-1])].copy()
    ok = np.zeros(n, np.uint8)
    assert raw_verify(engine, ok, sig_n, pk_n, blob_n, off_n, n) == 0 and ok.all()
    for spot in (1, n // 2, n - 1):
        bad = off_n.copy()
        bad[spot] = bad[spot + 1] + 5
        assert raw_verify(engine, ok, sig_n, pk_n, blob_n, bad, n) == -1, spot
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        out = np.zeros((n, 64), np.uint8)
        assert engine.library().ed25519_sign_batch(P(out), P(sk_n), P(pk_n), P(blob_n), P(bad), ctypes.c_size_t(0), ctypes.c_size_t(n)) == -1
    huge = off_n.copy(); huge[n] = 1 << 60
    assert raw_verify(engine, ok, sig_n, pk_n, blob_n, huge, n) == -1
    # the 2^46-byte limit is the whole table's: one whose chunks each stay below it is refused all the same, before anything is
    # touched (its messages do not exist)
    steep = (np.arange(n + 1, dtype=np.uint64) * np.uint64((1 << 47) // n))
    assert raw_verify(engine, ok, sig_n, pk_n, blob_n, steep, n) == -1
    ok[:] = 0
    assert raw_verify(engine, ok, sig_n, pk_n, blob_n, off_n, n) == 0 and ok.all()
    assert engine.secret_residue()[0] == 0


@pytest.mark.parametrize("n", [200, 40000, 300000])
def test_device_calls_clamp_an_offset_table_that_lies(engine, oracle, n):
    """a table in HBM is not inspected (that would be a pass of its own): the kernels clamp every span into
    [0, msg_off[n]) and to a length >= 0 (csrc/lanes.h: msg_span), on every route (n = 200: four lanes per item, 40000: the
    three-lane preparation, 300000: one lane per item), in verify, sign and the batch verification.  Entries that run
    backwards or point far outside therefore cost the items next to them their verdict and nothing else: no fault, and
    every item whose own two entries are intact is decided as the oracle decides it"""
    import torch
    m = min(n, 500)
    sk, pk, sig, blob, off = make_batch(oracle, m, n + 1)
    reps = (n + m - 1) // m
    pk_n, sig_n, sk_n = (np.tile(a, (reps, 1))[:n].copy() for a in (pk, sig, sk))
    lens = np.tile(np.diff(off.astype(np.int64)), reps)[:n]
    off_n = np.zeros(n + 1, np.int64); off_n[1:] = np.cumsum(lens)
    blob_n = np.tile(blob, reps)[:int(off_n[-1])].copy()
    bad = off_n.copy()
    spots = np.arange(7, n - 1, max(13, n // 50))
    bad[spots[0::3]] = bad[spots[0::3] + 1] + 9           # runs backwards
    bad[spots[1::3]] = (1 << 62) + 12345                  # far outside the buffer
    bad[spots[2::3]] = -8                                 # 2^64 - 8 as the kernels read it
    touched = np.zeros(n, bool)
    touched[spots] = True; touched[spots - 1] = True      # item k uses entries k and k + 1
    d = dict(sig=dev(sig_n), pk=dev(pk_n), sk=dev(sk_n), blob=dev(blob_n), off=torch.from_numpy(bad).cuda())
    ok = engine.ed25519_verify_batch(d["sig"], d["pk"], d["blob"], msg_off=d["off"]).cpu().numpy()
    assert ok[~touched].all() and not ok[touched].all()
    ok = engine.ed25519_verify_batch_rlc(d["sig"], d["pk"], d["blob"], msg_off=d["off"]).cpu().numpy()
    assert ok[~touched].all()
    engine.set_rlc_min_items(0)                            # ... and through the combination itself (its own hashing kernel)
    try:
        ok = engine.ed25519_verify_batch_rlc(d["sig"], d["pk"], d["blob"], msg_off=d["off"]).cpu().numpy()
    finally:
        engine.set_rlc_min_items(engine.RLC_MIN_ITEMS_DEFAULT)
    assert ok[~touched].all() and not ok[touched].all()
    out = engine.ed25519_sign_batch(d["sk"], d["pk"], d["blob"], msg_off=d["off"]).cpu().numpy()
    assert np.array_equal(out[~touched], sig_n[~touched])
    torch.cuda.synchronize()
    # a clamped span is still a span: the item is hashed over exactly msgs[lo', hi') - pinned on one entry of each kind
    total = int(off_n[-1])
    for k in (int(spots[0]), int(spots[1]), int(spots[2])):
        for item in (k - 1, k):
            lo, hi = int(bad[item]) % 2**64, int(bad[item + 1]) % 2**64
            lo = min(lo, total); hi = min(max(hi, lo), total)
            want = oracle.sign(sk_n[item].tobytes(), pk_n[item].tobytes(), bytes(blob_n[lo:hi]))
            assert out[item].tobytes() == want, (k, item)
