"""Deterministic synthetic workloads shared by bench.py, the tests and the golden
generator.  PRNG = SplitMix64 over a counter, keyed by (seed, config, field); little-endian
output, so every byte of every batch is reproducible anywhere from the numbers below alone."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def field_bytes(seed, config, field, n, width, first=0):
    """(n, width) uint8: bytes of items first..first+n-1 of the stream (seed, config, field)."""
    assert width % 8 == 0
    words = width // 8
    key = _splitmix64(np.uint64((seed << 16) ^ (config << 8) ^ field))
    with np.errstate(over="ignore"):
        idx = np.arange(first * words, (first + n) * words, dtype=np.uint64)
        out = _splitmix64(key * np.uint64(0x2545F4914F6CDD1D) + idx)
    return out.view(np.uint8).reshape(n, width).copy()


# field ids
F_SK, F_MSG, F_SCALAR, F_POINT, F_BITPOS = 1, 2, 3, 4, 5


def x25519_inputs(n, seed=2, first=0):
    """config 3: unclamped scalars, points with bit 255 uniformly random."""
    return field_bytes(seed, 3, F_SCALAR, n, 32, first), field_bytes(seed, 3, F_POINT, n, 32, first)


def sign_inputs(n, seed=4, config=5, first=0):
    """config 5: secret keys and 32-byte messages (public keys come from genpub)."""
    return field_bytes(seed, config, F_SK, n, 32, first), field_bytes(seed, config, F_MSG, n, 32, first)


def corrupt_for_verify(sig, pub, msg, seed=1, config=2, first=0):
    """config 2/4: items with global index i % 16 == 5 get one flipped bit in R, S, A or the message
    (round-robin over the corrupted items); returns the expected verdicts."""
    n = sig.shape[0]
    gidx = np.arange(first, first + n)
    bad = np.nonzero(gidx % 16 == 5)[0]
    which = (gidx[bad] // 16) % 4
    bitpos = field_bytes(seed, config, F_BITPOS, n, 8, first)[bad, 0].astype(np.int64)
    byte, bit = bitpos // 8, (1 << (bitpos % 8)).astype(np.uint8)
    for w, (buf, off) in enumerate(((sig, 0), (sig, 32), (pub, 0), (msg, 0))):
        sel = which == w
        buf[bad[sel], off + byte[sel]] ^= bit[sel]
    expect = np.ones(n, np.uint8)
    expect[bad] = 0
    return expect
