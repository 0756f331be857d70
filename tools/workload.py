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


P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
F_GARBAGE = 6

# the eight points of order dividing 8 (canonical encodings) and their orders
SMALL_ORDER = [
    ("0100000000000000000000000000000000000000000000000000000000000000", 1),
    ("ecffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f", 2),
    ("0000000000000000000000000000000000000000000000000000000000000000", 4),
    ("0000000000000000000000000000000000000000000000000000000000000080", 4),
    ("26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05", 8),
    ("26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc85", 8),
    ("c7176a703d4dd84fba3c0b760d10670f2a2053fa2c39ccc64ec7fd7792ac037a", 8),
    ("c7176a703d4dd84fba3c0b760d10670f2a2053fa2c39ccc64ec7fd7792ac03fa", 8),
]
EDGE_KINDS = 64          # edge case e of every 2^14-item block sits at item 2^14 m + 16 (4 e + 1) + 9
EDGE_BLOCK = 1 << 14


def edge_position(m, e):
    return EDGE_BLOCK * m + 16 * (4 * e + 1) + 9          # i % 16 == 9: never one of the corrupted items


def _le(x, n=32):
    return np.frombuffer(int(x).to_bytes(n, "little"), np.uint8)


def splice_edges(sig, pub, msg, expect, seed, config, first=0):
    """SURVEY 8(c)-3's edge vectors spliced at fixed indices of a config-2/4 batch: in every
    block of 2^14 items, 64 items (global index 2^14 m + 16 (4 e + 1) + 9, e = 0..63) are rewritten, each
    from the item's own genuine (sig, pub, msg): S + k l, S = 0 / l / 2^256-1, identity and non-canonical R,
    small-order and non-canonical A, off-curve A (y = 2..17, garbage keys), single flipped bits.  The
    expected verdicts follow from the reference's semantics (lib/ed25519-sha512.c:148-181, lib/ed.c:100-149:
    S reduced mod l unchecked, A decoded permissively, R compared as bytes) and are ASSERTED against the
    compiled reference by tools/gen_golden.py for every batch whose digest is pinned."""
    import hashlib
    n = sig.shape[0]
    ident = _le(1)
    for m in range(first // EDGE_BLOCK, (first + n + EDGE_BLOCK - 1) // EDGE_BLOCK):
        garbage = None
        for e in range(EDGE_KINDS):
            g = edge_position(m, e)
            if not first <= g < first + n:
                continue
            i = g - first
            S = int.from_bytes(sig[i, 32:].tobytes(), "little")

            def r_ident_s0(a_bytes, order):
                """R = identity, S = 0 under key A: accepted iff t A = identity, i.e. order | t"""
                sig[i, :32] = ident; sig[i, 32:] = 0; pub[i] = a_bytes
                if order is None:
                    return 0
                t = int.from_bytes(hashlib.sha512(sig[i, :32].tobytes() + pub[i].tobytes() + msg[i].tobytes()).digest(), "little") % L
                return int(t % order == 0)

            if e < 4:
                sig[i, 32:] = _le(S + (1, 2, 7, 14)[e] * L); v = 1       # S is not range-checked (sc.c:191-214)
            elif e == 4: sig[i, 32:] = 0; v = 0
            elif e == 5: sig[i, 32:] = _le(L); v = 0
            elif e == 6: sig[i, 32:] = 255; v = 0
            elif e == 7: sig[i, 32:] = _le((S - 1) % 2**256); v = 0
            elif e == 8: v = r_ident_s0(ident, 1)
            elif e == 9: v = r_ident_s0(ident, 1); sig[i, :32] = _le(P + 1); v = 0      # non-canonical R never matches
            elif e == 10: v = r_ident_s0(_le(1 | 1 << 255), 1)                           # x = 0 with the sign bit: accepted
            elif e < 19:
                enc, order = SMALL_ORDER[e - 11]
                v = r_ident_s0(np.frombuffer(bytes.fromhex(enc), np.uint8), order)
            elif e < 23:                                                  # y = p (order 4) and y = p + 1 (identity), both signs
                k, sign = (e - 19) // 2, (e - 19) % 2
                v = r_ident_s0(_le((P + k) | sign << 255), 4 if k == 0 else 1)
            elif e < 27: v = r_ident_s0(_le(P + (3, 6, 9, 18)[e - 23]), None)
            elif e < 43: pub[i] = _le(e - 27 + 2); v = 0                  # y = 2..17, genuine-looking signature
            elif e < 47: v = r_ident_s0(_le(e - 43 + 2), None)
            elif e < 55:
                buf, bit = ((sig, 0), (sig, 255), (sig, 256), (sig, 511), (pub, 255), (pub, 0), (msg, 0), (msg, 255))[e - 47]
                buf[i, bit // 8] ^= 1 << (bit % 8); v = 0
            else:
                if garbage is None:
                    garbage = field_bytes(seed, config, F_GARBAGE, EDGE_KINDS, 32, first=m * EDGE_KINDS)
                pub[i] = garbage[e]; v = 0
            expect[i] = v
    return expect


def corrupt_for_verify(sig, pub, msg, seed=1, config=2, first=0, edges=True):
    """config 2/4: items with global index i % 16 == 5 get one flipped bit in R, S, A or the message
    (round-robin over the corrupted items), then the edge vectors are spliced in (splice_edges);
    returns the expected verdicts."""
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
    return splice_edges(sig, pub, msg, expect, seed, config, first) if edges else expect


# ---------------------------------------------------------------------------------------------
# mixed-order inputs at scale: the inputs on which "v = u t mod 8 l" (csrc/halve.h) and the scalar
# mod 8 l of csrc/rlc.hip decide the verdict byte.  Test-data construction only: plain affine Edwards arithmetic on
# Python integers; the expected verdicts come from the oracle.
# ---------------------------------------------------------------------------------------------
_D = (-121665 * pow(121666, P - 2, P)) % P


def _xrecover(y, sign):
    x2 = (y * y - 1) * pow(_D * y * y + 1, P - 2, P) % P
    x = pow(x2, (P + 3) // 8, P)
    if (x * x - x2) % P:
        x = x * pow(2, (P - 1) // 4, P) % P
    assert (x * x - x2) % P == 0
    return P - x if (x & 1) != sign else x


def _dec(enc):
    v = int.from_bytes(enc, "little")
    y, sign = v & (2**255 - 1), v >> 255
    return (_xrecover(y, sign) if y not in (1, P - 1) else 0, y)


def _add(p1, p2):
    (x1, y1), (x2, y2) = p1, p2
    k = _D * x1 * x2 * y1 * y2 % P
    return ((x1 * y2 + x2 * y1) * pow(1 + k, P - 2, P) % P, (y1 * y2 + x1 * x2) * pow(1 - k, P - 2, P) % P)


def _enc(pt):
    return int(pt[1] | ((pt[0] & 1) << 255)).to_bytes(32, "little")


def add_torsion(sk, pk, sig, msg, seed):
    """In a third of the items (i % 3 == seed % 3) a random element of the 8-torsion subgroup - the neutral element
    included - is added to the key A, to the commitment R, or to both (by item), and S is what the honest signer computes for
    THAT key and commitment: S = r + H(R' || A' || M) a with r, a from the secret key as lib/ed25519-sha512.c:84-123 takes
    them.  The prime-order parts are genuine, so S B - t A' - R' = -(t T_A + T_R): the reference accepts exactly when
    t T_A + T_R = 0.  Returns (sig', pk', touched) - copies; `touched` marks the rewritten items."""
    import hashlib
    rng = np.random.default_rng(seed)
    tors = [_dec(bytes.fromhex(h)) for h, _ in SMALL_ORDER]
    n = sk.shape[0]
    sig2, pk2 = sig.copy(), pk.copy()
    touched = np.zeros(n, bool)
    for i in range(seed % 3, n, 3):
        kind = int(rng.integers(0, 3))                     # 0: A, 1: R, 2: both
        ta, tr = tors[int(rng.integers(0, 8))], tors[int(rng.integers(0, 8))]
        a_enc, r_enc, m = pk[i].tobytes(), sig[i, :32].tobytes(), msg[i].tobytes()
        if kind != 1:
            a_enc = _enc(_add(_dec(a_enc), ta))
        if kind != 0:
            r_enc = _enc(_add(_dec(r_enc), tr))
        h = hashlib.sha512(sk[i].tobytes()).digest()
        a = int.from_bytes(h[:32], "little") & ((1 << 254) - 8) | (1 << 254)
        r = int.from_bytes(hashlib.sha512(h[32:] + m).digest(), "little") % L
        t = int.from_bytes(hashlib.sha512(r_enc + a_enc + m).digest(), "little") % L
        pk2[i] = np.frombuffer(a_enc, np.uint8)
        sig2[i, :32] = np.frombuffer(r_enc, np.uint8)
        sig2[i, 32:] = np.frombuffer(((r + t * a) % L).to_bytes(32, "little"), np.uint8)
        touched[i] = True
    return sig2, pk2, touched
