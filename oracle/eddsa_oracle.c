/*
 * eddsa_oracle.c - CPU restatement of the libeddsa hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity checker for the HIP engine.  It is never linked into, loaded by or
 * called from the product library; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it (see eddsa_oracle.h).
 *
 * It restates, function by function, what the reference computes (file:line of the reference
 * given at every definition, relative to the reference checkout).  Representation is its own
 * (unsigned 5x51-bit field limbs, 64-bit-word scalars, Barrett with b = 2^64), which is allowed
 * because every decision and every output of the reference is taken on fully reduced values
 * (fld_reduce before each parity/equality/export), so outputs are functions of the input bytes
 * only.  Where the ORDER of group operations matters for the output (inputs that decode to a
 * pair (x,y) that is not on the curve), the same formulas are applied in the same order as the
 * reference, including the limb-boundary behaviour of its joint-sparse-form recoder.
 *
 * PARITY PINNED: checked against the compiled reference (oracle/_ref) function by function in
 * tests/test_oracle_vs_ref.py and against the committed golden vectors in
 * tests/test_oracle_golden.py.
 */
#include "eddsa_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t fe[5];             /* value = sum fe[i] * 2^(51 i)  (mod 2^255-19) */

#define MASK51 ((UINT64_C(1) << 51) - 1)

/* ------------------------------------------------------------------------------------------
 * GF(2^255-19)
 * ---------------------------------------------------------------------------------------- */

static void fe_copy(fe h, const fe f) { memcpy(h, f, sizeof(fe)); }

static void fe_set_small(fe h, uint64_t v)      /* fld.h:71 fld_set0 */
{
    h[0] = v; h[1] = h[2] = h[3] = h[4] = 0;
}

/* one full carry sweep 0->1->2->3->4->(x19)->0->1; keeps the residue, shrinks the limbs */
static void fe_carry(fe h)
{
    uint64_t c;
    for (int i = 0; i < 4; i++) { c = h[i] >> 51; h[i] &= MASK51; h[i + 1] += c; }
    c = h[4] >> 51; h[4] &= MASK51; h[0] += 19 * c;
    c = h[0] >> 51; h[0] &= MASK51; h[1] += c;
}

/* fld.c:136-156 fld_import: 256-bit little-endian; bit 255 is NOT dropped, it is folded in as
 * 2^255 = 19 (mod p). */
static void fe_frombytes(fe h, const uint8_t s[32])
{
    uint64_t w[4];
    for (int i = 0; i < 4; i++) {
        w[i] = 0;
        for (int j = 7; j >= 0; j--) w[i] = (w[i] << 8) | s[8 * i + j];
    }
    h[0] = w[0] & MASK51;
    h[1] = ((w[0] >> 51) | (w[1] << 13)) & MASK51;
    h[2] = ((w[1] >> 38) | (w[2] << 26)) & MASK51;
    h[3] = ((w[2] >> 25) | (w[3] << 39)) & MASK51;
    h[4] = (w[3] >> 12) & MASK51;
    h[0] += 19 * (w[3] >> 63);
}

/* fld.c:53-130 fld_reduce: the unique representative in [0,p) */
static void fe_canon(fe t, const fe f)
{
    fe_copy(t, f);
    fe_carry(t); fe_carry(t); fe_carry(t);
    /* t < 2^255 now; q = 1 iff t >= p */
    uint64_t q = (t[0] + 19) >> 51;
    for (int i = 1; i < 5; i++) q = (t[i] + q) >> 51;
    t[0] += 19 * q;
    uint64_t c;
    for (int i = 0; i < 4; i++) { c = t[i] >> 51; t[i] &= MASK51; t[i + 1] += c; }
    t[4] &= MASK51;
}

/* fld.c:162-178 fld_export */
static void fe_tobytes(uint8_t s[32], const fe f)
{
    fe t;
    uint64_t w[4];
    fe_canon(t, f);
    w[0] = t[0] | (t[1] << 51);
    w[1] = (t[1] >> 13) | (t[2] << 38);
    w[2] = (t[2] >> 26) | (t[3] << 25);
    w[3] = (t[3] >> 39) | (t[4] << 12);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) s[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}

/* fld.h:84-95 fld_add / fld_sub (the reference leaves limbs uncarried and signed; here the
 * difference is biased by 8p and both results are carried, same residue) */
static void fe_add(fe h, const fe f, const fe g)
{
    for (int i = 0; i < 5; i++) h[i] = f[i] + g[i];
    fe_carry(h);
}

static void fe_sub(fe h, const fe f, const fe g)
{
    static const uint64_t p8[5] = {
        (UINT64_C(1) << 54) - 152, (UINT64_C(1) << 54) - 8, (UINT64_C(1) << 54) - 8,
        (UINT64_C(1) << 54) - 8, (UINT64_C(1) << 54) - 8 };
    for (int i = 0; i < 5; i++) h[i] = f[i] + p8[i] - g[i];
    fe_carry(h);
}

static void fe_neg(fe h, const fe f)             /* fld.h:136 fld_neg */
{
    fe z; fe_set_small(z, 0); fe_sub(h, z, f);
}

/* fld.c:209-244 fld_mul */
static void fe_mul(fe h, const fe f, const fe g)
{
    u128 c[5];
    uint64_t g19[5];
    for (int i = 1; i < 5; i++) g19[i] = 19 * g[i];
    for (int k = 0; k < 5; k++) {
        c[k] = 0;
        for (int i = 0; i < 5; i++) {
            int j = k - i;
            c[k] += (j >= 0) ? (u128)f[i] * g[j] : (u128)f[i] * g19[j + 5];
        }
    }
    for (int k = 0; k < 4; k++) { c[k + 1] += c[k] >> 51; c[k] &= MASK51; }
    uint64_t top = (uint64_t)(c[4] >> 51); c[4] &= MASK51;
    for (int k = 0; k < 5; k++) h[k] = (uint64_t)c[k];
    h[0] += 19 * top;
    uint64_t cc = h[0] >> 51; h[0] &= MASK51; h[1] += cc;
}

static void fe_sq(fe h, const fe f) { fe_mul(h, f, f); }   /* fld.c:249-280 fld_sq */

static void fe_sqn(fe h, const fe f, int n)
{
    fe_sq(h, f);
    for (int i = 1; i < n; i++) fe_sq(h, h);
}

/* fld.c:183-204 fld_scale (only ever called with s = 121665) */
static void fe_mul_small(fe h, const fe f, uint64_t s)
{
    u128 c = 0;
    for (int i = 0; i < 5; i++) { c += (u128)f[i] * s; h[i] = (uint64_t)c & MASK51; c >>= 51; }
    h[0] += 19 * (uint64_t)c;
}

/* the common prefix of both nacl exponent chains: returns z^(2^250 - 1) and z^11, z
 * (fld.c:593-637 and fld.c:670-704 walk the same ladder 5,10,20,40,50,100,200,250) */
static void fe_pow_2_250_m1(fe out, fe z11, const fe z)
{
    fe z2, z9, a5, a10, a20, a50, a100, t;
    fe_sq(z2, z);                       /* 2 */
    fe_sqn(t, z2, 2);                   /* 8 */
    fe_mul(z9, t, z);                   /* 9 */
    fe_mul(z11, z9, z2);                /* 11 */
    fe_sq(t, z11);                      /* 22 */
    fe_mul(a5, t, z9);                  /* 2^5 - 1 */
    fe_sqn(t, a5, 5);   fe_mul(a10, t, a5);
    fe_sqn(t, a10, 10); fe_mul(a20, t, a10);
    fe_sqn(t, a20, 20); fe_mul(t, t, a20);       /* 2^40 - 1 */
    fe_sqn(t, t, 10);   fe_mul(a50, t, a10);
    fe_sqn(t, a50, 50); fe_mul(a100, t, a50);
    fe_sqn(t, a100, 100); fe_mul(t, t, a100);    /* 2^200 - 1 */
    fe_sqn(t, t, 50);   fe_mul(out, t, a50);     /* 2^250 - 1 */
}

/* fld.c:578-645 fld_inv: z^(p-2) = z^(2^255-21); inv(0) = 0 */
static void fe_inv(fe h, const fe z)
{
    fe t, z11;
    fe_pow_2_250_m1(t, z11, z);
    fe_sqn(t, t, 5);                    /* 2^255 - 2^5 */
    fe_mul(h, t, z11);
}

/* fld.c:657-709 fld_pow2523: z^((p-5)/8) = z^(2^252-3) */
static void fe_pow2523(fe h, const fe z)
{
    fe t, z11;
    fe_pow_2_250_m1(t, z11, z);
    fe_sqn(t, t, 2);                    /* 2^252 - 4 */
    fe_mul(h, t, z);
}

/* fld.c:546-568 fld_eq */
static int fe_eq(const fe a, const fe b)
{
    uint8_t x[32], y[32];
    fe_tobytes(x, a); fe_tobytes(y, b);
    return memcmp(x, y, 32) == 0;
}

static int fe_parity(const fe a)
{
    fe t; fe_canon(t, a); return (int)(t[0] & 1);
}

/* constants of fld.c:23-41, as canonical little-endian bytes */
static const uint8_t BYTES_D[32] = {
    0xa3,0x78,0x59,0x13,0xca,0x4d,0xeb,0x75,0xab,0xd8,0x41,0x41,0x4d,0x0a,0x70,0x00,
    0x98,0xe8,0x79,0x77,0x79,0x40,0xc7,0x8c,0x73,0xfe,0x6f,0x2b,0xee,0x6c,0x03,0x52 };
static const uint8_t BYTES_J[32] = {     /* sqrt(-1) = 2^((p-1)/4) */
    0xb0,0xa0,0x0e,0x4a,0x27,0x1b,0xee,0xc4,0x78,0xe4,0x2f,0xad,0x06,0x18,0x43,0x2f,
    0xa7,0xd7,0xfb,0x3d,0x99,0x00,0x4d,0x2b,0x0b,0xdf,0xc1,0x4f,0x80,0x24,0x83,0x2b };
static const uint8_t BYTES_BY[32] = {    /* y(B) = 4/5, sign bit clear */
    0x58,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,
    0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66,0x66 };

static fe K_D, K_2D, K_M2D, K_J;

/* ------------------------------------------------------------------------------------------
 * Z / l Z,  l = 2^252 + 27742317777372353535851937790883648493
 * ---------------------------------------------------------------------------------------- */

typedef uint64_t sc[4];             /* reduced scalar, little-endian 64-bit words */

static const uint64_t L_WORDS[4] = {
    UINT64_C(0x5812631a5cf5d3ed), UINT64_C(0x14def9dea2f79cd6), 0, UINT64_C(0x1000000000000000) };
/* mu = floor(2^512 / l), 5 words */
static const uint64_t MU_WORDS[5] = {
    UINT64_C(0xed9ce5a30a2c131b), UINT64_C(0x2106215d086329a7), UINT64_C(0xffffffffffffffeb),
    UINT64_C(0xffffffffffffffff), UINT64_C(0xf) };

/* r[0..na+nb) = a * b */
static void bn_mul(uint64_t *r, const uint64_t *a, int na, const uint64_t *b, int nb)
{
    memset(r, 0, sizeof(uint64_t) * (size_t)(na + nb));
    for (int i = 0; i < na; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < nb; j++) {
            u128 t = (u128)a[i] * b[j] + r[i + j] + carry;
            r[i + j] = (uint64_t)t; carry = (uint64_t)(t >> 64);
        }
        r[i + nb] = carry;
    }
}

/* r = a - b over n words, returns the borrow */
static uint64_t bn_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, int n)
{
    uint64_t borrow = 0;
    for (int i = 0; i < n; i++) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)t; borrow = (uint64_t)(t >> 64) & 1;
    }
    return borrow;
}

/* sc.c:79-158 sc_barrett (HAC 14.42), here with b = 2^64, k = 4: x < 2^512 -> x mod l */
static void sc_barrett(sc r, const uint64_t x[8])
{
    uint64_t q2[10], r2[10], r1[5], lw[5], t[5];
    bn_mul(q2, x + 3, 5, MU_WORDS, 5);              /* q1 = floor(x / b^(k-1)); q2 = q1*mu */
    bn_mul(r2, q2 + 5, 5, L_WORDS, 4);              /* q3 = floor(q2 / b^(k+1)); q3*l */
    bn_sub(r1, x, r2, 5);                           /* (x - q3*l) mod b^(k+1) */
    memcpy(lw, L_WORDS, 32); lw[4] = 0;
    for (int pass = 0; pass < 2; pass++)            /* at most two corrections */
        if (bn_sub(t, r1, lw, 5) == 0) memcpy(r1, t, sizeof(t));
    memcpy(r, r1, 32);
}

/* sc.c:191-214 sc_import: up to 64 little-endian bytes, reduced mod l (never rejected) */
static void sc_frombytes(sc r, const uint8_t *s, size_t len)
{
    uint64_t x[8] = {0};
    for (size_t i = 0; i < len && i < 64; i++) x[i / 8] |= (uint64_t)s[i] << (8 * (i % 8));
    sc_barrett(r, x);
}

/* sc.c:221-236 sc_export */
static void sc_tobytes(uint8_t s[32], const sc a)
{
    for (int i = 0; i < 32; i++) s[i] = (uint8_t)(a[i / 8] >> (8 * (i % 8)));
}

/* sc.c:241-266 sc_mul */
static void sc_mul(sc r, const sc a, const sc b)
{
    uint64_t x[8];
    bn_mul(x, a, 4, b, 4);
    sc_barrett(r, x);
}

/* sc.h:53-59 sc_add followed by the reduction its consumers apply (sc_export / sc_reduce) */
static void sc_add(sc r, const sc a, const sc b)
{
    uint64_t x[8] = {0};
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; x[i] = (uint64_t)c; c >>= 64; }
    x[4] = (uint64_t)c;
    sc_barrett(r, x);
}

/* the 52-bit limb i of a reduced scalar, as the reference's 64-bit build holds it (sc.h:26) */
static uint64_t sc_limb52(const sc a, int i)
{
    int bit = 52 * i, w = bit / 64, o = bit % 64;
    uint64_t v = a[w] >> o;
    if (o > 12 && w + 1 < 4) v |= a[w + 1] << (64 - o);
    return v & ((UINT64_C(1) << 52) - 1);
}

/* sc.c:272-281 jsfdigit */
static int jsf_digit(uint64_t a, uint64_t b)
{
    int u = 2 - (int)(a & 3);
    if (u == 2) return 0;
    if (((a & 7) == 3 || (a & 7) == 5) && (b & 3) == 2) return -u;
    return u;
}

/* sc.c:297-324 sc_jsf.  The reference feeds the recoder one 52-bit limb at a time, so the
 * three-bit look-ahead of jsfdigit does not see across a limb boundary; reproduced here. */
#define JSF_LEN 261
static int sc_jsf(int8_t u0[JSF_LEN], int8_t u1[JSF_LEN], const sc a, const sc b)
{
    int64_t n0 = 0, n1 = 0;
    int k = 0;
    for (int i = 0; i < 5; i++) {
        n0 += (int64_t)sc_limb52(a, i);
        n1 += (int64_t)sc_limb52(b, i);
        for (int j = 0; j < 52; j++, k++) {
            u0[k] = (int8_t)jsf_digit((uint64_t)n0, (uint64_t)n1);
            u1[k] = (int8_t)jsf_digit((uint64_t)n1, (uint64_t)n0);
            n0 = (n0 - u0[k]) >> 1;
            n1 = (n1 - u1[k]) >> 1;
        }
    }
    u0[k] = (int8_t)jsf_digit((uint64_t)n0, (uint64_t)n1);
    u1[k] = (int8_t)jsf_digit((uint64_t)n1, (uint64_t)n0);
    while (k >= 0 && u0[k] == 0 && u1[k] == 0) k--;
    return k;
}

/* ------------------------------------------------------------------------------------------
 * SHA-512  (sha512.c:83-210; one-shot over up to three segments)
 * ---------------------------------------------------------------------------------------- */

static const uint64_t SHA_K[80] = {
    0x428a2f98d728ae22ULL,0x7137449123ef65cdULL,0xb5c0fbcfec4d3b2fULL,0xe9b5dba58189dbbcULL,
    0x3956c25bf348b538ULL,0x59f111f1b605d019ULL,0x923f82a4af194f9bULL,0xab1c5ed5da6d8118ULL,
    0xd807aa98a3030242ULL,0x12835b0145706fbeULL,0x243185be4ee4b28cULL,0x550c7dc3d5ffb4e2ULL,
    0x72be5d74f27b896fULL,0x80deb1fe3b1696b1ULL,0x9bdc06a725c71235ULL,0xc19bf174cf692694ULL,
    0xe49b69c19ef14ad2ULL,0xefbe4786384f25e3ULL,0x0fc19dc68b8cd5b5ULL,0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL,0x4a7484aa6ea6e483ULL,0x5cb0a9dcbd41fbd4ULL,0x76f988da831153b5ULL,
    0x983e5152ee66dfabULL,0xa831c66d2db43210ULL,0xb00327c898fb213fULL,0xbf597fc7beef0ee4ULL,
    0xc6e00bf33da88fc2ULL,0xd5a79147930aa725ULL,0x06ca6351e003826fULL,0x142929670a0e6e70ULL,
    0x27b70a8546d22ffcULL,0x2e1b21385c26c926ULL,0x4d2c6dfc5ac42aedULL,0x53380d139d95b3dfULL,
    0x650a73548baf63deULL,0x766a0abb3c77b2a8ULL,0x81c2c92e47edaee6ULL,0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL,0xa81a664bbc423001ULL,0xc24b8b70d0f89791ULL,0xc76c51a30654be30ULL,
    0xd192e819d6ef5218ULL,0xd69906245565a910ULL,0xf40e35855771202aULL,0x106aa07032bbd1b8ULL,
    0x19a4c116b8d2d0c8ULL,0x1e376c085141ab53ULL,0x2748774cdf8eeb99ULL,0x34b0bcb5e19b48a8ULL,
    0x391c0cb3c5c95a63ULL,0x4ed8aa4ae3418acbULL,0x5b9cca4f7763e373ULL,0x682e6ff3d6b2b8a3ULL,
    0x748f82ee5defb2fcULL,0x78a5636f43172f60ULL,0x84c87814a1f0ab72ULL,0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL,0xa4506cebde82bde9ULL,0xbef9a3f7b2c67915ULL,0xc67178f2e372532bULL,
    0xca273eceea26619cULL,0xd186b8c721c0c207ULL,0xeada7dd6cde0eb1eULL,0xf57d4f7fee6ed178ULL,
    0x06f067aa72176fbaULL,0x0a637dc5a2c898a6ULL,0x113f9804bef90daeULL,0x1b710b35131c471bULL,
    0x28db77f523047d84ULL,0x32caab7b40c72493ULL,0x3c9ebe0a15c9bebcULL,0x431d67c49c100d4cULL,
    0x4cc5d4becb3e42b6ULL,0x597f299cfc657e2aULL,0x5fcb6fab3ad6faecULL,0x6c44198c4a475817ULL };

static inline uint64_t ror64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

static void sha_block(uint64_t st[8], const uint8_t blk[128])
{
    uint64_t w[16], s[8];
    for (int i = 0; i < 16; i++) {
        w[i] = 0;
        for (int j = 0; j < 8; j++) w[i] = (w[i] << 8) | blk[8 * i + j];
    }
    memcpy(s, st, sizeof(s));
    for (int r = 0; r < 80; r++) {
        if (r >= 16) {
            uint64_t w15 = w[(r + 1) & 15], w2 = w[(r + 14) & 15];
            w[r & 15] += (ror64(w15, 1) ^ ror64(w15, 8) ^ (w15 >> 7)) + w[(r + 9) & 15]
                       + (ror64(w2, 19) ^ ror64(w2, 61) ^ (w2 >> 6));
        }
        uint64_t a = s[0], b = s[1], c = s[2], e = s[4], f = s[5], g = s[6];
        uint64_t t1 = s[7] + (ror64(e, 14) ^ ror64(e, 18) ^ ror64(e, 41)) + ((e & f) ^ (~e & g))
                    + SHA_K[r] + w[r & 15];
        uint64_t t2 = (ror64(a, 28) ^ ror64(a, 34) ^ ror64(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
        s[7] = g; s[6] = f; s[5] = e; s[4] = s[3] + t1; s[3] = c; s[2] = b; s[1] = a; s[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) st[i] += s[i];
}

struct seg { const uint8_t *p; size_t n; };

static void sha512_segs(uint8_t out[64], const struct seg *segs, int nseg)
{
    uint64_t st[8] = {
        0x6a09e667f3bcc908ULL,0xbb67ae8584caa73bULL,0x3c6ef372fe94f82bULL,0xa54ff53a5f1d36f1ULL,
        0x510e527fade682d1ULL,0x9b05688c2b3e6c1fULL,0x1f83d9abfb41bd6bULL,0x5be0cd19137e2179ULL };
    uint8_t buf[128];
    size_t fill = 0;
    u128 total = 0;
    for (int s = 0; s < nseg; s++) {
        const uint8_t *p = segs[s].p; size_t n = segs[s].n;
        total += n;
        while (n > 0) {
            size_t take = 128 - fill; if (take > n) take = n;
            memcpy(buf + fill, p, take); fill += take; p += take; n -= take;
            if (fill == 128) { sha_block(st, buf); fill = 0; }
        }
    }
    buf[fill++] = 0x80;
    if (fill > 112) { memset(buf + fill, 0, 128 - fill); sha_block(st, buf); fill = 0; }
    memset(buf + fill, 0, 112 - fill);
    u128 bits = total << 3;
    for (int i = 0; i < 16; i++) buf[127 - i] = (uint8_t)(bits >> (8 * i));
    sha_block(st, buf);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(st[i] >> (56 - 8 * j));
}

void orc_sha512(uint8_t out[64], const uint8_t *data, size_t len)
{
    struct seg s = { data, len };
    sha512_segs(out, &s, 1);
}

/* ------------------------------------------------------------------------------------------
 * twisted Edwards group  -x^2 + y^2 = 1 + d x^2 y^2
 * ---------------------------------------------------------------------------------------- */

struct ge { fe X, Y, T, Z; };                   /* ed.h:13-18, t = xy/z */
struct pc { fe ymx, ypx, t2d; };                /* ed.c:30-34 struct pced, z = 1 implied */

static struct pc PC_B;                          /* ed.c:46-52 pced_B */
static struct pc COMB[32][8];                   /* ed.c:41-43 ed_lookup: (k+1) 16^(2i) B */
static sc SC_OFF;                               /* sc.c:39 con_off = 8 (16^64 - 1)/15 mod l */

static void ge_neutral(struct ge *p)            /* ed.c:72 ed_zero */
{
    fe_set_small(p->X, 0); fe_set_small(p->Y, 1); fe_set_small(p->T, 0); fe_set_small(p->Z, 1);
}

/* ed.c:100-149 ed_import.  Total: never fails.  Bit 255 is the sign; y is NOT range-checked
 * (y >= p wraps); if neither candidate is a root the second candidate is kept anyway. */
static void ge_frombytes(struct ge *p, const uint8_t in[32])
{
    uint8_t ybytes[32];
    fe u, v, a, b;
    memcpy(ybytes, in, 32); ybytes[31] &= 0x7f;
    fe_frombytes(p->Y, ybytes);

    fe one; fe_set_small(one, 1);
    fe_sq(u, p->Y);
    fe_mul(v, K_D, u);
    fe_sub(u, u, one);                          /* u = y^2 - 1 */
    fe_add(v, v, one);                          /* v = d y^2 + 1 */

    fe_sq(a, v);                                /* v^2 */
    fe_sq(b, a);                                /* v^4 */
    fe_mul(a, a, u); fe_mul(a, a, v);           /* a = u v^3 */
    fe_mul(b, b, a);                            /* u v^7 */
    fe_pow2523(b, b);
    fe_mul(b, b, a);                            /* beta = u v^3 (u v^7)^((p-5)/8) */

    fe_sq(a, b); fe_mul(a, a, v);               /* v beta^2 */
    if (fe_eq(a, u)) fe_copy(p->X, b);
    else             fe_mul(p->X, K_J, b);

    if (fe_parity(p->X) != (in[31] >> 7)) fe_neg(p->X, p->X);
    fe_mul(p->T, p->X, p->Y);
    fe_set_small(p->Z, 1);
}

/* ed.c:155-169 ed_export */
static void ge_tobytes(uint8_t out[32], const struct ge *p)
{
    fe zi, x, y;
    fe_inv(zi, p->Z);
    fe_mul(x, p->X, zi);
    fe_mul(y, p->Y, zi);
    fe_tobytes(out, y);
    out[31] |= (uint8_t)(fe_parity(x) << 7);
}

/* shared tail of every addition formula in ed.c (e.g. ed.c:193-202) */
static void ge_finish(struct ge *o, const fe a, const fe b, const fe c, const fe d)
{
    fe e, f, g, h;
    fe_sub(e, b, a); fe_sub(f, d, c); fe_add(g, d, c); fe_add(h, b, a);
    fe_mul(o->X, e, f); fe_mul(o->Y, g, h); fe_mul(o->T, e, h); fe_mul(o->Z, f, g);
}

/* ed.c:175-203 ed_add (sub = 0) and ed.c:245-273 ed_sub (sub = 1) */
static void ge_addsub(struct ge *o, const struct ge *p, const struct ge *q, int sub)
{
    fe a, b, c, d, t, qm, qp;
    fe_sub(qm, q->Y, q->X); fe_add(qp, q->Y, q->X);
    fe_sub(a, p->Y, p->X); fe_mul(a, a, sub ? qp : qm);
    fe_add(b, p->Y, p->X); fe_mul(b, b, sub ? qm : qp);
    fe_mul(c, p->T, q->T); fe_mul(c, c, sub ? K_M2D : K_2D);
    fe_mul(t, p->Z, q->Z); fe_add(d, t, t);
    ge_finish(o, a, b, c, d);
}

/* ed.c:211-237 ed_double: the addition law with P = Q (4 S + 5 M), not dbl-2008-hwcd */
static void ge_double(struct ge *o, const struct ge *p)
{
    fe a, b, c, d, t;
    fe_sub(a, p->Y, p->X); fe_sq(a, a);
    fe_add(b, p->Y, p->X); fe_sq(b, b);
    fe_sq(c, p->T); fe_mul(c, c, K_2D);
    fe_sq(t, p->Z); fe_add(d, t, t);
    ge_finish(o, a, b, c, d);
}

/* ed.c:282-305 ed_add_pc (sub = 0) and ed.c:310-335 ed_sub_pc (sub = 1) */
static void ge_addsub_pc(struct ge *o, const struct ge *p, const struct pc *q, int sub)
{
    fe a, b, c, d;
    fe_sub(a, p->Y, p->X); fe_mul(a, a, sub ? q->ypx : q->ymx);
    fe_add(b, p->Y, p->X); fe_mul(b, b, sub ? q->ymx : q->ypx);
    fe_mul(c, p->T, q->t2d); if (sub) fe_neg(c, c);
    fe_add(d, p->Z, p->Z);
    ge_finish(o, a, b, c, d);
}

/* ed.c:436-442 ed_precompute */
static void ge_to_pc(struct pc *o, const struct ge *p)
{
    fe_sub(o->ymx, p->Y, p->X); fe_add(o->ypx, p->Y, p->X); fe_mul(o->t2d, p->T, K_2D);
}

/* ed.c:346-391 scale16: digit * 16^(2 row) * B for digit in [-8,7] as a pc point.  The
 * reference scans the whole row under arithmetic masks (constant time); the oracle indexes. */
static void comb_entry(struct pc *o, int row, int digit)
{
    int mag = digit < 0 ? -digit : digit;
    if (mag == 0) {                             /* ed.c:73 pced_zero */
        fe_set_small(o->ymx, 1); fe_set_small(o->ypx, 1); fe_set_small(o->t2d, 0);
        return;
    }
    const struct pc *e = &COMB[row][mag - 1];
    if (digit > 0) { *o = *e; return; }
    fe_copy(o->ymx, e->ypx); fe_copy(o->ypx, e->ymx); fe_neg(o->t2d, e->t2d);
}

/* ed.c:397-430 ed_scale_base */
static void ge_scale_base(struct ge *out, const sc x)
{
    sc shifted; uint8_t pack[32];
    struct ge r0, r1; struct pc e;
    sc_add(shifted, x, SC_OFF);
    sc_tobytes(pack, shifted);
    ge_neutral(&r0); ge_neutral(&r1);
    for (int i = 0; i < 32; i++) {
        comb_entry(&e, i, (pack[i] & 15) - 8); ge_addsub_pc(&r0, &r0, &e, 0);
        comb_entry(&e, i, (pack[i] >> 4) - 8); ge_addsub_pc(&r1, &r1, &e, 0);
    }
    for (int i = 0; i < 4; i++) ge_addsub(&r1, &r1, &r1, 0);
    ge_addsub(out, &r0, &r1, 0);
}

/* ed.c:455-507 ed_dual_scale: x B + y Q, Q affine; Shamir's trick over the JSF digits */
static void ge_dual_scale(struct ge *r, const sc x, const sc y, const struct ge *q)
{
    int8_t ux[JSF_LEN], uy[JSF_LEN];
    struct ge qpb, qmb; struct pc pcq;
    ge_neutral(r);
    int n = sc_jsf(ux, uy, x, y);
    if (n < 0) return;
    ge_addsub_pc(&qpb, q, &PC_B, 0);
    ge_addsub_pc(&qmb, q, &PC_B, 1);
    ge_to_pc(&pcq, q);
    for (int i = n; ; i--) {
        int a = ux[i], b = uy[i];
        if (a == 1) {
            if (b == 1) ge_addsub(r, r, &qpb, 0);
            else if (b == -1) ge_addsub(r, r, &qmb, 1);
            else ge_addsub_pc(r, r, &PC_B, 0);
        } else if (a == -1) {
            if (b == 1) ge_addsub(r, r, &qmb, 0);
            else if (b == -1) ge_addsub(r, r, &qpb, 1);
            else ge_addsub_pc(r, r, &PC_B, 1);
        } else if (b == 1) ge_addsub_pc(r, r, &pcq, 0);
        else if (b == -1) ge_addsub_pc(r, r, &pcq, 1);
        if (i == 0) break;
        ge_double(r, r);
    }
}

/* ------------------------------------------------------------------------------------------
 * one-time constants (the reference ships them as literal tables; the oracle derives them)
 * ---------------------------------------------------------------------------------------- */

static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void ge_to_affine_pc(struct pc *o, const struct ge *p)
{
    fe zi, x, y, t;
    fe_inv(zi, p->Z);
    fe_mul(x, p->X, zi); fe_mul(y, p->Y, zi); fe_mul(t, x, y);
    fe_sub(o->ymx, y, x); fe_add(o->ypx, y, x); fe_mul(o->t2d, t, K_2D);
}

static void orc_init(void)
{
    fe_frombytes(K_D, BYTES_D);
    fe_add(K_2D, K_D, K_D);
    fe_neg(K_M2D, K_2D);
    fe_frombytes(K_J, BYTES_J);

    struct ge b, row, acc;
    ge_frombytes(&b, BYTES_BY);
    ge_to_affine_pc(&PC_B, &b);

    row = b;                                    /* 16^(2i) B */
    for (int i = 0; i < 32; i++) {
        acc = row;
        for (int k = 0; k < 8; k++) {
            ge_to_affine_pc(&COMB[i][k], &acc);
            ge_addsub(&acc, &acc, &row, 0);
        }
        for (int s = 0; s < 8; s++) ge_double(&row, &row);
    }

    /* off = 8 (16^64 - 1) / 15 = 0x8888...88 (64 nibbles), reduced mod l */
    uint8_t eights[32]; memset(eights, 0x88, 32);
    sc_frombytes(SC_OFF, eights, 32);
}

static void ensure_init(void) { pthread_once(&g_once, orc_init); }

/* ------------------------------------------------------------------------------------------
 * protocol layer
 * ---------------------------------------------------------------------------------------- */

/* ed25519-sha512.c:31-47 ed25519_key_setup */
static void key_setup(uint8_t h[64], const uint8_t sk[32])
{
    orc_sha512(h, sk, 32);
    h[31] &= 0x7f; h[31] |= 0x40; h[0] &= 0xf8;
}

/* ed25519-sha512.c:53-78 genpub / ed25519_genpub */
void orc_ed25519_genpub(uint8_t pub[32], const uint8_t sec[32])
{
    uint8_t h[64]; sc a; struct ge A;
    ensure_init();
    key_setup(h, sec);
    sc_frombytes(a, h, 32);
    ge_scale_base(&A, a);
    ge_tobytes(pub, &A);
}

/* ed25519-sha512.c:84-137 sign / ed25519_sign */
void orc_ed25519_sign(uint8_t sig[64], const uint8_t sec[32], const uint8_t pub[32],
                      const uint8_t *data, size_t len)
{
    uint8_t h[64]; sc a, r, t, S; struct ge R;
    ensure_init();
    key_setup(h, sec);
    sc_frombytes(a, h, 32);

    struct seg s1[2] = { { h + 32, 32 }, { data, len } };
    uint8_t hr[64];
    sha512_segs(hr, s1, 2);
    sc_frombytes(r, hr, 64);

    ge_scale_base(&R, r);
    ge_tobytes(sig, &R);

    struct seg s2[3] = { { sig, 32 }, { pub, 32 }, { data, len } };
    sha512_segs(h, s2, 3);
    sc_frombytes(t, h, 64);

    sc_mul(S, t, a);
    sc_add(S, r, S);
    sc_tobytes(sig + 32, S);
}

/* ed25519-sha512.c:148-181 ed25519_verify */
int orc_ed25519_verify(const uint8_t sig[64], const uint8_t pub[32], const uint8_t *data,
                       size_t len)
{
    uint8_t h[64], check[32]; struct ge A, C; sc S, t;
    ensure_init();
    ge_frombytes(&A, pub);
    sc_frombytes(S, sig + 32, 32);
    struct seg s[3] = { { sig, 32 }, { pub, 32 }, { data, len } };
    sha512_segs(h, s, 3);
    sc_frombytes(t, h, 64);
    fe_neg(A.X, A.X); fe_neg(A.T, A.T);
    ge_dual_scale(&C, S, t, &A);
    ge_tobytes(check, &C);
    return memcmp(check, sig, 32) == 0;
}

/* x25519.c:60-94 montgomery: A <- 2A, B <- A+B, given the affine difference x(A-B) = dx */
static void mont_step(fe ax, fe az, fe bx, fe bz, const fe dx)
{
    fe sa, da, sb, db, ssa, sda, t1, t2, t3;
    fe_add(sa, ax, az); fe_sq(ssa, sa);
    fe_sub(da, ax, az); fe_sq(sda, da);
    fe_mul(ax, sda, ssa);
    fe_sub(t1, ssa, sda);
    fe_mul_small(t2, t1, 121665); fe_add(t2, t2, ssa);
    fe_mul(az, t1, t2);
    fe_add(sb, bx, bz); fe_sub(db, bx, bz);
    fe_mul(t1, da, sb); fe_mul(t2, sa, db);
    fe_add(t3, t1, t2); fe_sq(bx, t3);
    fe_sub(t3, t1, t2); fe_sq(t3, t3);
    fe_mul(bz, t3, dx);
}

/* x25519.c:129-150 do_x25519 (with x25519.c:104-123 mg_scale inlined: 256 ladder steps over
 * all 32 bytes, most significant bit first) */
void orc_x25519(uint8_t out[32], const uint8_t scalar[32], const uint8_t point[32])
{
    uint8_t s[32]; fe px, ax, az, bx, bz;
    ensure_init();
    memcpy(s, scalar, 32);
    s[0] &= 0xf8; s[31] &= 0x7f; s[31] |= 0x40;
    fe_frombytes(px, point);                    /* bit 255 folded in, not masked */
    fe_set_small(ax, 1); fe_set_small(az, 0);
    fe_copy(bx, px); fe_set_small(bz, 1);
    for (int i = 255; i >= 0; i--) {
        int bit = (s[i >> 3] >> (i & 7)) & 1;
        if (bit) mont_step(bx, bz, ax, az, px);
        else     mont_step(ax, az, bx, bz, px);
    }
    fe_inv(az, az);
    fe_mul(ax, ax, az);
    fe_tobytes(out, ax);
}

/* x25519.c:158-197 do_x25519_base: clamp, reduce mod l, comb, u = (z+y)/(z-y) */
void orc_x25519_base(uint8_t out[32], const uint8_t scalar[32])
{
    uint8_t s[32]; sc x; struct ge R; fe u, t;
    ensure_init();
    memcpy(s, scalar, 32);
    s[0] &= 0xf8; s[31] &= 0x7f; s[31] |= 0x40;
    sc_frombytes(x, s, 32);
    ge_scale_base(&R, x);
    fe_sub(t, R.Z, R.Y); fe_inv(t, t);
    fe_add(u, R.Z, R.Y); fe_mul(u, u, t);
    fe_tobytes(out, u);
}

/* ed25519-sha512.c:187-232 pk_ed25519_to_x25519 */
void orc_pk_ed25519_to_x25519(uint8_t out[32], const uint8_t in[32])
{
    struct ge P; fe u, t;
    ensure_init();
    ge_frombytes(&P, in);
    fe_add(u, P.Z, P.Y);
    fe_sub(t, P.Z, P.Y); fe_inv(t, t);
    fe_mul(u, u, t);
    fe_tobytes(out, u);
}

/* ed25519-sha512.c:239-256 sk_ed25519_to_x25519 */
void orc_sk_ed25519_to_x25519(uint8_t out[32], const uint8_t in[32])
{
    uint8_t h[64];
    key_setup(h, in);
    memcpy(out, h, 32);
}

/* ------------------------------------------------------------------------------------------
 * layer probes
 * ---------------------------------------------------------------------------------------- */

void orc_fld_mul(uint8_t out[32], const uint8_t a[32], const uint8_t b[32])
{
    fe x, y; fe_frombytes(x, a); fe_frombytes(y, b); fe_mul(x, x, y); fe_tobytes(out, x);
}
void orc_fld_sq(uint8_t out[32], const uint8_t a[32])
{
    fe x; fe_frombytes(x, a); fe_sq(x, x); fe_tobytes(out, x);
}
void orc_fld_inv(uint8_t out[32], const uint8_t a[32])
{
    fe x; fe_frombytes(x, a); fe_inv(x, x); fe_tobytes(out, x);
}
void orc_fld_pow2523(uint8_t out[32], const uint8_t a[32])
{
    fe x; fe_frombytes(x, a); fe_pow2523(x, x); fe_tobytes(out, x);
}
void orc_sc_reduce_bytes(uint8_t out[32], const uint8_t *in, size_t len)
{
    sc x; sc_frombytes(x, in, len); sc_tobytes(out, x);
}
void orc_sc_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32])
{
    sc x, y, z; sc_frombytes(x, a, 32); sc_frombytes(y, b, 32); sc_frombytes(z, c, 32);
    sc_mul(x, x, y); sc_add(x, z, x); sc_tobytes(out, x);
}
void orc_ed_import_export(uint8_t out[32], const uint8_t in[32])
{
    struct ge P; ensure_init(); ge_frombytes(&P, in); ge_tobytes(out, &P);
}
void orc_ed_scale_base(uint8_t out[32], const uint8_t scalar[32])
{
    sc x; struct ge P; ensure_init();
    sc_frombytes(x, scalar, 32); ge_scale_base(&P, x); ge_tobytes(out, &P);
}
void orc_ed_dual_scale(uint8_t out[32], const uint8_t s[32], const uint8_t t[32], const uint8_t q[32])
{
    sc x, y; struct ge Q, R; ensure_init();
    sc_frombytes(x, s, 32); sc_frombytes(y, t, 32);
    ge_frombytes(&Q, q);
    ge_dual_scale(&R, x, y, &Q);
    ge_tobytes(out, &R);
}
void orc_ed_lookup_bytes(uint8_t out[32 * 8 * 96])
{
    ensure_init();
    for (int i = 0; i < 32; i++)
        for (int k = 0; k < 8; k++) {
            uint8_t *o = out + 96 * (8 * i + k);
            fe_tobytes(o, COMB[i][k].ymx); fe_tobytes(o + 32, COMB[i][k].ypx);
            fe_tobytes(o + 64, COMB[i][k].t2d);
        }
}

/* ------------------------------------------------------------------------------------------
 * batched forms: static contiguous partition over pthreads
 * ---------------------------------------------------------------------------------------- */

struct job {
    int kind; size_t lo, hi, msg_len;
    uint8_t *o; const uint8_t *a, *b, *c;
};

static void *job_run(void *arg)
{
    struct job *j = (struct job *)arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        switch (j->kind) {
        case 0: j->o[i] = (uint8_t)orc_ed25519_verify(j->a + 64 * i, j->b + 32 * i,
                                                     j->c + j->msg_len * i, j->msg_len); break;
        case 1: orc_x25519(j->o + 32 * i, j->a + 32 * i, j->b + 32 * i); break;
        case 2: orc_ed25519_sign(j->o + 64 * i, j->a + 32 * i, j->b + 32 * i,
                                 j->c + j->msg_len * i, j->msg_len); break;
        case 3: orc_ed25519_genpub(j->o + 32 * i, j->a + 32 * i); break;
        }
    }
    return NULL;
}

static void run_jobs(struct job proto, size_t n, int threads)
{
    ensure_init();
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    struct job *jobs = (struct job *)malloc(sizeof(struct job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = proto;
        jobs[t].lo = n * (size_t)t / (size_t)threads;
        jobs[t].hi = n * (size_t)(t + 1) / (size_t)threads;
        if (t > 0) pthread_create(&th[t], NULL, job_run, &jobs[t]);
    }
    job_run(&jobs[0]);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

void orc_ed25519_verify_batch(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs,
                              const uint8_t *msgs, size_t msg_len, size_t n, int threads)
{
    struct job j = { 0, 0, 0, msg_len, ok, sigs, pubs, msgs };
    run_jobs(j, n, threads);
}
void orc_x25519_batch(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n, int threads)
{
    struct job j = { 1, 0, 0, 0, out, scalars, points, NULL };
    run_jobs(j, n, threads);
}
void orc_ed25519_sign_batch(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs,
                            const uint8_t *msgs, size_t msg_len, size_t n, int threads)
{
    struct job j = { 2, 0, 0, msg_len, sigs, secs, pubs, msgs };
    run_jobs(j, n, threads);
}
void orc_ed25519_genpub_batch(uint8_t *pubs, const uint8_t *secs, size_t n, int threads)
{
    struct job j = { 3, 0, 0, 0, pubs, secs, NULL, NULL };
    run_jobs(j, n, threads);
}
