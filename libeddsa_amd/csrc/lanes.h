// lanes.h - what ONE lane computes in each kernel of kernels.hip, as plain functions over
// pointers (no threadIdx/blockIdx), so that the very same source also compiles for the host CPU
// under -DED_HOST_CHECK and is checked there against the oracle with every bound asserted
// (tests/host_check/, tests/test_device_source_on_host.py).
//
//   x25519_lane            x25519.c:129-150 do_x25519
//   table_entry_lane       one entry of the generated lib/ed_lookup64.h
//   verify_prepare_lane    ed25519-sha512.c:148-172 (hash, scalars, import of -A) + table 0..8 * -A
//   verify_main_lane       ed.c:455-507 ed_dual_scale (windowed, uniform control flow)
//   verify_encode_lane     ed.c:155-169 ed_export + ed25519-sha512.c:176-180 (given 1/Z)
//   verify_half_scalars_lane, verify_half_point_lane, verify_half_main_lane   the same verdict from half-length
//                          scalars (halve.h): u (S B - t A - R) = 0, 132 doublings
//   verify_exact_lane      the reference's own JSF/Shamir chain, for keys that are not on the curve (one lane per item: the host
//                          build and the layer probe; the kernels run its four-lane form, quad_lanes.h)
//   scale_base_lane        ed.c:397-430 ed_scale_base (comb, constant-time select)
//   genpub_point_lane, sign_point_lane, sign_finish_lane, encode_lane   ed25519-sha512.c:53-123
//   x25519_base_point_lane, x25519_base_finish_lane                      x25519.c:158-197
//   pk_to_x_lane, sk_to_x_lane   ed25519-sha512.c:187-256
#pragma once
#include "fe25519.h"
#include "ge25519.h"
#include "sc25519.h"
#include "sha512.h"
#include "halve.h"

#define TABLE_BASE16_ENTRIES 32769 /* k*B, k = 0..32768: 16-bit signed windows of S (4 MiB, L2/MALL) */
#define COMB_W 6                  /* signed window width of the fixed-base comb (the reference's is 4, ed.c:397-430) */
#define COMB_HALF (1 << (COMB_W - 1))          /* digits d in [-COMB_HALF, COMB_HALF - 1] */
#define COMB_DIGITS (COMB_W == 4 ? 64 : COMB_W == 5 ? 52 : 44)   /* digits of x + offset: 64 x 4, 52 x 5 or 44 x 6 bits */
#define COMB_ROWS (COMB_DIGITS / 2)            /* even digits and odd digits share a row */
#define TABLE_COMB_ENTRIES (COMB_ROWS * COMB_HALF) /* comb[i][k] = (k+1) * 2^(2*COMB_W*i) * B, k < COMB_HALF */
#define TABLE_ENTRY_WORDS 32      /* 3 x 10 limbs + 2 padding words */
#define COMB_IMG_ENTRIES COMB_HALF       /* LDS image of a comb row: entry m - 1 = m * 2^(2*COMB_W*i) * B, m = 1..COMB_HALF */
#define COMB_IMG_ENTRY_WORDS 36   /* 30 limbs + 6 padding words: entries start 4 banks apart */
#define COMB_IMG_WORDS (COMB_ROWS * COMB_IMG_ENTRIES * COMB_IMG_ENTRY_WORDS)
#define VERIFY_TABLE_ENTRIES 9    /* 0..8 times -A, cached form */
#define VERIFY_ENTRY_WORDS 32     /* ymx | ypx | t2d | z2, 255 bits packed into eight words each: one 128-byte line */

namespace ed {

struct alignas(16) word4 { uint32_t x, y, z, w; };   // one 16-byte load / store

// 256-bit little-endian value << s (s = 1, 4 or 16): the scalar is consumed from the top
template <int S>
ED_DEV void shl256(uint32_t w[8]) {
#pragma unroll
  for (int i = 7; i > 0; i--) w[i] = (w[i] << S) | (w[i - 1] >> (32 - S));
  w[0] <<= S;
}

// The message of item `item`: msgs + item * stride, len bytes (off == nullptr), or the ragged span off[item] .. off[item + 1].
// A ragged span is clamped into [0, *end) - *end is the table's own last entry, which the caller vouches for as the size of
// the message buffer - and to a length that is not negative: a table that is not non-decreasing (the host-pointer entry
// points refuse one with hipErrorInvalidValue; a table in device memory cannot be inspected without a pass of its own)
// never makes a lane read outside the aligned 4-byte words that hold [msgs, msgs + *end) (sha512.h loads a block's words whole: an
// unaligned span shares its first and last word with up to 3 neighbouring bytes).  Such an item is hashed over the clamped span.
ED_DEV void ragged_span(uint64_t& lo, uint64_t& hi, const uint64_t* off, const uint64_t* end, size_t item) {
  const uint64_t total = *end;
  lo = off[item]; hi = off[item + 1];
  lo = lo < total ? lo : total;
  hi = hi < lo ? lo : hi < total ? hi : total;
}
ED_DEV void msg_span(const uint8_t*& m, size_t& mlen, const uint8_t* msgs, const uint64_t* off, const uint64_t* end,
                     size_t len, size_t stride, size_t item) {
  if (off) {
    uint64_t lo, hi;
    ragged_span(lo, hi, off, end, item);
    m = msgs + lo; mlen = (size_t)(hi - lo);
  } else { m = msgs + item * stride; mlen = len; }
}

// x25519.c:137-140: clamp copy
ED_DEV void clamp(uint32_t s[8]) {
  s[0] &= 0xfffffff8u;
  s[7] = (s[7] & 0x7fffffffu) | 0x40000000u;
}

// ---------------------------------------------------------------------------------------------
// X25519: x25519.c:60-150 (montgomery, mg_scale, do_x25519)
// ---------------------------------------------------------------------------------------------

// the ladder of x25519.c:104-123 (mg_scale): (x2 : z2) = s * (pt : 1), s clamped here, both tight
ED_DEV void x25519_ladder_lane(fe& x2, fe& z2, uint32_t s[8], const uint32_t pt[8]) {
  fe x1, x3, z3;
  clamp(s);
  fe_frombytes(x1, pt);                          // bit 255 folded in as +19, not masked (fld.c:153)
  fe_set(x2, 1); fe_set(z2, 0); x3 = x1; fe_set(z3, 1);
  fe19 x1_19;                                    // x1 is the second operand of one multiplication in every step
  fe_premul19(x1_19, x1);
#ifndef ED_HOST_CHECK
#pragma unroll
  for (int j = 1; j < 10; j++) asm volatile("" : "+v"(x1_19.v[j]));   // keep them: the compiler would recompute 19 x1 in every step
#endif
  // bit 255 of the clamped scalar is 0 and the step for it maps (1:0),(x1:1) to itself
  // projectively, so the ladder starts at bit 254.  The scalar is consumed from the top, one word at a
  // time: `cur` holds the current word, shifted left once per step.
  uint32_t cur = s[7] << 1;
  // x25519.c:104-123 swaps (x2,z2) <-> (x3,z3) before and after every step.  The differential
  // addition is symmetric in the two points, so only the doubling has to know which slot holds
  // the point to double: with `swap` = (slots currently exchanged) ^ (this bit), the doubling's
  // inputs x+z, x-z are SELECTED (20 v_cndmask) instead of four elements being swapped (40);
  // afterwards slot 2 holds the double and slot 3 the sum, i.e. the slots are exchanged iff the
  // bit was 1.  Same field values as the reference's swap-step-swap.
  uint32_t swap = 0;
#pragma unroll 1
  for (int t = 254; t >= 3; t--) {
    const uint32_t bit = cur >> 31;
    cur <<= 1;
    if ((t & 31) == 0) {                         // word t / 32 is used up: fetch the next one (uniform, 7 times)
      const int w = (t >> 5) - 1;
      cur = w == 6 ? s[6] : w == 5 ? s[5] : w == 4 ? s[4] : w == 3 ? s[3] : w == 2 ? s[2] : w == 1 ? s[1] : s[0];
    }
    swap ^= bit;
    fe a, aa, b, bb, e, c, d, da, cb, t1;
    fe_add(a, x2, z2);                           // 2u
    fe_sub(b, x2, z2);                           // 3u
    fe_add(c, x3, z3);                           // 2u
    fe_sub(d, x3, z3);                           // 3u
    fe_mul(da, d, a);
    fe_mul(cb, c, b);
    fe_cmov(a, c, swap != 0);                    // the point to double: slot 3 when exchanged
    fe_cmov(b, d, swap != 0);
    swap = bit;
    fe_sq(aa, a);
    fe_sq(bb, b);
    fe_sub(e, aa, bb);                           // 3u
    fe_add(t1, da, cb);                          // 2u
    fe_sq(x3, t1);
    fe_sub(t1, da, cb);                          // 3u
    fe_sq(t1, t1);
    fe_mul_pre(z3, t1, x1, x1_19);
    fe_mul(x2, aa, bb);
    fe_mul121665(t1, e);                         // x25519.c:78 fld_scale(T2, T1, 121665)
    fe_add(t1, t1, aa);                          // 2u
    fe_mul(z2, e, t1);
  }
  fe_cswap(x2, x3, swap != 0);
  fe_cswap(z2, z3, swap != 0);
  // bits 2, 1, 0 of the clamped scalar are 0: their steps never swap, and (x2 : z2) -- all that is
  // used afterwards -- only goes through the doubling half of x25519.c:60-94 (same field values)
#pragma unroll 1
  for (int t = 2; t >= 0; t--) {
    fe a, aa, b, bb, e, t1;
    fe_add(a, x2, z2);                           // 2u
    fe_sq(aa, a);
    fe_sub(b, x2, z2);                           // 3u
    fe_sq(bb, b);
    fe_sub(e, aa, bb);                           // 3u
    fe_mul(x2, aa, bb);
    fe_mul121665(t1, e);
    fe_add(t1, t1, aa);                          // 2u
    fe_mul(z2, e, t1);
  }
}

// x25519.c:144-149: out = x2 / z2 given zinv = 1 / z2 (0 when z2 = 0, as fld_inv)
ED_DEV void x25519_finish_lane(uint32_t out[8], const fe& x2, const fe& zinv) {
  fe u;
  fe_mul(u, x2, zinv);
  fe_tobytes(out, u);
}

// the whole of do_x25519 for one item (host check and single-lane use; the kernels split it so that
// eight items per lane share the inversion)
ED_DEV void x25519_lane(uint32_t out[8], uint32_t s[8], const uint32_t pt[8]) {
  fe x2, z2;
  x25519_ladder_lane(x2, z2, s, pt);
  fe_inv(z2, z2);                                // z = 0 -> 0 (x25519.c:145)
  x25519_finish_lane(out, x2, z2);
}

// ---------------------------------------------------------------------------------------------
// base-point tables (what the reference ships as generated data, lib/ed_lookup64.h)
// ---------------------------------------------------------------------------------------------
// Entry = 32 words: y-x | y+x | 2dxy (10 canonical limbs each) + 2 words of padding.
//   base16[k], k = 0..32768 : k * B                (16-bit signed windows of S in verify)
//   comb[i][k], i < COMB_ROWS, k < COMB_HALF : (k+1) * 2^(2 COMB_W i) * B  (ed.c:41-43 ed_lookup has 32 rows of 8: w = 4)

ED_DEV void niels_store(uint32_t* dst, const ge_niels& n) {
#pragma unroll
  for (int j = 0; j < 10; j++) { dst[j] = n.ymx.v[j]; dst[10 + j] = n.ypx.v[j]; dst[20 + j] = n.t2d.v[j]; }
  dst[30] = 0; dst[31] = 0;
}

// entries are 128 bytes, 128-byte aligned: eight 16-byte loads (one cache line)
ED_DEV void niels_load(ge_niels& n, const uint32_t* src) {
  const word4* p = reinterpret_cast<const word4*>(src);
  uint32_t w[32];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const word4 v = p[q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
#pragma unroll
  for (int j = 0; j < 10; j++) { n.ymx.v[j] = w[j]; n.ypx.v[j] = w[10 + j]; n.t2d.v[j] = w[20 + j]; }
}

// dst = affine niels form of mult * 2^shift * B
ED_DEV void table_entry_lane(uint32_t* dst, uint32_t mult, uint32_t shift) {
  ge b, acc;
  ge_cached bc;
  ge_base(b);
  ge_to_cached(bc, b);
  ge_neutral(acc);
  for (int bit = 15; bit >= 0; bit--) {          // acc = mult * B, mult < 65536
    ge_dbl(acc, acc, true);
    if ((mult >> bit) & 1) ge_add_cached(acc, acc, bc, true);
  }
  for (uint32_t s = 0; s < shift; s++) ge_dbl(acc, acc, true);
  ge_niels n;
  ge_to_niels_affine(n, acc);                    // identity -> (1, 1, 0)
  niels_store(dst, n);
}

// ---------------------------------------------------------------------------------------------
// Ed25519 verify: ed25519-sha512.c:148-181
// ---------------------------------------------------------------------------------------------
// C = S*B + t*(-A) is evaluated as ONE left-to-right pass over 64 four-bit windows:
//     acc = 16*acc + d_i*(-A)  [+ e_j*B when i = 4j]
// with d_i in [-8,7] (ed.c:407-409's x + 0x88..8 recoding) looked up in a per-item table of
// 0..8 times -A, and e_j in [-32768,32767] (one every four windows, i = 4j) looked up in a
// 32769-entry table of multiples of B (4 MiB, shared by all lanes, L2 / Infinity-Cache resident).
// Control flow is uniform; the reference's 9-way data-dependent branch (ed.c:480-501) would
// serialise all 64 lanes of a wave.  Equality of the result with the reference's:
// DESIGN.md "Why the windowed evaluation is bit-exact".

// table entry = 32 words: ymx | ypx | t2d | z2, each packed into eight words (fe_pack), contiguous per item and
// 128-byte aligned, so that a lookup is ONE L2 line (eight 16-byte loads).  As forty plain limbs an entry straddled
// two lines, and the table reads - 3 TB/s at 2^20 items - cost the power-bound chip 10 % of the main kernel in clock
// (measurement); the unpacking costs 2 % in instructions.
ED_DEV void cached_store(uint32_t* tab, int entry, const ge_cached& c) {
  word4* p = reinterpret_cast<word4*>(tab + entry * VERIFY_ENTRY_WORDS);
  const fe* f[4] = {&c.ymx, &c.ypx, &c.t2d, &c.z2};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t w[8];
    fe_pack(w, *f[k]);
    p[2 * k] = word4{w[0], w[1], w[2], w[3]};
    p[2 * k + 1] = word4{w[4], w[5], w[6], w[7]};
  }
}
// an entry as loaded, so that the loads can be issued a window ahead of their use
struct cached_raw { word4 q[8]; };
ED_DEV void cached_load_raw(cached_raw& r, const uint32_t* tab, uint32_t entry) {
  const word4* p = reinterpret_cast<const word4*>(tab + entry * VERIFY_ENTRY_WORDS);
#pragma unroll
  for (int q = 0; q < 8; q++) r.q[q] = p[q];
}
ED_DEV void cached_from_raw(ge_cached& c, const cached_raw& r) {
  fe* f[4] = {&c.ymx, &c.ypx, &c.t2d, &c.z2};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t w[8] = {r.q[2 * k].x, r.q[2 * k].y, r.q[2 * k].z, r.q[2 * k].w,
                           r.q[2 * k + 1].x, r.q[2 * k + 1].y, r.q[2 * k + 1].z, r.q[2 * k + 1].w};
    fe_unpack(*f[k], w);
  }
}
ED_DEV void cached_load(ge_cached& c, const uint32_t* tab, uint32_t entry) {
  cached_raw r;
  cached_load_raw(r, tab, entry);
  cached_from_raw(c, r);
}

// The three steps of the prepare kernel, separate so that the kernel can load S only when it is needed
// (held across the hash it cost a spill) and store each result as soon as it exists:
// t = SHA-512(R || A || M) mod l as digit words (nibble - 8 is the signed digit)
ED_DEV void verify_hash_lane(uint32_t tw[8], const uint32_t rw[8], const uint32_t aw[8], const uint8_t* m, size_t mlen) {
  uint32_t pre[16], dig[16];
#pragma unroll
  for (int k = 0; k < 8; k++) { pre[k] = rw[k]; pre[8 + k] = aw[k]; }
  sha512_prefix_msg<16>(dig, pre, m, mlen);
  sc t;
  sc_from_words<16>(t, dig);
  sc_to_words(tw, t);
  words_add_pattern(tw, 0x88888888u);
}
// S mod l (not range-checked: sc.c:191-214) as digit words (halfword - 32768 is the signed digit)
ED_DEV void verify_s_lane(uint32_t sw[8]) {
  sc s;
  sc_from_words<8>(s, sw);
  sc_to_words(sw, s);
  words_add_pattern(sw, 0x80008000u);
}
// a (affine: Z = 1, as ge_frombytes leaves it) and its multiples 0..8 in cached form at tab; "+ a" is the
// seven-multiplication mixed addition
ED_DEV void verify_table_point_lane(uint32_t* tab, const ge& a) {
  ge p, q;
  ge_cached c;
  ge_niels n1;
  ge_neutral(p);
  ge_to_cached(c, p);  cached_store(tab, 0, c);
  ge_to_cached(c, a);  cached_store(tab, 1, c);
  n1.ymx = c.ymx; n1.ypx = c.ypx; n1.t2d = c.t2d;
  ge_dbl(p, a, true);                            // 2
  ge_to_cached(c, p);  cached_store(tab, 2, c);
  ge_add_niels(q, p, n1, true);                  // 3
  ge_to_cached(c, q);  cached_store(tab, 3, c);
  ge_dbl(p, p, true);                            // 4
  ge_to_cached(c, p);  cached_store(tab, 4, c);
  ge_dbl(q, q, true);                            // 6
  ge_to_cached(c, q);  cached_store(tab, 6, c);
  ge_add_niels(q, q, n1, true);                  // 7
  ge_to_cached(c, q);  cached_store(tab, 7, c);
  ge_add_niels(q, p, n1, true);                  // 5
  ge_to_cached(c, q);  cached_store(tab, 5, c);
  ge_dbl(p, p, true);                            // 8
  ge_to_cached(c, p);  cached_store(tab, 8, c);
}
// -A and its multiples 0..8 in cached form at tab; returns whether A is on the curve
ED_DEV bool verify_table_lane(uint32_t* tab, const uint32_t aw[8]) {
  bool oncurve;
  ge a;
  ge_frombytes(a, oncurve, aw, true);
  verify_table_point_lane(tab, a);
  return oncurve;
}

// rw = R, sw = S (raw, replaced by the digit words), aw = A as little-endian words.
// Writes tw/sw digit words and the 9-entry table at tab; returns whether A is on the curve.
ED_DEV bool verify_prepare_lane(uint32_t tw[8], uint32_t sw[8], uint32_t* tab, const uint32_t rw[8],
                                const uint32_t aw[8], const uint8_t* m, size_t mlen) {
  verify_hash_lane(tw, rw, aw, m, mlen);
  verify_s_lane(sw);
  return verify_table_lane(tab, aw);
}

// digits: the 16 digit words of this item (t + 0x88.. in [0,8), S + 0x8000.. in [8,16)), read from
// memory window by window rather than held in 16 registers (that is what lets the kernel fit 128
// VGPRs without scratch); tab: this item's table; base16: the k*B table
ED_DEV void verify_main_lane(ge& acc, const uint32_t* digits, const uint32_t* tab, const uint32_t* base16) {
  ge_neutral(acc);
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {
    if (w != 63) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) ge_dbl(acc, acc, k == 3);
    }
    {
      const int dig = (int)((digits[w >> 3] >> (4 * (w & 7))) & 15u) - 8;
      const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
      ge_cached c;
      cached_load(c, tab, mag);
      ge_cached_cneg(c, dig < 0);
      ge_add_cached(acc, acc, c, (w & 3) == 0);
    }
    if ((w & 3) == 0) {
      const int j = w >> 2;                        // digit j of S sits at bit 16 j
      const int dig = (int)((digits[8 + (j >> 1)] >> (16 * (j & 1))) & 0xffffu) - 32768;
      const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
      ge_niels nb;
      niels_load(nb, base16 + TABLE_ENTRY_WORDS * mag);
      ge_niels_cneg(nb, dig < 0);
      ge_add_niels(acc, acc, nb, false);
    }
  }
}

// ed_export given zinv = 1/Z, then the byte comparison with R (ed25519-sha512.c:176-180)
ED_DEV bool verify_encode_lane(const fe& X, const fe& Y, const fe& zinv, const uint32_t rw[8]) {
  fe x, y;
  fe_mul(x, X, zinv);
  fe_mul(y, Y, zinv);
  uint32_t cw[8];
  fe_tobytes(cw, y);
  cw[7] |= fe_parity(x) << 31;
  uint32_t diff = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) diff |= cw[q] ^ rw[q];
  return diff == 0;
}

// ---------------------------------------------------------------------------------------------
// Ed25519 verify with half-length scalars (halve.h): the same verdicts from 132 doublings.
//     acc = 16*acc + sigma*d_i*(-A) + e_i*(-R')  [+ f_j*B + g_j*(2^128 B) when i = 4j, j < 8]
// over 34 four-bit windows, d = digits of v, e = digits of |u|, sigma = sign of u, f / g = the 16-bit digits
// of the low / high half of s' = |u| S mod l; accept iff the result is the neutral element and R was a
// canonical encoding.  An item for which halve.h finds no pair keeps (u, v) = (1, t) and s' = S: the same equation,
// read over 64 windows ("long"; f_j*B then for every j < 16 and no g); its wave runs the long loop, in which the
// short items of the wave add neutral elements from window 34 on.  About one wave in 180 is long.
// hd (HALF_DIGIT_WORDS per item): v + 0x88.. [0,8) | |u| + 0x88.. [8,16) | s' + 0x8000.. [16,24) |
// [24]: bit 0 = (u < 0), bit 1 = long, bit 2 = the search returned a pair that failed the exact check (diagnostic: never seen).
// ---------------------------------------------------------------------------------------------
#define HALF_DIGIT_WORDS 28
constexpr int HALF_LONG_WINDOWS = 64;

// tdig / sdig: the digit words k_verify_prepare wrote (t + 0x88.., S mod l + 0x8000..); BITS: halve.h
template <int BITS = HALF_BITS>
ED_DEV void verify_half_scalars_lane(uint32_t hd[HALF_DIGIT_WORDS], const uint32_t tdig[8], const uint32_t sdig[8]) {
  uint32_t tw[8], sw[8], vw[5], uw[5], v8[8], u8[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { tw[k] = tdig[k]; sw[k] = sdig[k]; }
  words_sub_pattern(tw, 0x88888888u);
  words_sub_pattern(sw, 0x80008000u);
  bool uneg;
  bool found = halve_scalar_lane<BITS>(vw, uw, uneg, tw);
  // The pair search takes its quotients from doubles under hand-derived error margins (halve.h); a wrong pair would be
  // a silent wrong verdict, so the congruence it promises is re-verified here with integers before anything depends
  // on it:  u t = v (mod 8 l)  <=>  W := |u| t - sign(u) v  is divisible by l and by 8 (l is odd).  W (13 words, >= 0
  // for a correct pair: v < 8 l) by a 5 x 8-word schoolbook product, W mod l by the Barrett reduction every scalar
  // goes through, W mod 8 from its low word.  A pair that fails is dropped like a give-up: the item keeps
  // (u, v) = (1, t), the reference's own equation.
  bool mismatch;
  {
    uint32_t W[16];
#pragma unroll
    for (int k = 0; k < 16; k++) W[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      uint64_t c = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        c += (uint64_t)uw[i] * tw[j] + W[i + j];  // < 2^64: (2^32 - 1)^2 + 2 (2^32 - 1)
        W[i + j] = (uint32_t)c;
        c >>= 32;
      }
      W[i + 8] = (uint32_t)c;
    }
    int64_t c = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
      c += (int64_t)W[k] + (k < 5 ? (uneg ? (int64_t)vw[k] : -(int64_t)vw[k]) : 0);
      W[k] = (uint32_t)c;
      c >>= 32;
    }
    sc wl;
    sc_from_words<16>(wl, W);                    // W mod l (meaningless when W < 0: c != 0 then)
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) nz |= wl.v[k];
    const bool holds = c == 0 && nz == 0 && (W[0] & 7u) == 0 && (uw[0] & 1u) != 0;
    mismatch = found && !holds;
    found = found && holds;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    v8[k] = found ? (k < 5 ? vw[k] : 0u) : tw[k];
    u8[k] = found ? (k < 5 ? uw[k] : 0u) : (k == 0 ? 1u : 0u);
  }
  uneg = uneg && found;
  sc s, u, su;
  sc_from_words<8>(s, sw);
  sc_from_words<8>(u, u8);
  sc_mul(su, s, u);                              // |u| S mod l
  sc_to_words(sw, su);
  words_add_pattern(sw, 0x80008000u);
  words_add_pattern(v8, 0x88888888u);            // v < 2^253, |u| < 2^134: no carry out of the 64 nibbles
  words_add_pattern(u8, 0x88888888u);
#pragma unroll
  for (int k = 0; k < 8; k++) { hd[k] = v8[k]; hd[8 + k] = u8[k]; hd[16 + k] = sw[k]; }
  hd[24] = (uneg ? 1u : 0u) | (found ? 0u : 2u) | (mismatch ? 4u : 0u);
  hd[25] = 0; hd[26] = 0; hd[27] = 0;
}

// Table of 0..8 times -R' at tab, R' the point R encodes.  Returns whether R is what ed_export can produce
// (ed.c:155-169: y < p, a curve point, and not x = 0 with the sign bit set): only then can the reference's byte
// comparison (ed25519-sha512.c:176-180) succeed, and then it is the comparison of the points.
ED_DEV bool verify_half_point_lane(uint32_t* tab, const uint32_t rw[8]) {
  bool oncurve;
  ge r;
  ge_frombytes(r, oncurve, rw, true);
  bool top = (rw[7] & 0x7fffffffu) == 0x7fffffffu && rw[0] >= 0xffffffedu;
#pragma unroll
  for (int k = 1; k < 7; k++) top = top && rw[k] == 0xffffffffu;
  const bool zero_x_signed = (rw[7] >> 31) != 0 && fe_iszero(r.X);
  verify_table_point_lane(tab, r);
  return oncurve && !top && !zero_x_signed;
}

// returns whether the combination is the neutral element.  WITH_LONG = false: the caller keeps long items out (their
// result here is meaningless) and the loop is the short one; true: long_loop says whether some item of the wave is
// long (wave-uniform; the caller's vote).  The two per-item table entries of a window are requested before the
// window's four doublings and consumed after them (the kernel runs two waves per SIMD, which is not enough to hide a
// miss to HBM per addition otherwise).
template <bool WITH_LONG, int WINDOWS = HALF_WINDOWS>
ED_DEV bool verify_half_main_lane(const uint32_t* hd, const uint32_t* tab_a, const uint32_t* tab_r, const uint32_t* base16,
                                  bool long_loop, int hs = 1, const uint32_t* neutral = nullptr) {   // hs: distance in words between consecutive digit words (1: as k_verify_halve wrote them); neutral: a table whose entry 0 every lane may read for a zero digit (one line for the whole chip instead of one per item), or none
  ge acc;
  ge_neutral(acc);
  const bool uneg = (hd[24 * hs] & 1u) != 0, is_long = WITH_LONG && (hd[24 * hs] & 2u) != 0;
  const int top = (WITH_LONG && long_loop ? HALF_LONG_WINDOWS : WINDOWS) - 1;
#pragma unroll 1
  for (int w = top; w >= 0; w--) {
    const int dv = (int)((hd[(w >> 3) * hs] >> (4 * (w & 7))) & 15u) - 8;
    const int du = (int)((hd[(8 + (w >> 3)) * hs] >> (4 * (w & 7))) & 15u) - 8;
    cached_raw ra, rr;
    cached_load_raw(ra, neutral && dv == 0 ? neutral : tab_a, (uint32_t)(dv < 0 ? -dv : dv));
    cached_load_raw(rr, neutral && du == 0 ? neutral : tab_r, (uint32_t)(du < 0 ? -du : du));
    const bool base_here = (w & 3) == 0 && (WITH_LONG || w < 32);
    if (w != top) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) ge_dbl(acc, acc, k == 3);
    }
    {
      ge_cached c;
      cached_from_raw(c, ra);
      ge_cached_cneg(c, (dv < 0) != uneg);       // sigma * d * (-A)
      ge_add_cached(acc, acc, c, true);
      cached_from_raw(c, rr);
      ge_cached_cneg(c, du < 0);                 // e * (-R')
      ge_add_cached(acc, acc, c, base_here);
    }
    if (base_here) {
      const int j = w >> 2;                      // digit j of s' sits at bit 16 j, digit 8 + j at bit 128 + 16 j
#pragma unroll 1
      for (int h = 0; h < (j < 8 ? 2 : 1); h++) {
        // short items: digit j from k*B and digit 8 + j from k*2^128*B, j < 8; long items: digit j from k*B, j < 16
        const int jj = j + 8 * h;
        int dig = (int)((hd[(16 + (jj >> 1)) * hs] >> (16 * (jj & 1))) & 0xffffu) - 32768;
        if (WITH_LONG) dig = (is_long ? h == 0 : j < 8) ? dig : 0;
        const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
        ge_niels nb;
        niels_load(nb, base16 + TABLE_ENTRY_WORDS * ((size_t)mag + (h ? (size_t)TABLE_BASE16_ENTRIES : 0)));
        ge_niels_cneg(nb, dig < 0);
        ge_add_niels(acc, acc, nb, h == 0 && j < 8);
      }
    }
  }
  fe d;
  fe_sub(d, acc.Y, acc.Z);
  return fe_iszero(acc.X) && fe_iszero(d) && !fe_iszero(acc.Z);
}

// ---------------------------------------------------------------------------------------------
// Exact path for public keys that are NOT on the curve (slow, rare, data-dependent control flow)
// ---------------------------------------------------------------------------------------------
// ed_import (ed.c:100-149) never fails: when neither candidate root fits it keeps j*beta and the
// reference then applies its addition formulas to a pair (x,y) that is not a curve point.  The
// result is no longer independent of the evaluation order, so for those keys -- and only those --
// the reference's own chain is replayed formula by formula: joint-sparse-form digits produced one
// 52-bit limb at a time (sc.c:297-324, 64-bit build), Shamir's trick over B, Q+B, Q-B and pc(Q)
// (ed.c:455-507) with ed_add / ed_sub / ed_double / ed_add_pc / ed_sub_pc exactly as written
// (ed.c:175-335).  All values are field elements, so the limb representation does not matter.

// ed.c:175-203 ed_add (sub = false) / ed.c:245-273 ed_sub (sub = true)
ED_DEV void ref_add(ge& o, const ge& p, const ge& q, bool sub) {
  fe a, b, c, d, e, f, g, h, qm, qp;
  fe_sub(qm, q.Y, q.X);                          // 3u
  fe_add(qp, q.Y, q.X);                          // 2u
  fe_sub(a, p.Y, p.X);
  fe_mul(a, a, sub ? qp : qm);
  fe_add(b, p.Y, p.X);
  fe_mul(b, b, sub ? qm : qp);
  fe_mul(c, p.T, q.T);
  fe_mul(c, c, fe_const_2d());                   // ed_sub multiplies by -2d: handled by swapping f and g
  fe_mul(d, p.Z, q.Z);
  fe_add(d, d, d);                               // 2u
  fe_sub(e, b, a);                               // 3u
  fe_sub(f, d, c);                               // 4u: first operand only
  fe_add(g, d, c);                               // 3u
  fe_add(h, b, a);                               // 2u
  if (sub) { fe t = f; f = g; g = t; fe_carry(f); fe_carry(g); }
  fe_mul(o.X, f, e);
  fe_mul(o.Y, h, g);
  fe_mul(o.T, h, e);
  fe_mul(o.Z, f, g);
}

// ed.c:211-237 ed_double: the addition law with P = Q (4 S + 5 M)
ED_DEV void ref_double(ge& o, const ge& p) {
  fe a, b, c, d, e, f, g, h;
  fe_sub(a, p.Y, p.X); fe_sq(a, a);
  fe_add(b, p.Y, p.X); fe_sq(b, b);
  fe_sq(c, p.T); fe_mul(c, c, fe_const_2d());
  fe_sq(d, p.Z); fe_add(d, d, d);                // 2u
  fe_sub(e, b, a);                               // 3u
  fe_sub(f, d, c);                               // 4u
  fe_add(g, d, c);                               // 3u
  fe_add(h, b, a);                               // 2u
  fe_mul(o.X, f, e);
  fe_mul(o.Y, h, g);
  fe_mul(o.T, h, e);
  fe_mul(o.Z, f, g);
}

// ed.c:282-305 ed_add_pc / ed.c:310-335 ed_sub_pc
ED_DEV void ref_add_pc(ge& o, const ge& p, const ge_niels& q, bool sub) {
  ge_niels qq = q;
  ge_niels_cneg(qq, sub);                        // (sum, diff, -prod): the same values ed_sub_pc uses
  ge_add_niels(o, p, qq, true);
}

// sc.c:272-281 jsfdigit
ED_DEV int ref_jsf_digit(uint64_t a, uint64_t b) {
  const int u = 2 - (int)(a & 3);
  if (u == 2) return 0;
  if (((a & 7) == 3 || (a & 7) == 5) && (b & 3) == 2) return -u;
  return u;
}

// limb i (52 bits) of a reduced scalar held as eight 32-bit words (sc.h:26 in the 64-bit build)
ED_DEV uint64_t ref_limb52(const uint32_t w[8], int i) {
  uint64_t v = 0;
  for (int b = 0; b < 52; b++) {
    const int bit = 52 * i + b;
    if (bit < 256) v |= (uint64_t)((w[bit >> 5] >> (bit & 31)) & 1u) << b;
  }
  return v;
}

constexpr int REF_JSF_LEN = 261;

// sc.c:297-324 sc_jsf, limb boundaries included (the look-ahead of jsfdigit does not cross them)
ED_DEV int ref_jsf(int8_t* u0, int8_t* u1, const uint32_t aw[8], const uint32_t bw[8]) {
  int64_t n0 = 0, n1 = 0;
  int k = 0;
  for (int i = 0; i < 5; i++) {
    n0 += (int64_t)ref_limb52(aw, i);
    n1 += (int64_t)ref_limb52(bw, i);
    for (int j = 0; j < 52; j++, k++) {
      u0[k] = (int8_t)ref_jsf_digit((uint64_t)n0, (uint64_t)n1);
      u1[k] = (int8_t)ref_jsf_digit((uint64_t)n1, (uint64_t)n0);
      n0 = (n0 - u0[k]) >> 1;
      n1 = (n1 - u1[k]) >> 1;
    }
  }
  u0[k] = (int8_t)ref_jsf_digit((uint64_t)n0, (uint64_t)n1);
  u1[k] = (int8_t)ref_jsf_digit((uint64_t)n1, (uint64_t)n0);
  while (k >= 0 && u0[k] == 0 && u1[k] == 0) k--;
  return k;
}

// the same recoder writing digit k at [k * stride]
ED_DEV int ref_jsf_strided(int8_t* u0, int8_t* u1, int stride, const uint32_t aw[8], const uint32_t bw[8]) {
  int64_t n0 = 0, n1 = 0;
  int k = 0;
  for (int i = 0; i < 5; i++) {
    n0 += (int64_t)ref_limb52(aw, i);
    n1 += (int64_t)ref_limb52(bw, i);
    for (int j = 0; j < 52; j++, k++) {
      const int d0 = ref_jsf_digit((uint64_t)n0, (uint64_t)n1), d1 = ref_jsf_digit((uint64_t)n1, (uint64_t)n0);
      u0[k * stride] = (int8_t)d0; u1[k * stride] = (int8_t)d1;
      n0 = (n0 - d0) >> 1;
      n1 = (n1 - d1) >> 1;
    }
  }
  const int d0 = ref_jsf_digit((uint64_t)n0, (uint64_t)n1), d1 = ref_jsf_digit((uint64_t)n1, (uint64_t)n0);
  u0[k * stride] = (int8_t)d0; u1[k * stride] = (int8_t)d1;
  while (k >= 0 && u0[k * stride] == 0 && u1[k * stride] == 0) k--;
  return k;
}

// ed.c:455-507 ed_dual_scale: r = x*B + y*Q for reduced scalars given as words, Q affine (z = 1),
// pcB = the base point in niels form (ed.c:46-52 pced_B)
ED_DEV void ref_dual_scale(ge& r, const uint32_t xw[8], const uint32_t yw[8], const ge& q,
                           const ge_niels& pcB) {
  int8_t ux[REF_JSF_LEN], uy[REF_JSF_LEN];
  ge_neutral(r);
  const int n = ref_jsf(ux, uy, xw, yw);
  if (n < 0) return;
  ge qpb, qmb;
  ge_niels pcq;
  ref_add_pc(qpb, q, pcB, false);
  ref_add_pc(qmb, q, pcB, true);
  fe_sub(pcq.ymx, q.Y, q.X); fe_carry(pcq.ymx);          // ed.c:436-442 ed_precompute
  fe_add(pcq.ypx, q.Y, q.X); fe_carry(pcq.ypx);
  fe_mul(pcq.t2d, q.T, fe_const_2d());
  for (int i = n;; i--) {
    const int a = ux[i], b = uy[i];
    if (a == 1) {
      if (b == 1) ref_add(r, r, qpb, false);
      else if (b == -1) ref_add(r, r, qmb, true);
      else ref_add_pc(r, r, pcB, false);
    } else if (a == -1) {
      if (b == 1) ref_add(r, r, qmb, false);
      else if (b == -1) ref_add(r, r, qpb, true);
      else ref_add_pc(r, r, pcB, true);
    } else if (b == 1) ref_add_pc(r, r, pcq, false);
    else if (b == -1) ref_add_pc(r, r, pcq, true);
    if (i == 0) break;
    ref_double(r, r);
  }
}

// The same chain with uniform control flow, which is what the GPU runs (64 lanes with 64 different
// digit strings would otherwise serialise the nine cases of every step).  Three facts make it
// produce the reference's field values:
//   * ed_sub(P, Q) and ed_add(P, -Q) compute the same field elements (ed.c:245-273 vs :175-203:
//     the sums are swapped and 2d is negated, which is what negating Q.x and Q.t does);
//   * ed_add_pc(P, pc(Q)) and ed_add(P, Q) agree when Q.z = 1 (2d*Q.t is merely multiplied in
//     earlier, and 2*P.z*1 = 2*P.z);
//   * all formulas are homogeneous in the accumulator, so the doublings of the neutral element
//     above a lane's top digit rescale (X:Y:Z:T) without changing the affine result, and a step
//     whose digits are both zero leaves the accumulator untouched (select).
// Per-lane storage supplied by the caller (LDS on the device, so that the kernel needs neither
// scratch memory nor more than 128 VGPRs and its waves fit the slots k_verify_main's waves free):
//   ux, uy : 2 x REF_JSF_LEN digits, element i at [i * stride]
//   pts    : the four loop-invariant addends Q, B, Q+B, Q-B as 4 x 40 words (X|Y|Z|T),
//            word j of addend a at [(40 a + j) * stride]
ED_DEV void ref_pt_store(uint32_t* pts, int stride, int a, const ge& p) {
#pragma unroll
  for (int j = 0; j < 10; j++) {
    pts[(40 * a + j) * stride] = p.X.v[j];      pts[(40 * a + 10 + j) * stride] = p.Y.v[j];
    pts[(40 * a + 20 + j) * stride] = p.Z.v[j]; pts[(40 * a + 30 + j) * stride] = p.T.v[j];
  }
}
ED_DEV void ref_pt_load(ge& p, const uint32_t* pts, int stride, int a) {
#pragma unroll
  for (int j = 0; j < 10; j++) {
    p.X.v[j] = pts[(40 * a + j) * stride];      p.Y.v[j] = pts[(40 * a + 10 + j) * stride];
    p.Z.v[j] = pts[(40 * a + 20 + j) * stride]; p.T.v[j] = pts[(40 * a + 30 + j) * stride];
  }
}

// step 1: recode the scalars and lay out the four addends (sc.c:297-324, ed.c:473-476)
ED_DEV void ref_dual_scale_setup(const uint32_t xw[8], const uint32_t yw[8], const ge& q, const ge_niels& pcB,
                                 int8_t* ux, int8_t* uy, uint32_t* pts, int stride) {
  const int n = ref_jsf_strided(ux, uy, stride, xw, yw);
  for (int i = n + 1; i < REF_JSF_LEN; i++) { ux[i * stride] = 0; uy[i * stride] = 0; }
  ge t;
  ref_pt_store(pts, stride, 0, q);
  ge_base(t);                                    // B as an extended point (x, y, 1, xy)
  ref_pt_store(pts, stride, 1, t);
  ref_add_pc(t, q, pcB, false);
  ref_pt_store(pts, stride, 2, t);               // Q + B
  ref_add_pc(t, q, pcB, true);
  ref_pt_store(pts, stride, 3, t);               // Q - B
}

// step 2: the Shamir loop of ed.c:479-506 over all REF_JSF_LEN digit positions
ED_DEV void ref_dual_scale_chain(ge& r, const int8_t* ux, const int8_t* uy, const uint32_t* pts, int stride) {
  ge_neutral(r);
#pragma unroll 1
  for (int i = REF_JSF_LEN - 1; i >= 0; i--) {
    const int da = ux[i * stride], db = uy[i * stride];
    const bool both = (da != 0) && (db != 0);
    const bool skip = (da == 0) && (db == 0);
    // which addend: 2 = Q+B (digits equal), 3 = Q-B (digits opposite), 1 = B, 0 = Q
    const int which = both ? (da == db ? 2 : 3) : (da != 0 ? 1 : 0);
    const bool neg = which == 2 ? (da < 0) : which == 3 ? (da > 0) : which == 1 ? (da < 0) : (db < 0);
    ge add, sum;
    ref_pt_load(add, pts, stride, which);
    fe nx, nt;
    fe_neg(nx, add.X); fe_carry(nx);
    fe_neg(nt, add.T); fe_carry(nt);
    fe_cmov(add.X, nx, neg); fe_cmov(add.T, nt, neg);
    ref_add(sum, r, add, false);
    fe_cmov(r.X, sum.X, !skip); fe_cmov(r.Y, sum.Y, !skip); fe_cmov(r.Z, sum.Z, !skip); fe_cmov(r.T, sum.T, !skip);
    if (i != 0) ref_double(r, r);
  }
}

ED_DEV void ref_dual_scale_uniform(ge& r, const uint32_t xw[8], const uint32_t yw[8], const ge& q,
                                   const ge_niels& pcB, int8_t* ux, int8_t* uy, uint32_t* pts, int stride) {
  ref_dual_scale_setup(xw, yw, q, pcB, ux, uy, pts, stride);
  ref_dual_scale_chain(r, ux, uy, pts, stride);
}

// ed25519-sha512.c:148-175 in the reference's own order of operations, first half: hash, scalars,
// import of -A, digits and addends into the scratchpad.  base1 = entry 1 of the k*B table.
ED_DEV void verify_exact_setup_lane(const uint32_t rw[8], const uint32_t sraw[8], const uint32_t aw[8],
                                    const uint8_t* m, size_t mlen, const uint32_t* base1, int8_t* ux,
                                    int8_t* uy, uint32_t* pts, int stride) {
  uint32_t pre[16], dig[16], tw[8], sw[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { pre[k] = rw[k]; pre[8 + k] = aw[k]; }
  sha512_prefix_msg<16>(dig, pre, m, mlen);
  sc t, s;
  sc_from_words<16>(t, dig);
  sc_from_words<8>(s, sraw);
  sc_to_words(tw, t);
  sc_to_words(sw, s);
  bool oncurve;
  ge a;
  ge_frombytes(a, oncurve, aw, true);            // -A, as ed25519-sha512.c:174-175
  ge_niels pcB;
  niels_load(pcB, base1);
  ref_dual_scale_setup(sw, tw, a, pcB, ux, uy, pts, stride);
}

// The same first half from what k_verify_prepare already left in the workspace -- the digit words
// (t + 0x88..8, S + 0x8000..) and entry 1 of the item's table, -A in cached form (y-x, y+x, 2d*t, 2) --
// instead of hashing, reducing and decompressing again.  The point is rebuilt with all four
// coordinates doubled: (2x, 2y, 2, 2t) = (ypx - ymx, ypx + ymx, z2, t2d / d); every formula of the
// chain is homogeneous in each operand (of degree 2 in the accumulator and in an extended addend),
// so a common factor of an addend only rescales (X : Y : Z : T) and the affine result, all that is
// encoded, is the reference's.
ED_DEV void verify_exact_setup_reuse_lane(const uint32_t* digits, const uint32_t* tab, const uint32_t* base1,
                                          int8_t* ux, int8_t* uy, uint32_t* pts, int stride) {
  uint32_t tw[8], sw[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { tw[k] = digits[k]; sw[k] = digits[8 + k]; }
  words_sub_pattern(tw, 0x88888888u);
  words_sub_pattern(sw, 0x80008000u);
  ge_cached c1;
  cached_load(c1, tab, 1);
  ge a;
  fe_sub(a.X, c1.ypx, c1.ymx); fe_carry(a.X);
  fe_add(a.Y, c1.ypx, c1.ymx); fe_carry(a.Y);
  a.Z = c1.z2;
  fe_mul(a.T, c1.t2d, fe_const_inv_d());
  ge_niels pcB;
  niels_load(pcB, base1);
  ref_dual_scale_setup(sw, tw, a, pcB, ux, uy, pts, stride);
}

// second half (ed25519-sha512.c:176-180): the chain, ed_export, comparison with R
ED_DEV bool verify_exact_chain_lane(const uint32_t rw[8], const int8_t* ux, const int8_t* uy,
                                    const uint32_t* pts, int stride) {
  ge c;
  ref_dual_scale_chain(c, ux, uy, pts, stride);
  uint32_t cw[8];
  ge_tobytes(cw, c);                             // fld_inv(0) = 0 as in the reference
  uint32_t diff = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) diff |= cw[k] ^ rw[k];
  return diff == 0;
}

ED_DEV bool verify_exact_lane(const uint32_t rw[8], const uint32_t sraw[8], const uint32_t aw[8],
                              const uint8_t* m, size_t mlen, const uint32_t* base1, int8_t* ux, int8_t* uy,
                              uint32_t* pts, int stride) {
  verify_exact_setup_lane(rw, sraw, aw, m, mlen, base1, ux, uy, pts, stride);
  return verify_exact_chain_lane(rw, ux, uy, pts, stride);
}

// ---------------------------------------------------------------------------------------------
// The exact path as a THROUGHPUT kernel: one lane per item, for long work lists (a caller can send nothing but keys
// that are no curve points: ed_import never fails, ed.c:100-149).  Four lanes per item (quad_lanes.h) answer a short
// list soonest, but a wave of 16 items issues 300 k instructions - 1.2 M lane slots per item; one lane per item
// needs about 600 k.  Everything the chain reads is laid out by a set-up step in the item's OWN workspace, in the form
// the windowed kernels use for their tables (one packed 128-byte entry per lookup, cached_store):
//   tab entry 1      -Q = what k_verify_prepare left: (y-x, y+x, 2d t, 2) of the imported key, negated - the
//                    reference's ed_add_pc operand pc(Q) (ed.c:436-442) with 2 z = 2
//   tab entry 2, 3   Q + B, Q - B (ed.c:473-476) in cached form (scaled by a common factor, see
//                    verify_exact_setup_reuse_lane: every formula of the chain is homogeneous in the addend)
//   bentry           B in cached form, one entry shared by all items (EXACT_BENTRY_WORD of the work-list counters)
//   dig              the joint sparse form of (S, t), EXACT_DIGIT_WORDS words, one nibble per step
// and a step of the chain is ge_add_cached (the values of ed.c:175-203 ed_add / :282-305 ed_add_pc with the factors
// 2d t2 and 2 z2 multiplied in earlier; negating the addend swaps y-x with y+x and negates 2d t2: ed_sub / ed_sub_pc,
// ed.c:245-273, :310-335) followed by ref_double (ed.c:211-237), with the uniform control flow of ref_dual_scale_chain.
// ---------------------------------------------------------------------------------------------
#define EXACT_DIGIT_WORDS 33                           /* REF_JSF_LEN nibbles, eight per word */

// sc.c:272-281 jsfdigit without branches, from the low three bits of a and the low two of b: 0 for even a, else
// 2 - (a mod 4), negated when a mod 8 is 3 or 5 and b mod 4 is 2
ED_DEV int exact_jsf_digit(uint32_t a, uint32_t b) {
  const int u = (int)(a & 1u) - 2 * (int)(a & (a >> 1) & 1u);
  const uint32_t flip = a & ((a >> 1) ^ (a >> 2)) & (b >> 1) & ~b & 1u;
  return flip ? -u : u;
}

// limb I (52 bits) of a reduced scalar held as eight 32-bit words (sc.h:26 in the 64-bit build); I is a constant, so
// this is a shift or two (ref_limb52 walks the bits: it serves the host build and the layer probe)
template <int I>
ED_DEV uint64_t exact_limb52(const uint32_t w[8]) {
  constexpr int lo = 52 * I, k = lo >> 5, sh = lo & 31;
  uint64_t v = ((uint64_t)w[k] | ((uint64_t)(k + 1 < 8 ? w[k + 1 < 8 ? k + 1 : 7] : 0u) << 32)) >> sh;
  if (sh + 52 > 64 && k + 2 < 8) v |= (uint64_t)w[k + 2 < 8 ? k + 2 : 7] << (64 - sh);
  return v & (((uint64_t)1 << 52) - 1);
}

// sc.c:297-324 sc_jsf of (a, b), limb boundaries included (a limb joins the running value only when its turn comes, so the
// look-ahead of jsfdigit does not see across the boundary): step k as nibble (u0 + 1) | (u1 + 1) << 2 of word k / 8 at
// dig[(k / 8) * stride]; the nibbles past step 260 read "no digit"
template <int I>
ED_DEV void exact_jsf_limb(uint32_t* dig, int stride, uint64_t& n0, uint64_t& n1, uint32_t& word, const uint32_t aw[8], const uint32_t bw[8]) {
  n0 += exact_limb52<I>(aw);                     // (the running values stay non-negative: a digit of 1 is taken from an odd value)
  n1 += exact_limb52<I>(bw);
#pragma unroll 1
  for (int j = 0; j < 52; j++) {
    const int k = 52 * I + j;
    const int d0 = exact_jsf_digit((uint32_t)n0, (uint32_t)n1), d1 = exact_jsf_digit((uint32_t)n1, (uint32_t)n0);
    n0 = (uint64_t)((int64_t)(n0 - (uint64_t)(int64_t)d0) >> 1);
    n1 = (uint64_t)((int64_t)(n1 - (uint64_t)(int64_t)d1) >> 1);
    word |= (uint32_t)((d0 + 1) | ((d1 + 1) << 2)) << (4 * (k & 7));
    if ((k & 7) == 7) { dig[(k >> 3) * stride] = word; word = 0; }
  }
}
ED_DEV void exact_jsf_words(uint32_t* dig, int stride, const uint32_t aw[8], const uint32_t bw[8]) {
  uint64_t n0 = 0, n1 = 0;
  uint32_t word = 0;
  exact_jsf_limb<0>(dig, stride, n0, n1, word, aw, bw);
  exact_jsf_limb<1>(dig, stride, n0, n1, word, aw, bw);
  exact_jsf_limb<2>(dig, stride, n0, n1, word, aw, bw);
  exact_jsf_limb<3>(dig, stride, n0, n1, word, aw, bw);
  exact_jsf_limb<4>(dig, stride, n0, n1, word, aw, bw);
  {                                              // step 260 (sc.c:319-320), then "no digit" up to the end of the word
    constexpr int k = 260;
    const int d0 = exact_jsf_digit((uint32_t)n0, (uint32_t)n1), d1 = exact_jsf_digit((uint32_t)n1, (uint32_t)n0);
    word |= (uint32_t)((d0 + 1) | ((d1 + 1) << 2)) << (4 * (k & 7));
    for (int z = (k & 7) + 1; z < 8; z++) word |= 5u << (4 * z);
    dig[(k >> 3) * stride] = word;
  }
}

// B in cached form (ed.c:46-52 pced_B with 2 z = 2), packed: the shared entry
ED_DEV void exact_bentry_store(uint32_t* bentry) {
  ge b;
  ge_cached c;
  ge_base(b);
  ge_to_cached(c, b);
  cached_store(bentry, 0, c);
}

// set-up of one item from what k_verify_prepare left (digit words, entry 1 of the item's table): the digit string to
// dig, Q + B and Q - B to entries 2 and 3 of the item's table (its multiples of -A beyond the first are not needed:
// the windowed evaluation's result for this item is discarded)
ED_DEV void verify_exact_setup_table_lane(uint32_t* tab, uint32_t* dig, int dstride, const uint32_t* digits, const uint32_t* base1) {
  uint32_t tw[8], sw[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { tw[k] = digits[k]; sw[k] = digits[8 + k]; }
  words_sub_pattern(tw, 0x88888888u);
  words_sub_pattern(sw, 0x80008000u);
  exact_jsf_words(dig, dstride, sw, tw);
  ge_cached c1, c;
  cached_load(c1, tab, 1);
  ge qp, p;                                      // Q = -A with every coordinate doubled (verify_exact_setup_reuse_lane)
  fe_sub(qp.X, c1.ypx, c1.ymx); fe_carry(qp.X);
  fe_add(qp.Y, c1.ypx, c1.ymx); fe_carry(qp.Y);
  qp.Z = c1.z2;
  fe_mul(qp.T, c1.t2d, fe_const_inv_d());
  ge_niels pcB;
  niels_load(pcB, base1);
  ref_add_pc(p, qp, pcB, false);                 // Q + B
  ge_to_cached(c, p);
  cached_store(tab, 2, c);
  ref_add_pc(p, qp, pcB, true);                  // Q - B
  ge_to_cached(c, p);
  cached_store(tab, 3, c);
}

// which entry step `nib` adds, and with which sign (ed.c:480-501): 0 = Q, 1 = B, 2 = Q+B, 3 = Q-B
struct exact_step { int which; bool neg, skip; };
ED_DEV exact_step exact_step_of(uint32_t nib) {
  const int da = (int)(nib & 3u) - 1, db = (int)(nib >> 2) - 1;
  const bool both = (da != 0) && (db != 0);
  exact_step s;
  s.skip = (da == 0) && (db == 0);
  s.which = both ? (da == db ? 2 : 3) : (da != 0 ? 1 : 0);
  s.neg = s.which == 2 ? (da < 0) : s.which == 3 ? (da > 0) : s.which == 1 ? (da < 0) : (db < 0);
  return s;
}
ED_DEV const uint32_t* exact_entry_of(const exact_step& s, const uint32_t* tab, const uint32_t* bentry) {
  // a step without digits reads the shared entry (its sum is discarded): one line for the chip, not one per item
  return (s.skip || s.which == 1) ? bentry : tab + (s.which == 0 ? 1 : s.which) * VERIFY_ENTRY_WORDS;
}

// ed.c:479-506, steps hi .. lo of the chain (260 .. 0 is all of it): r = the accumulator on entry and on exit.  The entry of a
// step is requested one step ahead (its latency hides behind the current step's eighteen multiplications).
ED_DEV void exact_chain_steps(ge& r, const uint32_t* tab, const uint32_t* bentry, const uint32_t* dig, int dstride, int hi, int lo) {
  int i = hi;
  uint32_t w = dig[(i >> 3) * dstride];
  exact_step st = exact_step_of((w >> (4 * (i & 7))) & 15u);
  cached_raw raw;
  cached_load_raw(raw, exact_entry_of(st, tab, bentry), 0);
#pragma unroll 1
  for (;;) {
    const exact_step cur = st;
    ge_cached c;
    cached_from_raw(c, raw);
    if (i > lo) {
      const int j = i - 1;
      if ((j & 7) == 7) w = dig[(j >> 3) * dstride];
      st = exact_step_of((w >> (4 * (j & 7))) & 15u);
      cached_load_raw(raw, exact_entry_of(st, tab, bentry), 0);
    }
    ge sum;
    ge_cached_cneg(c, cur.neg);
    ge_add_cached(sum, r, c, true);
    fe_cmov(r.X, sum.X, !cur.skip); fe_cmov(r.Y, sum.Y, !cur.skip); fe_cmov(r.Z, sum.Z, !cur.skip); fe_cmov(r.T, sum.T, !cur.skip);
    if (i != 0) ref_double(r, r);                // ed.c:503-505: no doubling after the last digit
    if (i == lo) break;
    i--;
  }
}

// The chain in EXACT_SEGS stretches of steps, so that a wave's unit of work is a quarter of a chain: the kernel hands the
// (stretch, tile) units out one by one and the pass ends within a stretch's time of its last unit instead of a chain's
// (kernels.hip: k_verify_exact_lane_chain).  The accumulator crosses from one stretch to the next through the item's
// own workspace.
constexpr int EXACT_SEGS = 4;
ED_DEV constexpr int exact_seg_hi(int seg) { return REF_JSF_LEN - 1 - seg * ((REF_JSF_LEN + EXACT_SEGS - 1) / EXACT_SEGS); }
ED_DEV constexpr int exact_seg_lo(int seg) { return seg == EXACT_SEGS - 1 ? 0 : exact_seg_hi(seg + 1) + 1; }
#define EXACT_STATE_WORDS 40                           /* X | Y | Z | T, ten limbs each */
ED_DEV void exact_state_store(uint32_t* st, const ge& r) {
  word4* p = reinterpret_cast<word4*>(st);
  const fe* f[4] = {&r.X, &r.Y, &r.Z, &r.T};
  uint32_t w[EXACT_STATE_WORDS];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int j = 0; j < 10; j++) w[10 * k + j] = f[k]->v[j];
#pragma unroll
  for (int q = 0; q < EXACT_STATE_WORDS / 4; q++) p[q] = word4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
}
ED_DEV void exact_state_load(ge& r, const uint32_t* st) {
  const word4* p = reinterpret_cast<const word4*>(st);
  fe* f[4] = {&r.X, &r.Y, &r.Z, &r.T};
  uint32_t w[EXACT_STATE_WORDS];
#pragma unroll
  for (int q = 0; q < EXACT_STATE_WORDS / 4; q++) { const word4 v = p[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int j = 0; j < 10; j++) f[k]->v[j] = w[10 * k + j];
}

// ed_export + the byte comparison (ed25519-sha512.c:176-180)
ED_DEV bool exact_chain_verdict(const ge& r, const uint32_t rw[8]) {
  uint32_t cw[8];
  ge_tobytes(cw, r);                             // fld_inv(0) = 0 as in the reference
  uint32_t diff = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) diff |= cw[k] ^ rw[k];
  return diff == 0;
}

// one stretch of one item: state = the accumulator between stretches (EXACT_STATE_WORDS words of the item's workspace);
// returns the verdict after the last stretch (false before)
ED_DEV bool verify_exact_chain_segment_lane(int seg, const uint32_t rw[8], const uint32_t* tab, const uint32_t* bentry,
                                            const uint32_t* dig, int dstride, uint32_t* state, bool live) {
  ge r;
  if (seg == 0) ge_neutral(r); else exact_state_load(r, state);
  // (the bounds are runtime values on purpose: one copy of the loop, whatever the stretch)
  const int hi = seg == 0 ? exact_seg_hi(0) : seg == 1 ? exact_seg_hi(1) : seg == 2 ? exact_seg_hi(2) : exact_seg_hi(3);
  const int lo = seg == 0 ? exact_seg_lo(0) : seg == 1 ? exact_seg_lo(1) : seg == 2 ? exact_seg_lo(2) : exact_seg_lo(3);
  static_assert(EXACT_SEGS == 4, "the selects above list the stretches");
  exact_chain_steps(r, tab, bentry, dig, dstride, hi, lo);
  if (seg != EXACT_SEGS - 1) { if (live) exact_state_store(state, r); return false; }
  return exact_chain_verdict(r, rw);
}

// ---------------------------------------------------------------------------------------------
// TWO items per lane.  A joint sparse form has a digit pair in every second step, and a wave of 64 items has one in every
// step: run step by step, the chain's addition phase (8 M of the step's 13 M + 4 S) does useful work in half of its lanes.  Here a
// lane carries two items, each at its own step, and an iteration is ONE addition phase - for the item with an addition
// pending; of two, the one further behind - followed by a doubling phase per item: an item moves one step per iteration unless
// it loses the addition to its neighbour (a quarter of the iterations), ~300 iterations of 2 D + 1 A for two chains instead of
// 2 x 261 of D + A.  Same formulas on the same values in the same order per item (ed.c:479-506), so the same bytes.
// ---------------------------------------------------------------------------------------------
struct exact_walk { int i; bool pend; };         // i: the step whose addition (pend) or doubling comes next; -1: the chain is done

ED_DEV uint32_t exact_nibble(const uint32_t* dig, int dstride, int i) {
  const int j = i < 0 ? 0 : i;
  return (dig[(j >> 3) * dstride] >> (4 * (j & 7))) & 15u;
}
ED_DEV void ge_select(ge& r, const ge& v, bool flag) {
  fe_cmov(r.X, v.X, flag); fe_cmov(r.Y, v.Y, flag); fe_cmov(r.Z, v.Z, flag); fe_cmov(r.T, v.T, flag);
}
// the walk of an item before its first iteration: at the top digit, its addition pending if the digit pair is not (0, 0)
ED_DEV exact_walk exact_walk_start(const uint32_t* dig, int dstride, bool live) {
  exact_walk w;
  w.i = live ? REF_JSF_LEN - 1 : -1;
  w.pend = live && exact_nibble(dig, dstride, REF_JSF_LEN - 1) != 5u;
  return w;
}
// which item the next addition phase serves: 0 none, 1 the first, 2 the second
ED_DEV int exact_pair_choice(const exact_walk& a, const exact_walk& b) {
  return a.pend && b.pend ? (a.i >= b.i ? 1 : 2) : a.pend ? 1 : b.pend ? 2 : 0;
}
// r += (neg ? -q : q), q a packed cached entry as loaded: ge_cached_cneg + ge_add_cached with the entry's four factors unpacked
// one at a time, each right before its multiplication (forty registers less alive than with the entry unpacked up front)
ED_DEV void ge_add_raw(ge& r, const cached_raw& raw, bool neg) {
  fe a, b, c, d, e, f, g, h, m;
  uint32_t w[8];
  // negating the addend swaps y-x with y+x and negates 2d t (ed.c:245-273)
#define RAW_WORDS(k) { w[0] = raw.q[2 * (k)].x; w[1] = raw.q[2 * (k)].y; w[2] = raw.q[2 * (k)].z; w[3] = raw.q[2 * (k)].w; \
                       w[4] = raw.q[2 * (k) + 1].x; w[5] = raw.q[2 * (k) + 1].y; w[6] = raw.q[2 * (k) + 1].z; w[7] = raw.q[2 * (k) + 1].w; }
#define RAW_WORDS_SEL(k0, k1) { w[0] = neg ? raw.q[2 * (k1)].x : raw.q[2 * (k0)].x; w[1] = neg ? raw.q[2 * (k1)].y : raw.q[2 * (k0)].y; \
                                w[2] = neg ? raw.q[2 * (k1)].z : raw.q[2 * (k0)].z; w[3] = neg ? raw.q[2 * (k1)].w : raw.q[2 * (k0)].w; \
                                w[4] = neg ? raw.q[2 * (k1) + 1].x : raw.q[2 * (k0) + 1].x; w[5] = neg ? raw.q[2 * (k1) + 1].y : raw.q[2 * (k0) + 1].y; \
                                w[6] = neg ? raw.q[2 * (k1) + 1].z : raw.q[2 * (k0) + 1].z; w[7] = neg ? raw.q[2 * (k1) + 1].w : raw.q[2 * (k0) + 1].w; }
  RAW_WORDS_SEL(0, 1)
  fe_unpack(m, w);
  fe_sub(a, r.Y, r.X);                          // 3u
  fe_mul(a, a, m);
  RAW_WORDS_SEL(1, 0)
  fe_unpack(m, w);
  fe_add(b, r.Y, r.X);                          // 2u
  fe_mul(b, b, m);
  RAW_WORDS(2)
  fe_unpack(m, w);
  { fe n; fe_neg(n, m); fe_cmov(m, n, neg); }   // < 2u: fine as a second operand
  fe_mul(c, r.T, m);
  RAW_WORDS(3)
  fe_unpack(m, w);
  fe_mul(d, r.Z, m);
#undef RAW_WORDS
#undef RAW_WORDS_SEL
  fe_sub(e, b, a);                              // 3u
  fe_sub(f, d, c);                              // 3u
  fe_add(g, d, c);                              // 2u
  fe_add(h, b, a);                              // 2u
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, g, f);
  fe_mul(r.T, e, h);
}

ED_DEV void ge_cswap(ge& a, ge& b, bool flag) {
  fe_cswap(a.X, b.X, flag); fe_cswap(a.Y, b.Y, flag); fe_cswap(a.Z, b.Z, flag); fe_cswap(a.T, b.T, flag);
}
// `iters` iterations (iters < 0: until both items of every lane of the wave are done; the host build: of this lane).
// An item is named by its slot number: its table is at tabs + slot * slot_words, its digit string at digs + slot * slot_words
// (two 32-bit numbers per lane instead of four pointers: the kernel has no register to spare).
// The item the addition phase serves is brought into the FIRST register set by a conditional exchange of the two sets (and of
// their walks and slot numbers) and stays there: one exchange per iteration instead of a copy in and a copy out, and no third
// accumulator alive during the addition.  The sets are put back at the end.
ED_DEV void exact_pair_iterations(ge& ra, exact_walk& wa, ge& rb, exact_walk& wb, const uint32_t* tabs, const uint32_t* digs,
                                  uint32_t slot_words, uint32_t item_a, uint32_t item_b, const uint32_t* bentry, int iters) {
#define EXACT_TAB(item) (tabs + (size_t)(item) * slot_words)
#define EXACT_DIG(item) (digs + (size_t)(item) * slot_words)
  bool flipped = false;
  int sel = exact_pair_choice(wa, wb);
  exact_step st = exact_step_of(sel == 2 ? exact_nibble(EXACT_DIG(item_b), 1, wb.i) : exact_nibble(EXACT_DIG(item_a), 1, wa.i));
  st.skip = st.skip || sel == 0;
  const uint32_t* entry = exact_entry_of(st, EXACT_TAB(sel == 2 ? item_b : item_a), bentry);
  struct { word4 q[4]; } half;
  {
    const word4* e4 = reinterpret_cast<const word4*>(entry);
#pragma unroll
    for (int q = 0; q < 4; q++) half.q[q] = e4[q];
  }
#pragma unroll 1
  for (int it = 0;; it++) {
#ifdef ED_HOST_CHECK
    if (iters >= 0 ? it == iters : (wa.i < 0 && wb.i < 0)) break;
#else
    if (iters >= 0 ? it == iters : !__any(wa.i >= 0 || wb.i >= 0)) break;
#endif
    {                                            // the addition phase (ed.c:480-501) for the chosen item: in the first set
      // the second half of the entry (2d t, 2 z): requested now, used two multiplications further down
      cached_raw raw;
      {
        const word4* e4 = reinterpret_cast<const word4*>(entry);
#pragma unroll
        for (int q = 0; q < 4; q++) { raw.q[q] = half.q[q]; raw.q[4 + q] = e4[4 + q]; }
      }
      const bool x = sel == 2;
      ge_cswap(ra, rb, x);
      { const exact_walk t = wa; wa = x ? wb : wa; wb = x ? t : wb; }
      { const uint32_t t = item_a; item_a = x ? item_b : item_a; item_b = x ? t : item_b; }
      flipped = flipped != x;
      // (a branch, not a select: the lanes without an addition sit the phase out under the execution mask and their
      // accumulator needs no second copy - with selects here and below the kernel spilled 58 registers)
      if (sel != 0) ge_add_raw(ra, raw, st.neg);
      wa.pend = wa.pend && sel == 0;
    }
    // what the doubling phases below do, and whom the NEXT addition phase serves, follow from the digits alone
    const bool go_a = wa.i >= 0 && !wa.pend, go_b = wb.i >= 0 && !wb.pend;       // the item leaves its step in this iteration
    const bool dbl_a = go_a && wa.i > 0, dbl_b = go_b && wb.i > 0;               // ed.c:503-505: no doubling after the last digit
    if (go_a) { wa.i = wa.i - 1; wa.pend = wa.i >= 0 && exact_nibble(EXACT_DIG(item_a), 1, wa.i) != 5u; }
    if (go_b) { wb.i = wb.i - 1; wb.pend = wb.i >= 0 && exact_nibble(EXACT_DIG(item_b), 1, wb.i) != 5u; }
    sel = exact_pair_choice(wa, wb);
    st = exact_step_of(sel == 2 ? exact_nibble(EXACT_DIG(item_b), 1, wb.i) : exact_nibble(EXACT_DIG(item_a), 1, wa.i));
    st.skip = st.skip || sel == 0;
    entry = exact_entry_of(st, EXACT_TAB(sel == 2 ? item_b : item_a), bentry);
    // The FIRST half of the next entry (y-x, y+x: 16 registers) is requested between the two doublings and arrives behind the
    // second; the whole entry this early (32 registers across a doubling) left the kernel 10 registers short, and none of it
    // (a load in front of the addition) cost 5 % of VALU-busy.
    if (dbl_a) ref_double(ra, ra);
    {
      const word4* e4 = reinterpret_cast<const word4*>(entry);
#pragma unroll
      for (int q = 0; q < 4; q++) half.q[q] = e4[q];
    }
    if (dbl_b) ref_double(rb, rb);
  }
  ge_cswap(ra, rb, flipped);
  { const exact_walk t = wa; wa = flipped ? wb : wa; wb = flipped ? t : wb; }
#undef EXACT_TAB
#undef EXACT_DIG
}

// the accumulator and the walk of an item between two units of work: EXACT_STATE_WORDS + 2 words of its workspace
ED_DEV void exact_walk_store(uint32_t* st, const ge& r, const exact_walk& w) {
  exact_state_store(st, r);
  st[EXACT_STATE_WORDS] = (uint32_t)w.i; st[EXACT_STATE_WORDS + 1] = w.pend ? 1u : 0u;
}
ED_DEV void exact_walk_load(ge& r, exact_walk& w, const uint32_t* st) {
  exact_state_load(r, st);
  w.i = (int)st[EXACT_STATE_WORDS]; w.pend = st[EXACT_STATE_WORDS + 1] != 0;
}

// ed_export of both results on ONE inversion (ed.c:155-169; fld_inv(0) = 0: a zero Z contributes 1 to the shared product and gets the
// inverse 0 by hand), then the byte comparisons (ed25519-sha512.c:176-180)
ED_DEV void exact_pair_verdicts(bool& same_a, bool& same_b, const ge& ra, const ge& rb, const uint32_t rwa[8], const uint32_t rwb[8]) {
  fe one, zero, za = ra.Z, zb = rb.Z, p, u, zia, zib;
  fe_set(one, 1); fe_set(zero, 0);
  const bool nza = !fe_iszero(za), nzb = !fe_iszero(zb);
  fe_cmov(za, one, !nza); fe_cmov(zb, one, !nzb);
  fe_mul(p, za, zb);
  fe_inv(u, p);
  fe_mul(zia, u, zb);
  fe_mul(zib, u, za);
  fe_cmov(zia, zero, !nza); fe_cmov(zib, zero, !nzb);
  same_a = verify_encode_lane(ra.X, ra.Y, zia, rwa);
  same_b = verify_encode_lane(rb.X, rb.Y, zib, rwb);
}

// the iterations of a unit of work: the first EXACT_SEGS - 1 units of a tile run this many, the last one runs to the end
constexpr int EXACT_PAIR_ITERS = 75;

// the whole chain of one item in one go (host build; the kernels run it stretch by stretch)
ED_DEV bool verify_exact_chain_table_lane(const uint32_t rw[8], const uint32_t* tab, const uint32_t* bentry,
                                          const uint32_t* dig, int dstride) {
  ge r;
  ge_neutral(r);
  exact_chain_steps(r, tab, bentry, dig, dstride, REF_JSF_LEN - 1, 0);
  return exact_chain_verdict(r, rw);
}

// ---------------------------------------------------------------------------------------------
// fixed-base path: ed.c:346-430 (scale16, ed_scale_base) and its callers
// ---------------------------------------------------------------------------------------------
// The reference's comb (ed.c:397-430) with a wider window: x + offset is cut into COMB_DIGITS
// signed COMB_W-bit digits d_j in [-COMB_HALF, COMB_HALF - 1] (ed.c:407-409's recoding: add
// COMB_HALF at every digit position, then subtract it from every digit); even digits accumulate
// in R0, odd digits in R1, both from row i = j / 2 of the table comb[i][k] = (k+1) * 2^(2 w i) * B;
// then R1 <- 2^w R1 and out = R0 + R1.  The reference has w = 4 (64 digits, 32 rows of 8); here
// w = 6: 44 digits, 22 rows of 32, i.e. 44 mixed additions instead of 64 for two more doublings
// (round 1 had w = 5, 52 additions).  x * B is the same group element either way and only its affine
// encoding leaves the kernels (the addition law of ed.c:282-305 is complete on this curve), so the bytes
// are the reference's.
// The scalar is secret here, so the lookup keeps the reference's constant-time discipline
// (ed.c:359-390): no memory address and no branch depends on the digit.  On the device the table
// is staged in LDS as COMB_HALF entries per row, m * 2^(2 w i) * B for m = 1 .. COMB_HALF
// (`comb_image_entry_lane`: 22 x 32 x 144 bytes = 99 KiB, one 512-lane block per CU);
// lane L of every wave reads entry L mod 32 -- an address that depends on the lane number only -- and
// each lane then takes the entry it needs from lane |d| - 1 of its own wave with ds_bpermute_b32, the
// cross-lane shuffle of the LDS crossbar (30 per lookup, on the LDS pipe, beside the VALU work of the
// previous addition), replaces it by the neutral element when d = 0 and applies the digit's sign, both
// with selects in registers (ed.c:383-389: swap y-x with y+x, negate 2dxy).  Only lanes 0..31 are ever
// sources: the shuffle's crossbar has 32 banks for this purpose, and a 33rd source lane (an entry for
// m = 0 in lane 32) measurably collided with lane 0 whenever digits 0 and -32 met in one wave
// (SQ_LDS_BANK_CONFLICT 45 k for constant secrets, 430 k for random ones: tests/test_gpu_ct.py caught it).
// With w = 5 the image held both signs (2 x 16 entries per row) and needed no selects; with w = 6 both
// signs would not fit the CU's 160 KiB of LDS, and 8 fewer additions are worth more than 44 x 70 selects.
// The first version scanned all eight entries of a w = 4 row with v_cndmask: 240 selects + 40 for
// the conditional negation per lookup, 15 % of the kernel's instructions.

// entry m - 1 of the image of comb row `row` (comb = the table in its global layout): m * 2^(2 w row) * B
ED_DEV void comb_image_entry_lane(uint32_t* dst, const uint32_t* comb, int row, int m1) {
  ge_niels e;
  niels_load(e, comb + TABLE_ENTRY_WORDS * (COMB_HALF * row + m1));
#pragma unroll
  for (int j = 0; j < 10; j++) { dst[j] = e.ymx.v[j]; dst[10 + j] = e.ypx.v[j]; dst[20 + j] = e.t2d.v[j]; }
#pragma unroll
  for (int j = 30; j < COMB_IMG_ENTRY_WORDS; j++) dst[j] = 0;
}

#ifdef ED_HOST_CHECK
// host build: `table` is the comb in its global layout [COMB_ROWS][COMB_HALF]; plain indexed lookup
ED_DEV void comb_select(ge_niels& e, const uint32_t* table, int row, uint32_t digit) {
  const int d = (int)digit - COMB_HALF, mag = d < 0 ? -d : d;
  fe_set(e.ymx, 1); fe_set(e.ypx, 1); fe_set(e.t2d, 0);
  if (mag != 0) niels_load(e, table + TABLE_ENTRY_WORDS * (COMB_HALF * row + mag - 1));
  ge_niels_cneg(e, d < 0);                                      // ed.c:383-389: swap diff/sum, negate prod
}
#else
// device: `table` is the LDS image [COMB_ROWS][COMB_IMG_ENTRIES][COMB_IMG_ENTRY_WORDS]; every lane
// of the wave must be active (the point kernels give idle lanes a real item for that reason)
ED_DEV void comb_select(ge_niels& e, const uint32_t* table, int row, uint32_t digit) {
  const word4* p = reinterpret_cast<const word4*>(
      table + COMB_IMG_ENTRY_WORDS * (COMB_IMG_ENTRIES * row + (int)(threadIdx.x & (COMB_IMG_ENTRIES - 1))));
  uint32_t w[32];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const word4 v = p[q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
  const int d = (int)digit - COMB_HALF, mag = d < 0 ? -d : d;
  const int src = ((mag - 1) & (COMB_IMG_ENTRIES - 1)) << 2;    // byte address of the source lane's slot (lanes 0..31 only)
  const bool none = mag == 0;
#pragma unroll
  for (int j = 0; j < 30; j++) {
    const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)w[j]);
    w[j] = none ? ((j == 0 || j == 10) ? 1u : 0u) : v;          // d = 0: the neutral element (1, 1, 0), ed.c:73 pced_zero
  }
#pragma unroll
  for (int j = 0; j < 10; j++) { e.ymx.v[j] = w[j]; e.ypx.v[j] = w[10 + j]; e.t2d.v[j] = w[20 + j]; }
  ge_niels_cneg(e, d < 0);                                      // the sign, in registers: no branch, no address
}
#endif

// out = x * B for a reduced scalar (x < 2^253) given as eight little-endian words; comb = the LDS
// image on the device, the table in its global layout in the host build (see comb_select).
// PARTS = 4 (small passes, where the 44 additions in a row are the latency of the pass): four lanes - in four
// different WAVES of a block, `part` is uniform over a wave because the lanes of a wave share the row they look up
// (comb_select) - hold the same scalar; lane `part` adds up rows part, part + 4, ... (5 or 6 of the 22) and doubles
// its own odd-digit sum: out = the part's share of x * B (with T), and the caller adds the four shares
// (kernels.hip: point_reduce4).  12 + 1 + 2 additions deep instead of 44 + 1.  The lookups stay what they are: no
// address and no branch depends on a digit; which rows a lane takes depends on its wave's number only.
template <int PARTS = 1>
ED_DEV void scale_base_lane(ge& out, const uint32_t xw[8], const uint32_t* comb, int part = 0) {
  // y = x + sum_j COMB_HALF * 2^(w j): nine words (w = 6: 264 bits)
  uint32_t y[9];
  {
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      uint32_t pat = 0;                          // word k of the offset pattern
#pragma unroll
      for (int j = 0; j < COMB_DIGITS; j++) {
        const int bit = COMB_W * j + COMB_W - 1;
        if ((bit >> 5) == k) pat |= 1u << (bit & 31);
      }
      c += (uint64_t)(k < 8 ? xw[k] : 0u) + pat;
      y[k] = (uint32_t)c;
      c >>= 32;
    }
  }
#pragma unroll
  for (int s = 1; s < PARTS; s++) {              // bring the lane's first row down: `part` shifts by one row
    const bool go = s <= part;
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint32_t v = (y[k] >> (2 * COMB_W)) | (y[k + 1] << (32 - 2 * COMB_W)); y[k] = go ? v : y[k]; }
    y[8] = go ? y[8] >> (2 * COMB_W) : y[8];
  }
  ge r0, r1;
#pragma unroll 1
  for (int i = 0; i < (COMB_ROWS + PARTS - 1) / PARTS; i++) {
    const int row = PARTS * i + part;
    const bool valid = PARTS == 1 || row < COMB_ROWS;          // (a lane past the last row adds neutral elements)
    const uint32_t two = y[0] & ((1u << (2 * COMB_W)) - 1u);   // digits 2 row and 2 row + 1
#pragma unroll
    for (int s = 0; s < PARTS; s++) {
#pragma unroll
      for (int k = 0; k < 8; k++) y[k] = (y[k] >> (2 * COMB_W)) | (y[k + 1] << (32 - 2 * COMB_W));
      y[8] >>= 2 * COMB_W;
    }
    // both lookups of the row are issued ahead of its two additions: the second one's 30 shuffles and their latency run on
    // the LDS pipe beside the first addition's multiplications instead of between the two additions (fe_mul ends in a
    // scheduling barrier, so the compiler does not move them there by itself).  Measured, profiles/r04_sign_ab.txt:
    // sign 434 -> 444 M/s; thirty more registers (253 of 256: the secret scalars leave for the workspace before the comb).
    ge_niels e, e2;
    comb_select(e, comb, valid ? row : 0, valid ? two & ((1u << COMB_W) - 1u) : (uint32_t)COMB_HALF);
    comb_select(e2, comb, valid ? row : 0, valid ? two >> COMB_W : (uint32_t)COMB_HALF);
    if (i == 0) ge_from_niels(r0, e); else ge_add_niels(r0, r0, e, true);      // (the first entry IS the sum so far: one multiplication instead of seven)
    if (i == 0) ge_from_niels(r1, e2); else ge_add_niels(r1, r1, e2, true);
  }
#pragma unroll 1
  for (int k = 0; k < COMB_W; k++) ge_dbl(r1, r1, k == COMB_W - 1);
  ge_cached c;
  ge_to_cached(c, r1);
  ge_add_cached(out, r0, c, PARTS > 1);
}

// ed25519-sha512.c:31-47 ed25519_key_setup: h = SHA-512(sk), clamped
ED_DEV void key_setup(uint32_t h[16], const uint32_t sk[8]) {
  sha512_prefix_msg<8>(h, sk, nullptr, 0);
  h[0] &= 0xfffffff8u;
  h[7] = (h[7] & 0x7fffffffu) | 0x40000000u;
}

// The fixed-base operations are split in two, like verify: a "point" step that ends with the
// projective result, and a "finish" step that needs 1/Z (or 1/(Z-Y)); the kernels run the finish
// step for eight items per lane on ONE shared inversion (Montgomery's trick, kernels.hip).

// ed_export given zinv = 1/Z (ed.c:155-169)
ED_DEV void encode_lane(uint32_t out[8], const fe& X, const fe& Y, const fe& zinv) {
  fe x, y;
  fe_mul(x, X, zinv);
  fe_mul(y, Y, zinv);
  fe_tobytes(out, y);
  out[7] |= fe_parity(x) << 31;
}

// ed25519-sha512.c:53-67 genpub, up to the point A = a*B
template <int PARTS = 1>
ED_DEV void genpub_point_lane(ge& A, const uint32_t sk[8], const uint32_t* comb, int part = 0) {
  uint32_t h[16], aw[8];
  key_setup(h, sk);
  sc a;
  sc_from_words<8>(a, h);
  sc_to_words(aw, a);
  scale_base_lane<PARTS>(A, aw, comb, part);
}

// ed25519-sha512.c:84-105 sign, the secret scalars: aw = a (from the key hash) and rw = r = H(h[32..64) || M) mod l, as words
ED_DEV void sign_scalars_lane(uint32_t aw[8], uint32_t rw[8], const uint32_t sk[8], const uint8_t* m, size_t mlen) {
  uint32_t h[16], dig[16];
  key_setup(h, sk);
  sc a, r;
  sc_from_words<8>(a, h);
  sc_to_words(aw, a);
  sha512_prefix_msg<8>(dig, h + 8, m, mlen);     // r = H(h[32..64) || M)
  sc_from_words<16>(r, dig);
  sc_to_words(rw, r);
}

// ed25519-sha512.c:84-110 sign, up to R = r*B; aw, rw = the reduced scalars a and r as words
template <int PARTS = 1>
ED_DEV void sign_point_lane(ge& R, uint32_t aw[8], uint32_t rw[8], const uint32_t sk[8],
                            const uint8_t* m, size_t mlen, const uint32_t* comb, int part = 0) {
  sign_scalars_lane(aw, rw, sk, m, mlen);
  scale_base_lane<PARTS>(R, rw, comb, part);
}

// ed25519-sha512.c:112-122 sign, from the encoded R on: S = r + H(R || A || M) * a, in two steps so that
// the finish kernel holds the secret scalars only after the hash (fewer live registers: no scratch)
ED_DEV void sign_challenge_lane(sc& t, const uint32_t Rw[8], const uint32_t pub[8], const uint8_t* m, size_t mlen) {
  uint32_t pre[16], dig[16];
#pragma unroll
  for (int k = 0; k < 8; k++) { pre[k] = Rw[k]; pre[8 + k] = pub[k]; }
  sha512_prefix_msg<16>(dig, pre, m, mlen);      // t = H(R || A || M)
  sc_from_words<16>(t, dig);
}
ED_DEV void sign_response_lane(uint32_t Sw[8], const sc& t, const uint32_t aw[8], const uint32_t rw[8]) {
  sc a, r, S;
  sc_from_words<8>(a, aw);
  sc_from_words<8>(r, rw);
  sc_mul(S, t, a);
  sc_add(S, r, S);
  sc_to_words(Sw, S);
}
ED_DEV void sign_finish_lane(uint32_t Sw[8], const uint32_t Rw[8], const uint32_t aw[8],
                             const uint32_t rw[8], const uint32_t pub[8], const uint8_t* m, size_t mlen) {
  sc t;
  sign_challenge_lane(t, Rw, pub, m, mlen);
  sign_response_lane(Sw, t, aw, rw);
}

// x25519.c:158-190 do_x25519_base, up to R = x*B
template <int PARTS = 1>
ED_DEV void x25519_base_point_lane(ge& R, uint32_t s[8], const uint32_t* comb, int part = 0) {
  uint32_t xw[8];
  clamp(s);
  sc x;
  sc_from_words<8>(x, s);
  sc_to_words(xw, x);
  scale_base_lane<PARTS>(R, xw, comb, part);
}

// x25519.c:191-196: u = (z + y) / (z - y), given dinv = 1 / (z - y)  (0 when z = y, as fld_inv)
ED_DEV void x25519_base_finish_lane(uint32_t out[8], const fe& Y, const fe& Z, const fe& dinv) {
  fe u;
  fe_add(u, Z, Y);                               // 2u
  fe_mul(u, u, dinv);
  fe_tobytes(out, u);
}

// ed25519-sha512.c:187-232 pk_ed25519_to_x25519: u = (z + y) / (z - y) of the imported point.
// ed_import always returns z = 1 and y = the 255 low bits of the input taken mod p; x (and the
// square root that produces it) never reaches the output, so it is not computed.
ED_DEV void pk_to_x_lane(uint32_t out[8], uint32_t w[8]) {
  w[7] &= 0x7fffffffu;
  fe y, one, u, t;
  fe_frombytes(y, w);
  fe_set(one, 1);
  fe_sub(t, one, y);                             // 3u
  fe_inv(t, t);                                  // 1 - y = 0 -> 0, as fld_inv
  fe_add(u, one, y);
  fe_mul(u, u, t);
  fe_tobytes(out, u);
}

// ed25519-sha512.c:239-256 sk_ed25519_to_x25519: the first 32 bytes of the clamped key hash
ED_DEV void sk_to_x_lane(uint32_t out[8], const uint32_t sk[8]) {
  uint32_t h[16];
  key_setup(h, sk);
#pragma unroll
  for (int k = 0; k < 8; k++) out[k] = h[k];
}

}  // namespace ed
