// rlc_lanes.h - what ONE lane computes in the kernels of rlc.hip (batch verification by random linear
// combination), as plain functions over pointers like lanes.h, so that the same source compiles for the
// host under -DED_HOST_CHECK (tests/host_check/, every limb bound asserted) and is checked there:
// decoding and routing flags against big-integer arithmetic, the digit recoding against the scalars it
// encodes, and the whole combination of a small batch -- evaluated by plain double-and-add instead of the
// workgroup's buckets -- against the neutral element.  The workgroup-cooperative parts (LDS counting sort,
// bucket sums, scans) exist only in rlc.hip and are covered by the GPU tests.
#pragma once
#include "lanes.h"

namespace ed {

enum : uint8_t { RLC_R_VALID = 1, RLC_PER_ITEM = 2 };
constexpr int RLC_WINDOWS_A = 32, RLC_WINDOWS_R = 16, RLC_WINDOWS = RLC_WINDOWS_A + RLC_WINDOWS_R;
constexpr int RLC_TREE_FAN = 16;                // children per node of the batch hash tree (rlc.hip: k_rlc_tree)

ED_DEV bool ge_is_neutral(const ge& p) {
  fe d;
  fe_sub(d, p.Y, p.Z);
  return fe_iszero(p.X) && fe_iszero(d);
}
// Is the curve point p (Z = 1) one of the eight points of order dividing 8?  Those are exactly the points
// with y in {0, 1, -1, y8, -y8} (y8 = the y of a point of order 8), so five comparisons of the canonical y
// replace three doublings and a neutrality test.
ED_DEV bool ge_small_order(const ge& p) {
  constexpr uint32_t Y8[10] = {60155942, 32288931, 6862340, 26496934, 63071167, 28106709, 31680898, 18229030, 47743011, 1569101};
  constexpr uint32_t Y8N[10] = {6952903, 1265500, 60246523, 7057497, 4037696, 5447722, 35427965, 15325401, 19365852, 31985330};
  fe t;
  fe_canon(t, p.Y);
  uint32_t rest = 0, m1 = 0, d8 = 0, d8n = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    if (i) rest |= t.v[i];
    m1 |= t.v[i] ^ (i == 0 ? M26 - 19 : limb_mask(i));      // p - 1
    d8 |= t.v[i] ^ Y8[i];
    d8n |= t.v[i] ^ Y8N[i];
  }
  return (rest == 0 && t.v[0] <= 1u) || m1 == 0 || d8 == 0 || d8n == 0;
}
ED_DEV void niels_of_affine(ge_niels& n, const ge& p) {   // p.Z = 1: ed.c:436-442 ed_precompute
  fe_sub(n.ymx, p.Y, p.X); fe_carry(n.ymx);
  fe_add(n.ypx, p.Y, p.X); fe_carry(n.ypx);
  fe_mul(n.t2d, p.T, fe_const_2d());
}

// -A as an affine niels point: permissive decoding (ed.c:100-149), as the per-item path.  Returns the
// routing flag: a key that is no curve point or has small order sends its group to the per-item kernels.
ED_DEV uint8_t rlc_decode_key_lane(ge_niels& nl, const uint32_t aw[8]) {
  ge p;
  bool oncurve;
  ge_frombytes(p, oncurve, aw, true);
  const uint8_t fl = (!oncurve || ge_small_order(p)) ? RLC_PER_ITEM : 0;
  niels_of_affine(nl, p);
  return fl;
}

// -R as an affine niels point.  Only the canonical encoding of a curve point can equal what ed_export
// writes (ed.c:155-169: y < p, sign bit = parity of x), so anything else is rejected outright (flag
// RLC_R_VALID clear); a valid R of small order sends its group to the per-item kernels.
ED_DEV uint8_t rlc_decode_r_lane(ge_niels& nl, const uint32_t rw[8]) {
  const uint32_t sign = rw[7] >> 31, top = rw[7] & 0x7fffffffu;
  const bool y_ge_p = top == 0x7fffffffu && (rw[1] & rw[2] & rw[3] & rw[4] & rw[5] & rw[6]) == 0xffffffffu &&
                      rw[0] >= 0xffffffedu;
  ge p;
  bool oncurve;
  ge_frombytes(p, oncurve, rw, true);
  const bool valid = oncurve && !y_ge_p && !(sign != 0 && fe_iszero(p.X));
  uint8_t fl = 0;
  if (valid) {
    fl |= RLC_R_VALID;
    if (ge_small_order(p)) fl |= RLC_PER_ITEM;
  }
  niels_of_affine(nl, p);
  return fl;
}

// t = SHA-512(R || A || M) mod l and S mod l (ed25519-sha512.c:162-172) as words, and the item's leaf
// SHA-512(SHA-512(R || A || M) || S)[0..32) of the batch hash tree; s_bytes = the 32 bytes of S
ED_DEV void rlc_hash_lane(uint32_t tw[8], uint32_t sw[8], uint32_t leaf[8], const uint32_t rw[8], const uint32_t aw[8],
                          const uint8_t* s_bytes, const uint8_t* m, size_t mlen) {
  uint32_t pre[16], dig[16], lf[16];
#pragma unroll
  for (int k = 0; k < 8; k++) { pre[k] = rw[k]; pre[8 + k] = aw[k]; }
  sha512_prefix_msg<16>(dig, pre, m, mlen);
  sha512_prefix_msg<16>(lf, dig, s_bytes, 32);
  sc t, s;
  sc_from_words<16>(t, dig);
  sc_to_words(tw, t);
  sc_from_words<8>(s, sw);                       // not range-checked: sc.c:191-214
  sc_to_words(sw, s);
#pragma unroll
  for (int k = 0; k < 8; k++) leaf[k] = lf[k];
}

// l as eight little-endian words
ED_DEV constexpr uint32_t rlc_L(int i) {
  constexpr uint32_t Lw[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0x00000000u, 0x00000000u, 0x00000000u, 0x10000000u};
  return Lw[i];
}

// The key's scalar as an integer a' with |a'| <= 4 l and a' = z t (mod 8 l), from a0 = z t mod l: A may carry a
// component of order dividing 8 (the reference checks neither subgroup membership nor small order), on
// which z t and z t mod l act differently, so the class mod l alone does not determine a'*A.  z t = a0 + k l with
// k = (z t - a0) / l, and l = 5 (mod 8) is its own inverse there: k mod 8 = 5 (z t - a0) mod 8, which needs the
// low words only.  Centred, k' in [-4, 3]: a' = a0 + k' l is negative exactly when k' is (a0 < l).  mag = |a'|
// (below 2^255: no carry out of the recoding), returns a' < 0.
ED_DEV bool rlc_key_scalar_mod_8l(uint32_t mag[8], const uint32_t a0[8], uint32_t z_lo, uint32_t t_lo) {
  const uint32_t k = (((z_lo * t_lo - a0[0]) & 7u) * 5u) & 7u;
  const bool neg = k >= 4u;
  const uint32_t kk = neg ? 8u - k : k;          // |k'| <= 4
  uint64_t p = 0;
  int64_t c = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    p += (uint64_t)kk * rlc_L(q);
    c += (int64_t)(uint32_t)p + (neg ? -(int64_t)a0[q] : (int64_t)a0[q]);
    mag[q] = (uint32_t)c;
    c >>= 32;
    p >>= 32;
  }
  ED_CHECK(c + (int64_t)p == 0);                 // 0 <= |a'| <= 4 l < 2^255
  return neg;
}

// The coefficient z_i = the low 126 bits of SHA-512(seed || i || "rlc"), made odd; a' = z t mod 8 l, centred
// (rlc_key_scalar_mod_8l), and zs = z S mod l (9 words, the top one 0; B has order l); the signed byte digits of
// a' (32) and z (16): for a value x >= 0 digit j = byte j of (x + 0x80...80) - 128, the recoding of ed.c:407-409
// with 8-bit windows; a negative a' = -m takes the negated digits of m under the mirrored recoding (byte j of
// (m + 0x7f...7f) - 127, in [-127, 128]), so that every digit stays in [-128, 127].
ED_DEV void rlc_scalars_lane(int8_t dig_a[RLC_WINDOWS_A], int8_t dig_r[RLC_WINDOWS_R], uint32_t zs[9],
                             const uint32_t seed[8], uint64_t i, const uint32_t tw[8], const uint32_t sw[8]) {
  uint32_t pre[16], h[16], zw[8], aw[8];
#pragma unroll
  for (int q = 0; q < 8; q++) pre[q] = seed[q];
  pre[8] = (uint32_t)i; pre[9] = (uint32_t)(i >> 32); pre[10] = 0x00636c72u;
#pragma unroll
  for (int q = 11; q < 16; q++) pre[q] = 0;
  sha512_prefix_msg<16>(h, pre, nullptr, 0);
  zw[0] = h[0] | 1u; zw[1] = h[1]; zw[2] = h[2]; zw[3] = h[3] & 0x3fffffffu;
  zw[4] = zw[5] = zw[6] = zw[7] = 0;
  sc z, t, s, a;
  sc_from_words<8>(z, zw);
  sc_from_words<8>(t, tw);
  sc_from_words<8>(s, sw);
  sc_mul(a, z, t);
  sc_to_words(aw, a);
  sc_mul(s, z, s);
  sc_to_words(zs, s);
  zs[8] = 0;
  uint32_t tr[8];
  sc_to_words(tr, t);                            // t mod l as the per-item check takes it (sc.c:191-214); tw already is
  uint32_t m[8];
  const bool a_neg = rlc_key_scalar_mod_8l(m, aw, zw[0], tr[0]);
#pragma unroll
  for (int q = 0; q < 8; q++) aw[q] = m[q];
  words_add_pattern(aw, a_neg ? 0x7f7f7f7fu : 0x80808080u);   // |a'| < 2^255: no carry out of bit 255
  {
    uint64_t c = 0;                              // z < 2^126: no carry out of bit 127
#pragma unroll
    for (int q = 0; q < 4; q++) { c += (uint64_t)zw[q] + 0x80808080u; zw[q] = (uint32_t)c; c >>= 32; }
  }
#pragma unroll
  for (int wd = 0; wd < RLC_WINDOWS_A; wd++) {
    const int byte = (int)((aw[wd >> 2] >> (8 * (wd & 3))) & 0xffu);
    dig_a[wd] = (int8_t)(a_neg ? 127 - byte : byte - 128);
  }
#pragma unroll
  for (int wd = 0; wd < RLC_WINDOWS_R; wd++) dig_r[wd] = (int8_t)((int)((zw[wd >> 2] >> (8 * (wd & 3))) & 0xffu) - 128);
}

// the digits of a group's base-point scalar: sum (an integer of up to 16 words) mod l, recoded as above
ED_DEV void rlc_group_scalar_lane(int8_t dig[32], const uint32_t sum[16]) {
  sc s;
  uint32_t sw[8];
  sc_from_words<16>(s, sum);
  sc_to_words(sw, s);
  words_add_pattern(sw, 0x80808080u);
#pragma unroll
  for (int wd = 0; wd < 32; wd++) dig[wd] = (int8_t)((int)((sw[wd >> 2] >> (8 * (wd & 3))) & 0xffu) - 128);
}

}  // namespace ed
