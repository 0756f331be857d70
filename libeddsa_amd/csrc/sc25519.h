// sc25519.h - arithmetic modulo l = 2^252 + 27742317777372353535851937790883648493 on the device.
//
// Replaces the reference's lib/sc.c + lib/sc.h.  Like the reference's 32-bit-limb build
// (sc.h:31-35) a scalar is ten 26-bit limbs and reduction is Barrett (HAC 14.42) with b = 2^26,
// k = 10, the constants of sc.c:43-53.  All loops are fully unrolled so every array lives in
// registers; per verify this layer is < 0.3 % of the work.
//
// Semantics that matter for parity (sc.c:191-214 sc_import): every 32- or 64-byte string is
// accepted and reduced mod l -- S is NOT range-checked in verify.
#pragma once
#include "fe25519.h"

namespace ed {

struct sc { uint32_t v[10]; };      // radix 2^26, value < l

// sc.c:43-45 con_m (l) and sc.c:47-49 con_mu = floor(2^520 / l), radix 2^26
ED_DEV constexpr uint32_t sc_L(int i) {
  constexpr uint32_t L[11] = {16110573, 10012311, 30238081, 58362846, 1367801, 0, 0, 0, 0, 262144, 0};
  return L[i];
}
ED_DEV constexpr uint32_t sc_MU(int i) {
  constexpr uint32_t MU[11] = {1252153, 23642763, 41867726, 2198694, 17178973, 67107528,
                               67108863, 67108863, 67108863, 67108863, 255};
  return MU[i];
}

// r (11 limbs, < 2^286) -> r - l if r >= l   (sc.c:143-151, the masked final subtraction)
ED_DEV void sc_cond_sub_l(uint32_t r[11]) {
  uint32_t d[11];
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 11; i++) {
    int32_t t = (int32_t)r[i] - (int32_t)sc_L(i) + borrow;
    d[i] = (uint32_t)t & M26;
    borrow = t >> 26;
  }
  const bool ge = (borrow == 0);
#pragma unroll
  for (int i = 0; i < 11; i++) r[i] = ge ? d[i] : r[i];
}

// sc.c:79-158 sc_barrett: x (20 carried limbs, x < 2^520) -> x mod l
ED_DEV void sc_barrett(sc& out, const uint32_t x[20]) {
  uint32_t q3[11], r[11];
  uint64_t acc = 0;
  // q3 = floor(floor(x / b^9) * mu / b^11): all 21 columns, so the floor is exact
#pragma unroll
  for (int c = 0; c <= 20; c++) {
#pragma unroll
    for (int i = 0; i <= 10; i++) {
      const int j = c - i;
      if (j < 0 || j > 10) continue;
      acc = mad(x[9 + i], sc_MU(j), acc);
    }
    if (c >= 11) q3[c - 11] = (uint32_t)acc & M26;
    acc >>= 26;
  }
  q3[10] = (uint32_t)acc;
  // r = (x - q3 * l) mod b^11
  acc = 0;
  int32_t borrow = 0;
#pragma unroll
  for (int c = 0; c <= 10; c++) {
#pragma unroll
    for (int i = 0; i <= c; i++) {
      if (sc_L(c - i) == 0) continue;
      acc = mad(q3[i], sc_L(c - i), acc);
    }
    int32_t t = (int32_t)x[c] - (int32_t)((uint32_t)acc & M26) + borrow;
    r[c] = (uint32_t)t & M26;
    borrow = t >> 26;
    acc >>= 26;
  }
  // 0 <= r < 3 l  (HAC 14.42): at most two subtractions
  sc_cond_sub_l(r);
  sc_cond_sub_l(r);
#pragma unroll
  for (int i = 0; i < 10; i++) out.v[i] = r[i];
}

// sc.c:191-214 sc_import for len = 4*NW bytes (NW = 8 or 16 little-endian words)
template <int NW>
ED_DEV void sc_from_words(sc& out, const uint32_t w[NW]) {
  uint32_t x[20];
#pragma unroll
  for (int i = 0; i < 20; i++) {
    const int bit = 26 * i, wi = bit >> 5, sh = bit & 31;
    if (wi >= NW) { x[i] = 0; continue; }
    uint32_t v = w[wi] >> sh;
    if (sh > 6 && wi + 1 < NW) v |= w[wi + 1] << (32 - sh);
    x[i] = v & M26;
  }
  sc_barrett(out, x);
}

// sc.c:221-236 sc_export (the value is already fully reduced)
ED_DEV void sc_to_words(uint32_t w[8], const sc& a) {
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const int bit = 26 * i, wi = bit >> 5, sh = bit & 31;
    w[wi] |= a.v[i] << sh;
    if (sh > 6 && wi + 1 < 8) w[wi + 1] |= a.v[i] >> (32 - sh);
  }
}

// sc.c:241-266 sc_mul
ED_DEV void sc_mul(sc& out, const sc& a, const sc& b) {
  uint32_t x[20];
  uint64_t acc = 0;
#pragma unroll
  for (int c = 0; c <= 18; c++) {
#pragma unroll
    for (int i = 0; i <= 9; i++) {
      const int j = c - i;
      if (j < 0 || j > 9) continue;
      acc = mad(a.v[i], b.v[j], acc);
    }
    x[c] = (uint32_t)acc & M26;
    acc >>= 26;
  }
  x[19] = (uint32_t)acc;
  sc_barrett(out, x);
}

// sc.h:53-59 sc_add + the sc_reduce its consumers apply; a, b < l
ED_DEV void sc_add(sc& out, const sc& a, const sc& b) {
  uint32_t r[11];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    uint32_t t = a.v[i] + b.v[i] + c;
    r[i] = t & M26;
    c = t >> 26;
  }
  r[10] = c;
  sc_cond_sub_l(r);
#pragma unroll
  for (int i = 0; i < 10; i++) out.v[i] = r[i];
}

// 256-bit little-endian add of the same pattern word to each of the eight words (0x88888888 for
// 4-bit signed digits, 0x80008000 for 16-bit signed digits): the recoding of ed.c:407-409
// (x + con_off, then nibble - 8), done on the exported words.  Needs w < 2^253, so no carry
// leaves bit 255.
ED_DEV void words_add_pattern(uint32_t w[8], uint32_t pat32) {
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    c += (uint64_t)w[k] + pat32;
    w[k] = (uint32_t)c;
    c >>= 32;
  }
}

// the inverse of words_add_pattern: w - pat * (1, 1, ..., 1), w >= the pattern value
ED_DEV void words_sub_pattern(uint32_t w[8], uint32_t pat32) {
  int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    c += (int64_t)w[k] - (int64_t)pat32;
    w[k] = (uint32_t)c;
    c >>= 32;                                    // arithmetic: 0 or -1
  }
}

}  // namespace ed
