// fe25519.h - GF(2^255-19) for gfx950, one field element per lane.
//
// Replaces the reference's lib/fld.c + lib/fld.h on the device.  The limb formulation follows
// the measured gfx950 issue rates (profiles/r01_valu_rates.txt): v_mad_u64_u32 issues at the
// same 4 cycles per wave64 as every other VOP3 instruction, so a field element is ten unsigned
// 32-bit limbs in radix 2^25.5 (26,25,26,25,... bits; limb i sits at bit ceil(25.5 i)) and a
// product column is a chain of v_mad_u64_u32 into one 64-bit accumulator whose carry-in is the
// chain's initial addend (the carry add is free).
//
// Bounds ("u" = one nominal limb: 2^26 for even limbs, 2^25 for odd limbs):
//   tight    : what fe_mul / fe_sq / fe_carry / fe_mul121665 return: every limb < 1u, except
//              limb 1 which may reach 2^25 + 2^19.
//   fe_mul(f,g), fe_sq(f): f limbs < 8u; g (and fe_sq's f) limbs < 3.36u (19*g_even, 38*g_odd
//              must fit 32 bits); every 64-bit column then stays below 2^64.
//   fe_sub(a,b) = a + 2p - b needs b < 2u - 38 (so b tight, or tight + tight is NOT allowed);
//              fe_sub4(a,b) = a + 4p - b takes b up to 4u - 76.
// Every call site states its bounds, and they are CHECKED: with -DED_HOST_CHECK this same source
// compiles for the host CPU (tests/host_check/, a test binary, never part of the product) with an
// assertion at every precondition and on every 64-bit column sum; tests/test_device_source_on_host.py
// drives it with random and extreme inputs and compares with the oracle.
#pragma once
#include <stdint.h>

#ifdef ED_HOST_CHECK
// test build of the device source for the host CPU
#include <stddef.h>
namespace ed { void bound_violation(const char* file, int line, const char* what); }
#define ED_DEV inline
#define ED_SCHED_FENCE() ((void)0)
#define ED_CHECK(cond) do { if (!(cond)) ::ed::bound_violation(__FILE__, __LINE__, #cond); } while (0)
#define ED_CONSTANT_MEM static const
#define ED_ASSUME(cond) ((void)0)
namespace ed { static const uint64_t ed_opaque_zero = 0; }
#else
#include <hip/hip_runtime.h>
#define ED_DEV __device__ __forceinline__
// Each fe_mul / fe_sq ends with a scheduling fence: without it hipcc interleaves independent
// multiplications for ILP, which pushes the big kernels past 256 VGPRs into scratch spills;
// two to four waves per SIMD already hide the dependent-issue latency (profiles/r01_fe_rates.txt).
#define ED_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define ED_CHECK(cond) ((void)0)
#define ED_CONSTANT_MEM __device__ __constant__
#define ED_ASSUME(cond) __builtin_assume(cond)
#endif

namespace ed {

#ifndef ED_HOST_CHECK
// Always 0 (nothing ever stores to it), but externally visible, so the compiler cannot fold it.
// mad() below states its no-wrap fact through it: a plain `r >= c` is PROVABLE from the known
// bits of masked limbs, InstCombine then deletes the assumption and the protection is gone.
inline __device__ uint64_t ed_opaque_zero;
#endif

struct fe { uint32_t v[10]; };

constexpr uint32_t M26 = (1u << 26) - 1;
constexpr uint32_t M25 = (1u << 25) - 1;

ED_DEV constexpr int limb_bits(int i) { return (i & 1) ? 25 : 26; }
ED_DEV constexpr uint32_t limb_mask(int i) { return (i & 1) ? M25 : M26; }

// One multiply-accumulate of a product column: v_mad_u64_u32.  The assumption (true: the limb
// bounds keep every column below 2^64, so the sum never wraps; ed_opaque_zero is 0) gives each
// partial sum a second use in the IR, which stops LLVM's reassociation from moving the carry-in to the END of the
// chain -- there it costs a v_mul plus a 64-bit add per column instead of being the free initial
// addend.  It emits no code and, unlike an asm barrier, does not pin the instruction order.
ED_DEV uint64_t mad(uint32_t a, uint32_t b, uint64_t c) {
  const uint64_t r = (uint64_t)a * b + c;
  ED_CHECK(r >= c);                              // the column sum did not wrap
  ED_ASSUME((r ^ ed_opaque_zero) >= (c ^ ed_opaque_zero));
  return r;
}

ED_DEV void fe_set(fe& h, uint32_t x) {
  h.v[0] = x;
#pragma unroll
  for (int i = 1; i < 10; i++) h.v[i] = 0;
}

// fld.h:84 fld_add: limb-wise, no carry.  bounds add.
ED_DEV void fe_add(fe& h, const fe& f, const fe& g) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    ED_CHECK(f.v[i] <= 0xffffffffu - g.v[i]);
    h.v[i] = f.v[i] + g.v[i];
  }
}

// fld.h:91 fld_sub, with a 2p bias instead of signed limbs.  needs g < 2u-38; result < f + 2u.
ED_DEV void fe_sub(fe& h, const fe& f, const fe& g) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const uint32_t bias = (i == 0) ? 2 * (M26 - 18) : 2 * limb_mask(i);
    ED_CHECK(g.v[i] <= bias && f.v[i] <= 0xffffffffu - bias);
    h.v[i] = f.v[i] + bias - g.v[i];
  }
}

// same with a 4p bias: g < 4u-76; result < f + 4u.
ED_DEV void fe_sub4(fe& h, const fe& f, const fe& g) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const uint32_t bias = (i == 0) ? 4 * (M26 - 18) : 4 * limb_mask(i);
    ED_CHECK(g.v[i] <= bias && f.v[i] <= 0xffffffffu - bias);
    h.v[i] = f.v[i] + bias - g.v[i];
  }
}

// fld.h:136 fld_neg: 2p - f, f < 2u-38.
ED_DEV void fe_neg(fe& h, const fe& f) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const uint32_t bias = (i == 0) ? 2 * (M26 - 18) : 2 * limb_mask(i);
    ED_CHECK(f.v[i] <= bias);
    h.v[i] = bias - f.v[i];
  }
}

// one 32-bit carry sweep: any limbs < 2^32/19-ish in, tight out.
ED_DEV void fe_carry(fe& h) {
  uint32_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c = h.v[i] >> limb_bits(i);
    h.v[i] &= limb_mask(i);
    ED_CHECK(h.v[i + 1] <= 0xffffffffu - c);
    h.v[i + 1] += c;
  }
  c = h.v[9] >> 25;
  h.v[9] &= M25;
  h.v[0] += 19 * c;             // c < 2^7, 19c < 2^12
  c = h.v[0] >> 26;
  h.v[0] &= M26;
  h.v[1] += c;
}

// shared tail of fe_mul / fe_sq: r[] are the masked columns, top (< 2^39) the carry out of limb 9;
// fold 19*top into limb 0 and carry once more into limb 1.  h may alias the inputs of the caller.
ED_DEV void fe_fold_top(fe& h, const uint32_t r[10], uint64_t top) {
  ED_CHECK((top >> 32) < 128);
  uint64_t t = mad((uint32_t)top, 19u, (uint64_t)r[0]) + ((uint64_t)(19u * (uint32_t)(top >> 32)) << 32);
  h.v[0] = (uint32_t)t & M26;
  h.v[1] = r[1] + (uint32_t)(t >> 26);
#pragma unroll
  for (int i = 2; i < 10; i++) h.v[i] = r[i];
  ED_SCHED_FENCE();
}

// 255-bit packing of a tight element for tables in memory (lanes.h: cached entries): the integer sum f_i 2^off(i) as
// eight little-endian words (no reduction: any representative below 2^256 will do), and back.  Unpacked limbs are
// exact except the last, which keeps whatever lies above bit 230 (< 2^26).
ED_DEV constexpr int limb_offset(int i) { return 26 * ((i + 1) / 2) + 25 * (i / 2); }   // 0 26 51 77 102 128 153 179 204 230
ED_DEV void fe_pack(uint32_t w[8], const fe& f) {
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i < 10; i++)
      if ((limb_offset(i) >> 5) == k) acc += (uint64_t)f.v[i] << (limb_offset(i) & 31);
    w[k] = (uint32_t)acc;
    acc >>= 32;
  }
  ED_CHECK(acc == 0);                            // tight limbs: the sum stays below 2^256
}
ED_DEV void fe_unpack(fe& f, const uint32_t w[8]) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const int k = limb_offset(i) >> 5, sh = limb_offset(i) & 31;
    uint32_t v = w[k] >> sh;
    if (sh + limb_bits(i) > 32 && k < 7) v |= w[k + 1] << (32 - sh);
    f.v[i] = i < 9 ? (v & limb_mask(i)) : v;
  }
}

// 19 g, for a second operand that many multiplications share (the ladder's x1): computed once
struct fe19 { uint32_t v[10]; };
ED_DEV void fe_premul19(fe19& g19, const fe& g) {
  g19.v[0] = 0;
#pragma unroll
  for (int j = 1; j < 10; j++) { ED_CHECK(g.v[j] <= 0xffffffffu / 19u); g19.v[j] = 19u * g.v[j]; }
}

// fld.c:209-244 (fld.c:447-497 in the 32-bit build) fld_mul with 19 g supplied.  f < 8u, g < 3.36u, h tight.
ED_DEV void fe_mul_pre(fe& h, const fe& f, const fe& g, const fe19& g19) {
  uint32_t f2[10], r[10];
#pragma unroll
  for (int i = 1; i < 10; i += 2) { ED_CHECK(f.v[i] <= 0x7fffffffu); f2[i] = 2u * f.v[i]; }
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      const bool wrap = i > k;
      const bool odd2 = (i & 1) && (j & 1);
      acc = mad(odd2 ? f2[i] : f.v[i], wrap ? g19.v[j] : g.v[j], acc);
    }
    r[k] = (uint32_t)acc & limb_mask(k);
    acc >>= limb_bits(k);
  }
  fe_fold_top(h, r, acc);
}

// fld.c:209-244 (fld.c:447-497 in the 32-bit build) fld_mul.  f < 8u, g < 3.36u, h tight.
ED_DEV void fe_mul(fe& h, const fe& f, const fe& g) {
  fe19 g19;
  fe_premul19(g19, g);
  fe_mul_pre(h, f, g, g19);
}

// fld.c:249-280 (fld.c:502-531) fld_sq.  f < 3.36u, h tight.
// Thirteen premultiplied operands serve the 55 products: 2 f_i for i < 8, 19 f_j for the even and
// 38 f_j for the odd j >= 5.  A wrapped pair (i, j), i < j, i + j >= 10, carries the factor 2 * 19, times 2
// again when both are odd: 2 f_i * 38 f_j (both odd), f_i * 38 f_j (i even, j odd: the 2 sits in the 38),
// 2 f_i * 19 f_j (j even).
ED_DEV void fe_sq(fe& h, const fe& f) {
  uint32_t f2[10], fw[10], r[10];
#pragma unroll
  for (int i = 0; i < 8; i++) { ED_CHECK(f.v[i] <= 0x7fffffffu); f2[i] = 2u * f.v[i]; }
#pragma unroll
  for (int j = 5; j < 10; j++) {
    ED_CHECK(f.v[j] <= 0xffffffffu / ((j & 1) ? 38u : 19u));
    fw[j] = ((j & 1) ? 38u : 19u) * f.v[j];
  }
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      if (i > j) continue;                       // each unordered pair once
      const bool wrap = i + j >= 10;             // i + j = k + 10
      const bool odd2 = (i & 1) && (j & 1);
      uint32_t a, b;
      if (wrap) {
        b = fw[j];
        a = (i == j || (!(i & 1) && (j & 1))) ? f.v[i] : f2[i];   // the diagonal, and even i with odd j: 38 f_j already holds the 2
      } else {
        a = (i != j) ? f2[i] : f.v[i];
        b = odd2 ? f2[j] : f.v[j];
      }
      acc = mad(a, b, acc);
    }
    r[k] = (uint32_t)acc & limb_mask(k);
    acc >>= limb_bits(k);
  }
  fe_fold_top(h, r, acc);
}

ED_DEV void fe_sqn(fe& h, const fe& f, int n) {
  fe_sq(h, f);
  for (int i = 1; i < n; i++) fe_sq(h, h);
}

// fld.c:183-204 fld_scale with s = 121665 (its only caller, x25519.c:78).  f < 2^32, h tight.
ED_DEV void fe_mul121665(fe& h, const fe& f) {
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    acc = mad(f.v[k], 121665u, acc);     // reads f.v[k] before h.v[k] is written: alias-safe
    h.v[k] = (uint32_t)acc & limb_mask(k);
    acc >>= limb_bits(k);
  }
  uint32_t t = h.v[0] + 19u * (uint32_t)acc;     // acc < 2^24
  h.v[0] = t & M26;
  h.v[1] += t >> 26;
}

// fld.c:136-156 fld_import: 256-bit little-endian words; bit 255 is folded in as +19 (NOT masked:
// this is what the reference does and what its x25519 table pins).
ED_DEV void fe_frombytes(fe& h, const uint32_t w[8]) {
  h.v[0] = w[0] & M26;
  h.v[1] = ((w[0] >> 26) | (w[1] << 6)) & M25;
  h.v[2] = ((w[1] >> 19) | (w[2] << 13)) & M26;
  h.v[3] = ((w[2] >> 13) | (w[3] << 19)) & M25;
  h.v[4] = (w[3] >> 6);
  h.v[5] = w[4] & M25;
  h.v[6] = ((w[4] >> 25) | (w[5] << 7)) & M26;
  h.v[7] = ((w[5] >> 19) | (w[6] << 13)) & M25;
  h.v[8] = ((w[6] >> 12) | (w[7] << 20)) & M26;
  h.v[9] = (w[7] >> 6) & M25;
  h.v[0] += 19u * (w[7] >> 31);
}

// fld.c:53-130 fld_reduce: unique representative in [0,p), every limb strictly below 1u.
// any f with limbs < 2^31.
ED_DEV void fe_canon(fe& t, const fe& f) {
  t = f;
  fe_carry(t);
  fe_carry(t);
  uint32_t q = (t.v[0] + 19u) >> 26;
#pragma unroll
  for (int i = 1; i < 10; i++) q = (t.v[i] + q) >> limb_bits(i);
  t.v[0] += 19u * q;                             // q = 1 iff t >= p
  uint32_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c = t.v[i] >> limb_bits(i);
    t.v[i] &= limb_mask(i);
    t.v[i + 1] += c;
  }
  t.v[9] &= M25;
}

// fld.c:162-178 fld_export.
ED_DEV void fe_tobytes(uint32_t w[8], const fe& f) {
  fe t;
  fe_canon(t, f);
  w[0] = t.v[0] | (t.v[1] << 26);
  w[1] = (t.v[1] >> 6) | (t.v[2] << 19);
  w[2] = (t.v[2] >> 13) | (t.v[3] << 13);
  w[3] = (t.v[3] >> 19) | (t.v[4] << 6);
  w[4] = t.v[5] | (t.v[6] << 25);
  w[5] = (t.v[6] >> 7) | (t.v[7] << 19);
  w[6] = (t.v[7] >> 13) | (t.v[8] << 12);
  w[7] = (t.v[8] >> 20) | (t.v[9] << 6);
}

// is the canonical form of f zero?  (fld.c:546-568 fld_eq compares a-b with zero this way)
ED_DEV bool fe_iszero(const fe& f) {
  fe t;
  fe_canon(t, f);
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) r |= t.v[i];
  return r == 0;
}

ED_DEV uint32_t fe_parity(const fe& f) {
  fe t;
  fe_canon(t, f);
  return t.v[0] & 1u;
}

// h = flag ? g : h   (branch-free: v_cndmask per limb)
ED_DEV void fe_cmov(fe& h, const fe& g, bool flag) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = flag ? g.v[i] : h.v[i];
}

ED_DEV void fe_cswap(fe& a, fe& b, bool flag) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const uint32_t x = a.v[i], y = b.v[i];
    a.v[i] = flag ? y : x;
    b.v[i] = flag ? x : y;
  }
}

// z^(2^250-1) and z^11: the shared ladder of both nacl chains (fld.c:593-637, fld.c:670-704).
// z < 3.36u.
ED_DEV void fe_pow_2_250_m1(fe& out, fe& z11, const fe& z) {
  fe z2, z9, a5, a10, a50, a100, t;
  fe_sq(z2, z);
  fe_sqn(t, z2, 2);
  fe_mul(z9, t, z);
  fe_mul(z11, z9, z2);
  fe_sq(t, z11);
  fe_mul(a5, t, z9);                    // 2^5 - 1
  fe_sqn(t, a5, 5);    fe_mul(a10, t, a5);
  fe_sqn(t, a10, 10);  fe_mul(z2, t, a10);      // z2 := 2^20 - 1
  fe_sqn(t, z2, 20);   fe_mul(t, t, z2);        // 2^40 - 1
  fe_sqn(t, t, 10);    fe_mul(a50, t, a10);
  fe_sqn(t, a50, 50);  fe_mul(a100, t, a50);
  fe_sqn(t, a100, 100); fe_mul(t, t, a100);     // 2^200 - 1
  fe_sqn(t, t, 50);    fe_mul(out, t, a50);     // 2^250 - 1
}

// fld.c:578-645 fld_inv: z^(p-2); inv(0) = 0.
ED_DEV void fe_inv(fe& h, const fe& z) {
  fe t, z11;
  fe_pow_2_250_m1(t, z11, z);
  fe_sqn(t, t, 5);
  fe_mul(h, t, z11);
}

// fld.c:657-709 fld_pow2523: z^((p-5)/8).
ED_DEV void fe_pow2523(fe& h, const fe& z) {
  fe t, z11;
  fe_pow_2_250_m1(t, z11, z);
  fe_sqn(t, t, 2);
  fe_mul(h, t, z);
}

}  // namespace ed
