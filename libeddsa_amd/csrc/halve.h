// halve.h - half-length scalars for Ed25519 verification, one item per lane.
//
// The reference checks  encode(S*B - t*A) == R-bytes  (ed25519-sha512.c:166-180), which costs 252 doublings
// per item because t is a 253-bit scalar on a variable base.  For an item whose A and R both decode to curve
// points (R strictly: lanes.h verify_half_point_lane) that check is the group equation Q := S*B - t*A - R' = 0,
// and for any integer u coprime to the group order 8l it is equivalent to u*Q = 0:
//     (u S mod l)*B  -  v*A  -  u*R'  =  0        with  v = u t  (mod 8 l)
// (B has order l; A and R' have orders dividing 8l, so v*A = u*t*A also when A has a torsion component).
// The extended Euclidean algorithm on (8l, t), stopped half way, gives such a pair with |u|, v < 2^134
// (Antipa, Brown, Gallant, Lambert, Struik, Vanstone: "Accelerated verification of ECDSA signatures", SAC 2005,
// there mod the prime group order; mod 8l and with u odd here so that the equivalence is exact for every input
// on the curve, which is what bit-exactness with the reference's cofactorless check needs).  The evaluation is
// then 132 doublings and three half-length scalars.  Items for which no suitable pair turns up in the two
// candidates examined, or whose expansion has a quotient of 31 bits or more before that which the Lehmer rounds
// have not already used up in partial steps (together about 1 in 10^4 random t), are handed to the exact path like off-curve keys.
//
// Public data only (verification): control flow and timing depend on t.
#pragma once
#include "sc25519.h"

namespace ed {

// host test build only: how many Lehmer rounds and how many plain iterations ran (tests/test_device_source_on_host.py)
#ifdef ED_HOST_CHECK
inline thread_local long halve_counters[2];   // (per thread: the sanitizer build of the host side runs this source on several threads)
#define HALVE_COUNT(i) (++::ed::halve_counters[i])
// fault injection (tests only): n > 0 makes the n-th working half-step from now on take a quotient that is one too
// large - the failure the floating-point margins exist to exclude (the remainder wraps; the search then usually runs
// into its own give-up rules) - and n < 0 makes the |n|-th working half-step update the COFACTOR with a quotient one
// too large (remainders, hence the choice of the pair, stay right; the pair's congruence is broken).  Whatever comes
// out, lanes.h: verify_half_scalars_lane must not let a wrong pair through.
inline thread_local int halve_fault;
#define HALVE_FAULT(q, active) do { if ((active) && ::ed::halve_fault > 0 && --::ed::halve_fault == 0) (q) += 1u; } while (0)
#define HALVE_FAULT_COFACTOR(q, active) do { if ((active) && ::ed::halve_fault < 0 && ++::ed::halve_fault == 0) (q) += 1u; } while (0)
#else
#define HALVE_COUNT(i) ((void)0)
#define HALVE_FAULT(q, active) ((void)0)
#define HALVE_FAULT_COFACTOR(q, active) ((void)0)
#endif

constexpr int HALF_WINDOWS = 34;                 // 4-bit signed windows of v and |u|: values < 2^134
constexpr int HALF_BITS = 134;
constexpr int HALF_BITS_SMALL = 138;             // the bound for passes whose main kernel is shorter than the exact path: 35 windows
constexpr int HALF_WINDOWS_SMALL = 35;

// 8 l, little-endian words
ED_DEV constexpr uint32_t halve_N(int i) {
  constexpr uint32_t N[8] = {0xe7ae9f68u, 0xc09318d2u, 0x17bce6b2u, 0xa6f7cef5u, 0x00000000u, 0x00000000u, 0x00000000u, 0x80000000u};
  return N[i];
}

ED_DEV double halve_to_double(const uint32_t w[8]) {
  double d = (double)w[7];
#pragma unroll
  for (int k = 6; k >= 0; k--) d = __builtin_fma(d, 4294967296.0, (double)w[k]);
  return d;                                      // relative error < 2^-50
}

// a < b as 256-bit numbers
ED_DEV bool halve_less(const uint32_t a[8], const uint32_t b[8]) {
  int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { c += (int64_t)a[k] - (int64_t)b[k]; c >>= 32; }
  return c != 0;
}

// a < 2^bits
ED_DEV bool halve_below(const uint32_t a[8], int bits) {
  uint32_t hi = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (32 * k >= bits) hi |= a[k];
    else if (32 * (k + 1) > bits) hi |= a[k] >> (bits - 32 * k);
  }
  return hi == 0;
}

// a / b for a >= 0, b > 0, to a relative error far below the 2^-40 margin of its caller: reciprocal estimate
// (v_rcp_f64) and two Newton steps instead of the correctly rounded division's twenty instructions
ED_DEV double halve_ratio(double a, double b) {
#ifdef ED_HOST_CHECK
  return a / b;
#else
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  return a * y;
#endif
}

// One half-step of the Euclidean algorithm: (ra, ua) -= q * (rb, ub) with q = an underestimate (>= 1) of
// floor(ra / rb), for lanes with `active` set (which have ra >= rb > 0); q = 0 otherwise.  da, db = ra, rb as
// doubles (halve_to_double).  ua, ub: 160-bit two's complement.  Returns true, and changes nothing, when the
// quotient does not fit 31 bits (one step in 2^31: the caller gives the item up).
ED_DEV bool halve_reduce(uint32_t ra[8], uint32_t ua[5], const uint32_t rb[8], const uint32_t ub[5], double da, double db, bool active) {
  db = active ? db : 1.0;
  double qd = __builtin_floor(halve_ratio(da, db) * (1.0 - 0x1p-40));   // never above the true quotient
  const bool big = active && !(qd < 2147483647.0);
  qd = qd < 2147483647.0 ? qd : 0.0;
  uint32_t q = (uint32_t)qd;
  q = q < 1u ? 1u : q;
  q = active && !big ? q : 0u;
  HALVE_FAULT(q, active && !big);
  uint64_t carry = 0;
  int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    carry += (uint64_t)q * rb[k];
    c += (int64_t)ra[k] - (int64_t)(uint32_t)carry;
    ra[k] = (uint32_t)c;
    c >>= 32;
    carry >>= 32;
  }
  ED_CHECK(c == 0 && carry == 0);                // q <= floor(ra / rb): no borrow out
  HALVE_FAULT_COFACTOR(q, active && !big);
  carry = 0;
  c = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    carry += (uint64_t)q * ub[k];
    c += (int64_t)ua[k] - (int64_t)(uint32_t)carry;
    ua[k] = (uint32_t)c;
    c >>= 32;
    carry >>= 32;
  }
  return big;
}

// out = a * x - b * y for W-word numbers, a, b < 2^24.  W = 8: the result is known to lie in [0, 2^256) (checked
// on the host build); W = 5: 160-bit two's complement, modulo 2^160.
template <int W>
ED_DEV void halve_combine(uint32_t out[W], uint32_t a, const uint32_t x[W], uint32_t b, const uint32_t y[W]) {
  uint64_t ca = 0, cb = 0;
  int64_t c = 0;
#pragma unroll
  for (int k = 0; k < W; k++) {
    ca += (uint64_t)a * x[k];
    cb += (uint64_t)b * y[k];
    c += (int64_t)(uint32_t)ca - (int64_t)(uint32_t)cb;
    out[k] = (uint32_t)c;
    c >>= 32;
    ca >>= 32;
    cb >>= 32;
  }
  ED_CHECK(W != 8 || c + (int64_t)ca - (int64_t)cb == 0);
}

constexpr int HALF_LEHMER_MIN_BITS = 151;        // rounds run while both remainders have more bits than this
// One round of Lehmer's acceleration: many Euclidean half-steps on the remainders as doubles, then ONE exact update
// of the long numbers.  With nonnegative integers a0, b0, a1, b1 (kept below 2^24 as doubles)
//     R0' = a0 R0 - b0 R1,   R1' = b1 R1 - a1 R0      (and the same combinations of U0, U1)
// a half-step "R0' -= q R1'" is a0 += q a1, b0 += q b1.  d0, d1 carry a relative error below 2^-50, so the doubles
// y0, y1 that track R0', R1' are off by less than (a + b) max(d0, d1) 2^-50 < E := max(d0, d1) 2^-24; every
// quotient is taken as floor((y0 - E) / (y1 + E)), which cannot exceed the true one, so each half-step applied is a
// legitimate (possibly partial) Euclidean step and the true remainders stay nonnegative.  A half-step is refused
// when it cannot be shown to leave its remainder above 2^141 - nothing that halve_scalar_lane's rule would have
// to examine (remainders below 2^134, or 2^138) is ever produced here; the plain loop does the last bits.
// Returns whether anything was applied.
ED_DEV bool halve_lehmer_round(uint32_t r0[8], uint32_t r1[8], uint32_t u0[5], uint32_t u1[5], double& d0, double& d1) {
  const double E = (d0 > d1 ? d0 : d1) * 0x1p-24;
  double y0 = d0, y1 = d1, a0 = 1.0, b0 = 0.0, a1 = 0.0, b1 = 1.0;
  bool progress = false, live = true;
  for (int it = 0; it < 40 && live; it++) {
    bool any = false;
    {                                            // R0' by R1'
      const double q = __builtin_floor(halve_ratio(y0 - E, y1 + E) * (1.0 - 0x1p-40));
      const double na = __builtin_fma(q, a1, a0), nb = __builtin_fma(q, b1, b0), ny = __builtin_fma(-q, y1, y0);
      if (q >= 1.0 && na < 0x1p24 && nb < 0x1p24 && ny - E >= 0x1p141) { a0 = na; b0 = nb; y0 = ny; any = true; }
    }
    {                                            // R1' by R0'
      const double q = __builtin_floor(halve_ratio(y1 - E, y0 + E) * (1.0 - 0x1p-40));
      const double na = __builtin_fma(q, a0, a1), nb = __builtin_fma(q, b0, b1), ny = __builtin_fma(-q, y0, y1);
      if (q >= 1.0 && na < 0x1p24 && nb < 0x1p24 && ny - E >= 0x1p141) { a1 = na; b1 = nb; y1 = ny; any = true; }
    }
    live = any;
    progress = progress || any;
  }
  if (!progress) return false;
  uint32_t t0[8], t1[8], v0[5], v1[5];
  halve_combine<8>(t0, (uint32_t)a0, r0, (uint32_t)b0, r1);
  halve_combine<8>(t1, (uint32_t)b1, r1, (uint32_t)a1, r0);
  halve_combine<5>(v0, (uint32_t)a0, u0, (uint32_t)b0, u1);
  halve_combine<5>(v1, (uint32_t)b1, u1, (uint32_t)a1, u0);
#pragma unroll
  for (int k = 0; k < 8; k++) { r0[k] = t0[k]; r1[k] = t1[k]; }
#pragma unroll
  for (int k = 0; k < 5; k++) { u0[k] = v0[k]; u1[k] = v1[k]; }
  d0 = halve_to_double(r0);
  d1 = halve_to_double(r1);
  return true;
}

// |u| < 2^BITS for a 160-bit two's complement u; mag = |u|, neg = u < 0
template <int BITS = HALF_BITS>
ED_DEV bool halve_magnitude(uint32_t mag[5], bool& neg, const uint32_t u[5]) {
  neg = (u[4] >> 31) != 0;
  int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    c += neg ? -(int64_t)u[k] : (int64_t)u[k];
    mag[k] = (uint32_t)c;
    c >>= 32;
  }
  return (mag[4] >> (BITS - 128)) == 0;
}

// t (< l, eight words) -> v (five words, 0 <= v < 2^BITS), |u| (five words, < 2^BITS, u odd), uneg = (u < 0)
// with v = u t (mod 8 l).  Returns false when no such pair is among the candidates examined.  BITS = 134 (34 windows)
// gives up on 8.5 t in 10^5, BITS = 138 (35 windows; an even u is retried when r >= 2^118) on 2 in 10^7.
template <int BITS = HALF_BITS>
ED_DEV bool halve_scalar_lane(uint32_t vw[5], uint32_t uw[5], bool& uneg, const uint32_t tw[8]) {
  uint32_t r0[8], r1[8], u0[5], u1[5];
#pragma unroll
  for (int k = 0; k < 8; k++) { r0[k] = halve_N(k); r1[k] = tw[k]; }
#pragma unroll
  for (int k = 0; k < 5; k++) { u0[k] = 0; u1[k] = 0; }
  u1[0] = 1;
  // the candidate is the smaller remainder with its cofactor; which = 1: (r1, u1), 0: (r0, u0)
  bool done = false, good = false, tried = false;
  int which = 1;
  // examine the candidate a completed step has produced (lanes with `fresh` set)
  auto examine = [&](const uint32_t r[8], const uint32_t u[5], bool fresh, int idx) {
    if (!fresh || done) return;
    if (!halve_below(r, BITS)) return;
    if (u[0] & 1u) {
      uint32_t m[5]; bool ng;
      good = halve_magnitude<BITS>(m, ng, u);
      done = true; which = idx;
    } else if (tried || halve_below(r, 256 - BITS)) {   // |u'| <= 8l / r < 2^BITS needs r >= 2^(256 - BITS)
      done = true; good = false; which = idx;
    } else {
      tried = true;
    }
  };
  examine(r1, u1, true, 1);
  double d0 = halve_to_double(r0), d1 = halve_to_double(r1);
  // the bulk of the way down by Lehmer rounds (about 23 bits each) ...
  for (int round = 0; round < 8; round++) {
    if (done || halve_below(r0, HALF_LEHMER_MIN_BITS) || halve_below(r1, HALF_LEHMER_MIN_BITS)) break;
    if (!halve_lehmer_round(r0, r1, u0, u1, d0, d1)) break;
    HALVE_COUNT(0);
  }
  // ... and the rest step by step, every completed step examined; each half-step's comparison also tells the next
  // one whether it has work
  bool act = !done && !halve_less(r0, r1);       // the half-step about to run has ra >= rb
  for (int it = 0; it < 160; it++) {
    if (done) break;
    HALVE_COUNT(1);
    {                                            // r0 by r1
      if (halve_reduce(r0, u0, r1, u1, d0, d1, act)) { done = true; good = false; }
      d0 = halve_to_double(r0);
      const bool lt = halve_less(r0, r1);        // the step is complete
      examine(r0, u0, act && lt, 0);
      act = !done && lt;                         // then r1 > r0: the other half-step has work
    }
    {                                            // r1 by r0
      if (halve_reduce(r1, u1, r0, u0, d1, d0, act)) { done = true; good = false; }
      d1 = halve_to_double(r1);
      const bool lt = halve_less(r1, r0);
      examine(r1, u1, act && lt, 1);
      act = !done && !halve_less(r0, r1);        // r0 >= r1: also after a half-step that had no work or left a partial quotient
    }
  }
  good = good && done;
  uint32_t usel[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    vw[k] = which ? r1[k] : r0[k];
    usel[k] = which ? u1[k] : u0[k];
  }
  halve_magnitude<BITS>(uw, uneg, usel);
  return good;
}

}  // namespace ed
