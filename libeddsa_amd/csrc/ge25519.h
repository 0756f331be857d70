// ge25519.h - the twisted Edwards group -x^2 + y^2 = 1 + d x^2 y^2 on the device.
//
// Replaces the reference's lib/ed.c.  Decompression (ed.c:100-149) and compression
// (ed.c:155-169) reproduce the reference's permissive semantics exactly.  The group law itself
// is free to differ (outputs depend only on the mathematical point as long as the
// input is ON the curve, and the unified a = -1 formulas are complete there), so:
//   * doubling is dbl-2008-hwcd (4 S + 3 M, + 1 M when T is wanted), not ed.c:211-237's 4 S + 5 M;
//   * additions take the second operand in "cached" form (Y-X, Y+X, 2dT, 2Z): 8 M
//     (ed.c:175-203 ed_add spends 9 M), or affine "niels" form (y-x, y+x, 2dxy): 7 M as
//     ed.c:282-305 ed_add_pc.
// Bounds follow fe25519.h; every coordinate of a `ge` is tight.
#pragma once
#include "fe25519.h"

namespace ed {

struct ge { fe X, Y, Z, T; };
struct ge_cached { fe ymx, ypx, t2d, z2; };
struct ge_niels { fe ymx, ypx, t2d; };

// fld.c:23-41 con_d, con_2d, con_j and ed.c:46-52 (base point), radix 2^25.5
ED_DEV fe fe_const_d() { return fe{{56195235, 13857412, 51736253, 6949390, 114729, 24766616, 60832955, 30306712, 48412415, 21499315}}; }
ED_DEV fe fe_const_inv_d() { return fe{{30013507, 3972531, 42321084, 12719050, 2979674, 28954470, 51415654, 29910370, 18959708, 16925179}}; }   // 1 / d
ED_DEV fe fe_const_2d() { return fe{{45281625, 27714825, 36363642, 13898781, 229458, 15978800, 54557047, 27058993, 29715967, 9444199}}; }
ED_DEV fe fe_const_sqrtm1() { return fe{{34513072, 25610706, 9377949, 3500415, 12389472, 33281959, 41962654, 31548777, 326685, 11406482}}; }
ED_DEV fe fe_const_bx() { return fe{{52811034, 25909283, 16144682, 17082669, 27570973, 30858332, 40966398, 8378388, 20764389, 8758491}}; }
ED_DEV fe fe_const_by() { return fe{{40265304, 26843545, 13421772, 20132659, 26843545, 6710886, 53687091, 13421772, 40265318, 26843545}}; }

ED_DEV void ge_neutral(ge& p) {                  // ed.c:72 ed_zero
  fe_set(p.X, 0); fe_set(p.Y, 1); fe_set(p.Z, 1); fe_set(p.T, 0);
}

ED_DEV void ge_base(ge& p) {
  p.X = fe_const_bx(); p.Y = fe_const_by(); fe_set(p.Z, 1);
  fe_mul(p.T, p.X, p.Y);
}

// ed.c:100-149 ed_import.  Never fails.  y is taken mod p without a range check, the sign bit is
// applied even when x = 0, and when neither beta nor j*beta is a root the reference keeps
// j*beta (the result is then not a curve point): `oncurve` reports that case to the caller.
// negate = true returns -P instead (ed25519-sha512.c:174-175 negates A right after importing).
ED_DEV void ge_frombytes(ge& p, bool& oncurve, const uint32_t w[8], bool negate) {
  uint32_t yw[8];
#pragma unroll
  for (int i = 0; i < 8; i++) yw[i] = w[i];
  const uint32_t sign = yw[7] >> 31;
  yw[7] &= 0x7fffffffu;
  fe_frombytes(p.Y, yw);                        // tight (value < 2^255)

  fe one, u, v, a, b, chk;
  fe_set(one, 1);
  fe_sq(u, p.Y);                                // y^2, tight
  fe_mul(v, u, fe_const_d());
  v.v[0] += 1;                                  // v = d y^2 + 1, tight + 1
  fe_sub(u, u, one);                            // u = y^2 - 1, < 3u
  fe_sq(a, v);                                  // v^2
  fe_sq(b, a);                                  // v^4
  fe_mul(a, a, u);                              // g = u < 3u: ok
  fe_mul(a, a, v);                              // a = u v^3
  fe_mul(b, b, a);                              // u v^7
  fe_pow2523(b, b);
  fe_mul(b, b, a);                              // beta
  fe_sq(a, b);
  fe_mul(a, a, v);                              // v beta^2, tight
  fe_sub4(chk, a, u);                           // u < 3u needs the 4p bias
  const bool root1 = fe_iszero(chk);            // v beta^2 == u
  fe_add(chk, a, u);
  const bool root2 = fe_iszero(chk);            // v beta^2 == -u  => (j beta)^2 v == u
  oncurve = root1 || root2;
  fe_mul(a, b, fe_const_sqrtm1());
  p.X = a;
  fe_cmov(p.X, b, root1);
  const bool flip = ((fe_parity(p.X) ^ sign) != 0) != negate;
  fe_neg(a, p.X);                               // < 2u
  fe_carry(a);
  fe_cmov(p.X, a, flip);
  fe_mul(p.T, p.X, p.Y);
  fe_set(p.Z, 1);
}

// ed.c:155-169 ed_export
ED_DEV void ge_tobytes(uint32_t w[8], const ge& p) {
  fe zi, x, y;
  fe_inv(zi, p.Z);
  fe_mul(x, p.X, zi);
  fe_mul(y, p.Y, zi);
  fe_tobytes(w, y);
  w[7] |= fe_parity(x) << 31;
}

// r = 2p.  dbl-2008-hwcd with a = -1, all four outputs negated (same projective point):
//   e = 2XY, g = YY - XX, f = XX - YY + 2ZZ, h = XX + YY;  (X3,Y3,Z3,T3) = (f e, g h, f g, e h)
ED_DEV void ge_dbl(ge& r, const ge& p, bool need_t) {
  fe xx, yy, zz2, s, h, e, g, f;
  fe_sq(xx, p.X);
  fe_sq(yy, p.Y);
  fe_sq(zz2, p.Z);
  fe_add(zz2, zz2, zz2);                        // 2u
  fe_add(s, p.X, p.Y);                          // 2u
  fe_sq(s, s);
  fe_add(h, xx, yy);
  fe_carry(h);                                  // tight, so e fits the g-operand bound
  fe_sub(e, s, h);                              // 3u
  fe_sub(g, yy, xx);                            // 3u
  fe_sub4(f, zz2, g);                           // xx - yy + 2zz = 2zz - g, < 6u: first operand only
  // second operands are e (X, T) and g (Y, Z): their 19x premultiplies are computed once each
  fe_mul(r.X, f, e);
  fe_mul(r.Y, h, g);
  fe_mul(r.Z, f, g);
  if (need_t) fe_mul(r.T, h, e);
}

// r = p + q, q cached.  add-2008-hwcd-3 (the shape of ed.c:175-203 with 2dT2 and 2Z2 premultiplied)
ED_DEV void ge_add_cached(ge& r, const ge& p, const ge_cached& q, bool need_t) {
  fe a, b, c, d, e, f, g, h;
  fe_sub(a, p.Y, p.X);                          // 3u
  fe_mul(a, a, q.ymx);
  fe_add(b, p.Y, p.X);                          // 2u
  fe_mul(b, b, q.ypx);
  fe_mul(c, p.T, q.t2d);
  fe_mul(d, p.Z, q.z2);
  fe_sub(e, b, a);                              // 3u
  fe_sub(f, d, c);                              // 3u
  fe_add(g, d, c);                              // 2u
  fe_add(h, b, a);                              // 2u
  // second operands are f (X, Z) and h (Y, T): shared 19x premultiplies
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, g, f);
  if (need_t) fe_mul(r.T, e, h);
}

// r = p + q, q affine niels (ed.c:282-305 ed_add_pc)
ED_DEV void ge_add_niels(ge& r, const ge& p, const ge_niels& q, bool need_t) {
  fe a, b, c, d, e, f, g, h;
  fe_sub(a, p.Y, p.X);
  fe_mul(a, a, q.ymx);
  fe_add(b, p.Y, p.X);
  fe_mul(b, b, q.ypx);
  fe_mul(c, p.T, q.t2d);                        // q.t2d < 2u (may be a negated entry)
  fe_add(d, p.Z, p.Z);                          // 2u
  fe_sub(e, b, a);                              // 3u
  fe_sub(f, d, c);                              // 4u: first operand only
  fe_add(g, d, c);                              // 3u
  fe_add(h, b, a);                              // 2u
  // second operands are e (X, T) and g (Y, Z): shared 19x premultiplies
  fe_mul(r.X, f, e);
  fe_mul(r.Y, h, g);
  fe_mul(r.Z, f, g);
  if (need_t) fe_mul(r.T, h, e);
}

// cached form of p, every coordinate tight (ed.c:436-442 ed_precompute, plus 2Z)
ED_DEV void ge_to_cached(ge_cached& c, const ge& p) {
  fe_sub(c.ymx, p.Y, p.X); fe_carry(c.ymx);
  fe_add(c.ypx, p.Y, p.X); fe_carry(c.ypx);
  fe_mul(c.t2d, p.T, fe_const_2d());
  fe_add(c.z2, p.Z, p.Z);  fe_carry(c.z2);
}

// -q for table entries (ed.c:386-390: swap diff/sum, negate prod), applied when flag is set
// r = the affine point (x, y) of a niels entry (y-x, y+x, 2dxy), i.e. the neutral element plus the entry, as
// (X : Y : Z : T) = (4x : 4y : 4 : 4xy): one multiplication where the general addition has seven.  The neutral entry
// (1, 1, 0) gives (0 : 4 : 4 : 0).
ED_DEV void ge_from_niels(ge& r, const ge_niels& q) {
  fe e, h;
  fe_sub(e, q.ypx, q.ymx);                       // 2x, 3u
  fe_add(h, q.ypx, q.ymx);                       // 2y, 2u
  fe_mul(r.T, e, h);                             // 4xy
  fe_add(r.X, e, e); fe_carry(r.X);              // 4x, tight
  fe_add(r.Y, h, h); fe_carry(r.Y);              // 4y, tight
  fe_set(r.Z, 4);
}
ED_DEV void ge_cached_cneg(ge_cached& q, bool flag) {
  fe_cswap(q.ymx, q.ypx, flag);
  fe n;
  fe_neg(n, q.t2d);                             // < 2u: fine as a g operand
  fe_cmov(q.t2d, n, flag);
}
ED_DEV void ge_niels_cneg(ge_niels& q, bool flag) {
  fe_cswap(q.ymx, q.ypx, flag);
  fe n;
  fe_neg(n, q.t2d);
  fe_cmov(q.t2d, n, flag);
}

// affine niels form of p with canonical limbs (one inversion): the shape of the reference's
// generated table entries (lib/ed_lookup64.h)
ED_DEV void ge_to_niels_affine(ge_niels& n, const ge& p) {
  fe zi, x, y, t;
  fe_inv(zi, p.Z);
  fe_mul(x, p.X, zi);
  fe_mul(y, p.Y, zi);
  fe_mul(t, x, y);
  fe_sub(n.ymx, y, x); fe_canon(n.ymx, n.ymx);
  fe_add(n.ypx, y, x); fe_canon(n.ypx, n.ypx);
  fe_mul(n.t2d, t, fe_const_2d()); fe_canon(n.t2d, n.t2d);
}

}  // namespace ed
