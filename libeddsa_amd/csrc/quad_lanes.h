// quad_lanes.h - four lanes per item: the exact (reference-order) verify chain and the windowed evaluation for small passes.
//
// lanes.h: ref_dual_scale_chain replays the reference's JSF/Shamir chain (ed.c:455-507) for public
// keys that are not curve points with one lane per item: 261 steps of "uniform addition, then
// doubling", 18 dependent field multiplications per step -- 1.3 ms on an idle chip however few the
// items are, 4 ms beside k_verify_main.  That latency is what a pass of up to 2^19 items waits
// for.  Here the four coordinates of the accumulator live in the four lanes of a quad (lane & 3 =
// 0: X, 1: Y, 2: T, 3: Z) and the four multiplications of each stage of the addition law run side
// by side:
//
//   ed_add (ed.c:175-203)                            lane 0        lane 1        lane 2        lane 3
//   stage A   one fe_mul                             a=(Y-X)(y-x)  b=(Y+X)(y+x)  c=T*(2d*t)    d=Z*(2z)
//   stage B   e=b-a f=d-c g=d+c h=b+a, one fe_mul    X=f*e         Y=h*g         T=h*e         Z=f*g
//   ed_double (ed.c:211-237): stage A is a squaring of (Y-X, Y+X, T, Z), followed by one fe_mul
//   with the lane's constant (1, 1, 2d, 2); stage B is the same.
//
// The addend's factors (y-x, y+x, 2d*t, -2d*t, 2z) are loop invariants: the set-up (the same kernel,
// verify_exact_setup_quad below) stores them per item and a lane loads the one its role and the digit's sign select (negating the addend
// swaps y-x with y+x and negates 2d*t: ed_sub, ed.c:245-273).  Every value is a field element and
// multiplication in GF(p) is associative and commutative, so T*(2d*t) is the reference's (T*t)*2d
// and Z*(2z) its 2*(Z*z): the accumulator holds the reference's coordinates mod p (up to the common
// factor discussed in lanes.h: verify_exact_setup_reuse_lane) after every step; the arguments of
// lanes.h: ref_dual_scale_chain for the uniform control flow apply unchanged.
// Lanes exchange values with DPP quad permutations (v_mov_b32_dpp, no LDS).
//
// A step is about 1200 instructions instead of 2800 (the exchanges and selects cost 45 % on top of
// the five multiplications), the chain 0.7 ms on an idle chip.  Device only: this file is not part
// of the -DED_HOST_CHECK build; it is covered on the GPU by the parity tests (golden edge cases,
// random off-curve keys against the oracle at several batch sizes, and self-check mode 2, which
// sends genuine signatures through it).
#pragma once
#include "lanes.h"

namespace ed {

#define QUAD_VALUE_WORDS 12                            /* 10 limbs + 2 padding words: three 16-byte loads */
#define QUAD_ADDEND_WORDS (5 * QUAD_VALUE_WORDS)       /* y-x | y+x | 2d*t | -2d*t | 2z */
#define QUAD_ITEM_WORDS (4 * QUAD_ADDEND_WORDS)        /* the scratchpad of an item (HBM): Q, B, Q+B, Q-B */
#define QUAD_DIGIT_WORDS EXACT_DIGIT_WORDS              /* the digit pairs of an item (LDS): REF_JSF_LEN nibbles, eight per word */

// ---- set-up: one lane per item -------------------------------------------------------------------

ED_DEV void quad_value_store(uint32_t* dst, const fe& v) {
  word4* p = reinterpret_cast<word4*>(dst);
  p[0] = word4{v.v[0], v.v[1], v.v[2], v.v[3]};
  p[1] = word4{v.v[4], v.v[5], v.v[6], v.v[7]};
  p[2] = word4{v.v[8], v.v[9], 0, 0};
}

ED_DEV void quad_value_load(fe& v, const uint32_t* src) {
  const word4* p = reinterpret_cast<const word4*>(src);
  const word4 a = p[0], b = p[1], c = p[2];
  v.v[0] = a.x; v.v[1] = a.y; v.v[2] = a.z; v.v[3] = a.w;
  v.v[4] = b.x; v.v[5] = b.y; v.v[6] = b.z; v.v[7] = b.w;
  v.v[8] = c.x; v.v[9] = c.y;
}

// the five factors of one addend given as an extended point, every one tight
ED_DEV void quad_addend_store(uint32_t* dst, const ge& p) {
  fe t;
  fe_sub(t, p.Y, p.X); fe_carry(t); quad_value_store(dst, t);
  fe_add(t, p.Y, p.X); fe_carry(t); quad_value_store(dst + QUAD_VALUE_WORDS, t);
  fe_mul(t, p.T, fe_const_2d());    quad_value_store(dst + 2 * QUAD_VALUE_WORDS, t);
  fe_neg(t, t); fe_carry(t);        quad_value_store(dst + 3 * QUAD_VALUE_WORDS, t);
  fe_add(t, p.Z, p.Z); fe_carry(t); quad_value_store(dst + 4 * QUAD_VALUE_WORDS, t);
}

// The loop invariants of ed.c:455-478 from what k_verify_prepare left in the workspace (the digit words and entry 1
// of the item's table), as lanes.h: verify_exact_setup_reuse_lane, shared out over the item's quad so that set-up and
// chain are ONE kernel and one latency (round 2 ran a one-lane set-up kernel before the chain: 0.75 + 0.75 ms, with
// 84 spilled registers and a digit string written to memory byte by byte):
//   lane 0   the joint sparse form of (S, t) (sc.c:297-324, limb boundaries included), one nibble per step -
//            (u0 + 1) | (u1 + 1) << 2 - eight steps per word of `dig` (LDS, QUAD_DIGIT_WORDS words)
//   lane 1   the addends Q = -A (every coordinate doubled, see lanes.h) and B
//   lane 2   Q + B            lane 3   Q - B        (ed.c:473-476), five factors each, at `item` (HBM)
// The branches run one after the other (a wave executes them under partial masks): about 12 k instructions against the
// chain's 300 k.
ED_DEV void verify_exact_setup_quad(const uint32_t* digits, const uint32_t* tab, const uint32_t* base1, uint32_t* item,
                                    uint32_t* dig, int q) {
  if (q == 0) {
    uint32_t tw[8], sw[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { tw[k] = digits[k]; sw[k] = digits[8 + k]; }
    words_sub_pattern(tw, 0x88888888u);
    words_sub_pattern(sw, 0x80008000u);
    exact_jsf_words(dig, 1, sw, tw);             // lanes.h: one nibble per step, QUAD_DIGIT_WORDS words
  } else {
    ge_cached c1;
    cached_load(c1, tab, 1);
    ge qp, p;                                    // Q = -A with every coordinate doubled
    fe_sub(qp.X, c1.ypx, c1.ymx); fe_carry(qp.X);
    fe_add(qp.Y, c1.ypx, c1.ymx); fe_carry(qp.Y);
    qp.Z = c1.z2;
    fe_mul(qp.T, c1.t2d, fe_const_inv_d());
    if (q == 1) {
      quad_addend_store(item, qp);
      ge_base(p);
      quad_addend_store(item + QUAD_ADDEND_WORDS, p);
    } else {
      ge_niels pcB;
      niels_load(pcB, base1);
      ref_add_pc(p, qp, pcB, q == 3);            // Q + B / Q - B
      quad_addend_store(item + (q == 2 ? 2 : 3) * QUAD_ADDEND_WORDS, p);
    }
  }
}

// ---- the chain: four lanes per item ------------------------------------------------------------

// quad permutation: lane l of every quad reads lane P_l
template <int P0, int P1, int P2, int P3>
ED_DEV void fe_quad_perm(fe& o, const fe& a) {
  constexpr int ctrl = P0 | (P1 << 2) | (P2 << 4) | (P3 << 6);
#pragma unroll
  for (int j = 0; j < 10; j++)
    o.v[j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[j], ctrl, 0xf, 0xf, true);   // (bound_ctrl: no `old` value to set up - every lane of a quad is a valid source)
}

// the first factor of stage A: (Y - X, Y + X, T, Z); r tight, result < 3u
ED_DEV void quad_stage_a_operand(fe& first, const fe& r, int q) {
  fe p, s, d;
  fe_quad_perm<1, 0, 2, 3>(p, r);                // lane 0 sees Y, lane 1 sees X
  fe_add(s, r, p);                               // 2u
  fe_sub(d, p, r);                               // 3u
  first = r;
  fe_cmov(first, d, q == 0);
  fe_cmov(first, s, q == 1);
}

// stage B: m = (a, b, c, d) across the quad, tight -> (X, Y, T, Z) = (f*e, h*g, h*e, f*g)
ED_DEV void quad_stage_b(fe& out, const fe& m, int q) {
  fe a, b, c, d, e, f, g, h;
  fe_quad_perm<0, 0, 0, 0>(a, m);
  fe_quad_perm<1, 1, 1, 1>(b, m);
  fe_quad_perm<2, 2, 2, 2>(c, m);
  fe_quad_perm<3, 3, 3, 3>(d, m);
  fe_sub(e, b, a);                               // 3u
  fe_sub(f, d, c);                               // 3u
  fe_add(g, d, c);                               // 2u
  fe_add(h, b, a);                               // 2u
  fe x = h, y = g;
  fe_cmov(x, f, q == 0 || q == 3);               // first factor  < 8u
  fe_cmov(y, e, q == 0 || q == 2);               // second factor < 3.36u
  fe_mul(out, x, y);
}

// ed_export (ed.c:155-169) of the point whose coordinates (X, Y, T, Z) sit in the four lanes of the quad: every lane
// inverts Z (fld_inv(0) = 0 as in the reference); lane 1 ends up with the 32 encoded bytes in wd (the other lanes with
// the encodings of x, t and 1 under the same parity bit: not used)
ED_DEV void quad_encode(uint32_t wd[8], const fe& r) {
  fe z, zi, aff;
  fe_quad_perm<3, 3, 3, 3>(z, r);
  fe_inv(zi, z);
  fe_mul(aff, r, zi);
  fe_tobytes(wd, aff);
  const uint32_t xpar = (uint32_t)__builtin_amdgcn_mov_dpp((int)wd[0], 0, 0xf, 0xf, true) & 1u;
  wd[7] |= xpar << 31;
}

// ed.c:479-506 for one item; `item` = its addends (HBM), `dig` = its digit pairs (LDS), q = lane & 3.  Leaves the
// encoding of the result in wd (lane 1 of the quad: quad_encode).
// A step whose digit pair is (0, 0) in EVERY item of the wave skips the addition altogether (the select would discard
// it anyway): in a small pass, where a wave carries one or two items, that is half the additions.
ED_DEV void exact_chain_encode_quad(uint32_t wd[8], const uint32_t* item, const uint32_t* dig, int q) {
  fe r, k;
  fe_set(r, (uint32_t)(q & 1));                  // neutral element (0, 1, 0, 1) as (X, Y, T, Z)
  fe_set(k, 1);                                  // the doubling's constants (1, 1, 2d, 2)
  { fe two; fe_set(two, 2); fe_cmov(k, fe_const_2d(), q == 2); fe_cmov(k, two, q == 3); }
  // The factor a step multiplies by is loaded one step ahead, so that its latency (microseconds beside the main
  // kernel's table traffic) hides behind the current step.
  int i = REF_JSF_LEN - 1;
  uint32_t w = dig[i >> 3];                      // word holding step i
  fe mult_next;
  bool skip_next;
  // decode step j from the current word and start the load of its factor
#define QUAD_PREPARE_STEP(j)                                                                        \
  {                                                                                                 \
    const uint32_t nib = (w >> (4 * ((j) & 7))) & 15u;                                              \
    const int da = (int)(nib & 3u) - 1, db = (int)(nib >> 2) - 1;                                   \
    const bool both = (da != 0) && (db != 0);                                                       \
    skip_next = (da == 0) && (db == 0);                                                             \
    /* which addend: 2 = Q+B (digits equal), 3 = Q-B (digits opposite), 1 = B, 0 = Q; negated? */   \
    const int which = both ? (da == db ? 2 : 3) : (da != 0 ? 1 : 0);                                \
    const int neg = (which == 2 ? (da < 0) : which == 3 ? (da > 0) : which == 1 ? (da < 0) : (db < 0)) ? 1 : 0; \
    const int v = q == 0 ? neg : q == 1 ? 1 - neg : q == 2 ? 2 + neg : 4;                           \
    quad_value_load(mult_next, item + QUAD_ADDEND_WORDS * which + QUAD_VALUE_WORDS * v);            \
  }
  QUAD_PREPARE_STEP(i)
#pragma unroll 1
  for (;;) {
    const fe mult = mult_next;
    const bool skip = skip_next;
    if (i > 0) {
      const int j = i - 1;
      if ((j & 7) == 7) w = dig[j >> 3];         // step j starts the next word (wave-uniform)
      QUAD_PREPARE_STEP(j)
    }
    fe first, m, sum;
    if (__any(!skip)) {                          // wave-uniform
      quad_stage_a_operand(first, r, q);
      fe_mul(m, first, mult);
      quad_stage_b(sum, m, q);
      fe_cmov(r, sum, !skip);
    }
    if (i == 0) break;
    quad_stage_a_operand(first, r, q);
    fe_sq(m, first);
    fe_mul(m, m, k);
    quad_stage_b(r, m, q);
    i--;
  }
#undef QUAD_PREPARE_STEP
  quad_encode(wd, r);
}

// ... and the byte comparison with R (ed25519-sha512.c:176-180): returns (in every lane, but lane 1 is the one that
// holds y) whether the encoding of the result equals R's bytes
ED_DEV bool verify_exact_chain_quad(const uint32_t rw[8], const uint32_t* item, const uint32_t* dig, int q) {
  uint32_t wd[8];
  exact_chain_encode_quad(wd, item, dig, q);
  uint32_t diff = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) diff |= wd[j] ^ rw[j];
  return diff == 0;
}

// ---------------------------------------------------------------------------------------------
// The windowed evaluation of lanes.h: verify_main_lane with four lanes per item, for SMALL passes:
// below about 2^15 items one lane per item leaves most of the chip idle and a pass costs the
// latency of one item (0.7 ms); with a coordinate per lane the doubling is a squaring and a
// multiplication deep and the addition two multiplications, 2250 instructions per window instead
// of 5100.  Same digits, same tables, same field expressions as the one-lane form (T is computed at
// every step here, in the lane that would otherwise idle), so the same projective result.
// ---------------------------------------------------------------------------------------------

struct alignas(8) word2 { uint32_t x, y; };

// ten words at an 8-byte aligned address (table entries start 16-byte aligned, coordinates 40 bytes apart)
ED_DEV void quad_coord_load(fe& v, const uint32_t* src) {
  const word2* p = reinterpret_cast<const word2*>(src);
#pragma unroll
  for (int j = 0; j < 5; j++) { const word2 w = p[j]; v.v[2 * j] = w.x; v.v[2 * j + 1] = w.y; }
}

// r = 2r (ge25519.h: ge_dbl, dbl-2008-hwcd with all outputs negated), r = (X, Y, T, Z) across the quad
ED_DEV void quad_dbl(fe& r, int q) {
  fe xs, ys, s, op, m;
  fe_quad_perm<0, 1, 0, 3>(xs, r);               // lane 2 sees X
  fe_quad_perm<0, 1, 1, 3>(ys, r);               // lane 2 sees Y
  fe_add(s, xs, ys);                             // lane 2: X + Y (2u)
  op = r;
  fe_cmov(op, s, q == 2);
  fe_sq(m, op);                                  // (XX, YY, (X+Y)^2, ZZ)
  fe xx, yy, ss, zz, h, e, g, f;
  fe_quad_perm<0, 0, 0, 0>(xx, m);
  fe_quad_perm<1, 1, 1, 1>(yy, m);
  fe_quad_perm<2, 2, 2, 2>(ss, m);
  fe_quad_perm<3, 3, 3, 3>(zz, m);
  fe_add(zz, zz, zz);                            // 2u
  fe_add(h, xx, yy);
  fe_carry(h);                                   // tight, so e fits the second-operand bound
  fe_sub(e, ss, h);                              // 3u
  fe_sub(g, yy, xx);                             // 3u
  fe_sub4(f, zz, g);                             // xx - yy + 2zz = 2zz - g, < 6u: first operand only
  fe x = h, y = g;
  fe_cmov(x, f, q == 0 || q == 3);               // (X, Y, T, Z) = (f e, h g, h e, f g)
  fe_cmov(y, e, q == 0 || q == 2);
  fe_mul(r, x, y);
}

// r += +-entry, entry = words (ymx | ypx | t2d [| z2]) at `e`; has_z2 = false: affine (z2 = 2)
ED_DEV void quad_add_entry(fe& r, const uint32_t* e, bool neg, bool has_z2, int q) {
  // lane 0 takes y-x (y+x when negated), lane 1 the other, lane 2 2d*t (negated), lane 3 2z
  // coordinate index: per-item entries (has_z2) hold packed coordinates of eight words, the shared tables ten limbs
  const int co = q == 0 ? (neg ? 1 : 0) : q == 1 ? (neg ? 0 : 1) : q == 2 ? 2 : (has_z2 ? 3 : 0);
  fe mult, nm, first, m;
  if (has_z2) {
    const word4* p = reinterpret_cast<const word4*>(e + 8 * co);
    const word4 a = p[0], b = p[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    fe_unpack(mult, w);
  } else {
    quad_coord_load(mult, e + 10 * co);
    fe two; fe_set(two, 2); fe_cmov(mult, two, q == 3);
  }
  fe_neg(nm, mult);                              // < 2u: within the second operand's bound
  fe_cmov(mult, nm, neg && q == 2);
  quad_stage_a_operand(first, r, q);
  fe_mul(m, first, mult);
  quad_stage_b(r, m, q);
}

// verify_main_lane for one item; digits = its 16 digit words, tab = its table, q = lane & 3.
// Leaves coordinate q of the result in r.
ED_DEV void verify_main_quad(fe& r, const uint32_t* digits, const uint32_t* tab, const uint32_t* base16, int q) {
  fe_set(r, (uint32_t)(q & 1));                  // neutral element (0, 1, 0, 1) as (X, Y, T, Z)
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {
    if (w != 63) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) quad_dbl(r, q);
    }
    {
      const int dig = (int)((digits[w >> 3] >> (4 * (w & 7))) & 15u) - 8;
      const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
      quad_add_entry(r, tab + mag * VERIFY_ENTRY_WORDS, dig < 0, true, q);
    }
    if ((w & 3) == 0) {
      const int j = w >> 2;                        // digit j of S sits at bit 16 j
      const int dig = (int)((digits[8 + (j >> 1)] >> (16 * (j & 1))) & 0xffffu) - 32768;
      const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
      quad_add_entry(r, base16 + TABLE_ENTRY_WORDS * mag, dig < 0, false, q);
    }
  }
}


// lanes.h: verify_half_main_lane (the half-length evaluation, halve.h) with four lanes per item, for small passes:
// same digits, tables and field expressions, so the same projective result; returns, in every lane of the quad,
// whether that result is the neutral element.
template <int WINDOWS = HALF_WINDOWS>
ED_DEV bool verify_half_main_quad(const uint32_t* hd, const uint32_t* tab_a, const uint32_t* tab_r, const uint32_t* base16, int q) {
  fe r;
  fe_set(r, (uint32_t)(q & 1));                  // neutral element (0, 1, 0, 1) as (X, Y, T, Z)
  const bool uneg = (hd[24] & 1u) != 0, is_long = (hd[24] & 2u) != 0;
  const int top = (__any(is_long) ? HALF_LONG_WINDOWS : WINDOWS) - 1;   // the wave's loop (lanes.h: verify_half_main_lane)
#pragma unroll 1
  for (int w = top; w >= 0; w--) {
    // the lane's coordinate of the window's two per-item entries: requested before the doublings, consumed after them
    uint32_t raw[2][8];
    bool neg[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int dig = (int)((hd[8 * h + (w >> 3)] >> (4 * (w & 7))) & 15u) - 8;
      const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
      neg[h] = (dig < 0) != (h == 0 && uneg);
      const int co = q == 0 ? (neg[h] ? 1 : 0) : q == 1 ? (neg[h] ? 0 : 1) : q;
      const word4* p = reinterpret_cast<const word4*>((h ? tab_r : tab_a) + mag * VERIFY_ENTRY_WORDS + 8 * co);
      const word4 a = p[0], b = p[1];
      raw[h][0] = a.x; raw[h][1] = a.y; raw[h][2] = a.z; raw[h][3] = a.w;
      raw[h][4] = b.x; raw[h][5] = b.y; raw[h][6] = b.z; raw[h][7] = b.w;
    }
    if (w != top) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) quad_dbl(r, q);
    }
#pragma unroll 1
    for (int h = 0; h < 2; h++) {                // h = 0: v on -A (sign of u applied), h = 1: |u| on -R'
      fe mult, nm, first, m;
      fe_unpack(mult, raw[h]);
      fe_neg(nm, mult);
      fe_cmov(mult, nm, neg[h] && q == 2);
      quad_stage_a_operand(first, r, q);
      fe_mul(m, first, mult);
      quad_stage_b(r, m, q);
    }
    if ((w & 3) == 0) {
      const int j = w >> 2;
#pragma unroll 1
      for (int h = 0; h < (j < 8 ? 2 : 1); h++) {
        const int jj = j + 8 * h;
        int dig = (int)((hd[16 + (jj >> 1)] >> (16 * (jj & 1))) & 0xffffu) - 32768;
        dig = (is_long ? h == 0 : j < 8) ? dig : 0;
        const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
        quad_add_entry(r, base16 + TABLE_ENTRY_WORDS * ((size_t)mag + (h ? (size_t)TABLE_BASE16_ENTRIES : 0)), dig < 0, false, q);
      }
    }
  }
  // X = 0 (lane 0), Y = Z (lane 1 against lane 3), Z != 0 (lane 3); lane 2 (T) has no say
  fe z, d;
  fe_quad_perm<3, 3, 3, 3>(z, r);
  fe_sub(d, r, z);
  const bool mine = q == 0 ? fe_iszero(r) : q == 1 ? fe_iszero(d) : q == 3 ? !fe_iszero(r) : true;
  const uint64_t votes = __ballot(mine);
  const unsigned lane = __lane_id();
  return ((votes >> (lane & ~3u)) & 0xfu) == 0xfu;
}

// ---------------------------------------------------------------------------------------------
// The same evaluation for the SMALLEST passes (up to EDK_SUMS_MAX_ITEMS items), where the 66 + 16 additions of an item
// are a third of its latency although none of them depends on the accumulator: the sum of a window's addends
//   S_w = sigma d_w (-A) + e_w (-R') [+ the base point's entries of windows 0, 4, 8, ...]
// is computed for all windows of all items AT ONCE (k_verify_window_sums: one quad per window, a few additions deep),
// and the chain is 132 doublings and 34 additions of a ready sum instead of 82 additions with lookups in between.
// The result is the same group element (the addition law is complete on the curve's points, the order of additions
// does not matter), so the same verdict.
// ---------------------------------------------------------------------------------------------

#define HALF_SUM_WORDS 40                              /* a sum as its four stage-A multipliers y-x | y+x | 2d*t | 2z, ten limbs each */

// S_w of one item as multipliers at `out`; q = lane & 3.  hd, tab_a, tab_r as for verify_half_main_quad.
template <int WINDOWS = HALF_WINDOWS>
ED_DEV void verify_half_window_sum_quad(uint32_t* out, const uint32_t* hd, const uint32_t* tab_a, const uint32_t* tab_r,
                                        const uint32_t* base16, int w, int q) {
  const bool uneg = (hd[24] & 1u) != 0, is_long = (hd[24] & 2u) != 0;
  fe r;
  if (w >= (is_long ? HALF_LONG_WINDOWS : WINDOWS)) {
    fe_set(r, q == 3 ? 2u : q == 2 ? 0u : 1u);   // the neutral element's multipliers (1, 1, 0, 2): a wave with a long item runs 64 windows
  } else {
    {                                            // the neutral element plus an entry: stage A is (y-x, y+x, 0, 2z) itself
      const int dig = (int)((hd[w >> 3] >> (4 * (w & 7))) & 15u) - 8;
      const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
      const bool neg = (dig < 0) != uneg;
      const int co = q == 0 ? (neg ? 1 : 0) : q == 1 ? (neg ? 0 : 1) : 3;
      const word4* p = reinterpret_cast<const word4*>(tab_a + mag * VERIFY_ENTRY_WORDS + 8 * co);
      const word4 a = p[0], b = p[1];
      const uint32_t pw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      fe m, zero;
      fe_unpack(m, pw);
      fe_carry(m);                               // (the packed top limb may hold a carry: tight again for the subtractions)
      fe_set(zero, 0);
      fe_cmov(m, zero, q == 2);
      quad_stage_b(r, m, q);
    }
    {
      const int dig = (int)((hd[8 + (w >> 3)] >> (4 * (w & 7))) & 15u) - 8;
      const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
      quad_add_entry(r, tab_r + mag * VERIFY_ENTRY_WORDS, dig < 0, true, q);
    }
    if ((w & 3) == 0) {                          // lanes.h: verify_half_main_lane - digit j of s' from k*B, digit 8 + j from k*2^128*B
      const int j = w >> 2;
#pragma unroll 1
      for (int h = 0; h < (j < 8 ? 2 : 1); h++) {
        const int jj = j + 8 * h;
        int dig = (int)((hd[16 + (jj >> 1)] >> (16 * (jj & 1))) & 0xffffu) - 32768;
        dig = (is_long ? h == 0 : j < 8) ? dig : 0;
        const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
        quad_add_entry(r, base16 + TABLE_ENTRY_WORDS * ((size_t)mag + (h ? (size_t)TABLE_BASE16_ENTRIES : 0)), dig < 0, false, q);
      }
    }
    fe first, t2d, z2;                           // (X, Y, T, Z) -> (Y - X, Y + X, 2d T, 2 Z)
    quad_stage_a_operand(first, r, q);           // < 3u
    fe_mul(t2d, first, fe_const_2d());
    fe_add(z2, first, first);                    // lane 3: 2Z, 2u
    r = first;
    fe_cmov(r, t2d, q == 2);
    fe_cmov(r, z2, q == 3);
    fe_carry(r);
  }
  word2* o = reinterpret_cast<word2*>(out + 10 * q);
#pragma unroll
  for (int j = 0; j < 5; j++) o[j] = word2{r.v[2 * j], r.v[2 * j + 1]};
}

// the chain over the sums; returns, in every lane of the quad, whether the result is the neutral element
template <int WINDOWS = HALF_WINDOWS>
ED_DEV bool verify_half_main_sums_quad(const uint32_t* hd, const uint32_t* sums, int q) {
  fe r;
  fe_set(r, (uint32_t)(q & 1));                  // neutral element (0, 1, 0, 1) as (X, Y, T, Z)
  const bool is_long = (hd[24] & 2u) != 0;
  const int top = (__any(is_long) ? HALF_LONG_WINDOWS : WINDOWS) - 1;
#pragma unroll 1
  for (int w = top; w >= 0; w--) {
    fe mult, first, m;
    quad_coord_load(mult, sums + w * HALF_SUM_WORDS + 10 * q);   // requested before the doublings, consumed after them
    if (w != top) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) quad_dbl(r, q);
    }
    quad_stage_a_operand(first, r, q);
    fe_mul(m, first, mult);
    quad_stage_b(r, m, q);
  }
  fe z, d;
  fe_quad_perm<3, 3, 3, 3>(z, r);
  fe_sub(d, r, z);
  const bool mine = q == 0 ? fe_iszero(r) : q == 1 ? fe_iszero(d) : q == 3 ? !fe_iszero(r) : true;
  const uint64_t votes = __ballot(mine);
  const unsigned lane = __lane_id();
  return ((votes >> (lane & ~3u)) & 0xfu) == 0xfu;
}

// ---------------------------------------------------------------------------------------------
// X25519 with four lanes per item, for SMALL passes (lanes.h: x25519_ladder_lane is the one-lane form):
// a ladder step is nine multiplications / squarings deep with one lane; with the quad holding
// (x2, z2, x3, z3) it is three deep:
//   stage 0   (x3 - z3)(x2 + z2) | (x3 + z3)(x2 - z2) | P^2 | Q^2      P, Q = sum and difference of the point to double
//   stage 1   (da + cb)^2        | (da - cb)^2        | aa bb | 121665 (aa - bb)
//   stage 2         -            | e (121665 e + aa)  |   -   | (da - cb)^2 x1
// Same field expressions as x25519.c:60-94 (montgomery) and the same swap logic as the one-lane form, so the
// same (x2 : z2); about 760 instructions per step instead of 1260.  Leaves x2 in lane 0 and z2 in
// lane 1 of the quad.
// ---------------------------------------------------------------------------------------------
ED_DEV void x25519_ladder_quad(fe& r, uint32_t s[8], const uint32_t pt[8], int q) {
  fe x1, k121665;
  clamp(s);
  fe_frombytes(x1, pt);                          // bit 255 folded in as +19, not masked (fld.c:153)
  fe_set(k121665, 121665);
  r = x1;                                        // (x2, z2, x3, z3) = (1, 0, x1, 1)
  { fe one, zero; fe_set(one, 1); fe_set(zero, 0); fe_cmov(r, one, q == 0 || q == 3); fe_cmov(r, zero, q == 1); }
  uint32_t cur = s[7] << 1;                      // bit 255 of the clamped scalar is 0: start at bit 254
  uint32_t swap = 0;
#pragma unroll 1
  for (int t = 254; t >= 0; t--) {
    const uint32_t bit = cur >> 31;
    cur <<= 1;
    if ((t & 31) == 0 && t != 0) {
      const int w = (t >> 5) - 1;
      cur = w == 6 ? s[6] : w == 5 ? s[5] : w == 4 ? s[4] : w == 3 ? s[3] : w == 2 ? s[2] : w == 1 ? s[1] : s[0];
    }
    swap ^= bit;
    // lanes 0, 1 hold the pair (x2, z2), lanes 2, 3 the pair (x3, z3): sum and difference of the lane's OWN pair, and the
    // other pair's through one exchange (every operand below is one of those four, picked by the lane's role)
    fe X, Z, S, D, So, Do, u, v, m1;
    fe_quad_perm<0, 0, 2, 2>(X, r); fe_quad_perm<1, 1, 3, 3>(Z, r);
    fe_add(S, X, Z);                             // 2u: lanes 0, 1: a = x2 + z2, lanes 2, 3: c = x3 + z3
    fe_sub(D, X, Z);                             // 3u: lanes 0, 1: b = x2 - z2, lanes 2, 3: d = x3 - z3
    fe_quad_perm<2, 3, 0, 1>(So, S); fe_quad_perm<2, 3, 0, 1>(Do, D);
    // stage 0: lane 0: d * a, lane 1: c * b, lane 2: P * P, lane 3: Q * Q  (P, Q: the point to double - (a, b), or (c, d) when exchanged)
    {
      fe ts = So, td = Do;
      fe_cmov(ts, S, swap != 0); fe_cmov(td, D, swap != 0);
      u = ts;
      fe_cmov(u, Do, q == 0); fe_cmov(u, So, q == 1); fe_cmov(u, td, q == 3);
      v = u;
      fe_cmov(v, S, q == 0); fe_cmov(v, D, q == 1);
    }
    swap = bit;
    fe_mul(m1, u, v);                            // (da, cb, aa, bb)
    // stage 1: lane 0: (da + cb)^2, lane 1: (da - cb)^2, lane 2: aa * bb, lane 3: e * 121665 - own value and the neighbour's
    fe pm, s1, d1, m2;
    fe_quad_perm<1, 0, 3, 2>(pm, m1);            // (cb, da, bb, aa)
    fe_add(s1, m1, pm);                          // 2u: lane 0: da + cb
    fe_sub(d1, pm, m1);                          // 3u: lane 1: da - cb, lane 3: e = aa - bb
    u = s1;
    fe_cmov(u, d1, (q & 1) != 0); fe_cmov(u, m1, q == 2);
    v = u;
    fe_cmov(v, pm, q == 2); fe_cmov(v, k121665, q == 3);
    fe_mul(m2, u, v);                            // (x3', (da - cb)^2, x2', 121665 e)
    // stage 2: lane 1: e * (121665 e + aa), lane 3: (da - cb)^2 * x1; lanes 0 and 2 idle.  Lane 3 has e and aa (its
    // neighbour's m1), lane 1 the square: one exchange of e, one of (sum | square)
    fe w3, ex, wx, m3;
    fe_add(w3, m2, pm);                          // lane 3: 121665 e + aa, 2u
    fe_quad_perm<0, 3, 2, 1>(ex, d1);            // lane 1: e
    { fe t = m2; fe_cmov(t, w3, q == 3); fe_quad_perm<0, 3, 2, 1>(wx, t); }   // lane 1: the sum, lane 3: the square
    u = ex; fe_cmov(u, wx, q == 3);
    v = wx; fe_cmov(v, x1, q == 3);
    fe_mul(m3, u, v);                            // lane 1: z2', lane 3: z3'
    // (x2', z2', x3', z3') = (m2 of lane 2, m3, m2 of lane 0, m3)
    fe_quad_perm<2, 1, 0, 3>(r, m2);
    fe_cmov(r, m3, (q & 1) != 0);
  }
}

}  // namespace ed
