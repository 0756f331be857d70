// host_check.cpp - TEST BINARY: the device source (libeddsa_amd/csrc/lanes.h and the headers it
// includes) compiled for the host CPU with -DED_HOST_CHECK, i.e. with an assertion on every limb
// precondition and on every 64-bit column sum.  Lets the CPU test suite run the exact algorithms
// of the kernels (windowed double-scalar multiplication, comb, ladder, digit recoding, Barrett,
// SHA-512) against the oracle without a GPU, and proves the bounds quoted in fe25519.h on the
// inputs it is driven with.  It is not part of the product and nothing in the product loads it.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "lanes.h"
#include "rlc_lanes.h"

namespace ed {
static std::atomic<long> g_violations{0};
static char g_first[256];
static std::mutex g_mu;
void bound_violation(const char* file, int line, const char* what) {
  if (g_violations.fetch_add(1) == 0) {
    std::lock_guard<std::mutex> lk(g_mu);
    snprintf(g_first, sizeof(g_first), "%s:%d: %s", file, line, what);
  }
}
}  // namespace ed

using namespace ed;

static void rd(uint32_t w[8], const uint8_t* p) { memcpy(w, p, 32); }      // little-endian host
static void wr(uint8_t* p, const uint32_t w[8]) { memcpy(p, w, 32); }

struct Tables {
  // 16-byte aligned view of the base table (entries are read with 16-byte loads)
  uint32_t* b16() { return reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(base16.data()) + 15) & ~(uintptr_t)15); }
  std::vector<uint32_t> base16, comb;
  Tables() : base16((size_t)2 * TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS + 32), comb(TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS) {
    for (int k = 0; k < TABLE_BASE16_ENTRIES; k++) table_entry_lane(b16() + (size_t)TABLE_ENTRY_WORDS * k, (uint32_t)k, 0);
    // the second half, k * 2^128 * B (verify_half_main_lane): each entry by doubling the one before the shift
    for (int k = 0; k < TABLE_BASE16_ENTRIES; k++)
      table_entry_lane(b16() + (size_t)TABLE_ENTRY_WORDS * (TABLE_BASE16_ENTRIES + k), (uint32_t)k, 128);
    for (int c = 0; c < TABLE_COMB_ENTRIES; c++)
      table_entry_lane(&comb[TABLE_ENTRY_WORDS * c], (uint32_t)(c % COMB_HALF) + 1, 2u * COMB_W * (uint32_t)(c / COMB_HALF));
  }
};
static Tables& tables() { static Tables t; return t; }

extern "C" {

long hc_violations(void) { return g_violations.load(); }
const char* hc_first_violation(void) { return g_first; }
void hc_reset(void) { g_violations = 0; g_first[0] = 0; }

void hc_tables(uint32_t* base16, uint32_t* comb) {
  memcpy(base16, tables().b16(), (size_t)TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS * 4);
  memcpy(comb, tables().comb.data(), tables().comb.size() * 4);
}

void hc_x25519(uint8_t out[32], const uint8_t scalar[32], const uint8_t point[32]) {
  uint32_t s[8], p[8], o[8];
  rd(s, scalar); rd(p, point);
  x25519_lane(o, s, p);
  wr(out, o);
}

// the three verify kernels for one item (the finish step without the cross-item batching, plus
// the same Z = 0 / off-curve rule)
int hc_verify(const uint8_t sig[64], const uint8_t pub[32], const uint8_t* msg, size_t len) {
  alignas(16) uint32_t tab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS];
  uint32_t rw[8], sw[8], aw[8], tw[8];
  rd(rw, sig); rd(sw, sig + 32); rd(aw, pub);
  const bool oncurve = verify_prepare_lane(tw, sw, tab, rw, aw, msg, len);
  ge acc;
  uint32_t digits[16];
  memcpy(digits, tw, 32); memcpy(digits + 8, sw, 32);
  verify_main_lane(acc, digits, tab, tables().b16());
  if (!oncurve || fe_iszero(acc.Z)) return 0;
  fe zinv;
  fe_inv(zinv, acc.Z);
  return verify_encode_lane(acc.X, acc.Y, zinv, rw) ? 1 : 0;
}

// the half-length path (halve.h; k_verify_prepare + k_verify_halve + k_verify_main_half) for one item:
// 0 / 1 = its verdict (+ 4 when no short pair was found and the item ran the long loop, + 8 when the search returned
// a pair that the exact integer check of verify_half_scalars_lane refused), 2 = the item is handed to the exact path
// (key off the curve)
int hc_verify_half(const uint8_t sig[64], const uint8_t pub[32], const uint8_t* msg, size_t len) {
  alignas(16) uint32_t tab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS], rtab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS];
  uint32_t rw[8], sw[8], aw[8], tw[8], hd[HALF_DIGIT_WORDS];
  rd(rw, sig); rd(sw, sig + 32); rd(aw, pub);
  const bool oncurve = verify_prepare_lane(tw, sw, tab, rw, aw, msg, len);
  verify_half_scalars_lane(hd, tw, sw);
  const bool rvalid = verify_half_point_lane(rtab, rw);
  if (!oncurve) return 2;
  const bool is_long = (hd[24] & 2u) != 0;
  // short items as k_verify_main_half evaluates them, long ones as their wave of k_verify_main_half_quad does
  const bool neutral = is_long ? verify_half_main_lane<true>(hd, tab, rtab, tables().b16(), true)
                               : verify_half_main_lane<false>(hd, tab, rtab, tables().b16(), false);
  return (neutral && rvalid ? 1 : 0) + (is_long ? 4 : 0) + ((hd[24] & 4u) ? 8 : 0);   // + 8: the pair failed the exact check
}

// fault injection into the pair search (halve.h: HALVE_FAULT): the n-th working half-step takes a quotient one too large
void hc_halve_fault(int n) { halve_fault = n; }

// the wide form (pairs up to 2^138, 35 windows) that one-lane passes below 2^18 items use
int hc_verify_half_wide(const uint8_t sig[64], const uint8_t pub[32], const uint8_t* msg, size_t len) {
  alignas(16) uint32_t tab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS], rtab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS];
  uint32_t rw[8], sw[8], aw[8], tw[8], hd[HALF_DIGIT_WORDS];
  rd(rw, sig); rd(sw, sig + 32); rd(aw, pub);
  const bool oncurve = verify_prepare_lane(tw, sw, tab, rw, aw, msg, len);
  verify_half_scalars_lane<HALF_BITS_SMALL>(hd, tw, sw);
  const bool rvalid = verify_half_point_lane(rtab, rw);
  if (!oncurve || (hd[24] & 2u) != 0) return 2;   // the exact path
  return verify_half_main_lane<false, HALF_WINDOWS_SMALL>(hd, tab, rtab, tables().b16(), false) && rvalid ? 1 : 0;
}

// the same with the long loop forced, as a short item runs it in a wave that contains a long one
int hc_verify_half_in_long_wave(const uint8_t sig[64], const uint8_t pub[32], const uint8_t* msg, size_t len) {
  alignas(16) uint32_t tab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS], rtab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS];
  uint32_t rw[8], sw[8], aw[8], tw[8], hd[HALF_DIGIT_WORDS];
  rd(rw, sig); rd(sw, sig + 32); rd(aw, pub);
  const bool oncurve = verify_prepare_lane(tw, sw, tab, rw, aw, msg, len);
  verify_half_scalars_lane(hd, tw, sw);
  const bool rvalid = verify_half_point_lane(rtab, rw);
  if (!oncurve) return 2;
  return verify_half_main_lane<true>(hd, tab, rtab, tables().b16(), true) && rvalid ? 1 : 0;
}

// halve_scalar_lane: v (20 bytes), |u| (20 bytes), sign of u; returns whether a pair was found
int hc_halve(uint8_t v[20], uint8_t u[20], int* uneg, const uint8_t t[32], int wide) {
  uint32_t tw[8], vw[5], uw[5];
  rd(tw, t);
  bool ng;
  const bool good = wide ? halve_scalar_lane<HALF_BITS_SMALL>(vw, uw, ng, tw) : halve_scalar_lane<HALF_BITS>(vw, uw, ng, tw);
  memcpy(v, vw, 20); memcpy(u, uw, 20);
  *uneg = ng ? 1 : 0;
  return good ? 1 : 0;
}

void hc_halve_counters(long out[2], int reset) {
  out[0] = halve_counters[0]; out[1] = halve_counters[1];
  if (reset) halve_counters[0] = halve_counters[1] = 0;
}

// the exact (reference-order) path, as k_verify_exact runs it for off-curve keys
int hc_verify_exact(const uint8_t sig[64], const uint8_t pub[32], const uint8_t* msg, size_t len) {
  uint32_t rw[8], sw[8], aw[8];
  rd(rw, sig); rd(sw, sig + 32); rd(aw, pub);
  int8_t ux[REF_JSF_LEN], uy[REF_JSF_LEN];
  uint32_t pts[160];
  return verify_exact_lane(rw, sw, aw, msg, len, tables().b16() + TABLE_ENTRY_WORDS, ux, uy, pts, 1) ? 1 : 0;
}

// the exact path as the one-lane throughput kernels run it (k_verify_exact_lane_setup + _chain): set-up from what the prepare
// step left in the item's workspace (digit words, entry 1 of its table), Q + B and Q - B into entries 2 and 3, the chain
// over packed cached entries
int hc_verify_exact_table(const uint8_t sig[64], const uint8_t pub[32], const uint8_t* msg, size_t len) {
  alignas(16) uint32_t tab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS], bentry[VERIFY_ENTRY_WORDS];
  uint32_t rw[8], sw[8], aw[8], tw[8], digits[16], dig[EXACT_DIGIT_WORDS];
  rd(rw, sig); rd(sw, sig + 32); rd(aw, pub);
  verify_prepare_lane(tw, sw, tab, rw, aw, msg, len);
  memcpy(digits, tw, 32); memcpy(digits + 8, sw, 32);
  exact_bentry_store(bentry);
  verify_exact_setup_table_lane(tab, dig, 1, digits, tables().b16() + TABLE_ENTRY_WORDS);
  const bool whole = verify_exact_chain_table_lane(rw, tab, bentry, dig, 1);
  // ... and stretch by stretch with the accumulator handed on through memory, as k_verify_exact_lane_chain runs it
  alignas(16) uint32_t state[EXACT_STATE_WORDS];
  bool staged = false;
  for (int seg = 0; seg < EXACT_SEGS; seg++) staged = verify_exact_chain_segment_lane(seg, rw, tab, bentry, dig, 1, state, true);
  if (staged != whole) return 2;
  return whole ? 1 : 0;
}

// ed_dual_scale (s B + t Q, ed.c:455-507) on an arbitrary 32-byte "point" through the table form of the chain: what the
// one-lane kernels compute, as bytes - for a Q that is no curve point the bytes depend on the digit string and on the order
// of the formulas, which a verdict hardly ever shows
void hc_dual_scale_exact_table(uint8_t out[32], const uint8_t s[32], const uint8_t t[32], const uint8_t q[32], int staged) {
  alignas(16) uint32_t tab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS], bentry[VERIFY_ENTRY_WORDS], state[EXACT_STATE_WORDS];
  uint32_t w[8], digits[16], dig[EXACT_DIGIT_WORDS], o[8];
  sc x, y;
  rd(w, s); sc_from_words<8>(x, w); sc_to_words(digits + 8, x);
  rd(w, t); sc_from_words<8>(y, w); sc_to_words(digits, y);
  words_add_pattern(digits, 0x88888888u);        // as k_verify_prepare leaves them
  words_add_pattern(digits + 8, 0x80008000u);
  rd(w, q);
  ge Q, R; bool oc;
  ge_frombytes(Q, oc, w, false);
  ge_cached c;
  ge_to_cached(c, Q);
  cached_store(tab, 1, c);
  exact_bentry_store(bentry);
  verify_exact_setup_table_lane(tab, dig, 1, digits, tables().b16() + TABLE_ENTRY_WORDS);
  if (staged) {
    for (int seg = 0; seg < EXACT_SEGS; seg++) {
      if (seg == 0) ge_neutral(R); else exact_state_load(R, state);
      exact_chain_steps(R, tab, bentry, dig, 1, exact_seg_hi(seg), exact_seg_lo(seg));
      exact_state_store(state, R);
    }
  } else {
    ge_neutral(R);
    exact_chain_steps(R, tab, bentry, dig, 1, REF_JSF_LEN - 1, 0);
  }
  ge_tobytes(o, R); wr(out, o);
}

// two of those side by side in one lane, as k_verify_exact_lane_chain walks them (lanes.h: exact_pair_iterations): units of
// `unit` iterations with the accumulators and walks handed on through memory (unit <= 0: one go), the last one to the end
static void table_setup(uint32_t* tab, uint32_t* dig, const uint8_t s[32], const uint8_t t[32], const uint8_t q[32]) {
  uint32_t w[8], digits[16];
  sc x, y;
  rd(w, s); sc_from_words<8>(x, w); sc_to_words(digits + 8, x);
  rd(w, t); sc_from_words<8>(y, w); sc_to_words(digits, y);
  words_add_pattern(digits, 0x88888888u);
  words_add_pattern(digits + 8, 0x80008000u);
  rd(w, q);
  ge Q; bool oc;
  ge_frombytes(Q, oc, w, false);
  ge_cached c;
  ge_to_cached(c, Q);
  cached_store(tab, 1, c);
  verify_exact_setup_table_lane(tab, dig, 1, digits, tables().b16() + TABLE_ENTRY_WORDS);
}
int hc_dual_scale_exact_pair(uint8_t out_a[32], uint8_t out_b[32], const uint8_t a96[96], const uint8_t b96[96], int have_b, int unit) {
  constexpr uint32_t SLOT = VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS;       // two workspace slots each, as the kernel's table / rtable
  alignas(16) uint32_t tabs[2 * SLOT], digs[2 * SLOT], bentry[VERIFY_ENTRY_WORDS];
  alignas(16) uint32_t st_a[EXACT_STATE_WORDS + 4], st_b[EXACT_STATE_WORDS + 4];
  uint32_t *tab_a = tabs, *tab_b = tabs + SLOT, *dig_a = digs, *dig_b = digs + SLOT, o[8];
  exact_bentry_store(bentry);
  table_setup(tab_a, dig_a, a96, a96 + 32, a96 + 64);
  table_setup(tab_b, dig_b, b96, b96 + 32, b96 + 64);
  ge ra, rb;
  ge_neutral(ra); ge_neutral(rb);
  exact_walk wa = exact_walk_start(dig_a, 1, true), wb = exact_walk_start(dig_b, 1, have_b != 0);
  int units = 0;
  if (unit <= 0) {
    exact_pair_iterations(ra, wa, rb, wb, tabs, digs, SLOT, 0, 1, bentry, -1);
  } else {
    for (int seg = 0; seg < EXACT_SEGS; seg++, units++) {
      if (seg) { exact_walk_load(ra, wa, st_a); exact_walk_load(rb, wb, st_b); }
      exact_pair_iterations(ra, wa, rb, wb, tabs, digs, SLOT, 0, 1, bentry, seg == EXACT_SEGS - 1 ? -1 : unit);
      exact_walk_store(st_a, ra, wa); exact_walk_store(st_b, rb, wb);
    }
  }
  if (wa.i >= 0 || wb.i >= 0) return -1;
  // the encodings, through the shared inversion: compare with all-zero R and read the bytes off separately
  ge_tobytes(o, ra); wr(out_a, o);
  ge_tobytes(o, rb); wr(out_b, o);
  uint32_t rwa[8], rwb[8];
  rd(rwa, out_a); rd(rwb, out_b);
  bool sa, sb;
  exact_pair_verdicts(sa, sb, ra, rb, rwa, rwb);
  if (!sa || (have_b && !sb)) return -2;             // the shared-inversion export must reproduce ge_tobytes' bytes
  rwa[0] ^= 1u;
  exact_pair_verdicts(sa, sb, ra, rb, rwa, rwb);
  if (sa) return -3;
  return 0;
}

// ed_dual_scale in the reference's order on an arbitrary 32-byte "point" (cf. orc_ed_dual_scale)
void hc_dual_scale_exact(uint8_t out[32], const uint8_t s[32], const uint8_t t[32], const uint8_t q[32], int uniform) {
  uint32_t w[8], sw[8], tw[8], o[8];
  sc x, y;
  rd(w, s); sc_from_words<8>(x, w); sc_to_words(sw, x);
  rd(w, t); sc_from_words<8>(y, w); sc_to_words(tw, y);
  rd(w, q);
  ge Q, R; bool oc;
  ge_frombytes(Q, oc, w, false);
  ge_niels pcB;
  niels_load(pcB, tables().b16() + TABLE_ENTRY_WORDS);
  int8_t ux[REF_JSF_LEN], uy[REF_JSF_LEN];
  uint32_t pts[160];
  if (uniform) ref_dual_scale_uniform(R, sw, tw, Q, pcB, ux, uy, pts, 1); else ref_dual_scale(R, sw, tw, Q, pcB);
  ge_tobytes(o, R); wr(out, o);
}

// Montgomery's trick exactly as k_verify_finish applies it, on k <= 8 values: out[j] = 1/z[j]
void hc_batch_inverse(uint8_t* out, const uint8_t* zs, int k) {
  fe z[8], p[8], u, zi;
  uint32_t w[8];
  for (int j = 0; j < k; j++) { rd(w, zs + 32 * j); fe_frombytes(z[j], w); }
  p[0] = z[0];
  for (int j = 1; j < k; j++) fe_mul(p[j], p[j - 1], z[j]);
  fe_inv(u, p[k - 1]);
  for (int j = k - 1; j >= 1; j--) {
    fe_mul(zi, u, p[j - 1]);
    fe_tobytes(w, zi); wr(out + 32 * j, w);
    fe_mul(u, u, z[j]);
  }
  fe_tobytes(w, u); wr(out, w);
}

// point step, inversion, finish step: what the k_*_point / k_*_finish kernel pairs do for one item
void hc_genpub(uint8_t pub[32], const uint8_t sec[32]) {
  uint32_t sk[8], o[8];
  rd(sk, sec);
  ge A; fe zinv;
  genpub_point_lane(A, sk, tables().comb.data());
  fe_inv(zinv, A.Z);
  encode_lane(o, A.X, A.Y, zinv);
  wr(pub, o);
}

void hc_sign(uint8_t sig[64], const uint8_t sec[32], const uint8_t pub[32], const uint8_t* msg, size_t len) {
  uint32_t sk[8], pk[8], R[8], S[8], aw[8], rw[8];
  rd(sk, sec); rd(pk, pub);
  ge Rp; fe zinv;
  sign_point_lane(Rp, aw, rw, sk, msg, len, tables().comb.data());
  fe_inv(zinv, Rp.Z);
  encode_lane(R, Rp.X, Rp.Y, zinv);
  sign_finish_lane(S, R, aw, rw, pk, msg, len);
  wr(sig, R); wr(sig + 32, S);
}

void hc_x25519_base(uint8_t out[32], const uint8_t scalar[32]) {
  uint32_t s[8], o[8];
  rd(s, scalar);
  ge R; fe d;
  x25519_base_point_lane(R, s, tables().comb.data());
  fe_sub(d, R.Z, R.Y);
  fe_inv(d, d);                                   // 0 -> 0
  x25519_base_finish_lane(o, R.Y, R.Z, d);
  wr(out, o);
}

void hc_pk_to_x(uint8_t out[32], const uint8_t in[32]) {
  uint32_t w[8], o[8];
  rd(w, in);
  pk_to_x_lane(o, w);
  wr(out, o);
}

void hc_sk_to_x(uint8_t out[32], const uint8_t in[32]) {
  uint32_t w[8], o[8];
  rd(w, in);
  sk_to_x_lane(o, w);
  wr(out, o);
}

// ---- layer probes -------------------------------------------------------------------------

void hc_fe_mul(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) {
  uint32_t w[8]; fe x, y;
  rd(w, a); fe_frombytes(x, w); rd(w, b); fe_frombytes(y, w);
  fe_mul(x, x, y);
  fe_tobytes(w, x); wr(out, w);
}
void hc_fe_sq(uint8_t out[32], const uint8_t a[32]) {
  uint32_t w[8]; fe x;
  rd(w, a); fe_frombytes(x, w); fe_sq(x, x); fe_tobytes(w, x); wr(out, w);
}
void hc_fe_inv(uint8_t out[32], const uint8_t a[32]) {
  uint32_t w[8]; fe x;
  rd(w, a); fe_frombytes(x, w); fe_inv(x, x); fe_tobytes(w, x); wr(out, w);
}
void hc_fe_pow2523(uint8_t out[32], const uint8_t a[32]) {
  uint32_t w[8]; fe x;
  rd(w, a); fe_frombytes(x, w); fe_pow2523(x, x); fe_tobytes(w, x); wr(out, w);
}
// multiplication at the documented operand limits: f = ka * a (limb-wise, ka <= 7), g = kb * b (kb <= 3)
void hc_fe_mul_loose(uint8_t out[32], const uint8_t a[32], int ka, const uint8_t b[32], int kb) {
  uint32_t w[8]; fe x, y, fx, gy;
  rd(w, a); fe_frombytes(x, w); rd(w, b); fe_frombytes(y, w);
  fe_set(fx, 0); fe_set(gy, 0);
  for (int i = 0; i < ka; i++) fe_add(fx, fx, x);
  for (int i = 0; i < kb; i++) fe_add(gy, gy, y);
  fe_mul(x, fx, gy);
  fe_tobytes(w, x); wr(out, w);
}
void hc_sc_reduce(uint8_t out[32], const uint8_t* in, size_t len) {      // len = 32 or 64
  uint32_t w[16], o[8]; sc x;
  memcpy(w, in, len);
  if (len == 32) sc_from_words<8>(x, w); else sc_from_words<16>(x, w);
  sc_to_words(o, x); wr(out, o);
}
void hc_sc_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]) {
  uint32_t w[8], o[8]; sc x, y, z;
  rd(w, a); sc_from_words<8>(x, w); rd(w, b); sc_from_words<8>(y, w); rd(w, c); sc_from_words<8>(z, w);
  sc_mul(x, x, y); sc_add(x, z, x);
  sc_to_words(o, x); wr(out, o);
}
void hc_sha512(uint8_t out[64], const uint8_t* msg, size_t len) {
  uint32_t d[16];
  sha512_prefix_msg<0>(d, nullptr, msg, len);
  memcpy(out, d, 64);
}
int hc_ed_import_export(uint8_t out[32], const uint8_t in[32]) {         // returns the on-curve flag
  uint32_t w[8], o[8]; ge p; bool oc;
  rd(w, in);
  ge_frombytes(p, oc, w, false);
  ge_tobytes(o, p); wr(out, o);
  return oc ? 1 : 0;
}
void hc_scale_base(uint8_t out[32], const uint8_t scalar[32]) {           // scalar reduced mod l first
  uint32_t w[8], xw[8], o[8]; sc x; ge p;
  rd(w, scalar); sc_from_words<8>(x, w); sc_to_words(xw, x);
  scale_base_lane(p, xw, tables().comb.data());
  ge_tobytes(o, p); wr(out, o);
}


// ---- batch verification (rlc_lanes.h): the per-lane steps of rlc.hip, and the whole combination of a small
// batch evaluated the plain way (double-and-add over the signed byte digits) instead of by the workgroup's buckets

// flags (RLC_R_VALID = 1, RLC_PER_ITEM = 2) of a key (is_r = 0) or an R (is_r = 1); out = the affine niels
// entry of the NEGATED point as 3 x 32 canonical bytes (y-x | y+x | 2dxy)
int hc_rlc_decode(uint8_t out[96], const uint8_t in[32], int is_r) {
  uint32_t w[8];
  rd(w, in);
  ge_niels nl;
  const uint8_t fl = is_r ? rlc_decode_r_lane(nl, w) : rlc_decode_key_lane(nl, w);
  uint32_t o[8];
  fe_tobytes(o, nl.ymx); wr(out, o);
  fe_tobytes(o, nl.ypx); wr(out + 32, o);
  fe_tobytes(o, nl.t2d); wr(out + 64, o);
  return fl;
}

// digits of item i under `seed`: dig_a[32], dig_r[16], zs = z S mod l (32 bytes); t, s = reduced scalars
void hc_rlc_scalars(int8_t dig_a[32], int8_t dig_r[16], uint8_t zs[32], const uint8_t seed[32], uint64_t i,
                    const uint8_t t[32], const uint8_t s[32]) {
  uint32_t sd[8], tw[8], sw[8], z9[9];
  rd(sd, seed); rd(tw, t); rd(sw, s);
  rlc_scalars_lane(dig_a, dig_r, z9, sd, i, tw, sw);
  memcpy(zs, z9, 32);
}

// The whole check of rlc.hip for ONE group of n <= 64 items with fixed-length messages, every step taken from
// rlc_lanes.h: leaves -> seed (the hash tree), points and flags, coefficients and digits, then
//   sum_w 256^w ( sum_i digA[i][w] (-A_i) + sum_i digR[i][w] (-R_i) + digB[w] B )
// by Horner with plain additions.  Returns 1 if the total is the neutral element and no item is flagged for the
// per-item path, 0 otherwise; valid_r[i] receives the RLC_R_VALID flag of item i.
int hc_rlc_group(uint8_t* valid_r, const uint8_t* sigs, const uint8_t* pubs, const uint8_t* msgs, size_t mlen, int n) {
  if (n < 1 || n > 64) return -1;
  std::vector<uint32_t> tw(8 * n), sw(8 * n), leaves(8 * n);
  std::vector<ge_niels> na(n), nr(n);
  std::vector<uint8_t> fl(n);
  for (int i = 0; i < n; i++) {
    uint32_t rw[8], aw[8];
    rd(rw, sigs + 64 * i); rd(aw, pubs + 32 * i); rd(&sw[8 * i], sigs + 64 * i + 32);
    rlc_hash_lane(&tw[8 * i], &sw[8 * i], &leaves[8 * i], rw, aw, sigs + 64 * i + 32, msgs + mlen * i, mlen);
    fl[i] = rlc_decode_key_lane(na[i], aw) | rlc_decode_r_lane(nr[i], rw);
    valid_r[i] = fl[i] & RLC_R_VALID;
  }
  uint32_t seed16[16];
  {                                              // rlc.hip: k_rlc_tree, level by level until one node is left
    std::vector<uint32_t> level = leaves;
    size_t count = (size_t)n;
    do {
      const size_t next = (count + RLC_TREE_FAN - 1) / RLC_TREE_FAN;
      std::vector<uint32_t> up(8 * next);
      for (size_t j = 0; j < next; j++) {
        const size_t lo = j * RLC_TREE_FAN, cnt = count - lo < (size_t)RLC_TREE_FAN ? count - lo : (size_t)RLC_TREE_FAN;
        sha512_prefix_msg<0>(seed16, nullptr, reinterpret_cast<const uint8_t*>(level.data() + 8 * lo), 32 * cnt);
        for (int k = 0; k < 8; k++) up[8 * j + k] = seed16[k];
      }
      level = up;
      count = next;
    } while (count > 1);
    for (int k = 0; k < 8; k++) seed16[k] = level[k];
  }
  std::vector<int8_t> da(32 * n), dr(16 * n);
  uint32_t sum[16] = {0};
  bool flagged = false;
  for (int i = 0; i < n; i++) {
    flagged = flagged || (fl[i] & RLC_PER_ITEM);
    if (!(fl[i] & RLC_R_VALID)) { memset(&da[32 * i], 0, 32); memset(&dr[16 * i], 0, 16); continue; }
    uint32_t zs[9];
    rlc_scalars_lane(&da[32 * i], &dr[16 * i], zs, seed16, (uint64_t)i, &tw[8 * i], &sw[8 * i]);
    uint64_t c = 0;
    for (int k = 0; k < 10; k++) { c += (uint64_t)sum[k] + (k < 9 ? zs[k] : 0u); sum[k] = (uint32_t)c; c >>= 32; }
  }
  int8_t db[32];
  rlc_group_scalar_lane(db, sum);
  ge_niels nb;
  niels_load(nb, tables().b16() + TABLE_ENTRY_WORDS);        // B
  ge acc;
  ge_neutral(acc);
  auto add_digit = [&](const ge_niels& pt, int d) {          // acc += d * pt, |d| <= 128
    if (d == 0) return;
    ge_niels q = pt;
    ge_niels_cneg(q, d < 0);
    ge m;                                                    // m = |d| * pt by double-and-add
    ge_neutral(m);
    const int mag = d < 0 ? -d : d;
    for (int bit = 7; bit >= 0; bit--) {
      ge_dbl(m, m, true);
      if ((mag >> bit) & 1) ge_add_niels(m, m, q, true);
    }
    ge_cached c;
    ge_to_cached(c, m);
    ge_add_cached(acc, acc, c, true);
  };
  for (int w = 31; w >= 0; w--) {
    for (int k = 0; k < 8; k++) ge_dbl(acc, acc, true);
    for (int i = 0; i < n; i++) {
      add_digit(na[i], da[32 * i + w]);
      if (w < 16) add_digit(nr[i], dr[16 * i + w]);
    }
    add_digit(nb, db[w]);
  }
  return (ge_is_neutral(acc) && !flagged) ? 1 : 0;
}

}  // extern "C"
