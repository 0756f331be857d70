// kernels.hip - the batched Ed25519 / X25519 kernels for gfx950 and their launchers.
//
// One curve operation per lane, limbs in registers (fe25519.h), no MFMA.  Inputs and outputs are
// the packed, item-major byte arrays of the batch API (include/eddsa_amd.h).
//
//   k_x25519        x25519.c:129-150 do_x25519                      (config 3)
//   k_verify_*      ed25519-sha512.c:148-181 ed25519_verify        (config 2, 4)
//   k_sign          ed25519-sha512.c:84-123 sign                    (config 5)
//   k_genpub        ed25519-sha512.c:53-67 genpub
//   k_x25519_base   x25519.c:158-197 do_x25519_base
//   k_pk_to_x       ed25519-sha512.c:187-232 pk_ed25519_to_x25519
//   k_sk_to_x       ed25519-sha512.c:239-256 sk_ed25519_to_x25519
//   k_init_tables   generates what the reference ships as lib/ed_lookup64.h
#include "eddsa_kernels.h"

#include "fe25519.h"
#include "ge25519.h"
#include "sc25519.h"
#include "sha512.h"

namespace ed {

constexpr int BLOCK = 256;

// ---- packed byte-array access: 32 bytes per item as eight little-endian words ---------------

ED_DEV void load32(uint32_t w[8], const uint8_t* base, size_t item, size_t stride) {
  const uint8_t* p = base + item * stride;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0];
    const uint4 b = reinterpret_cast<const uint4*>(p)[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++)
      w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
             ((uint32_t)p[4 * i + 3] << 24);
  }
}

ED_DEV void store32(uint8_t* base, size_t item, size_t stride, const uint32_t w[8]) {
  uint8_t* p = base + item * stride;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    reinterpret_cast<uint4*>(p)[0] = make_uint4(w[0], w[1], w[2], w[3]);
    reinterpret_cast<uint4*>(p)[1] = make_uint4(w[4], w[5], w[6], w[7]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      p[4 * i] = (uint8_t)w[i]; p[4 * i + 1] = (uint8_t)(w[i] >> 8);
      p[4 * i + 2] = (uint8_t)(w[i] >> 16); p[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
  }
}

// 256-bit little-endian value << s (s = 1, 4 or 8): the scalar is consumed from the top
template <int S>
ED_DEV void shl256(uint32_t w[8]) {
#pragma unroll
  for (int i = 7; i > 0; i--) w[i] = (w[i] << S) | (w[i - 1] >> (32 - S));
  w[0] <<= S;
}

// x25519.c:137-140: clamp copy
ED_DEV void clamp(uint32_t s[8]) {
  s[0] &= 0xfffffff8u;
  s[7] = (s[7] & 0x7fffffffu) | 0x40000000u;
}

// ---------------------------------------------------------------------------------------------
// X25519: x25519.c:60-150 (montgomery, mg_scale, do_x25519)
// ---------------------------------------------------------------------------------------------

ED_DEV void x25519_lane(uint32_t out[8], uint32_t s[8], const uint32_t pt[8]) {
  fe x1, x2, z2, x3, z3;
  clamp(s);
  fe_frombytes(x1, pt);                          // bit 255 folded in as +19, not masked (fld.c:153)
  fe_set(x2, 1); fe_set(z2, 0); x3 = x1; fe_set(z3, 1);
  // bit 255 of the clamped scalar is 0 and the step for it maps (1:0),(x1:1) to itself
  // projectively, so the ladder starts at bit 254.
  shl256<1>(s);
  uint32_t swap = 0;
#pragma unroll 1
  for (int t = 254; t >= 0; t--) {
    const uint32_t bit = s[7] >> 31;
    shl256<1>(s);
    swap ^= bit;
    fe_cswap(x2, x3, swap != 0);
    fe_cswap(z2, z3, swap != 0);
    swap = bit;
    fe a, aa, b, bb, e, c, d, da, cb, t1;
    fe_add(a, x2, z2);                           // 2u
    fe_sq(aa, a);
    fe_sub(b, x2, z2);                           // 3u
    fe_sq(bb, b);
    fe_sub(e, aa, bb);                           // 3u
    fe_add(c, x3, z3);                           // 2u
    fe_sub(d, x3, z3);                           // 3u
    fe_mul(da, d, a);
    fe_mul(cb, c, b);
    fe_add(t1, da, cb);                          // 2u
    fe_sq(x3, t1);
    fe_sub(t1, da, cb);                          // 3u
    fe_sq(t1, t1);
    fe_mul(z3, t1, x1);
    fe_mul(x2, aa, bb);
    fe_mul121665(t1, e);                         // x25519.c:78 fld_scale(T2, T1, 121665)
    fe_add(t1, t1, aa);                          // 2u
    fe_mul(z2, e, t1);
  }
  fe_cswap(x2, x3, swap != 0);
  fe_cswap(z2, z3, swap != 0);
  fe_inv(z2, z2);                                // z = 0 -> 0 (x25519.c:145)
  fe_mul(x2, x2, z2);
  fe_tobytes(out, x2);
}

__global__ void __launch_bounds__(BLOCK, 2)
k_x25519(uint8_t* out, const uint8_t* scalars, const uint8_t* points, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8], pt[8], r[8];
  load32(s, scalars, i, 32);
  load32(pt, points, i, 32);
  x25519_lane(r, s, pt);
  store32(out, i, 32, r);
}

// ---------------------------------------------------------------------------------------------
// base-point tables (what the reference ships as generated data, lib/ed_lookup64.h)
// ---------------------------------------------------------------------------------------------
// Entry = 32 words: y-x | y+x | 2dxy (10 canonical limbs each) + 2 words of padding.
//   base8[k],  k = 0..128 : k * B                  (8-bit signed windows of S in verify)
//   comb[i][k], i < 32, k < 8 : (k+1) * 256^i * B  (ed.c:41-43 ed_lookup, sign/genpub/x25519_base)

ED_DEV void niels_store(uint32_t* dst, const ge_niels& n) {
#pragma unroll
  for (int j = 0; j < 10; j++) { dst[j] = n.ymx.v[j]; dst[10 + j] = n.ypx.v[j]; dst[20 + j] = n.t2d.v[j]; }
  dst[30] = 0; dst[31] = 0;
}

template <typename P>
ED_DEV void niels_load(ge_niels& n, const P* src) {
#pragma unroll
  for (int j = 0; j < 10; j++) { n.ymx.v[j] = src[j]; n.ypx.v[j] = src[10 + j]; n.t2d.v[j] = src[20 + j]; }
}

__global__ void __launch_bounds__(64) k_init_tables(uint32_t* base8, uint32_t* comb) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  if (id >= TABLE_BASE8_ENTRIES + 256) return;
  uint32_t mult, shift;                          // the entry is mult * 2^shift * B
  uint32_t* dst;
  if (id < TABLE_BASE8_ENTRIES) { mult = id; shift = 0; dst = base8 + 32 * id; }
  else { const int c = id - TABLE_BASE8_ENTRIES; mult = (c & 7) + 1; shift = 8 * (c >> 3); dst = comb + 32 * c; }
  ge b, acc;
  ge_cached bc;
  ge_base(b);
  ge_to_cached(bc, b);
  ge_neutral(acc);
  for (int bit = 7; bit >= 0; bit--) {           // acc = mult * B
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
//     acc = 16*acc + d_i*(-A)  [+ e_j*B when i = 2j]
// with d_i in [-8,7] (ed.c:407-409's x + 0x88..8 recoding) looked up in a per-item table of
// 0..8 times -A kept in HBM, and e_j in [-128,127] looked up in a 129-entry table of multiples
// of B staged in LDS.  Control flow is uniform across the wave; the reference's 9-way
// data-dependent branch (ed.c:480-501) would serialise all 64 lanes.
// Result equality with the reference: DESIGN.md "Why the windowed evaluation is bit-exact".
//
// Three kernels per chunk, so that each stays inside its register budget and its own I-cache
// footprint (one fused kernel spilled 4.5 KB/lane):
//   k_verify_prepare  hash, scalars -> digit words, decompress -A, table of 0..8 * -A
//   k_verify_main     the 252 doublings + 96 additions            (~85 % of the time)
//   k_verify_finish   invert Z, encode, compare with R
// Workspace (HBM; tile = 256 items):
//   digits [item][16]                 t + 0x88.., S + 0x80.. as little-endian words
//   table  [item][entry 9][word 40]         1440 contiguous bytes per item
//   acc    [tile][word 30][lane 256]   X, Y, Z of the result
//   flags  [item]                      1 = A decoded to a curve point

// table entry = 40 words (ymx | ypx | t2d | z2), contiguous per item so that one lookup reads
// 160 contiguous bytes (ten 16-byte loads) instead of touching one 128-byte line per word.
ED_DEV void cached_store(uint32_t* tab, int entry, const ge_cached& c) {
  uint4* p = reinterpret_cast<uint4*>(tab + entry * 40);
  const fe* f[4] = {&c.ymx, &c.ypx, &c.t2d, &c.z2};
  uint32_t w[40];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int j = 0; j < 10; j++) w[10 * k + j] = f[k]->v[j];
#pragma unroll
  for (int q = 0; q < 10; q++) p[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
ED_DEV void cached_load(ge_cached& c, const uint32_t* tab, uint32_t entry) {
  const uint4* p = reinterpret_cast<const uint4*>(tab + entry * 40);
  uint32_t w[40];
#pragma unroll
  for (int q = 0; q < 10; q++) {
    const uint4 v = p[q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
#pragma unroll
  for (int j = 0; j < 10; j++) {
    c.ymx.v[j] = w[j]; c.ypx.v[j] = w[10 + j]; c.t2d.v[j] = w[20 + j]; c.z2.v[j] = w[30 + j];
  }
}

__global__ void __launch_bounds__(BLOCK, 2)
k_verify_prepare(const uint8_t* sigs, const uint8_t* pubs, const uint8_t* msgs,
                 const uint64_t* msg_off, size_t msg_len, size_t n, uint32_t* digits,
                 uint32_t* table, uint8_t* flags) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;         // idle lanes redo the last item (same stores)
  uint32_t rw[8], aw[8], sw[8], tw[8];
  load32(rw, sigs, item, 64);
  load32(sw, sigs + 32, item, 64);
  load32(aw, pubs, item, 32);

  // t = SHA-512(R || A || M) mod l ; S mod l (not range-checked: sc.c:191-214)
  {
    uint32_t pre[16], dig[16];
#pragma unroll
    for (int k = 0; k < 8; k++) { pre[k] = rw[k]; pre[8 + k] = aw[k]; }
    const uint8_t* m; size_t mlen;
    if (msg_off) { m = msgs + msg_off[item]; mlen = (size_t)(msg_off[item + 1] - msg_off[item]); }
    else { m = msgs + item * msg_len; mlen = msg_len; }
    sha512_prefix_msg<16>(dig, pre, m, mlen);
    sc t, s;
    sc_from_words<16>(t, dig);
    sc_from_words<8>(s, sw);
    sc_to_words(tw, t);
    sc_to_words(sw, s);
    words_add_pattern(tw, 0x88888888u);          // nibble - 8 is the signed digit
    words_add_pattern(sw, 0x80808080u);          // byte - 128 is the signed digit
    uint4* d = reinterpret_cast<uint4*>(digits + 16 * i);
    if (i < n) {
      d[0] = make_uint4(tw[0], tw[1], tw[2], tw[3]); d[1] = make_uint4(tw[4], tw[5], tw[6], tw[7]);
      d[2] = make_uint4(sw[0], sw[1], sw[2], sw[3]); d[3] = make_uint4(sw[4], sw[5], sw[6], sw[7]);
    }
  }

  // -A and its multiples 0..8, cached form
  uint32_t* tab = table + i * (VERIFY_TABLE_ENTRIES * 40);
  bool oncurve;
  ge a, p, q;
  ge_cached c1, c;
  ge_frombytes(a, oncurve, aw, true);
  if (i < n) flags[i] = (uint8_t)oncurve;
  ge_neutral(p);
  ge_to_cached(c, p);  cached_store(tab, 0, c);
  ge_to_cached(c1, a); cached_store(tab, 1, c1);
  ge_dbl(p, a, true);                            // 2
  ge_to_cached(c, p);  cached_store(tab, 2, c);
  ge_add_cached(q, p, c1, true);                 // 3
  ge_to_cached(c, q);  cached_store(tab, 3, c);
  ge_dbl(p, p, true);                            // 4
  ge_to_cached(c, p);  cached_store(tab, 4, c);
  ge_dbl(q, q, true);                            // 6
  ge_to_cached(c, q);  cached_store(tab, 6, c);
  ge_add_cached(q, q, c1, true);                 // 7
  ge_to_cached(c, q);  cached_store(tab, 7, c);
  ge_add_cached(q, p, c1, true);                 // 5
  ge_to_cached(c, q);  cached_store(tab, 5, c);
  ge_dbl(p, p, true);                            // 8
  ge_to_cached(c, p);  cached_store(tab, 8, c);
}

__global__ void __launch_bounds__(BLOCK, 2)
k_verify_main(const uint32_t* digits, const uint32_t* table, const uint32_t* base8, uint32_t* accout,
              size_t n) {
  __shared__ uint32_t lds_base[TABLE_BASE8_ENTRIES * TABLE_ENTRY_WORDS];
  for (int j = threadIdx.x; j < TABLE_BASE8_ENTRIES * TABLE_ENTRY_WORDS; j += BLOCK) lds_base[j] = base8[j];
  __syncthreads();

  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;
  const uint32_t* tab = table + i * (VERIFY_TABLE_ENTRIES * 40);
  uint32_t tw[8], sw[8];
  {
    const uint4* d = reinterpret_cast<const uint4*>(digits + 16 * item);
    const uint4 a = d[0], b = d[1], c = d[2], e = d[3];
    tw[0] = a.x; tw[1] = a.y; tw[2] = a.z; tw[3] = a.w; tw[4] = b.x; tw[5] = b.y; tw[6] = b.z; tw[7] = b.w;
    sw[0] = c.x; sw[1] = c.y; sw[2] = c.z; sw[3] = c.w; sw[4] = e.x; sw[5] = e.y; sw[6] = e.z; sw[7] = e.w;
  }
  ge acc;
  ge_neutral(acc);
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {
    if (w != 63) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) ge_dbl(acc, acc, k == 3);
    }
    {
      const int dig = (int)(tw[7] >> 28) - 8;
      shl256<4>(tw);
      const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
      ge_cached c;
      cached_load(c, tab, mag);
      ge_cached_cneg(c, dig < 0);
      ge_add_cached(acc, acc, c, (w & 1) == 0);
    }
    if ((w & 1) == 0) {
      const int dig = (int)(sw[7] >> 24) - 128;
      shl256<8>(sw);
      const uint32_t mag = (uint32_t)(dig < 0 ? -dig : dig);
      ge_niels nb;
      niels_load(nb, lds_base + TABLE_ENTRY_WORDS * mag);
      ge_niels_cneg(nb, dig < 0);
      ge_add_niels(acc, acc, nb, false);
    }
  }
  uint32_t* o = accout + (size_t)blockIdx.x * (30 * BLOCK) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 10; j++) {
    o[j * BLOCK] = acc.X.v[j]; o[(10 + j) * BLOCK] = acc.Y.v[j]; o[(20 + j) * BLOCK] = acc.Z.v[j];
  }
}

// encode and compare with R as bytes (ed25519-sha512.c:176-180): a non-canonical R can never
// match.  An off-curve A is rejected outright: DESIGN.md "Off-curve public keys".
//
// The inversion of ed_export (ed.c:161, 254 S + 11 M) is shared by FINISH_K items per lane with
// Montgomery's trick: one inversion of the product of their Z plus 3 multiplications per item.
// Lane t of block b handles lane t of tiles b*K .. b*K+K-1, so every access stays coalesced.
// Z = 0 cannot occur for a curve point (the a = -1 law is complete); items whose key is off the
// curve are rejected anyway, and their Z is replaced by 1 so that it cannot poison the product.
constexpr int FINISH_K = 8;

// Z of lane threadIdx.x of tile blockIdx.x*K + k, or 1 when that item does not exist, its key is
// off the curve, or Z = 0
ED_DEV void finish_load_z(fe& zsel, bool& good, int k, const uint32_t* accin, const uint8_t* flags,
                          size_t n) {
  const size_t tile = (size_t)blockIdx.x * FINISH_K + k;
  const size_t i = tile * BLOCK + threadIdx.x;
  fe z;
  fe_set(z, 1);
  good = false;
  if (i < n) {
    const uint32_t* o = accin + tile * (30 * BLOCK) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 10; j++) z.v[j] = o[(20 + j) * BLOCK];
    good = flags[i] != 0 && !fe_iszero(z);
  }
  fe_set(zsel, 1);
  fe_cmov(zsel, z, good);
}

ED_DEV void finish_item(int k, const fe& zinv, bool good, uint8_t* ok, const uint8_t* sigs,
                        const uint32_t* accin, size_t n) {
  const size_t tile = (size_t)blockIdx.x * FINISH_K + k;
  const size_t i = tile * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t* o = accin + tile * (30 * BLOCK) + threadIdx.x;
  fe x, y;
#pragma unroll
  for (int j = 0; j < 10; j++) { x.v[j] = o[j * BLOCK]; y.v[j] = o[(10 + j) * BLOCK]; }
  fe_mul(x, x, zinv);
  fe_mul(y, y, zinv);
  uint32_t cw[8], rw[8];
  fe_tobytes(cw, y);
  cw[7] |= fe_parity(x) << 31;
  load32(rw, sigs, i, 64);
  uint32_t diff = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) diff |= cw[q] ^ rw[q];
  ok[i] = (uint8_t)((diff == 0) && good);
}

__global__ void __launch_bounds__(BLOCK, 2)
k_verify_finish(uint8_t* ok, const uint8_t* sigs, const uint32_t* accin, const uint8_t* flags, size_t n) {
  // straight-line on purpose: an fe[8] array indexed in a loop ends up in scratch
  fe z0, z1, z2, z3, z4, z5, z6, z7, p1, p2, p3, p4, p5, p6, p7, u, zi;
  bool g0, g1, g2, g3, g4, g5, g6, g7;
  finish_load_z(z0, g0, 0, accin, flags, n);
  finish_load_z(z1, g1, 1, accin, flags, n); fe_mul(p1, z0, z1);
  finish_load_z(z2, g2, 2, accin, flags, n); fe_mul(p2, p1, z2);
  finish_load_z(z3, g3, 3, accin, flags, n); fe_mul(p3, p2, z3);
  finish_load_z(z4, g4, 4, accin, flags, n); fe_mul(p4, p3, z4);
  finish_load_z(z5, g5, 5, accin, flags, n); fe_mul(p5, p4, z5);
  finish_load_z(z6, g6, 6, accin, flags, n); fe_mul(p6, p5, z6);
  finish_load_z(z7, g7, 7, accin, flags, n); fe_mul(p7, p6, z7);
  fe_inv(u, p7);                                 // u = 1 / (z0 ... z7)
  fe_mul(zi, u, p6); finish_item(7, zi, g7, ok, sigs, accin, n); fe_mul(u, u, z7);
  fe_mul(zi, u, p5); finish_item(6, zi, g6, ok, sigs, accin, n); fe_mul(u, u, z6);
  fe_mul(zi, u, p4); finish_item(5, zi, g5, ok, sigs, accin, n); fe_mul(u, u, z5);
  fe_mul(zi, u, p3); finish_item(4, zi, g4, ok, sigs, accin, n); fe_mul(u, u, z4);
  fe_mul(zi, u, p2); finish_item(3, zi, g3, ok, sigs, accin, n); fe_mul(u, u, z3);
  fe_mul(zi, u, p1); finish_item(2, zi, g2, ok, sigs, accin, n); fe_mul(u, u, z2);
  fe_mul(zi, u, z0); finish_item(1, zi, g1, ok, sigs, accin, n); fe_mul(u, u, z1);
  finish_item(0, u, g0, ok, sigs, accin, n);
}

// ---------------------------------------------------------------------------------------------
// fixed-base path: ed.c:346-430 (scale16, ed_scale_base) and its callers
// ---------------------------------------------------------------------------------------------
// Same comb as the reference: 64 signed 4-bit digits of (x + 0x88..8); even digits accumulate in
// R0, odd digits in R1, both from row i of comb[32][8] (staged in LDS); R1 <- 16 R1; R0 + R1.
// The scalar is secret here, so the lookup keeps the reference's constant-time discipline
// (ed.c:359-390): every lane reads all eight entries of the row (a wave-uniform LDS address,
// served as a broadcast) and keeps the one it needs with v_cndmask; no secret-dependent address,
// no secret-dependent branch.

ED_DEV void comb_select(ge_niels& e, const uint32_t* lds_row, int digit) {
  const uint32_t mag = (uint32_t)(digit < 0 ? -digit : digit);
  fe_set(e.ymx, 1); fe_set(e.ypx, 1); fe_set(e.t2d, 0);        // ed.c:73 pced_zero
#pragma unroll
  for (uint32_t k = 0; k < 8; k++) {
    ge_niels c;
    niels_load(c, lds_row + TABLE_ENTRY_WORDS * k);
    const bool hit = (mag == k + 1);
    fe_cmov(e.ymx, c.ymx, hit); fe_cmov(e.ypx, c.ypx, hit); fe_cmov(e.t2d, c.t2d, hit);
  }
  ge_niels_cneg(e, digit < 0);
}

// out = x * B for a reduced scalar given as eight little-endian words (consumed)
ED_DEV void ge_scale_base(ge& out, uint32_t xw[8], const uint32_t* lds_comb) {
  words_add_pattern(xw, 0x88888888u);            // ed.c:407-409
  ge r0, r1;
  ge_neutral(r0); ge_neutral(r1);
#pragma unroll 1
  for (int i = 0; i < 32; i++) {
    const uint32_t byte = xw[0] & 0xffu;
#pragma unroll
    for (int k = 0; k < 7; k++) xw[k] = (xw[k] >> 8) | (xw[k + 1] << 24);
    xw[7] >>= 8;
    ge_niels e;
    comb_select(e, lds_comb + TABLE_ENTRY_WORDS * 8 * i, (int)(byte & 15u) - 8);
    ge_add_niels(r0, r0, e, true);
    comb_select(e, lds_comb + TABLE_ENTRY_WORDS * 8 * i, (int)(byte >> 4) - 8);
    ge_add_niels(r1, r1, e, true);
  }
#pragma unroll 1
  for (int k = 0; k < 4; k++) ge_dbl(r1, r1, k == 3);
  ge_cached c;
  ge_to_cached(c, r1);
  ge_add_cached(out, r0, c, false);
}

ED_DEV void stage_comb(uint32_t* lds_comb, const uint32_t* comb) {
  for (int j = threadIdx.x; j < TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS; j += BLOCK) lds_comb[j] = comb[j];
  __syncthreads();
}

// ed25519-sha512.c:31-47 ed25519_key_setup: h = SHA-512(sk), clamped
ED_DEV void key_setup(uint32_t h[16], const uint32_t sk[8]) {
  sha512_prefix_msg<8>(h, sk, nullptr, 0);
  h[0] &= 0xfffffff8u;
  h[7] = (h[7] & 0x7fffffffu) | 0x40000000u;
}

ED_DEV void msg_span(const uint8_t*& m, size_t& mlen, const uint8_t* msgs, const uint64_t* msg_off,
                     size_t msg_len, size_t item) {
  if (msg_off) { m = msgs + msg_off[item]; mlen = (size_t)(msg_off[item + 1] - msg_off[item]); }
  else { m = msgs + item * msg_len; mlen = msg_len; }
}

// ed25519-sha512.c:53-67 genpub
__global__ void __launch_bounds__(BLOCK, 2)
k_genpub(uint8_t* pubs, const uint8_t* secs, size_t n, const uint32_t* comb) {
  __shared__ uint32_t lds_comb[TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS];
  stage_comb(lds_comb, comb);
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;
  uint32_t sk[8], h[16], aw[8], out[8];
  load32(sk, secs, item, 32);
  key_setup(h, sk);
  sc a;
  sc_from_words<8>(a, h);
  sc_to_words(aw, a);
  ge A;
  ge_scale_base(A, aw, lds_comb);
  ge_tobytes(out, A);
  if (i < n) store32(pubs, i, 32, out);
}

// ed25519-sha512.c:84-123 sign
__global__ void __launch_bounds__(BLOCK, 2)
k_sign(uint8_t* sigs, const uint8_t* secs, const uint8_t* pubs, const uint8_t* msgs,
       const uint64_t* msg_off, size_t msg_len, size_t n, const uint32_t* comb) {
  __shared__ uint32_t lds_comb[TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS];
  stage_comb(lds_comb, comb);
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;
  const uint8_t* m; size_t mlen;
  msg_span(m, mlen, msgs, msg_off, msg_len, item);

  uint32_t sk[8], h[16], dig[16], rw[8], Rw[8], Sw[8];
  load32(sk, secs, item, 32);
  key_setup(h, sk);
  sc a, r, t, S;
  sc_from_words<8>(a, h);
  sha512_prefix_msg<8>(dig, h + 8, m, mlen);     // r = H(h[32..64) || M)
  sc_from_words<16>(r, dig);
  sc_to_words(rw, r);
  {
    ge R;
    ge_scale_base(R, rw, lds_comb);
    ge_tobytes(Rw, R);
  }
  {
    uint32_t pre[16];
    load32(pre + 8, pubs, item, 32);
#pragma unroll
    for (int k = 0; k < 8; k++) pre[k] = Rw[k];
    sha512_prefix_msg<16>(dig, pre, m, mlen);    // t = H(R || A || M)
  }
  sc_from_words<16>(t, dig);
  sc_mul(S, t, a);
  sc_add(S, r, S);
  sc_to_words(Sw, S);
  if (i < n) { store32(sigs, i, 64, Rw); store32(sigs + 32, i, 64, Sw); }
}

// x25519.c:158-197 do_x25519_base
__global__ void __launch_bounds__(BLOCK, 2)
k_x25519_base(uint8_t* out, const uint8_t* scalars, size_t n, const uint32_t* comb) {
  __shared__ uint32_t lds_comb[TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS];
  stage_comb(lds_comb, comb);
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;
  uint32_t s[8], xw[8], o[8];
  load32(s, scalars, item, 32);
  clamp(s);
  sc x;
  sc_from_words<8>(x, s);
  sc_to_words(xw, x);
  ge R;
  ge_scale_base(R, xw, lds_comb);
  fe u, t;
  fe_sub(t, R.Z, R.Y);                           // 3u
  fe_inv(t, t);
  fe_add(u, R.Z, R.Y);                           // 2u
  fe_mul(u, u, t);
  fe_tobytes(o, u);
  if (i < n) store32(out, i, 32, o);
}

// ed25519-sha512.c:187-232 pk_ed25519_to_x25519: u = (z + y) / (z - y) of the imported point.
// ed_import always returns z = 1 and y = the 255 low bits of the input taken mod p; x (and the
// square root that produces it) never reaches the output, so it is not computed.
__global__ void __launch_bounds__(BLOCK, 2)
k_pk_to_x(uint8_t* out, const uint8_t* in, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8], o[8];
  load32(w, in, i, 32);
  w[7] &= 0x7fffffffu;
  fe y, one, u, t;
  fe_frombytes(y, w);
  fe_set(one, 1);
  fe_sub(t, one, y);                             // 3u
  fe_inv(t, t);                                  // 1 - y = 0 -> 0, as fld_inv
  fe_add(u, one, y);
  fe_mul(u, u, t);
  fe_tobytes(o, u);
  store32(out, i, 32, o);
}

// ed25519-sha512.c:239-256 sk_ed25519_to_x25519
__global__ void __launch_bounds__(BLOCK, 2)
k_sk_to_x(uint8_t* out, const uint8_t* in, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t sk[8], h[16];
  load32(sk, in, i, 32);
  key_setup(h, sk);
  store32(out, i, 32, h);
}

}  // namespace ed

// =============================================================================================
// launchers (host side of this translation unit)
// =============================================================================================

using namespace ed;

extern "C" {

hipError_t edk_init_tables(uint32_t* base8, uint32_t* comb, hipStream_t stream) {
  const int total = TABLE_BASE8_ENTRIES + 256;
  hipLaunchKernelGGL(k_init_tables, dim3((total + 63) / 64), dim3(64), 0, stream, base8, comb);
  return hipGetLastError();
}

hipError_t edk_x25519(uint8_t* out, const uint8_t* scalars, const uint8_t* points, size_t n,
                      hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const size_t blocks = (n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_x25519, dim3((unsigned)blocks), dim3(BLOCK), 0, stream, out, scalars, points, n);
  return hipGetLastError();
}

hipError_t edk_verify(uint8_t* ok, const uint8_t* sigs, const uint8_t* pubs, const uint8_t* msgs,
                      const uint64_t* msg_off, size_t msg_len, size_t n, const uint32_t* base8,
                      const edk_verify_ws* ws, hipEvent_t* marks, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + BLOCK - 1) / BLOCK);
  if (marks) (void)hipEventRecord(marks[0], stream);
  hipLaunchKernelGGL(k_verify_prepare, dim3(blocks), dim3(BLOCK), 0, stream, sigs, pubs, msgs, msg_off,
                     msg_len, n, ws->digits, ws->table, ws->flags);
  if (marks) (void)hipEventRecord(marks[1], stream);
  hipLaunchKernelGGL(k_verify_main, dim3(blocks), dim3(BLOCK), 0, stream, ws->digits, ws->table, base8,
                     ws->acc, n);
  if (marks) (void)hipEventRecord(marks[2], stream);
  hipLaunchKernelGGL(k_verify_finish, dim3((blocks + FINISH_K - 1) / FINISH_K), dim3(BLOCK), 0, stream, ok, sigs,
                     ws->acc, ws->flags, n);
  if (marks) (void)hipEventRecord(marks[3], stream);
  return hipGetLastError();
}

#define EDK_GRID(n) dim3((unsigned)(((n) + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream

hipError_t edk_genpub(uint8_t* pubs, const uint8_t* secs, size_t n, const uint32_t* comb, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_genpub, EDK_GRID(n), pubs, secs, n, comb);
  return hipGetLastError();
}

hipError_t edk_sign(uint8_t* sigs, const uint8_t* secs, const uint8_t* pubs, const uint8_t* msgs,
                    const uint64_t* msg_off, size_t msg_len, size_t n, const uint32_t* comb,
                    hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_sign, EDK_GRID(n), sigs, secs, pubs, msgs, msg_off, msg_len, n, comb);
  return hipGetLastError();
}

hipError_t edk_x25519_base(uint8_t* out, const uint8_t* scalars, size_t n, const uint32_t* comb,
                           hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_x25519_base, EDK_GRID(n), out, scalars, n, comb);
  return hipGetLastError();
}

hipError_t edk_pk_to_x(uint8_t* out, const uint8_t* in, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_pk_to_x, EDK_GRID(n), out, in, n);
  return hipGetLastError();
}

hipError_t edk_sk_to_x(uint8_t* out, const uint8_t* in, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_sk_to_x, EDK_GRID(n), out, in, n);
  return hipGetLastError();
}

}  // extern "C"
