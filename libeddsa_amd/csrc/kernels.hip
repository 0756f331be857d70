// kernels.hip - the batched Ed25519 / X25519 kernels for gfx950 and their launchers.
//
// One curve operation per lane, limbs in registers (fe25519.h), no MFMA.  What a lane computes is
// in lanes.h; this file maps items to lanes, stages the shared tables in LDS and moves the packed,
// item-major byte arrays of the batch API (include/eddsa_amd.h) in and out.
//
//   k_x25519         x25519.c:129-150 do_x25519                      (config 3)
//   k_verify_*       ed25519-sha512.c:148-181 ed25519_verify        (config 2, 4)
//   k_sign           ed25519-sha512.c:84-123 sign                    (config 5)
//   k_genpub         ed25519-sha512.c:53-67 genpub
//   k_x25519_base    x25519.c:158-197 do_x25519_base
//   k_pk_to_x        ed25519-sha512.c:187-232 pk_ed25519_to_x25519
//   k_sk_to_x        ed25519-sha512.c:239-256 sk_ed25519_to_x25519
//   k_init_tables    generates what the reference ships as lib/ed_lookup64.h
#include "eddsa_kernels.h"
#include "lanes.h"

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

ED_DEV void msg_span(const uint8_t*& m, size_t& mlen, const uint8_t* msgs, const uint64_t* msg_off,
                     size_t msg_len, size_t item) {
  if (msg_off) { m = msgs + msg_off[item]; mlen = (size_t)(msg_off[item + 1] - msg_off[item]); }
  else { m = msgs + item * msg_len; mlen = msg_len; }
}

// copy a table of `words` 32-bit words from HBM into LDS (whole block)
ED_DEV void stage_table(uint32_t* lds, const uint32_t* src, int words) {
  for (int j = threadIdx.x; j < words; j += BLOCK) lds[j] = src[j];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BLOCK, 4)
k_x25519(uint8_t* out, const uint8_t* scalars, const uint8_t* points, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8], pt[8], r[8];
  load32(s, scalars, i, 32);
  load32(pt, points, i, 32);
  x25519_lane(r, s, pt);
  store32(out, i, 32, r);
}

__global__ void __launch_bounds__(64) k_init_tables(uint32_t* base16, uint32_t* comb) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  if (id >= TABLE_BASE16_ENTRIES + TABLE_COMB_ENTRIES) return;
  if (id < TABLE_BASE16_ENTRIES) {
    table_entry_lane(base16 + (size_t)TABLE_ENTRY_WORDS * id, (uint32_t)id, 0);
  } else {
    const int c = id - TABLE_BASE16_ENTRIES;      // comb[i][k], c = 8 i + k
    table_entry_lane(comb + TABLE_ENTRY_WORDS * c, (uint32_t)(c & 7) + 1, 8u * (uint32_t)(c >> 3));
  }
}

// ---------------------------------------------------------------------------------------------
// Ed25519 verify.  Three kernels per chunk, so that each stays inside its register budget and its
// own I-cache footprint (one fused kernel spilled 4.5 KB/lane):
//   k_verify_prepare  hash, scalars -> digit words, decompress -A, table of 0..8 * -A
//   k_verify_main     the 252 doublings + 80 additions            (~85 % of the time)
//   k_verify_finish   invert Z (shared by 8 items per lane), encode, compare with R
// Workspace (HBM; tile = 256 items):
//   digits [item][16]                  t + 0x88.., S + 0x80.. as little-endian words
//   table  [item][entry 9][word 40]    1440 contiguous bytes per item
//   acc    [tile][word 30][lane 256]   X, Y, Z of the result
//   flags  [item]                      1 = A decoded to a curve point
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BLOCK, 2)
k_verify_prepare(const uint8_t* sigs, const uint8_t* pubs, const uint8_t* msgs,
                 const uint64_t* msg_off, size_t msg_len, size_t n, uint32_t* digits,
                 uint32_t* table, uint8_t* flags) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;         // idle lanes redo the last item into their own slot
  uint32_t rw[8], aw[8], sw[8], tw[8];
  load32(rw, sigs, item, 64);
  load32(sw, sigs + 32, item, 64);
  load32(aw, pubs, item, 32);
  const uint8_t* m; size_t mlen;
  msg_span(m, mlen, msgs, msg_off, msg_len, item);
  const bool oncurve = verify_prepare_lane(tw, sw, table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS),
                                           rw, aw, m, mlen);
  uint4* d = reinterpret_cast<uint4*>(digits + 16 * i);
  d[0] = make_uint4(tw[0], tw[1], tw[2], tw[3]); d[1] = make_uint4(tw[4], tw[5], tw[6], tw[7]);
  d[2] = make_uint4(sw[0], sw[1], sw[2], sw[3]); d[3] = make_uint4(sw[4], sw[5], sw[6], sw[7]);
  flags[i] = (uint8_t)oncurve;
}

__global__ void __launch_bounds__(BLOCK, 4)
k_verify_main(const uint32_t* digits, const uint32_t* table, const uint32_t* base16, uint32_t* accout) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;   // < workspace capacity
  uint32_t tw[8], sw[8];
  {
    const uint4* d = reinterpret_cast<const uint4*>(digits + 16 * i);
    const uint4 a = d[0], b = d[1], c = d[2], e = d[3];
    tw[0] = a.x; tw[1] = a.y; tw[2] = a.z; tw[3] = a.w; tw[4] = b.x; tw[5] = b.y; tw[6] = b.z; tw[7] = b.w;
    sw[0] = c.x; sw[1] = c.y; sw[2] = c.z; sw[3] = c.w; sw[4] = e.x; sw[5] = e.y; sw[6] = e.z; sw[7] = e.w;
  }
  ge acc;
  verify_main_lane(acc, tw, sw, table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS), base16);
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
  uint32_t rw[8];
  load32(rw, sigs, i, 64);
  ok[i] = (uint8_t)(verify_encode_lane(x, y, zinv, rw) && good);
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
// fixed-base kernels: the 256-entry comb (32 KiB) is staged in LDS by every block
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BLOCK, 2)
k_genpub(uint8_t* pubs, const uint8_t* secs, size_t n, const uint32_t* comb) {
  __shared__ alignas(16) uint32_t lds_comb[TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS];
  stage_table(lds_comb, comb, TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS);
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;
  uint32_t sk[8], out[8];
  load32(sk, secs, item, 32);
  genpub_lane(out, sk, lds_comb);
  if (i < n) store32(pubs, i, 32, out);
}

__global__ void __launch_bounds__(BLOCK, 2)
k_sign(uint8_t* sigs, const uint8_t* secs, const uint8_t* pubs, const uint8_t* msgs,
       const uint64_t* msg_off, size_t msg_len, size_t n, const uint32_t* comb) {
  __shared__ alignas(16) uint32_t lds_comb[TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS];
  stage_table(lds_comb, comb, TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS);
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;
  const uint8_t* m; size_t mlen;
  msg_span(m, mlen, msgs, msg_off, msg_len, item);
  uint32_t sk[8], pub[8], Rw[8], Sw[8];
  load32(sk, secs, item, 32);
  load32(pub, pubs, item, 32);
  sign_lane(Rw, Sw, sk, pub, m, mlen, lds_comb);
  if (i < n) { store32(sigs, i, 64, Rw); store32(sigs + 32, i, 64, Sw); }
}

__global__ void __launch_bounds__(BLOCK, 2)
k_x25519_base(uint8_t* out, const uint8_t* scalars, size_t n, const uint32_t* comb) {
  __shared__ alignas(16) uint32_t lds_comb[TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS];
  stage_table(lds_comb, comb, TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS);
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;
  uint32_t s[8], o[8];
  load32(s, scalars, item, 32);
  x25519_base_lane(o, s, lds_comb);
  if (i < n) store32(out, i, 32, o);
}

__global__ void __launch_bounds__(BLOCK, 2)
k_pk_to_x(uint8_t* out, const uint8_t* in, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8], o[8];
  load32(w, in, i, 32);
  pk_to_x_lane(o, w);
  store32(out, i, 32, o);
}

__global__ void __launch_bounds__(BLOCK, 2)
k_sk_to_x(uint8_t* out, const uint8_t* in, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t sk[8], o[8];
  load32(sk, in, i, 32);
  sk_to_x_lane(o, sk);
  store32(out, i, 32, o);
}

}  // namespace ed

// =============================================================================================
// launchers (host side of this translation unit)
// =============================================================================================

using namespace ed;

extern "C" {

hipError_t edk_init_tables(uint32_t* base16, uint32_t* comb, hipStream_t stream) {
  const int total = TABLE_BASE16_ENTRIES + TABLE_COMB_ENTRIES;
  hipLaunchKernelGGL(k_init_tables, dim3((total + 63) / 64), dim3(64), 0, stream, base16, comb);
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
                      const uint64_t* msg_off, size_t msg_len, size_t n, const uint32_t* base16,
                      const edk_verify_ws* ws, hipEvent_t* marks, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + BLOCK - 1) / BLOCK);
  if (marks) (void)hipEventRecord(marks[0], stream);
  hipLaunchKernelGGL(k_verify_prepare, dim3(blocks), dim3(BLOCK), 0, stream, sigs, pubs, msgs, msg_off,
                     msg_len, n, ws->digits, ws->table, ws->flags);
  if (marks) (void)hipEventRecord(marks[1], stream);
  hipLaunchKernelGGL(k_verify_main, dim3(blocks), dim3(BLOCK), 0, stream, ws->digits, ws->table, base16,
                     ws->acc);
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
