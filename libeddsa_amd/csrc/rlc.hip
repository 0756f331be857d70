// rlc.hip - opt-in batch verification by random linear combination (the reference's own
// TODO at lib/ed25519-sha512.c:13-14: "batch verification").  Never the default: see the caveat below.
//
// A group of G = 8192 items is accepted as a whole when
//
//     ( sum_i z_i S_i ) B  +  sum_i ( z_i t_i ) (-A_i)  +  sum_i z_i (-R_i)  =  neutral element
//
// with t_i = SHA-512(R_i || A_i || M_i) mod l and S_i mod l exactly as the per-item check takes them
// (lib/ed25519-sha512.c:162-172), and 126-bit odd coefficients z_i = SHA-512(seed || i) where seed is a
// SHA-512 hash tree over the WHOLE batch (signatures, keys and message digests), so the coefficients
// are fixed only after every byte of the batch is (Fiat-Shamir; deterministic, reproducible).  Every
// term z_i (S_i B - t_i A_i - R_i) of an item the per-item check accepts is the neutral element, so a
// group of accepted items always passes; if all points of a group lie in the prime-order subgroup
// (every honestly generated key and signature does) a group passes with an item the per-item check
// rejects only with probability about 2^-125.  A group that FAILS is re-verified by the per-item kernels
// (edk_verify: the bit-exact path), so the verdict vector returned is the per-item one in that case.
// Items the combination cannot represent are never combined: R that is not the canonical encoding of a
// curve point is rejected at once (the per-item check compares R as bytes, lib/ed25519-sha512.c:176-180,
// and an encoding produced by ed_export is always canonical); A that does not decode to a curve point
// (lib/ed.c:100-149 never fails) or has small order, and R of small order, send their group to the
// per-item kernels.
// The key's scalar z_i t_i is taken mod 8 l, not mod l (rlc_lanes.h: rlc_key_scalar_mod_8l; z_i itself is an
// integer on R_i and B has order l), so every term is exactly z_i Q_i with Q_i = S_i B - t_i A_i - R_i also
// when A_i or R_i carries a component of order dividing 8 (the reference checks neither subgroup membership
// nor small order).  z_i is odd and below l, hence coprime to the group order 8 l: a combination
// of ONE item passes exactly when the per-item check accepts it.
// CAVEAT (why this is opt-in): the small-order parts D_i of the Q_i live in a group of eight elements, where
// no choice of coefficients separates them: TWO OR MORE crafted items in one group can cancel -- equal
// defects of order 2 always do (odd z_1 + odd z_2 is even), of order 4 with probability 1/2, of order 8
// with 1/4, and whoever controls two items of a group can grind the deterministic coefficients -- so a group
// may pass although the reference's cofactorless per-item check rejects those items (or fail although all
// pass; that case only costs time).  tests/test_device_source_on_host.py pins both statements.
//
// The multi-scalar multiplication is a bucket method laid out for the wavefront: one wave of 64 lanes
// per (group, byte-window), 128 buckets for the signed 8-bit digits: the wave counting-sorts the group's
// digits of that window in LDS (18 KB), hands the buckets to its lanes in pairs by size rank (the fullest
// with the emptiest ...: a wave takes as long as its busiest lane), each lane adds the points of its two
// buckets (mixed additions, ed_add_pc's 7 M), and the weighted sum of the 128 buckets is fifteen additions
// deep through LDS; k_rlc_horner runs Horner over the 48 window points of each group (four lanes per
// point, quad_lanes.h: 248 dependent doublings are pure latency) BESIDE the bucket blocks, window by window
// as their points arrive.
// -A_i has 32 windows (z_i t_i mod 8 l, centred: |.| <= 4 l), -R_i 16 (z_i), B one entry per group and window: 48 mixed
// additions per item instead of the per-item kernel's 252 doublings + 80 additions.  Algorithmic HBM
// bytes: the same 129 per item as verify.  The per-lane arithmetic (decoding and routing flags,
// coefficients, digit recoding) is in rlc_lanes.h, which the host-check build also compiles.
#include "eddsa_kernels.h"
#include "edk_checked.h"
#include "rlc_lanes.h"
#include "quad_lanes.h"

namespace ed {

constexpr int RLC_G = 8192;                      // items per group
constexpr int RLC_BUCKETS = 128;                 // |digit| in 1..128
constexpr int RLC_SEG_WINDOWS = 1;               // windows per workgroup (2 and 4, with the doublings in between, measured slower)
constexpr int RLC_SEGS_A = RLC_WINDOWS_A / RLC_SEG_WINDOWS, RLC_SEGS_R = RLC_WINDOWS_R / RLC_SEG_WINDOWS;
constexpr int RLC_SEGS = RLC_SEGS_A + RLC_SEGS_R;   // window points per group: 32 for -A, 16 for -R
constexpr int RLC_BLOCK = 256;
constexpr uint32_t RLC_BASE_IDX = RLC_G;         // list entry that stands for the base point B
constexpr int RLC_GROUP_WORDS = 64;              // per group: word 0 the routing flag, words 1..48 "window point s is in memory" (k_rlc_bucket)
constexpr int RLC_HORNER_GROUPS = 16;            // groups per Horner wave (four lanes each)
constexpr uint32_t RLC_HORNER_PATIENCE = 1u << 15;   // polls (0.4 us each, measured) for ONE window before a wave of k_rlc_horner leaves its groups
                                                     // to k_rlc_final: 14 ms, where the whole bucket launch takes 2
constexpr uint8_t RLC_UNDECIDED = 2;             // gok[g]: k_rlc_horner gave the group up


// workspace carving (bytes), capacity = a multiple of 2048 items
struct rlc_layout {
  size_t groups;
  size_t ts, leaf, niels_a, niels_r, dig, flags, bsum, bdig, gflags, gok, seg, tree, hook, total;
};
__host__ __device__ inline size_t rlc_align(size_t x) { return (x + 255) & ~(size_t)255; }
__host__ inline rlc_layout rlc_carve(size_t cap) {
  rlc_layout L;
  L.groups = (cap + RLC_G - 1) / RLC_G;
  size_t o = 0;
  L.ts = o;      o += rlc_align(cap * 64);                       // t | S mod l, 8 + 8 words per item
  L.leaf = o;    o += rlc_align(cap * 32);
  L.niels_a = o; o += rlc_align(cap * 128);
  L.niels_r = o; o += rlc_align(cap * 128);
  L.dig = o;     o += rlc_align(L.groups * RLC_WINDOWS * (size_t)RLC_G);
  L.flags = o;   o += rlc_align(cap);
  L.bsum = o;    o += rlc_align((L.groups * (RLC_G / RLC_BLOCK) + 1) * 40);   // per block of 256 items: sum of z S, 9 words (+1 pad)
  L.bdig = o;    o += rlc_align(L.groups * 32);
  L.gflags = o;  o += rlc_align(L.groups * RLC_GROUP_WORDS * 4);         // per group: routing flag, window flags
  L.gok = o;     o += rlc_align(L.groups);
  L.seg = o;     o += rlc_align(L.groups * RLC_SEGS * VERIFY_ENTRY_WORDS * 4);      // window points, cached form (packed)
  L.tree = o;    o += rlc_align((cap / RLC_TREE_FAN + 2) * 32 * 2);
  L.hook = o;    o += 256;                                       // word 0: test hook (eddsa_amd_debug_withhold_handoff); word 1: waves of k_rlc_horner that gave up; zeroed at allocation
  L.total = o;
  return L;
}

ED_DEV void load_words8(uint32_t w[8], const uint8_t* p) {
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++)
      w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
  }
}

// ---------------------------------------------------------------------------------------------
// R1: per item, t and S mod l (ed25519-sha512.c:162-172) and the item's leaf of the batch hash tree
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RLC_BLOCK, 2)
k_rlc_hash(edk_verify_src src, size_t n, uint32_t* ts, uint32_t* leaf, const uint32_t* perm) {
  const size_t g = (size_t)blockIdx.x * RLC_BLOCK + threadIdx.x;
  if (g >= n) return;
  const size_t i = perm ? perm[g] : g;           // ragged messages: a wave hashes items of one length (edk_msg_order)
  uint32_t rw[8], aw[8], sw[8], tw[8], lf[8];
  load_words8(rw, src.sigs + i * src.sig_stride);
  load_words8(aw, src.pubs + i * src.pub_stride);
  load_words8(sw, src.sigs + i * src.sig_stride + 32);
  const uint8_t* m; size_t mlen;
  msg_span(m, mlen, src.msgs, src.msg_off, src.msg_end, src.msg_len, src.msg_stride, i);
  rlc_hash_lane(tw, sw, lf, rw, aw, src.sigs + i * src.sig_stride + 32, m, mlen);
  uint4* o = reinterpret_cast<uint4*>(ts + 16 * i);
  o[0] = make_uint4(tw[0], tw[1], tw[2], tw[3]); o[1] = make_uint4(tw[4], tw[5], tw[6], tw[7]);
  o[2] = make_uint4(sw[0], sw[1], sw[2], sw[3]); o[3] = make_uint4(sw[4], sw[5], sw[6], sw[7]);
  uint4* l = reinterpret_cast<uint4*>(leaf + 8 * i);
  l[0] = make_uint4(lf[0], lf[1], lf[2], lf[3]); l[1] = make_uint4(lf[4], lf[5], lf[6], lf[7]);
}

// R2: one level of the hash tree: node j = SHA-512(children 16 j .. 16 j + 15)[0..32).  (A lane hashes its children's 512 bytes
// block after block: the tree is latency, five blocks per level and five levels at 2^20 items.  With a fan-in of 64 - 17 blocks
// per level, three levels and a fourth of one node - it ran 0.6-1.0 ms beside k_rlc_points and was the critical path of passes
// below 2^19 items: 2^17 items 1.48 -> 1.32 ms, 2^18 1.80 -> 1.62, 2^20 4.47 -> 4.41: profiles/r06_rlc_bucket.txt.)
__global__ void __launch_bounds__(64)
k_rlc_tree(const uint32_t* in, uint32_t* out, size_t count) {
  const size_t j = (size_t)blockIdx.x * 64 + threadIdx.x;
  const size_t lo = j * RLC_TREE_FAN;
  if (lo >= count) return;
  const size_t cnt = count - lo < (size_t)RLC_TREE_FAN ? count - lo : (size_t)RLC_TREE_FAN;
  uint32_t d[16];
  sha512_prefix_msg<0>(d, nullptr, reinterpret_cast<const uint8_t*>(in + 8 * lo), 32 * cnt);
#pragma unroll
  for (int k = 0; k < 8; k++) out[8 * j + k] = d[k];
}

// ---------------------------------------------------------------------------------------------
// R3: per item, -A and -R as affine niels points; the routing flags
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RLC_BLOCK, 2)
k_rlc_points(edk_verify_src src, size_t n, uint32_t* niels_a, uint32_t* niels_r, uint8_t* flags, uint32_t* gflags) {
  const size_t i = (size_t)blockIdx.x * RLC_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  ge_niels nl;
  load_words8(w, src.pubs + i * src.pub_stride);
  uint8_t fl = rlc_decode_key_lane(nl, w);
  niels_store(niels_a + 32 * i, nl);
  load_words8(w, src.sigs + i * src.sig_stride);
  fl |= rlc_decode_r_lane(nl, w);
  niels_store(niels_r + 32 * i, nl);
  flags[i] = fl;
  if (fl & RLC_PER_ITEM) atomicOr(gflags + RLC_GROUP_WORDS * (i / RLC_G), 1u);
}

// ---------------------------------------------------------------------------------------------
// R4: per item, the coefficient z_i, the scalars z_i t_i and z_i S_i mod l, their signed byte digits
// ---------------------------------------------------------------------------------------------
ED_DEV void add_words9(uint32_t a[9], const uint32_t b[9]) {
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { c += (uint64_t)a[k] + b[k]; a[k] = (uint32_t)c; c >>= 32; }
}

__global__ void __launch_bounds__(RLC_BLOCK, 2)
k_rlc_scalars(size_t n, const uint32_t* ts, const uint32_t* seed, const uint8_t* flags, int8_t* dig, uint32_t* bsum) {
  __shared__ uint32_t red[RLC_BLOCK * 9];
  const size_t i = (size_t)blockIdx.x * RLC_BLOCK + threadIdx.x;
  uint32_t zs[9];
#pragma unroll
  for (int k = 0; k < 9; k++) zs[k] = 0;
  {                                              // (the grid covers whole groups)
    const size_t g = i / RLC_G, k = i % RLC_G;
    int8_t* d = dig + g * (size_t)RLC_WINDOWS * RLC_G + k;
    if (i < n && (flags[i] & RLC_R_VALID)) {
      const uint4* p = reinterpret_cast<const uint4*>(ts + 16 * i);
      const uint4 t0 = p[0], t1 = p[1], s0 = p[2], s1 = p[3];
      const uint32_t tw[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      const uint32_t sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      uint32_t sd[8];
#pragma unroll
      for (int q = 0; q < 8; q++) sd[q] = seed[q];
      int8_t da[RLC_WINDOWS_A], dr[RLC_WINDOWS_R];
      rlc_scalars_lane(da, dr, zs, sd, (uint64_t)i, tw, sw);
#pragma unroll
      for (int wd = 0; wd < RLC_WINDOWS_A; wd++) d[(size_t)wd * RLC_G] = da[wd];
#pragma unroll
      for (int wd = 0; wd < RLC_WINDOWS_R; wd++) d[(size_t)(RLC_WINDOWS_A + wd) * RLC_G] = dr[wd];
    } else {                                     // rejected outright, or a slot past the end: contributes nothing
#pragma unroll
      for (int wd = 0; wd < RLC_WINDOWS; wd++) d[(size_t)wd * RLC_G] = 0;
    }
  }
  // sum of z_i S_i over the block's 256 items (a 9-word integer; reduced mod l per group later)
#pragma unroll
  for (int k = 0; k < 9; k++) red[k * RLC_BLOCK + threadIdx.x] = zs[k];
  __syncthreads();
  for (int stride = RLC_BLOCK / 2; stride >= 1; stride >>= 1) {
    if ((int)threadIdx.x < stride) {
      uint32_t o[9];
#pragma unroll
      for (int k = 0; k < 9; k++) o[k] = red[k * RLC_BLOCK + threadIdx.x + stride];
      add_words9(zs, o);
#pragma unroll
      for (int k = 0; k < 9; k++) red[k * RLC_BLOCK + threadIdx.x] = zs[k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) bsum[10 * (size_t)blockIdx.x + k] = zs[k];
  }
}

// R4b: per group, s_g = sum of z_i S_i mod l and its signed byte digits (the base point's windows)
__global__ void __launch_bounds__(64)
k_rlc_group_scalar(size_t n, const uint32_t* bsum, int8_t* bdig) {
  const size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
  const size_t groups = (n + RLC_G - 1) / RLC_G;
  if (g >= groups) return;
  const size_t per = RLC_G / RLC_BLOCK;
  uint32_t acc[16];
#pragma unroll
  for (int k = 0; k < 16; k++) acc[k] = 0;
  for (size_t b = g * per; b < (g + 1) * per; b++) {
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) { c += (uint64_t)acc[k] + (k < 9 ? bsum[10 * b + k] : 0u); acc[k] = (uint32_t)c; c >>= 32; }
  }
  int8_t dg[32];
  rlc_group_scalar_lane(dg, acc);
#pragma unroll
  for (int wd = 0; wd < 32; wd++) bdig[32 * g + wd] = dg[wd];
}

// ---------------------------------------------------------------------------------------------
// R5: the buckets.  One wave per (group, window); a lane adds up two buckets, a full and a sparse one.
// ---------------------------------------------------------------------------------------------
constexpr int RLC_LANES = 64;
struct rlc_lds {
  uint32_t hist[RLC_BUCKETS + 4];
  uint32_t cursor[RLC_BUCKETS + 4];
  uint32_t first[RLC_BUCKETS + 4];               // where each bucket's run starts in the sorted list
  uint32_t perm[RLC_BUCKETS];                    // rank by size -> bucket
  union {
    uint16_t list[RLC_G + 2];                    // entry = item index in the group (or RLC_BASE_IDX) | sign << 15
    uint32_t pts[RLC_LANES * 40];                // exchange area of the weighted sum (after the list's last use)
  };
};

ED_DEV void lds_put_at(uint32_t* pts, const ge& p, int slot) {
#pragma unroll
  for (int j = 0; j < 10; j++) {
    pts[j * RLC_LANES + slot] = p.X.v[j];        pts[(10 + j) * RLC_LANES + slot] = p.Y.v[j];
    pts[(20 + j) * RLC_LANES + slot] = p.Z.v[j]; pts[(30 + j) * RLC_LANES + slot] = p.T.v[j];
  }
}
ED_DEV void lds_put(uint32_t* pts, const ge& p) { lds_put_at(pts, p, (int)threadIdx.x); }
ED_DEV void lds_get(ge& p, const uint32_t* pts, int lane) {
#pragma unroll
  for (int j = 0; j < 10; j++) {
    p.X.v[j] = pts[j * RLC_LANES + lane];        p.Y.v[j] = pts[(10 + j) * RLC_LANES + lane];
    p.Z.v[j] = pts[(20 + j) * RLC_LANES + lane]; p.T.v[j] = pts[(30 + j) * RLC_LANES + lane];
  }
}
ED_DEV void ge_add_full(ge& r, const ge& p, const ge& q) {
  ge_cached c;
  ge_to_cached(c, q);
  ge_add_cached(r, p, c, true);
}
ED_DEV void ge_cmov(ge& r, const ge& p, bool flag) {
  fe_cmov(r.X, p.X, flag); fe_cmov(r.Y, p.Y, flag); fe_cmov(r.Z, p.Z, flag); fe_cmov(r.T, p.T, flag);
}

// Window points pass from the bucket blocks to the Horner waves INSIDE the launch, and the eight XCDs' L2 caches are
// not coherent with each other: an acquire / release pair at agent scope writes back and invalidates the whole L2
// (buffer_wbl2 sc1 / buffer_inv sc1) - issued by each of the 6144 blocks it emptied the cache under the other blocks'
// point loads (tried: + 0.4 ms).  So these few words travel by agent-scope accesses (sc1: written through to and read
// from memory) and nothing else in the kernel is touched: the writer waits for its stores before it raises the flag,
// the reader's loads follow a branch on the flag it read.
ED_DEV void coherent_store8(uint32_t* dst, const uint32_t w[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) __hip_atomic_store(dst + j, w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ED_DEV void coherent_load8(uint32_t w[8], const uint32_t* src) {
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// r += entry, a packed cached entry (Y-X | Y+X | 2dT | 2Z) written by another block of this launch (quad_lanes.h: quad_add_entry)
ED_DEV void quad_add_coherent(fe& r, const uint32_t* e, int q) {
  uint32_t w[8];
  coherent_load8(w, e + 8 * q);
  fe mult, first, m;
  fe_unpack(mult, w);
  quad_stage_a_operand(first, r, q);
  fe_mul(m, first, mult);
  quad_stage_b(r, m, q);
}

// the group's verdict from the total r = (X, Y, T, Z) across the quad: accepted when r is the neutral element and no item
// of the group sent it to the per-item kernels
ED_DEV void rlc_group_verdict(const fe& r, int q, bool live, size_t g, size_t n, const uint32_t* gflags, uint8_t* gok, uint32_t* stats) {
  // neutral element: X = 0, Y = Z and Z != 0 (lane 0 holds X, lane 1 Y, lane 3 Z), as verify_half_main_quad tests it:
  // (0, 0, *, 0) is no point, and is unreachable only while every combined point is on the curve
  fe z, d;
  fe_quad_perm<3, 3, 3, 3>(z, r);
  fe_sub(d, r, z);                               // lane 1: Y - Z
  const bool mine = q == 0 ? fe_iszero(r) : q == 1 ? fe_iszero(d) : q == 3 ? !fe_iszero(r) : true;
  const int all = (int)mine & __shfl_xor((int)mine, 1);
  const bool neutral = (all & __shfl_xor(all, 2)) != 0;
  if (q != 0 || !live) return;
  const bool ok = neutral && gflags[RLC_GROUP_WORDS * g] == 0;
  gok[g] = (uint8_t)ok;
  if (stats) {
    const uint32_t cnt = (uint32_t)(n - g * RLC_G < (size_t)RLC_G ? n - g * RLC_G : (size_t)RLC_G);
    atomicAdd(stats + (ok ? 0 : 1), cnt);        // items decided by the combination / by the per-item kernels
    if (!ok) atomicAdd(stats + 2, 1u);           // groups sent to the per-item kernels
    else atomicAdd(stats + 3, 1u);               // groups decided by the combination
  }
}

// R6: per group, Horner over the window points (highest window first; the 16 windows of -R carry the weights of the 16
// lowest windows of -A): the total must be the neutral element.  248 dependent doublings per group: pure latency, the
// price of any 253-bit multi-scalar multiplication, so it runs in the four-lanes-per-point form of quad_lanes.h (a
// doubling is one squaring and one multiplication deep), sixteen groups to a wave.  The kernel runs on the side stream
// BESIDE k_rlc_bucket (60 registers, no LDS: its eight waves fit next to the bucket blocks' two per SIMD) and takes each
// window's points as its flags go up - the bucket blocks run in order of weight, so when the last of them is done all that
// is left is the last windows' share of the chain.  (As a kernel after the bucket blocks the evaluation was 0.26 ms
// during which the chip did nothing else.  Tried and dropped: run by the bucket block that completes a part of the windows
// it held that block's wave slot, the SIMD's later blocks started late and the launch ended 0.4 ms later - 6144 blocks are
// exactly three rounds of the chip's 2048 slots, and for the same reason these waves cannot be blocks of the bucket launch;
// as kernels between launches of the bucket blocks by weight on two streams the hardware interleaved the launches and the
// heavy windows were done no sooner.)
// Forward progress: a bucket block waits for nothing and its launch is queued BEFORE this kernel's, so every flag is
// raised whether or not the two launches overlap.  Where they do not - a profiler collecting counters runs one kernel at
// a time, in an order of its own, and may run this one FIRST - a wave that has polled RLC_HORNER_PATIENCE times for one
// window marks its groups RLC_UNDECIDED and ends; k_rlc_final, queued behind both launches on the pass's stream, then
// evaluates those groups the way round 5 evaluated all of them (and ends at once when there are none): the pass costs
// the patience and the old 0.26 ms, the verdicts are the same.
__global__ void __launch_bounds__(4 * RLC_HORNER_GROUPS)
k_rlc_horner(size_t n, const uint32_t* segpts, const uint32_t* gflags, uint8_t* gok, uint32_t* stats, uint32_t* giveups) {
  const size_t groups = (n + RLC_G - 1) / RLC_G;
  const size_t g = (size_t)blockIdx.x * RLC_HORNER_GROUPS + (threadIdx.x >> 2);
  const bool live = g < groups;
  const int q = (int)(threadIdx.x & 3u);
  const uint32_t* gpts = segpts + (live ? g : 0) * RLC_SEGS * VERIFY_ENTRY_WORDS;
  const uint32_t* gw = gflags + RLC_GROUP_WORDS * (live ? g : 0);
  fe r;
  fe_set(r, (uint32_t)(q & 1));                  // neutral element (0, 1, 0, 1) as (X, Y, T, Z)
  bool gave_up = false;
#pragma unroll 1
  for (int w = RLC_WINDOWS_A - 1; w >= 0 && !gave_up; w--) {
    if (w != RLC_WINDOWS_A - 1) {
#pragma unroll 1
      for (int k = 0; k < 8; k++) quad_dbl(r, q);
    }
    const int sa = RLC_WINDOWS_A - 1 - w, sr = RLC_SEGS_A + RLC_WINDOWS_R - 1 - w;
    uint32_t polls = 0;
    for (;;) {
      bool ready = !live || __hip_atomic_load(gw + 1 + sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      if (w < RLC_WINDOWS_R) ready = ready && (!live || __hip_atomic_load(gw + 1 + sr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
      if (__all(ready)) break;
      if (++polls == RLC_HORNER_PATIENCE) { gave_up = true; break; }
      __builtin_amdgcn_s_sleep(32);
    }
    if (gave_up) break;
    quad_add_coherent(r, gpts + sa * VERIFY_ENTRY_WORDS, q);
    if (w < RLC_WINDOWS_R) quad_add_coherent(r, gpts + sr * VERIFY_ENTRY_WORDS, q);
  }
  if (gave_up) {
    if (q == 0 && live) gok[g] = RLC_UNDECIDED;
    if (threadIdx.x == 0) atomicAdd(giveups, 1u);
    return;
  }
  rlc_group_verdict(r, q, live, g, n, gflags, gok, stats);
}

// the groups k_rlc_horner left undecided (none, unless the two launches did not overlap: see there), after both launches:
// every window point is in memory, plain loads
__global__ void __launch_bounds__(4 * RLC_HORNER_GROUPS)
k_rlc_final(size_t n, const uint32_t* segpts, const uint32_t* gflags, uint8_t* gok, uint32_t* stats) {
  const size_t groups = (n + RLC_G - 1) / RLC_G;
  const size_t g = (size_t)blockIdx.x * RLC_HORNER_GROUPS + (threadIdx.x >> 2);
  const bool live = g < groups && gok[g] == RLC_UNDECIDED;
  if (!__any(live)) return;
  const int q = (int)(threadIdx.x & 3u);
  const uint32_t* gpts = segpts + (live ? g : 0) * RLC_SEGS * VERIFY_ENTRY_WORDS;
  fe r;
  fe_set(r, (uint32_t)(q & 1));
#pragma unroll 1
  for (int w = RLC_WINDOWS_A - 1; w >= 0; w--) {
    if (w != RLC_WINDOWS_A - 1) {
#pragma unroll 1
      for (int k = 0; k < 8; k++) quad_dbl(r, q);
    }
    quad_add_entry(r, gpts + (RLC_WINDOWS_A - 1 - w) * VERIFY_ENTRY_WORDS, false, true, q);
    if (w < RLC_WINDOWS_R) quad_add_entry(r, gpts + (RLC_SEGS_A + RLC_WINDOWS_R - 1 - w) * VERIFY_ENTRY_WORDS, false, true, q);
  }
  rlc_group_verdict(r, q, live, g, n, gflags, gok, stats);
}

__global__ void __launch_bounds__(RLC_LANES, 2)
k_rlc_bucket(size_t n, const int8_t* dig, const int8_t* bdig, const uint32_t* niels_a, const uint32_t* niels_r,
             const uint32_t* base16, uint32_t* segpts, uint32_t* gflags, const uint32_t* hook) {
  __shared__ rlc_lds s;
  const size_t groups = (n + RLC_G - 1) / RLC_G;
  // heaviest windows first (by_weight = 0..47): -A's windows 31..16, then windows 15..0 of -A and -R in turn (k_rlc_horner)
  const int by_weight = (int)(blockIdx.x / groups);
  const size_t g = blockIdx.x % groups;
  const bool is_a = by_weight < 16 || ((by_weight - 16) & 1) == 0;
  const int w = by_weight < 16 ? RLC_WINDOWS_A - 1 - by_weight : 15 - ((by_weight - 16) >> 1);
  const int seg = is_a ? RLC_WINDOWS_A - 1 - w : RLC_SEGS_A + RLC_WINDOWS_R - 1 - w;   // the window point's place among the group's
  const uint32_t* pts = (is_a ? niels_a : niels_r) + g * (size_t)RLC_G * 32;
  const int8_t* drow = dig + (g * RLC_WINDOWS + (is_a ? 0 : RLC_WINDOWS_A)) * (size_t)RLC_G;
  const int l = (int)threadIdx.x;
  static_assert(RLC_SEG_WINDOWS == 1 && RLC_BUCKETS == 2 * RLC_LANES, "one window per wave, two buckets per lane");

  // counting sort of the window's digits by magnitude; for the bookkeeping lane l owns buckets l + 1 and l + 65
  s.hist[l + 1] = 0;
  s.hist[l + 1 + RLC_LANES] = 0;
  if (l == 0) s.hist[0] = 0;
  __syncthreads();
  // this lane's 128 digits of the window: items 16 (64 j + l) .. + 15, j < 8, one 16-byte load per j
  // (rows are whole: k_rlc_scalars zeroes the digits of the slots past the end of the batch)
  const uint4* dw = reinterpret_cast<const uint4*>(drow + (size_t)w * RLC_G);
  const int bd = (is_a && l == 0) ? (int)bdig[32 * g + w] : 0;               // lane 0 also files the base point
#pragma unroll 1
  for (int j = 0; j < RLC_G / (16 * RLC_LANES); j++) {
    const uint4 v = dw[j * RLC_LANES + l];
    const uint32_t dd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int d = (int)(int8_t)(dd[t >> 2] >> (8 * (t & 3)));
      if (d != 0) atomicAdd(&s.hist[d < 0 ? -d : d], 1u);
    }
  }
  if (bd != 0) atomicAdd(&s.hist[bd < 0 ? -bd : bd], 1u);
  __syncthreads();
  {
    uint32_t start0 = 0, start1 = 0, rank0 = 0, rank1 = 0;
    const uint32_t own0 = s.hist[l + 1], own1 = s.hist[l + 1 + RLC_LANES];
    for (int q = 1; q <= RLC_BUCKETS; q++) {     // prefix sums, and the two buckets' ranks by size (fullest first)
      const uint32_t c = s.hist[q];
      if (q <= l) start0 += c;
      if (q <= l + RLC_LANES) start1 += c;
      rank0 += (c > own0 || (c == own0 && q < l + 1)) ? 1u : 0u;
      rank1 += (c > own1 || (c == own1 && q < l + 1 + RLC_LANES)) ? 1u : 0u;
    }
    s.cursor[l + 1] = start0;              s.first[l + 1] = start0;              s.perm[rank0] = (uint32_t)l;
    s.cursor[l + 1 + RLC_LANES] = start1;  s.first[l + 1 + RLC_LANES] = start1;  s.perm[rank1] = (uint32_t)(l + RLC_LANES);
  }
  __syncthreads();
#pragma unroll 1
  for (int j = 0; j < RLC_G / (16 * RLC_LANES); j++) {
    const uint4 v = dw[j * RLC_LANES + l];
    const uint32_t dd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int d = (int)(int8_t)(dd[t >> 2] >> (8 * (t & 3)));
      const int k = 16 * (j * RLC_LANES + l) + t;
      if (d != 0) s.list[atomicAdd(&s.cursor[d < 0 ? -d : d], 1u)] = (uint16_t)(k | (d < 0 ? 0x8000 : 0));
    }
  }
  if (bd != 0) s.list[atomicAdd(&s.cursor[bd < 0 ? -bd : bd], 1u)] = (uint16_t)(RLC_BASE_IDX | (bd < 0 ? 0x8000 : 0));
  __syncthreads();
  // Lane l adds up the bucket whose size has rank l and then the bucket of rank 127 - l: a wave takes as long as its
  // busiest lane, a bucket holds Poisson(64) points (the fullest of 128 about 86), and the fullest plus the emptiest,
  // the second fullest plus the second emptiest ... all come to about twice the mean: the busiest lane has 130.5
  // additions where the mean is 127.5.  (Round 5 gave the 64 fullest buckets to one wave of a two-wave block and the
  // 64 emptiest to the other; the block met at a barrier, so both waves paid for the fullest bucket: 2 x 86.)
  const int mb1 = (int)s.perm[l], mb2 = (int)s.perm[RLC_BUCKETS - 1 - l];
  const uint32_t lo1 = s.first[mb1 + 1], cnt1 = s.hist[mb1 + 1];
  const uint32_t lo2 = s.first[mb2 + 1], total = cnt1 + s.hist[mb2 + 1];
  ge acc, acc1;
  ge_neutral(acc);
  ge_neutral(acc1);
#pragma unroll 1
  for (uint32_t q = 0; q < total; q++) {
    const uint32_t e = s.list[q < cnt1 ? lo1 + q : lo2 + (q - cnt1)], idx = e & 0x7fffu;
    ge_niels nl;
    niels_load(nl, idx == RLC_BASE_IDX ? base16 + TABLE_ENTRY_WORDS : pts + 32 * (size_t)idx);
    ge_niels_cneg(nl, (e & 0x8000u) != 0);
    ge_add_niels(acc, acc, nl, true);
    if (q + 1 == cnt1) { acc1 = acc; ge_neutral(acc); }     // the first bucket is done (some 18 distinct counts per wave)
  }
  // (Fetching the next entry's line during the current addition - two register sets taken in turn, so that the loads
  // really are in flight - measured the same: the SIMD's other wave covers the wait.)
  __syncthreads();
  // Back to bucket order, two buckets per lane again: lane j takes the sums of buckets 2 j + 1 and 2 j + 2 (sa, sb; a
  // bucket's number is its digit's magnitude = its weight) - in two rounds, the odd-numbered buckets and then the
  // even-numbered ones, so that the exchange area is 64 points (10 KB) and not 128
  ge sa, sb;
  if ((mb1 & 1) == 0) lds_put_at(s.pts, acc1, mb1 >> 1);
  if ((mb2 & 1) == 0) lds_put_at(s.pts, acc, mb2 >> 1);
  __syncthreads();
  lds_get(sa, s.pts, l);
  __syncthreads();
  if ((mb1 & 1) != 0) lds_put_at(s.pts, acc1, mb1 >> 1);
  if ((mb2 & 1) != 0) lds_put_at(s.pts, acc, mb2 >> 1);
  __syncthreads();
  lds_get(sb, s.pts, l);
  __syncthreads();
  // sum over buckets of weight * sum = sum_j (2 j + 1) sa_j + (2 j + 2) sb_j = 2 sum_j (j + 1) u_j - sum_j sa_j with
  // u_j = sa_j + sb_j; sum_j (j + 1) u_j is the sum of the suffix sums t_j = u_j + u_(j+1) + ...  Fifteen additions deep
  // for 128 buckets: one for u, six for the scan, six for the two tree sums (of the t_j in lanes 0..31 and of the sa_j
  // in lanes 32..63, side by side), a doubling and the last addition.
  ge_add_full(acc, sa, sb);                      // u
#pragma unroll 1
  for (int stride = 1; stride < RLC_LANES; stride <<= 1) {                   // inclusive suffix scan
    lds_put(s.pts, acc);
    __syncthreads();
    if (l + stride < RLC_LANES) {
      ge o;
      lds_get(o, s.pts, l + stride);
      ge_add_full(acc, acc, o);
    }
    __syncthreads();
  }
  {                                              // first step of both trees: lanes 0..31 t_l + t_(l+32), lanes 32..63 sa_(l-32) + sa_l
    ge o, o2;
    lds_put(s.pts, acc);
    __syncthreads();
    lds_get(o, s.pts, l ^ 32);
    __syncthreads();
    lds_put(s.pts, sa);
    __syncthreads();
    lds_get(o2, s.pts, l ^ 32);
    __syncthreads();
    ge_cmov(acc, sa, l >= 32);
    ge_cmov(o, o2, l >= 32);
    ge_add_full(acc, acc, o);
  }
#pragma unroll 1
  for (int stride = RLC_LANES / 4; stride >= 1; stride >>= 1) {
    lds_put(s.pts, acc);
    __syncthreads();
    if ((l & 31) < stride) {
      ge o;
      lds_get(o, s.pts, l + stride);
      ge_add_full(acc, acc, o);
    }
    __syncthreads();
  }
  lds_put(s.pts, acc);
  __syncthreads();
  if (l == 0) {                                  // the window's weight 2^(8 w) is applied by the Horner wave
    ge o;
    lds_get(o, s.pts, 32);                       // sum of the sa_j: subtracted
    fe_neg(o.X, o.X); fe_carry(o.X);
    fe_neg(o.T, o.T); fe_carry(o.T);
    ge_dbl(acc, acc, true);
    ge_add_full(acc, acc, o);
    ge_cached c;                                 // stored in cached form (Y-X | Y+X | 2dT | 2Z, packed)
    ge_to_cached(c, acc);
    uint32_t* dst = segpts + (g * RLC_SEGS + seg) * VERIFY_ENTRY_WORDS;
    const fe* f[4] = {&c.ymx, &c.ypx, &c.t2d, &c.z2};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t pw[8];
      fe_pack(pw, *f[k]);
      coherent_store8(dst + 8 * k, pw);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the point is in memory before its flag is
    // (test hook: the flag of window point *hook - 1 of group 0 is never raised; the word is 0 unless a test set it)
    if (!(g == 0 && *hook == (uint32_t)seg + 1u))
      __hip_atomic_store(gflags + RLC_GROUP_WORDS * g + 1 + seg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// R7: verdicts of the items of accepted groups
__global__ void __launch_bounds__(RLC_BLOCK)
k_rlc_verdicts(size_t n, const uint8_t* gok, const uint8_t* flags, uint8_t* ok) {
  const size_t i = (size_t)blockIdx.x * RLC_BLOCK + threadIdx.x;
  if (i >= n) return;
  if (gok[i / RLC_G]) ok[i] = flags[i] & RLC_R_VALID;
}

}  // namespace ed

using namespace ed;

namespace ed {
__global__ void k_rlc_note_per_item(uint32_t* stats, uint32_t items, uint32_t groups) {
  atomicAdd(stats + 1, items);
  atomicAdd(stats + 2, groups);
}
}  // namespace ed

// a pass that the caller routes to the per-item kernels without trying the combination: only the statistics
extern "C" hipError_t edk_rlc_note_per_item(uint32_t* stats, size_t n, hipStream_t stream) {
  if (stats && n) hipLaunchKernelGGL(ed::k_rlc_note_per_item, dim3(1), dim3(1), 0, stream, stats, (uint32_t)n,
                                     (uint32_t)((n + ed::RLC_G - 1) / ed::RLC_G));
  return hipGetLastError();
}

extern "C" size_t edk_rlc_ws_bytes(size_t capacity) { return capacity ? rlc_carve(capacity).total : 0; }
extern "C" size_t edk_rlc_hook_offset(size_t capacity) { return rlc_carve(capacity).hook; }

// First half of a pass: everything the combination itself needs, and the copy of the group verdicts to the
// host (pinned).  The caller synchronises `stream` and then calls edk_verify_rlc_fallback.
extern "C" hipError_t edk_verify_rlc(uint8_t* ok, uint32_t* stats, const edk_verify_src* srcp, size_t n,
                                     const uint32_t* base16, const edk_verify_ws* ws, const edk_rlc_ws* rws,
                                     hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const edk_verify_src src = *srcp;
  const rlc_layout L = rlc_carve(rws->capacity);
  uint8_t* base = static_cast<uint8_t*>(rws->base);
  uint32_t* ts = reinterpret_cast<uint32_t*>(base + L.ts);
  uint32_t* leaf = reinterpret_cast<uint32_t*>(base + L.leaf);
  uint32_t* niels_a = reinterpret_cast<uint32_t*>(base + L.niels_a);
  uint32_t* niels_r = reinterpret_cast<uint32_t*>(base + L.niels_r);
  int8_t* dig = reinterpret_cast<int8_t*>(base + L.dig);
  uint8_t* flags = base + L.flags;
  uint32_t* bsum = reinterpret_cast<uint32_t*>(base + L.bsum);
  int8_t* bdig = reinterpret_cast<int8_t*>(base + L.bdig);
  uint32_t* gflags = reinterpret_cast<uint32_t*>(base + L.gflags);
  uint8_t* gok = base + L.gok;
  uint32_t* segpts = reinterpret_cast<uint32_t*>(base + L.seg);
  uint32_t* tree = reinterpret_cast<uint32_t*>(base + L.tree);
  const size_t groups = (n + RLC_G - 1) / RLC_G;
  const unsigned blocks = (unsigned)((n + RLC_BLOCK - 1) / RLC_BLOCK);
  hipError_t e;

  EDK_DO(hipMemsetAsync(gflags, 0, groups * RLC_GROUP_WORDS * 4, stream));
  const uint32_t* perm = nullptr;
  EDK_DO(edk_msg_order(&perm, ws, src.msg_off, src.msg_end, n, stream));
  EDK_LAUNCH(k_rlc_hash, dim3(blocks), dim3(RLC_BLOCK), 0, stream, src, n, ts, leaf, perm);
  // the batch seed: a SHA-512 tree of fan-in 64 over the leaves, on the side stream beside k_rlc_points
  EDK_DO(hipEventRecord(ws->ev_prepared, stream));
  EDK_DO(hipStreamWaitEvent(ws->side, ws->ev_prepared, 0));
  const uint32_t* level = leaf;
  uint32_t* bufs[2] = {tree, tree + 8 * (rws->capacity / RLC_TREE_FAN + 2)};
  size_t count = n;
  int flip = 0;
  do {
    const size_t next = (count + RLC_TREE_FAN - 1) / RLC_TREE_FAN;
    EDK_LAUNCH(k_rlc_tree, dim3((unsigned)((next + 63) / 64)), dim3(64), 0, ws->side, level, bufs[flip], count);
    level = bufs[flip];
    flip ^= 1;
    count = next;
  } while (count > 1);
  const uint32_t* seed = level;
  EDK_DO(hipEventRecord(ws->ev_exact, ws->side));
  EDK_LAUNCH(k_rlc_points, dim3(blocks), dim3(RLC_BLOCK), 0, stream, src, n, niels_a, niels_r, flags, gflags);
  EDK_DO(hipStreamWaitEvent(stream, ws->ev_exact, 0));
  EDK_LAUNCH(k_rlc_scalars, dim3((unsigned)(groups * (RLC_G / RLC_BLOCK))), dim3(RLC_BLOCK), 0, stream, n, ts, seed, flags, dig, bsum);
  EDK_LAUNCH(k_rlc_group_scalar, dim3((unsigned)((groups + 63) / 64)), dim3(64), 0, stream, n, bsum, bdig);
  EDK_DO(hipEventRecord(ws->ev_prepared, stream));
  EDK_LAUNCH(k_rlc_bucket, dim3((unsigned)(groups * RLC_SEGS)), dim3(RLC_LANES), 0, stream, n, dig, bdig, niels_a, niels_r, base16, segpts, gflags,
             reinterpret_cast<const uint32_t*>(base + L.hook));
  EDK_DO(hipStreamWaitEvent(ws->side, ws->ev_prepared, 0));    // (queued after the bucket launch: see k_rlc_horner)
  EDK_LAUNCH(k_rlc_horner, dim3((unsigned)((groups + RLC_HORNER_GROUPS - 1) / RLC_HORNER_GROUPS)), dim3(4 * RLC_HORNER_GROUPS), 0, ws->side,
             n, segpts, gflags, gok, stats, reinterpret_cast<uint32_t*>(base + L.hook) + 1);
  EDK_DO(hipEventRecord(ws->ev_exact, ws->side));
  EDK_DO(hipStreamWaitEvent(stream, ws->ev_exact, 0));
  EDK_LAUNCH(k_rlc_final, dim3((unsigned)((groups + RLC_HORNER_GROUPS - 1) / RLC_HORNER_GROUPS)), dim3(4 * RLC_HORNER_GROUPS), 0, stream,
             n, segpts, gflags, gok, stats);
  EDK_LAUNCH(k_rlc_verdicts, dim3(blocks), dim3(RLC_BLOCK), 0, stream, n, gok, flags, ok);

  // groups the combination did not accept: the per-item kernels decide (edk_verify_rlc_fallback).  This is the
  // one place where the host looks at a result: the caller synchronises the stream once per pass.
  EDK_DO(hipMemcpyAsync(rws->host_gok, gok, groups, hipMemcpyDeviceToHost, stream));
  return hipSuccess;
}

// Second half, once the stream has been synchronised: runs of groups that did not pass go to the per-item kernels
extern "C" hipError_t edk_verify_rlc_fallback(uint8_t* ok, const edk_verify_src* srcp, size_t n, const uint32_t* base16,
                                              const edk_verify_ws* ws, const edk_rlc_ws* rws, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const edk_verify_src src = *srcp;
  const size_t groups = (n + RLC_G - 1) / RLC_G;
  const uint8_t* h_gok = static_cast<const uint8_t*>(rws->host_gok);
  hipError_t e;
  for (size_t g = 0; g < groups;) {
    if (h_gok[g]) { g++; continue; }
    size_t g1 = g;
    while (g1 < groups && !h_gok[g1]) g1++;
    const size_t lo = g * RLC_G, hi = g1 * RLC_G < n ? g1 * RLC_G : n;
    edk_verify_src sub = src;
    sub.sigs += lo * src.sig_stride;
    sub.pubs += lo * src.pub_stride;
    if (src.msg_off) sub.msg_off += lo; else sub.msgs += lo * src.msg_stride;
    if ((e = edk_verify(ok + lo, &sub, hi - lo, base16, ws, nullptr, nullptr, 0, stream)) != hipSuccess) return e;
    g = g1;
  }
  return hipSuccess;
}
