// kernels.hip - the batched Ed25519 / X25519 kernels for gfx950 and their launchers.
//
// One curve operation per lane, limbs in registers (fe25519.h), no MFMA.  What a lane computes is
// in lanes.h; this file maps items to lanes, stages the shared tables in LDS and moves the packed,
// item-major byte arrays of the batch API (include/eddsa_amd.h) in and out.
//
//   k_x25519_*       x25519.c:129-150 do_x25519                      (config 3)
//   k_verify_*       ed25519-sha512.c:148-181 ed25519_verify        (config 2, 4): prepare / halve / main_half by
//                    default (half-length scalars, halve.h), prepare / main / finish for the full-length route,
//                    *_pair / *_quad forms for small passes; for keys that are not curve points k_verify_exact_quad (short
//                    work lists, four lanes per item) and k_verify_exact_lane_setup / _chain (long ones, two items per lane)
//   k_sign_*         ed25519-sha512.c:84-123 sign                    (config 5)
//   k_genpub_point + k_encode_finish          ed25519-sha512.c:53-67 genpub
//   k_x25519_base_*  x25519.c:158-197 do_x25519_base
//   k_pk_to_x        ed25519-sha512.c:187-232 pk_ed25519_to_x25519
//   k_sk_to_x        ed25519-sha512.c:239-256 sk_ed25519_to_x25519
//   k_init_tables    generates what the reference ships as lib/ed_lookup64.h
#include "eddsa_kernels.h"
#include "edk_checked.h"
#include "kernel_io.h"
#include "lanes.h"
#include "quad_lanes.h"

#include <atomic>

namespace ed {

constexpr int BLOCK = 256;

// one verify item: R, S, A as words and the message span (packed arrays or fixed-size records)
ED_DEV void verify_item(uint32_t rw[8], uint32_t sw[8], uint32_t aw[8], const uint8_t*& m, size_t& mlen,
                        const edk_verify_src& s, size_t item) {
  load32(rw, s.sigs, item, s.sig_stride);
  load32(sw, s.sigs + 32, item, s.sig_stride);
  load32(aw, s.pubs, item, s.pub_stride);
  msg_span(m, mlen, s.msgs, s.msg_off, s.msg_end, s.msg_len, s.msg_stride, item);
}

// Append to a work list: the lanes of the wave that `want` a slot get consecutive ones from ONE atomic (a pass of 2^20
// items would otherwise send as many atomics to one address).  Callable under divergence: the ballot counts the active
// lanes only, and the leader is one of them.
ED_DEV uint32_t wave_append(uint32_t* counter, bool want) {
  const uint64_t m = __ballot(want);
  if (m == 0) return 0;
  const unsigned lane = __lane_id();
  const int leader = __ffsll((unsigned long long)m) - 1;
  uint32_t base = 0;
  if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, leader);
  return base + (uint32_t)__popcll(m & (((uint64_t)1 << lane) - 1));
}

// ---------------------------------------------------------------------------------------------

// X25519 in two kernels, like the fixed-base operations: the ladder leaves (x2 : z2) in the point
// workspace (X and Z slots), the finish kernel inverts z2 once per eight items per lane
// (finish_batch8 below) -- the per-item inversion was 7 % of the work.
__global__ void __launch_bounds__(BLOCK, 4)
k_x25519_ladder(uint32_t* accout, const uint8_t* scalars, const uint8_t* points, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t item = i < n ? i : n - 1;
  uint32_t s[8], pt[8];
  load32(s, scalars, item, 32);
  load32(pt, points, item, 32);
  fe x2, z2;
  x25519_ladder_lane(x2, z2, s, pt);
  if (i >= n) return;                            // an idle lane must not leave a copy of the last item's secret
  uint32_t* o = accout + (size_t)blockIdx.x * (ACC_WORDS * BLOCK) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 10; j++) { o[j * BLOCK] = x2.v[j]; o[(20 + j) * BLOCK] = z2.v[j]; }
}

// small passes: four lanes per item (quad_lanes.h: x25519_ladder_quad); writes the same workspace slots
constexpr size_t X25519_QUAD_MAX_N = (size_t)1 << 14;
__global__ void __launch_bounds__(BLOCK, 2)
k_x25519_ladder_quad(uint32_t* accout, const uint8_t* scalars, const uint8_t* points, size_t n) {
  const size_t i = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 2;          // quads are all-or-nothing
  if (i >= n) return;
  const int q = (int)(threadIdx.x & 3u);
  uint32_t s[8], pt[8];
  load32(s, scalars, i, 32);
  load32(pt, points, i, 32);
  fe r;
  x25519_ladder_quad(r, s, pt, q);
  if (q > 1) return;                             // lane 0: x2 -> the X slot, lane 1: z2 -> the Z slot
  uint32_t* o = accout + (i / BLOCK) * (ACC_WORDS * BLOCK) + (i % BLOCK) + (q == 0 ? 0 : 20) * BLOCK;
#pragma unroll
  for (int j = 0; j < 10; j++) o[j * BLOCK] = r.v[j];
}

__global__ void __launch_bounds__(64) k_init_tables(uint32_t* base16, uint32_t* comb) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  if (id >= 2 * TABLE_BASE16_ENTRIES + TABLE_COMB_ENTRIES) return;
  if (id < TABLE_BASE16_ENTRIES) {
    table_entry_lane(base16 + (size_t)TABLE_ENTRY_WORDS * id, (uint32_t)id, 0);
  } else if (id < 2 * TABLE_BASE16_ENTRIES) {     // k * 2^128 * B: the high half of s' in the half-length verification
    table_entry_lane(base16 + (size_t)TABLE_ENTRY_WORDS * id, (uint32_t)(id - TABLE_BASE16_ENTRIES), 128);
  } else {
    const int c = id - 2 * TABLE_BASE16_ENTRIES;  // comb[i][k], c = COMB_HALF i + k
    table_entry_lane(comb + TABLE_ENTRY_WORDS * c, (uint32_t)(c % COMB_HALF) + 1, 2u * COMB_W * (uint32_t)(c / COMB_HALF));
  }
}

// the LDS image of the comb (lanes.h: comb_select), one thread per entry
__global__ void __launch_bounds__(64) k_init_comb_image(uint32_t* img, const uint32_t* comb) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  if (id >= COMB_ROWS * COMB_IMG_ENTRIES) return;
  comb_image_entry_lane(img + COMB_IMG_ENTRY_WORDS * id, comb, id / COMB_IMG_ENTRIES, id % COMB_IMG_ENTRIES);
}

// ---------------------------------------------------------------------------------------------
// Ragged messages in order of length.  A lane hashes its message block by block and a wave ends with its longest message:
// with lengths drawn from 0 .. 4 KiB a pass of 2^20 items spent a third of its hashing beside finished lanes (verify 15.7 ->
// 12.4 ms, sign 13.3 -> 8.6 ms when the SAME items are handed over sorted: profiles/r05_msglen.txt).  So the kernels that hash
// take their items through a permutation - position g of the grid works on item perm[g] - built per pass by a counting sort on
// the number of 128-byte blocks, longest first (three small kernels, ~20 us); everything an item leaves in the workspace stays
// indexed by the item, so no other kernel knows.  Passes of at least MSG_ORDER_MIN_N items with an offset table only.
// ---------------------------------------------------------------------------------------------
constexpr int LEN_BINS = EDK_LEN_BINS;           // key = min(length / 128, LEN_BINS - 1), counted from the long end
constexpr size_t MSG_ORDER_MIN_N = (size_t)1 << 12;

ED_DEV uint32_t len_bin(const uint64_t* off, const uint64_t* end, size_t item) {
  uint64_t lo, hi;
  ragged_span(lo, hi, off, end, item);                         // (the clamped span: what the hashing kernels will read)
  const size_t blocks = (size_t)(hi - lo) >> 7;
  return (uint32_t)(LEN_BINS - 1) - (uint32_t)(blocks < (size_t)LEN_BINS - 1 ? blocks : (size_t)LEN_BINS - 1);
}

// bins[b] += items of bin b (per block in LDS first: a pass whose messages all have one length would otherwise send 2^20 atomics to one word)
__global__ void __launch_bounds__(BLOCK) k_len_count(uint32_t* bins, const uint64_t* off, const uint64_t* end, size_t n) {
  __shared__ uint32_t h[LEN_BINS];
  for (int b = threadIdx.x; b < LEN_BINS; b += BLOCK) h[b] = 0;
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) atomicAdd(&h[len_bin(off, end, i)], 1u);
  __syncthreads();
  for (int b = threadIdx.x; b < LEN_BINS; b += BLOCK) if (h[b]) atomicAdd(bins + b, h[b]);
}
// bins[LEN_BINS + b] = items in the bins before b (one block)
__global__ void __launch_bounds__(BLOCK) k_len_starts(uint32_t* bins) {
  __shared__ uint32_t part[BLOCK];
  constexpr int PER = LEN_BINS / BLOCK;
  uint32_t local[PER], sum = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) { local[k] = sum; sum += bins[threadIdx.x * PER + k]; }
  part[threadIdx.x] = sum;
  __syncthreads();
  uint32_t before = 0;
  for (int t = 0; t < (int)threadIdx.x; t++) before += part[t];
#pragma unroll
  for (int k = 0; k < PER; k++) bins[LEN_BINS + threadIdx.x * PER + k] = before + local[k];
}
// perm[start of the item's bin ...] = item: a block reserves its share of every bin with one atomic per bin
__global__ void __launch_bounds__(BLOCK) k_len_place(uint32_t* perm, uint32_t* bins, const uint64_t* off, const uint64_t* end, size_t n) {
  __shared__ uint32_t h[LEN_BINS], base[LEN_BINS];
  for (int b = threadIdx.x; b < LEN_BINS; b += BLOCK) h[b] = 0;
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  uint32_t bin = 0, rank = 0;
  if (i < n) { bin = len_bin(off, end, i); rank = atomicAdd(&h[bin], 1u); }
  __syncthreads();
  for (int b = threadIdx.x; b < LEN_BINS; b += BLOCK) if (h[b]) base[b] = atomicAdd(bins + LEN_BINS + b, h[b]);
  __syncthreads();
  if (i < n) perm[base[bin] + rank] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------------
// Ed25519 verify, the full-length route (passes of up to QUAD_MAIN_MAX_N items, eddsa_amd_set_verify_algo(1), the
// reject mode; larger passes take the half-length route further down, which shares k_verify_prepare and the exact
// path).  Three kernels per chunk, so that each stays inside its register budget and its own I-cache footprint
// (one fused kernel spilled 4.5 KB/lane):
//   k_verify_prepare  hash, scalars -> digit words, decompress -A, table of 0..8 * -A
//   k_verify_main     the 252 doublings + 80 additions            (~85 % of the time)
//   k_verify_finish   invert Z (shared by 8 items per lane), encode, compare with R
//   k_verify_exact_quad (side stream, beside main) the reference's own chain for the items whose key is not a curve
//                     point; owns their verdict bytes
// Workspace (HBM; tile = 256 items):
//   digits [item][16]                  t + 0x88.., S + 0x80.. as little-endian words
//   table  [item][entry 9][word 32]    1152 contiguous bytes per item; an entry = ymx | ypx | t2d | z2 packed into 255 bits each
//                                      (fe_pack), one 128-byte line
//   acc    [tile][word 30][lane 256]   X, Y, Z of the result
//   flags  [item]                      bit 0: A decoded to a curve point; bit 1: Z usable (set by finish)
//   offlist[..], offcount              items whose A is off the curve, for k_verify_exact_quad
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BLOCK, 2)
k_verify_prepare(edk_verify_src src, size_t n, uint32_t* digits,
                 uint32_t* table, uint8_t* flags, uint32_t* onlist, uint32_t* offlist, uint32_t* offcount, int all_exact, const uint32_t* perm) {
  const size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  // position g of the grid works on item perm[g] (ragged messages in order of length, above) or on item g
  const size_t item = perm ? perm[g < n ? g : n - 1] : (g < n ? g : n - 1);
  const size_t i = g < n ? item : g;             // its slot of the workspace: the item's own; idle lanes redo the last item into a slot past the pass
  uint32_t rw[8], aw[8], sw[8], tw[8];
  const uint8_t* m; size_t mlen;
  load32(rw, src.sigs, item, src.sig_stride);
  load32(aw, src.pubs, item, src.pub_stride);
  msg_span(m, mlen, src.msgs, src.msg_off, src.msg_end, src.msg_len, src.msg_stride, item);
  uint4* d = reinterpret_cast<uint4*>(digits + 16 * i);
  verify_hash_lane(tw, rw, aw, m, mlen);
  d[0] = make_uint4(tw[0], tw[1], tw[2], tw[3]); d[1] = make_uint4(tw[4], tw[5], tw[6], tw[7]);
  load32(sw, src.sigs + 32, item, src.sig_stride);       // S is fetched only now: nothing to hold across the hash
  verify_s_lane(sw);
  d[2] = make_uint4(sw[0], sw[1], sw[2], sw[3]); d[3] = make_uint4(sw[4], sw[5], sw[6], sw[7]);
  const bool oncurve = verify_table_lane(table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS), aw);
  // keys that are not curve points go to the exact (reference-order) kernels: append to their work
  // list (all_exact, a self-check mode: every item does, and the windowed result is not used)
  const bool windowed = oncurve && !all_exact;
  flags[i] = (uint8_t)windowed;
  // Two work lists: the items the windowed evaluation decides (the half-length route's k_verify_halve and
  // k_verify_main_half run over THIS list, not over the pass: a caller who sends nothing but garbage keys then pays for
  // the exact path only, not for a windowed evaluation whose result is discarded on top), and the exact path's.
  const bool live = g < n;
  const uint32_t on_slot = wave_append(offcount + EDK_ONLIST_WORD, live && windowed);
  if (live && windowed) onlist[on_slot] = (uint32_t)i;
  const uint32_t off_slot = wave_append(offcount, live && !windowed);
  if (live && !windowed) offlist[off_slot] = (uint32_t)i;
}

// The exact path: ed25519-sha512.c:148-181 replayed in the reference's own order for the items k_verify_prepare (and, on
// the half-length route, k_verify_halve) listed, with four lanes per item (quad_lanes.h), so that a chain step is a
// squaring and a multiplication deep.  Set-up (joint sparse form of the two scalars and the addends Q, B, Q+B, Q-B, all
// from what k_verify_prepare left in the workspace) and chain are one kernel on the side stream beside k_verify_main,
// which is launched with MAIN_LDS_RESERVE and therefore leaves wave slots, registers and a little LDS free on every CU.
// Why this shape (tools/exact_path_time.py, tools/exact_trace.py timelines, profiles/r02_verify_ab.txt):
// k_verify_main's grid for 2^20 items is an exact number of rounds of resident blocks, so it has no slack:
// whatever holds up ONE of its blocks -- a displaced tile (round 1: chain blocks of four waves at 149 VGPRs
// took a main block's place), or main waves starved by a long chain wave of raised priority on their SIMD --
// costs the latency of one tile at the end (0.7 ms), for 512 off-curve keys as for 65536.  The remedy is to
// disturb every SIMD a little instead of a few a lot: single-wave chain blocks (16 items) spread over the
// whole chip, the short four-lane chain, no displacement.  Cost of the exact path on config 2: 0.94 ms in
// round 1, 0.4 ms in round 2, 0.05 ms now (profiles/r04_small_grid.txt has what it costs passes of one or two rounds).
// The work list may be as long as the pass (a caller can send nothing but garbage keys: ed_import never fails, ed.c:100-149).
// This kernel serves SHORT lists - below EXACT_LANE_MIN_LISTED entries in passes of EXACT_LANE_MIN_N items or more, every list of
// a smaller pass (at most its 65 535 items: EDK_EXACT_SLOTS scratchpad slots, 4096 waves of 16) - and ends at once when the list
// is longer: that one belongs to k_verify_exact_lane_* below, one lane per item at the occupancy of a kernel that fills the
// chip.  (Rounds 1-4 walked long lists with this kernel in stretches of 65 536: 34 M items/s, and a pass of 2^20 random keys
// 26 ms; now 14: profiles/r05_exact_lane.txt.)
constexpr int QUAD_BLOCK = 256;                  // k_verify_main_quad
constexpr int QUAD_CHAIN_BLOCK = 64;             // k_verify_exact_quad: one wave, up to 16 items
constexpr int QUAD_CHAIN_ITEMS = QUAD_CHAIN_BLOCK / 4;
constexpr int QUAD_SPREAD_WAVES = 256;           // a short work list is spread over this many waves, a long one packed 16 items to the wave
constexpr size_t EXACT_DENSE_MIN_N = (size_t)1 << 16;   // passes from this size on pack the chain's items 16 to the wave whatever their number
constexpr size_t EXACT_LANE_MIN_LISTED = (size_t)1 << 13;   // (the threshold's table: profiles/r05_exact_lane.txt, 5.) work lists from this length on go to k_verify_exact_lane_* ...
constexpr size_t EXACT_LANE_MIN_N = (size_t)1 << 16;      // ... in passes of at least this many items
static_assert((size_t)EDK_EXACT_SLOTS * QUAD_ITEM_WORDS * 4 <= EDK_EXACT_PAD_BYTES, "scratchpad too small");
static_assert(EDK_EXACT_SLOTS % QUAD_CHAIN_ITEMS == 0, "whole waves");

// Set-up and chain in ONE kernel (round 2 had two, 0.37-0.75 + 0.75 ms however few the items: a pass of 2^14 items
// with a handful of off-curve keys took 1.02 ms instead of 0.39; now 0.79).  A wave carries 16 items when the list is
// long - beside a full k_verify_main_half that is what costs it least - and as few as one when it is short (up to
// QUAD_SPREAD_WAVES waves: a small pass waits for this kernel, and a wave with one item skips the addition of every step
// without digits, quad_lanes.h; spreading 540 items of a 2^16-item pass one to the wave was measured slower than 34
// full waves - 1.28 against 1.18 ms - their instructions are taken from the main kernel's waves).  The digit pairs
// live in LDS (33 words per item), the addends in the HBM scratchpad, one slot per quad of the grid.
__global__ void __launch_bounds__(QUAD_CHAIN_BLOCK, 2)
k_verify_exact_quad(uint8_t* ok, const uint8_t* sigs, size_t sig_stride, const uint32_t* digits, const uint32_t* table,
                    const uint32_t* offlist, const uint32_t* offcount, const uint32_t* base16, uint32_t* pad, int dense,
                    uint32_t lane_min) {
  __shared__ uint32_t lds_dig[QUAD_CHAIN_ITEMS * QUAD_DIGIT_WORDS];
  __builtin_amdgcn_s_setprio(3);                 // small passes wait for the chain: 1-3 % there; no difference beside a full k_verify_main
  const size_t listed = *offcount;
  if (listed == 0) return;
  if (lane_min != 0 && listed >= lane_min) return;   // a list this long belongs to the one-lane kernels (k_verify_exact_lane_*)
  const size_t count = listed < (size_t)EDK_EXACT_SLOTS ? listed : (size_t)EDK_EXACT_SLOTS;   // (never more: see the launcher's static_assert)
  // items per wave: 16 in a pass whose main kernel outlasts the chain anyway (dense: the chain's total work is what counts
  // there - every wave issues the chain's 300 k instructions however few items it carries, and in the host pipeline, where
  // the chunks' kernels fill one another's gaps, 557 items spread three to the wave cost a 2^16-item chunk a quarter of its
  // main kernel's work on top); in a small pass, which WAITS for the chain, as few as one (its wave then skips the
  // addition of every step without digits)
  size_t per = dense ? (size_t)QUAD_CHAIN_ITEMS : (count + QUAD_SPREAD_WAVES - 1) / QUAD_SPREAD_WAVES;
  per = per > (size_t)QUAD_CHAIN_ITEMS ? (size_t)QUAD_CHAIN_ITEMS : per;
  const size_t quad = threadIdx.x >> 2;          // quads are all-or-nothing
  const size_t g = (size_t)blockIdx.x * per + quad;
  if (quad >= per || g >= count) return;
  const int q = (int)(threadIdx.x & 3u);
  const size_t i = offlist[g];
  uint32_t* item = pad + g * QUAD_ITEM_WORDS;
  uint32_t* dig = lds_dig + quad * QUAD_DIGIT_WORDS;
  verify_exact_setup_quad(digits + 16 * i, table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS), base16 + TABLE_ENTRY_WORDS, item, dig, q);
  __syncthreads();                               // one wave: orders the quad's stores (LDS digits, HBM addends) before the other lanes' loads
  uint32_t rw[8];
  load32(rw, sigs, i, sig_stride);
  const bool same = verify_exact_chain_quad(rw, item, dig, q);
  if (q == 1) ok[i] = (uint8_t)same;
}

// The exact path for LONG work lists: lanes.h: verify_exact_setup_table_lane, one lane per item, then exact_pair_iterations,
// TWO items per lane sharing one addition phase - two kernels of persistent blocks on the side stream, ahead of the four-lane
// launch.  Which form serves a pass is decided on the device, by the length of the list: below `min_listed` entries these
// kernels end at once and k_verify_exact_quad does the work (its single-wave blocks answer soonest and disturb a full main
// kernel least); from there on it is the other way round - a wave of the four-lane chain carries 16 items through 300 k
// instructions (1.2 M lane slots per item), a wave of this one 128 through 1.2 M (594 k per item), and a caller who sends
// nothing but garbage keys (47 % of random strings are no curve point) is served at the rate of a kernel that fills the chip:
// profiles/r05_exact_lane.txt.
// The set-up writes into the item's OWN workspace - Q + B and Q - B over entries 2 and 3 of its table, the digit string
// over the first words of its rtable slot - which nobody reads for such an item: the windowed kernels skip it (on the
// half-length route they do not even visit it, see k_verify_prepare) or discard what they compute from it.
// Work is handed out to WAVES, one (stretch, tile) unit at a time from a counter: a tile is 128 entries of the list (two per lane),
// a stretch EXACT_PAIR_ITERS iterations of the walk (lanes.h: exact_pair_iterations; the last one runs to the end), the accumulators cross from stretch to stretch through the items' workspace and
// a wave that draws stretch s of a tile waits for the tile's stretch s - 1 (tile_done; units are drawn in the order stretch 0 of
// every tile, stretch 1 of every tile, ..., so the wait is over before it starts unless the list is shorter than the chip).
// Why not one chain per lane per launch slot: beside k_verify_main_half half of these blocks become resident late, and with a
// fixed share of 4 chains of 2.5 ms each per lane the kernel's tail was 3 ms of a 17 ms pass (profiles/r05_exact_lane.txt).
constexpr uint32_t EXACT_CHAIN_COST = 594;       // executed instructions per item of k_verify_exact_lane_chain, thousands
constexpr unsigned EXACT_LANE_BLOCKS = 512;      // two resident blocks per CU
constexpr int EXACT_DIGITS_AT = 0, EXACT_STATE_AT = 64;   // words of the item's rtable slot: the digit string; the accumulator between stretches (lines of its own)
__global__ void __launch_bounds__(BLOCK, 2)
k_verify_exact_lane_setup(const uint32_t* digits, uint32_t* table, uint32_t* rtable, const uint32_t* offlist, uint32_t* offcount,
                          const uint32_t* base16, uint32_t* tile_done, uint32_t min_listed) {
  const size_t listed = *offcount;
  if (listed < min_listed) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) exact_bentry_store(offcount + EDK_BENTRY_WORD);   // (the same bytes in every pass)
  for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < listed; g += (size_t)gridDim.x * BLOCK) {
    if ((g & 63) == 0) tile_done[g >> 6] = 0;
    const size_t i = offlist[g];
    verify_exact_setup_table_lane(table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS),
                                  rtable + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS) + EXACT_DIGITS_AT, 1, digits + 16 * i, base16 + TABLE_ENTRY_WORDS);
  }
}

__global__ void __launch_bounds__(BLOCK, 2)
k_verify_exact_lane_chain(uint8_t* ok, const uint8_t* sigs, size_t sig_stride, const uint32_t* table, uint32_t* rtable,
                          const uint32_t* offlist, uint32_t* offcount, uint32_t* tile_done, uint32_t min_listed, size_t n,
                          uint32_t main_cost, int main_all, uint32_t* status) {
  const size_t listed = *offcount;
  if (listed < min_listed) return;
  // a tile is 128 entries of the list: a lane walks TWO chains, entries L and 64 + L of the tile (lanes.h: exact_pair_iterations)
  const unsigned tiles = (unsigned)((listed + 127) / 128), units = tiles * (unsigned)EXACT_SEGS;
  const unsigned lane = threadIdx.x & 63u;
  // no more waves than tiles: one that drew a later stretch of a tile still in its first would only hold, waiting, a slot
  // that a wave of k_verify_main_half could use
  unsigned waves = tiles;
  // ... and no more than this kernel's SHARE of the wave slots: its persistent waves keep what they take until the list is
  // done, the main kernel beside it fills what is left, and whichever ends first leaves the other to its own tail.  With the
  // slots split like the work (main_cost: the main kernel's instructions per item in thousands, over the on-curve list or -
  // main_all - over the whole pass) the two end together: a pass with every second key random 13.7 -> 12.8 ms, all random unchanged
  // (profiles/r05_exact_lane.txt).
  {
    const uint64_t mine = (uint64_t)listed * EXACT_CHAIN_COST;
    const uint64_t other = (uint64_t)(main_all ? n : (size_t)offcount[EDK_ONLIST_WORD]) * main_cost;
    const uint64_t slots = (uint64_t)gridDim.x * (BLOCK / 64);
    const uint64_t share = (slots * mine + (mine + other) - 1) / (mine + other);
    // (never fewer than 512, or than there are tiles: a short list is done soonest with a wave per tile, long before the main kernel)
    const uint64_t least = tiles < 512u ? tiles : 512u;
    // (and only when the main kernel has the larger part: a chain that is most of the pass is soonest done with every slot it can get
    // - all keys random, a share of 0.70: 14.3 ms with all slots, 14.7 with its share)
    if (share * 5 < slots * 3 && share < waves) waves = (unsigned)(share < least ? least : share);
  }
  if ((blockIdx.x * (unsigned)BLOCK + threadIdx.x) / 64u >= waves) return;
  for (;;) {
    unsigned u = 0;
    if (lane == 0)
      u = __hip_atomic_load(offcount + EDK_STALL_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? units
                                                                                                        : atomicAdd(offcount + EDK_EXACT_UNIT_WORD, 1u);
    u = (unsigned)__builtin_amdgcn_readfirstlane((int)u);
    if (u >= units) break;
    const unsigned seg = u / tiles, tile = u - seg * tiles;
    if (seg > 0) {
      // the tile's previous stretch, possibly on another CU: its stores are in L2 once its count is; the acquire drops this
      // CU's stale lines.
      // Why the wait ends: units are drawn in order from ONE counter - stretch s - 1 of every tile before stretch s of any -
      // so the unit waited for was drawn earlier, by a wave that is resident (a wave draws only while it runs) and that
      // itself waits, if at all, for a unit drawn earlier still: the oldest unit in flight waits for nothing.
      // Why it is bounded all the same: if that wave never publishes (a fault in it; the test hook below), spinning for
      // ever would turn one lost hand-off into a hung queue.  After EDK_EXACT_PATIENCE polls - seconds, where a stretch
      // takes a millisecond - the wave raises EDK_STALL_WORD and leaves; every wave looks at that word before it draws or
      // while it waits, so the launch drains, and the host side returns EDDSA_AMD_STALLED for the pass.
      uint32_t stalled = 0;
      if (lane == 0) {
        uint32_t polls = 0;
        while (__hip_atomic_load(tile_done + tile, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seg) {
          if (++polls == EDK_EXACT_PATIENCE || __hip_atomic_load(offcount + EDK_STALL_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            stalled = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(16);
        }
        if (stalled) {
          __hip_atomic_store(offcount + EDK_STALL_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (status) __hip_atomic_store(status, EDK_STATUS_STALLED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      if (__builtin_amdgcn_readfirstlane((int)stalled)) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const size_t ga = (size_t)tile * 128 + lane, gb = ga + 64;
    const bool live_a = ga < listed, live_b = gb < listed;
    const uint32_t ia = offlist[live_a ? ga : listed - 1], ib = offlist[live_b ? gb : listed - 1];   // (an idle half walks nothing and stores nothing)
    constexpr uint32_t SLOT = VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS;
    ge ra, rb;
    exact_walk wa, wb;
    if (seg == 0) {
      ge_neutral(ra); ge_neutral(rb);
      wa = exact_walk_start(rtable + (size_t)ia * SLOT + EXACT_DIGITS_AT, 1, live_a);
      wb = exact_walk_start(rtable + (size_t)ib * SLOT + EXACT_DIGITS_AT, 1, live_b);
    } else {
      exact_walk_load(ra, wa, rtable + (size_t)ia * SLOT + EXACT_STATE_AT);
      exact_walk_load(rb, wb, rtable + (size_t)ib * SLOT + EXACT_STATE_AT);
      if (!live_a) { wa.i = -1; wa.pend = false; }
      if (!live_b) { wb.i = -1; wb.pend = false; }
    }
    const bool last = seg == (unsigned)EXACT_SEGS - 1;
    exact_pair_iterations(ra, wa, rb, wb, table, rtable + EXACT_DIGITS_AT, SLOT, ia, ib, offcount + EDK_BENTRY_WORD, last ? -1 : EXACT_PAIR_ITERS);
    if (last) {
      uint32_t rwa[8], rwb[8];
      load32(rwa, sigs, ia, sig_stride);
      load32(rwb, sigs, ib, sig_stride);
      bool same_a, same_b;
      exact_pair_verdicts(same_a, same_b, ra, rb, rwa, rwb);
      if (live_a) ok[ia] = (uint8_t)same_a;
      if (live_b) ok[ib] = (uint8_t)same_b;
    } else {
      if (live_a) exact_walk_store(rtable + (size_t)ia * SLOT + EXACT_STATE_AT, ra, wa);
      if (live_b) exact_walk_store(rtable + (size_t)ib * SLOT + EXACT_STATE_AT, rb, wb);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // every lane's accumulators are out before the count says so
      // (test hook: the first hand-off of tile EDK_WITHHOLD_WORD - 1 is never published; the word is 0 unless a test set it)
      if (lane == 0 && !(seg == 0 && offcount[EDK_WITHHOLD_WORD] == tile + 1u))
        __hip_atomic_fetch_add(tile_done + tile, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void __launch_bounds__(BLOCK, 4)
k_verify_main(const uint32_t* digits, const uint32_t* table, const uint32_t* base16, uint32_t* accout) {
  const size_t tile = blockIdx.x;
  const size_t i = tile * BLOCK + threadIdx.x;   // < workspace capacity
  ge acc;
  verify_main_lane(acc, digits + 16 * i, table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS), base16);
  uint32_t* o = accout + tile * (ACC_WORDS * BLOCK) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 10; j++) {
    o[j * BLOCK] = acc.X.v[j]; o[(10 + j) * BLOCK] = acc.Y.v[j]; o[(20 + j) * BLOCK] = acc.Z.v[j];
  }
}

// ---------------------------------------------------------------------------------------------
// Half-length verification (halve.h), one lane per item: the route of passes above HALF_QUAD_MAX_N items (smaller
// ones: k_verify_prepare_pair + k_verify_main_half_quad further down):
//   k_verify_prepare    as above
//   k_verify_halve      (u, v) with v = u t mod 8l from t; s' = |u| S; decompress R strictly, table of 0..8 * -R'
//   k_verify_main_half  132 doublings + 68 + 16 additions, neutral-element test, verdict byte
// No finish kernel: "is the neutral element" needs no inversion.  Items the pair search gives up on join the
// off-curve keys on the exact path's work list (so the list is complete only after k_verify_halve).
// Workspace beside the one above: hdigits [item][HALF_DIGIT_WORDS], rtable [item][entry 9][word 32] (the same packed form);
// flags bit 0: this path owns the verdict, bit 2: R is a canonical encoding of a curve point.
// ---------------------------------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(BLOCK, 2)
k_verify_halve(const uint8_t* sigs, size_t sig_stride, const uint32_t* digits, uint32_t* hdigits,
               uint32_t* rtable, uint8_t* flags, const uint32_t* onlist, uint32_t* offlist, uint32_t* offcount) {
  // over the list of items the windowed evaluation decides (k_verify_prepare), not over the pass: the grid is sized for
  // the pass, the blocks beyond the list end at once
  const size_t slot = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (slot >= (size_t)offcount[EDK_ONLIST_WORD]) return;
  const size_t i = onlist[slot], item = i;
  const uint8_t fl = flags[i];
  uint32_t tdig[8], sdig[8], hd[HALF_DIGIT_WORDS];
  {
    const uint4* d = reinterpret_cast<const uint4*>(digits + 16 * i);
    const uint4 a = d[0], b = d[1], c = d[2], e = d[3];
    tdig[0] = a.x; tdig[1] = a.y; tdig[2] = a.z; tdig[3] = a.w; tdig[4] = b.x; tdig[5] = b.y; tdig[6] = b.z; tdig[7] = b.w;
    sdig[0] = c.x; sdig[1] = c.y; sdig[2] = c.z; sdig[3] = c.w; sdig[4] = e.x; sdig[5] = e.y; sdig[6] = e.z; sdig[7] = e.w;
  }
  verify_half_scalars_lane<BITS>(hd, tdig, sdig);
  if ((hd[24] & 4u) != 0) atomicAdd(offcount + EDK_REFUSED_WORD, 1u);   // a pair the exact check refused: never seen (diagnostic)
  uint4* o = reinterpret_cast<uint4*>(hdigits + HALF_DIGIT_WORDS * i);
#pragma unroll
  for (int q = 0; q < HALF_DIGIT_WORDS / 4; q++) o[q] = make_uint4(hd[4 * q], hd[4 * q + 1], hd[4 * q + 2], hd[4 * q + 3]);
  uint32_t rw[8];
  load32(rw, sigs, item, sig_stride);
  const bool rvalid = verify_half_point_lane(rtable + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS), rw);
  // An item without a short pair ("long", lanes.h) joins the exact path's work list here: k_verify_main_half's grid
  // has no slack, and both alternatives measured slower at 2^20 items - the long loop inside that kernel for the
  // waves that contain such an item (one in 180: 7.2 -> 7.8 ms) and a separate four-lane kernel for them beside it
  // (7.4 -> 7.65 ms).  Small passes, whose waves have the chip to themselves, do run the long loop in place
  // (k_verify_main_half_quad).
  const bool mine = (fl & 1) != 0 && (hd[24] & 2u) == 0;
  flags[i] = (uint8_t)((mine ? 1 : 0) | (rvalid ? 4 : 0));
  if ((fl & 1) != 0 && !mine) offlist[atomicAdd(offcount, 1u)] = (uint32_t)i;
}

// dynamic LDS the main kernels are launched with: what keeps a third block off the CU, and where k_verify_main_half keeps its digit words
constexpr unsigned MAIN_LDS_RESERVE = 74 * 1024;   // (2 x 74 of the CU's 160 KB: a third block does not fit, four waves of k_verify_exact_quad do)
// (measured and dropped: three blocks of 128 lanes with 50 KB each, 1.5 waves per SIMD - a resident set of 226 MB, which fits the
// 256 MB Infinity Cache: profiles/r03_verify_ab.txt)
constexpr int MAIN_HALF_BLOCK = BLOCK, MAIN_HALF_BLOCKS_PER_CU = 2;
// WITH_LONG: after k_verify_prepare_pair (mid-size passes), which leaves the items without a short pair to this kernel:
// the wave of such an item (2 in 10^7 with the wide search) runs the long loop
template <int WINDOWS, bool WITH_LONG = false>
__global__ void __launch_bounds__(MAIN_HALF_BLOCK, MAIN_HALF_BLOCKS_PER_CU)
k_verify_main_half(uint8_t* ok, const uint32_t* hdigits, const uint32_t* table, const uint32_t* rtable,
                   const uint32_t* base16, const uint8_t* flags, const uint32_t* onlist, const uint32_t* offcount) {
  // over the list of items the windowed evaluation decides (k_verify_prepare / k_verify_prepare_pair), like k_verify_halve
  const size_t slot = (size_t)blockIdx.x * MAIN_HALF_BLOCK + threadIdx.x;
  if (slot >= (size_t)offcount[EDK_ONLIST_WORD]) return;
  const size_t i = onlist[slot];
  const uint32_t* hd = hdigits + HALF_DIGIT_WORDS * i;
  // The lane's digit words go to LDS once (word w of lane t at [w][t]: no bank conflicts).  Read from memory window by
  // window they were the hottest lines of L2 - one per resident item, 16 MB of its 32 - and L2 is what the table lines
  // need: its hits are worth 12 % to this kernel (profiles/r04_main_half_loads_ab.txt), and with the digits out of it the
  // pass gained 1.5 %.  (The launch reserves MAIN_LDS_RESERVE bytes per block anyway; this uses 28 KB of them.)
  extern __shared__ uint32_t digit_words[];
  {
    const uint4* g = reinterpret_cast<const uint4*>(hd);
#pragma unroll
    for (int q = 0; q < HALF_DIGIT_WORDS / 4; q++) {
      const uint4 v = g[q];
      digit_words[(4 * q) * MAIN_HALF_BLOCK + threadIdx.x] = v.x; digit_words[(4 * q + 1) * MAIN_HALF_BLOCK + threadIdx.x] = v.y;
      digit_words[(4 * q + 2) * MAIN_HALF_BLOCK + threadIdx.x] = v.z; digit_words[(4 * q + 3) * MAIN_HALF_BLOCK + threadIdx.x] = v.w;
    }
  }
  const uint32_t* hl = digit_words + threadIdx.x;
  const bool long_loop = WITH_LONG && __any((hl[24 * MAIN_HALF_BLOCK] & 2u) != 0);
  // a zero digit reads item 0's entry 0 - the neutral element, like every item's own: one line for the chip, not one per item
  const bool neutral = verify_half_main_lane<WITH_LONG, WINDOWS>(hl, table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS),
                                             rtable + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS), base16, long_loop, MAIN_HALF_BLOCK, table);
  const uint8_t fl = flags[i];
  if ((fl & 1) == 0) return;                     // k_verify_halve handed the item to the exact path, which owns its verdict
  ok[i] = (uint8_t)(neutral && (fl & 4) != 0);
}

// ---------------------------------------------------------------------------------------------
// "finish" kernels: everything that needs an inversion (ed_export ed.c:161, x25519.c:192) shares
// ONE inversion between FINISH_K items per lane (Montgomery's trick: 254 S + 11 M once, plus three
// multiplications per item).  Lane t of block b handles lane t of tiles b*K .. b*K+K-1 of the
// lane-interleaved point workspace acc[tile][30][256], so every access stays coalesced.
// A policy P supplies  den(k, z, good): the value to invert for item k (1 when there is nothing to
// invert: item past the end, rejected key, or a zero denominator, which must not poison the shared
// product) and  item(k, zinv, good): the rest of the work of item k.
// ---------------------------------------------------------------------------------------------
constexpr int FINISH_K = 8;
// k_verify_main uses no LDS; launching it with 74 KB (78 until round 3) of dynamic LDS per block limits it to two blocks (eight
// waves) per CU.  Measured (profiles/r02_verify_ab.txt): the kernel itself is about 1 % FASTER that way (two
// waves per SIMD already saturate VALU issue; fewer resident tables), and the exact path's waves fit beside
// it on every CU without taking the place of any of its blocks.

struct finish_pos {
  size_t tile, i;            // tile index and global item index of slot k for this lane
  const uint32_t* acc;       // this lane's column of the tile: X at [j*BLOCK], Y at [(10+j)*BLOCK], Z at [(20+j)*BLOCK]
};
ED_DEV finish_pos finish_at(int k, const uint32_t* accin, int K) {
  finish_pos p;
  p.tile = (size_t)blockIdx.x * K + k;
  p.i = p.tile * BLOCK + threadIdx.x;
  p.acc = accin + p.tile * (ACC_WORDS * BLOCK) + threadIdx.x;
  return p;
}
ED_DEV void acc_load(fe& f, const uint32_t* acc, int coord) {
#pragma unroll
  for (int j = 0; j < 10; j++) f.v[j] = acc[(10 * coord + j) * BLOCK];
}
// i = the lane's global item slot (tile i / BLOCK, lane i % BLOCK), whatever the block size
ED_DEV void acc_store(uint32_t* accout, size_t i, const ge& p) {
  uint32_t* o = accout + (i / BLOCK) * (ACC_WORDS * BLOCK) + (i % BLOCK);
#pragma unroll
  for (int j = 0; j < 10; j++) {
    o[j * BLOCK] = p.X.v[j]; o[(10 + j) * BLOCK] = p.Y.v[j]; o[(20 + j) * BLOCK] = p.Z.v[j];
  }
}

// Phase A multiplies the eight denominators together, inverts once and unwinds.  Each denominator is
// written to its Z slot by den(), each prefix product z0 ... zk to the fourth slot (W) of item k, and
// both are re-read during the unwinding: with the eight prefix products in registers beside the
// inversion's temporaries every finish kernel needed more than 256 VGPRs and spilled 660 bytes per
// lane.  The same lane writes and later reads these slots.  Phase B is an ordinary loop over the
// items, so the (large) per-item code exists once.
ED_DEV void slot_store(uint32_t* acc, int k, int coord, const fe& f, int K) {
  uint32_t* o = acc + ((size_t)blockIdx.x * K + k) * (ACC_WORDS * BLOCK) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 10; j++) o[(10 * coord + j) * BLOCK] = f.v[j];
}
ED_DEV void slot_load(fe& f, const uint32_t* acc, int k, int coord, int K) {
  const uint32_t* o = acc + ((size_t)blockIdx.x * K + k) * (ACC_WORDS * BLOCK) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 10; j++) f.v[j] = o[(10 * coord + j) * BLOCK];
}
ED_DEV void zinv_store(uint32_t* acc, int k, const fe& zi, int K) { slot_store(acc, k, 2, zi, K); }

template <class P>
ED_DEV void finish_batch8(const P& pol, uint32_t* acc) {
  const int K = pol.K;                           // items per lane that share the inversion: FINISH_K, fewer in small passes
  fe z, p, u, zi;
  pol.den(0, p);
  slot_store(acc, 0, 3, p, K);
#pragma unroll 1
  for (int k = 1; k < K; k++) {
    pol.den(k, z);
    fe_mul(p, p, z);
    slot_store(acc, k, 3, p, K);                 // z0 ... zk
  }
  fe_inv(u, p);                                  // u = 1 / (z0 ... z(K-1))
#pragma unroll 1
  for (int k = K - 1; k >= 1; k--) {
    slot_load(p, acc, k - 1, 3, K);
    fe_mul(zi, u, p);                            // 1 / zk
    slot_load(z, acc, k, 2, K);
    fe_mul(u, u, z);                             // 1 / (z0 ... z(k-1))
    zinv_store(acc, k, zi, K);
  }
  zinv_store(acc, 0, u, K);
#pragma unroll 1
  for (int k = 0; k < K; k++) pol.item(k);
}

// z := good ? z : 1, committed to slot k's Z so that the unwinding re-reads exactly this value
ED_DEV void den_commit(fe& z, bool good, uint32_t* acc, int k, int K) {
  fe one;
  fe_set(one, 1);
  fe_cmov(one, z, good);
  z = one;
  zinv_store(acc, k, z, K);
}

// verify: encode and compare with R as bytes (ed25519-sha512.c:176-180): a non-canonical R can
// never match.  Items whose A is off the curve are skipped here (k_verify_exact_quad writes their
// verdict; DESIGN.md "Off-curve public keys"); Z = 0 cannot occur for a curve point (the a = -1 law
// is complete) and is rejected defensively.
struct verify_finish_policy {
  uint8_t* ok; const uint8_t* sigs; size_t sig_stride; uint32_t* acc; uint8_t* flags; size_t n; int exact_offcurve; int K;
  ED_DEV void den(int k, fe& z) const {
    const finish_pos p = finish_at(k, acc, K);
    fe_set(z, 1);
    bool good = false;
    if (p.i < n) {
      acc_load(z, p.acc, 2);
      const uint8_t fl = flags[p.i];
      good = (fl & 1) != 0 && !fe_iszero(z);
      flags[p.i] = (uint8_t)((fl & 1) | (good ? 2 : 0));   // phase B reads it back
    }
    den_commit(z, good, acc, k, K);
  }
  ED_DEV void item(int k) const {
    const finish_pos p = finish_at(k, acc, K);
    if (p.i >= n) return;
    fe x, y, zinv;
    acc_load(x, p.acc, 0); acc_load(y, p.acc, 1); acc_load(zinv, p.acc, 2);
    uint32_t rw[8];
    load32(rw, sigs, p.i, sig_stride);
    const uint8_t fl = flags[p.i];
    if ((fl & 1) == 0) {                         // off-curve key
      if (!exact_offcurve) ok[p.i] = 0;          // reject mode; otherwise k_verify_exact_quad owns ok[i]
      return;
    }
    ok[p.i] = (uint8_t)(verify_encode_lane(x, y, zinv, rw) && (fl & 2) != 0);
  }
};

// small passes: four lanes per item (quad_lanes.h: verify_main_quad); writes the same workspace
constexpr size_t QUAD_MAIN_MAX_N = (size_t)1 << 14;   // measured: 0.57 vs 0.86 ms at 2^14, equal at 2^15 (tools/verify_sizes.py)
// the upper bound of k_verify_prepare_pair + k_verify_main_half_quad when the mid-size arrangement below is switched off
// (algo 0 takes that one above PAIR_ONE_MIN_N items, so the four-lane evaluation serves passes of up to 24 576 items)
constexpr size_t HALF_QUAD_MAX_N = (size_t)1 << 15;
// Between 24 576 and 2^18 items: k_verify_prepare_pair, then the ONE-lane evaluation with the long loop in place.  Measured
// (tools/verify_mid.py, valid signatures, ms): 2^15 0.68 (four-lane evaluation) / 0.70 (one lane per item throughout) -> 0.58,
// 2^16 0.71-0.76 -> 0.66, 2^17 1.30 -> 1.28; the config-2 mix, whose floor is the exact path: 2^16 1.16 -> 1.11, else equal.
constexpr size_t PAIR_ONE_MIN_N = (size_t)3 << 13;   // the four-lane evaluation steps up with every 8192 items (0.35 / 0.51 / 0.69 ms: tools/verify_cross.py), this one stays at 0.58
constexpr size_t QUAD_WIDE_MIN_N = 256;               // four-lane passes above this search pairs up to 2^138 as well (see edk_verify)
constexpr size_t PAIR_ONE_MAX_N = (size_t)1 << 18;    // the mid-size arrangement (three-lane preparation, one-lane evaluation with the long loop in place) ends here
constexpr size_t HALF_WIDE_MIN_N = (size_t)1 << 19;   // one-lane passes below this search pairs up to 2^138 (see edk_verify)
__global__ void __launch_bounds__(QUAD_BLOCK, 2)
k_verify_main_quad(const uint32_t* digits, const uint32_t* table, const uint32_t* base16, uint32_t* accout, size_t n) {
  const size_t i = ((size_t)blockIdx.x * QUAD_BLOCK + threadIdx.x) >> 2;       // quads are all-or-nothing
  if (i >= n) return;
  const int q = (int)(threadIdx.x & 3u);
  fe r;
  verify_main_quad(r, digits + 16 * i, table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS), base16, q);
  if (q == 2) return;                            // T is not part of the result
  uint32_t* o = accout + (i / BLOCK) * (ACC_WORDS * BLOCK) + (i % BLOCK) + (q == 0 ? 0 : q == 1 ? 10 : 20) * BLOCK;
#pragma unroll
  for (int j = 0; j < 10; j++) o[j * BLOCK] = r.v[j];
}

__global__ void __launch_bounds__(BLOCK, 2)
k_verify_finish(uint8_t* ok, const uint8_t* sigs, size_t sig_stride, uint32_t* acc, uint8_t* flags, size_t n,
                int exact_offcurve, int K) {
  finish_batch8(verify_finish_policy{ok, sigs, sig_stride, acc, flags, n, exact_offcurve, K}, acc);
}

// x25519.c:144-149: x2 / z2, and 0 when z2 = 0 (fld_inv(0) = 0): such an item contributes 1 to the
// shared product and its "inverse" is forced to 0
struct x25519_finish_policy {
  uint8_t* out; uint32_t* acc; size_t n; int K;
  ED_DEV void den(int k, fe& z) const {
    const finish_pos p = finish_at(k, acc, K);
    fe_set(z, 1);
    bool good = false;
    if (p.i < n) {
      acc_load(z, p.acc, 2);
      good = !fe_iszero(z);
      if (!good) {                               // remember it: X := 0 makes the product 0 whatever the "inverse"
        uint32_t* o = acc + p.tile * (ACC_WORDS * BLOCK) + threadIdx.x;
#pragma unroll
        for (int j = 0; j < 10; j++) o[j * BLOCK] = 0;
      }
    }
    den_commit(z, good, acc, k, K);
  }
  ED_DEV void item(int k) const {
    const finish_pos p = finish_at(k, acc, K);
    fe x, zinv;
    if (p.i < n) { acc_load(x, p.acc, 0); acc_load(zinv, p.acc, 2); }
    // (x2 : z2), 1/z2 and the prefix products of the shared inversion determine shared secrets: they
    // do not outlive the call in HBM (x25519.c:221 burnstack); slots past the end hold the committed 1
    uint32_t* o = acc + p.tile * (ACC_WORDS * BLOCK) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 10; j++) { o[j * BLOCK] = 0; o[(20 + j) * BLOCK] = 0; o[(30 + j) * BLOCK] = 0; }
    if (p.i >= n) return;
    uint32_t w[8];
    x25519_finish_lane(w, x, zinv);
    store32(out, p.i, 32, w);
  }
};

__global__ void __launch_bounds__(BLOCK, 2)
k_x25519_finish(uint8_t* out, uint32_t* acc, size_t n, int K) {
  finish_batch8(x25519_finish_policy{out, acc, n, K}, acc);
}

// ---------------------------------------------------------------------------------------------
// fixed-base kernels: the image of the comb (22 rows x 32 entries m * 4096^i * B, m = 1..32, 99 KiB of the CU's
// 160 KiB LDS; one 512-lane block per CU, two waves per SIMD) is staged by every block of a
// "point" kernel; the matching "finish" kernel encodes (and, for sign, hashes and computes S)
// ---------------------------------------------------------------------------------------------

constexpr int POINT_BLOCK = COMB_IMG_WORDS * 4 > 80 * 1024 ? 512 : 256;   // 8 waves per CU either way (768 lanes: 6-101 spilled registers, 1024: 94-314)

// PARTS = 4: four lanes per item for small passes (lanes.h: scale_base_lane<4>) - a block is four waves, wave w holds
// share w of 64 items - and the shares are added through LDS, pairwise: waves 1 and 3 hand theirs to waves 0 and 2,
// then wave 2 to wave 0, which delivers.  The shares are secrets (partial sums of a secret multiple): the slots are
// zeroed once read.
constexpr int POINT_SPLIT = 4, POINT_SPLIT_ITEMS = 64;
// The blocks of a pass that fills the chip are PERSISTENT - one block per CU stages the 99 KB image once - and their WAVES take
// 64-item tiles from a counter until none is left (point_tile below).  Why not a tile per block, or a fixed share per wave:
// the image leaves room for ONE block of eight waves per CU, two waves per SIMD, and the SIMD issues its older wave first; with
// equal shares the older wave of every pair finished at ~60 % of the kernel's time and the younger one ran the rest alone, at
// the issue rate of a single wave (SQ_WAVE_CYCLES: 1.52 waves resident per SIMD on average, VALU-busy 0.92 - the "idle 10 %" of
// VERDICT r03 #8; not the staging, not LDS latency, not the instruction cache: profiles/r04_sign_ab.txt).  Handed out one by
// one, the tiles go to whichever wave is ahead and all waves finish together.  Small passes (no more blocks than CUs, or four
// lanes per item) keep the fixed mapping: block b takes tile b.
constexpr unsigned POINT_MAX_BLOCKS = 256u * (512 / POINT_BLOCK);   // as many as are resident at once

// the next item of this lane, or false when the pass is done.  `it` = the lane's iteration count so far (0 at the start).
// Dynamic hand-out: lane 0 of the wave draws a tile number, the wave takes items 64 t .. 64 t + 63, until the counter is past
// the end (the launcher zeroes it on the pass's stream before every such launch).
template <int PARTS>
ED_DEV bool point_tile(size_t& i, unsigned it, size_t n, uint32_t* tiles) {
  if (PARTS == 1 && tiles != nullptr) {
    unsigned t = 0;
    if ((threadIdx.x & 63u) == 0) t = atomicAdd(tiles, 1u);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    if ((size_t)t * 64 < n) { i = (size_t)t * 64 + (threadIdx.x & 63u); return true; }
    return false;
  }
  const size_t span = PARTS == 1 ? (size_t)blockDim.x : (size_t)POINT_SPLIT_ITEMS;
  const size_t base = ((size_t)blockIdx.x + (size_t)it * gridDim.x) * span;
  i = base + (PARTS == 1 ? threadIdx.x : (threadIdx.x & 63u));
  return base < n;
}

ED_DEV void share_store(uint32_t* slot, const ge& p) {
#pragma unroll
  for (int j = 0; j < 10; j++) {
    slot[j * 64] = p.X.v[j]; slot[(10 + j) * 64] = p.Y.v[j]; slot[(20 + j) * 64] = p.Z.v[j]; slot[(30 + j) * 64] = p.T.v[j];
  }
}
ED_DEV void share_take(ge& p, uint32_t* slot) {
#pragma unroll
  for (int j = 0; j < 10; j++) {
    p.X.v[j] = slot[j * 64]; p.Y.v[j] = slot[(10 + j) * 64]; p.Z.v[j] = slot[(20 + j) * 64]; p.T.v[j] = slot[(30 + j) * 64];
    slot[j * 64] = 0; slot[(10 + j) * 64] = 0; slot[(20 + j) * 64] = 0; slot[(30 + j) * 64] = 0;
  }
}
// every wave of the block calls it; afterwards the lanes of wave 0 hold the sums (Z, Y, X; T is not computed)
ED_DEV void point_reduce4(ge& a, uint32_t* lds_shares, int part) {
  uint32_t* slot = lds_shares + (part >> 1) * (40 * 64) + (threadIdx.x & 63u);
  ge o;
  ge_cached c;
  if (part & 1) share_store(slot, a);
  __syncthreads();
  if (!(part & 1)) { share_take(o, slot); ge_to_cached(c, o); ge_add_cached(a, a, c, true); }
  __syncthreads();
  slot = lds_shares + (threadIdx.x & 63u);
  if (part == 2) share_store(slot, a);
  __syncthreads();
  if (part == 0) { share_take(o, slot); ge_to_cached(c, o); ge_add_cached(a, a, c, false); }
}

template <int PARTS>
__global__ void __launch_bounds__(POINT_BLOCK, 512 / POINT_BLOCK)
k_genpub_point(uint32_t* accout, const uint8_t* secs, size_t n, const uint32_t* comb, uint32_t* tiles) {
  __shared__ alignas(16) uint32_t lds_comb[COMB_IMG_WORDS];
  stage_table(lds_comb, comb, COMB_IMG_WORDS);
  __shared__ uint32_t lds_shares[PARTS == 1 ? 1 : 2 * 40 * 64];
  // blocks of POINT_BLOCK lanes, or smaller ones for small passes (EDK_POINT_GRID); PARTS = 4: four waves, 64 items
  const int part = PARTS == 1 ? 0 : (int)(threadIdx.x >> 6);
  size_t i;
#pragma unroll 1
  for (unsigned it = 0; point_tile<PARTS>(i, it, n, tiles); it++) {
    uint32_t sk[8];
    load32(sk, secs, i < n ? i : n - 1, 32);
    ge A;
    genpub_point_lane<PARTS>(A, sk, lds_comb, part);
    if (PARTS > 1) point_reduce4(A, lds_shares, part);
    if (PARTS == 1 || (part == 0 && i < n)) acc_store(accout, i, A);
  }
}

// Z of a comb result is never 0 (B and its multiples are curve points)
struct encode_finish_policy {
  uint8_t* out; uint32_t* acc; size_t n; int K;
  ED_DEV void den(int k, fe& z) const {
    const finish_pos p = finish_at(k, acc, K);
    fe_set(z, 1);
    if (p.i < n) acc_load(z, p.acc, 2);
    den_commit(z, true, acc, k, K);
  }
  ED_DEV void item(int k) const {
    const finish_pos p = finish_at(k, acc, K);
    if (p.i >= n) return;
    fe x, y, zinv;
    acc_load(x, p.acc, 0); acc_load(y, p.acc, 1); acc_load(zinv, p.acc, 2);
    uint32_t w[8];
    encode_lane(w, x, y, zinv);
    store32(out, p.i, 32, w);
  }
};

__global__ void __launch_bounds__(BLOCK, 2)
k_encode_finish(uint8_t* out, uint32_t* acc, size_t n, int K) {
  finish_batch8(encode_finish_policy{out, acc, n, K}, acc);
}

template <int PARTS>
__global__ void __launch_bounds__(POINT_BLOCK, 512 / POINT_BLOCK)
k_sign_point(uint32_t* accout, uint32_t* aux, const uint8_t* secs, const uint8_t* msgs,
             const uint64_t* msg_off, const uint64_t* msg_end, size_t msg_len, size_t n, const uint32_t* comb, const uint32_t* perm,
             uint32_t* tiles) {
  __shared__ alignas(16) uint32_t lds_comb[COMB_IMG_WORDS];
  stage_table(lds_comb, comb, COMB_IMG_WORDS);
  __shared__ uint32_t lds_shares[PARTS == 1 ? 1 : 2 * 40 * 64];
  // blocks of POINT_BLOCK lanes, or smaller ones for small passes (EDK_POINT_GRID); PARTS = 4: four waves, 64 items
  const int part = PARTS == 1 ? 0 : (int)(threadIdx.x >> 6);
  size_t i;
#pragma unroll 1
  for (unsigned it = 0; point_tile<PARTS>(i, it, n, tiles); it++) {
    // position i works on item perm[i] (ragged messages in order of length); the point and the secret scalars stay at the
    // POSITION's slots of the workspace, where k_sign_finish finds them
    const size_t item = perm ? perm[i < n ? i : n - 1] : (i < n ? i : n - 1);
    const uint8_t* m; size_t mlen;
    msg_span(m, mlen, msgs, msg_off, msg_end, msg_len, msg_len, item);
    uint32_t sk[8], aw[8], rw[8];
    load32(sk, secs, item, 32);
    sign_scalars_lane(aw, rw, sk, m, mlen);
    // the secret scalars a and r, for the finish step: handed over BEFORE the comb (sixteen registers less to hold across
    // its 44 additions, which is what lets a row's two lookups be issued ahead of its two additions without a spill).
    // An idle lane must not leave a copy of the last item's secrets.
    if (i < n && part == 0) {
      uint4* d = reinterpret_cast<uint4*>(aux + 16 * i);
      d[0] = make_uint4(aw[0], aw[1], aw[2], aw[3]); d[1] = make_uint4(aw[4], aw[5], aw[6], aw[7]);
      d[2] = make_uint4(rw[0], rw[1], rw[2], rw[3]); d[3] = make_uint4(rw[4], rw[5], rw[6], rw[7]);
    }
    ge R;
    scale_base_lane<PARTS>(R, rw, lds_comb, part);
    if (PARTS > 1) point_reduce4(R, lds_shares, part);
    if (PARTS == 1 || (part == 0 && i < n)) acc_store(accout, i, R);
  }
}

struct sign_finish_policy {
  uint8_t* sigs; uint32_t* acc; uint32_t* aux; const uint8_t* pubs; const uint8_t* msgs;
  const uint64_t* msg_off; const uint64_t* msg_end; size_t msg_len; size_t n; int K; const uint32_t* perm;
  ED_DEV void den(int k, fe& z) const {
    const finish_pos p = finish_at(k, acc, K);
    fe_set(z, 1);
    if (p.i < n) acc_load(z, p.acc, 2);
    den_commit(z, true, acc, k, K);
  }
  ED_DEV void item(int k) const {
    const finish_pos p = finish_at(k, acc, K);
    if (p.i >= n) return;
    const size_t it = perm ? perm[p.i] : p.i;      // the item this position carries (k_sign_point)
    fe x, y, zinv;
    acc_load(x, p.acc, 0); acc_load(y, p.acc, 1); acc_load(zinv, p.acc, 2);
    uint32_t Rw[8], Sw[8], pub[8];
    encode_lane(Rw, x, y, zinv);
    load32(pub, pubs, it, 32);
    const uint8_t* m; size_t mlen;
    msg_span(m, mlen, msgs, msg_off, msg_end, msg_len, msg_len, it);
    sc t;
    sign_challenge_lane(t, Rw, pub, m, mlen);              // the secret scalars are fetched only after the hash
    uint32_t aw[8], rw[8];
    uint4* d = reinterpret_cast<uint4*>(aux + 16 * p.i);
    const uint4 a0 = d[0], a1 = d[1], r0 = d[2], r1 = d[3];
    aw[0] = a0.x; aw[1] = a0.y; aw[2] = a0.z; aw[3] = a0.w; aw[4] = a1.x; aw[5] = a1.y; aw[6] = a1.z; aw[7] = a1.w;
    rw[0] = r0.x; rw[1] = r0.y; rw[2] = r0.z; rw[3] = r0.w; rw[4] = r1.x; rw[5] = r1.y; rw[6] = r1.z; rw[7] = r1.w;
    const uint4 zero = make_uint4(0, 0, 0, 0);           // the secrets do not outlive the call in HBM
    d[0] = zero; d[1] = zero; d[2] = zero; d[3] = zero;
    sign_response_lane(Sw, t, aw, rw);
    store32(sigs, it, 64, Rw);
    store32(sigs + 32, it, 64, Sw);
  }
};

__global__ void __launch_bounds__(BLOCK, 2)
k_sign_finish(uint8_t* sigs, uint32_t* acc, uint32_t* aux, const uint8_t* pubs, const uint8_t* msgs,
              const uint64_t* msg_off, const uint64_t* msg_end, size_t msg_len, size_t n, int K, const uint32_t* perm) {
  finish_batch8(sign_finish_policy{sigs, acc, aux, pubs, msgs, msg_off, msg_end, msg_len, n, K, perm}, acc);
}

template <int PARTS>
__global__ void __launch_bounds__(POINT_BLOCK, 512 / POINT_BLOCK)
k_x25519_base_point(uint32_t* accout, const uint8_t* scalars, size_t n, const uint32_t* comb, uint32_t* tiles) {
  __shared__ alignas(16) uint32_t lds_comb[COMB_IMG_WORDS];
  stage_table(lds_comb, comb, COMB_IMG_WORDS);
  __shared__ uint32_t lds_shares[PARTS == 1 ? 1 : 2 * 40 * 64];
  // blocks of POINT_BLOCK lanes, or smaller ones for small passes (EDK_POINT_GRID); PARTS = 4: four waves, 64 items
  const int part = PARTS == 1 ? 0 : (int)(threadIdx.x >> 6);
  size_t i;
#pragma unroll 1
  for (unsigned it = 0; point_tile<PARTS>(i, it, n, tiles); it++) {
    uint32_t s[8];
    load32(s, scalars, i < n ? i : n - 1, 32);
    ge R;
    x25519_base_point_lane<PARTS>(R, s, lds_comb, part);
    if (PARTS > 1) point_reduce4(R, lds_shares, part);
    fe_add(R.X, R.Z, R.Y);                       // the finish step needs z + y, not x
    fe_carry(R.X);
    if (PARTS == 1 || (part == 0 && i < n)) acc_store(accout, i, R);
  }
}

// u = (z + y) / (z - y); z = y gives 0 in the reference (fld_inv(0) = 0, x25519.c:192): such an
// item contributes 1 to the shared product and gets the "inverse" 0 by hand.
struct x25519_base_finish_policy {
  uint8_t* out; uint32_t* acc; size_t n; int K;
  ED_DEV void den(int k, fe& d) const {
    const finish_pos p = finish_at(k, acc, K);
    fe_set(d, 1);
    bool good = false;
    if (p.i < n) {
      fe y, z;
      acc_load(y, p.acc, 1); acc_load(z, p.acc, 2);
      fe_sub(d, z, y);                           // 3u
      good = !fe_iszero(d);
    }
    den_commit(d, good, acc, k, K);
  }
  ED_DEV void item(int k) const {
    const finish_pos p = finish_at(k, acc, K);
    if (p.i >= n) return;
    fe x, y, dinv, z, d;
    acc_load(y, p.acc, 1); acc_load(dinv, p.acc, 2);     // the Z slot now holds 1/(z - y) (or 1/1)
    // z itself was overwritten; recover the numerator z + y = (z - y) + 2y from X's slot instead:
    // the point kernel stores z + y there (x is not needed for x25519_base)
    acc_load(x, p.acc, 0);                               // = z + y
    fe_sub(d, x, y); fe_sub(d, d, y);                    // z - y, to detect the zero denominator
    fe zero;
    fe_set(zero, 0);
    fe_cmov(dinv, zero, fe_iszero(d));
    fe u;
    fe_mul(u, x, dinv);
    uint32_t w[8];
    fe_tobytes(w, u);
    store32(out, p.i, 32, w);
  }
};

__global__ void __launch_bounds__(BLOCK, 2)
k_x25519_base_finish(uint8_t* out, uint32_t* acc, size_t n, int K) {
  finish_batch8(x25519_base_finish_policy{out, acc, n, K}, acc);
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

// ---------------------------------------------------------------------------------------------
// The half-length route for SMALL passes (up to QUAD_MAIN_MAX_N items), where what counts is the latency of one
// item: k_verify_prepare_pair does the work of k_verify_prepare + k_verify_halve with three lanes per item in two kinds
// of blocks - the points' blocks, where lane 0 of a pair decompresses A and builds its table while lane 1 does the same
// for R, and the scalars' blocks, where one lane hashes and searches the pair (nothing of the one depends on the other:
// a single item's 0.13 ms became 0.09) - and k_verify_main_half_quad is verify_half_main_lane with a coordinate per
// lane (quad_lanes.h).  Two kernels, 132 doublings and no inversion instead of three, 252 and one.
// ---------------------------------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(BLOCK, 2)
k_verify_prepare_pair(edk_verify_src src, size_t n, uint32_t* digits, uint32_t* hdigits, uint32_t* table, uint32_t* rtable,
                      uint8_t* flags, uint32_t* onlist, uint32_t* offlist, uint32_t* offcount, int all_exact, unsigned point_blocks,
                      const uint32_t* perm) {
  if (blockIdx.x >= point_blocks) {
    // the scalars' blocks, one lane per item: hash, reduce, search the pair.  They need nothing from the points' blocks
    // and those nothing from here: the two square-root chains of an item run BESIDE its hash and its search
    const size_t g = (size_t)(blockIdx.x - point_blocks) * BLOCK + threadIdx.x;
    if (g >= n) return;
    const size_t i = perm ? perm[g] : g;         // (ragged messages in order of length)
    uint32_t rw[8], aw[8], sw[8], tw[8], hd[HALF_DIGIT_WORDS];
    const uint8_t* m; size_t mlen;
    verify_item(rw, sw, aw, m, mlen, src, i);
    verify_hash_lane(tw, rw, aw, m, mlen);
    verify_s_lane(sw);
    verify_half_scalars_lane<BITS>(hd, tw, sw);
    uint4* d = reinterpret_cast<uint4*>(digits + 16 * i);
    d[0] = make_uint4(tw[0], tw[1], tw[2], tw[3]); d[1] = make_uint4(tw[4], tw[5], tw[6], tw[7]);
    d[2] = make_uint4(sw[0], sw[1], sw[2], sw[3]); d[3] = make_uint4(sw[4], sw[5], sw[6], sw[7]);
    uint4* o = reinterpret_cast<uint4*>(hdigits + HALF_DIGIT_WORDS * i);
#pragma unroll
    for (int q = 0; q < HALF_DIGIT_WORDS / 4; q++) o[q] = make_uint4(hd[4 * q], hd[4 * q + 1], hd[4 * q + 2], hd[4 * q + 3]);
    if ((hd[24] & 4u) != 0) atomicAdd(offcount + EDK_REFUSED_WORD, 1u);
    return;
  }
  const size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t i = g >> 1;                       // pairs are all-or-nothing
  if (i >= n) return;
  const bool second = (g & 1) != 0;
  // lane 0: -A permissively (ed.c:100-149), lane 1: -R' strictly (lanes.h: verify_half_point_lane)
  uint32_t pw[8];
  load32(pw, second ? src.sigs : src.pubs, i, second ? src.sig_stride : src.pub_stride);
  uint32_t* tab = (second ? rtable : table) + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS);
  bool oncurve;
  ge p;
  ge_frombytes(p, oncurve, pw, true);
  bool top = (pw[7] & 0x7fffffffu) == 0x7fffffffu && pw[0] >= 0xffffffedu;
#pragma unroll
  for (int k = 1; k < 7; k++) top = top && pw[k] == 0xffffffffu;
  const bool strict = oncurve && !top && !((pw[7] >> 31) != 0 && fe_iszero(p.X));
  verify_table_point_lane(tab, p);
  const int mine = second ? (strict ? 1 : 0) : (oncurve ? 1 : 0);
  const int other = __shfl_xor(mine, 1);
  if (!second) {
    const bool keep = oncurve && !all_exact;
    flags[i] = (uint8_t)((keep ? 1 : 0) | (other ? 4 : 0));
    // the two work lists of k_verify_prepare (the four-lane evaluations of small passes do not use the first)
    const uint32_t on_slot = wave_append(offcount + EDK_ONLIST_WORD, keep);
    if (keep) onlist[on_slot] = (uint32_t)i;
    const uint32_t off_slot = wave_append(offcount, !keep);
    if (!keep) offlist[off_slot] = (uint32_t)i;
  }
}

template <int WINDOWS>
__global__ void __launch_bounds__(QUAD_BLOCK, 2)
k_verify_main_half_quad(uint8_t* ok, const uint32_t* hdigits, const uint32_t* table, const uint32_t* rtable,
                        const uint32_t* base16, const uint8_t* flags, size_t n, int exact_offcurve) {
  const size_t i = ((size_t)blockIdx.x * QUAD_BLOCK + threadIdx.x) >> 2;      // quads are all-or-nothing
  if (i >= n) return;
  const int q = (int)(threadIdx.x & 3u);
  const bool neutral = verify_half_main_quad<WINDOWS>(hdigits + HALF_DIGIT_WORDS * i, table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS),
                                             rtable + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS), base16, q);
  if (q != 0) return;
  const uint8_t fl = flags[i];
  if ((fl & 1) == 0) {                           // the exact path owns this verdict
    if (!exact_offcurve) ok[i] = 0;
    return;
  }
  ok[i] = (uint8_t)(neutral && (fl & 4) != 0);
}

// The smallest passes (quad_lanes.h: verify_half_window_sum_quad): block i adds up the 64 windows of item i, one quad each;
// then the chain over the sums.
template <int WINDOWS>
__global__ void __launch_bounds__(QUAD_BLOCK, 2)
k_verify_window_sums(uint32_t* sums, const uint32_t* hdigits, const uint32_t* table, const uint32_t* rtable, const uint32_t* base16) {
  const size_t i = blockIdx.x;
  const int w = (int)(threadIdx.x >> 2), q = (int)(threadIdx.x & 3u);
  verify_half_window_sum_quad<WINDOWS>(sums + (i * HALF_LONG_WINDOWS + (size_t)w) * HALF_SUM_WORDS, hdigits + HALF_DIGIT_WORDS * i,
                              table + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS), rtable + i * (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS),
                              base16, w, q);
}

template <int WINDOWS>
__global__ void __launch_bounds__(QUAD_BLOCK, 2)
k_verify_main_sums_quad(uint8_t* ok, const uint32_t* hdigits, const uint32_t* sums, const uint8_t* flags, size_t n, int exact_offcurve) {
  const size_t i = ((size_t)blockIdx.x * QUAD_BLOCK + threadIdx.x) >> 2;      // quads are all-or-nothing
  if (i >= n) return;
  const int q = (int)(threadIdx.x & 3u);
  const bool neutral = verify_half_main_sums_quad<WINDOWS>(hdigits + HALF_DIGIT_WORDS * i, sums + i * (HALF_LONG_WINDOWS * HALF_SUM_WORDS), q);
  if (q != 0) return;
  const uint8_t fl = flags[i];
  if ((fl & 1) == 0) {                           // the exact path owns this verdict
    if (!exact_offcurve) ok[i] = 0;
    return;
  }
  ok[i] = (uint8_t)(neutral && (fl & 4) != 0);
}

}  // namespace ed

// =============================================================================================
// launchers (host side of this translation unit)
// =============================================================================================

using namespace ed;

// Items per lane that share one inversion in the finish kernels: FINISH_K (8) in a pass that fills the chip; in a smaller
// pass as few as it takes to keep one block of 256 lanes per CU - a lane with one item runs 31 k dependent instructions,
// with eight 70 k, and a pass of 2^12 items would otherwise occupy two CUs.
static unsigned finish_k(size_t n) {
  const size_t tiles = (n + BLOCK - 1) / BLOCK;
  for (unsigned k = 1; k < (unsigned)FINISH_K; k *= 2)
    if ((tiles + k - 1) / k <= 256) return k;
  return FINISH_K;
}

extern "C" {

hipError_t edk_init_tables(uint32_t* base16, uint32_t* comb, uint32_t* comb_img, hipStream_t stream) {
  const int total = 2 * TABLE_BASE16_ENTRIES + TABLE_COMB_ENTRIES;
  hipLaunchKernelGGL(k_init_tables, dim3((total + 63) / 64), dim3(64), 0, stream, base16, comb);
  hipLaunchKernelGGL(k_init_comb_image, dim3((COMB_ROWS * COMB_IMG_ENTRIES + 63) / 64), dim3(64), 0, stream, comb_img, comb);
  return hipGetLastError();
}

hipError_t edk_x25519(uint8_t* out, const uint8_t* scalars, const uint8_t* points, size_t n,
                      const edk_fixed_ws* ws, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + BLOCK - 1) / BLOCK);
  if (n <= X25519_QUAD_MAX_N)
    hipLaunchKernelGGL(k_x25519_ladder_quad, dim3((unsigned)((4 * n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, ws->acc,
                       scalars, points, n);
  else
    hipLaunchKernelGGL(k_x25519_ladder, dim3(blocks), dim3(BLOCK), 0, stream, ws->acc, scalars, points, n);
  hipLaunchKernelGGL(k_x25519_finish, dim3((blocks + finish_k(n) - 1) / finish_k(n)), dim3(BLOCK), 0, stream, out, ws->acc, n, (int)finish_k(n));
  return hipGetLastError();
}

// the test hook of edk_checked.h: checked calls made since the count was restarted, and the call that is to fail
static std::atomic<int> g_checked{0}, g_fail_at{0}, g_counting{0};
int edk_fault_tick(void) {
  // Unarmed (every production pass): one relaxed load of a line nobody writes.  The calls are counted only while a fault is
  // pending or a test has restarted the count (g_counting): no shared counter is bumped on the hot path otherwise.
  int at = g_fail_at.load(std::memory_order_relaxed);
  if (at == 0 && !g_counting.load(std::memory_order_relaxed)) return 0;
  const int k = g_checked.fetch_add(1, std::memory_order_relaxed) + 1;
  return at != 0 && k == at && g_fail_at.compare_exchange_strong(at, 0);
}
void edk_debug_counting(int on) { g_counting.store(on != 0); }
int edk_debug_fail_in(int nth) {
  if (nth < 0) return g_checked.load();
  g_fail_at.store(0);
  g_checked.store(0);
  g_fail_at.store(nth);
  return 0;
}

// perm[0..n) = the items in order of message length, longest first (k_len_*): bins = 2 * LEN_BINS words of the workspace
static hipError_t msg_order(uint32_t* perm, uint32_t* bins, const uint64_t* msg_off, const uint64_t* msg_end, size_t n, hipStream_t stream) {
  const unsigned blocks = (unsigned)((n + BLOCK - 1) / BLOCK);
  hipError_t e = hipMemsetAsync(bins, 0, (size_t)LEN_BINS * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  static_assert(LEN_BINS % BLOCK == 0, "k_len_starts: a thread takes LEN_BINS / BLOCK bins");
  hipLaunchKernelGGL(k_len_count, dim3(blocks), dim3(BLOCK), 0, stream, bins, msg_off, msg_end, n);
  hipLaunchKernelGGL(k_len_starts, dim3(1), dim3(BLOCK), 0, stream, bins);
  hipLaunchKernelGGL(k_len_place, dim3(blocks), dim3(BLOCK), 0, stream, perm, bins, msg_off, msg_end, n);
  return hipGetLastError();
}

// the verify workspace's order for a pass of n items, or *perm = nullptr where the pass takes its items as they come
// (no offset table, or fewer than MSG_ORDER_MIN_N items); rlc.hip's hashing kernel shares it
hipError_t edk_msg_order(const uint32_t** perm, const edk_verify_ws* ws, const uint64_t* msg_off, const uint64_t* msg_end, size_t n,
                         hipStream_t stream) {
  *perm = nullptr;
  if (!msg_off || n < MSG_ORDER_MIN_N) return hipSuccess;
  const hipError_t e = msg_order(ws->perm, ws->lenbins, msg_off, msg_end, n, stream);
  if (e == hipSuccess) *perm = ws->perm;
  return e;
}

// Every HIP call below that orders work or moves data is checked (edk_checked.h): the first failure ends the pass with
// that error.  What has been queued by then still runs; eddsa_amd.c: verify_on waits for it (both streams) before the
// workspace can be handed out again, and the caller learns that the outputs are unspecified.
hipError_t edk_verify(uint8_t* ok, const edk_verify_src* srcp, size_t n, const uint32_t* base16,
                      const edk_verify_ws* ws, hipEvent_t* marks, hipEvent_t bulk_done, int bulk_early, hipStream_t stream) {
  const edk_verify_src src = *srcp;
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + BLOCK - 1) / BLOCK);
  EDK_DO(hipMemsetAsync(ws->offcount, 0, EDK_PASS_WORDS * sizeof(uint32_t), stream));   // both work lists' lengths, the one-lane exact path's unit counter and its stall flag
  if (marks) EDK_DO(hipEventRecord(marks[0], stream));
  // algo 0: half-length scalars - the three-lane preparation and four lanes per item up to 24 576 items, the same preparation and one lane per item up to 2^18, one lane per item above;
  // 3: the mid-size arrangement at any size below 2^18;
  // 1: full-length windows (one lane per item above QUAD_MAIN_MAX_N items, quads below); 2: half-length, one lane per item
  const bool small = n <= QUAD_MAIN_MAX_N;
  const bool small_half = n <= HALF_QUAD_MAX_N;
  const int algo = ws->exact_offcurve ? ws->algo : 1;   // the half-length route relies on the exact path for its give-ups
  const bool wide = n < HALF_WIDE_MIN_N;
  // mid-size passes: the three-lane preparation of the small route (its blocks fill the SIMDs that one lane per item leaves
  // with a single wave) and the one-lane evaluation; algo 3 forces it for measurements
  const bool pair_one = n < PAIR_ONE_MAX_N && (algo == 3 || (algo == 0 && n > PAIR_ONE_MIN_N));
  const bool half = !pair_one && (algo == 2 || (algo == 0 && !small_half));
  const bool half_quad = !pair_one && algo == 0 && small_half;
  const unsigned pair_point_blocks = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
  // The four-lane evaluation runs 64 windows in the wave of an item without a short pair, and the pass waits for that wave:
  // with pairs up to 2^134 (8.5 items in 10^5) a pass of 2048 items has such an item one time in six, one of 2^14 three
  // times in four.  Above QUAD_WIDE_MIN_N items the search goes up to 2^138 (2 in 10^7) for a 35th window in every item.
  const bool quad_wide = half_quad && n > QUAD_WIDE_MIN_N;
  // ragged messages: the hashing kernels take their items in order of length
  const uint32_t* perm = nullptr;
  EDK_DO(edk_msg_order(&perm, ws, src.msg_off, src.msg_end, n, stream));
  if (pair_one || quad_wide)
    EDK_LAUNCH(k_verify_prepare_pair<HALF_BITS_SMALL>, dim3(pair_point_blocks + blocks), dim3(BLOCK), 0, stream, src, n,
               ws->digits, ws->hdigits, ws->table, ws->rtable, ws->flags, ws->onlist, ws->offlist, ws->offcount, ws->exact_offcurve == 2,
               pair_point_blocks, perm);
  else if (half_quad)
    EDK_LAUNCH(k_verify_prepare_pair<HALF_BITS>, dim3(pair_point_blocks + blocks), dim3(BLOCK), 0, stream, src, n,
               ws->digits, ws->hdigits, ws->table, ws->rtable, ws->flags, ws->onlist, ws->offlist, ws->offcount, ws->exact_offcurve == 2,
               pair_point_blocks, perm);
  else
    EDK_LAUNCH(k_verify_prepare, dim3(blocks), dim3(BLOCK), 0, stream, src, n, ws->digits, ws->table, ws->flags, ws->onlist, ws->offlist,
               ws->offcount, ws->exact_offcurve == 2, perm);
  // Below HALF_WIDE_MIN_N items the pass searches pairs up to 2^138 and runs 35 windows (2 t in 10^7 without a pair instead
  // of 8.5 in 10^5; 3 % more instructions in the main kernel): an item without a short pair goes through the exact path's
  // chain, and beside a main kernel of one or two rounds of resident blocks that chain costs the pass 0.3-0.4 ms (the
  // SIMDs its waves sit on finish their tiles that much later and the grid has no slack: profiles/r04_small_grid.txt)
  if (half && wide)
    EDK_LAUNCH(k_verify_halve<HALF_BITS_SMALL>, dim3(blocks), dim3(BLOCK), 0, stream, src.sigs, src.sig_stride, ws->digits,
               ws->hdigits, ws->rtable, ws->flags, ws->onlist, ws->offlist, ws->offcount);
  else if (half)
    EDK_LAUNCH(k_verify_halve<HALF_BITS>, dim3(blocks), dim3(BLOCK), 0, stream, src.sigs, src.sig_stride, ws->digits,
               ws->hdigits, ws->rtable, ws->flags, ws->onlist, ws->offlist, ws->offcount);
  if (marks) EDK_DO(hipEventRecord(marks[1], stream));
  if (bulk_done && bulk_early) EDK_DO(hipEventRecord(bulk_done, stream));   // the next pass may start beside this one's main kernel
  // the exact path depends only on what came before: it runs beside the main kernel on the side stream
  if (ws->exact_offcurve) {
    // Passes of EXACT_LANE_MIN_N items or more hand a work list of EXACT_LANE_MIN_LISTED entries or more to the one-lane
    // kernels; which form runs is decided on the device (the host does not know the list's length): the other ends at once.
    const uint32_t lane_min = n >= EXACT_LANE_MIN_N ? (uint32_t)EXACT_LANE_MIN_LISTED : 0u;
    EDK_DO(hipEventRecord(ws->ev_prepared, stream));
    EDK_DO(hipStreamWaitEvent(ws->side, ws->ev_prepared, 0));
    // The one-lane kernels go FIRST on the side stream: when the list is short they end at once, and they must do so while
    // the chip still has room for their (large) blocks - queued behind the four-lane chain they reached the dispatcher a
    // millisecond into a full k_verify_main_half, where each of their 512 idle blocks had to wait for two wave slots of one
    // SIMD to fall free together (profiles/r05_exact_lane.txt: same-box A/B, 107.2 -> 107.9 M/s on config 2).
    if (lane_min != 0) {
      const unsigned lane_blocks = blocks < EXACT_LANE_BLOCKS ? blocks : EXACT_LANE_BLOCKS;
      // (the scratchpad of the four-lane chain is free when these run: it holds their per-tile counts)
      EDK_LAUNCH(k_verify_exact_lane_setup, dim3(lane_blocks), dim3(BLOCK), 0, ws->side, ws->digits, ws->table, ws->rtable, ws->offlist,
                 ws->offcount, base16, ws->exact_pad, lane_min);
      // (what runs beside it: the half-length evaluation over the on-curve list, 226 k instructions per item, or the
      // full-length one over the whole pass, 323 k: profiles/pmc_summary.json)
      const bool half_main = half || pair_one;
      EDK_LAUNCH(k_verify_exact_lane_chain, dim3(lane_blocks), dim3(BLOCK), 0, ws->side, ok, src.sigs, src.sig_stride, ws->table, ws->rtable,
                 ws->offlist, ws->offcount, ws->exact_pad, lane_min, n, half_main ? 226u : 323u, half_main ? 0 : 1, ws->status);
    }
    {
      // one launch: a list it would serve has fewer entries than the scratchpad has slots (lane_min != 0: fewer than lane_min;
      // else the pass itself is smaller than EXACT_LANE_MIN_N)
      static_assert(EXACT_LANE_MIN_LISTED <= (size_t)EDK_EXACT_SLOTS && EXACT_LANE_MIN_N <= (size_t)EDK_EXACT_SLOTS, "the four-lane chain's list fits its scratchpad");
      const size_t qi = lane_min != 0 ? (size_t)lane_min : n;
      const size_t dense = (qi + QUAD_CHAIN_ITEMS - 1) / QUAD_CHAIN_ITEMS, spread = qi < (size_t)QUAD_SPREAD_WAVES ? qi : (size_t)QUAD_SPREAD_WAVES;
      EDK_LAUNCH(k_verify_exact_quad, dim3((unsigned)(dense > spread ? dense : spread)), dim3(QUAD_CHAIN_BLOCK), 0, ws->side, ok,
                 src.sigs, src.sig_stride, ws->digits, ws->table, ws->offlist, ws->offcount, base16, ws->exact_pad,
                 (int)(n >= EXACT_DENSE_MIN_N), lane_min);
    }
    EDK_DO(hipEventRecord(ws->ev_exact, ws->side));
  }
  if (half_quad && n <= EDK_SUMS_MAX_ITEMS) {
    static_assert(QUAD_BLOCK == 4 * HALF_LONG_WINDOWS, "a block of k_verify_window_sums is the windows of one item");
    if (quad_wide) {
      EDK_LAUNCH(k_verify_window_sums<HALF_WINDOWS_SMALL>, dim3((unsigned)n), dim3(QUAD_BLOCK), 0, stream, ws->sums, ws->hdigits, ws->table, ws->rtable, base16);
      EDK_LAUNCH(k_verify_main_sums_quad<HALF_WINDOWS_SMALL>, dim3((unsigned)((4 * n + QUAD_BLOCK - 1) / QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, stream,
                 ok, ws->hdigits, ws->sums, ws->flags, n, ws->exact_offcurve);
    } else {
      EDK_LAUNCH(k_verify_window_sums<HALF_WINDOWS>, dim3((unsigned)n), dim3(QUAD_BLOCK), 0, stream, ws->sums, ws->hdigits, ws->table, ws->rtable, base16);
      EDK_LAUNCH(k_verify_main_sums_quad<HALF_WINDOWS>, dim3((unsigned)((4 * n + QUAD_BLOCK - 1) / QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, stream,
                 ok, ws->hdigits, ws->sums, ws->flags, n, ws->exact_offcurve);
    }
    if (marks) { EDK_DO(hipEventRecord(marks[2], stream)); EDK_DO(hipEventRecord(marks[3], stream)); }
  } else if (half_quad) {
    if (quad_wide)
      EDK_LAUNCH(k_verify_main_half_quad<HALF_WINDOWS_SMALL>, dim3((unsigned)((4 * n + QUAD_BLOCK - 1) / QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, stream,
                 ok, ws->hdigits, ws->table, ws->rtable, base16, ws->flags, n, ws->exact_offcurve);
    else
      EDK_LAUNCH(k_verify_main_half_quad<HALF_WINDOWS>, dim3((unsigned)((4 * n + QUAD_BLOCK - 1) / QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, stream,
                 ok, ws->hdigits, ws->table, ws->rtable, base16, ws->flags, n, ws->exact_offcurve);
    if (marks) { EDK_DO(hipEventRecord(marks[2], stream)); EDK_DO(hipEventRecord(marks[3], stream)); }
  } else if (half || pair_one) {
    constexpr unsigned half_lds = MAIN_LDS_RESERVE;
    static_assert(HALF_DIGIT_WORDS * MAIN_HALF_BLOCK * 4 <= half_lds, "k_verify_main_half keeps its digit words in the block's LDS");
    const unsigned hblocks = (blocks * BLOCK + MAIN_HALF_BLOCK - 1) / MAIN_HALF_BLOCK;
    if (pair_one)
      EDK_LAUNCH((k_verify_main_half<HALF_WINDOWS_SMALL, true>), dim3(hblocks), dim3(MAIN_HALF_BLOCK), half_lds, stream, ok, ws->hdigits,
                 ws->table, ws->rtable, base16, ws->flags, ws->onlist, ws->offcount);
    else if (wide)
      EDK_LAUNCH(k_verify_main_half<HALF_WINDOWS_SMALL>, dim3(hblocks), dim3(MAIN_HALF_BLOCK), half_lds, stream, ok, ws->hdigits,
                 ws->table, ws->rtable, base16, ws->flags, ws->onlist, ws->offcount);
    else
      EDK_LAUNCH(k_verify_main_half<HALF_WINDOWS>, dim3(hblocks), dim3(MAIN_HALF_BLOCK), half_lds, stream, ok, ws->hdigits,
                 ws->table, ws->rtable, base16, ws->flags, ws->onlist, ws->offcount);
    if (marks) { EDK_DO(hipEventRecord(marks[2], stream)); EDK_DO(hipEventRecord(marks[3], stream)); }
  } else {
    if (small)
      EDK_LAUNCH(k_verify_main_quad, dim3((unsigned)((4 * n + QUAD_BLOCK - 1) / QUAD_BLOCK)), dim3(QUAD_BLOCK), 0,
                 stream, ws->digits, ws->table, base16, ws->acc, n);
    else
      EDK_LAUNCH(k_verify_main, dim3(blocks), dim3(BLOCK), MAIN_LDS_RESERVE, stream, ws->digits, ws->table, base16, ws->acc);
    if (marks) EDK_DO(hipEventRecord(marks[2], stream));
    EDK_LAUNCH(k_verify_finish, dim3((blocks + finish_k(n) - 1) / finish_k(n)), dim3(BLOCK), 0, stream, ok, src.sigs,
               src.sig_stride, ws->acc, ws->flags, n, ws->exact_offcurve, (int)finish_k(n));
    if (marks) EDK_DO(hipEventRecord(marks[3], stream));
  }
  // everything that fills the chip has been queued; what follows on this stream only waits for the exact path's few
  // latency-bound waves: a caller that pipelines passes over several workspaces starts the next pass from here
  if (bulk_done && !bulk_early) EDK_DO(hipEventRecord(bulk_done, stream));
  if (ws->exact_offcurve) EDK_DO(hipStreamWaitEvent(stream, ws->ev_exact, 0));   // complete when both paths are
  return hipSuccess;
}

#define EDK_GRID(n) dim3((unsigned)(((n) + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream

// Point kernels stage the comb in LDS per block, one block per CU.  A pass that fills the chip runs blocks of POINT_BLOCK
// lanes (two waves per SIMD).  A smaller pass runs smaller blocks: a block of 512 lanes puts two waves on every SIMD of
// ONE CU, which then take turns (a single ed25519_genpub ran its one useful wave beside seven idle-lane waves redoing
// the same item: 0.23 ms in the point kernel against 0.12 with a wave to itself), and 4096 items occupied 8 CUs of 256.
// So: the smallest block that covers the pass with one block per CU - one, two or four waves, each with a SIMD to itself.
#define POINT_LANES(n) ((n) <= (size_t)64 * 256 ? 64 : (n) <= (size_t)128 * 256 ? 128 : (n) <= (size_t)256 * 256 ? 256 : POINT_BLOCK)
#define POINT_GRID_BLOCKS(n) ((unsigned)(((n) + POINT_LANES(n) - 1) / POINT_LANES(n)))
// Passes of up to POINT_SPLIT_MAX_N items spend four lanes on an item (lanes.h: scale_base_lane<4>, k_genpub_point above): the 44 additions of
// the comb in a row were the latency of the pass (a single ed25519_sign: 0.17 ms in k_sign_point, now 0.09).  A pass with more
// tiles than CUs runs POINT_MAX_BLOCKS persistent blocks whose waves draw their tiles from ws->tiles (point_tile above); a smaller
// one a block per tile, without the counter.
constexpr size_t POINT_SPLIT_MAX_N = (size_t)1 << 14;
#define EDK_POINT_LAUNCH(kernel, n, ...) do { \
    if ((n) <= POINT_SPLIT_MAX_N) hipLaunchKernelGGL((kernel<POINT_SPLIT>), dim3((unsigned)(((n) + POINT_SPLIT_ITEMS - 1) / POINT_SPLIT_ITEMS)), \
                                                     dim3(POINT_SPLIT * 64), 0, stream, __VA_ARGS__, (uint32_t*)nullptr); \
    else if (POINT_GRID_BLOCKS(n) > POINT_MAX_BLOCKS) { \
      const hipError_t z_ = hipMemsetAsync(ws->tiles, 0, sizeof(uint32_t), stream); if (z_ != hipSuccess) return z_; \
      hipLaunchKernelGGL((kernel<1>), dim3(POINT_MAX_BLOCKS), dim3(POINT_LANES(n)), 0, stream, __VA_ARGS__, ws->tiles); } \
    else hipLaunchKernelGGL((kernel<1>), dim3(POINT_GRID_BLOCKS(n)), dim3(POINT_LANES(n)), 0, stream, __VA_ARGS__, (uint32_t*)nullptr); } while (0)
#define EDK_FINISH_GRID(n) dim3((unsigned)((((n) + BLOCK - 1) / BLOCK + finish_k(n) - 1) / finish_k(n))), dim3(BLOCK), 0, stream

hipError_t edk_genpub(uint8_t* pubs, const uint8_t* secs, size_t n, const uint32_t* comb,
                      const edk_fixed_ws* ws, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  EDK_POINT_LAUNCH(k_genpub_point, n, ws->acc, secs, n, comb);
  hipLaunchKernelGGL(k_encode_finish, EDK_FINISH_GRID(n), pubs, ws->acc, n, (int)finish_k(n));
  return hipGetLastError();
}

hipError_t edk_sign(uint8_t* sigs, const uint8_t* secs, const uint8_t* pubs, const uint8_t* msgs,
                    const uint64_t* msg_off, const uint64_t* msg_end, size_t msg_len, size_t n, const uint32_t* comb,
                    const edk_fixed_ws* ws, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const uint32_t* perm = nullptr;                  // ragged messages: both kernels hash, both take their items in order of length
  if (msg_off && n >= MSG_ORDER_MIN_N) {
    const hipError_t e = msg_order(ws->perm, ws->lenbins, msg_off, msg_end, n, stream);
    if (e != hipSuccess) return e;
    perm = ws->perm;
  }
  EDK_POINT_LAUNCH(k_sign_point, n, ws->acc, ws->aux, secs, msgs, msg_off, msg_end, msg_len, n, comb, perm);
  hipLaunchKernelGGL(k_sign_finish, EDK_FINISH_GRID(n), sigs, ws->acc, ws->aux, pubs, msgs, msg_off, msg_end, msg_len, n, (int)finish_k(n), perm);
  return hipGetLastError();
}

hipError_t edk_x25519_base(uint8_t* out, const uint8_t* scalars, size_t n, const uint32_t* comb,
                           const edk_fixed_ws* ws, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  EDK_POINT_LAUNCH(k_x25519_base_point, n, ws->acc, scalars, n, comb);
  hipLaunchKernelGGL(k_x25519_base_finish, EDK_FINISH_GRID(n), out, ws->acc, n, (int)finish_k(n));
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
