// probe.hip - the layer probes: ONE layer of the device code on caller-given inputs, as a library of its own
// (libeddsa_amd_probe.so; include/eddsa_amd_probe.h).  Test infrastructure: the product library (libeddsa_amd.so) holds no
// probe kernel and exports none of this - until round 5 these kernels shipped inside it (k_debug_layer alone: 511 VGPRs,
// 107 of them spilled).  The probes compile the SAME device source the product's kernels are made of (lanes.h,
// quad_lanes.h, kernel_io.h), so the golden layer vectors (tests/golden/layer_kats.json, pinned to the reference's static
// library) reach the GPU as layers and not only as parts of whole operations - what the device toolchain makes of the limb
// arithmetic and of the DPP exchanges is the one thing the host build of this source (tests/host_check/) cannot vouch for.
// Self-contained: its own copy of the generated tables (built on first use, per device), plain hipMalloc / hipMemcpy.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "eddsa_amd_probe.h"
#include "kernel_io.h"
#include "lanes.h"
#include "quad_lanes.h"

namespace ed {

constexpr int BLOCK = 256;
constexpr int QUAD_CHAIN_BLOCK = 64, QUAD_CHAIN_ITEMS = QUAD_CHAIN_BLOCK / 4;       // as kernels.hip: k_verify_exact_quad
constexpr int POINT_BLOCK = COMB_IMG_WORDS * 4 > 80 * 1024 ? 512 : 256;             // as kernels.hip: the point kernels

// the generated tables, as kernels.hip: k_init_tables / k_init_comb_image make them
__global__ void __launch_bounds__(64) k_probe_tables(uint32_t* base16, uint32_t* comb) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  if (id >= 2 * TABLE_BASE16_ENTRIES + TABLE_COMB_ENTRIES) return;
  if (id < TABLE_BASE16_ENTRIES) table_entry_lane(base16 + (size_t)TABLE_ENTRY_WORDS * id, (uint32_t)id, 0);
  else if (id < 2 * TABLE_BASE16_ENTRIES) table_entry_lane(base16 + (size_t)TABLE_ENTRY_WORDS * id, (uint32_t)(id - TABLE_BASE16_ENTRIES), 128);
  else {
    const int c = id - 2 * TABLE_BASE16_ENTRIES;
    table_entry_lane(comb + TABLE_ENTRY_WORDS * c, (uint32_t)(c % COMB_HALF) + 1, 2u * COMB_W * (uint32_t)(c / COMB_HALF));
  }
}
__global__ void __launch_bounds__(64) k_probe_comb_image(uint32_t* img, const uint32_t* comb) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  if (id >= COMB_ROWS * COMB_IMG_ENTRIES) return;
  comb_image_entry_lane(img + COMB_IMG_ENTRY_WORDS * id, comb, id / COMB_IMG_ENTRIES, id % COMB_IMG_ENTRIES);
}

// diagnostic (eddsa_amd_probe_halve): halve_scalar_lane on the device for given t; out = v (20 bytes) | |u| (20) |
// u < 0 (1) | found (1) | 6 bytes of padding per item
template <int BITS>
__global__ void __launch_bounds__(BLOCK) k_debug_halve(uint8_t* out, const uint8_t* t, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t tw[8], vw[5], uw[5];
  load32(tw, t, i, 32);
  bool uneg;
  const bool found = halve_scalar_lane<BITS>(vw, uw, uneg, tw);
  uint32_t* o = reinterpret_cast<uint32_t*>(out + 48 * i);
#pragma unroll
  for (int k = 0; k < 5; k++) { o[k] = vw[k]; o[5 + k] = uw[k]; }
  o[10] = (uneg ? 1u : 0u) | (found ? 0x100u : 0u);
  o[11] = 0;
}

// ---------------------------------------------------------------------------------------------
// Layer probes (include/eddsa_amd_probe.h: eddsa_amd_probe_layer): ONE layer of the device code on caller-given inputs,
// so that the golden layer vectors (tests/golden/layer_kats.json, pinned to the reference's static library) reach the
// GPU as such and not only as parts of whole operations - what the device toolchain makes of the limb arithmetic and of
// the DPP exchanges is the one thing the host build of this source (tests/host_check/) cannot vouch for.
// One lane per item; the four-lane forms below.  The op codes are those of the header.
// ---------------------------------------------------------------------------------------------
enum { L_FE_MUL = 1, L_FE_SQ, L_FE_INV, L_FE_POW2523, L_FE_MUL_LOOSE, L_SC_REDUCE32, L_SC_REDUCE64, L_SC_MULADD, L_SHA512,
       L_ED_IMPORT_EXPORT, L_ED_SCALE_BASE, L_ED_DUAL_SCALE, L_GE_DBL_ADD };

ED_DEV void ldw(uint32_t w[8], const uint8_t* p) { load32(w, p, 0, 0); }
ED_DEV void stw(uint8_t* p, const uint32_t w[8]) { store32(p, 0, 0, w); }

// scratch (EDL_ED_DUAL_SCALE only): per item 2 x REF_JSF_LEN digit bytes + 160 words (form 2: the uniform chain's storage)
constexpr size_t LAYER_SCRATCH_BYTES = 2 * ((REF_JSF_LEN + 3) / 4 * 4) + 160 * 4;
// form 3: four table entries | 36 digit words | the accumulator between stretches | 16 digit words | padding | the shared entry; 128-byte aligned slots
constexpr size_t LAYER_TABLE_SCRATCH_BYTES = (4 * VERIFY_ENTRY_WORDS + 36 + EXACT_STATE_WORDS + 16 + 4 + VERIFY_ENTRY_WORDS) * 4;
static_assert(LAYER_TABLE_SCRATCH_BYTES % 128 == 0, "entries are 128-byte lines");
// form 4: two table slots | two slots for digits (at +64 words) and walks (at +128) | the shared entry
constexpr size_t LAYER_PAIR_SCRATCH_BYTES = (4 * VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS + VERIFY_ENTRY_WORDS) * 4;
static_assert(LAYER_PAIR_SCRATCH_BYTES % 128 == 0, "entries are 128-byte lines");

__global__ void __launch_bounds__(64)
k_debug_layer(int op, int form, uint8_t* out, size_t out_w, const uint8_t* in, size_t in_w, size_t n, const uint32_t* base16,
              uint8_t* scratch) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* a = in + i * in_w;
  uint8_t* o = out + i * out_w;
  uint32_t w[8], r[8];
  fe x, y;
  if (op == L_FE_MUL || op == L_FE_MUL_LOOSE) {
    ldw(w, a); fe_frombytes(x, w); ldw(w, a + 32); fe_frombytes(y, w);
    if (op == L_FE_MUL_LOOSE) {                  // f = ka a (ka <= 7), g = kb b (kb <= 3): the documented operand limits
      const int ka = a[64], kb = a[65];
      fe fx, gy;
      fe_set(fx, 0); fe_set(gy, 0);
      for (int k = 0; k < ka; k++) fe_add(fx, fx, x);
      for (int k = 0; k < kb; k++) fe_add(gy, gy, y);
      x = fx; y = gy;
    }
    fe_mul(x, x, y);
    fe_tobytes(r, x); stw(o, r);
  } else if (op == L_FE_SQ || op == L_FE_INV || op == L_FE_POW2523) {
    ldw(w, a); fe_frombytes(x, w);
    if (op == L_FE_SQ) fe_sq(x, x); else if (op == L_FE_INV) fe_inv(x, x); else fe_pow2523(x, x);
    fe_tobytes(r, x); stw(o, r);
  } else if (op == L_SC_REDUCE32 || op == L_SC_REDUCE64) {
    uint32_t w16[16];
    sc t;
    ldw(w16, a);
    if (op == L_SC_REDUCE64) { ldw(w16 + 8, a + 32); sc_from_words<16>(t, w16); } else sc_from_words<8>(t, w16);
    sc_to_words(r, t); stw(o, r);
  } else if (op == L_SC_MULADD) {
    sc p, q, c;
    ldw(w, a); sc_from_words<8>(p, w); ldw(w, a + 32); sc_from_words<8>(q, w); ldw(w, a + 64); sc_from_words<8>(c, w);
    sc_mul(p, p, q); sc_add(p, c, p);
    sc_to_words(r, p); stw(o, r);
  } else if (op == L_SHA512) {
    uint32_t d[16];
    size_t len = 0;
    for (int k = 7; k >= 0; k--) len = (len << 8) | a[k];
    if (len > in_w - 8) len = in_w - 8;            // (the host side refuses such an item before the launch; never read past the slot)
    sha512_prefix_msg<0>(d, nullptr, a + 8, len);
    stw(o, d); stw(o + 32, d + 8);
  } else if (op == L_ED_IMPORT_EXPORT) {
    ge p; bool oc;
    ldw(w, a);
    ge_frombytes(p, oc, w, false);
    ge_tobytes(r, p); stw(o, r);
    o[32] = oc ? 1 : 0;
  } else if (op == L_ED_DUAL_SCALE) {            // forms 0 (the literal chain) and 2 (uniform control flow); form 1 is k_debug_dual_scale_quad
    uint32_t sw[8], tw[8];
    sc s, t;
    ldw(w, a); sc_from_words<8>(s, w); sc_to_words(sw, s);
    ldw(w, a + 32); sc_from_words<8>(t, w); sc_to_words(tw, t);
    ldw(w, a + 64);
    ge Q, R; bool oc;
    ge_frombytes(Q, oc, w, false);
    ge_niels pcB;
    niels_load(pcB, base16 + TABLE_ENTRY_WORDS);
    if (form == 4) {                              // two items per lane, as k_verify_exact_lane_chain walks them: this item and the next one of the batch
      constexpr uint32_t SLOT = VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS;
      uint32_t* sp = reinterpret_cast<uint32_t*>(scratch + i * LAYER_PAIR_SCRATCH_BYTES);
      uint32_t* tabs = sp; uint32_t* digs = sp + 2 * SLOT; uint32_t* bentry = sp + 4 * SLOT;
      exact_bentry_store(bentry);
      for (int which = 0; which < 2; which++) {
        const uint8_t* b = in + ((i + (size_t)which) % n) * in_w;
        uint32_t digits[16], bw[8];
        sc bs, bt;
        ldw(bw, b); sc_from_words<8>(bs, bw); sc_to_words(digits + 8, bs);
        ldw(bw, b + 32); sc_from_words<8>(bt, bw); sc_to_words(digits, bt);
        words_add_pattern(digits, 0x88888888u);
        words_add_pattern(digits + 8, 0x80008000u);
        ldw(bw, b + 64);
        ge P; bool poc;
        ge_frombytes(P, poc, bw, false);
        ge_cached c;
        ge_to_cached(c, P);
        cached_store(tabs + which * SLOT, 1, c);
        verify_exact_setup_table_lane(tabs + which * SLOT, digs + which * SLOT + 64, 1, digits, base16 + TABLE_ENTRY_WORDS);
      }
      ge rb;
      ge_neutral(R); ge_neutral(rb);
      exact_walk wa = exact_walk_start(digs + 64, 1, true), wb = exact_walk_start(digs + SLOT + 64, 1, true);
      for (int seg = 0; seg < EXACT_SEGS; seg++) {   // units of work with the accumulators and walks handed on through memory
        if (seg) { exact_walk_load(R, wa, digs + 128); exact_walk_load(rb, wb, digs + SLOT + 128); }
        exact_pair_iterations(R, wa, rb, wb, tabs, digs + 64, SLOT, 0, 1, bentry, seg == EXACT_SEGS - 1 ? -1 : EXACT_PAIR_ITERS);
        exact_walk_store(digs + 128, R, wa); exact_walk_store(digs + SLOT + 128, rb, wb);
      }
    } else if (form == 3) {                       // the one-lane throughput form: set-up into the item's own table, the chain stretch by stretch
      uint32_t* sp = reinterpret_cast<uint32_t*>(scratch + i * LAYER_TABLE_SCRATCH_BYTES);
      uint32_t* tab = sp; uint32_t* dig = sp + 4 * VERIFY_ENTRY_WORDS; uint32_t* state = dig + 36; uint32_t* digits = state + EXACT_STATE_WORDS;
      uint32_t* bentry = sp + LAYER_TABLE_SCRATCH_BYTES / 4 - VERIFY_ENTRY_WORDS;
      words_add_pattern(tw, 0x88888888u);          // as k_verify_prepare leaves them
      words_add_pattern(sw, 0x80008000u);
      for (int k = 0; k < 8; k++) { digits[k] = tw[k]; digits[8 + k] = sw[k]; }
      ge_cached c;
      ge_to_cached(c, Q);
      cached_store(tab, 1, c);
      exact_bentry_store(bentry);
      verify_exact_setup_table_lane(tab, dig, 1, digits, base16 + TABLE_ENTRY_WORDS);
      for (int seg = 0; seg < EXACT_SEGS; seg++) {
        if (seg == 0) ge_neutral(R); else exact_state_load(R, state);
        exact_chain_steps(R, tab, bentry, dig, 1, seg == 0 ? exact_seg_hi(0) : seg == 1 ? exact_seg_hi(1) : seg == 2 ? exact_seg_hi(2) : exact_seg_hi(3),
                          seg == 0 ? exact_seg_lo(0) : seg == 1 ? exact_seg_lo(1) : seg == 2 ? exact_seg_lo(2) : exact_seg_lo(3));
        exact_state_store(state, R);
      }
    } else if (form == 2) {
      uint8_t* sp = scratch + i * LAYER_SCRATCH_BYTES;
      int8_t* ux = reinterpret_cast<int8_t*>(sp);
      int8_t* uy = ux + (REF_JSF_LEN + 3) / 4 * 4;
      uint32_t* pts = reinterpret_cast<uint32_t*>(sp + 2 * ((REF_JSF_LEN + 3) / 4 * 4));
      ref_dual_scale_uniform(R, sw, tw, Q, pcB, ux, uy, pts, 1);
    } else {
      ref_dual_scale(R, sw, tw, Q, pcB);
    }
    ge_tobytes(r, R); stw(o, r);
  } else if (op == L_GE_DBL_ADD) {               // enc(2 P + k B), k < 65536 from the k B table: the windowed evaluation's two steps
    ge P; bool oc;
    ldw(w, a);
    ge_frombytes(P, oc, w, false);
    const uint32_t k = (uint32_t)a[32] | ((uint32_t)a[33] << 8);
    ge_dbl(P, P, true);
    ge_niels nb;
    niels_load(nb, base16 + TABLE_ENTRY_WORDS * (k > 32768u ? 32768u : k));
    ge_add_niels(P, P, nb, false);
    ge_tobytes(r, P); stw(o, r);
  }
}

// ed_scale_base with the comb staged in LDS and the shuffle lookup, as the point kernels run it (every lane of a wave
// active: idle lanes redo the last item)
__global__ void __launch_bounds__(POINT_BLOCK, 512 / POINT_BLOCK)
k_debug_scale_base(uint8_t* out, const uint8_t* in, size_t n, const uint32_t* comb) {
  __shared__ alignas(16) uint32_t lds_comb[COMB_IMG_WORDS];
  stage_table(lds_comb, comb, COMB_IMG_WORDS);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t w[8], xw[8], r[8];
  load32(w, in, i < n ? i : n - 1, 32);
  sc x;
  sc_from_words<8>(x, w); sc_to_words(xw, x);
  ge p;
  scale_base_lane<1>(p, xw, lds_comb, 0);
  if (i >= n) return;
  ge_tobytes(r, p);
  store32(out, i, 32, r);
}

// The four-lane forms (quad_lanes.h), one wave per block, 16 items per wave.
// EDL_ED_DUAL_SCALE: set-up and chain of the exact path on a given (s, t, q): the lanes first build what k_verify_prepare
// would have left - the digit words and entry 1 of the item's table, here the cached form of Q itself - in scratch.
// scratch per item: 16 digit words | 2 table entries (64 words) | the chain's slot (QUAD_ITEM_WORDS)
constexpr size_t LAYER_QUAD_SCRATCH_WORDS = 16 + 2 * VERIFY_ENTRY_WORDS + QUAD_ITEM_WORDS;
__global__ void __launch_bounds__(QUAD_CHAIN_BLOCK, 2)
k_debug_dual_scale_quad(uint8_t* out, const uint8_t* in, size_t n, const uint32_t* base16, uint32_t* scratch) {
  __shared__ uint32_t lds_dig[QUAD_CHAIN_ITEMS * QUAD_DIGIT_WORDS];
  const size_t quad = threadIdx.x >> 2;
  const size_t g = (size_t)blockIdx.x * QUAD_CHAIN_ITEMS + quad;
  const bool live = g < n;
  const size_t i = live ? g : n - 1;             // (whole waves run: a quad past the end redoes the last item into its own slot)
  const int q = (int)(threadIdx.x & 3u);
  uint32_t* sp = scratch + g * LAYER_QUAD_SCRATCH_WORDS;
  uint32_t* digits = sp; uint32_t* tab = sp + 16; uint32_t* item = sp + 16 + 2 * VERIFY_ENTRY_WORDS;
  const uint8_t* a = in + i * 96;
  if (q == 0) {
    uint32_t w[8], sw[8], tw[8];
    sc s, t;
    ldw(w, a); sc_from_words<8>(s, w); sc_to_words(sw, s);
    ldw(w, a + 32); sc_from_words<8>(t, w); sc_to_words(tw, t);
    words_add_pattern(tw, 0x88888888u);
    words_add_pattern(sw, 0x80008000u);
    for (int k = 0; k < 8; k++) { digits[k] = tw[k]; digits[8 + k] = sw[k]; }
  } else if (q == 1) {
    uint32_t w[8];
    ge Q; bool oc;
    ldw(w, a + 64);
    ge_frombytes(Q, oc, w, false);
    ge_cached c;
    ge_to_cached(c, Q);
    cached_store(tab, 1, c);
  }
  __syncthreads();
  uint32_t* dig = lds_dig + quad * QUAD_DIGIT_WORDS;
  verify_exact_setup_quad(digits, tab, base16 + TABLE_ENTRY_WORDS, item, dig, q);
  __syncthreads();
  uint32_t wd[8];
  exact_chain_encode_quad(wd, item, dig, q);
  if (live && q == 1) store32(out, g, 32, wd);
}

// EDL_GE_DBL_ADD with a coordinate per lane: quad_dbl, then quad_add_entry from the k B table
__global__ void __launch_bounds__(QUAD_CHAIN_BLOCK, 2)
k_debug_dbl_add_quad(uint8_t* out, const uint8_t* in, size_t n, const uint32_t* base16) {
  const size_t g = ((size_t)blockIdx.x * QUAD_CHAIN_BLOCK + threadIdx.x) >> 2;
  const bool live = g < n;
  const uint8_t* a = in + (live ? g : n - 1) * 40;
  const int q = (int)(threadIdx.x & 3u);
  uint32_t w[8];
  ge P; bool oc;
  ldw(w, a);
  ge_frombytes(P, oc, w, false);
  const uint32_t k = (uint32_t)a[32] | ((uint32_t)a[33] << 8);
  fe r = q == 0 ? P.X : q == 1 ? P.Y : q == 2 ? P.T : P.Z;
  quad_dbl(r, q);
  quad_add_entry(r, base16 + TABLE_ENTRY_WORDS * (k > 32768u ? 32768u : k), false, false, q);
  uint32_t wd[8];
  quad_encode(wd, r);
  if (live && q == 1) store32(out, g, 32, wd);
}

}  // namespace ed

using namespace ed;

static int layer_widths_ok(int op, int form, size_t in_w, size_t out_w) {
  if (form < 0 || form > 4) return 0;
  if (form >= 3 && op != L_ED_DUAL_SCALE) return 0;
  switch (op) {
    case L_FE_MUL: return form == 0 && in_w == 64 && out_w == 32;
    case L_FE_SQ: case L_FE_INV: case L_FE_POW2523: case L_SC_REDUCE32: case L_ED_SCALE_BASE: return form == 0 && in_w == 32 && out_w == 32;
    case L_FE_MUL_LOOSE: return form == 0 && in_w == 72 && out_w == 32;
    case L_SC_REDUCE64: return form == 0 && in_w == 64 && out_w == 32;
    case L_SC_MULADD: return form == 0 && in_w == 96 && out_w == 32;
    case L_SHA512: return form == 0 && in_w >= 8 && out_w == 64;
    case L_ED_IMPORT_EXPORT: return form == 0 && in_w == 32 && out_w == 33;
    case L_ED_DUAL_SCALE: return in_w == 96 && out_w == 32;                         // forms 0..4
    case L_GE_DBL_ADD: return form <= 1 && in_w == 40 && out_w == 32;
  }
  return 0;
}

static hipError_t layer_launch(int op, int form, uint8_t* out, size_t out_w, const uint8_t* in, size_t in_w, size_t n,
                           const uint32_t* base16, const uint32_t* comb_img, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if (!layer_widths_ok(op, form, in_w, out_w)) return hipErrorInvalidValue;
  void* scratch = nullptr;
  hipError_t e = hipSuccess;
  const unsigned qblocks = (unsigned)((n + QUAD_CHAIN_ITEMS - 1) / QUAD_CHAIN_ITEMS);
  if (op == L_ED_SCALE_BASE) {
    hipLaunchKernelGGL(k_debug_scale_base, dim3((unsigned)((n + POINT_BLOCK - 1) / POINT_BLOCK)), dim3(POINT_BLOCK), 0, stream, out, in, n, comb_img);
  } else if (op == L_ED_DUAL_SCALE && form == 1) {
    if ((e = hipMalloc(&scratch, (size_t)qblocks * QUAD_CHAIN_ITEMS * LAYER_QUAD_SCRATCH_WORDS * 4)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_debug_dual_scale_quad, dim3(qblocks), dim3(QUAD_CHAIN_BLOCK), 0, stream, out, in, n, base16, (uint32_t*)scratch);
  } else if (op == L_GE_DBL_ADD && form == 1) {
    hipLaunchKernelGGL(k_debug_dbl_add_quad, dim3(qblocks), dim3(QUAD_CHAIN_BLOCK), 0, stream, out, in, n, base16);
  } else {
    if (op == L_ED_DUAL_SCALE && form == 2 && (e = hipMalloc(&scratch, n * LAYER_SCRATCH_BYTES)) != hipSuccess) return e;
    if (op == L_ED_DUAL_SCALE && form == 3 && (e = hipMalloc(&scratch, n * LAYER_TABLE_SCRATCH_BYTES)) != hipSuccess) return e;
    if (op == L_ED_DUAL_SCALE && form == 4 && (e = hipMalloc(&scratch, n * LAYER_PAIR_SCRATCH_BYTES)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_debug_layer, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, op, form, out, out_w, in, in_w, n, base16, (uint8_t*)scratch);
  }
  e = hipGetLastError();
  if (scratch) {                                   // a probe, not a hot path: wait, then release
    const hipError_t e2 = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = e2;
    (void)hipFree(scratch);                        // (teardown of a test buffer: public data, nothing to report to)
  }
  return e;
}


// ---- the exported functions (host pointers; the calling thread's current device) ----------------------------------------

namespace {
struct probe_tables { uint32_t *base16 = nullptr, *comb = nullptr, *comb_img = nullptr; };
std::mutex g_lk;
probe_tables g_tab[64];

hipError_t tables_for_current_device(probe_tables& t) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> g(g_lk);
  if (!g_tab[dev].base16) {
    probe_tables n;
    if ((e = hipMalloc((void**)&n.base16, (size_t)2 * TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS * 4)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&n.comb, (size_t)TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS * 4)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&n.comb_img, (size_t)COMB_IMG_WORDS * 4)) != hipSuccess) return e;
    const int total = 2 * TABLE_BASE16_ENTRIES + TABLE_COMB_ENTRIES;
    hipLaunchKernelGGL(k_probe_tables, dim3((total + 63) / 64), dim3(64), 0, nullptr, n.base16, n.comb);
    hipLaunchKernelGGL(k_probe_comb_image, dim3((COMB_ROWS * COMB_IMG_ENTRIES + 63) / 64), dim3(64), 0, nullptr, n.comb_img, n.comb);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    g_tab[dev] = n;
  }
  t = g_tab[dev];
  return hipSuccess;
}
}  // namespace

#define PTRY(call) do { e = (call); if (e != hipSuccess) goto out; } while (0)

extern "C" {

int eddsa_amd_probe_layer(int op, int form, uint8_t* out, size_t out_w, const uint8_t* in, size_t in_w, size_t n) {
  if (!layer_widths_ok(op, form, in_w, out_w)) return -(int)hipErrorInvalidValue;
  if (n == 0) return 0;
  if (op == L_SHA512)                               // an item's length field must fit its slot
    for (size_t i = 0; i < n; i++) {
      uint64_t len = 0;
      for (int k = 7; k >= 0; k--) len = (len << 8) | in[i * in_w + k];
      if (len > in_w - 8) return -(int)hipErrorInvalidValue;
    }
  probe_tables t;
  uint8_t *d_in = nullptr, *d_out = nullptr;
  hipError_t e = tables_for_current_device(t);
  if (e != hipSuccess) return -(int)e;
  PTRY(hipMalloc((void**)&d_in, n * in_w));
  PTRY(hipMalloc((void**)&d_out, n * out_w));
  PTRY(hipMemcpy(d_in, in, n * in_w, hipMemcpyHostToDevice));
  PTRY(hipMemset(d_out, 0, n * out_w));
  PTRY(layer_launch(op, form, d_out, out_w, d_in, in_w, n, t.base16, t.comb_img, nullptr));
  PTRY(hipStreamSynchronize(nullptr));
  PTRY(hipMemcpy(out, d_out, n * out_w, hipMemcpyDeviceToHost));
out:
  if (d_in) (void)hipFree(d_in);                    // probe buffers: public test data
  if (d_out) (void)hipFree(d_out);
  return e == hipSuccess ? 0 : -(int)e;
}

int eddsa_amd_probe_halve(uint8_t* out48, const uint8_t* t32, size_t n, int wide) {
  if (n == 0) return 0;
  uint8_t *d_t = nullptr, *d_o = nullptr;
  hipError_t e;
  PTRY(hipMalloc((void**)&d_t, n * 32));
  PTRY(hipMalloc((void**)&d_o, n * 48));
  PTRY(hipMemcpy(d_t, t32, n * 32, hipMemcpyHostToDevice));
  if (wide) hipLaunchKernelGGL(k_debug_halve<HALF_BITS_SMALL>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, nullptr, d_o, d_t, n);
  else hipLaunchKernelGGL(k_debug_halve<HALF_BITS>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, nullptr, d_o, d_t, n);
  PTRY(hipGetLastError());
  PTRY(hipMemcpy(out48, d_o, n * 48, hipMemcpyDeviceToHost));
out:
  if (d_t) (void)hipFree(d_t);
  if (d_o) (void)hipFree(d_o);
  return e == hipSuccess ? 0 : -(int)e;
}

}  // extern "C"
