// fake_kernels.cpp - TEST INFRASTRUCTURE: the launchers of libeddsa_amd/csrc/eddsa_kernels.h on the CPU, for the sanitizer
// build of the product's host side (see fake_hip.c).  Every launcher computes what its kernels compute by calling the
// -DED_HOST_CHECK build of the device source item by item (lanes.h: the same functions tests/host_check/ drives against the
// oracle, every limb bound asserted), and first checks that every pointer it was given is memory of the calling thread's
// current device and that the stream belongs to it - the rank <-> device mix-ups a one-GPU box cannot show.  The work
// itself is handed to the fake runtime as a task of the launch's stream (fake_hip_enqueue): at once in the eager model, as
// late as the API allows with FAKE_HIP_DEFER=1 - arguments that a kernel receives by value are captured at the launch,
// device memory is read when the task runs.
// Never part of the product.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include "eddsa_kernels.h"
#include "fake_hip.h"
#include "lanes.h"

namespace ed {
static std::atomic<long> g_violations{0};
void bound_violation(const char* file, int line, const char* what) {
  if (g_violations.fetch_add(1) == 0) fprintf(stderr, "fake_kernels: limb bound violated at %s:%d: %s\n", file, line, what);
}
}  // namespace ed
using namespace ed;

extern "C" long fake_kernels_bound_violations(void) { return g_violations.load(); }

static void rd(uint32_t w[8], const uint8_t* p) { memcpy(w, p, 32); }
static void wr(uint8_t* p, const uint32_t w[8]) { memcpy(p, w, 32); }

static int cur() { return fake_hip_current_device(); }
static void own(const void* p, size_t bytes, const char* what) { if (bytes) fake_hip_require_device(p, bytes, cur(), what); }
static void run_closure(void* a) { auto* f = static_cast<std::function<void()>*>(a); (*f)(); delete f; }
static void enq(hipStream_t s, std::function<void()> f) { fake_hip_enqueue(s, run_closure, new std::function<void()>(std::move(f))); }
static void own_stream(hipStream_t s) {
  if (fake_hip_stream_device(s) != cur()) { fprintf(stderr, "fake_kernels: launch on a stream of device %d while device %d is current\n", fake_hip_stream_device(s), cur()); abort(); }
}

// the generated tables, computed once per process and copied into every engine's buffers.  Same entries as
// lanes.h: table_entry_lane produces (affine niels form, canonical limbs), but k * P by repeated addition of P and one
// inversion per 64 entries (Montgomery's trick) instead of a double-and-add chain and an inversion per entry: the
// sanitizer builds run the 66 000 entries in a second instead of half a minute.  Every 997th entry is cross-checked
// against table_entry_lane itself.
static void niels_from_affine_parts(uint32_t* dst, const ge& p, const fe& zinv) {
  fe x, y, t;
  ge_niels n;
  fe_mul(x, p.X, zinv); fe_mul(y, p.Y, zinv); fe_mul(t, x, y);
  fe_sub(n.ymx, y, x); fe_canon(n.ymx, n.ymx);
  fe_add(n.ypx, y, x); fe_canon(n.ypx, n.ypx);
  fe_mul(n.t2d, t, fe_const_2d()); fe_canon(n.t2d, n.t2d);
  niels_store(dst, n);
}
static void multiples(uint32_t* dst, int count, uint32_t shift) {
  alignas(16) uint32_t one[TABLE_ENTRY_WORDS];
  table_entry_lane(one, 1, shift);               // P = 2^shift B, affine niels
  ge_niels p1;
  niels_load(p1, one);
  ge acc;
  ge_neutral(acc);
  constexpr int CH = 64;
  for (int base = 0; base < count; base += CH) {
    const int m = count - base < CH ? count - base : CH;
    ge pts[CH];
    fe pre[CH], u, zi;
    for (int j = 0; j < m; j++) {                // pts[j] = (base + j) P
      pts[j] = acc;
      if (j == 0) pre[0] = acc.Z; else fe_mul(pre[j], pre[j - 1], acc.Z);
      ge_add_niels(acc, acc, p1, true);
    }
    fe_inv(u, pre[m - 1]);
    for (int j = m - 1; j >= 0; j--) {
      if (j > 0) { fe_mul(zi, u, pre[j - 1]); fe_mul(u, u, pts[j].Z); } else zi = u;
      niels_from_affine_parts(dst + (size_t)TABLE_ENTRY_WORDS * (base + j), pts[j], zi);
    }
  }
  for (int k = 0; k < count; k += 997) {
    alignas(16) uint32_t want[TABLE_ENTRY_WORDS];
    table_entry_lane(want, (uint32_t)k, shift);
    if (memcmp(want, dst + (size_t)TABLE_ENTRY_WORDS * k, sizeof(want)) != 0) { fprintf(stderr, "fake_kernels: table entry %d (shift %u) differs from table_entry_lane\n", k, shift); abort(); }
  }
}
struct Tables {
  std::vector<uint32_t> base16, comb;
  Tables() : base16((size_t)2 * TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS + 4), comb((size_t)TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS) {
    multiples(b16(), TABLE_BASE16_ENTRIES, 0);
    multiples(b16() + (size_t)TABLE_ENTRY_WORDS * TABLE_BASE16_ENTRIES, TABLE_BASE16_ENTRIES, 128);
    for (int c = 0; c < TABLE_COMB_ENTRIES; c++)
      table_entry_lane(&comb[(size_t)TABLE_ENTRY_WORDS * c], (uint32_t)(c % COMB_HALF) + 1, 2u * COMB_W * (uint32_t)(c / COMB_HALF));
  }
  uint32_t* b16() { return reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(base16.data()) + 15) & ~(uintptr_t)15); }
  const uint32_t* b16() const { return const_cast<Tables*>(this)->b16(); }
};
static const Tables& tables() { static Tables t; return t; }

// one verify item as the default route decides it: the half-length evaluation (long items in place), the reference-order
// chain for keys that are no curve points
static int verify_item(const edk_verify_src& s, size_t i, const uint32_t* base16, int exact_offcurve) {
  alignas(16) uint32_t tab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS], rtab[VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS];
  uint32_t rw[8], sw[8], aw[8], tw[8], hd[HALF_DIGIT_WORDS];
  rd(rw, s.sigs + i * s.sig_stride); rd(sw, s.sigs + i * s.sig_stride + 32); rd(aw, s.pubs + i * s.pub_stride);
  const uint8_t* m; size_t mlen;
  if (s.msg_off) { m = s.msgs + s.msg_off[i]; mlen = (size_t)(s.msg_off[i + 1] - s.msg_off[i]); } else { m = s.msgs + i * s.msg_stride; mlen = s.msg_len; }
  uint32_t sraw[8];
  memcpy(sraw, sw, 32);
  const bool oncurve = verify_prepare_lane(tw, sw, tab, rw, aw, m, mlen);
  if (!oncurve || exact_offcurve == 2) {
    if (!exact_offcurve) return 0;
    int8_t ux[REF_JSF_LEN], uy[REF_JSF_LEN];
    uint32_t pts[160];
    return verify_exact_lane(rw, sraw, aw, m, mlen, base16 + TABLE_ENTRY_WORDS, ux, uy, pts, 1) ? 1 : 0;
  }
  verify_half_scalars_lane(hd, tw, sw);
  const bool rvalid = verify_half_point_lane(rtab, rw);
  const bool is_long = (hd[24] & 2u) != 0;
  const bool neutral = is_long ? verify_half_main_lane<true>(hd, tab, rtab, base16, true) : verify_half_main_lane<false>(hd, tab, rtab, base16, false);
  return neutral && rvalid ? 1 : 0;
}

static std::atomic<int> g_checked{0}, g_fail_at{0};

extern "C" {

int edk_fault_tick(void) {
  const int k = g_checked.fetch_add(1) + 1;
  int at = g_fail_at.load();
  return at != 0 && k == at && g_fail_at.compare_exchange_strong(at, 0);
}
void edk_debug_counting(int) {}
int edk_debug_fail_in(int nth) {
  if (nth < 0) return g_checked.load();
  g_fail_at.store(0); g_checked.store(0); g_fail_at.store(nth);
  return 0;
}

hipError_t edk_init_tables(uint32_t* base16, uint32_t* comb, uint32_t* comb_img, hipStream_t stream) {
  own_stream(stream);
  const Tables& t = tables();
  const size_t b16_bytes = (size_t)2 * TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS * 4;
  own(base16, b16_bytes, "base16"); own(comb, t.comb.size() * 4, "comb"); own(comb_img, (size_t)COMB_IMG_WORDS * 4, "comb image");
  // (the host build of comb_select reads the comb in its global layout: that is what this "image" holds here)
  static_assert((size_t)COMB_IMG_WORDS >= (size_t)TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS, "the image buffer holds the plain comb");
  enq(stream, [=, &t] {
    memcpy(base16, t.b16(), b16_bytes);
    memcpy(comb, t.comb.data(), t.comb.size() * 4);
    memset(comb_img, 0, (size_t)COMB_IMG_WORDS * 4);
    memcpy(comb_img, t.comb.data(), t.comb.size() * 4);
  });
  return hipSuccess;
}

hipError_t edk_verify(uint8_t* ok, const edk_verify_src* src, size_t n, const uint32_t* base16, const edk_verify_ws* ws,
                      hipEvent_t* marks, hipEvent_t bulk_done, int bulk_early, hipStream_t stream) {
  (void)bulk_early;
  if (n == 0) return hipSuccess;
  if (edk_fault_tick()) return hipErrorUnknown;
  own_stream(stream);
  own(ok, n, "verdicts");
  own(src->sigs, (n - 1) * src->sig_stride + 64, "signatures"); own(src->pubs, (n - 1) * src->pub_stride + 32, "keys");
  if (src->msg_off) own(src->msg_off, (n + 1) * 8, "message offsets");
  else own(src->msgs, src->msg_len ? (n - 1) * src->msg_stride + src->msg_len : 0, "messages");
  own(base16, (size_t)2 * TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS * 4, "base16");
  if (n > ws->capacity) { fprintf(stderr, "fake_kernels: pass of %zu items through a workspace of %zu\n", n, ws->capacity); abort(); }
  own(ws->digits, ws->capacity * 64, "workspace digits"); own(ws->table, ws->capacity * VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS * 4, "workspace table");
  own(ws->flags, ws->capacity, "workspace flags"); own(ws->offcount, 256, "work-list counter");
  if (fake_hip_stream_device(ws->side) != cur()) { fprintf(stderr, "fake_kernels: the workspace's side stream belongs to another device\n"); abort(); }
  // the events of kernels.hip: edk_verify, in its order
  if (marks) for (int k = 0; k < 2; k++) if (hipEventRecord(marks[k], stream) != hipSuccess) return hipErrorUnknown;
  if (bulk_done && bulk_early && hipEventRecord(bulk_done, stream) != hipSuccess) return hipErrorUnknown;
  const edk_verify_src s = *src;                 // a kernel argument: by value, at the launch
  const int exact = ws->exact_offcurve, dev = cur();
  enq(stream, [=] {
    // (the offset table is device memory: its contents exist when the task runs, not before)
    if (s.msg_off) fake_hip_require_device(s.msgs + s.msg_off[0], (size_t)(s.msg_off[n] - s.msg_off[0]) ? (size_t)(s.msg_off[n] - s.msg_off[0]) : 1, dev, "messages");
    for (size_t i = 0; i < n; i++) ok[i] = (uint8_t)verify_item(s, i, base16, exact);
  });
  if (hipEventRecord(ws->ev_prepared, stream) != hipSuccess || hipStreamWaitEvent(ws->side, ws->ev_prepared, 0) != hipSuccess ||
      hipEventRecord(ws->ev_exact, ws->side) != hipSuccess) return hipErrorUnknown;
  if (marks) for (int k = 2; k < 4; k++) if (hipEventRecord(marks[k], stream) != hipSuccess) return hipErrorUnknown;
  if (bulk_done && !bulk_early && hipEventRecord(bulk_done, stream) != hipSuccess) return hipErrorUnknown;
  if (exact && hipStreamWaitEvent(stream, ws->ev_exact, 0) != hipSuccess) return hipErrorUnknown;
  return hipSuccess;
}

size_t edk_rlc_ws_bytes(size_t capacity) { return capacity ? capacity : 0; }
size_t edk_rlc_hook_offset(size_t) { return 0; }                       // (this build's workspace is `capacity` bytes: the hook words at its start)
hipError_t edk_rlc_note_per_item(uint32_t* stats, size_t n, hipStream_t stream) {
  own_stream(stream);
  if (stats && n) { own(stats, 16, "statistics"); enq(stream, [=] { stats[1] += (uint32_t)n; stats[2] += (uint32_t)((n + 8191) / 8192); }); }
  return hipSuccess;
}
// the combination is device code (rlc.hip) that this build does not contain: the items are decided one by one, every
// group counts as "decided per item", and the second half has nothing left to do
hipError_t edk_verify_rlc(uint8_t* ok, uint32_t* stats, const edk_verify_src* src, size_t n, const uint32_t* base16,
                          const edk_verify_ws* ws, const edk_rlc_ws* rws, hipStream_t stream) {
  const hipError_t e = edk_verify(ok, src, n, base16, ws, nullptr, nullptr, 0, stream);
  if (e != hipSuccess) return e;
  void* gok = rws->host_gok;
  enq(stream, [=] { memset(gok, 1, (n + 8191) / 8192); });
  return edk_rlc_note_per_item(stats, n, stream);
}
hipError_t edk_verify_rlc_fallback(uint8_t*, const edk_verify_src*, size_t, const uint32_t*, const edk_verify_ws*, const edk_rlc_ws*, hipStream_t) {
  return hipSuccess;
}

static void own_fixed(const edk_fixed_ws* ws, size_t n, hipStream_t stream) {
  own_stream(stream);
  if (n > ws->capacity) { fprintf(stderr, "fake_kernels: pass of %zu items through a workspace of %zu\n", n, ws->capacity); abort(); }
  own(ws->acc, ws->capacity * ACC_WORDS * 4, "point workspace"); own(ws->aux, ws->capacity * 64, "scalar workspace");
}

hipError_t edk_x25519(uint8_t* out, const uint8_t* scalars, const uint8_t* points, size_t n, const edk_fixed_ws* ws, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  own_fixed(ws, n, stream); own(out, 32 * n, "x25519 out"); own(scalars, 32 * n, "scalars"); own(points, 32 * n, "points");
  enq(stream, [=] {
    for (size_t i = 0; i < n; i++) {
      uint32_t s[8], p[8], o[8];
      rd(s, scalars + 32 * i); rd(p, points + 32 * i);
      x25519_lane(o, s, p);
      wr(out + 32 * i, o);
    }
  });
  return hipSuccess;
}

hipError_t edk_genpub(uint8_t* pubs, const uint8_t* secs, size_t n, const uint32_t* comb, const edk_fixed_ws* ws, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  own_fixed(ws, n, stream); own(pubs, 32 * n, "public keys"); own(secs, 32 * n, "secret keys"); own(comb, (size_t)COMB_IMG_WORDS * 4, "comb");
  enq(stream, [=] {
    for (size_t i = 0; i < n; i++) {
      uint32_t sk[8], o[8];
      rd(sk, secs + 32 * i);
      ge A; fe zinv;
      genpub_point_lane(A, sk, comb);
      fe_inv(zinv, A.Z);
      encode_lane(o, A.X, A.Y, zinv);
      wr(pubs + 32 * i, o);
    }
  });
  return hipSuccess;
}

hipError_t edk_sign(uint8_t* sigs, const uint8_t* secs, const uint8_t* pubs, const uint8_t* msgs, const uint64_t* msg_off, const uint64_t* /*msg_end*/, size_t msg_len,
                    size_t n, const uint32_t* comb, const edk_fixed_ws* ws, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  own_fixed(ws, n, stream); own(sigs, 64 * n, "signatures"); own(secs, 32 * n, "secret keys"); own(pubs, 32 * n, "public keys");
  if (msg_off) own(msg_off, (n + 1) * 8, "message offsets"); else own(msgs, n * msg_len, "messages");
  const int dev = cur();
  enq(stream, [=] {
    if (msg_off) fake_hip_require_device(msgs + msg_off[0], (size_t)(msg_off[n] - msg_off[0]) ? (size_t)(msg_off[n] - msg_off[0]) : 1, dev, "messages");
    for (size_t i = 0; i < n; i++) {
      const uint8_t* m = msg_off ? msgs + msg_off[i] : msgs + i * msg_len;
      const size_t mlen = msg_off ? (size_t)(msg_off[i + 1] - msg_off[i]) : msg_len;
      uint32_t sk[8], pk[8], R[8], S[8], aw[8], rw[8];
      rd(sk, secs + 32 * i); rd(pk, pubs + 32 * i);
      ge Rp; fe zinv;
      sign_point_lane(Rp, aw, rw, sk, m, mlen, comb);
      fe_inv(zinv, Rp.Z);
      encode_lane(R, Rp.X, Rp.Y, zinv);
      sign_finish_lane(S, R, aw, rw, pk, m, mlen);
      wr(sigs + 64 * i, R); wr(sigs + 64 * i + 32, S);
    }
  });
  return hipSuccess;
}

hipError_t edk_x25519_base(uint8_t* out, const uint8_t* scalars, size_t n, const uint32_t* comb, const edk_fixed_ws* ws, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  own_fixed(ws, n, stream); own(out, 32 * n, "x25519_base out"); own(scalars, 32 * n, "scalars");
  enq(stream, [=] {
    for (size_t i = 0; i < n; i++) {
      uint32_t s[8], o[8];
      rd(s, scalars + 32 * i);
      ge R; fe d;
      x25519_base_point_lane(R, s, comb);
      fe_sub(d, R.Z, R.Y);
      fe_inv(d, d);
      x25519_base_finish_lane(o, R.Y, R.Z, d);
      wr(out + 32 * i, o);
    }
  });
  return hipSuccess;
}

hipError_t edk_pk_to_x(uint8_t* out, const uint8_t* in, size_t n, hipStream_t stream) {
  own_stream(stream); own(out, 32 * n, "pk->x out"); own(in, 32 * n, "pk->x in");
  enq(stream, [=] { for (size_t i = 0; i < n; i++) { uint32_t w[8], o[8]; rd(w, in + 32 * i); pk_to_x_lane(o, w); wr(out + 32 * i, o); } });
  return hipSuccess;
}
hipError_t edk_sk_to_x(uint8_t* out, const uint8_t* in, size_t n, hipStream_t stream) {
  own_stream(stream); own(out, 32 * n, "sk->x out"); own(in, 32 * n, "sk->x in");
  enq(stream, [=] { for (size_t i = 0; i < n; i++) { uint32_t w[8], o[8]; rd(w, in + 32 * i); sk_to_x_lane(o, w); wr(out + 32 * i, o); } });
  return hipSuccess;
}

}  // extern "C"
