// eddsa_kernels.h - internal interface between the C host library (eddsa_amd.c) and the HIP
// translation unit (kernels.hip).  Not installed; the public contract is include/eddsa_amd.h.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#ifndef TABLE_BASE16_ENTRIES      /* also defined, identically, by lanes.h for the device side */
#define TABLE_BASE16_ENTRIES 32769 /* k*B, k = 0..32768; the table holds twice that: k*2^128*B follows */
#define COMB_W 6                  /* signed window width of the fixed-base comb (the reference's is 4, ed.c:397-430) */
#define COMB_HALF (1 << (COMB_W - 1))          /* digits d in [-COMB_HALF, COMB_HALF - 1] */
#define COMB_DIGITS (COMB_W == 4 ? 64 : COMB_W == 5 ? 52 : 44)   /* digits of x + offset: 64 x 4, 52 x 5 or 44 x 6 bits */
#define COMB_ROWS (COMB_DIGITS / 2)            /* even digits and odd digits share a row */
#define TABLE_COMB_ENTRIES (COMB_ROWS * COMB_HALF) /* comb[i][k] = (k+1) * 2^(2*COMB_W*i) * B, k < COMB_HALF */
#define TABLE_ENTRY_WORDS 32      /* 3 x 10 limbs + 2 padding words */
#define VERIFY_TABLE_ENTRIES 9    /* 0..8 times -A, cached form */
#define VERIFY_ENTRY_WORDS 32     /* ymx | ypx | t2d | z2, 255 bits packed into eight words each: one 128-byte line */
#define COMB_IMG_ENTRIES COMB_HALF       /* LDS image of a comb row: entry m - 1 = m * 2^(2*COMB_W*i) * B, m = 1..COMB_HALF */
#define COMB_IMG_ENTRY_WORDS 36
#define COMB_IMG_WORDS (COMB_ROWS * COMB_IMG_ENTRIES * COMB_IMG_ENTRY_WORDS)
#endif
#define VERIFY_TABLE_WORDS_PER_TILE (VERIFY_TABLE_ENTRIES * VERIFY_ENTRY_WORDS * 256)
#define VERIFY_TILE 256            /* items per tile = threads per block */
#define EDK_HALF_DIGIT_WORDS 28     /* = HALF_DIGIT_WORDS of lanes.h */
#define EDK_LEN_BINS 2048           /* bins of the per-pass counting sort of ragged messages by length (kernels.hip: k_len_*) */
#define EDK_REFUSED_WORD 8
#define EDK_ONLIST_WORD 1           /* word of edk_verify_ws.offcount: the length of onlist */
#define EDK_EXACT_UNIT_WORD 2        /* word of edk_verify_ws.offcount: the next unit of work of k_verify_exact_lane_chain */
#define EDK_STALL_WORD 3             /* ... set by a wave of that kernel that gave up waiting for a hand-off: the others then leave too */
#define EDK_PASS_WORDS 4             /* words 0..3 are zeroed by every pass */
#define EDK_WITHHOLD_WORD 9          /* test hook (eddsa_amd_debug_withhold_handoff): tile + 1 whose first hand-off is never published; 0: none */
#define EDK_EXACT_PATIENCE (1u << 21) /* polls (a microsecond or two each) a wave of k_verify_exact_lane_chain spends on one hand-off */
#define EDK_BENTRY_WORD 32          /* words 32..63 of edk_verify_ws.offcount: the base point as a packed cached entry (lanes.h: exact_bentry_store) */
#define ACC_WORDS 40               /* point workspace per item: X, Y, Z and one slot for the finish kernels' prefix products */

#ifdef __cplusplus
extern "C" {
#endif

hipError_t edk_init_tables(uint32_t* base16, uint32_t* comb, uint32_t* comb_img, hipStream_t stream);

#define EDK_SUMS_MAX_ITEMS ((size_t)1 << 11)          /* passes of up to this many items add their windows' sums first (kernels.hip: k_verify_window_sums) */
#define EDK_SUMS_BYTES (EDK_SUMS_MAX_ITEMS * 64 * 40 * sizeof(uint32_t))   /* 64 windows x four multipliers of ten limbs per item */
#define EDK_STATUS_STALLED 1u
#define EDK_EXACT_SLOTS 65536                        /* items k_verify_exact_quad has in flight at once (4096 waves of 16); a longer work list is walked in strides of this */
#define EDK_EXACT_PAD_BYTES ((size_t)EDK_EXACT_SLOTS * 960)   /* per slot: four addends x five factors x 12 words (quad_lanes.h: QUAD_ITEM_WORDS) */

/* verify workspace for up to `capacity` items (a multiple of VERIFY_TILE), all in HBM */
typedef struct edk_verify_ws {
  size_t capacity;
  uint32_t* digits;   /* capacity * 16 words */
  uint32_t* table;    /* capacity / 256 tiles * VERIFY_TABLE_WORDS_PER_TILE words */
  uint32_t* acc;      /* capacity * ACC_WORDS words */
  uint32_t* hdigits;  /* capacity * EDK_HALF_DIGIT_WORDS words: the half-length scalars (kernels.hip k_verify_halve) */
  uint32_t* rtable;   /* like table: 0..8 times -R' */
  uint8_t* flags;     /* capacity bytes */
  uint32_t* offlist;  /* capacity words: the exact path's work list (keys off the curve; large passes: items without a short pair) */
  uint32_t* onlist;   /* capacity words: the items the windowed evaluation decides (every other item); the half-length route's
                         k_verify_halve / k_verify_main_half run over this list */
  uint32_t* perm;     /* capacity words: ragged passes: the items in order of message length (kernels.hip: msg_order) */
  uint32_t* lenbins;  /* 2 * EDK_LEN_BINS words: that sort's counts and cursors */
  uint32_t* offcount; /* 64 words, zeroed at allocation: [0] the length of offlist, [EDK_ONLIST_WORD] the length of onlist, [EDK_EXACT_UNIT_WORD],
                         [EDK_STALL_WORD] (all four zeroed by every pass), [EDK_REFUSED_WORD] half-length pairs that the exact check of lanes.h:
                         verify_half_scalars_lane refused since allocation (diagnostic), [EDK_WITHHOLD_WORD] (test hook), [EDK_BENTRY_WORD..] the
                         shared entry of the one-lane exact path */
  uint32_t* exact_pad;/* EDK_EXACT_PAD_BYTES: scratchpad of k_verify_exact_quad: the addends of the items in flight, one slot per quad */
  uint32_t* sums;     /* EDK_SUMS_BYTES: the windows' sums of a small pass */
  uint32_t* status;   /* one word of page-locked host memory shared by the engine's workspaces (or NULL): a kernel that had to give up
                         stores EDK_STATUS_STALLED there; the host side turns it into EDDSA_AMD_STALLED (eddsa_amd.c: take_async_error) */
  hipStream_t side;   /* the exact path runs here, beside the main kernel */
  hipEvent_t ev_prepared, ev_exact;
  int algo;           /* 0: half-length scalars (four lanes per item up to 24 576 items, one above); 1: always full-length; 2: half-length, one lane per item; 3: the mid-size arrangement below 2^18 items */
  int exact_offcurve; /* 1: replay the reference's chain for off-curve keys (default); 0: reject them; 2: replay for every item */
} edk_verify_ws;

/* where the items of a verify pass live: item i has its signature at sigs + i * sig_stride, its key
 * at pubs + i * pub_stride and its message at msgs + i * msg_stride (msg_len bytes) or, for ragged
 * messages, at msgs + msg_off[i].  Packed arrays: strides 64 / 32 / msg_len; records: one stride. */
typedef struct {
  const uint8_t *sigs, *pubs, *msgs;
  const uint64_t* msg_off;
  size_t msg_len, sig_stride, pub_stride, msg_stride;
  const uint64_t* msg_end;   /* ragged messages: the LAST entry of the call's offset table = the size of the message buffer; every
                                item's span is clamped into it (lanes.h: msg_span).  verify_on / rlc_on fill it in. */
} edk_verify_src;

/* bulk_done (or NULL): recorded on `stream` once every kernel that fills the chip has been queued, before the stream
 * waits for the exact path's side stream (bulk_early != 0: already before the main kernel, so that a following pass on
 * another workspace starts beside it) */
hipError_t edk_verify(uint8_t* ok, const edk_verify_src* src, size_t n, const uint32_t* base16,
                      const edk_verify_ws* ws, hipEvent_t* marks /* 4 events or NULL */, hipEvent_t bulk_done, int bulk_early,
                      hipStream_t stream);

/* workspace of the fixed-base operations for up to `capacity` items (a multiple of VERIFY_TILE) */
typedef struct edk_fixed_ws {
  size_t capacity;
  uint32_t* acc;      /* capacity * ACC_WORDS words: projective result, lane-interleaved per tile */
  uint32_t* aux;      /* capacity * 16 words: sign's secret scalars a, r between its two kernels (zeroed after use) */
  uint32_t* perm;     /* capacity words, and */
  uint32_t* lenbins;  /* 2 * EDK_LEN_BINS words: sign's ragged messages in order of length, as in edk_verify_ws */
  uint32_t* tiles;    /* 64 words; [0] the next 64-item tile a wave of a persistent point kernel takes (zeroed by the launcher before
                         every such launch): kernels.hip, point_tile */
} edk_fixed_ws;

/* workspace of the batch (random-linear-combination) verification for up to `capacity` items: one
 * allocation, carved up by rlc.hip */
typedef struct edk_rlc_ws {
  size_t capacity;
  void* base;
  void* host_gok;     /* pinned host memory, one byte per group of the largest pass: the group verdicts */
} edk_rlc_ws;
#define EDK_RLC_HOST_BYTES 4096
/* ragged messages: *perm = ws->perm filled with the pass's items in order of message length, or NULL (kernels.hip) */
hipError_t edk_msg_order(const uint32_t** perm, const edk_verify_ws* ws, const uint64_t* msg_off, const uint64_t* msg_end, size_t n,
                         hipStream_t stream);
size_t edk_rlc_ws_bytes(size_t capacity);
size_t edk_rlc_hook_offset(size_t capacity);   /* of the workspace's test-hook word (256 bytes to zero at allocation; rlc.hip: k_rlc_bucket) */
hipError_t edk_rlc_note_per_item(uint32_t* stats, size_t n, hipStream_t stream);
/* one pass in two halves: edk_verify_rlc enqueues the combination and the copy of the group verdicts to rws->host_gok;
 * the caller synchronises `stream`; edk_verify_rlc_fallback hands the groups that did not pass to the per-item kernels */
hipError_t edk_verify_rlc(uint8_t* ok, uint32_t* stats, const edk_verify_src* src, size_t n, const uint32_t* base16,
                          const edk_verify_ws* ws, const edk_rlc_ws* rws, hipStream_t stream);
hipError_t edk_verify_rlc_fallback(uint8_t* ok, const edk_verify_src* src, size_t n, const uint32_t* base16,
                                   const edk_verify_ws* ws, const edk_rlc_ws* rws, hipStream_t stream);

hipError_t edk_x25519(uint8_t* out, const uint8_t* scalars, const uint8_t* points, size_t n,
                      const edk_fixed_ws* ws, hipStream_t stream);
hipError_t edk_genpub(uint8_t* pubs, const uint8_t* secs, size_t n, const uint32_t* comb,
                      const edk_fixed_ws* ws, hipStream_t stream);
/* msg_end: as in edk_verify_src (NULL with msg_off == NULL) */
hipError_t edk_sign(uint8_t* sigs, const uint8_t* secs, const uint8_t* pubs, const uint8_t* msgs,
                    const uint64_t* msg_off, const uint64_t* msg_end, size_t msg_len, size_t n, const uint32_t* comb,
                    const edk_fixed_ws* ws, hipStream_t stream);
hipError_t edk_x25519_base(uint8_t* out, const uint8_t* scalars, size_t n, const uint32_t* comb,
                           const edk_fixed_ws* ws, hipStream_t stream);
/* test surface (include/eddsa_amd_debug.h).  edk_debug_fail_in: nth > 0 arms the fault (the nth checked HIP call of the
 * verify passes from now on reports hipErrorUnknown instead of being made) and restarts the count; 0 disarms; < 0 only
 * returns the number of checked calls made since the count was restarted */
int edk_debug_fail_in(int nth);
/* the checked calls are counted only while this is on (armed hooks): an unarmed process bumps no shared counter in its passes */
void edk_debug_counting(int on);
hipError_t edk_pk_to_x(uint8_t* out, const uint8_t* in, size_t n, hipStream_t stream);
hipError_t edk_sk_to_x(uint8_t* out, const uint8_t* in, size_t n, hipStream_t stream);

#ifdef __cplusplus
}
#endif
