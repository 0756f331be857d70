/*
 * eddsa_amd_probe.h - the layer probes of the device code: libeddsa_amd_probe.so, TEST INFRASTRUCTURE.
 *
 * A library of its own, built from the same device source as the product's kernels (csrc/lanes.h, quad_lanes.h,
 * fe25519.h ...) but linked into nothing a caller binds: libeddsa_amd.so exports none of these names and contains none of
 * these kernels.  The repository's GPU tests load it beside the product to run ONE layer - a field multiplication, a
 * scalar reduction, SHA-512, point import / export, the fixed-base comb, the reference-order double-scalar chain in each of
 * its forms - on caller-given inputs and compare with the golden layer vectors of the reference
 * (tests/golden/layer_kats.json).  Host pointers; runs on the calling thread's current HIP device; returns 0 or the
 * negated hipError_t.
 */
#ifndef EDDSA_AMD_PROBE_H
#define EDDSA_AMD_PROBE_H

#include <stddef.h>
#include <stdint.h>

#if defined(EDDSA_PROBE_BUILD)
#define EDDSA_PROBE_DECL __attribute__((visibility("default")))
#else
#define EDDSA_PROBE_DECL
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- layer probes: one layer of the device code on caller-given inputs (host memory), one lane per item
 *      (form 0) or, where a four-lane form exists, a quad per item exchanging operands by DPP (form 1) ----
 * in: n items of in_w bytes, out: n items of out_w bytes; the widths are fixed per op and checked.
 *   op                        in (bytes)                                   out
 *   EDL_FE_MUL                a 32 | b 32                                  32   fld_mul  reference lib/fld.c:209-244
 *   EDL_FE_SQ                 a 32                                         32   fld_sq   lib/fld.c:249-280
 *   EDL_FE_INV                a 32                                         32   fld_inv  lib/fld.c:578-645
 *   EDL_FE_POW2523            a 32                                         32   fld_pow2523 lib/fld.c:657-709
 *   EDL_FE_MUL_LOOSE          a 32 | b 32 | ka 1 | kb 1 | pad 6            32   (ka a)(kb b), ka <= 7, kb <= 3: operands at the
 *                                                                               documented limb bounds (csrc/fe25519.h:10-16)
 *   EDL_SC_REDUCE32 / 64      x 32 / x 64                                  32   sc_import lib/sc.c:191-214
 *   EDL_SC_MULADD             a 32 | b 32 | c 32                           32   a b + c mod l, lib/sc.c:241-266
 *   EDL_SHA512                len 8 (LE) | message, padded to in_w - 8     64   lib/sha512.c:127-210
 *   EDL_ED_IMPORT_EXPORT      enc 32                                       33   ed_import, ed_export lib/ed.c:100-169 | on-curve flag
 *   EDL_ED_SCALE_BASE         x 32 (reduced mod l first)                   32   ed_scale_base lib/ed.c:397-430 (comb from LDS, shuffle select)
 *   EDL_ED_DUAL_SCALE         s 32 | t 32 | q 32                           32   ed_dual_scale lib/ed.c:455-507 in the reference's order;
 *                                                                               form 0 literal, form 2 with uniform control flow,
 *                                                                               form 1 the four-lane chain (set-up + chain of the exact path),
 *                                                                               form 3 the one-lane throughput form (set-up into the item's
 *                                                                               table, the chain stretch by stretch), form 4 two items per
 *                                                                               lane (this item and the next of the batch) in units of work, as
 *                                                                               k_verify_exact_lane_chain walks them
 *   EDL_GE_DBL_ADD            p 32 | k 2 (LE) | pad 6                      32   enc(2 P + k B): form 0 ge_dbl + ge_add_niels, form 1 quad_dbl +
 *                                                                               quad_add_entry (the windowed evaluation's two steps)
 */
enum { EDL_FE_MUL = 1, EDL_FE_SQ, EDL_FE_INV, EDL_FE_POW2523, EDL_FE_MUL_LOOSE, EDL_SC_REDUCE32, EDL_SC_REDUCE64,
       EDL_SC_MULADD, EDL_SHA512, EDL_ED_IMPORT_EXPORT, EDL_ED_SCALE_BASE, EDL_ED_DUAL_SCALE, EDL_GE_DBL_ADD };
EDDSA_PROBE_DECL int eddsa_amd_probe_layer(int op, int form, uint8_t *out, size_t out_w, const uint8_t *in, size_t in_w, size_t n);


/* the device's search for the half-length pair (u, v), v = u*t mod 8l, on n given scalars t < l (32 bytes each, host
 * memory); out48 per item: v (20 bytes, little-endian) | |u| (20) | u < 0 (1) | found (1) | 6 bytes of padding.
 * wide != 0: |u|, v < 2^138 (what passes below 2^18 items use) instead of 2^134 */
EDDSA_PROBE_DECL int eddsa_amd_probe_halve(uint8_t *out48, const uint8_t *t32, size_t n, int wide);

#ifdef __cplusplus
}
#endif
#endif /* EDDSA_AMD_PROBE_H */
