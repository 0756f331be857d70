/*
 * eddsa_oracle.h - CPU restatement of the libeddsa hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker.  The product (libeddsa_amd.so) never links, loads or calls it.
 *
 * Every entry point restates one function of the reference (file:line given at each
 * definition in eddsa_oracle.c).  Parity is PINNED: tests/test_oracle_vs_ref.py compares every
 * function below with the compiled reference (oracle/_ref/libeddsa_ref.so) and
 * tests/test_oracle_golden.py with the committed golden vectors under tests/golden/.
 */
#ifndef EDDSA_ORACLE_H
#define EDDSA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- single-item protocol functions: reference lib/eddsa.h:44-80 ---- */
void orc_ed25519_genpub(uint8_t pub[32], const uint8_t sec[32]);
void orc_ed25519_sign(uint8_t sig[64], const uint8_t sec[32], const uint8_t pub[32],
                      const uint8_t *data, size_t len);
int  orc_ed25519_verify(const uint8_t sig[64], const uint8_t pub[32],
                        const uint8_t *data, size_t len);
void orc_x25519_base(uint8_t out[32], const uint8_t scalar[32]);
void orc_x25519(uint8_t out[32], const uint8_t scalar[32], const uint8_t point[32]);
void orc_pk_ed25519_to_x25519(uint8_t out[32], const uint8_t in[32]);
void orc_sk_ed25519_to_x25519(uint8_t out[32], const uint8_t in[32]);

/* ---- hash: reference lib/sha512.c:127-210 (one-shot form) ---- */
void orc_sha512(uint8_t out[64], const uint8_t *data, size_t len);

/* ---- layer probes (byte in / byte out so any limb radix can be compared) ---- */
void orc_fld_mul(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);   /* fld.c:209 */
void orc_fld_sq(uint8_t out[32], const uint8_t a[32]);                          /* fld.c:249 */
void orc_fld_inv(uint8_t out[32], const uint8_t a[32]);                         /* fld.c:578 */
void orc_fld_pow2523(uint8_t out[32], const uint8_t a[32]);                     /* fld.c:657 */
void orc_sc_reduce_bytes(uint8_t out[32], const uint8_t *in, size_t len);       /* sc.c:191 + 221 */
void orc_sc_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32],
                   const uint8_t c[32]);                                         /* sc.c:241, sc.h:53 */
void orc_ed_import_export(uint8_t out[32], const uint8_t in[32]);               /* ed.c:100 + 155 */
void orc_ed_scale_base(uint8_t out[32], const uint8_t scalar[32]);              /* ed.c:397 */
void orc_ed_dual_scale(uint8_t out[32], const uint8_t s[32], const uint8_t t[32],
                       const uint8_t q[32]);                                     /* ed.c:455 */
/* the comb table the reference ships as generated data (lib/ed_lookup64.h): entry [i][k] is
 * (k+1)*16^(2i)*B as canonical bytes y-x | y+x | 2dxy, 96 bytes each, 32*8 entries. */
void orc_ed_lookup_bytes(uint8_t out[32 * 8 * 96]);

/* ---- batched forms (item-major packed arrays), threaded; the cpu_baseline leg ---- */
void orc_ed25519_verify_batch(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs,
                              const uint8_t *msgs, size_t msg_len, size_t n, int threads);
void orc_x25519_batch(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n,
                      int threads);
void orc_ed25519_sign_batch(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs,
                            const uint8_t *msgs, size_t msg_len, size_t n, int threads);
void orc_ed25519_genpub_batch(uint8_t *pubs, const uint8_t *secs, size_t n, int threads);

#ifdef __cplusplus
}
#endif
#endif
