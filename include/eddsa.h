/*
 * eddsa.h - the libeddsa function surface, served by the MI355X engine (libeddsa_amd.so).
 *
 * Drop-in for the reference's public header (reference lib/eddsa.h:1-122): the same thirteen
 * exported C symbols with the same argument meaning and byte-level behaviour.  Each call is a
 * batch of one on the GPU (see eddsa_amd.h for the batched entry points, which is what a caller
 * that cares about throughput should bind); there is NO CPU implementation behind these symbols.
 * If no usable gfx950 device is present the library reports the HIP error on stderr and aborts:
 * the reference's signatures have no error channel (reference lib/eddsa.h:44-80 are void / bool).
 *
 * Behaviour pinned by the reference and reproduced bit for bit (tests/golden/):
 *   - ed25519_verify is the permissive cofactorless encode-and-compare check
 *     (reference lib/ed25519-sha512.c:148-181): S is reduced mod l, not range-checked; the
 *     public key is decoded without any validity check; R is compared as bytes.
 *   - x25519 does not mask bit 255 of the input u-coordinate (reference lib/fld.c:137-156,
 *     lib/x25519.c:142); its own table (reference test/x25519-table.h) pins that.
 */
#ifndef EDDSA_H
#define EDDSA_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__) && defined(EDDSA_BUILD)
#define EDDSA_DECL __attribute__((visibility("default")))
#else
#define EDDSA_DECL
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define ED25519_KEY_LEN 32
#define ED25519_SIG_LEN 64
#define X25519_KEY_LEN 32

/* reference lib/eddsa.h:44 */
EDDSA_DECL void ed25519_genpub(uint8_t pub[ED25519_KEY_LEN], const uint8_t sec[ED25519_KEY_LEN]);
/* reference lib/eddsa.h:47 -- the caller supplies pub; it is hashed, not checked */
EDDSA_DECL void ed25519_sign(uint8_t sig[ED25519_SIG_LEN], const uint8_t sec[ED25519_KEY_LEN],
                             const uint8_t pub[ED25519_KEY_LEN], const uint8_t *data, size_t len);
/* reference lib/eddsa.h:52 */
EDDSA_DECL bool ed25519_verify(const uint8_t sig[ED25519_SIG_LEN], const uint8_t pub[ED25519_KEY_LEN],
                               const uint8_t *data, size_t len);

/* reference lib/eddsa.h:64 and :67 */
EDDSA_DECL void x25519_base(uint8_t out[X25519_KEY_LEN], const uint8_t scalar[X25519_KEY_LEN]);
EDDSA_DECL void x25519(uint8_t out[X25519_KEY_LEN], const uint8_t scalar[X25519_KEY_LEN],
                       const uint8_t point[X25519_KEY_LEN]);

/* reference lib/eddsa.h:77 and :80 */
EDDSA_DECL void pk_ed25519_to_x25519(uint8_t out[X25519_KEY_LEN], const uint8_t in[ED25519_KEY_LEN]);
EDDSA_DECL void sk_ed25519_to_x25519(uint8_t out[X25519_KEY_LEN], const uint8_t in[ED25519_KEY_LEN]);

/* obsolete aliases kept by the reference (lib/eddsa.h:92-113); same semantics as above.
 * NOTE: `DH` collides with OpenSSL's typedef of the same name; do not include both headers. */
EDDSA_DECL void eddsa_genpub(uint8_t pub[32], const uint8_t sec[32]);
EDDSA_DECL void eddsa_sign(uint8_t sig[64], const uint8_t sec[32], const uint8_t pub[32],
                           const uint8_t *data, size_t len);
EDDSA_DECL bool eddsa_verify(const uint8_t sig[64], const uint8_t pub[32], const uint8_t *data, size_t len);
EDDSA_DECL void DH(uint8_t out[32], const uint8_t sec[32], const uint8_t point[32]);
EDDSA_DECL void eddsa_pk_eddsa_to_dh(uint8_t out[32], const uint8_t in[32]);
EDDSA_DECL void eddsa_sk_eddsa_to_dh(uint8_t out[32], const uint8_t in[32]);

#ifdef __cplusplus
}
#endif
#endif /* EDDSA_H */
