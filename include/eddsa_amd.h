/*
 * eddsa_amd.h - batched entry points of the MI355X Ed25519 / X25519 engine.
 *
 * Plain C ABI (pointers and sizes only).  Every function is the batched form of one function
 * of the reference's public header and returns, item by item, exactly what a loop over the
 * reference function returns:
 *
 *   ed25519_verify_batch   loop of ed25519_verify        reference lib/eddsa.h:52
 *   ed25519_verify_records the same over fixed-size (sig, pub, msg) records
 *   ed25519_verify_batch_rlc  the same verdicts by batch verification (opt-in; the reference's TODO,
 *                          lib/ed25519-sha512.c:13-14)
 *   ed25519_sign_batch     loop of ed25519_sign          reference lib/eddsa.h:47
 *   ed25519_genpub_batch   loop of ed25519_genpub        reference lib/eddsa.h:44
 *   x25519_batch           loop of x25519                reference lib/eddsa.h:67
 *   x25519_base_batch      loop of x25519_base           reference lib/eddsa.h:64
 *   pk_ed25519_to_x25519_batch / sk_ed25519_to_x25519_batch   reference lib/eddsa.h:77, :80
 *
 * Data layout: packed, item-major, no padding: item i of a 32-byte field lives at base + 32*i
 * (signatures: base + 64*i).  Messages are either fixed-length (msg_off == NULL: item i is
 * msgs[i*msg_len .. (i+1)*msg_len)) or ragged (msg_off[0..n]: item i is
 * msgs[msg_off[i] .. msg_off[i+1])).  Verdicts are one byte per item (1 = accept, 0 = reject).
 * The offset table must not decrease, and msg_off[n] is taken for the size of the message buffer.  A host-pointer call
 * checks its table - the span msg_off[n] - msg_off[0] (at most 2^46 bytes, not negative) before anything is touched, the
 * order of the entries chunk by chunk, before each chunk's bytes are - and returns -hipErrorInvalidValue for a bad one; a
 * table that decreases further on is found when its chunk comes up, after earlier chunks have run (outputs unspecified,
 * as on every error).  A device-pointer call cannot inspect a table in HBM without a pass of its own: its kernels CLAMP
 * every item's span into [0, msg_off[n]) and to a length >= 0 instead, so whatever the table holds no lane reads outside
 * the aligned 4-byte words that hold the bytes [msgs, msgs + msg_off[n]) (the hashing kernels load whole words: up to
 * 3 bytes before an unaligned msgs and after the last byte lie in the same word as a byte of the buffer); an item with an
 * inconsistent span is hashed over the clamped one (its verdict / signature is then meaningless, like the table).
 *
 * Two flavours of every entry point:
 *   *_batch      host pointers; copies in, runs, copies out, returns when done.
 *   *_batch_dev  device pointers (HBM-resident buffers, e.g. torch tensors' data_ptr());
 *                enqueues on `stream` (a hipStream_t passed as void*, NULL = default stream)
 *                and returns without synchronising.
 *
 * Return value: 0 on success, otherwise the negated hipError_t of the failing HIP call (the
 * reference has no error channel; a loop over it cannot fail), or EDDSA_AMD_STALLED: a kernel gave up
 * waiting for a hand-off between its own waves (seconds, where the hand-off takes a millisecond: a
 * fault on the device) - the pass's outputs are incomplete.  A host-pointer call reports it itself; a
 * device-pointer call has returned by then, and the next verify call on that device reports it instead
 * (and is not run).  On error the outputs are unspecified.  Ownership: the caller owns every buffer; the library keeps no pointer after a
 * host-pointer call returns / after the stream work of a device-pointer call completes.
 * Devices: the library keeps one engine (tables, workspaces, staging pipeline) per HIP device,
 * created on first use.  A device-pointer call runs on the device that holds its output buffer; a
 * host-pointer call runs on the default device (eddsa_amd_init; otherwise the calling thread's
 * current device at the first call); the *_multi calls run on every device of the set bound by
 * eddsa_amd_init_devices, one contiguous shard each.  Every call makes its device current for its
 * own duration and restores the caller's before it returns.
 * Threading: calls may be issued from several host threads.  Device-pointer calls on different
 * streams use different workspaces (a pool of four per device) and overlap on the GPU; calls on one
 * stream are ordered by the stream.  Host-pointer calls on one device share its staging pipeline: large
 * calls run one after the other; SMALL calls (up to 64 items, fixed-length messages - among them every call
 * of the eddsa.h single-item functions) issued by several threads at once are merged: the calls queued for
 * one operation travel as ONE launch (launches of different operations side by side) and every caller gets
 * its own results back (the reference is
 * reentrant and scales with its caller's threads, lib/eddsa.h:44-80; a GPU pass costs about 0.4 ms however
 * few items it carries).  eddsa_amd_shutdown waits for calls in flight.
 * Host memory: ordinary (malloc) memory is staged through page-locked buffers by a few copier threads
 * (eddsa_amd_set_host_threads); page-locked caller memory (eddsa_amd_host_alloc, hipHostMalloc,
 * hipHostRegister) is used in place.
 * Test and measurement hooks (fault injection, route selection, probes, traces) are NOT part of this contract: they are
 * declared in eddsa_amd_debug.h and inert unless a test arms them.
 * Secrets: the staging copies of secret keys / scalars / shared secrets and the secret scalars that
 * cross kernel boundaries are zeroed in HBM before a call's stream work completes (the reference
 * wipes its stack after the same operations, lib/ed25519-sha512.c:77,136, lib/x25519.c:208,221).
 */
#ifndef EDDSA_AMD_H
#define EDDSA_AMD_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__) && defined(EDDSA_BUILD)
#define EDDSA_AMD_DECL __attribute__((visibility("default")))
#else
#define EDDSA_AMD_DECL
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Make HIP device `device` (>= 0) the default device of the host-pointer entry points and of the
 * eddsa.h single-item functions, and build its engine: the base-point tables (what the reference
 * ships as lib/ed_lookup64.h) and the workspace pool.  Idempotent; engines of other devices stay.
 * Without it the first host-pointer call adopts the calling thread's current HIP device. */
EDDSA_AMD_DECL int eddsa_amd_init(int device);
/* Release every engine on every device (waits for calls in flight); the next call builds anew. */
EDDSA_AMD_DECL void eddsa_amd_shutdown(void);

/* ---- several devices in one process (host thread or stream per device, one result
 *      gather over RCCL/xGMI; the loop being sharded is ed25519_verify, lib/ed25519-sha512.c:148-181) ----
 * Bind the device set of the *_multi entry points: an engine on each device of `devices` (n entries,
 * no duplicates; NULL / 0 = every visible device) and, for the device-pointer form's result gather,
 * one RCCL communicator per device (ncclCommInitAll; librccl.so.1 is loaded here, not at link time).
 * Device d of the set owns items [lo, hi) = eddsa_amd_shard_bounds(n, d, count): contiguous,
 * balanced (shards differ by at most one item). */
EDDSA_AMD_DECL int eddsa_amd_init_devices(const int *devices, int n);
EDDSA_AMD_DECL int eddsa_amd_device_count(void);
EDDSA_AMD_DECL int eddsa_amd_device_at(int index);   /* the HIP device that owns shard `index` of the set, or -1 */
EDDSA_AMD_DECL void eddsa_amd_shard_bounds(size_t n, int rank, int world, size_t *lo, size_t *hi);
/* host pointers, whole batch in host memory: one host thread per device runs that device's streaming
 * pipeline on its shard and copies the results straight into the caller's buffer (no collective) */
EDDSA_AMD_DECL int ed25519_verify_batch_multi(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs,
                                              const uint8_t *msgs, const uint64_t *msg_off,
                                              size_t msg_len, size_t n);
EDDSA_AMD_DECL int ed25519_sign_batch_multi(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs,
                                            const uint8_t *msgs, const uint64_t *msg_off,
                                            size_t msg_len, size_t n);
EDDSA_AMD_DECL int x25519_batch_multi(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n);
/* device pointers: sigs[d] / pubs[d] / msgs[d] (fixed msg_len) hold shard d in device d's HBM,
 * ok_full[d] is a buffer of n_total bytes there, streams[d] a stream of device d (NULL entries, or
 * streams == NULL = default stream).  Shard d is verified on device d into its slice of ok_full[d]; then ONE RCCL
 * all-gather (grouped broadcasts when the shards differ in length) leaves the whole verdict vector
 * in every ok_full[d].  Enqueues and returns; the streams order the work. */
EDDSA_AMD_DECL int ed25519_verify_batch_multi_dev(uint8_t *const ok_full[], const uint8_t *const sigs[],
                                                  const uint8_t *const pubs[], const uint8_t *const msgs[],
                                                  size_t msg_len, size_t n_total, void *const streams[]);
/* Page-locked host memory for the arrays of the host-pointer entry points: such buffers are read and written by the DMA
 * engines in place, without the staging copy ordinary memory needs.  NULL when the allocation fails. */
EDDSA_AMD_DECL void *eddsa_amd_host_alloc(size_t bytes);
EDDSA_AMD_DECL void eddsa_amd_host_free(void *p);
/* helper threads that copy ordinary caller memory into the staging buffers beside the calling thread (default 6, at
 * most 16; 0 = the caller copies alone) */
EDDSA_AMD_DECL void eddsa_amd_set_host_threads(int n);
#define EDDSA_AMD_STALLED (-100003)
/* human-readable text for a negative return value */
EDDSA_AMD_DECL const char *eddsa_amd_strerror(int err);
/* How ed25519_verify* treats a public key that does not decode to a curve point (the reference's
 * ed_import never fails, lib/ed.c:100-149).  exact != 0 (default): such items are evaluated in the
 * reference's own order of operations -- the only way to reproduce its bytes for them.  exact == 0:
 * they are rejected outright, which differs from the reference only on a SHA-512 fixed point and
 * saves about 1 ms per pass that contains such keys.  exact == 2: EVERY item is evaluated in the
 * reference's order of operations (JSF/Shamir chain, lib/ed.c:455-507) and the windowed kernel's
 * result is ignored -- same verdicts, latency-bound, meant for self-checks. */
EDDSA_AMD_DECL void eddsa_amd_set_offcurve_mode(int exact);
EDDSA_AMD_DECL void eddsa_amd_set_rlc_min_items(size_t items);   /* see ed25519_verify_batch_rlc */
/* ---- host-pointer entry points ---- */
EDDSA_AMD_DECL int ed25519_verify_batch(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs,
                                        const uint8_t *msgs, const uint64_t *msg_off, size_t msg_len,
                                        size_t n);
/* The same loop of ed25519_verify (reference lib/eddsa.h:52) over n fixed-size RECORDS, for callers
 * that hold (signature, public key, message) together: record i starts at records + i * stride and
 * has its 64-byte signature at sig_off, its 32-byte key at pub_off and its msg_len-byte message at
 * msg_off (e.g. stride 128 = sig | pub | 32-byte digest).  Fields may overlap nothing in particular
 * and need no alignment; every field must lie inside the record, otherwise hipErrorInvalidValue is
 * returned (negated).  One upload instead of three on the host path. */
EDDSA_AMD_DECL int ed25519_verify_records(uint8_t *ok, const uint8_t *records, size_t stride,
                                          size_t sig_off, size_t pub_off, size_t msg_off,
                                          size_t msg_len, size_t n);
/* OPT-IN batch verification by random linear combination: the reference's own TODO
 * (lib/ed25519-sha512.c:13-14, "batch verification"); same arguments and verdict bytes as
 * ed25519_verify_batch, about twice its rate on batches whose signatures are (nearly) all valid.
 * Groups of 8192 items are checked as  (sum z_i S_i) B - sum (z_i t_i) A_i - sum z_i R_i = 0  with
 * 126-bit odd coefficients z_i derived from a SHA-512 tree over the whole batch; a group that fails, or
 * that contains a key that is no curve point or a key / R of small order, is decided item by item by the
 * ordinary kernels, and an R that is not a canonical point encoding is rejected at once.  The verdicts
 * equal ed25519_verify_batch's whenever every A and R lies in the prime-order subgroup (all honestly
 * generated keys and signatures) -- up to a 2^-125 chance per group; crafted inputs with small-order
 * COMPONENTS in several items of one group can make the group pass although the reference's
 * cofactorless check rejects one of them.  That is why this is never the default.
 * stats (4 words, may be NULL) receives: items decided by the combination, items decided per item,
 * groups sent to the per-item kernels, groups decided by the combination.
 * The device-pointer form synchronises `stream` once per pass (it reads the group verdicts).
 * The combination has about 1 ms of latency of its own (hash tree over the batch, one serial Horner per
 * group), so it pays from about 2^17 items (x1.5 at 2^18, x2.1 at 2^19, x2.2 from 2^20): calls with fewer items than
 * eddsa_amd_set_rlc_min_items (default EDDSA_AMD_RLC_MIN_ITEMS_DEFAULT; 0 = always combine) go straight to the per-item
 * kernels. */
#define EDDSA_AMD_RLC_MIN_ITEMS_DEFAULT ((size_t)5 << 15)
EDDSA_AMD_DECL int ed25519_verify_batch_rlc(uint8_t *ok, uint32_t stats[4], const uint8_t *sigs,
                                            const uint8_t *pubs, const uint8_t *msgs,
                                            const uint64_t *msg_off, size_t msg_len, size_t n);
EDDSA_AMD_DECL int ed25519_sign_batch(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs,
                                      const uint8_t *msgs, const uint64_t *msg_off, size_t msg_len,
                                      size_t n);
EDDSA_AMD_DECL int ed25519_genpub_batch(uint8_t *pubs, const uint8_t *secs, size_t n);
EDDSA_AMD_DECL int x25519_batch(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n);
EDDSA_AMD_DECL int x25519_base_batch(uint8_t *out, const uint8_t *scalars, size_t n);
EDDSA_AMD_DECL int pk_ed25519_to_x25519_batch(uint8_t *out, const uint8_t *in, size_t n);
EDDSA_AMD_DECL int sk_ed25519_to_x25519_batch(uint8_t *out, const uint8_t *in, size_t n);

/* ---- device-pointer entry points (asynchronous on `stream`) ---- */
EDDSA_AMD_DECL int ed25519_verify_batch_dev(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs,
                                            const uint8_t *msgs, const uint64_t *msg_off,
                                            size_t msg_len, size_t n, void *stream);
EDDSA_AMD_DECL int ed25519_verify_records_dev(uint8_t *ok, const uint8_t *records, size_t stride,
                                              size_t sig_off, size_t pub_off, size_t msg_off,
                                              size_t msg_len, size_t n, void *stream);
EDDSA_AMD_DECL int ed25519_verify_batch_rlc_dev(uint8_t *ok, uint32_t *stats /* device, 4 words, or NULL */,
                                                const uint8_t *sigs, const uint8_t *pubs,
                                                const uint8_t *msgs, const uint64_t *msg_off,
                                                size_t msg_len, size_t n, void *stream);
EDDSA_AMD_DECL int ed25519_sign_batch_dev(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs,
                                          const uint8_t *msgs, const uint64_t *msg_off,
                                          size_t msg_len, size_t n, void *stream);
EDDSA_AMD_DECL int ed25519_genpub_batch_dev(uint8_t *pubs, const uint8_t *secs, size_t n, void *stream);
EDDSA_AMD_DECL int x25519_batch_dev(uint8_t *out, const uint8_t *scalars, const uint8_t *points,
                                    size_t n, void *stream);
EDDSA_AMD_DECL int x25519_base_batch_dev(uint8_t *out, const uint8_t *scalars, size_t n, void *stream);
EDDSA_AMD_DECL int pk_ed25519_to_x25519_batch_dev(uint8_t *out, const uint8_t *in, size_t n, void *stream);
EDDSA_AMD_DECL int sk_ed25519_to_x25519_batch_dev(uint8_t *out, const uint8_t *in, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EDDSA_AMD_H */
