/*
 * eddsa_amd_debug.h - the TEST and MEASUREMENT surface of libeddsa_amd.so.
 *
 * Nothing here is part of the contract a caller binds (that is include/eddsa.h, the reference's own 13 symbols,
 * lib/eddsa.h:44-113, plus the batched forms of include/eddsa_amd.h).  The functions below exist for the repository's
 * tests, bench.py and the scripts under tools/: route selection (same verdicts on every route), counters, traces, and
 * two fault injectors.  (The probes that run single layers of the device code on caller-given inputs are a library of
 * their own, libeddsa_amd_probe.so / include/eddsa_amd_probe.h: this library holds no kernel that is not a product kernel.)
 *
 * The fault injectors are INERT unless the hooks were armed with eddsa_amd_debug_init(device, EDDSA_AMD_TEST_HOOKS):
 * unarmed they return EDDSA_AMD_HOOKS_OFF and change nothing, so a stray call in a production process cannot make
 * anybody's next signature check fail.  eddsa_amd_shutdown disarms them.
 */
#ifndef EDDSA_AMD_DEBUG_H
#define EDDSA_AMD_DEBUG_H

#include "eddsa_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define EDDSA_AMD_TEST_HOOKS 1u
#define EDDSA_AMD_HOOKS_OFF (-100002)   /* a hook was called without being armed */

/* eddsa_amd_init(device), then arm (flags & EDDSA_AMD_TEST_HOOKS) or disarm (flags == 0) the hooks marked [armed] below */
EDDSA_AMD_DECL int eddsa_amd_debug_init(int device, unsigned flags);

/* ---- fault injection [armed] ---- */
/* the next host-pointer call fails with hipErrorUnknown after its inputs were staged and its kernels launched
 * (exercises the error path: the staging copies of secrets are wiped there as on success) */
EDDSA_AMD_DECL int eddsa_amd_debug_fail_next_host_call(void);
/* the nth (1-based, counted from now, over all threads) checked HIP call inside a verify pass - event records, stream
 * waits, the work-list reset, launches - reports hipErrorUnknown instead of being made; 0 disarms.  The pass must
 * return a negative value and leave the engine usable. */
EDDSA_AMD_DECL int eddsa_amd_debug_fail_hip_call(int nth);
/* The two places where device code waits for other device code.  (i) The hand-off between the waves of
 * k_verify_exact_lane_chain (csrc/kernels.hip): the first hand-off of tile `tile_plus_1 - 1` of every later pass is never
 * published, so the wave that waits for it must give up after its bound and the pass must come back as EDDSA_AMD_STALLED
 * instead of hanging.  (ii) The batch verification's Horner waves (csrc/rlc.hip: k_rlc_horner), which take the window points
 * of k_rlc_bucket as they arrive: the flag of window point `tile_plus_1 - 1` of group 0 is never raised, so that group's
 * wave must give up and leave its groups to k_rlc_final - same verdicts, no error.  0 switches both off.  Waits for the
 * device; acts on the workspaces allocated so far.  Returns a negative error, or the number of Horner waves that have given up
 * since the previous call (0 in a process whose launches overlap as they should). */
EDDSA_AMD_DECL int eddsa_amd_debug_withhold_handoff(int tile_plus_1);
/* checked HIP calls the verify passes have made since the last eddsa_amd_debug_fail_hip_call (so that a test can walk
 * nth over every one of them) */
EDDSA_AMD_DECL int eddsa_amd_debug_hip_calls(void);

/* ---- route selection (a measurement and test aid; the verdicts are the same on every route) ----
 * 0 (default): every pass checks u*(S*B - t*A - R) = 0 with half-length u, v = u*t mod 8l (132 doublings instead of
 * 252; csrc/halve.h) - passes of up to 24 576 items with four lanes per item, larger ones with one; passes of 257 ..
 * 2^19 - 1 items search u, v < 2^138 and run 35 windows, larger ones 2^134 and 34; passes of 24 577 .. 2^18 - 1 items
 * prepare with three lanes per item (the constants: csrc/kernels.hip PAIR_ONE_MIN_N, PAIR_ONE_MAX_N, HALF_WIDE_MIN_N);
 * 1: the full-length evaluation of S*B - t*A (four lanes per item up to 2^14 items); 2: the half-length one with one
 * lane per item whatever the size; 3: the arrangement of 24 577 .. 2^18 - 1 items at any size below 2^18. */
EDDSA_AMD_DECL void eddsa_amd_set_verify_algo(int algo);
/* items of the first chunk of a host-pointer call and the cap of its later chunks; 0 = the defaults: the first chunk 2^16
 * (verify) or 2^17 items, every later one twice its predecessor up to 2^20 (verify: CHUNK_MAX, one workspace pass) or 2^18 (the
 * other operations), a tail shorter than half a chunk travelling with the last one - so a verify call of 2^20 items is
 * 2^16 + 2^17 + 2^18 + the rest in one, and one of 2^22 adds chunks of 2^19 and 2^20 (a last chunk of up to 1.5 x 2^20 is two
 * passes on one workspace) */
EDDSA_AMD_DECL void eddsa_amd_set_pipeline(size_t first_chunk, size_t stage_chunk);
/* how the kernels of consecutive chunks of a host-pointer call are ordered.  -1 (default): each operation's own
 * setting; 0: side by side; 1: in chunk order; 2: in chunk order, a verify chunk starting beside the previous chunk's
 * main kernel */
EDDSA_AMD_DECL void eddsa_amd_set_pipeline_chain(int mode);

/* ---- counters, probes, traces ---- */
/* host-side time stamps of the host-pointer pipeline (csrc/host_pipe.c); on != 0 switches recording on for the calls
 * that follow; returns the stamps of the last call: tag (0 call start, 1 lane drained, 2 inputs staged and queued,
 * 3 kernels queued, 4 download queued, 5 all lanes drained, 6 call end; of a combined launch also 7 leader elected,
 * 8 callers gathered, 9 requests packed, 10 results handed back), chunk index, ms since the call started.  on = 2:
 * recording stops by itself after the 20th combined launch of 32 calls or more.  One trace per process: stamps of
 * concurrent calls interleave. */
EDDSA_AMD_DECL int eddsa_amd_debug_pipe_trace(int on, int *tags, unsigned *chunks, double *ms, int max);
/* out[0] = launches the combiner of small host-pointer calls has made on the default device, out[1] = the calls they
 * carried (equal when no two calls ever met) */
EDDSA_AMD_DECL int eddsa_amd_combiner_stats(uint64_t out[2]);
/* copy the device's generated tables out: base16 = 32769 entries k*B; comb = 704 entries comb[i][k] = (k+1)*4096^i*B,
 * i < 22, k < 32 (the reference's ed_lookup[i][k], lib/ed.c:41-43, is (k+1)*256^i*B: the same comb with 4-bit windows;
 * row 0 coincides for k < 8); each entry 32 words: 10 radix-2^25.5 limbs of y-x, y+x, 2dxy, then 2 words of padding */
EDDSA_AMD_DECL int eddsa_amd_dump_tables(uint32_t *base16_words, uint32_t *comb_words);
/* the half-length route re-verifies every pair with integers before it is used (u t = v mod 8 l) and falls back to
 * (u, v) = (1, t) when the check fails; *count = how often that has happened on the default device since its
 * workspaces were allocated.  Waits for the device.  Expected, and observed over the 2^24-item batch: 0 */
EDDSA_AMD_DECL int eddsa_amd_halve_rejected(uint64_t *count);
/* when on, HIP events are recorded on the launch stream around the kernels of every verify pass (up to 256 passes);
 * eddsa_amd_verify_phase_ms() waits for them and returns the average duration of each phase in milliseconds: out[0]
 * k_verify_prepare (+ k_verify_halve), out[1] the main kernel, out[2] k_verify_finish (the half-length route has
 * none: 0) */
EDDSA_AMD_DECL void eddsa_amd_set_profiling(int on);
EDDSA_AMD_DECL int eddsa_amd_verify_phase_ms(float out[3]);
/* HIP calls on teardown and clean-up paths (frees, destroys, restoring the caller's device, waiting for what a failed pass
 * had queued) have no caller to report to: their failures are counted instead.  Returns the count since the library was
 * loaded, *first_hip_error (may be NULL) the first hipError_t.  Expected: 0. */
EDDSA_AMD_DECL int eddsa_amd_debug_teardown_errors(int *first_hip_error);
/* for the secret-hygiene tests: waits for the default device to go idle and counts the non-zero bytes left in out[0]
 * the scalar workspace (sign's secret scalars a, r between its two kernels), out[1] the point workspace (x25519's
 * (x2 : z2); public for the other operations), out[2] the host pipeline's first input staging buffers (secret keys /
 * scalars), out[3] its output staging buffer */
EDDSA_AMD_DECL int eddsa_amd_secret_residue(uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* EDDSA_AMD_DEBUG_H */
