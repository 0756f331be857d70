/*
 * engine.h - what the two C files of the host side share (not installed; the public contract is
 * include/eddsa.h + include/eddsa_amd.h).
 *
 *   eddsa_amd.c   engines (one per HIP device: tables, workspace pool), settings, the device-pointer entry
 *                 points, the multi-device device-pointer form with its RCCL gather, diagnostics
 *   host_pipe.c   everything that takes HOST pointers: the copier pool and the pinned staging lanes of the
 *                 streaming pipeline, the combiner for concurrent small calls, the host-pointer entry points
 *                 (single device and *_multi) and the thirteen eddsa.h functions (batches of one)
 *
 * Locks, always taken in this order:  g_table (rwlock: read for the duration of every call, write to create /
 * destroy engines)  ->  g_rccl_lk (the multi-device device-pointer call)  ->  engine.comb_q.lk (the combiner's
 * queue; never held while work runs)  ->  engine.pipe_lk (hands out the lanes of the host pipeline; never held while
 * work runs either: a job owns its lanes through their busy marks)  ->  engine.lk (workspace pool, profiling marks).
 */
#ifndef EDDSA_AMD_ENGINE_H
#define EDDSA_AMD_ENGINE_H

#ifndef _POSIX_C_SOURCE
#define _POSIX_C_SOURCE 200809L
#endif
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <pthread.h>
#include <stddef.h>
#include <stdint.h>

#include "eddsa.h"
#include "eddsa_amd.h"
#include "eddsa_amd_debug.h"
#include "eddsa_kernels.h"

#ifndef CHUNK_MAX                     /* (tests/fake_hip builds one variant with tiny passes: tests/c/multi_passes.c) */
#define CHUNK_MAX ((size_t)1 << 20)   /* verify items per workspace pass: 1.6 GB of HBM workspace */
#endif
#define MARK_SLOTS 256                /* profiled verify passes kept for eddsa_amd_verify_phase_ms */
#define MAX_DEVICES 64

#define ERR_NOT_GFX950 (-100000)
#define ERR_RCCL_MISSING (-100001)
#define ERR_RCCL_BASE (-200000)       /* ERR_RCCL_BASE - ncclResult_t */

#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = -(int)e_; goto out; } } while (0)

/* HIP calls on teardown and clean-up paths - frees, destroys, restoring the caller's device, waiting for what a failed
 * pass had queued - have nobody to return an error to.  They are not ignored: hip_note() counts the failures and keeps
 * the first (eddsa_amd_debug_teardown_errors; the GPU tests require zero after shutdown / re-initialisation cycles). */
void hip_note(hipError_t e, const char *what);
#define HIP_NOTE(call) hip_note((call), #call)
/* hipPointerGetAttributes / hipHostMalloc report "no such allocation" / "out of memory" through the sticky last-error
 * slot as well; when that answer has been handled, the slot is cleared so that a later hipGetLastError() of the caller's
 * does not see it */
void hip_forget_error(void);

/* Workspaces: a small pool, so that passes issued on DIFFERENT streams (host threads that each own a stream, the
 * lanes of the host pipeline) overlap on the GPU instead of queueing behind one workspace.  A stream keeps the
 * slot it used last (passes on one stream are ordered anyway, and the slot has the right size); another stream
 * takes an idle slot, or the least recently used one. */
#define VERIFY_SLOTS 4
struct vslot {
    edk_verify_ws ws;                 /* grown on demand up to CHUNK_MAX items; owns a side stream and two events */
    edk_fixed_ws fws;                 /* sign / genpub / x25519_base / x25519 workspace, grown on demand */
    edk_rlc_ws rws;                   /* batch (random-linear-combination) verification workspace */
    hipEvent_t free;                  /* recorded after the last kernel that touches ws, fws or rws */
    hipStream_t last_stream;
    unsigned long stamp;              /* for least-recently-used */
    int busy;                         /* held by a batch-verification pass that has left e->lk to wait for its stream */
};

/* host-pointer entry points: the lanes of the streaming pipeline (host_pipe.c) */
#define PIPE_MAX_IN 3
#define PIPE_LANES 3
struct lane {
    hipStream_t st;                   /* upload, kernels and download of the chunk this lane carries, in order */
    hipEvent_t kdone;                 /* the chunk's chip-filling kernels are queued up to here: the next chunk's kernels start behind it */
    void *d_in[PIPE_MAX_IN]; size_t d_in_cap[PIPE_MAX_IN];     /* HBM */
    void *d_msgs; size_t d_msgs_cap;
    void *d_out; size_t d_out_cap;
    void *h_in[PIPE_MAX_IN]; size_t h_in_cap[PIPE_MAX_IN];     /* pinned host staging */
    void *h_msgs; size_t h_msgs_cap;
    void *h_out; size_t h_out_cap;
    int busy;                         /* held by a job (under pipe_lk) */
    void *d_off; size_t d_off_cap;    /* ragged message offsets (calls of one chunk) */
    void *h_off; size_t h_off_cap;
    /* the chunk in flight on this lane: its results go from pend_dev (HBM) to pend_dst (the caller's memory), through
     * pend_via (pinned staging) unless the caller's memory is page-locked itself, once the lane's kernels are done */
    uint8_t *pend_dst, *pend_via, *pend_dev; size_t pend_bytes;
    int pend_queued;                  /* the download is already queued behind the kernels (calls of one chunk) */
    size_t used_in0;                  /* bytes of h_in[0] the chunk staged (zeroed after the chunk when they were secret) */
};
struct pipe {
    int ready;
    struct lane lane[PIPE_LANES];
    pthread_cond_t lane_cv;           /* (with engine.pipe_lk) a lane was released */
    int big_waiting;                  /* calls of several chunks waiting for all lanes: calls of one chunk queue up behind them */
    uint32_t *d_stats;                /* 4 words: statistics of the batch verification */
    uint32_t *h_stats;                /* pinned */
};

/* the combiner for concurrent small host-pointer calls (host_pipe.c): one queue, per operation a leader at a time */
struct creq;
#define COMB_KINDS 8                  /* hjob.kind: 1 verify, 2 sign, 3 x25519, 4 genpub, 5 x25519_base, 6 pk->x, 7 sk->x; 0: never combined */
struct comb_kind {
    uint32_t gen;                     /* bumped when a launch of this operation completes; its waiters sleep on it (futex) */
    int active;                       /* some thread is packing / running a launch of this operation */
    unsigned queued, last_reqs, waiting_at_end;   /* requests waiting; calls the last launch carried; requests waiting when it ended */
    void *h_in[PIPE_MAX_IN]; size_t h_in_cap[PIPE_MAX_IN];     /* pinned: the packed batch */
    void *h_msgs; size_t h_msgs_cap;
    void *h_out; size_t h_out_cap;
    uint64_t *h_off; size_t h_off_cap;
};
struct combiner {
    pthread_mutex_t lk;
    struct creq *head, *tail;
    struct comb_kind kind[COMB_KINDS];
    unsigned long batches, items;     /* statistics: combined launches, calls they carried */
};

struct engine {
    int device;
    pthread_mutex_t lk, pipe_lk;
    pthread_cond_t slot_cv;            /* signalled (under lk) when a busy workspace slot is released */
    uint32_t *base16, *comb;           /* generated base-point tables (HBM) */
    uint32_t *comb_img;                /* the comb as the point kernels stage it in LDS (lanes.h: comb_select) */
    uint32_t *status;                  /* page-locked host word the kernels report through (edk_verify_ws.status); read with take_async_error */
    struct vslot vs[VERIFY_SLOTS];
    unsigned long clock;
    int marks_used;                   /* passes recorded since profiling was switched on */
    hipEvent_t marks[MARK_SLOTS][4];
    struct pipe pipe;
    struct combiner comb_q;
};

/* the device set of the *_multi entry points (eddsa_amd_init_devices) */
typedef struct ncclComm *ncclComm_t;  /* as in rccl.h; the library is dlopen()ed on first use (it is 570 MB) */
struct multi {
    int n;
    int dev[MAX_DEVICES];
    ncclComm_t comm[MAX_DEVICES];
    void *rccl;
    int (*CommInitAll)(ncclComm_t *, int, const int *);
    int (*CommDestroy)(ncclComm_t);
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    const char *(*GetErrorString)(int);
};

extern pthread_rwlock_t g_table;
extern struct engine *g_eng[MAX_DEVICES];
extern struct multi g_multi;
extern int g_hooks_armed;            /* eddsa_amd_debug_init(.., EDDSA_AMD_TEST_HOOKS): the fault injectors and the layer probe act (atomic) */

/* Every call brackets its work with enter()/leave(): enter() resolves the engine (creating it on first use),
 * holds g_table for reading and makes the engine's device current for the calling thread; leave() restores the
 * caller's device.  device < 0: the default device. */
struct call { struct engine *e; int saved; };
int enter(struct call *c, int device);
void leave(struct call *c);

/* device buffers that held secrets (or may have) are zeroed before they go back to the allocator */
void wipe_free(void *p, size_t bytes);
/* What a kernel of an earlier or (after the stream was waited for) of this pass reported through e->status: 0, or
 * EDDSA_AMD_STALLED once per report.  Host-pointer calls ask after their last wait, so the call that stalled is the one that
 * fails; device-pointer calls return before their kernels run, so there the NEXT verify call on the engine fails in
 * its place (it is not run), as a sticky HIP error would surface. */
int take_async_error(struct engine *e);

/* device-pointer work on one engine (the engine's device is current); all asynchronous on `st` except rlc_on,
 * which waits for the stream once per pass */
/* bulk_done (or NULL): see edk_verify */
int verify_on(struct engine *e, uint8_t *ok, const edk_verify_src *all, size_t n, hipStream_t st, hipEvent_t bulk_done, int bulk_early);
int rlc_on(struct engine *e, uint8_t *ok, uint32_t *stats, const edk_verify_src *all, size_t n, hipStream_t st);
int sign_on(struct engine *e, uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
            const uint64_t *msg_off, size_t msg_len, size_t n, hipStream_t st);
int genpub_on(struct engine *e, uint8_t *pubs, const uint8_t *secs, size_t n, hipStream_t st);
int x25519_on(struct engine *e, uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n, hipStream_t st);
int xbase_on(struct engine *e, uint8_t *out, const uint8_t *scalars, size_t n, hipStream_t st);
int pk_to_x_on(struct engine *e, uint8_t *out, const uint8_t *in, size_t n, hipStream_t st);
int sk_to_x_on(struct engine *e, uint8_t *out, const uint8_t *in, size_t n, hipStream_t st);
int records_ok(size_t stride, size_t sig_off, size_t pub_off, size_t msg_off, size_t msg_len);

/* host_pipe.c, called by the engine's life cycle and diagnostics in eddsa_amd.c */
void pipe_setup(struct pipe *p);                   /* once per engine */
void pipe_release(struct pipe *p);                 /* device current, no call in flight */
void combiner_init(struct combiner *q);
void combiner_release(struct combiner *q);
int pipe_residue(struct engine *e, uint64_t *in0, uint64_t *out);   /* non-zero bytes left in the secret-bearing staging buffers (HBM and pinned host) */
int count_nonzero_dev(const void *dev, size_t bytes, uint64_t *count);
void host_pool_stop(void);

#endif /* EDDSA_AMD_ENGINE_H */
