/*
 * host_pipe.c - everything that takes HOST pointers, plain C.
 *
 * The reference's functions take ordinary (malloc) memory, are reentrant and scale with the caller's threads
 * (reference lib/eddsa.h:44-80).  Three pieces give the GPU engine the same manners:
 *
 *   1. A streaming pipeline over PIPE_LANES lanes.  A lane carries one chunk of the batch at a time on its own
 *      stream - upload, kernels, download, in order - so up to three chunks are in flight: one uploading, one
 *      computing, one downloading, and the kernels of consecutive chunks overlap on the GPU (each lane's stream
 *      draws its own workspace from the engine's pool).  Chunks start small (2^16 or 2^17 items: the GPU is working
 *      0.2 ms after the call) and double up to the job's stage size.
 *   2. Pinned staging.  hipMemcpyAsync from pageable memory is staged by the HIP runtime on one thread at about
 *      11 GB/s and blocks the caller (round 2: 86.8 M verifies/s host to host against 109.5 kernel-only, sign 113
 *      against 182 from pinned memory).  Here every lane owns pinned staging buffers; caller memory is copied into
 *      them by a small pool of copier threads, piece by piece, each piece handed to the DMA engine as soon as it is
 *      staged; results come back the same way.  Caller memory that is already pinned (eddsa_amd_host_alloc,
 *      hipHostMalloc, hipHostRegister) is used in place.
 *   3. A combiner for concurrent small calls (flat combining).  A GPU pass costs about 0.4 ms however few items it
 *      carries, so T threads that each loop over ed25519_verify would get 1 / 0.4 ms verifies per second IN TOTAL
 *      if their calls ran one after the other.  Instead a small call queues its request; one of the waiting
 *      threads becomes the leader of its operation, packs everything queued for that operation into one batch, runs
 *      it as a single pipeline job (one lane: the launches of different operations overlap) and hands the results back.
 *
 * Jobs that carry secrets (secret keys, scalars, shared secrets) zero their staging copies - in HBM, in the
 * pinned host buffers and in the combiner's buffers - before the call returns, on the error path too (the
 * reference wipes its stack after the same operations: lib/ed25519-sha512.c:77,136, lib/x25519.c:208,221).
 */
#define _GNU_SOURCE                    /* syscall(): the combiner's waiters sleep on a futex */
#include "engine.h"

#include <limits.h>
#include <linux/futex.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------------------------------
 * copier pool: memcpy / memset of large host ranges on several threads
 * ---------------------------------------------------------------------------------------- */

#define POOL_MAX_THREADS 16
#define POOL_QUEUE 256
#define POOL_MIN_SLICE ((size_t)1 << 20)     /* below 2 slices of this the caller copies by itself */

struct ptask { uint8_t *dst; const uint8_t *src; size_t bytes; int *pending; };   /* src == NULL: zero fill */

static struct {
    pthread_mutex_t lk;
    pthread_cond_t work_cv, done_cv;
    pthread_t th[POOL_MAX_THREADS];
    int started, want, stop;
    struct ptask q[POOL_QUEUE];
    unsigned head, tail;                      /* tasks q[head % POOL_QUEUE .. tail) */
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 6, 0, {{0}}, 0, 0 };

static void ptask_run(const struct ptask *t)
{
    if (t->src) memcpy(t->dst, t->src, t->bytes); else memset(t->dst, 0, t->bytes);
}

static void *pool_worker(void *arg)
{
    (void)arg;
    pthread_mutex_lock(&g_pool.lk);
    for (;;) {
        while (!g_pool.stop && g_pool.head == g_pool.tail) pthread_cond_wait(&g_pool.work_cv, &g_pool.lk);
        if (g_pool.head == g_pool.tail) break;   /* stopping, and nothing queued is left undone */
        struct ptask t = g_pool.q[g_pool.head++ % POOL_QUEUE];
        pthread_mutex_unlock(&g_pool.lk);
        ptask_run(&t);
        pthread_mutex_lock(&g_pool.lk);
        if (--*t.pending == 0) pthread_cond_broadcast(&g_pool.done_cv);
    }
    pthread_mutex_unlock(&g_pool.lk);
    return NULL;
}

/* helper threads beside the calling thread (default 6; 0 = the caller copies alone).  Takes effect for the threads
 * not yet started; eddsa_amd_shutdown stops the pool, the next large call starts it again. */
void eddsa_amd_set_host_threads(int n)
{
    pthread_mutex_lock(&g_pool.lk);
    g_pool.want = n < 0 ? 0 : n > POOL_MAX_THREADS ? POOL_MAX_THREADS : n;
    pthread_mutex_unlock(&g_pool.lk);
}

/* (two threads may shut the library down at once: the second waits for the first one's joins instead of joining the
 * same threads again) */
void host_pool_stop(void)
{
    pthread_mutex_lock(&g_pool.lk);
    while (g_pool.stop) pthread_cond_wait(&g_pool.done_cv, &g_pool.lk);
    const int n = g_pool.started;
    g_pool.stop = 1;
    pthread_cond_broadcast(&g_pool.work_cv);
    pthread_mutex_unlock(&g_pool.lk);
    for (int i = 0; i < n; i++) pthread_join(g_pool.th[i], NULL);
    pthread_mutex_lock(&g_pool.lk);
    g_pool.started = 0;
    g_pool.stop = 0;
    pthread_cond_broadcast(&g_pool.done_cv);
    pthread_mutex_unlock(&g_pool.lk);
}

/* Queue dst[0..bytes) = src[0..bytes) (src == NULL: zeros) for the pool, in slices; *pending counts the slices still
 * to be done (read and written under the pool's lock only; pool_wait).  What the pool cannot take - no helper threads,
 * no room in the queue - is done here and now.  Returns the bytes queued. */
static size_t pool_submit(int *pending, uint8_t *d, const uint8_t *s, size_t bytes, size_t slice)
{
    size_t off = 0;
    if (bytes >= POOL_MIN_SLICE) {
        pthread_mutex_lock(&g_pool.lk);
        while (g_pool.started < g_pool.want && !g_pool.stop) {
            if (pthread_create(&g_pool.th[g_pool.started], NULL, pool_worker, NULL) != 0) break;
            g_pool.started++;
        }
        if (g_pool.started > 0 && !g_pool.stop) {   /* (a pool that is being stopped takes no new work: done here instead) */
            if (slice == 0) {
                size_t parts = bytes / POOL_MIN_SLICE;
                if (parts > (size_t)g_pool.started) parts = (size_t)g_pool.started;
                slice = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
            }
            int added = 0;
            while (off < bytes && g_pool.tail - g_pool.head < POOL_QUEUE) {
                const size_t b = bytes - off < slice ? bytes - off : slice;
                const struct ptask t = { d + off, s ? s + off : NULL, b, pending };
                g_pool.q[g_pool.tail++ % POOL_QUEUE] = t;
                ++*pending;
                added++;
                off += b;
            }
            if (added) pthread_cond_broadcast(&g_pool.work_cv);
        }
        pthread_mutex_unlock(&g_pool.lk);
    }
    if (off < bytes) { const struct ptask t = { d + off, s ? s + off : NULL, bytes - off, NULL }; ptask_run(&t); }
    return off;
}

static void pool_wait(int *pending)
{
    pthread_mutex_lock(&g_pool.lk);
    while (*pending) pthread_cond_wait(&g_pool.done_cv, &g_pool.lk);
    pthread_mutex_unlock(&g_pool.lk);
}

/* dst[0..bytes) = src[0..bytes) (src == NULL: zeros) on the caller and the pool together; returns when all of it is done */
static void par_copy(void *dst, const void *src, size_t bytes)
{
    uint8_t *d = (uint8_t *)dst;
    const uint8_t *s = (const uint8_t *)src;
    int pending = 0;
    size_t own = bytes;
    if (bytes >= 2 * POOL_MIN_SLICE) {
        pthread_mutex_lock(&g_pool.lk);
        const size_t helpers = (size_t)(g_pool.started > g_pool.want ? g_pool.started : g_pool.want);
        pthread_mutex_unlock(&g_pool.lk);
        size_t parts = bytes / POOL_MIN_SLICE;
        if (parts > helpers + 1) parts = helpers + 1;
        const size_t slice = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        own = slice < bytes ? slice : bytes;
        if (own < bytes) pool_submit(&pending, d + own, s ? s + own : NULL, bytes - own, slice);
    }
    if (own) { const struct ptask t = { d, s, own, NULL }; ptask_run(&t); }
    if (own < bytes) pool_wait(&pending);
}

/* ------------------------------------------------------------------------------------------
 * pinned memory for callers
 * ---------------------------------------------------------------------------------------- */

/* Page-locked host memory: buffers from here are used in place by the host-pointer entry points (no staging copy).
 * NULL when the allocation fails.  Freed by eddsa_amd_host_free only. */
void *eddsa_amd_host_alloc(size_t bytes)
{
    struct call c;
    void *p = NULL;
    if (enter(&c, -1)) return NULL;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { hip_forget_error(); p = NULL; }
    leave(&c);
    return p;
}

void eddsa_amd_host_free(void *p)
{
    if (p) HIP_NOTE(hipHostFree(p));
}

/* is [p, p + bytes) page-locked memory the DMA engines can read directly?  BOTH ends are asked (an array whose head was
 * registered and whose tail is pageable - a view that runs past a registered window - is staged like ordinary memory);
 * the runtime has no query for "one allocation", so a range glued together from two registrations passes: each of its
 * pages is page-locked, which is what the copy needs */
static int is_pinned(const void *p, size_t bytes)
{
    hipPointerAttribute_t a, b;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { hip_forget_error(); return 0; }   /* "not registered" is an answer, not a failure */
    if (a.type != hipMemoryTypeHost) return 0;
    if (bytes <= 1) return 1;
    if (hipPointerGetAttributes(&b, (const uint8_t *)p + bytes - 1) != hipSuccess) { hip_forget_error(); return 0; }
    return b.type == hipMemoryTypeHost;
}

/* ------------------------------------------------------------------------------------------
 * the pipeline
 * ---------------------------------------------------------------------------------------- */

#define PIPE_FIRST_CHUNK ((size_t)1 << 17)   /* the first chunk of a call (512 blocks: the chip is full 0.3 ms after the call); later ones double up to the job's stage size */
#define PIPE_CHUNK ((size_t)1 << 18)         /* stage size: 1024 blocks of 256 lanes */
#define PIPE_PIECE ((size_t)8 << 20)         /* staging granularity: a piece is handed to the DMA engine while the next is copied */
#define PIN_CHECK_MIN ((size_t)1 << 20)      /* smaller arrays are staged without asking whether they are pinned */

static size_t g_pipe_first, g_pipe_stage;   /* eddsa_amd_set_pipeline: 0 = the defaults above / the job's own stage size */
static int g_pipe_chain = -1;               /* -1: the job's own; 0: the lanes' kernels run side by side; 1: in chunk order; 2: verify's next chunk starts beside the main kernel */

#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
void eddsa_amd_set_pipeline_chain(int mode)
{
    pthread_rwlock_wrlock(&g_table);
    g_pipe_chain = mode;
    pthread_rwlock_unlock(&g_table);
}
#endif

/* tuning: items of the first chunk of a host-pointer call and of its later stages (0 = default).  A measurement aid. */
#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
void eddsa_amd_set_pipeline(size_t first_chunk, size_t stage_chunk)
{
    pthread_rwlock_wrlock(&g_table);
    g_pipe_first = first_chunk;
    g_pipe_stage = stage_chunk;
    pthread_rwlock_unlock(&g_table);
}
#endif

enum { WIPE_NONE = 0, WIPE_IN0 = 1, WIPE_OUT = 2 };   /* which staging buffers held secrets */

struct hjob {
    int n_in; const uint8_t *in[PIPE_MAX_IN]; size_t in_w[PIPE_MAX_IN];   /* fixed-width inputs */
    int has_msgs; const uint8_t *msgs; const uint64_t *msg_off; size_t msg_len;
    uint8_t *out; size_t out_w;
    int (*run)(struct engine *e, const struct hjob *j, uint8_t *d_out, uint8_t *const d_in[PIPE_MAX_IN],
               const uint8_t *d_msgs, const uint64_t *d_off, size_t msg_len, size_t m, hipStream_t st, hipEvent_t kdone);
    size_t rec_sig, rec_pub, rec_msg;          /* records: offsets inside in[0]'s items (in_w[0] = stride) */
    size_t chunk;                              /* items per pipeline stage (0: PIPE_CHUNK) */
    int wipe;                                  /* WIPE_* */
    uint32_t *stats;                           /* rlc: host copy of the pass statistics (4 words) or NULL */
    size_t first_chunk;                        /* items of the first chunk (0: PIPE_FIRST_CHUNK); later ones double up to `chunk` */
    int chain;                                 /* the kernels of consecutive chunks run in chunk order (else side by side) */
    int kind;                                  /* combiner slot (engine.h: COMB_KINDS): small calls of this operation may be merged; 0: never */
    int src_pinned;                            /* every host array of the job is page-locked: no staging */
    int traced;                                /* the caller (the combiner's leader) has started the call's trace already */
};

/* (new memory starts zeroed, on the stream that will use it: what the allocator hands out is whatever a freed buffer held,
 * and the residue check of the tests - pipe_residue - looks at whole buffers) */
static int dev_grow(void **buf, size_t *cap, size_t need, hipStream_t st)
{
    if (need <= *cap) return 0;
    if (*buf) { wipe_free(*buf, *cap); *buf = NULL; *cap = 0; }
    need = need < 256 ? 256 : need;
    hipError_t e = hipMalloc(buf, need);
    if (e != hipSuccess) return -(int)e;
    e = hipMemsetAsync(*buf, 0, need, st);
    if (e != hipSuccess) { HIP_NOTE(hipFree(*buf)); *buf = NULL; return -(int)e; }
    *cap = need;
    return 0;
}

static void host_free_wiped(void **buf, size_t *cap)
{
    if (*buf) { memset(*buf, 0, *cap); HIP_NOTE(hipHostFree(*buf)); }
    *buf = NULL; *cap = 0;
}

static int host_grow(void **buf, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    host_free_wiped(buf, cap);
    need = need < 4096 ? 4096 : need;
    hipError_t e = hipHostMalloc(buf, need, hipHostMallocDefault);
    if (e != hipSuccess) { *buf = NULL; return -(int)e; }
    memset(*buf, 0, need);
    *cap = need;
    return 0;
}

static int pipe_init(struct pipe *p)
{
    int rc = 0;
    if (p->ready) return 0;
    for (int l = 0; l < PIPE_LANES; l++) {
        /* (a call that failed half-way through here is retried by the next one: what exists is kept) */
        if (!p->lane[l].st) TRY(hipStreamCreateWithFlags(&p->lane[l].st, hipStreamNonBlocking));
        if (!p->lane[l].kdone) TRY(hipEventCreateWithFlags(&p->lane[l].kdone, hipEventDisableTiming));
    }
    if (!p->d_stats) TRY(hipMalloc((void **)&p->d_stats, 256));
    if (!p->h_stats) TRY(hipHostMalloc((void **)&p->h_stats, 256, hipHostMallocDefault));
    p->ready = 1;
out:
    return rc;
}

void pipe_setup(struct pipe *p)
{
    pthread_cond_init(&p->lane_cv, NULL);
}

void pipe_release(struct pipe *p)
{
    pthread_cond_destroy(&p->lane_cv);
    for (int l = 0; l < PIPE_LANES; l++) {
        struct lane *L = &p->lane[l];
        for (int i = 0; i < PIPE_MAX_IN; i++) { wipe_free(L->d_in[i], L->d_in_cap[i]); host_free_wiped(&L->h_in[i], &L->h_in_cap[i]); }
        wipe_free(L->d_msgs, 0); host_free_wiped(&L->h_msgs, &L->h_msgs_cap);
        wipe_free(L->d_out, L->d_out_cap); host_free_wiped(&L->h_out, &L->h_out_cap);
        wipe_free(L->d_off, 0); host_free_wiped(&L->h_off, &L->h_off_cap);
        if (L->st) HIP_NOTE(hipStreamDestroy(L->st));
        if (L->kdone) HIP_NOTE(hipEventDestroy(L->kdone));
    }
    if (p->d_stats) HIP_NOTE(hipFree(p->d_stats));
    if (p->h_stats) HIP_NOTE(hipHostFree(p->h_stats));
    memset(p, 0, sizeof(*p));
}

/* Lanes are handed out under pipe_lk and owned through their busy marks, so that the lock is never held while work
 * runs: a call of ONE chunk takes any free lane - three such calls (the combined launches of three operations, say)
 * are in flight side by side -, a call of several chunks takes all three; while one of those waits, no new one-chunk
 * call starts.  Returns the first lane of the job (0 when it has them all), or a negative error. */
static int lanes_acquire(struct engine *e, int all)
{
    struct pipe *p = &e->pipe;
    int got = -1;
    pthread_mutex_lock(&e->pipe_lk);
    int rc = pipe_init(p);
    if (rc) { pthread_mutex_unlock(&e->pipe_lk); return rc; }
    if (all) {
        p->big_waiting++;
        while (p->lane[0].busy || p->lane[1].busy || p->lane[2].busy) pthread_cond_wait(&p->lane_cv, &e->pipe_lk);
        p->big_waiting--;
        for (int l = 0; l < PIPE_LANES; l++) p->lane[l].busy = 1;
        got = 0;
    } else {
        for (;;) {
            for (int l = 0; l < PIPE_LANES && got < 0 && !p->big_waiting; l++) if (!p->lane[l].busy) got = l;
            if (got >= 0) break;
            pthread_cond_wait(&p->lane_cv, &e->pipe_lk);
        }
        p->lane[got].busy = 1;
    }
    pthread_mutex_unlock(&e->pipe_lk);
    return got;
}

static void lanes_release(struct engine *e, int all, int first)
{
    pthread_mutex_lock(&e->pipe_lk);
    for (int l = 0; l < PIPE_LANES; l++) if (all || l == first) e->pipe.lane[l].busy = 0;
    pthread_cond_broadcast(&e->pipe.lane_cv);
    pthread_mutex_unlock(&e->pipe_lk);
}

int pipe_residue(struct engine *e, uint64_t *in0, uint64_t *out)
{
    int rc = lanes_acquire(e, 1);                  /* no job in flight while the buffers are read */
    if (rc < 0) return rc;
    rc = 0;
    for (int l = 0; l < PIPE_LANES && !rc; l++) {
        const struct lane *L = &e->pipe.lane[l];
        rc = count_nonzero_dev(L->d_in[0], L->d_in_cap[0], in0);
        if (!rc) rc = count_nonzero_dev(L->d_out, L->d_out_cap, out);
        for (size_t k = 0; L->h_in[0] && k < L->h_in_cap[0]; k++) *in0 += ((const uint8_t *)L->h_in[0])[k] != 0;
        for (size_t k = 0; L->h_out && k < L->h_out_cap; k++) *out += ((const uint8_t *)L->h_out)[k] != 0;
    }
    lanes_release(e, 1, 0);
    pthread_mutex_lock(&e->comb_q.lk);
    for (int c = 0; c < COMB_KINDS; c++) {
        const struct comb_kind *K = &e->comb_q.kind[c];
        for (size_t k = 0; K->h_in[0] && k < K->h_in_cap[0]; k++) *in0 += ((const uint8_t *)K->h_in[0])[k] != 0;
    }
    pthread_mutex_unlock(&e->comb_q.lk);
    return rc;
}

/* host -> HBM on the lane's stream: staged through *hbuf piece by piece, or straight from page-locked memory */
static int lane_upload(struct lane *L, void *dst, void **hbuf, size_t *hcap, const uint8_t *src, size_t bytes, int staged)
{
    int rc = 0;
    if (!bytes) return 0;
    if (!staged) { TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, L->st)); return 0; }
    if ((rc = host_grow(hbuf, hcap, bytes))) return rc;
    for (size_t off = 0; off < bytes; off += PIPE_PIECE) {
        const size_t pb = bytes - off < PIPE_PIECE ? bytes - off : PIPE_PIECE;
        par_copy((uint8_t *)*hbuf + off, src + off, pb);
        TRY(hipMemcpyAsync((uint8_t *)dst + off, (uint8_t *)*hbuf + off, pb, hipMemcpyHostToDevice, L->st));
    }
out:
    return rc;
}

/* Wait for the kernels of the chunk the lane carries, fetch its results, deliver them and zero the staging copies of
 * secrets.  The download is queued only now, when it can run at once: a copy queued behind kernels still to run sits at
 * the head of one of the DMA engines' queues, and the UPLOADS that the runtime later hands to the same engine wait
 * behind it (measured: the last two pieces of a chunk's upload ran 5 ms late, the whole call 12.4 instead of 10 ms). */
static int lane_drain(struct lane *L, int wipe, int *wipes)
{
    int rc = 0;
    TRY(hipStreamSynchronize(L->st));
    if (L->pend_bytes) {
        if (!L->pend_queued) {
            TRY(hipMemcpyAsync(L->pend_via ? L->pend_via : (void *)L->pend_dst, L->pend_dev, L->pend_bytes, hipMemcpyDeviceToHost, L->st));
            if (wipe & 2) TRY(hipMemsetAsync(L->pend_dev, 0, L->pend_bytes, L->st));    /* shared secrets leave HBM with the call */
            TRY(hipStreamSynchronize(L->st));
        }
        if (L->pend_via) par_copy(L->pend_dst, L->pend_via, L->pend_bytes);
    }
    /* wipes == NULL: the lane is about to be reused, zero it now; otherwise queue the zeroing (end of the call) */
    if ((wipe & 1) && L->used_in0) { if (wipes) pool_submit(wipes, (uint8_t *)L->h_in[0], NULL, L->used_in0, 0); else par_copy(L->h_in[0], NULL, L->used_in0); }
    if ((wipe & 2) && L->pend_bytes && L->pend_via) { if (wipes) pool_submit(wipes, (uint8_t *)L->pend_via, NULL, L->pend_bytes, 0); else par_copy(L->pend_via, NULL, L->pend_bytes); }
out:
    L->pend_bytes = 0;
    L->used_in0 = 0;
    return rc;
}

static int g_fail_next_host_call;      /* eddsa_amd_debug_fail_next_host_call: set and consumed atomically */

/* measurement aid: host-side time stamps of the last host-pointer call (eddsa_amd_debug_pipe_trace) */
#define TRACE_MAX 512
/* one trace per process; stamps of concurrent calls interleave.  Slots are reserved atomically (leaders of different
 * operations stamp at the same time) and filled with atomic stores (a call that restarts the trace may hand a slot out
 * again while its previous owner is still writing: the stamp is then a mix of two calls' - it is a diagnostic - but never a
 * data race); on / n / seen are read and written atomically; the reader holds g_table for writing, i.e. no call is in flight */
static struct { int on, n, seen; int tag[TRACE_MAX]; unsigned chunk[TRACE_MAX]; int64_t t_ns[TRACE_MAX]; } g_trace;
#define TRACE_ON() __atomic_load_n(&g_trace.on, __ATOMIC_RELAXED)
static int64_t trace_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
}
#define TRACE(tag_, k_) do { if (TRACE_ON()) { const int i_ = __atomic_fetch_add(&g_trace.n, 1, __ATOMIC_RELAXED); \
    if (i_ >= 0 && i_ < TRACE_MAX) { __atomic_store_n(&g_trace.tag[i_], (tag_), __ATOMIC_RELAXED); \
        __atomic_store_n(&g_trace.chunk[i_], (unsigned)(k_), __ATOMIC_RELAXED); __atomic_store_n(&g_trace.t_ns[i_], trace_now(), __ATOMIC_RELAXED); } } } while (0)
#define TRACE_RESTART() __atomic_store_n(&g_trace.n, 0, __ATOMIC_RELAXED)

/* on != 0: record host-side time stamps in every host-pointer call from now on; returns the number of stamps of the last
 * call and copies up to `max` of them: tag (0 call start, 1 lane drained, 2 inputs staged and queued, 3 kernels queued,
 * 4 download queued, 5 all lanes drained, 6 call end; of a combined launch also 7 leader elected, 8 callers gathered,
 * 9 requests packed, 10 results handed back), chunk index, milliseconds since the call started.  on = 2: recording stops
 * by itself after the 20th combined launch of 32 calls or more, so that a typical launch under load can be read */
#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
int eddsa_amd_debug_pipe_trace(int on, int *tags, unsigned *chunks, double *ms, int max)
{
    pthread_rwlock_wrlock(&g_table);       /* no call in flight: nobody is stamping */
    int n = __atomic_load_n(&g_trace.n, __ATOMIC_RELAXED);
    n = n > TRACE_MAX ? TRACE_MAX : n;
    n = n < max ? n : max;
    for (int i = 0; i < n; i++) {
        tags[i] = __atomic_load_n(&g_trace.tag[i], __ATOMIC_RELAXED); chunks[i] = __atomic_load_n(&g_trace.chunk[i], __ATOMIC_RELAXED);
        ms[i] = 1e-6 * (double)(__atomic_load_n(&g_trace.t_ns[i], __ATOMIC_RELAXED) - __atomic_load_n(&g_trace.t_ns[0], __ATOMIC_RELAXED));
    }
    __atomic_store_n(&g_trace.on, on, __ATOMIC_RELAXED);
    __atomic_store_n(&g_trace.seen, 0, __ATOMIC_RELAXED);
    pthread_rwlock_unlock(&g_table);
    return n;
}
#endif

/* test hook (inert unless armed, include/eddsa_amd_debug.h): the next host-pointer call fails (hipErrorUnknown) after its
 * inputs were staged and its kernels launched, so that the error path's clean-up (the staging copies of secrets are wiped
 * there too) can be exercised */
#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
int eddsa_amd_debug_fail_next_host_call(void)
{
    if (!__atomic_load_n(&g_hooks_armed, __ATOMIC_ACQUIRE)) return EDDSA_AMD_HOOKS_OFF;
    __atomic_store_n(&g_fail_next_host_call, 1, __ATOMIC_RELEASE);
    return 0;
}
#endif

/* One host-pointer job on engine e (its device is current).
 * Chunk k travels on lane k mod 3: [wait for the lane's previous chunk, fetch and deliver its results] - stage and
 * upload - kernels, all on the lane's stream; copies and kernels of different lanes overlap.  How the KERNELS of
 * consecutive chunks are ordered is the job's choice (hjob.chain; tools/pipe_sweep.py has the measurements):
 *   in chunk order (each chunk's kernels wait for the previous lane's `kdone`): x25519, sign and the other fixed-base
 *     operations, whose chunk is one long kernel - side by side three of them share the chip, finish together, and the
 *     pipeline drains and refills in bursts (x25519 84-93 M/s against 103-105 in order);
 *   side by side: verify, whose three kernels per chunk leave ramps and tails that the neighbours fill (95.7 M/s
 *     against 79-86 in order).
 * `kdone` is recorded before a verify pass waits for its exact path, so that those few latency-bound waves never hold
 * up the next chunk (each lane's stream has its own workspace). */
/* A ragged call's offset table (m + 1 entries of the caller's memory) must not decrease, and the bytes it spans must be an
 * amount a buffer can hold: everything below takes msg_off[k + 1] - msg_off[k] for a length.  Checked chunk by chunk, right
 * before the chunk's offsets are first used (the walk of chunk k runs beside the GPU work of chunk k - 1); a table that
 * fails ends the call with -hipErrorInvalidValue before a byte of that chunk is read. */
#define MSG_BYTES_MAX ((uint64_t)1 << 46)
static int offsets_ok(const uint64_t *off, size_t m)
{
    uint64_t bad = 0;
    for (size_t t = 0; t < m; t++) bad |= (uint64_t)(off[t + 1] < off[t]);
    return !bad && off[m] - off[0] <= MSG_BYTES_MAX;
}

static int pipe_run_on(struct engine *e, const struct hjob *j, size_t n)
{
    int rc = 0, wipes = 0;
    struct pipe *p = &e->pipe;
    int staged_in[PIPE_MAX_IN] = { 0 }, staged_msgs = 0, staged_out = 0;
    if (n == 0) return 0;
    const int ragged = j->has_msgs && j->msg_off != NULL;
    const int tunable = j->stats == NULL;                       /* (a combination covers a fixed number of items) */
    const size_t stage = tunable && g_pipe_stage ? g_pipe_stage : j->chunk ? j->chunk : PIPE_CHUNK;
    const size_t first = tunable && g_pipe_first ? g_pipe_first : j->first_chunk ? j->first_chunk : PIPE_FIRST_CHUNK;
    /* The offset table as a whole, before anything is touched: it must not end before it starts, nor span more than 2^46
     * bytes (include/eddsa_amd.h).  Whether it DEcreases somewhere in between is checked chunk by chunk below, as each chunk is
     * about to be staged (one pass over the table in step with the copies instead of a second one in front of them): a table
     * that is bad further on is found after earlier chunks have run, the call then returns -hipErrorInvalidValue and its
     * outputs are unspecified, as on every error. */
    if (ragged && (j->msg_off[n] < j->msg_off[0] || j->msg_off[n] - j->msg_off[0] > MSG_BYTES_MAX)) return -(int)hipErrorInvalidValue;
    const size_t msg_total = !j->has_msgs ? 0 : ragged ? (size_t)(j->msg_off[n] - j->msg_off[0]) : n * j->msg_len;
    /* one chunk, one lane (any); several chunks - and the batch verification, whose statistics live in the pipe - all of them */
    const int all = j->stats != NULL || n > first;
    const int base = lanes_acquire(e, all);
    if (base < 0) return base;
#define LANE_OF(k) (&p->lane[all ? (k) % PIPE_LANES : (unsigned)base])
    if (TRACE_ON() && !j->traced) TRACE_RESTART();
    TRACE(0, 0);
    {
        struct lane *prev = NULL;
        for (int i = 0; i < j->n_in; i++)
            staged_in[i] = !j->src_pinned && !(n * j->in_w[i] >= PIN_CHECK_MIN && is_pinned(j->in[i], n * j->in_w[i]));
        staged_msgs = !j->src_pinned && !(msg_total >= PIN_CHECK_MIN && is_pinned(ragged ? j->msgs + j->msg_off[0] : j->msgs, msg_total));
        staged_out = !j->src_pinned && !(n * j->out_w >= PIN_CHECK_MIN && is_pinned(j->out, n * j->out_w));
        if (j->stats) {
            TRY(hipMemsetAsync(p->d_stats, 0, 16, p->lane[0].st));
            TRY(hipStreamSynchronize(p->lane[0].st));           /* the other lanes' kernels add to it too */
        }
        size_t lo = 0;
        for (unsigned k = 0; lo < n; k++) {
            struct lane *L = LANE_OF(k);
            size_t m = first << (k < 8 ? k : 8);
            if (m > stage) m = stage;
            if (!all || m > n - lo || n - lo - m < m / 2) m = n - lo;   /* (a short tail travels with the last chunk) */
            if (ragged && !offsets_ok(j->msg_off + lo, m)) { rc = -(int)hipErrorInvalidValue; goto out; }   /* before anything of the chunk is staged */
            /* the lane's previous chunk (k - 3): the two chunks after it keep the GPU busy meanwhile */
            if (k >= PIPE_LANES && (rc = lane_drain(L, j->wipe, NULL))) goto out;    /* (every call leaves its lanes drained) */
            TRACE(1, k);
            for (int i = 0; i < j->n_in; i++) {
                if ((rc = dev_grow(&L->d_in[i], &L->d_in_cap[i], m * j->in_w[i], L->st))) goto out;
                if ((rc = lane_upload(L, L->d_in[i], &L->h_in[i], &L->h_in_cap[i], j->in[i] + lo * j->in_w[i], m * j->in_w[i], staged_in[i]))) goto out;
            }
            if (staged_in[0]) L->used_in0 = m * j->in_w[0];
            if (j->has_msgs) {
                /* ragged messages: the chunk's own bytes, msg_off[lo] .. msg_off[lo + m) */
                const size_t bytes = ragged ? (size_t)(j->msg_off[lo + m] - j->msg_off[lo]) : m * j->msg_len;
                const uint8_t *src = ragged ? j->msgs + j->msg_off[lo] : j->msgs + lo * j->msg_len;
                if ((rc = dev_grow(&L->d_msgs, &L->d_msgs_cap, bytes, L->st))) goto out;
                if ((rc = lane_upload(L, L->d_msgs, &L->h_msgs, &L->h_msgs_cap, src, bytes, staged_msgs))) goto out;
            }
            if (ragged) {
                /* ... and its own offset table, rebased to the chunk's first byte (the kernels index it by the item's
                 * number inside the chunk).  A table that starts at 0 in page-locked memory - the combiner's - goes as it is. */
                const size_t ob = (m + 1) * sizeof(uint64_t);
                if ((rc = dev_grow(&L->d_off, &L->d_off_cap, ob, L->st))) goto out;
                if (j->src_pinned && j->msg_off[lo] == 0) {
                    TRY(hipMemcpyAsync(L->d_off, j->msg_off + lo, ob, hipMemcpyHostToDevice, L->st));
                } else {
                    if ((rc = host_grow(&L->h_off, &L->h_off_cap, ob))) goto out;
                    uint64_t *ho = (uint64_t *)L->h_off;
                    const uint64_t base0 = j->msg_off[lo];
                    for (size_t t = 0; t <= m; t++) ho[t] = j->msg_off[lo + t] - base0;
                    TRY(hipMemcpyAsync(L->d_off, ho, ob, hipMemcpyHostToDevice, L->st));
                }
            }
            if ((rc = dev_grow(&L->d_out, &L->d_out_cap, m * j->out_w, L->st))) goto out;
            TRACE(2, k);
            if (prev && (g_pipe_chain < 0 ? j->chain : g_pipe_chain)) TRY(hipStreamWaitEvent(L->st, prev->kdone, 0));      /* kernels in chunk order */
            {
                struct hjob jj = *j;
                jj.stats = j->stats ? p->d_stats : NULL;
                rc = j->run(e, &jj, (uint8_t *)L->d_out, (uint8_t *const *)L->d_in, (const uint8_t *)L->d_msgs,
                            ragged ? (const uint64_t *)L->d_off : NULL, j->msg_len, m, L->st, L->kdone);
            }
            prev = L;
            TRACE(3, k);
            if (!rc && __atomic_exchange_n(&g_fail_next_host_call, 0, __ATOMIC_ACQ_REL)) rc = -(int)hipErrorUnknown;
            if (rc) goto out;
            /* the download: queued by lane_drain, once the kernels are done */
            L->pend_via = NULL;
            if (staged_out) {
                if ((rc = host_grow(&L->h_out, &L->h_out_cap, m * j->out_w))) goto out;
                L->pend_via = (uint8_t *)L->h_out;
            }
            L->pend_dst = j->out + lo * j->out_w; L->pend_dev = (uint8_t *)L->d_out; L->pend_bytes = m * j->out_w;
            L->pend_queued = 0;
            if (k == 0 && m == n) {            /* a call of one chunk has no uploads to hold up: everything in order, one wait */
                TRY(hipMemcpyAsync(L->pend_via ? L->pend_via : (void *)L->pend_dst, L->pend_dev, L->pend_bytes, hipMemcpyDeviceToHost, L->st));
                if (j->wipe & WIPE_OUT) TRY(hipMemsetAsync(L->pend_dev, 0, L->pend_bytes, L->st));
                L->pend_queued = 1;
            }
            /* secrets do not outlive the call in HBM */
            if (j->wipe & WIPE_IN0) TRY(hipMemsetAsync(L->d_in[0], 0, m * j->in_w[0], L->st));
            lo += m;
            TRACE(4, k);
            if (lo >= n) {                     /* the chunks still in flight, oldest first */
                for (unsigned t = k + 1 < PIPE_LANES ? PIPE_LANES - k : 1; t <= PIPE_LANES; t++)
                    if ((rc = lane_drain(LANE_OF(k + t), j->wipe, &wipes))) goto out;
                TRACE(5, k);
            }
        }
        if (j->stats) {
            TRY(hipMemcpyAsync(p->h_stats, p->d_stats, 16, hipMemcpyDeviceToHost, p->lane[0].st));
            TRY(hipStreamSynchronize(p->lane[0].st));
            memcpy(j->stats, p->h_stats, 16);
        }
        /* every lane of the call has been waited for: what a verify pass's kernels reported (a hand-off given up) is this
         * call's error (the other operations' kernels have nothing to report: a sign call does not take a verify call's word) */
        if (j->out_w == 1 && (rc = take_async_error(e))) goto out;
    }
out:
    if (rc) {
        /* a failed call must not leave its secrets behind either (best effort: whole buffers, HBM and pinned) */
        for (int l = 0; l < PIPE_LANES; l++) {
            struct lane *L = &p->lane[l];
            if (!all && l != base) continue;
            HIP_NOTE(hipStreamSynchronize(L->st));
            L->pend_bytes = 0; L->used_in0 = 0;
            if ((j->wipe & WIPE_IN0) && L->d_in[0]) HIP_NOTE(hipMemsetAsync(L->d_in[0], 0, L->d_in_cap[0], L->st));
            if ((j->wipe & WIPE_OUT) && L->d_out) HIP_NOTE(hipMemsetAsync(L->d_out, 0, L->d_out_cap, L->st));
            if (j->wipe) HIP_NOTE(hipStreamSynchronize(L->st));
            if ((j->wipe & WIPE_IN0) && L->h_in[0]) pool_submit(&wipes, (uint8_t *)L->h_in[0], NULL, L->h_in_cap[0], 0);
            if ((j->wipe & WIPE_OUT) && L->h_out) pool_submit(&wipes, (uint8_t *)L->h_out, NULL, L->h_out_cap, 0);
        }
    }
    if (j->wipe) pool_wait(&wipes);            /* nothing secret outlives the call in the pinned staging buffers */
    TRACE(6, 0);
#undef LANE_OF
    lanes_release(e, all, base);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * the combiner: concurrent small calls of one operation become one launch
 * ---------------------------------------------------------------------------------------- */

#define COMBINE_MAX_N 64                     /* items of a call that may be merged */
#define COMBINE_MAX_BYTES ((size_t)1 << 20)  /* ... and its message bytes */
#define COMBINE_MAX_BATCH 16384              /* items per combined launch */
#define COMBINE_MAX_BATCH_BYTES ((size_t)64 << 20)   /* ... and its message bytes (pinned staging of the packed batch) */
#define COMBINE_RETRY_GIVE_UP 4               /* request-by-request retries of a failed launch that may fail alike in a row before the rest are given the launch's error */
#define COMBINE_GATHER_NS 80000              /* how long a leader elected under contention waits for the callers of the previous batch */

struct creq { const struct hjob *j; size_t n; int rc, done; struct creq *next; };

void combiner_init(struct combiner *q)
{
    memset(q, 0, sizeof(*q));
    pthread_mutex_init(&q->lk, NULL);
}

void combiner_release(struct combiner *q)
{
    for (int c = 0; c < COMB_KINDS; c++) {
        struct comb_kind *K = &q->kind[c];
        for (int i = 0; i < PIPE_MAX_IN; i++) host_free_wiped(&K->h_in[i], &K->h_in_cap[i]);
        host_free_wiped(&K->h_msgs, &K->h_msgs_cap);
        host_free_wiped(&K->h_out, &K->h_out_cap);
        { void *p = K->h_off; size_t cap = K->h_off_cap; host_free_wiped(&p, &cap); K->h_off = NULL; K->h_off_cap = 0; }
    }
    pthread_mutex_destroy(&q->lk);
}

/* diagnostic: combined launches and the items they carried on the default device since its engine was built */
#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
int eddsa_amd_combiner_stats(uint64_t out[2])
{
    struct call c;
    int rc = enter(&c, -1);
    if (rc) return rc;
    pthread_mutex_lock(&c.e->comb_q.lk);
    out[0] = c.e->comb_q.batches; out[1] = c.e->comb_q.items;
    pthread_mutex_unlock(&c.e->comb_q.lk);
    leave(&c);
    return 0;
}
#endif

static int64_t now_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
}

/* the leader's work: pack the requests of `batch` (a list through ->next, all of one operation), run them as one job
 * from the operation's pinned buffers, scatter the results.  Returns the job's status. */
static int combiner_run(struct engine *e, struct comb_kind *K, struct creq *batch, size_t total)
{
    const struct hjob *j0 = batch->j;
    int rc = 0, same_len = 1;
    size_t msg_bytes = 0;
    for (struct creq *r = batch; r; r = r->next) {
        if (r->j->msg_len != j0->msg_len) same_len = 0;
        msg_bytes += r->n * r->j->msg_len;
    }
    for (int i = 0; i < j0->n_in; i++) if ((rc = host_grow(&K->h_in[i], &K->h_in_cap[i], total * j0->in_w[i]))) return rc;
    if (j0->has_msgs && (rc = host_grow(&K->h_msgs, &K->h_msgs_cap, msg_bytes))) return rc;
    if ((rc = host_grow(&K->h_out, &K->h_out_cap, total * j0->out_w))) return rc;
    if (j0->has_msgs && !same_len) {
        void *p = K->h_off;
        rc = host_grow(&p, &K->h_off_cap, (total + 1) * sizeof(uint64_t));
        K->h_off = (uint64_t *)p;
        if (rc) return rc;
    }
    size_t at = 0, mat = 0;
    for (struct creq *r = batch; r; r = r->next) {
        for (int i = 0; i < j0->n_in; i++) memcpy((uint8_t *)K->h_in[i] + at * j0->in_w[i], r->j->in[i], r->n * j0->in_w[i]);
        if (j0->has_msgs) {
            if (r->n * r->j->msg_len != 0) memcpy((uint8_t *)K->h_msgs + mat, r->j->msgs, r->n * r->j->msg_len);
            if (!same_len) for (size_t k = 0; k < r->n; k++) K->h_off[at + k] = mat + k * r->j->msg_len;
            mat += r->n * r->j->msg_len;
        }
        at += r->n;
    }
    if (j0->has_msgs && !same_len) K->h_off[total] = mat;
    struct hjob big = *j0;
    for (int i = 0; i < j0->n_in; i++) big.in[i] = (const uint8_t *)K->h_in[i];
    big.msgs = (const uint8_t *)K->h_msgs;
    big.msg_off = j0->has_msgs && !same_len ? K->h_off : NULL;
    big.out = (uint8_t *)K->h_out;
    big.kind = 0;
    big.src_pinned = 1;
    big.traced = 1;
    TRACE(9, (unsigned)total);
    rc = pipe_run_on(e, &big, total);
    at = 0;
    for (struct creq *r = batch; r && !rc; r = r->next) {
        memcpy(r->j->out, (uint8_t *)K->h_out + at * j0->out_w, r->n * j0->out_w);
        at += r->n;
    }
    /* the packed copies of secrets go as well */
    if (j0->wipe & WIPE_IN0) memset(K->h_in[0], 0, total * j0->in_w[0]);
    if (j0->wipe & WIPE_OUT) memset(K->h_out, 0, total * j0->out_w);
    TRACE(10, (unsigned)total);
    if (TRACE_ON() == 2 && total >= 32 && __atomic_add_fetch(&g_trace.seen, 1, __ATOMIC_RELAXED) == 20)
        __atomic_store_n(&g_trace.on, 0, __ATOMIC_RELAXED);   /* on = 2: keep the 20th launch that carried 32 calls or more */
    return rc;
}

/* Waiters sleep on their operation's generation counter (one per operation: with one for all, every completion woke
 * the callers of the other operations too - 256 threads split over four operations spent their time being woken for
 * nothing, 38-80 k calls/s), not on a condition variable: when a launch completes, every
 * caller it carried is woken at once and leaves on its own `done` flag without touching the queue's mutex (with a
 * condition variable the 64 callers of a launch re-acquired the mutex one after the other, a context switch each,
 * which cost more than the GPU pass). */
static void gen_wait(uint32_t *gen, uint32_t seen)
{
    (void)syscall(SYS_futex, gen, FUTEX_WAIT_PRIVATE, seen, NULL, NULL, 0);
}
static void gen_wake_all(uint32_t *gen)
{
    (void)syscall(SYS_futex, gen, FUTEX_WAKE_PRIVATE, INT_MAX, NULL, NULL, 0);
}

/* One leader per OPERATION at a time: the calls queued for verify travel in one launch while, side by side on another
 * lane of the pipeline, the calls queued for sign travel in theirs (threads that issue different operations would
 * otherwise take turns: 64 threads split over four operations got 28 k calls/s that way). */
static int combiner_submit(struct engine *e, const struct hjob *j, size_t n)
{
    struct combiner *q = &e->comb_q;
    struct comb_kind *K = &q->kind[j->kind];
    struct creq me = { j, n, 0, 0, NULL };
    pthread_mutex_lock(&q->lk);
    if (q->tail) q->tail->next = &me; else q->head = &me;
    q->tail = &me;
    K->queued++;
    for (;;) {
        if (__atomic_load_n(&me.done, __ATOMIC_ACQUIRE)) break;
        if (K->active) {
            const uint32_t seen = __atomic_load_n(&K->gen, __ATOMIC_RELAXED);
            pthread_mutex_unlock(&q->lk);
            gen_wait(&K->gen, seen);                   /* returns at once if a launch of this operation completed in between */
            if (__atomic_load_n(&me.done, __ATOMIC_ACQUIRE)) return me.rc;
            pthread_mutex_lock(&q->lk);
            continue;
        }
        K->active = 1;
        if (TRACE_ON()) TRACE_RESTART();
        TRACE(7, 0);
        /* Under contention the callers of the launch that has just finished are about to queue again (they do within
         * microseconds of being woken): give them a moment, or the callers split into two camps that take turns and
         * every launch carries half of them.  Expected: whoever was already waiting when that launch ended, plus the
         * calls it carried.  Never long; a lone caller (the previous launch carried one call) never waits. */
        {
            const unsigned want = K->waiting_at_end + K->last_reqs;
            if (K->last_reqs > 1 && K->queued < want) {
                const int64_t until = now_ns() + COMBINE_GATHER_NS;
                pthread_mutex_unlock(&q->lk);
                for (;;) {
                    sched_yield();
                    pthread_mutex_lock(&q->lk);
                    if (K->queued >= want || now_ns() >= until) break;
                    pthread_mutex_unlock(&q->lk);
                }
            }
        }
        /* everything queued for this operation (this thread's own request is among it unless thousands are ahead of it:
         * then it leads again) */
        TRACE(8, K->queued);
        struct creq *batch = NULL, *btail = NULL, **pp = &q->head, *last = NULL;
        size_t total = 0, reqs = 0, bytes = 0;
        while (*pp) {
            struct creq *r = *pp;
            const size_t rb = r->j->has_msgs ? r->n * r->j->msg_len : 0;
            if (r->j->kind == j->kind && total + r->n <= COMBINE_MAX_BATCH && (reqs == 0 || bytes + rb <= COMBINE_MAX_BATCH_BYTES)) {
                *pp = r->next;
                r->next = NULL;
                if (btail) btail->next = r; else batch = r;
                btail = r;
                total += r->n; reqs++; bytes += rb;
                K->queued--;
            } else {
                last = r;
                pp = &r->next;
            }
        }
        q->tail = last;
        pthread_mutex_unlock(&q->lk);
        struct hjob own = *j;
        own.traced = 1;
        const int rc = reqs == 1 && batch == &me ? pipe_run_on(e, &own, n) : combiner_run(e, K, batch, total);
        for (struct creq *r = batch; r; r = r->next) r->rc = rc;
        if (rc && reqs > 1) {
            /* A merged launch shares one status, and the eddsa.h callers abort() on failure: before anybody is told, every
             * request gets a run of its own (a transient error, or the packed batch's staging allocation, need not concern
             * the others; a request that is itself the cause fails again, alone) */
            /* ... but not without end: a batch holds up to 16 384 requests, and when the cause is the device (lost, out of
             * memory) every retry stages secrets, fails, waits and wipes while all callers stay blocked.  After
             * COMBINE_RETRY_GIVE_UP retries in a row that end with the batch's own error the rest are told that error. */
            /* Only for errors of the DEVICE: an error a caller can bring about by itself (an invalid argument, a bad offset
             * table) says nothing about the next request - four faulty callers in a row must not fail everybody behind
             * them - so such a batch is retried request by request to the end. */
            const int device_fault = rc != -(int)hipErrorInvalidValue && rc != -(int)hipErrorInvalidDevicePointer;
            int same = 0;
            for (struct creq *r = batch; r; r = r->next) {
                if (device_fault && same >= COMBINE_RETRY_GIVE_UP) { r->rc = rc; continue; }
                struct hjob alone = *r->j;
                alone.traced = 1;
                r->rc = pipe_run_on(e, &alone, r->n);
                same = r->rc == rc ? same + 1 : 0;
            }
        }
        pthread_mutex_lock(&q->lk);
        q->batches++; q->items += reqs;
        K->last_reqs = (unsigned)reqs;
        K->waiting_at_end = K->queued;
        for (struct creq *r = batch; r;) {             /* a caller may return (and its request vanish) the moment `done` is set */
            struct creq *nx = r->next;
            __atomic_store_n(&r->done, 1, __ATOMIC_RELEASE);
            r = nx;
        }
        K->active = 0;
        __atomic_add_fetch(&K->gen, 1, __ATOMIC_RELEASE);
        pthread_mutex_unlock(&q->lk);
        gen_wake_all(&K->gen);
        pthread_mutex_lock(&q->lk);
    }
    pthread_mutex_unlock(&q->lk);
    return me.rc;
}

/* on the default device */
static int pipe_run(const struct hjob *j, size_t n)
{
    struct call c;
    int rc;
    if (n == 0) return 0;
    rc = enter(&c, -1);
    if (rc) return rc;
    if (j->kind && n <= COMBINE_MAX_N && !(j->has_msgs && (j->msg_off || n * j->msg_len > COMBINE_MAX_BYTES)))
        rc = combiner_submit(c.e, j, n);
    else
        rc = pipe_run_on(c.e, j, n);
    leave(&c);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * jobs
 * ---------------------------------------------------------------------------------------- */

/* verify: chunks of 2^16, 2^17, 2^18 items, then the rest in one (measured, tools/pipe_verify_sweep.py, 2^20 items from
 * malloc memory, round 4: stage 2^20 98.8 M/s on the config-2 mix and 102.2 on valid signatures, stage 2^19 98.2 / 101.4):
 * the large last chunk hides its exact chain behind a main kernel of several rounds, as one big pass does */
#define PIPE_CHUNK_VERIFY CHUNK_MAX
#define PIPE_FIRST_CHUNK_VERIFY ((size_t)1 << 16)   /* the first chunk of a verify call (the other operations: PIPE_FIRST_CHUNK) */

#define RUN_ARGS struct engine *e, const struct hjob *j, uint8_t *d_out, uint8_t *const d_in[PIPE_MAX_IN], \
                 const uint8_t *d_msgs, const uint64_t *d_off, size_t msg_len, size_t m, hipStream_t st, hipEvent_t kdone
/* record the lane's "kernels queued" event (verify does it itself, before its stream waits for the exact path) */
#define RUN_DONE(rc_) do { int r_ = (rc_); if (!r_ && hipEventRecord(kdone, st) != hipSuccess) r_ = -(int)hipErrorUnknown; return r_; } while (0)
static int run_verify(RUN_ARGS)
{
    (void)j;
    const edk_verify_src src = { d_in[0], d_in[1], d_msgs, d_off, msg_len, 64, 32, msg_len, NULL };
    return verify_on(e, d_out, &src, m, st, kdone, g_pipe_chain == 2);
}
static int run_verify_rlc(RUN_ARGS)
{
    const edk_verify_src src = { d_in[0], d_in[1], d_msgs, d_off, msg_len, 64, 32, msg_len, NULL };
    RUN_DONE(rlc_on(e, d_out, j->stats, &src, m, st));
}
static int run_verify_records(RUN_ARGS)
{
    (void)d_msgs; (void)d_off;
    const edk_verify_src src = { d_in[0] + j->rec_sig, d_in[0] + j->rec_pub, d_in[0] + j->rec_msg, NULL, msg_len,
                                 j->in_w[0], j->in_w[0], j->in_w[0], NULL };
    return verify_on(e, d_out, &src, m, st, kdone, g_pipe_chain == 2);
}
static int run_sign(RUN_ARGS)
{
    (void)j;
    RUN_DONE(sign_on(e, d_out, d_in[0], d_in[1], d_msgs, d_off, msg_len, m, st));
}
static int run_x25519(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    RUN_DONE(x25519_on(e, d_out, d_in[0], d_in[1], m, st));
}
static int run_genpub(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    RUN_DONE(genpub_on(e, d_out, d_in[0], m, st));
}
static int run_xbase(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    RUN_DONE(xbase_on(e, d_out, d_in[0], m, st));
}
static int run_pk_to_x(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    RUN_DONE(pk_to_x_on(e, d_out, d_in[0], m, st));
}
static int run_sk_to_x(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    RUN_DONE(sk_to_x_on(e, d_out, d_in[0], m, st));
}

static struct hjob job_verify(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs, const uint8_t *msgs,
                              const uint64_t *msg_off, size_t msg_len)
{
    /* Measured (tools/pipe_sweep.py, 2^20 items from malloc memory): the three kernels of a verify chunk leave ramps and
     * tails that the neighbouring chunks' kernels fill when the lanes run side by side (95.7 M/s; in chunk order 79-86),
     * and the small first chunk gets the chip working 0.2 ms after the call. */
    const struct hjob j = { .n_in = 2, .in = { sigs, pubs, NULL }, .in_w = { 64, 32, 0 }, .has_msgs = 1, .msgs = msgs,
                            .msg_off = msg_off, .msg_len = msg_len, .out = ok, .out_w = 1, .run = run_verify,
                            .chunk = PIPE_CHUNK_VERIFY, .wipe = WIPE_NONE, .first_chunk = PIPE_FIRST_CHUNK_VERIFY,
                            .chain = 0, .kind = 1 };
    return j;
}
static struct hjob job_sign(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                            const uint64_t *msg_off, size_t msg_len)
{
    const struct hjob j = { .n_in = 2, .in = { secs, pubs, NULL }, .in_w = { 32, 32, 0 }, .has_msgs = 1, .msgs = msgs,
                            .msg_off = msg_off, .msg_len = msg_len, .out = sigs, .out_w = 64, .run = run_sign,
                            .wipe = WIPE_IN0, .chain = 1, .kind = 2 };
    return j;
}
static struct hjob job_x25519(uint8_t *out, const uint8_t *scalars, const uint8_t *points)
{
    const struct hjob j = { .n_in = 2, .in = { scalars, points, NULL }, .in_w = { 32, 32, 0 }, .out = out, .out_w = 32,
                            .run = run_x25519, .wipe = WIPE_IN0 | WIPE_OUT, .chain = 1, .kind = 3 };
    return j;
}
static struct hjob job_1in(int (*run)(RUN_ARGS), uint8_t *out, const uint8_t *in, int wipe, int kind)
{
    const struct hjob j = { .n_in = 1, .in = { in, NULL, NULL }, .in_w = { 32, 0, 0 }, .out = out, .out_w = 32, .run = run,
                            .wipe = wipe, .chain = 1, .kind = kind };
    return j;
}

/* ------------------------------------------------------------------------------------------
 * host-pointer entry points (include/eddsa_amd.h)
 * ---------------------------------------------------------------------------------------- */

int ed25519_verify_batch(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs, const uint8_t *msgs,
                         const uint64_t *msg_off, size_t msg_len, size_t n)
{
    struct hjob j = job_verify(ok, sigs, pubs, msgs, msg_off, msg_len);
    return pipe_run(&j, n);
}

int ed25519_verify_batch_rlc(uint8_t *ok, uint32_t stats[4], const uint8_t *sigs, const uint8_t *pubs,
                             const uint8_t *msgs, const uint64_t *msg_off, size_t msg_len, size_t n)
{
    uint32_t local[4] = { 0, 0, 0, 0 };
    struct hjob j = job_verify(ok, sigs, pubs, msgs, msg_off, msg_len);
    j.run = run_verify_rlc;
    j.chunk = j.first_chunk = CHUNK_MAX;   /* one combination per 2^20 items */
    j.stats = local;
    j.kind = 0;
    int rc = pipe_run(&j, n);
    if (stats) memcpy(stats, local, sizeof(local));
    return rc;
}

int ed25519_verify_records(uint8_t *ok, const uint8_t *records, size_t stride, size_t sig_off, size_t pub_off,
                           size_t msg_off, size_t msg_len, size_t n)
{
    if (!records_ok(stride, sig_off, pub_off, msg_off, msg_len)) return -(int)hipErrorInvalidValue;
    const struct hjob j = { .n_in = 1, .in = { records, NULL, NULL }, .in_w = { stride, 0, 0 }, .msg_len = msg_len, .out = ok,
                            .out_w = 1, .run = run_verify_records, .rec_sig = sig_off, .rec_pub = pub_off, .rec_msg = msg_off,
                            .chunk = PIPE_CHUNK_VERIFY, .wipe = WIPE_NONE, .first_chunk = PIPE_FIRST_CHUNK_VERIFY };
    return pipe_run(&j, n);
}

int ed25519_sign_batch(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                       const uint64_t *msg_off, size_t msg_len, size_t n)
{
    struct hjob j = job_sign(sigs, secs, pubs, msgs, msg_off, msg_len);
    return pipe_run(&j, n);
}

int x25519_batch(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n)
{
    struct hjob j = job_x25519(out, scalars, points);
    return pipe_run(&j, n);
}

int ed25519_genpub_batch(uint8_t *pubs, const uint8_t *secs, size_t n)
{
    struct hjob j = job_1in(run_genpub, pubs, secs, WIPE_IN0, 4);
    return pipe_run(&j, n);
}
int x25519_base_batch(uint8_t *out, const uint8_t *scalars, size_t n)
{
    struct hjob j = job_1in(run_xbase, out, scalars, WIPE_IN0, 5);
    return pipe_run(&j, n);
}
int pk_ed25519_to_x25519_batch(uint8_t *out, const uint8_t *in, size_t n)
{
    struct hjob j = job_1in(run_pk_to_x, out, in, WIPE_NONE, 6);
    return pipe_run(&j, n);
}
int sk_ed25519_to_x25519_batch(uint8_t *out, const uint8_t *in, size_t n)
{
    struct hjob j = job_1in(run_sk_to_x, out, in, WIPE_IN0 | WIPE_OUT, 7);
    return pipe_run(&j, n);
}

/* ------------------------------------------------------------------------------------------
 * several devices in one process, host-pointer forms: thread d runs the ordinary pipeline of device d
 * on shard d; results are copied device -> caller's buffer slice directly, so there is nothing to gather
 * ---------------------------------------------------------------------------------------- */

struct shard_job { struct hjob j; size_t n; int device; int rc; };

static void *shard_thread(void *arg)
{
    struct shard_job *s = (struct shard_job *)arg;
    struct call c;
    s->rc = enter(&c, s->device);
    if (s->rc) return NULL;
    s->rc = pipe_run_on(c.e, &s->j, s->n);
    leave(&c);
    return NULL;
}

/* split job j over the device set: item ranges for the fixed-width arrays, message bytes for ragged ones */
static int multi_run(const struct hjob *j, size_t n)
{
    struct shard_job jobs[MAX_DEVICES];
    pthread_t th[MAX_DEVICES];
    uint64_t *offs[MAX_DEVICES];
    int started[MAX_DEVICES];
    int rc = 0, g;
    pthread_rwlock_rdlock(&g_table);
    g = g_multi.n;
    for (int d = 0; d < g; d++) jobs[d].device = g_multi.dev[d];
    pthread_rwlock_unlock(&g_table);
    if (g == 0) return -(int)hipErrorNotInitialized;
    if (n == 0) return 0;
    memset(offs, 0, sizeof(offs));
    memset(started, 0, sizeof(started));
    for (int d = 0; d < g; d++) {
        size_t lo, hi;
        eddsa_amd_shard_bounds(n, d, g, &lo, &hi);
        jobs[d].j = *j;
        jobs[d].n = hi - lo;
        jobs[d].rc = 0;
        for (int i = 0; i < j->n_in; i++) jobs[d].j.in[i] = j->in[i] + lo * j->in_w[i];
        jobs[d].j.out = j->out + lo * j->out_w;
        if (j->has_msgs && j->msg_off) {            /* ragged: the shard's own offset table, rebased to 0 */
            offs[d] = (uint64_t *)malloc((hi - lo + 1) * sizeof(uint64_t));
            if (!offs[d]) { rc = -(int)hipErrorOutOfMemory; break; }
            for (size_t k = 0; k <= hi - lo; k++) offs[d][k] = j->msg_off[lo + k] - j->msg_off[lo];
            jobs[d].j.msg_off = offs[d];
            jobs[d].j.msgs = j->msgs + j->msg_off[lo];
        } else if (j->has_msgs) {
            jobs[d].j.msgs = j->msgs + lo * j->msg_len;
        }
    }
    for (int d = 0; d < g && !rc; d++) {
        if (jobs[d].n == 0) continue;
        if (pthread_create(&th[d], NULL, shard_thread, &jobs[d]) != 0) { rc = -(int)hipErrorOutOfMemory; break; }
        started[d] = 1;
    }
    for (int d = 0; d < g; d++) {
        if (started[d]) { pthread_join(th[d], NULL); if (!rc) rc = jobs[d].rc; }
        free(offs[d]);
    }
    return rc;
}

int ed25519_verify_batch_multi(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs, const uint8_t *msgs,
                               const uint64_t *msg_off, size_t msg_len, size_t n)
{
    struct hjob j = job_verify(ok, sigs, pubs, msgs, msg_off, msg_len);
    return multi_run(&j, n);
}

int ed25519_sign_batch_multi(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                             const uint64_t *msg_off, size_t msg_len, size_t n)
{
    struct hjob j = job_sign(sigs, secs, pubs, msgs, msg_off, msg_len);
    return multi_run(&j, n);
}

int x25519_batch_multi(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n)
{
    struct hjob j = job_x25519(out, scalars, points);
    return multi_run(&j, n);
}

/* ------------------------------------------------------------------------------------------
 * the eddsa.h surface: batches of one.  No error channel in these signatures, so fail loudly.
 * ---------------------------------------------------------------------------------------- */

static void must(int rc, const char *what)
{
    if (rc == 0) return;
    fprintf(stderr, "libeddsa_amd: %s failed on the GPU path: %s (no CPU fallback exists)\n", what,
            eddsa_amd_strerror(rc));
    abort();
}

void ed25519_genpub(uint8_t pub[32], const uint8_t sec[32])
{
    must(ed25519_genpub_batch(pub, sec, 1), "ed25519_genpub");
}

void ed25519_sign(uint8_t sig[64], const uint8_t sec[32], const uint8_t pub[32], const uint8_t *data, size_t len)
{
    must(ed25519_sign_batch(sig, sec, pub, data, NULL, len, 1), "ed25519_sign");
}

bool ed25519_verify(const uint8_t sig[64], const uint8_t pub[32], const uint8_t *data, size_t len)
{
    uint8_t ok = 0;
    must(ed25519_verify_batch(&ok, sig, pub, data, NULL, len, 1), "ed25519_verify");
    return ok != 0;
}

void x25519_base(uint8_t out[32], const uint8_t scalar[32])
{
    must(x25519_base_batch(out, scalar, 1), "x25519_base");
}

void x25519(uint8_t out[32], const uint8_t scalar[32], const uint8_t point[32])
{
    must(x25519_batch(out, scalar, point, 1), "x25519");
}

void pk_ed25519_to_x25519(uint8_t out[32], const uint8_t in[32])
{
    must(pk_ed25519_to_x25519_batch(out, in, 1), "pk_ed25519_to_x25519");
}

void sk_ed25519_to_x25519(uint8_t out[32], const uint8_t in[32])
{
    must(sk_ed25519_to_x25519_batch(out, in, 1), "sk_ed25519_to_x25519");
}

/* reference lib/ed25519-sha512.c:270-324 and lib/x25519.c:232-243: the obsolete names */
void eddsa_genpub(uint8_t pub[32], const uint8_t sec[32]) { ed25519_genpub(pub, sec); }
void eddsa_sign(uint8_t sig[64], const uint8_t sec[32], const uint8_t pub[32], const uint8_t *data, size_t len)
{
    ed25519_sign(sig, sec, pub, data, len);
}
bool eddsa_verify(const uint8_t sig[64], const uint8_t pub[32], const uint8_t *data, size_t len)
{
    return ed25519_verify(sig, pub, data, len);
}
void DH(uint8_t out[32], const uint8_t sec[32], const uint8_t point[32]) { x25519(out, sec, point); }
void eddsa_pk_eddsa_to_dh(uint8_t out[32], const uint8_t in[32]) { pk_ed25519_to_x25519(out, in); }
void eddsa_sk_eddsa_to_dh(uint8_t out[32], const uint8_t in[32]) { sk_ed25519_to_x25519(out, in); }
