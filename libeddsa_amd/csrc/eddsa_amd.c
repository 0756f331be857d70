/*
 * eddsa_amd.c - host side of libeddsa_amd.so, plain C.
 *
 * Exports the thirteen symbols of the reference's public header (include/eddsa.h, reference
 * lib/eddsa.h:44-113) and the batched entry points of include/eddsa_amd.h.  Everything is
 * computed by the HIP kernels in kernels.hip; there is no CPU arithmetic in this file and no
 * fallback: without a usable gfx950 device every entry point fails (batch API: negative return;
 * eddsa.h API, which has no error channel: message on stderr + abort()).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "eddsa.h"
#include "eddsa_amd.h"
#include "eddsa_kernels.h"

#define CHUNK_MAX ((size_t)1 << 20)   /* verify items per workspace pass: 1.6 GB of HBM workspace */
#define MARK_SLOTS 256                /* profiled verify passes kept for eddsa_amd_verify_phase_ms */

/* Workspaces: a small pool, so that passes issued on DIFFERENT streams (host threads that
 * each own a stream) overlap on the GPU instead of queueing behind one workspace.  A stream keeps the
 * slot it used last (passes on one stream are ordered anyway, and the slot has the right size);
 * another stream takes an idle slot, or the least recently used one. */
#define VERIFY_SLOTS 4
struct vslot {
    edk_verify_ws ws;                 /* grown on demand up to CHUNK_MAX items; owns a side stream and two events */
    edk_fixed_ws fws;                 /* sign / genpub / x25519_base / x25519 workspace, grown on demand */
    hipEvent_t free;                  /* recorded after the last kernel that touches ws or fws */
    hipStream_t last_stream;
    unsigned long stamp;              /* for least-recently-used */
};

struct engine {
    int ready;
    int device;
    uint32_t *base16, *comb;           /* generated base-point tables (HBM) */
    uint32_t *comb_img;                /* the comb as the point kernels stage it in LDS (lanes.h: comb_select) */
    struct vslot vs[VERIFY_SLOTS];
    unsigned long clock;
    int offcurve_mode;                /* eddsa_amd_set_offcurve_mode: 0 reject, 1 exact (default), 2 all exact */
    int profiling;                    /* record marks around the three verify kernels */
    int marks_used;                   /* passes recorded since profiling was switched on */
    hipEvent_t marks[MARK_SLOTS][4];
};

static struct engine g_eng;
static void pipe_release(void);
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = -(int)e_; goto out; } } while (0)

const char *eddsa_amd_strerror(int err)
{
    if (err == 0) return "success";
    if (err == -100000) return "eddsa_amd: device is not gfx950 (MI355X); no code object for it";
    return hipGetErrorString((hipError_t)(-err));
}

static void ws_release(struct vslot *v)
{
    if (v->ws.digits) (void)hipFree(v->ws.digits);
    if (v->ws.table) (void)hipFree(v->ws.table);
    if (v->ws.acc) (void)hipFree(v->ws.acc);
    if (v->ws.flags) (void)hipFree(v->ws.flags);
    if (v->ws.offlist) (void)hipFree(v->ws.offlist);
    if (v->ws.offcount) (void)hipFree(v->ws.offcount);
    if (v->ws.exact_pad) (void)hipFree(v->ws.exact_pad);
    v->ws.capacity = 0;
    v->ws.digits = v->ws.table = v->ws.acc = v->ws.offlist = v->ws.offcount = v->ws.exact_pad = NULL;
    v->ws.flags = NULL;
}

static void fws_release(struct vslot *v)
{
    if (v->fws.acc) (void)hipFree(v->fws.acc);
    if (v->fws.aux) (void)hipFree(v->fws.aux);
    memset(&v->fws, 0, sizeof(v->fws));
}

/* caller holds g_lock */
static int fws_reserve(struct vslot *v, size_t items)
{
    int rc = 0;
    size_t cap = (items + VERIFY_TILE - 1) / VERIFY_TILE * VERIFY_TILE;
    cap = (cap + 8 * VERIFY_TILE - 1) / (8 * VERIFY_TILE) * (8 * VERIFY_TILE);   /* whole finish blocks */
    if (cap <= v->fws.capacity) return 0;
    TRY(hipEventSynchronize(v->free));
    fws_release(v);
    TRY(hipMalloc((void **)&v->fws.acc, cap * 30 * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->fws.aux, cap * 16 * sizeof(uint32_t)));
    v->fws.capacity = cap;
out:
    if (rc) fws_release(v);
    return rc;
}

/* caller holds g_lock */
static int ws_reserve(struct vslot *v, size_t items)
{
    int rc = 0;
    size_t cap = (items + VERIFY_TILE - 1) / VERIFY_TILE * VERIFY_TILE;
    cap = (cap + 8 * VERIFY_TILE - 1) / (8 * VERIFY_TILE) * (8 * VERIFY_TILE);   /* whole finish blocks */
    if (cap <= v->ws.capacity) return 0;
    /* the old buffers may still be in use by enqueued kernels */
    TRY(hipEventSynchronize(v->free));
    ws_release(v);
    TRY(hipMalloc((void **)&v->ws.digits, cap * 16 * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.table, cap / VERIFY_TILE * (size_t)VERIFY_TABLE_WORDS_PER_TILE * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.acc, cap * 30 * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.flags, cap));
    TRY(hipMalloc((void **)&v->ws.offlist, cap * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.offcount, 256));
    TRY(hipMalloc((void **)&v->ws.exact_pad, EDK_EXACT_PAD_BYTES));
    v->ws.capacity = cap;
out:
    if (rc) ws_release(v);
    return rc;
}

/* caller holds g_lock: the slot a pass on stream `st` uses */
static struct vslot *ws_pick(struct engine *e, hipStream_t st)
{
    struct vslot *idle = NULL, *lru = &e->vs[0];
    for (int i = 0; i < VERIFY_SLOTS; i++) {
        struct vslot *v = &e->vs[i];
        if (v->stamp && v->last_stream == st) { lru = v; idle = NULL; goto found; }
        if (!idle && (v->stamp == 0 || hipEventQuery(v->free) == hipSuccess)) idle = v;
        if (v->stamp < lru->stamp) lru = v;
    }
found:
    if (idle) lru = idle;
    lru->last_stream = st;
    lru->stamp = ++e->clock;
    return lru;
}

/* everything the engine holds on the device; caller holds g_lock */
static void engine_release(struct engine *e)
{
    (void)hipDeviceSynchronize();
    pipe_release();
    for (int i = 0; i < VERIFY_SLOTS; i++) {
        struct vslot *v = &e->vs[i];
        ws_release(v);
        fws_release(v);
        if (v->ws.side) (void)hipStreamDestroy(v->ws.side);
        if (v->ws.ev_prepared) (void)hipEventDestroy(v->ws.ev_prepared);
        if (v->ws.ev_exact) (void)hipEventDestroy(v->ws.ev_exact);
        if (v->free) (void)hipEventDestroy(v->free);
    }
    (void)hipFree(e->base16); (void)hipFree(e->comb); (void)hipFree(e->comb_img);
    for (int s = 0; s < MARK_SLOTS; s++)
        for (int i = 0; i < 4; i++) if (e->marks[s][i]) (void)hipEventDestroy(e->marks[s][i]);
    memset(e, 0, sizeof(*e));
}

int eddsa_amd_init(int device)
{
    int rc = 0;
    hipDeviceProp_t prop;
    pthread_mutex_lock(&g_lock);
    if (g_eng.ready && g_eng.device == device) goto out;
    if (g_eng.ready) engine_release(&g_eng);   /* re-bind to another device */
    TRY(hipSetDevice(device));
    TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { rc = -100000; goto out; }
    TRY(hipMalloc((void **)&g_eng.base16, (size_t)TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&g_eng.comb, TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&g_eng.comb_img, COMB_IMG_WORDS * sizeof(uint32_t)));
    for (int i = 0; i < VERIFY_SLOTS; i++) {
        /* highest queue priority for the side streams: their few workgroups must be dispatched while
         * k_verify_main still has thousands waiting, not after them */
        int lo = 0, hi = 0;
        struct vslot *v = &g_eng.vs[i];
        TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        TRY(hipStreamCreateWithPriority(&v->ws.side, hipStreamNonBlocking, hi));
        TRY(hipEventCreateWithFlags(&v->ws.ev_prepared, hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&v->ws.ev_exact, hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&v->free, hipEventDisableTiming));
        TRY(hipEventRecord(v->free, NULL));
    }
    g_eng.offcurve_mode = 1;
    for (int s = 0; s < MARK_SLOTS; s++)
        for (int i = 0; i < 4; i++) TRY(hipEventCreate(&g_eng.marks[s][i]));
    TRY(edk_init_tables(g_eng.base16, g_eng.comb, g_eng.comb_img, NULL));
    TRY(hipDeviceSynchronize());
    g_eng.device = device;
    g_eng.ready = 1;
out:
    pthread_mutex_unlock(&g_lock);
    return rc;
}

void eddsa_amd_shutdown(void)
{
    pthread_mutex_lock(&g_lock);
    if (g_eng.ready) engine_release(&g_eng);
    pthread_mutex_unlock(&g_lock);
}

static int ensure_init(void)
{
    int dev = 0;
    if (g_eng.ready) return 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return -(int)e;
    return eddsa_amd_init(dev);
}

int eddsa_amd_dump_tables(uint32_t *base16_words, uint32_t *comb_words)
{
    int rc = ensure_init();
    if (rc) return rc;
    TRY(hipMemcpy(base16_words, g_eng.base16, (size_t)TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    TRY(hipMemcpy(comb_words, g_eng.comb, TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
out:
    return rc;
}

/* How verify treats a public key that does not decode to a curve point (ed_import never fails,
 * reference lib/ed.c:100-149).  EXACT (default): such items are evaluated in the reference's own
 * order of operations, which is the only way to reproduce its bytes there.  REJECT: they are
 * rejected outright; this differs from the reference only if encode(C) == R for a C that depends on
 * SHA-512(R || A || M), i.e. on a fixed point of a random function, and saves the ~1 ms the exact
 * pass costs whenever a batch contains such keys.  ALL (2): every item takes the reference-order
 * path and the windowed evaluation's result is ignored -- slow (latency-bound), for self-checks. */
void eddsa_amd_set_offcurve_mode(int exact)
{
    pthread_mutex_lock(&g_lock);
    g_eng.offcurve_mode = exact == 2 ? 2 : exact != 0;
    pthread_mutex_unlock(&g_lock);
}

/* per-kernel timing of the verify pass, for bench.py's roofline line: HIP events recorded on the
 * launch stream around k_verify_prepare / k_verify_main / k_verify_finish of the LAST chunk */
void eddsa_amd_set_profiling(int on)
{
    g_eng.profiling = on != 0;
    g_eng.marks_used = 0;
}

/* average duration (ms) of each of the three kernels over the passes recorded since profiling was
 * switched on (at most MARK_SLOTS; later passes are not recorded) */
int eddsa_amd_verify_phase_ms(float out[3])
{
    int rc = 0;
    int used = g_eng.marks_used < MARK_SLOTS ? g_eng.marks_used : MARK_SLOTS;
    if (!g_eng.ready || used == 0) return -(int)hipErrorNotReady;
    out[0] = out[1] = out[2] = 0.0f;
    for (int s = 0; s < used; s++) {
        TRY(hipEventSynchronize(g_eng.marks[s][3]));
        for (int i = 0; i < 3; i++) {
            float ms = 0.0f;
            TRY(hipEventElapsedTime(&ms, g_eng.marks[s][i], g_eng.marks[s][i + 1]));
            out[i] += ms / (float)used;
        }
    }
out:
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * device-pointer entry points
 * ---------------------------------------------------------------------------------------- */

/* both device-pointer verify entry points: chunks of at most CHUNK_MAX items through the workspace */
static int verify_dev(uint8_t *ok, const edk_verify_src *all, size_t n, hipStream_t st)
{
    int rc = ensure_init();
    if (rc || n == 0) return rc;
    pthread_mutex_lock(&g_lock);
    struct vslot *v = ws_pick(&g_eng, st);
    rc = ws_reserve(v, n < CHUNK_MAX ? n : CHUNK_MAX);
    if (rc) goto out;
    v->ws.exact_offcurve = g_eng.offcurve_mode;
    /* the slot may have served another stream: order this pass behind its previous one */
    TRY(hipStreamWaitEvent(st, v->free, 0));
    for (size_t done = 0; done < n; done += CHUNK_MAX) {
        size_t m = n - done < CHUNK_MAX ? n - done : CHUNK_MAX;
        edk_verify_src src = *all;
        src.sigs += done * all->sig_stride;
        src.pubs += done * all->pub_stride;
        if (all->msg_off) src.msg_off += done; else src.msgs += done * all->msg_stride;
        hipEvent_t *marks = NULL;
        if (g_eng.profiling && g_eng.marks_used < MARK_SLOTS) marks = g_eng.marks[g_eng.marks_used++];
        TRY(edk_verify(ok + done, &src, m, g_eng.base16, &v->ws, marks, st));
    }
    TRY(hipEventRecord(v->free, st));
out:
    pthread_mutex_unlock(&g_lock);
    return rc;
}

int ed25519_verify_batch_dev(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs, const uint8_t *msgs,
                             const uint64_t *msg_off, size_t msg_len, size_t n, void *stream)
{
    const edk_verify_src src = { sigs, pubs, msgs, msg_off, msg_len, 64, 32, msg_len };
    return verify_dev(ok, &src, n, (hipStream_t)stream);
}

/* fixed-size records: see include/eddsa_amd.h */
static int records_ok(size_t stride, size_t sig_off, size_t pub_off, size_t msg_off, size_t msg_len)
{
    return sig_off <= stride && 64 <= stride - sig_off && pub_off <= stride && 32 <= stride - pub_off &&
           msg_off <= stride && msg_len <= stride - msg_off;
}

int ed25519_verify_records_dev(uint8_t *ok, const uint8_t *records, size_t stride, size_t sig_off, size_t pub_off,
                               size_t msg_off, size_t msg_len, size_t n, void *stream)
{
    if (!records_ok(stride, sig_off, pub_off, msg_off, msg_len)) return -(int)hipErrorInvalidValue;
    const edk_verify_src src = { records + sig_off, records + pub_off, records + msg_off, NULL, msg_len,
                                 stride, stride, stride };
    return verify_dev(ok, &src, n, (hipStream_t)stream);
}

/* the three fixed-base operations share one driver: chunks of at most CHUNK_MAX items through fws */
typedef hipError_t (*fixed_step)(size_t done, size_t m, const void *ctx, const edk_fixed_ws *fws, hipStream_t st);

static int run_fixed(size_t n, fixed_step step, const void *ctx, void *stream)
{
    int rc = ensure_init();
    hipStream_t st = (hipStream_t)stream;
    if (rc || n == 0) return rc;
    pthread_mutex_lock(&g_lock);
    struct vslot *v = ws_pick(&g_eng, st);
    rc = fws_reserve(v, n < CHUNK_MAX ? n : CHUNK_MAX);
    if (rc) goto out;
    TRY(hipStreamWaitEvent(st, v->free, 0));
    for (size_t done = 0; done < n; done += CHUNK_MAX)
        TRY(step(done, n - done < CHUNK_MAX ? n - done : CHUNK_MAX, ctx, &v->fws, st));
    TRY(hipEventRecord(v->free, st));
out:
    pthread_mutex_unlock(&g_lock);
    return rc;
}

struct sign_ctx { uint8_t *sigs; const uint8_t *secs, *pubs, *msgs; const uint64_t *msg_off; size_t msg_len; };

static hipError_t sign_step(size_t done, size_t m, const void *vctx, const edk_fixed_ws *fws, hipStream_t st)
{
    const struct sign_ctx *c = (const struct sign_ctx *)vctx;
    const uint8_t *mp = c->msg_off ? c->msgs : c->msgs + done * c->msg_len;
    const uint64_t *op = c->msg_off ? c->msg_off + done : NULL;
    return edk_sign(c->sigs + 64 * done, c->secs + 32 * done, c->pubs + 32 * done, mp, op, c->msg_len, m,
                    g_eng.comb_img, fws, st);
}

int ed25519_sign_batch_dev(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                           const uint64_t *msg_off, size_t msg_len, size_t n, void *stream)
{
    struct sign_ctx c = { sigs, secs, pubs, msgs, msg_off, msg_len };
    return run_fixed(n, sign_step, &c, stream);
}

struct io_ctx { uint8_t *out; const uint8_t *in; };

static hipError_t genpub_step(size_t done, size_t m, const void *vctx, const edk_fixed_ws *fws, hipStream_t st)
{
    const struct io_ctx *c = (const struct io_ctx *)vctx;
    return edk_genpub(c->out + 32 * done, c->in + 32 * done, m, g_eng.comb_img, fws, st);
}

int ed25519_genpub_batch_dev(uint8_t *pubs, const uint8_t *secs, size_t n, void *stream)
{
    struct io_ctx c = { pubs, secs };
    return run_fixed(n, genpub_step, &c, stream);
}

struct io2_ctx { uint8_t *out; const uint8_t *a, *b; };

static hipError_t x25519_step(size_t done, size_t m, const void *vctx, const edk_fixed_ws *fws, hipStream_t st)
{
    const struct io2_ctx *c = (const struct io2_ctx *)vctx;
    return edk_x25519(c->out + 32 * done, c->a + 32 * done, c->b + 32 * done, m, fws, st);
}

int x25519_batch_dev(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n, void *stream)
{
    struct io2_ctx c = { out, scalars, points };
    return run_fixed(n, x25519_step, &c, stream);
}

static hipError_t xbase_step(size_t done, size_t m, const void *vctx, const edk_fixed_ws *fws, hipStream_t st)
{
    const struct io_ctx *c = (const struct io_ctx *)vctx;
    return edk_x25519_base(c->out + 32 * done, c->in + 32 * done, m, g_eng.comb_img, fws, st);
}

int x25519_base_batch_dev(uint8_t *out, const uint8_t *scalars, size_t n, void *stream)
{
    struct io_ctx c = { out, scalars };
    return run_fixed(n, xbase_step, &c, stream);
}

int pk_ed25519_to_x25519_batch_dev(uint8_t *out, const uint8_t *in, size_t n, void *stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    TRY(edk_pk_to_x(out, in, n, (hipStream_t)stream));
out:
    return rc;
}

int sk_ed25519_to_x25519_batch_dev(uint8_t *out, const uint8_t *in, size_t n, void *stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    TRY(edk_sk_to_x(out, in, n, (hipStream_t)stream));
out:
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * host-pointer entry points: a streaming pipeline over chunks of PIPE_CHUNK items.
 *
 * Three streams: `up` copies chunk k+1 host -> HBM while `exec` runs the kernels of chunk k and
 * `down` copies the results of chunk k-1 back, on two alternating sets of device input buffers
 * that persist across calls (grown on demand).  With pageable caller memory the HIP runtime stages
 * the copies itself and blocks the calling thread for their duration, which is why the download of
 * chunk k-1 is issued only after the kernels of chunk k were launched; with pinned caller memory
 * (hipHostMalloc / hipHostRegister) the copies are asynchronous as well.
 * Ragged messages (msg_off != NULL) go through the same buffers as a single chunk.
 * ---------------------------------------------------------------------------------------- */

#define PIPE_CHUNK ((size_t)1 << 18)   /* 1024 blocks of 256 lanes: one full residency of the chip */
/* verify: two residencies per chunk, so that the exact path's chain for off-curve keys (4 ms beside
 * the main kernel, tools/verify_sizes.py) stays hidden behind k_verify_main as it is in one big pass */
#define PIPE_CHUNK_VERIFY ((size_t)1 << 19)
#define PIPE_MAX_IN 3

struct hjob {
    int n_in; const uint8_t *in[PIPE_MAX_IN]; size_t in_w[PIPE_MAX_IN];   /* fixed-width inputs */
    int has_msgs; const uint8_t *msgs; const uint64_t *msg_off; size_t msg_len;
    uint8_t *out; size_t out_w;
    int (*run)(const struct hjob *j, uint8_t *d_out, uint8_t *const d_in[PIPE_MAX_IN], const uint8_t *d_msgs,
               const uint64_t *d_off, size_t msg_len, size_t m, void *stream);
    size_t rec_sig, rec_pub, rec_msg;          /* records: offsets inside in[0]'s items (in_w[0] = stride) */
    size_t chunk;                              /* items per pipeline stage (0: PIPE_CHUNK) */
};

struct pipe {
    int ready;
    hipStream_t up, exec, down;
    hipEvent_t in_ready[2], exec_done[2], slot_free[2];
    void *d_in[2][PIPE_MAX_IN]; size_t in_cap[2][PIPE_MAX_IN];
    void *d_msgs[2]; size_t msgs_cap[2];
    void *d_off; size_t off_cap;
    void *d_out; size_t out_cap;
};
static struct pipe g_pipe;
static pthread_mutex_t g_pipe_lock = PTHREAD_MUTEX_INITIALIZER;

static int pipe_grow(void **buf, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    if (*buf) { (void)hipFree(*buf); *buf = NULL; *cap = 0; }
    hipError_t e = hipMalloc(buf, need < 256 ? 256 : need);
    if (e != hipSuccess) return -(int)e;
    *cap = need < 256 ? 256 : need;
    return 0;
}

static int pipe_init(void)
{
    int rc = 0;
    if (g_pipe.ready) return 0;
    TRY(hipStreamCreateWithFlags(&g_pipe.up, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&g_pipe.exec, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&g_pipe.down, hipStreamNonBlocking));
    for (int s = 0; s < 2; s++) {
        TRY(hipEventCreateWithFlags(&g_pipe.in_ready[s], hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&g_pipe.exec_done[s], hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&g_pipe.slot_free[s], hipEventDisableTiming));
    }
    g_pipe.ready = 1;
out:
    return rc;
}

static void pipe_release(void)
{
    if (!g_pipe.ready) return;
    for (int s = 0; s < 2; s++) {
        for (int i = 0; i < PIPE_MAX_IN; i++) if (g_pipe.d_in[s][i]) (void)hipFree(g_pipe.d_in[s][i]);
        if (g_pipe.d_msgs[s]) (void)hipFree(g_pipe.d_msgs[s]);
        (void)hipEventDestroy(g_pipe.in_ready[s]); (void)hipEventDestroy(g_pipe.exec_done[s]);
        (void)hipEventDestroy(g_pipe.slot_free[s]);
    }
    if (g_pipe.d_off) (void)hipFree(g_pipe.d_off);
    if (g_pipe.d_out) (void)hipFree(g_pipe.d_out);
    (void)hipStreamDestroy(g_pipe.up); (void)hipStreamDestroy(g_pipe.exec); (void)hipStreamDestroy(g_pipe.down);
    memset(&g_pipe, 0, sizeof(g_pipe));
}

static int pipe_run(const struct hjob *j, size_t n)
{
    int rc = ensure_init();
    if (rc || n == 0) return rc;
    pthread_mutex_lock(&g_pipe_lock);
    rc = pipe_init();
    if (rc) goto out;
    {
        const int ragged = j->has_msgs && j->msg_off != NULL;
        const size_t stage = j->chunk ? j->chunk : PIPE_CHUNK;
        const size_t chunk = ragged ? n : (n < stage ? n : stage);
        const size_t nchunks = (n + chunk - 1) / chunk;
        if ((rc = pipe_grow(&g_pipe.d_out, &g_pipe.out_cap, n * j->out_w))) goto out;
        if (ragged && (rc = pipe_grow(&g_pipe.d_off, &g_pipe.off_cap, (n + 1) * sizeof(uint64_t)))) goto out;
        for (int s = 0; s < (nchunks > 1 ? 2 : 1); s++) {
            for (int i = 0; i < j->n_in; i++)
                if ((rc = pipe_grow(&g_pipe.d_in[s][i], &g_pipe.in_cap[s][i], chunk * j->in_w[i]))) goto out;
            if (j->has_msgs) {
                const size_t need = ragged ? (size_t)j->msg_off[n] : chunk * j->msg_len;
                if ((rc = pipe_grow(&g_pipe.d_msgs[s], &g_pipe.msgs_cap[s], need))) goto out;
            }
        }
        if (ragged) TRY(hipMemcpyAsync(g_pipe.d_off, j->msg_off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g_pipe.up));
        for (size_t k = 0; k < nchunks; k++) {
            const int s = (int)(k & 1);
            const size_t lo = k * chunk, m = n - lo < chunk ? n - lo : chunk;
            /* upload chunk k into slot s once the kernels of chunk k-2 have consumed it */
            if (k >= 2) TRY(hipStreamWaitEvent(g_pipe.up, g_pipe.exec_done[s], 0));
            for (int i = 0; i < j->n_in; i++)
                TRY(hipMemcpyAsync(g_pipe.d_in[s][i], j->in[i] + lo * j->in_w[i], m * j->in_w[i], hipMemcpyHostToDevice, g_pipe.up));
            if (j->has_msgs) {
                const size_t bytes = ragged ? (size_t)j->msg_off[n] : m * j->msg_len;
                const uint8_t *src = ragged ? j->msgs : j->msgs + lo * j->msg_len;
                if (bytes) TRY(hipMemcpyAsync(g_pipe.d_msgs[s], src, bytes, hipMemcpyHostToDevice, g_pipe.up));
            }
            TRY(hipEventRecord(g_pipe.in_ready[s], g_pipe.up));
            /* kernels of chunk k */
            TRY(hipStreamWaitEvent(g_pipe.exec, g_pipe.in_ready[s], 0));
            rc = j->run(j, (uint8_t *)g_pipe.d_out + lo * j->out_w, (uint8_t *const *)g_pipe.d_in[s],
                        (const uint8_t *)g_pipe.d_msgs[s], ragged ? (const uint64_t *)g_pipe.d_off : NULL,
                        j->msg_len, m, g_pipe.exec);
            if (rc) goto out;
            TRY(hipEventRecord(g_pipe.exec_done[s], g_pipe.exec));
            /* download chunk k-1 (its kernels were launched one iteration ago) */
            if (k >= 1) {
                const size_t plo = (k - 1) * chunk;
                TRY(hipStreamWaitEvent(g_pipe.down, g_pipe.exec_done[s ^ 1], 0));
                TRY(hipMemcpyAsync(j->out + plo * j->out_w, (uint8_t *)g_pipe.d_out + plo * j->out_w, chunk * j->out_w,
                                   hipMemcpyDeviceToHost, g_pipe.down));
            }
        }
        {
            const size_t plo = (nchunks - 1) * chunk;
            TRY(hipStreamWaitEvent(g_pipe.down, g_pipe.exec_done[(nchunks - 1) & 1], 0));
            TRY(hipMemcpyAsync(j->out + plo * j->out_w, (uint8_t *)g_pipe.d_out + plo * j->out_w, (n - plo) * j->out_w,
                               hipMemcpyDeviceToHost, g_pipe.down));
        }
        TRY(hipStreamSynchronize(g_pipe.down));
        TRY(hipStreamSynchronize(g_pipe.exec));
        TRY(hipStreamSynchronize(g_pipe.up));
    }
out:
    if (rc) { (void)hipStreamSynchronize(g_pipe.up); (void)hipStreamSynchronize(g_pipe.exec); (void)hipStreamSynchronize(g_pipe.down); }
    pthread_mutex_unlock(&g_pipe_lock);
    return rc;
}

#define RUN_ARGS const struct hjob *j, uint8_t *d_out, uint8_t *const d_in[PIPE_MAX_IN], const uint8_t *d_msgs, \
                 const uint64_t *d_off, size_t msg_len, size_t m, void *stream
static int run_verify(RUN_ARGS)
{
    (void)j;
    return ed25519_verify_batch_dev(d_out, d_in[0], d_in[1], d_msgs, d_off, msg_len, m, stream);
}
static int run_verify_records(RUN_ARGS)
{
    (void)d_msgs; (void)d_off;
    return ed25519_verify_records_dev(d_out, d_in[0], j->in_w[0], j->rec_sig, j->rec_pub, j->rec_msg, msg_len, m, stream);
}
static int run_sign(RUN_ARGS)
{
    (void)j;
    return ed25519_sign_batch_dev(d_out, d_in[0], d_in[1], d_msgs, d_off, msg_len, m, stream);
}
static int run_x25519(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    return x25519_batch_dev(d_out, d_in[0], d_in[1], m, stream);
}
#define RUN_1IN(name, devfn) \
static int name(RUN_ARGS) \
{ (void)j; (void)d_msgs; (void)d_off; (void)msg_len; return devfn(d_out, d_in[0], m, stream); }
RUN_1IN(run_genpub, ed25519_genpub_batch_dev)
RUN_1IN(run_xbase, x25519_base_batch_dev)
RUN_1IN(run_pk_to_x, pk_ed25519_to_x25519_batch_dev)
RUN_1IN(run_sk_to_x, sk_ed25519_to_x25519_batch_dev)

int ed25519_verify_batch(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs, const uint8_t *msgs,
                         const uint64_t *msg_off, size_t msg_len, size_t n)
{
    struct hjob j = { 2, { sigs, pubs, NULL }, { 64, 32, 0 }, 1, msgs, msg_off, msg_len, ok, 1, run_verify, 0, 0, 0, PIPE_CHUNK_VERIFY };
    return pipe_run(&j, n);
}

int ed25519_verify_records(uint8_t *ok, const uint8_t *records, size_t stride, size_t sig_off, size_t pub_off,
                           size_t msg_off, size_t msg_len, size_t n)
{
    if (!records_ok(stride, sig_off, pub_off, msg_off, msg_len)) return -(int)hipErrorInvalidValue;
    struct hjob j = { 1, { records, NULL, NULL }, { stride, 0, 0 }, 0, NULL, NULL, msg_len, ok, 1, run_verify_records,
                      sig_off, pub_off, msg_off, PIPE_CHUNK_VERIFY };
    return pipe_run(&j, n);
}

int ed25519_sign_batch(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                       const uint64_t *msg_off, size_t msg_len, size_t n)
{
    struct hjob j = { 2, { secs, pubs, NULL }, { 32, 32, 0 }, 1, msgs, msg_off, msg_len, sigs, 64, run_sign, 0, 0, 0, 0 };
    return pipe_run(&j, n);
}

int x25519_batch(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n)
{
    struct hjob j = { 2, { scalars, points, NULL }, { 32, 32, 0 }, 0, NULL, NULL, 0, out, 32, run_x25519, 0, 0, 0, 0 };
    return pipe_run(&j, n);
}

static int run_1in(int (*run)(RUN_ARGS), uint8_t *out, const uint8_t *in, size_t n)
{
    struct hjob j = { 1, { in, NULL, NULL }, { 32, 0, 0 }, 0, NULL, NULL, 0, out, 32, run, 0, 0, 0, 0 };
    return pipe_run(&j, n);
}

int ed25519_genpub_batch(uint8_t *pubs, const uint8_t *secs, size_t n) { return run_1in(run_genpub, pubs, secs, n); }
int x25519_base_batch(uint8_t *out, const uint8_t *scalars, size_t n) { return run_1in(run_xbase, out, scalars, n); }
int pk_ed25519_to_x25519_batch(uint8_t *out, const uint8_t *in, size_t n) { return run_1in(run_pk_to_x, out, in, n); }
int sk_ed25519_to_x25519_batch(uint8_t *out, const uint8_t *in, size_t n) { return run_1in(run_sk_to_x, out, in, n); }

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
