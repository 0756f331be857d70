/*
 * eddsa_amd.c - host side of libeddsa_amd.so, plain C.
 *
 * Exports the thirteen symbols of the reference's public header (include/eddsa.h, reference
 * lib/eddsa.h:44-113) and the batched entry points of include/eddsa_amd.h.  Everything is
 * computed by the HIP kernels in kernels.hip; there is no CPU arithmetic in this file and no
 * fallback: without a usable gfx950 device every entry point fails (batch API: negative return;
 * eddsa.h API, which has no error channel: message on stderr + abort()).
 *
 * State: one `struct engine` per HIP device (tables, workspace pool, staging pipeline), created on
 * first use.  A device-pointer call runs on the device its output buffer lives on; a host-pointer
 * call runs on the default device (eddsa_amd_init) or, for the *_multi entry points, on every
 * device of the set bound by eddsa_amd_init_devices.  Every call makes its device current for its
 * own duration and restores the caller's.
 *
 * Locks, always taken in this order:  g_table (rwlock: read for the duration of every call, write
 * to create / destroy engines)  ->  g_rccl_lk (the multi-device device-pointer call)  ->  engine.pipe_lk
 * (host staging pipeline)  ->  engine.lk (workspace pool, profiling marks).
 */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "eddsa.h"
#include "eddsa_amd.h"
#include "eddsa_kernels.h"

#define CHUNK_MAX ((size_t)1 << 20)   /* verify items per workspace pass: 1.6 GB of HBM workspace */
#define MARK_SLOTS 256                /* profiled verify passes kept for eddsa_amd_verify_phase_ms */
#define MAX_DEVICES 64

#define ERR_NOT_GFX950 (-100000)
#define ERR_RCCL_MISSING (-100001)
#define ERR_RCCL_BASE (-200000)       /* ERR_RCCL_BASE - ncclResult_t */

/* Workspaces: a small pool, so that passes issued on DIFFERENT streams (host threads that
 * each own a stream) overlap on the GPU instead of queueing behind one workspace.  A stream keeps the
 * slot it used last (passes on one stream are ordered anyway, and the slot has the right size);
 * another stream takes an idle slot, or the least recently used one. */
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

/* host-pointer entry points: staging buffers and streams of the streaming pipeline (pipe_run) */
#define PIPE_MAX_IN 3
struct pipe {
    int ready;
    hipStream_t up, exec, down;
    hipEvent_t in_ready[2], exec_done[2], slot_free[2];
    void *d_in[2][PIPE_MAX_IN]; size_t in_cap[2][PIPE_MAX_IN];
    void *d_msgs[2]; size_t msgs_cap[2];
    void *d_off; size_t off_cap;
    void *d_out; size_t out_cap;
};

struct engine {
    int device;
    pthread_mutex_t lk, pipe_lk;
    pthread_cond_t slot_cv;            /* signalled (under lk) when a busy workspace slot is released */
    uint32_t *base16, *comb;           /* generated base-point tables (HBM) */
    uint32_t *comb_img;                /* the comb as the point kernels stage it in LDS (lanes.h: comb_select) */
    struct vslot vs[VERIFY_SLOTS];
    unsigned long clock;
    int marks_used;                   /* passes recorded since profiling was switched on */
    hipEvent_t marks[MARK_SLOTS][4];
    struct pipe pipe;
};

static pthread_rwlock_t g_table = PTHREAD_RWLOCK_INITIALIZER;
static struct engine *g_eng[MAX_DEVICES];
static int g_default = -1;            /* device of the host-pointer entry points; -1: the caller's current device at first use */
static int g_verify_algo = 0;         /* eddsa_amd_set_verify_algo: 0 by pass size (default), 1 full-length windows, 2 half-length scalars */
static int g_offcurve_mode = 1;       /* eddsa_amd_set_offcurve_mode: 0 reject, 1 exact (default), 2 all exact */
static int g_profiling;               /* record marks around the three verify kernels */
static int g_fail_next_host_call;      /* eddsa_amd_debug_fail_next_host_call: test hook for the error path of pipe_run_on */
static size_t g_rlc_min_items = (size_t)3 << 17;   /* eddsa_amd_set_rlc_min_items: smaller calls go to the per-item kernels */

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
static struct multi g_multi;
static pthread_mutex_t g_rccl_lk = PTHREAD_MUTEX_INITIALIZER;   /* one grouped RCCL call at a time (taken after g_table) */
#define NCCL_UINT8 1                   /* ncclUint8, rccl.h */

#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = -(int)e_; goto out; } } while (0)

const char *eddsa_amd_strerror(int err)
{
    if (err == 0) return "success";
    if (err == ERR_NOT_GFX950) return "eddsa_amd: device is not gfx950 (MI355X); no code object for it";
    if (err == ERR_RCCL_MISSING) return "eddsa_amd: librccl.so.1 could not be loaded (needed for the multi-device result gather)";
    if (err <= ERR_RCCL_BASE) {
        if (g_multi.GetErrorString) return g_multi.GetErrorString(ERR_RCCL_BASE - err);
        return "eddsa_amd: RCCL error";
    }
    return hipGetErrorString((hipError_t)(-err));
}

/* ------------------------------------------------------------------------------------------
 * engines
 * ---------------------------------------------------------------------------------------- */

/* device buffers that held secrets (or may have) are zeroed before they go back to the allocator:
 * the reference wipes after its secret-key operations (lib/ed25519-sha512.c:77,136, lib/x25519.c:208,221) */
static void wipe_free(void *p, size_t bytes)
{
    if (!p) return;
    if (bytes) { (void)hipMemset(p, 0, bytes); (void)hipDeviceSynchronize(); }
    (void)hipFree(p);
}

static void ws_release(struct vslot *v)
{
    if (v->ws.digits) (void)hipFree(v->ws.digits);
    if (v->ws.table) (void)hipFree(v->ws.table);
    if (v->ws.acc) (void)hipFree(v->ws.acc);
    if (v->ws.flags) (void)hipFree(v->ws.flags);
    if (v->ws.hdigits) (void)hipFree(v->ws.hdigits);
    if (v->ws.rtable) (void)hipFree(v->ws.rtable);
    if (v->ws.offlist) (void)hipFree(v->ws.offlist);
    if (v->ws.offcount) (void)hipFree(v->ws.offcount);
    if (v->ws.exact_pad) (void)hipFree(v->ws.exact_pad);
    v->ws.capacity = 0;
    v->ws.digits = v->ws.table = v->ws.acc = v->ws.offlist = v->ws.offcount = v->ws.exact_pad = NULL;
    v->ws.hdigits = v->ws.rtable = NULL;
    v->ws.flags = NULL;
}

static void fws_release(struct vslot *v)
{
    wipe_free(v->fws.acc, v->fws.capacity * ACC_WORDS * sizeof(uint32_t));
    wipe_free(v->fws.aux, v->fws.capacity * 16 * sizeof(uint32_t));
    memset(&v->fws, 0, sizeof(v->fws));
}

static void rws_release(struct vslot *v)
{
    if (v->rws.base) (void)hipFree(v->rws.base);
    if (v->rws.host_gok) (void)hipHostFree(v->rws.host_gok);
    memset(&v->rws, 0, sizeof(v->rws));
}

static size_t round_capacity(size_t items)
{
    size_t cap = (items + VERIFY_TILE - 1) / VERIFY_TILE * VERIFY_TILE;
    return (cap + 8 * VERIFY_TILE - 1) / (8 * VERIFY_TILE) * (8 * VERIFY_TILE);   /* whole finish blocks */
}

/* caller holds e->lk; st = the stream the pass is about to use */
static int fws_reserve(struct vslot *v, size_t items, hipStream_t st)
{
    int rc = 0;
    const size_t cap = round_capacity(items);
    if (cap <= v->fws.capacity) return 0;
    TRY(hipEventSynchronize(v->free));
    fws_release(v);
    TRY(hipMalloc((void **)&v->fws.acc, cap * ACC_WORDS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->fws.aux, cap * 16 * sizeof(uint32_t)));
    /* recycled memory: start clean.  On the pass's own stream: a hipMemset on the null stream is not ordered
     * with the kernels of a non-blocking stream and could land after their first stores */
    TRY(hipMemsetAsync(v->fws.acc, 0, cap * ACC_WORDS * sizeof(uint32_t), st));
    TRY(hipMemsetAsync(v->fws.aux, 0, cap * 16 * sizeof(uint32_t), st));
    v->fws.capacity = cap;
out:
    if (rc) fws_release(v);
    return rc;
}

/* caller holds e->lk */
static int ws_reserve(struct vslot *v, size_t items)
{
    int rc = 0;
    const size_t cap = round_capacity(items);
    if (cap <= v->ws.capacity) return 0;
    /* the old buffers may still be in use by enqueued kernels */
    TRY(hipEventSynchronize(v->free));
    ws_release(v);
    TRY(hipMalloc((void **)&v->ws.digits, cap * 16 * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.table, cap / VERIFY_TILE * (size_t)VERIFY_TABLE_WORDS_PER_TILE * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.acc, cap * ACC_WORDS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.hdigits, cap * EDK_HALF_DIGIT_WORDS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.rtable, cap / VERIFY_TILE * (size_t)VERIFY_TABLE_WORDS_PER_TILE * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.flags, cap));
    TRY(hipMalloc((void **)&v->ws.offlist, cap * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.offcount, 256));
    TRY(hipMemset(v->ws.offcount, 0, 256));
    TRY(hipStreamSynchronize(NULL));      /* the pass's stream does not wait for the null stream */
    TRY(hipMalloc((void **)&v->ws.exact_pad, EDK_EXACT_PAD_BYTES));
    v->ws.capacity = cap;
out:
    if (rc) ws_release(v);
    return rc;
}

/* caller holds e->lk */
static int rws_reserve(struct vslot *v, size_t items)
{
    int rc = 0;
    const size_t cap = round_capacity(items);
    if (cap <= v->rws.capacity) return 0;
    TRY(hipEventSynchronize(v->free));
    rws_release(v);
    TRY(hipMalloc((void **)&v->rws.base, edk_rlc_ws_bytes(cap)));
    TRY(hipHostMalloc(&v->rws.host_gok, EDK_RLC_HOST_BYTES, hipHostMallocDefault));
    v->rws.capacity = cap;
out:
    if (rc) rws_release(v);
    return rc;
}

/* caller holds e->lk: the slot a pass on stream `st` uses.  Slots marked busy (rlc_on, while it waits for its
 * stream outside the lock) are not handed out; when all are, the caller waits for one to be released. */
static struct vslot *ws_pick(struct engine *e, hipStream_t st)
{
    for (;;) {
        struct vslot *idle = NULL, *lru = NULL;
        for (int i = 0; i < VERIFY_SLOTS; i++) {
            struct vslot *v = &e->vs[i];
            if (v->busy) continue;
            if (v->stamp && v->last_stream == st) { lru = v; idle = NULL; break; }
            if (!idle && (v->stamp == 0 || hipEventQuery(v->free) == hipSuccess)) idle = v;
            if (!lru || v->stamp < lru->stamp) lru = v;
        }
        if (idle) lru = idle;
        if (lru) {
            lru->last_stream = st;
            lru->stamp = ++e->clock;
            return lru;
        }
        pthread_cond_wait(&e->slot_cv, &e->lk);
    }
}

static void pipe_release(struct pipe *p);

/* everything an engine holds on its device.  Caller holds g_table for writing (no call is in
 * flight) and has made e->device current. */
static void engine_destroy(struct engine *e)
{
    (void)hipDeviceSynchronize();
    pipe_release(&e->pipe);
    for (int i = 0; i < VERIFY_SLOTS; i++) {
        struct vslot *v = &e->vs[i];
        ws_release(v);
        fws_release(v);
        rws_release(v);
        if (v->ws.side) (void)hipStreamDestroy(v->ws.side);
        if (v->ws.ev_prepared) (void)hipEventDestroy(v->ws.ev_prepared);
        if (v->ws.ev_exact) (void)hipEventDestroy(v->ws.ev_exact);
        if (v->free) (void)hipEventDestroy(v->free);
    }
    if (e->base16) (void)hipFree(e->base16);
    if (e->comb) (void)hipFree(e->comb);
    if (e->comb_img) (void)hipFree(e->comb_img);
    for (int s = 0; s < MARK_SLOTS; s++)
        for (int i = 0; i < 4; i++) if (e->marks[s][i]) (void)hipEventDestroy(e->marks[s][i]);
    pthread_mutex_destroy(&e->lk);
    pthread_mutex_destroy(&e->pipe_lk);
    pthread_cond_destroy(&e->slot_cv);
    free(e);
}

/* caller holds g_table for writing */
static int engine_create(int device)
{
    int rc = 0, saved = -1;
    hipDeviceProp_t prop;
    struct engine *e;
    if (device < 0 || device >= MAX_DEVICES) return -(int)hipErrorInvalidDevice;
    if (g_eng[device]) return 0;
    e = (struct engine *)calloc(1, sizeof(*e));
    if (!e) return -(int)hipErrorOutOfMemory;
    e->device = device;
    pthread_mutex_init(&e->lk, NULL);
    pthread_mutex_init(&e->pipe_lk, NULL);
    pthread_cond_init(&e->slot_cv, NULL);
    (void)hipGetDevice(&saved);
    TRY(hipSetDevice(device));
    TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { rc = ERR_NOT_GFX950; goto out; }
    TRY(hipMalloc((void **)&e->base16, (size_t)2 * TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&e->comb, TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&e->comb_img, COMB_IMG_WORDS * sizeof(uint32_t)));
    for (int i = 0; i < VERIFY_SLOTS; i++) {
        /* highest queue priority for the side streams: their few workgroups must be dispatched while
         * k_verify_main still has thousands waiting, not after them */
        int lo = 0, hi = 0;
        struct vslot *v = &e->vs[i];
        TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        TRY(hipStreamCreateWithPriority(&v->ws.side, hipStreamNonBlocking, hi));
        TRY(hipEventCreateWithFlags(&v->ws.ev_prepared, hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&v->ws.ev_exact, hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&v->free, hipEventDisableTiming));
        TRY(hipEventRecord(v->free, NULL));
    }
    for (int s = 0; s < MARK_SLOTS; s++)
        for (int i = 0; i < 4; i++) TRY(hipEventCreate(&e->marks[s][i]));
    TRY(edk_init_tables(e->base16, e->comb, e->comb_img, NULL));
    TRY(hipDeviceSynchronize());
    g_eng[device] = e;
    e = NULL;
out:
    if (e) engine_destroy(e);          /* a half-built engine leaks nothing */
    if (saved >= 0) (void)hipSetDevice(saved);
    return rc;
}

/* Every call brackets its work with enter()/leave(): enter() resolves the engine (creating it on
 * first use), holds g_table for reading and makes the engine's device current for the calling
 * thread; leave() restores the caller's device.  device < 0: the default device. */
struct call { struct engine *e; int saved; };

static int enter(struct call *c, int device)
{
    c->e = NULL;
    c->saved = -1;
    for (int attempt = 0; attempt < 2; attempt++) {
        pthread_rwlock_rdlock(&g_table);
        int dev = device >= 0 ? device : g_default;
        if (dev < 0) {                     /* never bound: the caller's current device becomes the default */
            hipError_t er = hipGetDevice(&dev);
            if (er != hipSuccess) { pthread_rwlock_unlock(&g_table); return -(int)er; }
        }
        if (dev >= MAX_DEVICES) { pthread_rwlock_unlock(&g_table); return -(int)hipErrorInvalidDevice; }
        if (g_eng[dev]) {
            hipError_t er;
            c->e = g_eng[dev];
            (void)hipGetDevice(&c->saved);
            er = hipSetDevice(dev);
            if (er != hipSuccess) { pthread_rwlock_unlock(&g_table); return -(int)er; }
            return 0;
        }
        pthread_rwlock_unlock(&g_table);
        pthread_rwlock_wrlock(&g_table);
        int rc = engine_create(dev);
        if (rc == 0 && device < 0 && g_default < 0) g_default = dev;
        pthread_rwlock_unlock(&g_table);
        if (rc) return rc;
    }
    return -(int)hipErrorNotReady;         /* shut down again between the two steps */
}

static void leave(struct call *c)
{
    if (c->saved >= 0) (void)hipSetDevice(c->saved);
    pthread_rwlock_unlock(&g_table);
}

/* the device a device-pointer call runs on: where its output buffer lives */
static int device_of(const void *p, int *device)
{
    hipPointerAttribute_t a;
    hipError_t er = hipPointerGetAttributes(&a, p);
    if (er != hipSuccess) { (void)hipGetLastError(); return -(int)hipErrorInvalidValue; }
    if (a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeManaged) return -(int)hipErrorInvalidValue;
    *device = a.device;
    return 0;
}

int eddsa_amd_init(int device)
{
    struct call c;
    int rc;
    if (device < 0) return -(int)hipErrorInvalidDevice;
    rc = enter(&c, device);
    if (rc) return rc;
    leave(&c);
    pthread_rwlock_wrlock(&g_table);
    g_default = device;
    pthread_rwlock_unlock(&g_table);
    return 0;
}

static void multi_release(void)
{
    for (int i = 0; i < g_multi.n; i++)
        if (g_multi.comm[i] && g_multi.CommDestroy) (void)g_multi.CommDestroy(g_multi.comm[i]);
    void *h = g_multi.rccl;
    memset(&g_multi, 0, sizeof(g_multi));
    g_multi.rccl = h;                  /* the library stays loaded; its symbols are looked up again */
}

void eddsa_amd_shutdown(void)
{
    int saved = -1;
    pthread_rwlock_wrlock(&g_table);
    (void)hipGetDevice(&saved);
    multi_release();
    for (int d = 0; d < MAX_DEVICES; d++) {
        if (!g_eng[d]) continue;
        (void)hipSetDevice(d);
        engine_destroy(g_eng[d]);
        g_eng[d] = NULL;
    }
    g_default = -1;
    if (saved >= 0) (void)hipSetDevice(saved);
    pthread_rwlock_unlock(&g_table);
}

int eddsa_amd_dump_tables(uint32_t *base16_words, uint32_t *comb_words)
{
    struct call c;
    int rc = enter(&c, -1);
    if (rc) return rc;
    TRY(hipMemcpy(base16_words, c.e->base16, (size_t)TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    TRY(hipMemcpy(comb_words, c.e->comb, TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
out:
    leave(&c);
    return rc;
}

/* Which evaluation ed25519_verify* uses (same verdicts; a measurement and test aid).  0 (default): half-length
 * scalars (csrc/halve.h), four lanes per item up to 2^15 items and one above; 1: full-length windows always;
 * 2: half-length scalars with one lane per item always. */
void eddsa_amd_set_verify_algo(int algo)
{
    pthread_rwlock_wrlock(&g_table);
    g_verify_algo = algo == 1 || algo == 2 ? algo : 0;
    pthread_rwlock_unlock(&g_table);
}

/* diagnostic for the tests: the device's pair search (csrc/halve.h) on n given scalars t (32 bytes each, < l);
 * out48: v (20 bytes) | |u| (20) | u < 0 (1) | found (1) | padding (6) per item.  wide: the bound 2^138 of passes
 * below 2^18 items instead of 2^134.  Host pointers. */
int eddsa_amd_debug_halve(uint8_t *out48, const uint8_t *t32, size_t n, int wide)
{
    struct call c;
    uint8_t *d_t = NULL, *d_o = NULL;
    int rc = enter(&c, -1);
    if (rc) return rc;
    if (n == 0) goto out;
    TRY(hipMalloc((void **)&d_t, n * 32));
    TRY(hipMalloc((void **)&d_o, n * 48));
    TRY(hipMemcpy(d_t, t32, n * 32, hipMemcpyHostToDevice));
    TRY(edk_debug_halve(d_o, d_t, n, wide, NULL));
    TRY(hipMemcpy(out48, d_o, n * 48, hipMemcpyDeviceToHost));
out:
    if (d_t) (void)hipFree(d_t);
    if (d_o) (void)hipFree(d_o);
    leave(&c);
    return rc;
}

/* test hook: the next host-pointer call fails (hipErrorUnknown) after its inputs were staged and its kernels launched,
 * so that the error path's clean-up (tests: the staging copies of secrets are wiped there too) can be exercised */
void eddsa_amd_debug_fail_next_host_call(void)
{
    pthread_rwlock_wrlock(&g_table);
    g_fail_next_host_call = 1;
    pthread_rwlock_unlock(&g_table);
}

/* diagnostic for the tests: how many half-length pairs the exact integer check (csrc/lanes.h: verify_half_scalars_lane)
 * has refused on the default device since its workspaces were allocated.  Waits for the device.  Expected: 0. */
int eddsa_amd_halve_rejected(uint64_t *count)
{
    struct call c;
    int rc = enter(&c, -1);
    if (rc) return rc;
    *count = 0;
    pthread_mutex_lock(&c.e->lk);
    TRY(hipDeviceSynchronize());
    for (int i = 0; i < VERIFY_SLOTS; i++) {
        uint32_t w = 0;
        if (!c.e->vs[i].ws.offcount) continue;
        TRY(hipMemcpy(&w, c.e->vs[i].ws.offcount + EDK_REFUSED_WORD, sizeof(w), hipMemcpyDeviceToHost));
        *count += w;
    }
out:
    pthread_mutex_unlock(&c.e->lk);
    leave(&c);
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
    pthread_rwlock_wrlock(&g_table);
    g_offcurve_mode = exact == 2 ? 2 : exact != 0;
    pthread_rwlock_unlock(&g_table);
}

/* ed25519_verify_batch_rlc[_dev] calls of fewer than `items` items use the per-item kernels (default 3 x 2^17:
 * the measured break-even; 0 = always try the combination) */
void eddsa_amd_set_rlc_min_items(size_t items)
{
    pthread_rwlock_wrlock(&g_table);
    g_rlc_min_items = items;
    pthread_rwlock_unlock(&g_table);
}

/* per-kernel timing of the verify pass, for bench.py's roofline line: HIP events recorded on the
 * launch stream around k_verify_prepare / k_verify_main / k_verify_finish of every chunk */
void eddsa_amd_set_profiling(int on)
{
    pthread_rwlock_wrlock(&g_table);       /* no call in flight: nobody is bumping marks_used */
    g_profiling = on != 0;
    for (int d = 0; d < MAX_DEVICES; d++) if (g_eng[d]) g_eng[d]->marks_used = 0;
    pthread_rwlock_unlock(&g_table);
}

/* average duration (ms) of each of the three kernels over the passes recorded on the default device
 * since profiling was switched on (at most MARK_SLOTS; later passes are not recorded) */
int eddsa_amd_verify_phase_ms(float out[3])
{
    struct call c;
    int rc = enter(&c, -1);
    if (rc) return rc;
    pthread_mutex_lock(&c.e->lk);
    const int used = c.e->marks_used < MARK_SLOTS ? c.e->marks_used : MARK_SLOTS;
    if (used == 0) { rc = -(int)hipErrorNotReady; goto out; }
    out[0] = out[1] = out[2] = 0.0f;
    for (int s = 0; s < used; s++) {
        TRY(hipEventSynchronize(c.e->marks[s][3]));
        for (int i = 0; i < 3; i++) {
            float ms = 0.0f;
            TRY(hipEventElapsedTime(&ms, c.e->marks[s][i], c.e->marks[s][i + 1]));
            out[i] += ms / (float)used;
        }
    }
out:
    pthread_mutex_unlock(&c.e->lk);
    leave(&c);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * device-pointer work on one engine (the engine's device is current)
 * ---------------------------------------------------------------------------------------- */

/* a pass failed half-way: kernels already queued may still use the slot; wait for them before the
 * slot can be handed out (and possibly re-allocated) again */
static void slot_quiesce(struct vslot *v, hipStream_t st)
{
    (void)hipStreamSynchronize(st);
    if (v->ws.side) (void)hipStreamSynchronize(v->ws.side);
    (void)hipEventRecord(v->free, st);
}

/* both verify forms: chunks of at most CHUNK_MAX items through the workspace */
static int verify_on(struct engine *e, uint8_t *ok, const edk_verify_src *all, size_t n, hipStream_t st)
{
    int rc = 0;
    if (n == 0) return 0;
    pthread_mutex_lock(&e->lk);
    struct vslot *v = ws_pick(e, st);
    rc = ws_reserve(v, n < CHUNK_MAX ? n : CHUNK_MAX);
    if (rc) goto unlock;
    v->ws.exact_offcurve = g_offcurve_mode;
    v->ws.algo = g_offcurve_mode ? g_verify_algo : 1;   /* the reject mode has no exact path for the items the pair search gives up on */
    /* the slot may have served another stream: order this pass behind its previous one */
    TRY(hipStreamWaitEvent(st, v->free, 0));
    for (size_t done = 0; done < n; done += CHUNK_MAX) {
        size_t m = n - done < CHUNK_MAX ? n - done : CHUNK_MAX;
        edk_verify_src src = *all;
        src.sigs += done * all->sig_stride;
        src.pubs += done * all->pub_stride;
        if (all->msg_off) src.msg_off += done; else src.msgs += done * all->msg_stride;
        hipEvent_t *marks = NULL;
        if (g_profiling && e->marks_used < MARK_SLOTS) marks = e->marks[e->marks_used++];
        TRY(edk_verify(ok + done, &src, m, e->base16, &v->ws, marks, st));
    }
    TRY(hipEventRecord(v->free, st));
out:
    if (rc) slot_quiesce(v, st);
unlock:
    pthread_mutex_unlock(&e->lk);
    return rc;
}

/* the fixed-base operations and x25519 share one driver: chunks of at most CHUNK_MAX items through fws */
typedef hipError_t (*fixed_step)(struct engine *e, size_t done, size_t m, const void *ctx, const edk_fixed_ws *fws, hipStream_t st);

static int fixed_on(struct engine *e, size_t n, fixed_step step, const void *ctx, hipStream_t st)
{
    int rc = 0;
    if (n == 0) return 0;
    pthread_mutex_lock(&e->lk);
    struct vslot *v = ws_pick(e, st);
    rc = fws_reserve(v, n < CHUNK_MAX ? n : CHUNK_MAX, st);
    if (rc) goto unlock;
    TRY(hipStreamWaitEvent(st, v->free, 0));
    for (size_t done = 0; done < n; done += CHUNK_MAX)
        TRY(step(e, done, n - done < CHUNK_MAX ? n - done : CHUNK_MAX, ctx, &v->fws, st));
    TRY(hipEventRecord(v->free, st));
out:
    if (rc) slot_quiesce(v, st);
unlock:
    pthread_mutex_unlock(&e->lk);
    return rc;
}

struct sign_ctx { uint8_t *sigs; const uint8_t *secs, *pubs, *msgs; const uint64_t *msg_off; size_t msg_len; };

static hipError_t sign_step(struct engine *e, size_t done, size_t m, const void *vctx, const edk_fixed_ws *fws, hipStream_t st)
{
    const struct sign_ctx *c = (const struct sign_ctx *)vctx;
    const uint8_t *mp = c->msg_off ? c->msgs : c->msgs + done * c->msg_len;
    const uint64_t *op = c->msg_off ? c->msg_off + done : NULL;
    return edk_sign(c->sigs + 64 * done, c->secs + 32 * done, c->pubs + 32 * done, mp, op, c->msg_len, m,
                    e->comb_img, fws, st);
}

struct io_ctx { uint8_t *out; const uint8_t *in; };
struct io2_ctx { uint8_t *out; const uint8_t *a, *b; };

static hipError_t genpub_step(struct engine *e, size_t done, size_t m, const void *vctx, const edk_fixed_ws *fws, hipStream_t st)
{
    const struct io_ctx *c = (const struct io_ctx *)vctx;
    return edk_genpub(c->out + 32 * done, c->in + 32 * done, m, e->comb_img, fws, st);
}

static hipError_t x25519_step(struct engine *e, size_t done, size_t m, const void *vctx, const edk_fixed_ws *fws, hipStream_t st)
{
    const struct io2_ctx *c = (const struct io2_ctx *)vctx;
    (void)e;
    return edk_x25519(c->out + 32 * done, c->a + 32 * done, c->b + 32 * done, m, fws, st);
}

static hipError_t xbase_step(struct engine *e, size_t done, size_t m, const void *vctx, const edk_fixed_ws *fws, hipStream_t st)
{
    const struct io_ctx *c = (const struct io_ctx *)vctx;
    return edk_x25519_base(c->out + 32 * done, c->in + 32 * done, m, e->comb_img, fws, st);
}

static int sign_on(struct engine *e, uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                   const uint64_t *msg_off, size_t msg_len, size_t n, hipStream_t st)
{
    struct sign_ctx c = { sigs, secs, pubs, msgs, msg_off, msg_len };
    return fixed_on(e, n, sign_step, &c, st);
}

static int pk_to_x_on(struct engine *e, uint8_t *out, const uint8_t *in, size_t n, hipStream_t st)
{
    (void)e;
    hipError_t er = edk_pk_to_x(out, in, n, st);
    return er == hipSuccess ? 0 : -(int)er;
}

static int sk_to_x_on(struct engine *e, uint8_t *out, const uint8_t *in, size_t n, hipStream_t st)
{
    (void)e;
    hipError_t er = edk_sk_to_x(out, in, n, st);
    return er == hipSuccess ? 0 : -(int)er;
}

/* Batch verification by random linear combination (reference lib/ed25519-sha512.c:13-14, its TODO;
 * SURVEY 8(f)-3): see include/eddsa_amd.h.  One pass of at most CHUNK_MAX items; larger batches are
 * split into independent sub-batches. */
static int rlc_on(struct engine *e, uint8_t *ok, uint32_t *stats, const edk_verify_src *all, size_t n, hipStream_t st)
{
    int rc = 0;
    if (n == 0) return 0;
    if (n < g_rlc_min_items) {
        /* the combination has about 2 ms of latency of its own (hash tree, one serial Horner per group): below
         * ~3 x 2^17 items the per-item kernels are faster (tools/rlc_sizes.py), so such calls go straight to them */
        rc = verify_on(e, ok, all, n, st);
        if (!rc) { hipError_t er = edk_rlc_note_per_item(stats, n, st); if (er != hipSuccess) rc = -(int)er; }
        return rc;
    }
    /* The combination's group verdicts are read by the host, once per pass.  The wait for the stream happens OUTSIDE
     * e->lk (other threads keep enqueueing on this engine meanwhile); the workspace slot stays reserved through its
     * busy mark, which ws_pick honours. */
    pthread_mutex_lock(&e->lk);
    struct vslot *v = ws_pick(e, st);
    v->busy = 1;
    rc = ws_reserve(v, n < CHUNK_MAX ? n : CHUNK_MAX);
    if (!rc) rc = rws_reserve(v, n < CHUNK_MAX ? n : CHUNK_MAX);
    if (rc) goto release;
    v->ws.exact_offcurve = g_offcurve_mode ? g_offcurve_mode : 1;
    v->ws.algo = g_verify_algo;
    TRY(hipStreamWaitEvent(st, v->free, 0));
    for (size_t done = 0; done < n; done += CHUNK_MAX) {
        size_t m = n - done < CHUNK_MAX ? n - done : CHUNK_MAX;
        edk_verify_src src = *all;
        src.sigs += done * all->sig_stride;
        src.pubs += done * all->pub_stride;
        if (all->msg_off) src.msg_off += done; else src.msgs += done * all->msg_stride;
        TRY(edk_verify_rlc(ok + done, stats, &src, m, e->base16, &v->ws, &v->rws, st));
        pthread_mutex_unlock(&e->lk);
        hipError_t er = hipStreamSynchronize(st);
        pthread_mutex_lock(&e->lk);
        TRY(er);
        TRY(edk_verify_rlc_fallback(ok + done, &src, m, e->base16, &v->ws, &v->rws, st));
    }
    TRY(hipEventRecord(v->free, st));
out:
    if (rc) slot_quiesce(v, st);
release:
    v->busy = 0;
    pthread_cond_broadcast(&e->slot_cv);
    pthread_mutex_unlock(&e->lk);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * device-pointer entry points: run on the device that holds the output buffer
 * ---------------------------------------------------------------------------------------- */

#define DEV_ENTER(outptr) \
    struct call c; int dev_ = -1, rc; \
    if (n == 0) return 0; \
    rc = device_of(outptr, &dev_); if (rc) return rc; \
    rc = enter(&c, dev_); if (rc) return rc

int ed25519_verify_batch_dev(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs, const uint8_t *msgs,
                             const uint64_t *msg_off, size_t msg_len, size_t n, void *stream)
{
    const edk_verify_src src = { sigs, pubs, msgs, msg_off, msg_len, 64, 32, msg_len };
    DEV_ENTER(ok);
    rc = verify_on(c.e, ok, &src, n, (hipStream_t)stream);
    leave(&c);
    return rc;
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
    DEV_ENTER(ok);
    rc = verify_on(c.e, ok, &src, n, (hipStream_t)stream);
    leave(&c);
    return rc;
}

int ed25519_verify_batch_rlc_dev(uint8_t *ok, uint32_t *stats, const uint8_t *sigs, const uint8_t *pubs,
                                 const uint8_t *msgs, const uint64_t *msg_off, size_t msg_len, size_t n, void *stream)
{
    const edk_verify_src src = { sigs, pubs, msgs, msg_off, msg_len, 64, 32, msg_len };
    DEV_ENTER(ok);
    rc = rlc_on(c.e, ok, stats, &src, n, (hipStream_t)stream);
    leave(&c);
    return rc;
}

int ed25519_sign_batch_dev(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                           const uint64_t *msg_off, size_t msg_len, size_t n, void *stream)
{
    DEV_ENTER(sigs);
    rc = sign_on(c.e, sigs, secs, pubs, msgs, msg_off, msg_len, n, (hipStream_t)stream);
    leave(&c);
    return rc;
}

int ed25519_genpub_batch_dev(uint8_t *pubs, const uint8_t *secs, size_t n, void *stream)
{
    struct io_ctx x = { pubs, secs };
    DEV_ENTER(pubs);
    rc = fixed_on(c.e, n, genpub_step, &x, (hipStream_t)stream);
    leave(&c);
    return rc;
}

int x25519_batch_dev(uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n, void *stream)
{
    struct io2_ctx x = { out, scalars, points };
    DEV_ENTER(out);
    rc = fixed_on(c.e, n, x25519_step, &x, (hipStream_t)stream);
    leave(&c);
    return rc;
}

int x25519_base_batch_dev(uint8_t *out, const uint8_t *scalars, size_t n, void *stream)
{
    struct io_ctx x = { out, scalars };
    DEV_ENTER(out);
    rc = fixed_on(c.e, n, xbase_step, &x, (hipStream_t)stream);
    leave(&c);
    return rc;
}

int pk_ed25519_to_x25519_batch_dev(uint8_t *out, const uint8_t *in, size_t n, void *stream)
{
    DEV_ENTER(out);
    rc = pk_to_x_on(c.e, out, in, n, (hipStream_t)stream);
    leave(&c);
    return rc;
}

int sk_ed25519_to_x25519_batch_dev(uint8_t *out, const uint8_t *in, size_t n, void *stream)
{
    DEV_ENTER(out);
    rc = sk_to_x_on(c.e, out, in, n, (hipStream_t)stream);
    leave(&c);
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
 * Jobs that carry secrets (secret keys, scalars, shared secrets) zero their staging buffers before
 * the call returns: nothing secret outlives the call in HBM.
 * ---------------------------------------------------------------------------------------- */

#define PIPE_CHUNK ((size_t)1 << 18)   /* 1024 blocks of 256 lanes: one full residency of the chip */
/* verify: two residencies per chunk, so that the exact path's chain for off-curve keys (4 ms beside
 * the main kernel, tools/verify_sizes.py) stays hidden behind k_verify_main as it is in one big pass */
#define PIPE_CHUNK_VERIFY ((size_t)1 << 19)

enum { WIPE_NONE = 0, WIPE_IN0 = 1, WIPE_OUT = 2 };   /* which staging buffers held secrets */

struct hjob {
    int n_in; const uint8_t *in[PIPE_MAX_IN]; size_t in_w[PIPE_MAX_IN];   /* fixed-width inputs */
    int has_msgs; const uint8_t *msgs; const uint64_t *msg_off; size_t msg_len;
    uint8_t *out; size_t out_w;
    int (*run)(struct engine *e, const struct hjob *j, uint8_t *d_out, uint8_t *const d_in[PIPE_MAX_IN],
               const uint8_t *d_msgs, const uint64_t *d_off, size_t msg_len, size_t m, hipStream_t st);
    size_t rec_sig, rec_pub, rec_msg;          /* records: offsets inside in[0]'s items (in_w[0] = stride) */
    size_t chunk;                              /* items per pipeline stage (0: PIPE_CHUNK) */
    int wipe;                                  /* WIPE_* */
    uint32_t *stats;                           /* rlc: host copy of the pass statistics (4 words) or NULL */
};

static int pipe_grow(void **buf, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    if (*buf) { wipe_free(*buf, *cap); *buf = NULL; *cap = 0; }
    hipError_t e = hipMalloc(buf, need < 256 ? 256 : need);
    if (e != hipSuccess) return -(int)e;
    *cap = need < 256 ? 256 : need;
    return 0;
}

static int pipe_init(struct pipe *p)
{
    int rc = 0;
    if (p->ready) return 0;
    TRY(hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&p->exec, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking));
    for (int s = 0; s < 2; s++) {
        TRY(hipEventCreateWithFlags(&p->in_ready[s], hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&p->exec_done[s], hipEventDisableTiming));
        TRY(hipEventCreateWithFlags(&p->slot_free[s], hipEventDisableTiming));
    }
    p->ready = 1;
out:
    return rc;
}

static void pipe_release(struct pipe *p)
{
    for (int s = 0; s < 2; s++) {
        for (int i = 0; i < PIPE_MAX_IN; i++) wipe_free(p->d_in[s][i], p->in_cap[s][i]);
        wipe_free(p->d_msgs[s], 0);
        if (p->in_ready[s]) (void)hipEventDestroy(p->in_ready[s]);
        if (p->exec_done[s]) (void)hipEventDestroy(p->exec_done[s]);
        if (p->slot_free[s]) (void)hipEventDestroy(p->slot_free[s]);
    }
    wipe_free(p->d_off, 0);
    wipe_free(p->d_out, p->out_cap);
    if (p->up) (void)hipStreamDestroy(p->up);
    if (p->exec) (void)hipStreamDestroy(p->exec);
    if (p->down) (void)hipStreamDestroy(p->down);
    memset(p, 0, sizeof(*p));
}

/* one host-pointer job on engine e (its device is current) */
static int pipe_run_on(struct engine *e, const struct hjob *j, size_t n)
{
    int rc = 0;
    struct pipe *p = &e->pipe;
    uint32_t *d_stats = NULL;
    if (n == 0) return 0;
    pthread_mutex_lock(&e->pipe_lk);
    rc = pipe_init(p);
    if (rc) goto out;
    {
        const int ragged = j->has_msgs && j->msg_off != NULL;
        const size_t stage = j->chunk ? j->chunk : PIPE_CHUNK;
        const size_t chunk = ragged ? n : (n < stage ? n : stage);
        const size_t nchunks = (n + chunk - 1) / chunk;
        const int nslots = nchunks > 1 ? 2 : 1;
        if ((rc = pipe_grow(&p->d_out, &p->out_cap, n * j->out_w + (j->stats ? 64 : 0)))) goto out;
        if (j->stats) {
            d_stats = (uint32_t *)((uint8_t *)p->d_out + (n * j->out_w + 15) / 16 * 16);
            TRY(hipMemsetAsync(d_stats, 0, 16, p->exec));
        }
        if (ragged && (rc = pipe_grow(&p->d_off, &p->off_cap, (n + 1) * sizeof(uint64_t)))) goto out;
        for (int s = 0; s < nslots; s++) {
            for (int i = 0; i < j->n_in; i++)
                if ((rc = pipe_grow(&p->d_in[s][i], &p->in_cap[s][i], chunk * j->in_w[i]))) goto out;
            if (j->has_msgs) {
                const size_t need = ragged ? (size_t)j->msg_off[n] : chunk * j->msg_len;
                if ((rc = pipe_grow(&p->d_msgs[s], &p->msgs_cap[s], need))) goto out;
            }
        }
        /* a call of one chunk has nothing to overlap: everything in order on the kernels' stream, one sync */
        const int single = nchunks == 1;
        hipStream_t up = single ? p->exec : p->up, down = single ? p->exec : p->down;
        if (ragged) TRY(hipMemcpyAsync(p->d_off, j->msg_off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up));
        for (size_t k = 0; k < nchunks; k++) {
            const int s = (int)(k & 1);
            const size_t lo = k * chunk, m = n - lo < chunk ? n - lo : chunk;
            /* upload chunk k into slot s once the kernels of chunk k-2 have consumed it */
            if (k >= 2) TRY(hipStreamWaitEvent(up, p->exec_done[s], 0));
            for (int i = 0; i < j->n_in; i++)
                TRY(hipMemcpyAsync(p->d_in[s][i], j->in[i] + lo * j->in_w[i], m * j->in_w[i], hipMemcpyHostToDevice, up));
            if (j->has_msgs) {
                const size_t bytes = ragged ? (size_t)j->msg_off[n] : m * j->msg_len;
                const uint8_t *src = ragged ? j->msgs : j->msgs + lo * j->msg_len;
                if (bytes) TRY(hipMemcpyAsync(p->d_msgs[s], src, bytes, hipMemcpyHostToDevice, up));
            }
            if (!single) TRY(hipEventRecord(p->in_ready[s], up));
            /* kernels of chunk k */
            if (!single) TRY(hipStreamWaitEvent(p->exec, p->in_ready[s], 0));
            {
                struct hjob jj = *j;
                jj.stats = d_stats;
                rc = j->run(e, &jj, (uint8_t *)p->d_out + lo * j->out_w, (uint8_t *const *)p->d_in[s],
                            (const uint8_t *)p->d_msgs[s], ragged ? (const uint64_t *)p->d_off : NULL,
                            j->msg_len, m, p->exec);
            }
            if (!rc && g_fail_next_host_call) { g_fail_next_host_call = 0; rc = -(int)hipErrorUnknown; }
            if (rc) goto out;
            if (!single) TRY(hipEventRecord(p->exec_done[s], p->exec));
            /* download chunk k-1 (its kernels were launched one iteration ago) */
            if (k >= 1) {
                const size_t plo = (k - 1) * chunk;
                TRY(hipStreamWaitEvent(down, p->exec_done[s ^ 1], 0));
                TRY(hipMemcpyAsync(j->out + plo * j->out_w, (uint8_t *)p->d_out + plo * j->out_w, chunk * j->out_w,
                                   hipMemcpyDeviceToHost, down));
            }
        }
        {
            const size_t plo = (nchunks - 1) * chunk;
            if (!single) TRY(hipStreamWaitEvent(down, p->exec_done[(nchunks - 1) & 1], 0));
            TRY(hipMemcpyAsync(j->out + plo * j->out_w, (uint8_t *)p->d_out + plo * j->out_w, (n - plo) * j->out_w,
                               hipMemcpyDeviceToHost, down));
            if (j->stats) TRY(hipMemcpyAsync(j->stats, d_stats, 16, hipMemcpyDeviceToHost, down));
        }
        /* secrets do not outlive the call in the staging buffers */
        if (j->wipe & WIPE_IN0)
            for (int s = 0; s < nslots; s++) TRY(hipMemsetAsync(p->d_in[s][0], 0, chunk * j->in_w[0], p->exec));
        if (!single) TRY(hipStreamSynchronize(down));
        if (j->wipe & WIPE_OUT) TRY(hipMemsetAsync(p->d_out, 0, n * j->out_w, p->exec));
        TRY(hipStreamSynchronize(p->exec));
        if (!single) TRY(hipStreamSynchronize(up));
    }
out:
    if (rc && p->ready) {
        (void)hipStreamSynchronize(p->up); (void)hipStreamSynchronize(p->exec); (void)hipStreamSynchronize(p->down);
        /* a failed call must not leave its secrets behind either (best effort: whatever was staged, whole buffers) */
        if (j->wipe & WIPE_IN0)
            for (int s = 0; s < 2; s++) if (p->d_in[s][0]) (void)hipMemsetAsync(p->d_in[s][0], 0, p->in_cap[s][0], p->exec);
        if ((j->wipe & WIPE_OUT) && p->d_out) (void)hipMemsetAsync(p->d_out, 0, p->out_cap, p->exec);
        if (j->wipe) (void)hipStreamSynchronize(p->exec);
    }
    pthread_mutex_unlock(&e->pipe_lk);
    return rc;
}

/* on the default device */
static int pipe_run(const struct hjob *j, size_t n)
{
    struct call c;
    int rc;
    if (n == 0) return 0;
    rc = enter(&c, -1);
    if (rc) return rc;
    rc = pipe_run_on(c.e, j, n);
    leave(&c);
    return rc;
}

#define RUN_ARGS struct engine *e, const struct hjob *j, uint8_t *d_out, uint8_t *const d_in[PIPE_MAX_IN], \
                 const uint8_t *d_msgs, const uint64_t *d_off, size_t msg_len, size_t m, hipStream_t st
static int run_verify(RUN_ARGS)
{
    (void)j;
    const edk_verify_src src = { d_in[0], d_in[1], d_msgs, d_off, msg_len, 64, 32, msg_len };
    return verify_on(e, d_out, &src, m, st);
}
static int run_verify_rlc(RUN_ARGS)
{
    const edk_verify_src src = { d_in[0], d_in[1], d_msgs, d_off, msg_len, 64, 32, msg_len };
    return rlc_on(e, d_out, j->stats, &src, m, st);
}
static int run_verify_records(RUN_ARGS)
{
    (void)d_msgs; (void)d_off;
    const edk_verify_src src = { d_in[0] + j->rec_sig, d_in[0] + j->rec_pub, d_in[0] + j->rec_msg, NULL, msg_len,
                                 j->in_w[0], j->in_w[0], j->in_w[0] };
    return verify_on(e, d_out, &src, m, st);
}
static int run_sign(RUN_ARGS)
{
    (void)j;
    return sign_on(e, d_out, d_in[0], d_in[1], d_msgs, d_off, msg_len, m, st);
}
static int run_x25519(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    struct io2_ctx x = { d_out, d_in[0], d_in[1] };
    return fixed_on(e, m, x25519_step, &x, st);
}
static int run_genpub(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    struct io_ctx x = { d_out, d_in[0] };
    return fixed_on(e, m, genpub_step, &x, st);
}
static int run_xbase(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    struct io_ctx x = { d_out, d_in[0] };
    return fixed_on(e, m, xbase_step, &x, st);
}
static int run_pk_to_x(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    return pk_to_x_on(e, d_out, d_in[0], m, st);
}
static int run_sk_to_x(RUN_ARGS)
{
    (void)j; (void)d_msgs; (void)d_off; (void)msg_len;
    return sk_to_x_on(e, d_out, d_in[0], m, st);
}

static struct hjob job_verify(uint8_t *ok, const uint8_t *sigs, const uint8_t *pubs, const uint8_t *msgs,
                              const uint64_t *msg_off, size_t msg_len)
{
    struct hjob j = { 2, { sigs, pubs, NULL }, { 64, 32, 0 }, 1, msgs, msg_off, msg_len, ok, 1, run_verify, 0, 0, 0,
                      PIPE_CHUNK_VERIFY, WIPE_NONE, NULL };
    return j;
}
static struct hjob job_sign(uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                            const uint64_t *msg_off, size_t msg_len)
{
    struct hjob j = { 2, { secs, pubs, NULL }, { 32, 32, 0 }, 1, msgs, msg_off, msg_len, sigs, 64, run_sign, 0, 0, 0, 0,
                      WIPE_IN0, NULL };
    return j;
}
static struct hjob job_x25519(uint8_t *out, const uint8_t *scalars, const uint8_t *points)
{
    struct hjob j = { 2, { scalars, points, NULL }, { 32, 32, 0 }, 0, NULL, NULL, 0, out, 32, run_x25519, 0, 0, 0, 0,
                      WIPE_IN0 | WIPE_OUT, NULL };
    return j;
}
static struct hjob job_1in(int (*run)(RUN_ARGS), uint8_t *out, const uint8_t *in, int wipe)
{
    struct hjob j = { 1, { in, NULL, NULL }, { 32, 0, 0 }, 0, NULL, NULL, 0, out, 32, run, 0, 0, 0, 0, wipe, NULL };
    return j;
}

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
    j.chunk = CHUNK_MAX;                /* one combination per 2^20 items */
    j.stats = local;
    int rc = pipe_run(&j, n);
    if (stats) memcpy(stats, local, sizeof(local));
    return rc;
}

int ed25519_verify_records(uint8_t *ok, const uint8_t *records, size_t stride, size_t sig_off, size_t pub_off,
                           size_t msg_off, size_t msg_len, size_t n)
{
    if (!records_ok(stride, sig_off, pub_off, msg_off, msg_len)) return -(int)hipErrorInvalidValue;
    struct hjob j = { 1, { records, NULL, NULL }, { stride, 0, 0 }, 0, NULL, NULL, msg_len, ok, 1, run_verify_records,
                      sig_off, pub_off, msg_off, PIPE_CHUNK_VERIFY, WIPE_NONE, NULL };
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
    struct hjob j = job_1in(run_genpub, pubs, secs, WIPE_IN0);
    return pipe_run(&j, n);
}
int x25519_base_batch(uint8_t *out, const uint8_t *scalars, size_t n)
{
    struct hjob j = job_1in(run_xbase, out, scalars, WIPE_IN0);
    return pipe_run(&j, n);
}
int pk_ed25519_to_x25519_batch(uint8_t *out, const uint8_t *in, size_t n)
{
    struct hjob j = job_1in(run_pk_to_x, out, in, WIPE_NONE);
    return pipe_run(&j, n);
}
int sk_ed25519_to_x25519_batch(uint8_t *out, const uint8_t *in, size_t n)
{
    struct hjob j = job_1in(run_sk_to_x, out, in, WIPE_IN0 | WIPE_OUT);
    return pipe_run(&j, n);
}

/* ------------------------------------------------------------------------------------------
 * several devices in one process: contiguous shards, one host thread per device for the
 * host-pointer forms, no data-path collective; the only exchange is the gather of the result bytes
 * (RCCL over xGMI) in the device-pointer form.
 * ---------------------------------------------------------------------------------------- */

void eddsa_amd_shard_bounds(size_t n, int rank, int world, size_t *lo, size_t *hi)
{
    const size_t base = n / (size_t)world, extra = n % (size_t)world, r = (size_t)rank;
    *lo = r * base + (r < extra ? r : extra);
    *hi = *lo + base + (r < extra ? 1 : 0);
}

static int rccl_load(void)
{
    if (!g_multi.rccl) g_multi.rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!g_multi.rccl) g_multi.rccl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!g_multi.rccl) return ERR_RCCL_MISSING;
#define SYM(field, name) do { *(void **)&g_multi.field = dlsym(g_multi.rccl, name); if (!g_multi.field) return ERR_RCCL_MISSING; } while (0)
    SYM(CommInitAll, "ncclCommInitAll"); SYM(CommDestroy, "ncclCommDestroy"); SYM(AllGather, "ncclAllGather");
    SYM(Broadcast, "ncclBroadcast"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return 0;
}

int eddsa_amd_init_devices(const int *devices, int n)
{
    int rc = 0, all[MAX_DEVICES];
    if (devices == NULL || n <= 0) {       /* every visible device */
        hipError_t er = hipGetDeviceCount(&n);
        if (er != hipSuccess) return -(int)er;
        if (n > MAX_DEVICES) n = MAX_DEVICES;
        for (int i = 0; i < n; i++) all[i] = i;
        devices = all;
    }
    if (n < 1 || n > MAX_DEVICES) return -(int)hipErrorInvalidValue;
    for (int i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= MAX_DEVICES) return -(int)hipErrorInvalidDevice;
        for (int k = 0; k < i; k++) if (devices[k] == devices[i]) return -(int)hipErrorInvalidValue;   /* one engine, one RCCL rank per device */
    }
    pthread_rwlock_wrlock(&g_table);
    multi_release();
    for (int i = 0; i < n && !rc; i++) rc = engine_create(devices[i]);
    if (!rc) rc = rccl_load();
    if (!rc) {
        int saved = -1, r;
        (void)hipGetDevice(&saved);
        r = g_multi.CommInitAll(g_multi.comm, n, devices);     /* single process, one communicator per device */
        if (saved >= 0) (void)hipSetDevice(saved);
        if (r) rc = ERR_RCCL_BASE - r;
    }
    if (!rc) {
        g_multi.n = n;
        memcpy(g_multi.dev, devices, (size_t)n * sizeof(int));
        if (g_default < 0) g_default = devices[0];
    } else {
        multi_release();
    }
    pthread_rwlock_unlock(&g_table);
    return rc;
}

int eddsa_amd_device_count(void)
{
    pthread_rwlock_rdlock(&g_table);
    const int n = g_multi.n;
    pthread_rwlock_unlock(&g_table);
    return n;
}

/* the HIP device that owns shard `index` of the set (the order given to eddsa_amd_init_devices), or -1 */
int eddsa_amd_device_at(int index)
{
    pthread_rwlock_rdlock(&g_table);
    const int dev = index >= 0 && index < g_multi.n ? g_multi.dev[index] : -1;
    pthread_rwlock_unlock(&g_table);
    return dev;
}

/* host-pointer forms: thread d runs the ordinary streaming pipeline of device d on shard d; results
 * are copied device -> caller's buffer slice directly, so there is nothing to gather */
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

/* Device-pointer form: device d of the set holds shard d of the inputs and a buffer ok_full[d] of
 * n_total bytes; shard d is verified on device d into its own slice of ok_full[d], then one RCCL
 * all-gather (grouped broadcasts when the shards differ in length) completes every ok_full[d].
 * Everything is enqueued from the calling thread; streams[d] orders the work on device d. */
int ed25519_verify_batch_multi_dev(uint8_t *const ok_full[], const uint8_t *const sigs[], const uint8_t *const pubs[],
                                   const uint8_t *const msgs[], size_t msg_len, size_t n_total, void *const streams[])
{
    int rc = 0, saved = -1, g, r, even = 1, locked = 0;
    pthread_rwlock_rdlock(&g_table);
    g = g_multi.n;
    if (g == 0) { rc = -(int)hipErrorNotInitialized; goto unlock; }
    if (n_total == 0) goto unlock;
    (void)hipGetDevice(&saved);
    /* concurrent callers must not interleave their launches or their grouped RCCL calls on the shared communicators */
    pthread_mutex_lock(&g_rccl_lk);
    locked = 1;
    for (int d = 0; d < g && !rc; d++) {
        size_t lo, hi;
        eddsa_amd_shard_bounds(n_total, d, g, &lo, &hi);
        if (hi - lo != n_total / (size_t)g) even = 0;
        const edk_verify_src src = { sigs[d], pubs[d], msgs[d], NULL, msg_len, 64, 32, msg_len };
        TRY(hipSetDevice(g_multi.dev[d]));
        rc = verify_on(g_eng[g_multi.dev[d]], ok_full[d] + lo, &src, hi - lo, (hipStream_t)streams[d]);
    }
    if (rc) goto out;
    /* the final result gather: the only exchange of the path */
    if ((r = g_multi.GroupStart())) { rc = ERR_RCCL_BASE - r; goto out; }
    for (int d = 0; d < g; d++) {
        if (even) {
            size_t lo, hi;
            eddsa_amd_shard_bounds(n_total, d, g, &lo, &hi);
            r = g_multi.AllGather(ok_full[d] + lo, ok_full[d], hi - lo, NCCL_UINT8, g_multi.comm[d], (hipStream_t)streams[d]);
            if (r && !rc) rc = ERR_RCCL_BASE - r;
        } else {
            for (int root = 0; root < g; root++) {
                size_t lo, hi;
                eddsa_amd_shard_bounds(n_total, root, g, &lo, &hi);
                r = g_multi.Broadcast(ok_full[d] + lo, ok_full[d] + lo, hi - lo, NCCL_UINT8, root, g_multi.comm[d],
                                      (hipStream_t)streams[d]);
                if (r && !rc) rc = ERR_RCCL_BASE - r;
            }
        }
    }
    r = g_multi.GroupEnd();
    if (r && !rc) rc = ERR_RCCL_BASE - r;
out:
    if (locked) pthread_mutex_unlock(&g_rccl_lk);
    if (saved >= 0) (void)hipSetDevice(saved);
unlock:
    pthread_rwlock_unlock(&g_table);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * diagnostics
 * ---------------------------------------------------------------------------------------- */

static int count_nonzero(const void *dev, size_t bytes, uint64_t *count)
{
    int rc = 0;
    if (!dev || !bytes) return 0;
    uint8_t *h = (uint8_t *)malloc(bytes);
    if (!h) return -(int)hipErrorOutOfMemory;
    TRY(hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < bytes; i++) *count += h[i] != 0;
out:
    free(h);
    return rc;
}

/* Secret hygiene check (tests): non-zero bytes left on the default device in out[0] the scalar
 * workspace `aux` (sign's a and r), out[1] the point workspace `acc` (x25519's (x2 : z2); public for
 * the other operations), out[2] the host pipeline's first input staging buffers (secret keys /
 * scalars), out[3] its output staging buffer.  Waits for the device to go idle first. */
int eddsa_amd_secret_residue(uint64_t out[4])
{
    struct call c;
    int rc = enter(&c, -1);
    if (rc) return rc;
    out[0] = out[1] = out[2] = out[3] = 0;
    pthread_mutex_lock(&c.e->pipe_lk);
    pthread_mutex_lock(&c.e->lk);
    TRY(hipDeviceSynchronize());
    for (int i = 0; i < VERIFY_SLOTS && !rc; i++) {
        const struct vslot *v = &c.e->vs[i];
        rc = count_nonzero(v->fws.aux, v->fws.capacity * 16 * sizeof(uint32_t), &out[0]);
        if (!rc) rc = count_nonzero(v->fws.acc, v->fws.capacity * ACC_WORDS * sizeof(uint32_t), &out[1]);
    }
    for (int s = 0; s < 2 && !rc; s++) rc = count_nonzero(c.e->pipe.d_in[s][0], c.e->pipe.in_cap[s][0], &out[2]);
    if (!rc) rc = count_nonzero(c.e->pipe.d_out, c.e->pipe.out_cap, &out[3]);
out:
    pthread_mutex_unlock(&c.e->lk);
    pthread_mutex_unlock(&c.e->pipe_lk);
    leave(&c);
    return rc;
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
