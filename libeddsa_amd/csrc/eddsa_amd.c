/*
 * eddsa_amd.c - engines and the device-pointer side of libeddsa_amd.so, plain C.
 *
 * One `struct engine` per HIP device (generated tables, a pool of workspaces, the host pipeline's lanes and the
 * combiner's queue, both driven by host_pipe.c), created on first use.  A device-pointer call runs on the device its
 * output buffer lives on; a host-pointer call (host_pipe.c) on the default device (eddsa_amd_init) or, for the
 * *_multi entry points, on every device of the set bound by eddsa_amd_init_devices.  Every call makes its device
 * current for its own duration and restores the caller's.
 *
 * Everything is computed by the HIP kernels in kernels.hip / rlc.hip; there is no CPU arithmetic on this side and no
 * fallback: without a usable gfx950 device every entry point fails (batch API: negative return; eddsa.h API, which has
 * no error channel: message on stderr + abort()).  Shared declarations and the lock order: engine.h.
 */
#include "engine.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

pthread_rwlock_t g_table = PTHREAD_RWLOCK_INITIALIZER;
struct engine *g_eng[MAX_DEVICES];
static int g_default = -1;            /* device of the host-pointer entry points; -1: the caller's current device at first use */
static int g_verify_algo = 0;         /* eddsa_amd_set_verify_algo: 0 by pass size (default), 1 full-length windows, 2 half-length scalars */
static int g_offcurve_mode = 1;       /* eddsa_amd_set_offcurve_mode: 0 reject, 1 exact (default), 2 all exact */
static int g_profiling;               /* record marks around the three verify kernels */
static size_t g_rlc_min_items = EDDSA_AMD_RLC_MIN_ITEMS_DEFAULT;   /* eddsa_amd_set_rlc_min_items: smaller calls go to the per-item kernels */

int g_hooks_armed;                     /* eddsa_amd_debug_init: read and written atomically */
struct multi g_multi;                  /* the device set of the *_multi entry points (eddsa_amd_init_devices) */
static pthread_mutex_t g_rccl_lk = PTHREAD_MUTEX_INITIALIZER;   /* one grouped RCCL call at a time (taken after g_table) */
#define NCCL_UINT8 1                   /* ncclUint8, rccl.h */

const char *eddsa_amd_strerror(int err)
{
    if (err == 0) return "success";
    if (err == ERR_NOT_GFX950) return "eddsa_amd: device is not gfx950 (MI355X); no code object for it";
    if (err == ERR_RCCL_MISSING) return "eddsa_amd: librccl.so.1 could not be loaded (needed for the multi-device result gather)";
    if (err == EDDSA_AMD_HOOKS_OFF) return "eddsa_amd: test hooks not armed (eddsa_amd_debug_init)";
    if (err == EDDSA_AMD_STALLED) return "eddsa_amd: a verify kernel gave up waiting for a hand-off between its waves; the pass's outputs are incomplete";
    if (err <= ERR_RCCL_BASE) {
        if (g_multi.GetErrorString) return g_multi.GetErrorString(ERR_RCCL_BASE - err);
        return "eddsa_amd: RCCL error";
    }
    return hipGetErrorString((hipError_t)(-err));
}

static struct { int count, first; char what[96]; } g_teardown;   /* hip_note: count and first are atomic; what is written by the first */

void hip_note(hipError_t e, const char *what)
{
    if (e == hipSuccess) return;
    if (__atomic_fetch_add(&g_teardown.count, 1, __ATOMIC_ACQ_REL) == 0) {
        __atomic_store_n(&g_teardown.first, (int)e, __ATOMIC_RELEASE);
        strncpy(g_teardown.what, what, sizeof(g_teardown.what) - 1);
    }
    (void)hipGetLastError();               /* noted: the sticky slot is cleared (the only unchecked HIP calls of the host side are this one and hip_forget_error's) */
}

void hip_forget_error(void)
{
    (void)hipGetLastError();               /* the caller has handled the answer it stands for */
}

#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
int eddsa_amd_debug_teardown_errors(int *first_hip_error)
{
    if (first_hip_error) *first_hip_error = __atomic_load_n(&g_teardown.first, __ATOMIC_ACQUIRE);
    return __atomic_load_n(&g_teardown.count, __ATOMIC_ACQUIRE);
}
#endif

/* ------------------------------------------------------------------------------------------
 * engines
 * ---------------------------------------------------------------------------------------- */

/* device buffers that held secrets (or may have) are zeroed before they go back to the allocator:
 * the reference wipes after its secret-key operations (lib/ed25519-sha512.c:77,136, lib/x25519.c:208,221) */
void wipe_free(void *p, size_t bytes)
{
    if (!p) return;
    if (bytes) { HIP_NOTE(hipMemset(p, 0, bytes)); HIP_NOTE(hipDeviceSynchronize()); }
    HIP_NOTE(hipFree(p));
}

static void ws_release(struct vslot *v)
{
    if (v->ws.digits) HIP_NOTE(hipFree(v->ws.digits));
    if (v->ws.table) HIP_NOTE(hipFree(v->ws.table));
    if (v->ws.acc) HIP_NOTE(hipFree(v->ws.acc));
    if (v->ws.flags) HIP_NOTE(hipFree(v->ws.flags));
    if (v->ws.hdigits) HIP_NOTE(hipFree(v->ws.hdigits));
    if (v->ws.rtable) HIP_NOTE(hipFree(v->ws.rtable));
    if (v->ws.offlist) HIP_NOTE(hipFree(v->ws.offlist));
    if (v->ws.onlist) HIP_NOTE(hipFree(v->ws.onlist));
    if (v->ws.perm) HIP_NOTE(hipFree(v->ws.perm));
    if (v->ws.lenbins) HIP_NOTE(hipFree(v->ws.lenbins));
    if (v->ws.offcount) HIP_NOTE(hipFree(v->ws.offcount));
    if (v->ws.exact_pad) HIP_NOTE(hipFree(v->ws.exact_pad));
    if (v->ws.sums) HIP_NOTE(hipFree(v->ws.sums));
    v->ws.capacity = 0;
    v->ws.digits = v->ws.table = v->ws.acc = v->ws.offlist = v->ws.offcount = v->ws.exact_pad = NULL;
    v->ws.hdigits = v->ws.rtable = v->ws.sums = v->ws.onlist = v->ws.perm = v->ws.lenbins = NULL;
    v->ws.flags = NULL;
}

static void fws_release(struct vslot *v)
{
    wipe_free(v->fws.acc, v->fws.capacity * ACC_WORDS * sizeof(uint32_t));
    wipe_free(v->fws.aux, v->fws.capacity * 16 * sizeof(uint32_t));
    if (v->fws.tiles) HIP_NOTE(hipFree(v->fws.tiles));
    if (v->fws.perm) HIP_NOTE(hipFree(v->fws.perm));
    if (v->fws.lenbins) HIP_NOTE(hipFree(v->fws.lenbins));
    memset(&v->fws, 0, sizeof(v->fws));
}

static void rws_release(struct vslot *v)
{
    if (v->rws.base) HIP_NOTE(hipFree(v->rws.base));
    if (v->rws.host_gok) HIP_NOTE(hipHostFree(v->rws.host_gok));
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
    TRY(hipMalloc((void **)&v->fws.perm, cap * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->fws.lenbins, 2 * EDK_LEN_BINS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->fws.tiles, 256));
    TRY(hipMemsetAsync(v->fws.tiles, 0, 256, st));
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
    TRY(hipMalloc((void **)&v->ws.onlist, cap * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.perm, cap * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.lenbins, 2 * EDK_LEN_BINS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&v->ws.offcount, 256));
    TRY(hipMemset(v->ws.offcount, 0, 256));
    TRY(hipStreamSynchronize(NULL));      /* the pass's stream does not wait for the null stream */
    TRY(hipMalloc((void **)&v->ws.exact_pad, EDK_EXACT_PAD_BYTES));
    TRY(hipMalloc((void **)&v->ws.sums, EDK_SUMS_BYTES));
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
    TRY(hipMemset((char *)v->rws.base + edk_rlc_hook_offset(cap), 0, 256));
    TRY(hipStreamSynchronize(NULL));      /* the pass's stream does not wait for the null stream */
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

/* everything an engine holds on its device.  Caller holds g_table for writing (no call is in
 * flight) and has made e->device current. */
static void engine_destroy(struct engine *e)
{
    HIP_NOTE(hipDeviceSynchronize());
    pipe_release(&e->pipe);
    combiner_release(&e->comb_q);
    for (int i = 0; i < VERIFY_SLOTS; i++) {
        struct vslot *v = &e->vs[i];
        ws_release(v);
        fws_release(v);
        rws_release(v);
        if (v->ws.side) HIP_NOTE(hipStreamDestroy(v->ws.side));
        if (v->ws.ev_prepared) HIP_NOTE(hipEventDestroy(v->ws.ev_prepared));
        if (v->ws.ev_exact) HIP_NOTE(hipEventDestroy(v->ws.ev_exact));
        if (v->free) HIP_NOTE(hipEventDestroy(v->free));
    }
    if (e->base16) HIP_NOTE(hipFree(e->base16));
    if (e->comb) HIP_NOTE(hipFree(e->comb));
    if (e->comb_img) HIP_NOTE(hipFree(e->comb_img));
    if (e->status) HIP_NOTE(hipHostFree(e->status));
    for (int s = 0; s < MARK_SLOTS; s++)
        for (int i = 0; i < 4; i++) if (e->marks[s][i]) HIP_NOTE(hipEventDestroy(e->marks[s][i]));
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
    combiner_init(&e->comb_q);
    pipe_setup(&e->pipe);
    HIP_NOTE(hipGetDevice(&saved));
    TRY(hipSetDevice(device));
    TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { rc = ERR_NOT_GFX950; goto out; }
    TRY(hipMalloc((void **)&e->base16, (size_t)2 * TABLE_BASE16_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&e->comb, TABLE_COMB_ENTRIES * TABLE_ENTRY_WORDS * sizeof(uint32_t)));
    TRY(hipMalloc((void **)&e->comb_img, COMB_IMG_WORDS * sizeof(uint32_t)));
    TRY(hipHostMalloc((void **)&e->status, 64, hipHostMallocDefault));
    memset(e->status, 0, 64);
    for (int i = 0; i < VERIFY_SLOTS; i++) {
        /* highest queue priority for the side streams: their few workgroups must be dispatched while
         * k_verify_main still has thousands waiting, not after them */
        int lo = 0, hi = 0;
        struct vslot *v = &e->vs[i];
        v->ws.status = e->status;
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
    if (saved >= 0) HIP_NOTE(hipSetDevice(saved));
    return rc;
}

/* Every call brackets its work with enter()/leave(): enter() resolves the engine (creating it on
 * first use), holds g_table for reading and makes the engine's device current for the calling
 * thread; leave() restores the caller's device.  device < 0: the default device. */
int enter(struct call *c, int device)
{
    c->e = NULL;
    c->saved = -1;
    /* (an engine built here can be torn down again by a concurrent eddsa_amd_shutdown before this thread holds the table
     * for reading: then it is built once more; only a storm of shutdowns could use up the attempts) */
    for (int attempt = 0; attempt < 16; attempt++) {
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
            HIP_NOTE(hipGetDevice(&c->saved));
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

void leave(struct call *c)
{
    if (c->saved >= 0) HIP_NOTE(hipSetDevice(c->saved));
    pthread_rwlock_unlock(&g_table);
}

/* the device a device-pointer call runs on: where its output buffer lives */
static int device_of(const void *p, int *device)
{
    hipPointerAttribute_t a;
    hipError_t er = hipPointerGetAttributes(&a, p);
    if (er != hipSuccess) { hip_forget_error(); return -(int)hipErrorInvalidValue; }
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
    HIP_NOTE(hipGetDevice(&saved));
    multi_release();
    for (int d = 0; d < MAX_DEVICES; d++) {
        if (!g_eng[d]) continue;
        HIP_NOTE(hipSetDevice(d));
        engine_destroy(g_eng[d]);
        g_eng[d] = NULL;
    }
    g_default = -1;
    __atomic_store_n(&g_hooks_armed, 0, __ATOMIC_RELEASE);
    edk_debug_counting(0);
    (void)edk_debug_fail_in(0);
    if (saved >= 0) HIP_NOTE(hipSetDevice(saved));   /* teardown: nothing to report to */
    pthread_rwlock_unlock(&g_table);
    host_pool_stop();
}

/* ------------------------------------------------------------------------------------------
 * the test surface (include/eddsa_amd_debug.h): inert unless armed
 * ---------------------------------------------------------------------------------------- */

#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
int eddsa_amd_debug_init(int device, unsigned flags)
{
    const int rc = eddsa_amd_init(device);
    if (rc) return rc;
    __atomic_store_n(&g_hooks_armed, (flags & EDDSA_AMD_TEST_HOOKS) != 0, __ATOMIC_RELEASE);
    edk_debug_counting((flags & EDDSA_AMD_TEST_HOOKS) != 0);
    if (!(flags & EDDSA_AMD_TEST_HOOKS)) (void)edk_debug_fail_in(0);
    return 0;
}
#endif

#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
int eddsa_amd_debug_fail_hip_call(int nth)
{
    if (!__atomic_load_n(&g_hooks_armed, __ATOMIC_ACQUIRE)) return EDDSA_AMD_HOOKS_OFF;
    (void)edk_debug_fail_in(nth < 0 ? 0 : nth);
    return 0;
}
#endif

#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
int eddsa_amd_debug_hip_calls(void)
{
    return edk_debug_fail_in(-1);
}
#endif

#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
int eddsa_amd_debug_withhold_handoff(int tile_plus_1)
{
    struct call c;
    int rc, gave_up = 0;
    if (!__atomic_load_n(&g_hooks_armed, __ATOMIC_ACQUIRE)) return EDDSA_AMD_HOOKS_OFF;
    if ((rc = enter(&c, -1))) return rc;
    pthread_mutex_lock(&c.e->lk);
    TRY(hipDeviceSynchronize());
    for (int i = 0; i < VERIFY_SLOTS; i++) {
        const uint32_t w = tile_plus_1 > 0 ? (uint32_t)tile_plus_1 : 0u;
        if (c.e->vs[i].ws.offcount) TRY(hipMemcpy(c.e->vs[i].ws.offcount + EDK_WITHHOLD_WORD, &w, sizeof(w), hipMemcpyHostToDevice));
        if (c.e->vs[i].rws.base) {
            uint32_t *hook = (uint32_t *)((char *)c.e->vs[i].rws.base + edk_rlc_hook_offset(c.e->vs[i].rws.capacity));
            uint32_t two[2] = { 0, 0 };
            TRY(hipMemcpy(two, hook, sizeof(two), hipMemcpyDeviceToHost));
            gave_up += (int)two[1];
            two[0] = w; two[1] = 0;
            TRY(hipMemcpy(hook, two, sizeof(two), hipMemcpyHostToDevice));
        }
    }
    rc = gave_up;
out:
    pthread_mutex_unlock(&c.e->lk);
    leave(&c);
    return rc;
}
#endif

#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
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
#endif

/* Which evaluation ed25519_verify* uses (same verdicts; a measurement and test aid).  0 (default): half-length
 * scalars (csrc/halve.h), four lanes per item up to 2^15 items and one above; 1: full-length windows always;
 * 2: half-length scalars with one lane per item always. */
#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
void eddsa_amd_set_verify_algo(int algo)
{
    pthread_rwlock_wrlock(&g_table);
    g_verify_algo = algo >= 1 && algo <= 3 ? algo : 0;
    pthread_rwlock_unlock(&g_table);
}
#endif

/* diagnostic for the tests: how many half-length pairs the exact integer check (csrc/lanes.h: verify_half_scalars_lane)
 * has refused on the default device since its workspaces were allocated.  Waits for the device.  Expected: 0. */
#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
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
#endif

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

/* ed25519_verify_batch_rlc[_dev] calls of fewer than `items` items use the per-item kernels (default EDDSA_AMD_RLC_MIN_ITEMS_DEFAULT:
 * above the measured break-even; 0 = always try the combination) */
void eddsa_amd_set_rlc_min_items(size_t items)
{
    pthread_rwlock_wrlock(&g_table);
    g_rlc_min_items = items;
    pthread_rwlock_unlock(&g_table);
}

/* per-kernel timing of the verify pass, for bench.py's roofline line: HIP events recorded on the
 * launch stream around k_verify_prepare / k_verify_main / k_verify_finish of every chunk */
#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
void eddsa_amd_set_profiling(int on)
{
    pthread_rwlock_wrlock(&g_table);       /* no call in flight: nobody is bumping marks_used */
    g_profiling = on != 0;
    for (int d = 0; d < MAX_DEVICES; d++) if (g_eng[d]) g_eng[d]->marks_used = 0;
    pthread_rwlock_unlock(&g_table);
}
#endif

/* average duration (ms) of each of the three kernels over the passes recorded on the default device
 * since profiling was switched on (at most MARK_SLOTS; later passes are not recorded) */
#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
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
#endif

/* ------------------------------------------------------------------------------------------
 * device-pointer work on one engine (the engine's device is current)
 * ---------------------------------------------------------------------------------------- */

/* a pass failed half-way: kernels already queued may still use the slot; wait for them before the
 * slot can be handed out (and possibly re-allocated) again */
static void slot_quiesce(struct vslot *v, hipStream_t st)
{
    HIP_NOTE(hipStreamSynchronize(st));
    if (v->ws.side) HIP_NOTE(hipStreamSynchronize(v->ws.side));
    HIP_NOTE(hipEventRecord(v->free, st));
}

int take_async_error(struct engine *e)
{
    return __atomic_exchange_n(e->status, 0u, __ATOMIC_ACQ_REL) == EDK_STATUS_STALLED ? EDDSA_AMD_STALLED : 0;
}

/* both verify forms: chunks of at most CHUNK_MAX items through the workspace */
int verify_on(struct engine *e, uint8_t *ok, const edk_verify_src *all, size_t n, hipStream_t st, hipEvent_t bulk_done, int bulk_early)
{
    int rc = 0;
    if (n == 0) return 0;
    if ((rc = take_async_error(e))) return rc;   /* an earlier pass's kernels gave up: see engine.h */
    pthread_mutex_lock(&e->lk);
    struct vslot *v = ws_pick(e, st);
    rc = ws_reserve(v, n < CHUNK_MAX ? n : CHUNK_MAX);
    if (rc) goto unlock;
    v->ws.exact_offcurve = g_offcurve_mode;
    v->ws.algo = g_offcurve_mode ? g_verify_algo : 1;   /* the reject mode has no exact path for the items the pair search gives up on */
    /* the slot may have served another stream: order this pass behind its previous one */
    TRY(hipStreamWaitEvent(st, v->free, 0));
    edk_verify_src whole = *all;
    if (whole.msg_off && !whole.msg_end) whole.msg_end = whole.msg_off + n;   /* the kernels clamp every span into [0, msg_off[n]) */
    for (size_t done = 0; done < n; done += CHUNK_MAX) {
        size_t m = n - done < CHUNK_MAX ? n - done : CHUNK_MAX;
        edk_verify_src src = whole;
        src.sigs += done * all->sig_stride;
        src.pubs += done * all->pub_stride;
        if (all->msg_off) src.msg_off += done; else src.msgs += done * all->msg_stride;
        hipEvent_t *marks = NULL;
        if (g_profiling && e->marks_used < MARK_SLOTS) marks = e->marks[e->marks_used++];
        TRY(edk_verify(ok + done, &src, m, e->base16, &v->ws, marks, done + CHUNK_MAX >= n ? bulk_done : NULL, bulk_early, st));
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

struct sign_ctx { uint8_t *sigs; const uint8_t *secs, *pubs, *msgs; const uint64_t *msg_off, *msg_end; size_t msg_len; };

static hipError_t sign_step(struct engine *e, size_t done, size_t m, const void *vctx, const edk_fixed_ws *fws, hipStream_t st)
{
    const struct sign_ctx *c = (const struct sign_ctx *)vctx;
    const uint8_t *mp = c->msg_off ? c->msgs : c->msgs + done * c->msg_len;
    const uint64_t *op = c->msg_off ? c->msg_off + done : NULL;
    return edk_sign(c->sigs + 64 * done, c->secs + 32 * done, c->pubs + 32 * done, mp, op, c->msg_end, c->msg_len, m,
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

int sign_on(struct engine *e, uint8_t *sigs, const uint8_t *secs, const uint8_t *pubs, const uint8_t *msgs,
                   const uint64_t *msg_off, size_t msg_len, size_t n, hipStream_t st)
{
    struct sign_ctx c = { sigs, secs, pubs, msgs, msg_off, msg_off ? msg_off + n : NULL, msg_len };   /* spans are clamped into [0, msg_off[n]) */
    return fixed_on(e, n, sign_step, &c, st);
}

int genpub_on(struct engine *e, uint8_t *pubs, const uint8_t *secs, size_t n, hipStream_t st)
{
    struct io_ctx x = { pubs, secs };
    return fixed_on(e, n, genpub_step, &x, st);
}

int x25519_on(struct engine *e, uint8_t *out, const uint8_t *scalars, const uint8_t *points, size_t n, hipStream_t st)
{
    struct io2_ctx x = { out, scalars, points };
    return fixed_on(e, n, x25519_step, &x, st);
}

int xbase_on(struct engine *e, uint8_t *out, const uint8_t *scalars, size_t n, hipStream_t st)
{
    struct io_ctx x = { out, scalars };
    return fixed_on(e, n, xbase_step, &x, st);
}

int pk_to_x_on(struct engine *e, uint8_t *out, const uint8_t *in, size_t n, hipStream_t st)
{
    (void)e;
    hipError_t er = edk_pk_to_x(out, in, n, st);
    return er == hipSuccess ? 0 : -(int)er;
}

int sk_to_x_on(struct engine *e, uint8_t *out, const uint8_t *in, size_t n, hipStream_t st)
{
    (void)e;
    hipError_t er = edk_sk_to_x(out, in, n, st);
    return er == hipSuccess ? 0 : -(int)er;
}

/* Batch verification by random linear combination (reference lib/ed25519-sha512.c:13-14, its TODO;
 * SURVEY 8(f)-3): see include/eddsa_amd.h.  One pass of at most CHUNK_MAX items; larger batches are
 * split into independent sub-batches. */
int rlc_on(struct engine *e, uint8_t *ok, uint32_t *stats, const edk_verify_src *all, size_t n, hipStream_t st)
{
    int rc = 0;
    if (n == 0) return 0;
    if (n < g_rlc_min_items) {
        /* the combination has about 1 ms of latency of its own (hash tree, one serial Horner per group): below
         * ~2^17 items the per-item kernels are faster (tools/rlc_sizes.py), so such calls go straight to them */
        rc = verify_on(e, ok, all, n, st, NULL, 0);
        if (!rc) { hipError_t er = edk_rlc_note_per_item(stats, n, st); if (er != hipSuccess) rc = -(int)er; }
        return rc;
    }
    if ((rc = take_async_error(e))) return rc;
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
    edk_verify_src whole = *all;
    if (whole.msg_off && !whole.msg_end) whole.msg_end = whole.msg_off + n;
    for (size_t done = 0; done < n; done += CHUNK_MAX) {
        size_t m = n - done < CHUNK_MAX ? n - done : CHUNK_MAX;
        edk_verify_src src = whole;
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
    const edk_verify_src src = { sigs, pubs, msgs, msg_off, msg_len, 64, 32, msg_len, NULL };
    DEV_ENTER(ok);
    rc = verify_on(c.e, ok, &src, n, (hipStream_t)stream, NULL, 0);
    leave(&c);
    return rc;
}

/* fixed-size records: see include/eddsa_amd.h */
int records_ok(size_t stride, size_t sig_off, size_t pub_off, size_t msg_off, size_t msg_len)
{
    return sig_off <= stride && 64 <= stride - sig_off && pub_off <= stride && 32 <= stride - pub_off &&
           msg_off <= stride && msg_len <= stride - msg_off;
}

int ed25519_verify_records_dev(uint8_t *ok, const uint8_t *records, size_t stride, size_t sig_off, size_t pub_off,
                               size_t msg_off, size_t msg_len, size_t n, void *stream)
{
    if (!records_ok(stride, sig_off, pub_off, msg_off, msg_len)) return -(int)hipErrorInvalidValue;
    const edk_verify_src src = { records + sig_off, records + pub_off, records + msg_off, NULL, msg_len,
                                 stride, stride, stride, NULL };
    DEV_ENTER(ok);
    rc = verify_on(c.e, ok, &src, n, (hipStream_t)stream, NULL, 0);
    leave(&c);
    return rc;
}

int ed25519_verify_batch_rlc_dev(uint8_t *ok, uint32_t *stats, const uint8_t *sigs, const uint8_t *pubs,
                                 const uint8_t *msgs, const uint64_t *msg_off, size_t msg_len, size_t n, void *stream)
{
    const edk_verify_src src = { sigs, pubs, msgs, msg_off, msg_len, 64, 32, msg_len, NULL };
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
        HIP_NOTE(hipGetDevice(&saved));
        r = g_multi.CommInitAll(g_multi.comm, n, devices);     /* single process, one communicator per device */
        if (saved >= 0) HIP_NOTE(hipSetDevice(saved));
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

/* Device-pointer form: device d of the set holds shard d of the inputs and a buffer ok_full[d] of
 * n_total bytes; shard d is verified on device d into its own slice of ok_full[d], then one RCCL
 * all-gather (grouped broadcasts when the shards differ in length) completes every ok_full[d].
 * Everything is enqueued from the calling thread; streams[d] orders the work on device d. */
int ed25519_verify_batch_multi_dev(uint8_t *const ok_full[], const uint8_t *const sigs[], const uint8_t *const pubs[],
                                   const uint8_t *const msgs[], size_t msg_len, size_t n_total, void *const streams[])
{
    int rc = 0, saved = -1, g, r, even = 1, locked = 0;
    void *const no_streams[MAX_DEVICES] = { 0 };
    if (!streams) streams = no_streams;          /* no array: every device's default stream */
    pthread_rwlock_rdlock(&g_table);
    g = g_multi.n;
    if (g == 0) { rc = -(int)hipErrorNotInitialized; goto unlock; }
    if (n_total == 0) goto unlock;
    HIP_NOTE(hipGetDevice(&saved));
    /* concurrent callers must not interleave their launches or their grouped RCCL calls on the shared communicators */
    pthread_mutex_lock(&g_rccl_lk);
    locked = 1;
    for (int d = 0; d < g && !rc; d++) {
        size_t lo, hi;
        eddsa_amd_shard_bounds(n_total, d, g, &lo, &hi);
        if (hi - lo != n_total / (size_t)g) even = 0;
        const edk_verify_src src = { sigs[d], pubs[d], msgs[d], NULL, msg_len, 64, 32, msg_len, NULL };
        TRY(hipSetDevice(g_multi.dev[d]));
        rc = verify_on(g_eng[g_multi.dev[d]], ok_full[d] + lo, &src, hi - lo, (hipStream_t)streams[d], NULL, 0);
    }
    if (rc) goto out;
    /* the final result gather: the only exchange of the path */
    if ((r = g_multi.GroupStart())) { rc = ERR_RCCL_BASE - r; goto out; }
    for (int d = 0; d < g; d++) {
        /* a NULL stream is the default stream of the CURRENT device: rank d's calls are made with its device current.
         * (A failure here is reported, but the rank still makes its calls: a group that lacks a rank never completes.) */
        hipError_t er = hipSetDevice(g_multi.dev[d]);
        if (er != hipSuccess && !rc) rc = -(int)er;
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
    if (saved >= 0) HIP_NOTE(hipSetDevice(saved));
unlock:
    pthread_rwlock_unlock(&g_table);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * diagnostics
 * ---------------------------------------------------------------------------------------- */

int count_nonzero_dev(const void *dev, size_t bytes, uint64_t *count)
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
#ifdef EDDSA_AMD_DEBUG_BUILD   /* include/eddsa_amd_debug.h: only libeddsa_amd_debug.so has it */
int eddsa_amd_secret_residue(uint64_t out[4])
{
    struct call c;
    int rc = enter(&c, -1);
    if (rc) return rc;
    out[0] = out[1] = out[2] = out[3] = 0;
    pthread_mutex_lock(&c.e->lk);
    TRY(hipDeviceSynchronize());
    for (int i = 0; i < VERIFY_SLOTS && !rc; i++) {
        const struct vslot *v = &c.e->vs[i];
        rc = count_nonzero_dev(v->fws.aux, v->fws.capacity * 16 * sizeof(uint32_t), &out[0]);
        if (!rc) rc = count_nonzero_dev(v->fws.acc, v->fws.capacity * ACC_WORDS * sizeof(uint32_t), &out[1]);
    }
out:
    pthread_mutex_unlock(&c.e->lk);
    if (!rc) rc = pipe_residue(c.e, &out[2], &out[3]);
    leave(&c);
    return rc;
}
#endif
