/*
 * fake_hip.c - TEST INFRASTRUCTURE: a stand-in for the HIP runtime on host memory, so that the product's host side
 * (libeddsa_amd/csrc/eddsa_amd.c + host_pipe.c, unchanged) runs on the CPU of the build container under
 * -fsanitize=thread and -fsanitize=address,undefined, with MORE THAN ONE "device" (the single-process
 * multi-device path had only ever met one device, and the flat combiner had no race evidence).
 *
 * Never part of the product and never loaded by it: tests/fake_hip/Makefile links these objects into test binaries
 * (tests/c/multi_device.c, tests/c/threaded_callers.c) INSTEAD of libamdhip64; the kernels' launchers are replaced by
 * fake_kernels.cpp, which calls the -DED_HOST_CHECK build of the device source (the same one tests/host_check/ uses).
 *
 * Model: FAKE_HIP_DEVICES devices (default 2); "device memory" is host memory tagged with the device that allocated
 * it; streams and events are tagged objects.  Every copy, memset, launcher and collective checks that the pointers it is
 * given live on the device it runs on, and that the stream belongs to the calling thread's current device.  A violation
 * prints a message and aborts.
 *
 * Two execution models:
 *   eager (default)       every operation completes inside the call that issues it: races between HOST threads, lifetime
 *                         errors and device / rank mix-ups are in reach, ordering between streams is not;
 *   FAKE_HIP_DEFER=1      asynchronous work (async copies and memsets, the launchers' work, event records and waits) is
 *                         QUEUED on its stream and runs as LATE as the API allows: only when the host synchronises with
 *                         that stream or event (hipStreamSynchronize, hipEventSynchronize, hipDeviceSynchronize, a blocking
 *                         copy on the legacy default stream, hipFree) or when a stream being run reaches a
 *                         hipStreamWaitEvent whose record is still queued elsewhere.  A stream that nobody waits for never
 *                         gets ahead, so a consumer that lacks its wait - a download queued without the producer's event, a
 *                         host read before the synchronisation, a staging buffer refilled or wiped before the copy out of it
 *                         has run - reads stale bytes, deterministically, and the caller's check of the result fails.
 *                         Waits that can never be satisfied (a cycle between streams) abort as the hang they would be.
 *                         Rules kept from the real runtime: the legacy default stream orders itself with BLOCKING streams
 *                         only; a copy between pageable host memory and the device does not return before the pageable
 *                         side has been read / written; hipFree and hipHostFree wait for the device.
 */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fake_hip.h"

#define MAX_ALLOCS 4096
struct alloc { uint8_t *p; size_t bytes; int device; int host; };   /* host: page-locked host memory (device = -1) */
static struct alloc g_allocs[MAX_ALLOCS];
static pthread_mutex_t g_lk = PTHREAD_MUTEX_INITIALIZER;
static __thread int t_device = 0;
static __thread hipError_t t_last = hipSuccess;
static long g_live_streams, g_live_events;

/* deferred model: a stream is a queue of tasks; an event remembers how often it was recorded, how many of those records
 * have run, and the stream that holds the latest one */
enum { T_RUN, T_RECORD, T_WAIT };
struct task { struct task *next; int kind; void (*fn)(void *); void *arg; struct ihipEvent_t *ev; long gen; struct ihipStream_t *from; };
struct ihipStream_t { unsigned magic; int device, blocking, running; struct task *head, *tail; struct ihipStream_t *next_live; };
struct ihipEvent_t { unsigned magic; int device; long recorded, done, asked; struct ihipStream_t *where; };
#define STREAM_MAGIC 0x57ea3u
#define EVENT_MAGIC 0xe7e47u

static void die(const char *what)
{
    fprintf(stderr, "fake_hip: %s (current device %d)\n", what, t_device);
    abort();
}

int fake_hip_device_count(void)
{
    const char *e = getenv("FAKE_HIP_DEVICES");
    int n = e ? atoi(e) : 2;
    return n < 1 ? 1 : n > 16 ? 16 : n;
}

static hipError_t fail(hipError_t e) { t_last = e; return e; }

/* Fault injection for tests/c/host_fault_walk.c: the call with ordinal g_fail_at among the FALLIBLE runtime calls
 * (allocations, copies, memsets, stream / event creation, records, waits, synchronisations, hipSetDevice - the ones a
 * real runtime fails when the device is lost or memory runs out; releases are not failed) returns an error instead of
 * doing its work.  One shot: the ordinal is consumed when it fires. */
static long g_calls, g_fail_at, g_fired;         /* atomics */
long fake_hip_calls(void) { return __atomic_load_n(&g_calls, __ATOMIC_SEQ_CST); }
void fake_hip_fail_call(long ordinal) { __atomic_store_n(&g_fail_at, ordinal, __ATOMIC_SEQ_CST); }
long fake_hip_faults_fired(void) { return __atomic_load_n(&g_fired, __ATOMIC_SEQ_CST); }
static int tick(void)
{
    const long c = __atomic_add_fetch(&g_calls, 1, __ATOMIC_SEQ_CST);
    long at = c;
    if (!__atomic_compare_exchange_n(&g_fail_at, &at, 0, 0, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return 0;
    __atomic_add_fetch(&g_fired, 1, __ATOMIC_SEQ_CST);
    return 1;
}
#define FALLIBLE(err) do { if (tick()) return fail(err); } while (0)

/* the allocation that holds [p, p + bytes), or NULL */
static struct alloc *find(const void *p, size_t bytes)
{
    const uint8_t *q = (const uint8_t *)p;
    for (int i = 0; i < MAX_ALLOCS; i++)
        if (g_allocs[i].p && q >= g_allocs[i].p && q + bytes <= g_allocs[i].p + g_allocs[i].bytes) return &g_allocs[i];
    return NULL;
}

/* device of the allocation that holds the range: >= 0 device memory, -1 page-locked host memory, -2 unknown (pageable) */
int fake_hip_owner(const void *p, size_t bytes)
{
    pthread_mutex_lock(&g_lk);
    const struct alloc *a = find(p, bytes ? bytes : 1);
    const int r = !a ? -2 : a->host ? -1 : a->device;
    pthread_mutex_unlock(&g_lk);
    return r;
}

void fake_hip_require_device(const void *p, size_t bytes, int device, const char *what)
{
    const int o = fake_hip_owner(p, bytes);
    if (o != device) {
        fprintf(stderr, "fake_hip: %s: %zu bytes at %p live on %s %d, expected device %d\n", what, bytes, p,
                o == -2 ? "no allocation" : o == -1 ? "the host, pinned" : "device", o, device);
        abort();
    }
}

int fake_hip_current_device(void) { return t_device; }
int fake_hip_stream_device(hipStream_t s) { return s ? (s->magic == STREAM_MAGIC ? s->device : (die("not a live stream"), -1)) : t_device; }

static void check_stream(hipStream_t s, const char *what)
{
    if (s && s->magic != STREAM_MAGIC) die("use of a destroyed or foreign stream");
    if (s && s->device != t_device) { fprintf(stderr, "fake_hip: %s on a stream of device %d", what, s->device); die(" while another device is current"); }
}

static hipError_t add(void **out, size_t bytes, int device, int host)
{
    uint8_t *p = NULL;
    if (posix_memalign((void **)&p, 256, bytes ? bytes : 1) != 0) return fail(hipErrorOutOfMemory);
    /* fresh device memory holds garbage (of a large buffer only both ends are filled: the fault walk allocates the
     * 63 MB scratchpads of the verify workspaces a few thousand times) */
    if (bytes <= ((size_t)4 << 20)) memset(p, 0xa5, bytes);
    else { memset(p, 0xa5, (size_t)1 << 20); memset(p + bytes - ((size_t)1 << 20), 0xa5, (size_t)1 << 20); }
    pthread_mutex_lock(&g_lk);
    int i = 0;
    while (i < MAX_ALLOCS && g_allocs[i].p) i++;
    if (i == MAX_ALLOCS) { pthread_mutex_unlock(&g_lk); free(p); return fail(hipErrorOutOfMemory); }
    g_allocs[i].p = p; g_allocs[i].bytes = bytes ? bytes : 1; g_allocs[i].device = device; g_allocs[i].host = host;
    pthread_mutex_unlock(&g_lk);
    *out = p;
    return hipSuccess;
}

static hipError_t drop(void *p, int host)
{
    if (!p) return hipSuccess;
    pthread_mutex_lock(&g_lk);
    struct alloc *a = find(p, 1);
    if (!a || a->p != p || a->host != host) { pthread_mutex_unlock(&g_lk); die(host ? "hipHostFree of something hipHostMalloc did not return" : "hipFree of something hipMalloc did not return"); }
    a->p = NULL;
    pthread_mutex_unlock(&g_lk);
    free(p);
    return hipSuccess;
}

long fake_hip_live_allocations(void)
{
    long n = 0;
    pthread_mutex_lock(&g_lk);
    for (int i = 0; i < MAX_ALLOCS; i++) n += g_allocs[i].p != NULL;
    n += g_live_streams + g_live_events;
    pthread_mutex_unlock(&g_lk);
    return n;
}

/* ---- devices ---- */
hipError_t hipGetDeviceCount(int *n) { *n = fake_hip_device_count(); return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d)
{
    if (d < 0 || d >= fake_hip_device_count()) return fail(hipErrorInvalidDevice);
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int d)
{
    if (d < 0 || d >= fake_hip_device_count()) return fail(hipErrorInvalidDevice);
    memset(prop, 0, sizeof(*prop));
    strcpy(prop->gcnArchName, "gfx950:fake");
    strcpy(prop->name, "fake MI355X");
    prop->multiProcessorCount = 256;
    return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipGetLastError(void) { const hipError_t e = t_last; t_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e)
{
    switch (e) {
        case hipSuccess: return "no error";
        case hipErrorInvalidValue: return "invalid argument";
        case hipErrorOutOfMemory: return "out of memory";
        case hipErrorInvalidDevice: return "invalid device ordinal";
        case hipErrorNotInitialized: return "initialization error";
        case hipErrorNotReady: return "device not ready";
        case hipErrorUnknown: return "unknown error";
        default: return "some HIP error (fake runtime)";
    }
}

/* ---- memory ---- */
hipError_t hipMalloc(void **p, size_t bytes) { FALLIBLE(hipErrorOutOfMemory); return add(p, bytes, t_device, 0); }
static pthread_mutex_t g_q;                               /* (defined with the queues, further down) */
static void drain_device(int device);
static void wait_for_all_devices(void)
{
    if (!fake_hip_deferred()) return;
    pthread_mutex_lock(&g_q);
    for (int d = 0; d < fake_hip_device_count(); d++) drain_device(d);
    pthread_mutex_unlock(&g_q);
}
hipError_t hipFree(void *p) { if (p) wait_for_all_devices(); return drop(p, 0); }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags) { (void)flags; FALLIBLE(hipErrorOutOfMemory); return add(p, bytes, -1, 1); }
hipError_t hipHostFree(void *p) { if (p) wait_for_all_devices(); return drop(p, 1); }

hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p)
{
    pthread_mutex_lock(&g_lk);
    const struct alloc *f = find(p, 1);
    memset(a, 0, sizeof(*a));
    if (!f) { pthread_mutex_unlock(&g_lk); return fail(hipErrorInvalidValue); }
    a->type = f->host ? hipMemoryTypeHost : hipMemoryTypeDevice;
    a->device = f->host ? 0 : f->device;
    a->hostPointer = f->host ? (void *)p : NULL;
    a->devicePointer = (void *)p;
    pthread_mutex_unlock(&g_lk);
    return hipSuccess;
}

/* ------------------------------------------------------------------------------------------------------------
 * the queues (FAKE_HIP_DEFER=1).  One lock, g_q, guards every queue, every event counter and the running of tasks:
 * "device" work runs on whichever host thread forces it, one task at a time.
 * ---------------------------------------------------------------------------------------------------------- */
static pthread_mutex_t g_q = PTHREAD_MUTEX_INITIALIZER;
static struct ihipStream_t *g_live;                        /* created streams */
static struct ihipStream_t g_null[16];                     /* the legacy default stream of each device (blocking by definition) */
static int g_defer = -1;

int fake_hip_deferred(void)
{
    if (g_defer < 0) { const char *e = getenv("FAKE_HIP_DEFER"); g_defer = e && atoi(e) != 0; }
    return g_defer;
}

static struct ihipStream_t *S(hipStream_t s)
{
    if (s) return s;
    struct ihipStream_t *n = &g_null[t_device];
    if (!n->magic) { n->magic = STREAM_MAGIC; n->device = t_device; n->blocking = 1; }
    return n;
}

static void put(struct ihipStream_t *s, struct task t)
{
    struct task *n = (struct task *)malloc(sizeof(*n));
    if (!n) die("out of memory for a task");
    *n = t;
    n->next = NULL;
    if (s->tail) s->tail->next = n; else s->head = n;
    s->tail = n;
}

/* deferred model: queued tasks that something has FORCED to run so far (a synchronisation, an event wait, a blocking copy) - a host
 * side that enqueues without ever waiting leaves it unchanged (tests/c/multi_passes.c) */
static long g_tasks_run;
long fake_hip_tasks_run(void) { pthread_mutex_lock(&g_q); long v = g_tasks_run; pthread_mutex_unlock(&g_q); return v; }

/* run the head of s until it is empty, or (ev != NULL) until record `gen` of ev has run.  g_q is held. */
static void drain(struct ihipStream_t *s, struct ihipEvent_t *ev, long gen)
{
    if (ev && ev->done >= gen) return;
    if (s->running) die("a stream waits for work that is queued behind that very wait: on the device this never completes");
    s->running = 1;
    while (s->head && !(ev && ev->done >= gen)) {
        struct task *t = s->head;
        if (t->kind == T_WAIT && t->ev && t->ev->done < t->gen) drain(t->from, t->ev, t->gen);    /* the record it waits for, and what precedes it there */
        s->head = t->next;
        if (!s->head) s->tail = NULL;
        if (t->kind == T_RUN) { g_tasks_run++; t->fn(t->arg); }
        else if (t->kind == T_RECORD && t->ev && t->ev->done < t->gen) t->ev->done = t->gen;
        free(t);
    }
    s->running = 0;
    if (ev && ev->done < gen) die("an event is waited for whose record is in no queue");
}

/* what the legacy default stream implies: everything queued on the device's blocking streams and on the default stream */
static void drain_blocking(int device)
{
    for (struct ihipStream_t *s = g_live; s; s = s->next_live) if (s->device == device && s->blocking) drain(s, NULL, 0);
    drain(&g_null[device], NULL, 0);
}
static void drain_device(int device)
{
    for (struct ihipStream_t *s = g_live; s; s = s->next_live) if (s->device == device) drain(s, NULL, 0);
    drain(&g_null[device], NULL, 0);
}

/* queue fn(arg) on the stream (eager model: run it now).  The launchers of fake_kernels.cpp come through here. */
void fake_hip_enqueue(hipStream_t stream, void (*fn)(void *), void *arg)
{
    if (!fake_hip_deferred()) { fn(arg); return; }
    pthread_mutex_lock(&g_q);
    struct ihipStream_t *s = S(stream);
    /* default stream <-> blocking streams order themselves */
    if (s == &g_null[s->device]) drain_blocking(s->device); else if (s->blocking) drain(&g_null[s->device], NULL, 0);
    struct task t = { NULL, T_RUN, fn, arg, NULL, 0, NULL };
    put(s, t);
    pthread_mutex_unlock(&g_q);
}

/* for the fake RCCL: one task per stream, all queued before any of them can run (a collective's parts) */
void fake_hip_enqueue_group(int n, hipStream_t const *streams, void (*fn)(void *), void *const *args)
{
    if (!fake_hip_deferred()) { for (int i = 0; i < n; i++) fn(args[i]); return; }
    pthread_mutex_lock(&g_q);
    for (int i = 0; i < n; i++) {
        struct task t = { NULL, T_RUN, fn, args[i], NULL, 0, NULL };
        put(streams[i], t);
    }
    pthread_mutex_unlock(&g_q);
}
/* for the fake RCCL: the stream a NULL handle means right now (the current device's default stream) ... */
hipStream_t fake_hip_resolve_stream(hipStream_t s)
{
    pthread_mutex_lock(&g_q);
    struct ihipStream_t *r = S(s);
    pthread_mutex_unlock(&g_q);
    return r;
}
/* ... and, CALLED FROM INSIDE A TASK ONLY (g_q is held): run that stream until *flag is set by one of its tasks */
void fake_hip_run_until(hipStream_t stream, const volatile int *flag)
{
    struct ihipStream_t *s = stream;
    if (*flag) return;
    if (s->running) die("a collective waits for a rank whose stream waits for this one: on the device this never completes");
    s->running = 1;
    while (s->head && !*flag) {
        struct task *t = s->head;
        if (t->kind == T_WAIT && t->ev && t->ev->done < t->gen) drain(t->from, t->ev, t->gen);
        s->head = t->next;
        if (!s->head) s->tail = NULL;
        if (t->kind == T_RUN) t->fn(t->arg);
        else if (t->kind == T_RECORD && t->ev && t->ev->done < t->gen) t->ev->done = t->gen;
        free(t);
    }
    s->running = 0;
    if (!*flag) die("a collective waits for a rank that never issued its part");
}

/* a copy's device side must be memory of the CURRENT device (what the real runtime would reach over xGMI by
 * accident is a bug here: every shard is meant to stay on its own device) */
static void check_copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (!bytes) return;
    if (kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice) fake_hip_require_device(dst, bytes, t_device, "copy destination");
    if (kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice) fake_hip_require_device(src, bytes, t_device, "copy source");
    if (kind == hipMemcpyHostToDevice && fake_hip_owner(src, bytes) >= 0) die("host-to-device copy from device memory");
    if (kind == hipMemcpyDeviceToHost && fake_hip_owner(dst, bytes) >= 0) die("device-to-host copy into device memory");
}

struct copy_task { void *dst; const void *src; void *snapshot; size_t bytes; };
static void run_copy(void *a)
{
    struct copy_task *c = (struct copy_task *)a;
    memmove(c->dst, c->src, c->bytes);
    free(c->snapshot);
    free(c);
}
struct set_task { void *dst; int v; size_t bytes; };
static void run_set(void *a)
{
    struct set_task *c = (struct set_task *)a;
    memset(c->dst, c->v, c->bytes);
    free(c);
}

hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    FALLIBLE(hipErrorUnknown);
    check_copy(dst, src, bytes, kind);
    if (fake_hip_deferred()) {                             /* a blocking copy on the legacy default stream */
        pthread_mutex_lock(&g_q);
        drain_blocking(t_device);
        if (bytes) memmove(dst, src, bytes);
        pthread_mutex_unlock(&g_q);
        return hipSuccess;
    }
    if (bytes) memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s)
{
    FALLIBLE(hipErrorUnknown);
    check_stream(s, "hipMemcpyAsync");
    check_copy(dst, src, bytes, kind);
    if (!bytes) return hipSuccess;
    if (!fake_hip_deferred()) { memmove(dst, src, bytes); return hipSuccess; }
    struct copy_task *c = (struct copy_task *)calloc(1, sizeof(*c));
    if (!c) return fail(hipErrorOutOfMemory);
    c->dst = dst; c->src = src; c->bytes = bytes;
    if (kind == hipMemcpyDeviceToHost && fake_hip_owner(dst, bytes) == -2) {
        /* into pageable memory: the call returns when the bytes are there */
        pthread_mutex_lock(&g_q);
        if (!s) drain_blocking(t_device); else { if (s->blocking) drain(&g_null[s->device], NULL, 0); drain(s, NULL, 0); }
        pthread_mutex_unlock(&g_q);
        run_copy(c);
        return hipSuccess;
    }
    if (kind == hipMemcpyHostToDevice && fake_hip_owner(src, bytes) == -2) {
        /* out of pageable memory: the source has been read when the call returns (the runtime stages it) */
        c->snapshot = malloc(bytes);
        if (!c->snapshot) { free(c); return fail(hipErrorOutOfMemory); }
        memcpy(c->snapshot, src, bytes);
        c->src = c->snapshot;
    }
    fake_hip_enqueue(s, run_copy, c);
    return hipSuccess;
}
static hipError_t set_on(void *dst, int v, size_t bytes, hipStream_t s)
{
    if (!bytes) return hipSuccess;
    fake_hip_require_device(dst, bytes, t_device, "hipMemset");
    if (!fake_hip_deferred()) { memset(dst, v, bytes); return hipSuccess; }
    struct set_task *c = (struct set_task *)calloc(1, sizeof(*c));
    if (!c) return fail(hipErrorOutOfMemory);
    c->dst = dst; c->v = v; c->bytes = bytes;
    fake_hip_enqueue(s, run_set, c);
    return hipSuccess;
}
/* hipMemset of device memory does not wait for the device: it is work of the default stream */
hipError_t hipMemset(void *dst, int v, size_t bytes) { FALLIBLE(hipErrorUnknown); return set_on(dst, v, bytes, NULL); }
hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t s)
{
    FALLIBLE(hipErrorUnknown);
    check_stream(s, "hipMemsetAsync");
    return set_on(dst, v, bytes, s);
}

/* ---- streams and events ---- */
static hipError_t new_stream(hipStream_t *s, int blocking)
{
    FALLIBLE(hipErrorOutOfMemory);
    *s = (hipStream_t)calloc(1, sizeof(**s));
    if (!*s) return fail(hipErrorOutOfMemory);
    (*s)->magic = STREAM_MAGIC; (*s)->device = t_device; (*s)->blocking = blocking;
    pthread_mutex_lock(&g_q); (*s)->next_live = g_live; g_live = *s; pthread_mutex_unlock(&g_q);
    pthread_mutex_lock(&g_lk); g_live_streams++; pthread_mutex_unlock(&g_lk);
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s) { return new_stream(s, 1); }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags) { return new_stream(s, !(flags & hipStreamNonBlocking)); }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int prio) { (void)prio; return new_stream(s, !(flags & hipStreamNonBlocking)); }
hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s || s->magic != STREAM_MAGIC) die("hipStreamDestroy of a dead stream");
    pthread_mutex_lock(&g_q);
    drain(s, NULL, 0);                                     /* (the runtime lets queued work finish) */
    for (struct ihipStream_t **q = &g_live; *q; q = &(*q)->next_live) if (*q == s) { *q = s->next_live; break; }
    pthread_mutex_unlock(&g_q);
    s->magic = 0;
    free(s);
    pthread_mutex_lock(&g_lk); g_live_streams--; pthread_mutex_unlock(&g_lk);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
    if (s && s->magic != STREAM_MAGIC) die("hipStreamSynchronize of a dead stream");
    FALLIBLE(hipErrorUnknown);
    if (fake_hip_deferred()) {
        pthread_mutex_lock(&g_q);
        if (!s) drain_blocking(t_device); else { if (s->blocking) drain(&g_null[s->device], NULL, 0); drain(s, NULL, 0); }
        pthread_mutex_unlock(&g_q);
    }
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void)
{
    FALLIBLE(hipErrorUnknown);
    if (fake_hip_deferred()) { pthread_mutex_lock(&g_q); drain_device(t_device); pthread_mutex_unlock(&g_q); }
    return hipSuccess;
}
static hipError_t new_event(hipEvent_t *e)
{
    FALLIBLE(hipErrorOutOfMemory);
    *e = (hipEvent_t)calloc(1, sizeof(**e));
    if (!*e) return fail(hipErrorOutOfMemory);
    (*e)->magic = EVENT_MAGIC; (*e)->device = t_device;
    pthread_mutex_lock(&g_lk); g_live_events++; pthread_mutex_unlock(&g_lk);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return new_event(e); }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags) { (void)flags; return new_event(e); }
static void finish_event(hipEvent_t e)                     /* g_q held: the latest record has run when this returns */
{
    if (e->done < e->recorded) drain(e->where, e, e->recorded);
}
hipError_t hipEventDestroy(hipEvent_t e)
{
    if (!e || e->magic != EVENT_MAGIC) die("hipEventDestroy of a dead event");
    pthread_mutex_lock(&g_q);
    finish_event(e);
    /* waits queued on it are satisfied now, and must not look at the freed object */
    for (struct ihipStream_t *s = g_live; s; s = s->next_live) for (struct task *t = s->head; t; t = t->next) if (t->ev == e) t->ev = NULL;
    for (int d = 0; d < 16; d++) for (struct task *t = g_null[d].head; t; t = t->next) if (t->ev == e) t->ev = NULL;
    pthread_mutex_unlock(&g_q);
    e->magic = 0;
    free(e);
    pthread_mutex_lock(&g_lk); g_live_events--; pthread_mutex_unlock(&g_lk);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    if (!e || e->magic != EVENT_MAGIC) die("hipEventRecord of a dead event");
    check_stream(s, "hipEventRecord");
    if (e->device != fake_hip_stream_device(s)) die("hipEventRecord: the event belongs to another device than the stream");
    FALLIBLE(hipErrorUnknown);
    pthread_mutex_lock(&g_q);
    struct ihipStream_t *q = S(s);
    const long gen = ++e->recorded;
    if (!fake_hip_deferred()) e->done = gen;
    else {
        if (!s) drain_blocking(q->device); else if (q->blocking) drain(&g_null[q->device], NULL, 0);
        e->where = q;
        struct task t = { NULL, T_RECORD, NULL, NULL, e, gen, NULL };
        put(q, t);
    }
    pthread_mutex_unlock(&g_q);
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    if (!e || e->magic != EVENT_MAGIC) die("hipEventSynchronize of a dead event");
    FALLIBLE(hipErrorUnknown);
    pthread_mutex_lock(&g_q);
    finish_event(e);
    pthread_mutex_unlock(&g_q);
    return hipSuccess;
}
/* the first question about a record that has not run is answered "not ready" (and nothing moves); the second one runs it:
 * a caller that polls makes progress, a caller that takes the first answer as a hint sees the unfavourable one */
hipError_t hipEventQuery(hipEvent_t e)
{
    if (!e || e->magic != EVENT_MAGIC) die("hipEventQuery of a dead event");
    hipError_t r = hipSuccess;
    pthread_mutex_lock(&g_q);
    if (e->done < e->recorded) {
        if (e->asked < e->recorded) { e->asked = e->recorded; r = hipErrorNotReady; }
        else finish_event(e);
    }
    pthread_mutex_unlock(&g_q);
    return r;                                              /* (hipErrorNotReady is an answer, not a sticky error) */
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    if (!a || !b || a->magic != EVENT_MAGIC || b->magic != EVENT_MAGIC) die("hipEventElapsedTime of a dead event");
    pthread_mutex_lock(&g_q);
    const int pending = a->done < a->recorded || b->done < b->recorded;
    pthread_mutex_unlock(&g_q);
    if (pending) return fail(hipErrorNotReady);            /* as the runtime: both records must have completed */
    *ms = 0.001f;
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags)
{
    (void)flags;
    if (!e || e->magic != EVENT_MAGIC) die("hipStreamWaitEvent on a dead event");
    if (s && s->magic != STREAM_MAGIC) die("hipStreamWaitEvent of a dead stream");
    FALLIBLE(hipErrorUnknown);
    pthread_mutex_lock(&g_q);
    if (fake_hip_deferred() && e->done < e->recorded) {    /* waits for the latest record at the time of THIS call */
        struct ihipStream_t *q = S(s);
        struct task t = { NULL, T_WAIT, NULL, NULL, e, e->recorded, e->where };
        put(q, t);
    }
    pthread_mutex_unlock(&g_q);
    return hipSuccess;
}
