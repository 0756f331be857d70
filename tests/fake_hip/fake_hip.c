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
 * it; streams and events are tagged objects and every operation completes at once (so ordering bugs between streams are
 * out of reach here - races between HOST threads, lifetime errors and device/rank mix-ups are not: every copy, memset,
 * launcher and collective checks that the pointers it is given live on the device it runs on, and that the stream
 * belongs to the calling thread's current device).  A violation prints a message and aborts.
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

struct ihipStream_t { unsigned magic; int device; };
struct ihipEvent_t { unsigned magic; int device; };
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
hipError_t hipDeviceSynchronize(void) { FALLIBLE(hipErrorUnknown); return hipSuccess; }
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
hipError_t hipFree(void *p) { return drop(p, 0); }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags) { (void)flags; FALLIBLE(hipErrorOutOfMemory); return add(p, bytes, -1, 1); }
hipError_t hipHostFree(void *p) { return drop(p, 1); }

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
static hipError_t copy_now(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    check_copy(dst, src, bytes, kind);
    if (bytes) memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    FALLIBLE(hipErrorUnknown);
    check_copy(dst, src, bytes, kind);
    if (bytes) memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s)
{
    FALLIBLE(hipErrorUnknown);
    check_stream(s, "hipMemcpyAsync");
    return copy_now(dst, src, bytes, kind);
}
static hipError_t set_now(void *dst, int v, size_t bytes)
{
    if (bytes) { fake_hip_require_device(dst, bytes, t_device, "hipMemset"); memset(dst, v, bytes); }
    return hipSuccess;
}
hipError_t hipMemset(void *dst, int v, size_t bytes)
{
    FALLIBLE(hipErrorUnknown);
    if (bytes) { fake_hip_require_device(dst, bytes, t_device, "hipMemset"); memset(dst, v, bytes); }
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t s)
{
    FALLIBLE(hipErrorUnknown);
    check_stream(s, "hipMemsetAsync");
    return set_now(dst, v, bytes);
}

/* ---- streams and events: tagged objects; everything has completed by the time a call returns ---- */
static hipError_t new_stream(hipStream_t *s)
{
    FALLIBLE(hipErrorOutOfMemory);
    *s = (hipStream_t)malloc(sizeof(**s));
    if (!*s) return fail(hipErrorOutOfMemory);
    (*s)->magic = STREAM_MAGIC; (*s)->device = t_device;
    pthread_mutex_lock(&g_lk); g_live_streams++; pthread_mutex_unlock(&g_lk);
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s) { return new_stream(s); }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags) { (void)flags; return new_stream(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int prio) { (void)flags; (void)prio; return new_stream(s); }
hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s || s->magic != STREAM_MAGIC) die("hipStreamDestroy of a dead stream");
    s->magic = 0;
    free(s);
    pthread_mutex_lock(&g_lk); g_live_streams--; pthread_mutex_unlock(&g_lk);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { if (s && s->magic != STREAM_MAGIC) die("hipStreamSynchronize of a dead stream"); FALLIBLE(hipErrorUnknown); return hipSuccess; }
static hipError_t new_event(hipEvent_t *e)
{
    FALLIBLE(hipErrorOutOfMemory);
    *e = (hipEvent_t)malloc(sizeof(**e));
    if (!*e) return fail(hipErrorOutOfMemory);
    (*e)->magic = EVENT_MAGIC; (*e)->device = t_device;
    pthread_mutex_lock(&g_lk); g_live_events++; pthread_mutex_unlock(&g_lk);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return new_event(e); }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags) { (void)flags; return new_event(e); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    if (!e || e->magic != EVENT_MAGIC) die("hipEventDestroy of a dead event");
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
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { if (!e || e->magic != EVENT_MAGIC) die("hipEventSynchronize of a dead event"); FALLIBLE(hipErrorUnknown); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { if (!e || e->magic != EVENT_MAGIC) die("hipEventQuery of a dead event"); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    if (!a || !b || a->magic != EVENT_MAGIC || b->magic != EVENT_MAGIC) die("hipEventElapsedTime of a dead event");
    *ms = 0.001f;
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags)
{
    (void)flags;
    if (!e || e->magic != EVENT_MAGIC) die("hipStreamWaitEvent on a dead event");
    if (s && s->magic != STREAM_MAGIC) die("hipStreamWaitEvent of a dead stream");
    FALLIBLE(hipErrorUnknown);
    return hipSuccess;
}
