/* fake_hip.h - what the other pieces of the CPU test build (fake_kernels.cpp, fake_rccl.c) ask the fake runtime.  Test infrastructure. */
#ifndef FAKE_HIP_H
#define FAKE_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
int fake_hip_device_count(void);
int fake_hip_current_device(void);
int fake_hip_owner(const void *p, size_t bytes);          /* >= 0: device memory of that device; -1: page-locked host memory; -2: unknown */
void fake_hip_require_device(const void *p, size_t bytes, int device, const char *what);   /* aborts with a message otherwise */
int fake_hip_stream_device(struct ihipStream_t *s);       /* NULL: the current device */
long fake_hip_live_allocations(void);                     /* allocations, streams and events not yet released */
int fake_hip_deferred(void);                              /* FAKE_HIP_DEFER=1: asynchronous work is queued and runs as late as the API allows (fake_hip.c) */
void fake_hip_enqueue(struct ihipStream_t *s, void (*fn)(void *), void *arg);   /* queue fn(arg) on the stream (eager model: run it now) */
void fake_hip_enqueue_group(int n, struct ihipStream_t *const *streams, void (*fn)(void *), void *const *args);   /* one task per (resolved) stream, queued together */
struct ihipStream_t *fake_hip_resolve_stream(struct ihipStream_t *s);           /* NULL -> the current device's default stream */
void fake_hip_run_until(struct ihipStream_t *s, const volatile int *flag);      /* inside a task only: run s until one of its tasks sets *flag */
long fake_hip_tasks_run(void);                            /* deferred model: queued tasks that a host-side wait has forced to run so far */
long fake_hip_calls(void);                                /* fallible runtime calls made so far (all threads) */
void fake_hip_fail_call(long ordinal);                    /* the call with that ordinal fails instead of doing its work (one shot; 0: disarm) */
long fake_hip_faults_fired(void);                         /* how many armed ordinals were reached */
#ifdef __cplusplus
}
#endif
#endif
