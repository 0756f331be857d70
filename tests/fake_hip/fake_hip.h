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
long fake_hip_calls(void);                                /* fallible runtime calls made so far (all threads) */
void fake_hip_fail_call(long ordinal);                    /* the call with that ordinal fails instead of doing its work (one shot; 0: disarm) */
long fake_hip_faults_fired(void);                         /* how many armed ordinals were reached */
#ifdef __cplusplus
}
#endif
#endif
