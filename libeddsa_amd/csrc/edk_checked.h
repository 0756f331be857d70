// edk_checked.h - the HIP calls of a verify pass that carry correctness (event records and stream waits that order the
// side stream against the caller's, the work-list reset, every launch) go through these macros: the first failure ends the
// pass and is returned (eddsa_amd.c: verify_on then waits for whatever was queued before the workspace is handed out again).
// edk_fault_tick() counts the checked calls and lets a test make the nth one fail (include/eddsa_amd_debug.h:
// eddsa_amd_debug_fail_hip_call; inert unless armed).  Host side of kernels.hip / rlc.hip only.
#pragma once
#include <hip/hip_runtime.h>

extern "C" int edk_fault_tick(void);   // kernels.hip: 1 when the armed fault falls on this call

#define EDK_DO(expr) do { if (edk_fault_tick()) return hipErrorUnknown; \
    const hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)
#define EDK_LAUNCH(...) do { if (edk_fault_tick()) return hipErrorUnknown; \
    hipLaunchKernelGGL(__VA_ARGS__); const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)
