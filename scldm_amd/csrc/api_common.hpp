// Shared host-side helpers of the C ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdio>

#include "../../include/scldm_hip.h"

int scldm_fail(int code, const char* fmt, ...);   // records the thread-local message returned by scldm_last_error()
#define fail scldm_fail
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(SCLDM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define LAUNCH_CHECK()                                                                             \
  do {                                                                                             \
    hipError_t e_ = hipGetLastError();                                                             \
    if (e_ != hipSuccess) return fail(SCLDM_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

static inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
