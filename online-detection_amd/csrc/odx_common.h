// Shared host-side helpers of libodx (gfx950 only; no other backend exists).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/odx.h"

namespace odx {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(odx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

}  // namespace odx

#define ODX_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      odx::set_error(__VA_ARGS__);             \
      return ODX_ERR_INVALID;                  \
    }                                          \
  } while (0)

#define ODX_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      odx::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return ODX_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)

#define ODX_CHECK_HIP(expr)                                                      \
  do {                                                                           \
    hipError_t e__ = (expr);                                                     \
    if (e__ != hipSuccess) {                                                     \
      odx::set_error("%s failed: %s", #expr, hipGetErrorString(e__));            \
      return ODX_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)

#define ODX_PROPAGATE(expr)        \
  do {                             \
    int rc__ = (expr);             \
    if (rc__ != ODX_OK) return rc__; \
  } while (0)
