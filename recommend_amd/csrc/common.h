// Shared helpers for the librecommend_amd kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/recommend_amd.h"

namespace rk {

// thread-local last-error text returned by rk_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

#define RK_CHECK_ARG(cond, ...)                      \
  do {                                               \
    if (!(cond)) return rk::fail(RK_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define RK_CHECK_HIP(expr)                                                    \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess)                                                     \
      return rk::fail(RK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// launch check: a bad configuration shows up in hipGetLastError right away
#define RK_CHECK_LAUNCH(name)                                                  \
  do {                                                                         \
    hipError_t _e = hipGetLastError();                                         \
    if (_e != hipSuccess)                                                      \
      return rk::fail(RK_ERR_HIP, "launch of %s failed: %s", name,             \
                      hipGetErrorString(_e));                                  \
  } while (0)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;

// v_permlane32_swap: lanes 32..63 of `lo` and lanes 0..31 of `hi` change places.  Through the
// compiler's builtin, not inline assembly: the instruction reads both operands as VALU sources, and
// the wait states it needs behind an MFMA or a vector instruction that has just written them are
// placed by the compiler's hazard recogniser, which does not look inside inline assembly.
__device__ __forceinline__ void swap_halves32(float &lo, float &hi) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
  lo = __uint_as_float(r[0]);
  hi = __uint_as_float(r[1]);
}

}  // namespace rk
