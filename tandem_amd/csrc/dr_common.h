// dr_common.h -- shared host-side helpers of libdr_mi355x.so (error plumbing, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dr_mi355x.h"

namespace dr {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

std::string &last_error_slot();

[[noreturn]] inline void fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

#define DR_HIP(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      ::dr::fail(DR_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// Wraps a C-ABI body: exceptions -> status code + dr_last_error().
template <class F>
inline int guarded(F &&f) {
  try {
    last_error_slot().clear();
    f();
    return DR_OK;
  } catch (const Error &e) {
    last_error_slot() = e.what();
    return e.code;
  } catch (const std::exception &e) {
    last_error_slot() = e.what();
    return DR_ERR_DEVICE;
  }
}

template <class T>
inline T *dalloc(size_t n) {
  void *p = nullptr;
  DR_HIP(hipMalloc(&p, n * sizeof(T) > 0 ? n * sizeof(T) : 16));
  return reinterpret_cast<T *>(p);
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// A DR_* switch that selects a superseded kernel generation or the losing side of a settled A/B exists only in the PARITY build
// (-DDR_PARITY_HOOKS, libdr_mi355x_hooks.so: test infrastructure).  In the product library hook_env() is a constant: the variable is not read, the
// branch behind it folds away.  What the product library does read from the environment is listed in INTEGRATION.md ("Environment switches").
#ifdef DR_PARITY_HOOKS
inline const char *hook_env(const char *name) { return getenv(name); }
#else
inline const char *hook_env(const char *) { return nullptr; }
#endif

}  // namespace dr
