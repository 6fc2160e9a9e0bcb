// rsx_common.h -- shared host-side plumbing of librsx.so (error reporting, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <stdexcept>
#include <string>

#include "rsx.h"
#include "rsx_diag.h"

namespace rsx {

std::string &last_error();

inline int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

// Experiment / tuning knobs read from the environment exist only in -DRSX_EXPERIMENTS builds (make EXPERIMENTS=1, what
// tools/prof*.sh and tools/spectral/* build): the product library has no hidden switches.
inline const char *exp_env(const char *name) {
#ifdef RSX_EXPERIMENTS
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Exception firewall of the C-ABI ("no C++ exceptions cross the ABI"; the reference lets nanoflann's
// std::runtime_error escape, NF.hpp:1228,1324).  EVERY extern "C" entry that returns a status is a function-try-block
//     int rsx_xxx(...) try { ... } RSX_CATCH_ALL
// so a std::bad_alloc from a host container (std::vector growth, new), a std::system_error from a mutex or anything else
// thrown below comes back as a status with rsx_last_error_string() set.
inline int on_exception() noexcept {
  int code = RSX_ERR_INTERNAL;
  try {
    try {
      throw;
    } catch (const std::bad_alloc &) {
      code = RSX_ERR_OOM;
      return fail(code, "host allocation failed (std::bad_alloc)");
    } catch (const std::length_error &e) {
      code = RSX_ERR_OOM;
      return fail(code, "host allocation failed (%s)", e.what());
    } catch (const std::exception &e) {
      return fail(code, "unexpected exception: %s", e.what());
    } catch (...) {
      return fail(code, "unexpected exception");
    }
  } catch (...) {  // (recording the message itself failed)
    return code;
  }
}
#define RSX_CATCH_ALL \
  catch (...) {       \
    return rsx::on_exception(); \
  }

#define RSX_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return rsx::fail(_e == hipErrorOutOfMemory ? RSX_ERR_OOM : RSX_ERR_HIP, "%s failed: %s", \
                       #expr, hipGetErrorString(_e));                                         \
  } while (0)

#define RSX_TRY(expr)          \
  do {                         \
    int _s = (expr);           \
    if (_s != RSX_OK) return _s; \
  } while (0)

// growable device buffer (never shrinks); contents preserved on growth when keep=true
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int reserve(size_t want, hipStream_t s, bool keep) {
    if (want <= bytes) return RSX_OK;
    size_t nb = bytes ? bytes : 4096;
    while (nb < want) nb *= 2;
    void *np = nullptr;
    RSX_HIP(hipMalloc(&np, nb));
    if (keep && p && bytes) {
      hipError_t e = hipMemcpyAsync(np, p, bytes, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        (void)hipFree(np);
        return fail(RSX_ERR_HIP, "DevBuf grow copy: %s", hipGetErrorString(e));
      }
    } else if (p) {
      hipError_t e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        (void)hipFree(np);
        return fail(RSX_ERR_HIP, "DevBuf grow sync: %s", hipGetErrorString(e));
      }
    }
    if (p) (void)hipFree(p);
    p = np;
    bytes = nb;
    return RSX_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <typename T>
  T *as() const {
    return static_cast<T *>(p);
  }
};

}  // namespace rsx
