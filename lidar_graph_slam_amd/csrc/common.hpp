// Host-side plumbing shared by the translation units of libmi355_lidar.so: error handling, grow-only device
// buffers, the context object behind mi355_ctx.  gfx950 only — no CPU fallback exists: every entry point
// needs a HIP device and fails with MI355_EHIP when there is none.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi355_lidar.h"

namespace mi355 {

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct ArgError : std::runtime_error {
  int code;
  ArgError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define MI_HIP(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess)                                                                                  \
      throw ::mi355::HipError(std::string(#expr) + " -> " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

#define MI_LAUNCH_CHECK() MI_HIP(hipGetLastError())

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  // grow-only; contents are NOT preserved
  void ensure(size_t need) {
    if (need <= n) return;
    release();
    size_t cap = need + need / 8 + 64;
    MI_HIP(hipMalloc((void**)&p, cap * sizeof(T)));
    n = cap;
  }
};

template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  size_t n = 0;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  void ensure(size_t need) {
    if (need <= n) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    MI_HIP(hipHostMalloc((void**)&p, need * sizeof(T), hipHostMallocDefault));
    n = need;
  }
};

inline int div_up(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace mi355

struct mi355_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  float grid_cell = 0.5f;
  int compute_units = 0;
  char name[128] = {0};
};
