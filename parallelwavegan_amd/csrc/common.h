// Internal helpers shared by the gfx950 kernels.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/pwg_kernels.h"

namespace pwg {

void set_error(const char* fmt, ...);

#define PWG_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      ::pwg::set_error(__VA_ARGS__);  \
      return (code);                  \
    }                                 \
  } while (0)

#define PWG_CHECK_LAUNCH(what)                                               \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      ::pwg::set_error("%s: launch failed: %s", what, hipGetErrorString(e_)); \
      return PWG_ERR_LAUNCH;                                                 \
    }                                                                        \
  } while (0)

// Optional per-launch timing with HIP events recorded on the launch stream (pwg_prof_* in the
// public header).  Zero cost when disabled.
bool prof_enabled();
void prof_record(hipStream_t stream, const char* kernel, double flops, double bytes, bool begin);
struct ProfScope {
  hipStream_t s;
  const char* k;
  double f, b;
  bool on;
  ProfScope(hipStream_t stream, const char* kernel, double flops, double bytes)
      : s(stream), k(kernel), f(flops), b(bytes), on(prof_enabled()) {
    if (on) prof_record(s, k, f, b, true);
  }
  ~ProfScope() {
    if (on) prof_record(s, k, f, b, false);
  }
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case PWG_ACT_LEAKY_RELU: return v > 0.f ? v : v * slope;
    case PWG_ACT_RELU: return v > 0.f ? v : 0.f;
    case PWG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

}  // namespace pwg
