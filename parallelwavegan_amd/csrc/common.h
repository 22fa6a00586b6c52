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
// PWG_PROF_SHAPES=1: launches are accounted per problem shape instead of per kernel family; returns an
// interned (pointer-stable) name built from the format, else `family` itself
const char* prof_shape_name(const char* family, const char* fmt, ...);
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

// Raw buffer descriptor whose inputs are forced wave-uniform with readfirstlane, so that hipcc keeps
// it in SGPRs instead of wrapping every buffer op in a waterfall loop (cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_buffer_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  const unsigned n = __builtin_amdgcn_readfirstlane(bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}

// remembers the largest dynamic-LDS limit already configured per kernel (so that launches inside a
// hipGraph capture never call hipFuncSetAttribute again)
bool lds_limit_is_set(const void* kern, size_t bytes);
// PWG_POISON_LDS=1 (debugging): NaN-fill every CU's LDS before an MFMA kernel launch (capi.hip)
void maybe_poison_lds(hipStream_t stream);

// Zero n floats with an ordinary kernel launch.  Used instead of hipMemsetAsync on every path that a training
// step captures into a hipGraph: a captured memset becomes a memset NODE, and the one failure mode the round-2
// bench ever showed (C2: garbage STFT-loss sums on a graph replay) sat exactly on the only buffer whose
// correctness depended on such a node (the unwritten tail of the partial-sum array).  Kernel nodes only.
void zero_fill(float* p, long n, hipStream_t stream);

// 1.0 = a launch has the chip to itself; < 1: it shares it with concurrent launches (pwg_set_concurrency_hint)
float concurrency_hint();

// gconv.hip: grouped k = 41 strided convolutions with 4 / 8 input channels per group on the 16 x 16 x 4 MFMA
// (`d` is always the FORWARD descriptor of the layer)
bool gconv_forward_applicable(const pwg_conv1d_desc* d, const float* add1, const float* add2);
int gconv_forward(const pwg_conv1d_desc* d, const float* x, const float* wp, const float* bias, float* y, hipStream_t stream);
bool gconv_dgrad_applicable(const pwg_conv1d_desc* d);
int gconv_backward_data(const pwg_conv1d_desc* d, const float* dy, const float* wp_bwd, const float* x_fwd, const float* accum,
                        float* dx, hipStream_t stream);
bool gconv_wgrad_applicable(const pwg_conv1d_desc* d);
size_t gconv_wgrad_workspace_floats(const pwg_conv1d_desc* d);
int gconv_backward_weight(const pwg_conv1d_desc* d, const float* x, const float* dy, float* dw, float* db, float* workspace,
                          size_t ws_floats, hipStream_t stream);

// wgrad_k1.hip: weight gradient of 1 x 1 convolutions with <= 96 channels and a long reduction (`d` flattened, forward
// descriptor); one tap-major slab (+ bias row when `write_bias`) per workgroup, `slab_stride` floats apart
bool k1_wgrad_applicable(const pwg_conv1d_desc* d);
int k1_wgrad_slabs(const pwg_conv1d_desc* d);
int k1_wgrad_launch(const pwg_conv1d_desc* d, const float* x, const float* dy, float* slabs, long slab_stride, int nslabs,
                    float slope_x, bool write_bias, hipStream_t stream);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// A stride-1 (k,1) convolution over (rows, width) is a plain 1-D convolution over the flattened
// rows*width axis with dilation*width and pad*width (the width index survives shifts by multiples of
// width, and zero padding above/below the rows is zero padding of the flat axis).  Flattening lets
// the period discriminators' 1024-channel layers run the fast width-1 paths.
static inline pwg_conv1d_desc flatten_width(const pwg_conv1d_desc& d) {
  pwg_conv1d_desc f = d;
  if (d.width > 1 && d.stride == 1 && d.pad_mode == PWG_PAD_ZERO) {
    f.t_in = d.t_in * d.width;
    f.t_out = d.t_out * d.width;
    f.dilation = d.dilation * d.width;
    f.pad_left = d.pad_left * d.width;
    f.width = 1;
  }
  return f;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case PWG_ACT_LEAKY_RELU: return v > 0.f ? v : v * slope;
    case PWG_ACT_RELU: return v > 0.f ? v : 0.f;
    case PWG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

}  // namespace pwg
