// reduce_optim.hip -- loss reductions (deterministic two-stage sums) and fused multi-tensor
// optimizer steps.  All HBM-bound: one read of each operand, 16-B accesses where aligned.
#include "common.h"

namespace pwg {

enum { RED_ABS_DIFF = 0, RED_SQ_DIFF = 1, RED_SQ = 2, RED_SQ_DIFF_CONST = 3, RED_SUM = 4, RED_HINGE_REAL = 5,
       RED_HINGE_FAKE = 6,
       // |lrelu(a) - lrelu(b)| with slope c: the feature-matching term on feature maps kept in pre-activation form
       // (layers/activation.py: PreActivated); d/da carries the activation's derivative, so no separate activation-
       // gradient pass exists
       RED_ABS_DIFF_LRELU = 7, RED_NUM_MODES = 8 };
__device__ __forceinline__ float red_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
constexpr int RED_BLOCKS = 512;

__device__ __forceinline__ float red_term(int mode, float a, float b, float c) {
  switch (mode) {
    case RED_ABS_DIFF: return fabsf(a - b);
    case RED_SQ_DIFF: return (a - b) * (a - b);
    case RED_SQ: return a * a;
    case RED_SQ_DIFF_CONST: return (a - c) * (a - c);
    case RED_HINGE_REAL: return -fminf(a - 1.f, 0.f);   // -min(x - 1, 0)   (adversarial_loss.py:119-120)
    case RED_HINGE_FAKE: return -fminf(-a - 1.f, 0.f);  // -min(-x - 1, 0)  (adversarial_loss.py:122-123)
    case RED_ABS_DIFF_LRELU: return fabsf(red_lrelu(a, c) - red_lrelu(b, c));
    default: return a;
  }
}
// d term / d a  (for the diff modes d/db = -d/da); ties of the hinge's min(., 0) get 1/2 like torch.minimum
__device__ __forceinline__ float red_dterm(int mode, float a, float b, float c) {
  switch (mode) {
    case RED_ABS_DIFF: {
      const float d = a - b;
      return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    }
    case RED_SQ_DIFF: return 2.f * (a - b);
    case RED_SQ: return 2.f * a;
    case RED_SQ_DIFF_CONST: return 2.f * (a - c);
    case RED_HINGE_REAL: return a < 1.f ? -1.f : (a == 1.f ? -0.5f : 0.f);
    case RED_HINGE_FAKE: return a > -1.f ? 1.f : (a == -1.f ? 0.5f : 0.f);
    case RED_ABS_DIFF_LRELU: {
      const float d = red_lrelu(a, c) - red_lrelu(b, c);
      return (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (a > 0.f ? 1.f : c);
    }
    default: return 1.f;
  }
}

__device__ __forceinline__ float block_sum(float s, float* red) {
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

// stage 1: partial[blockIdx] = sum over a grid-strided slice (fixed grid => deterministic)
__global__ void reduce_stage1_kernel(const float* a, const float* b, float c, long n, int mode, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    s += red_term(mode, a[i], b ? b[i] : 0.f, c);
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// stage 2: out[0] = (sum partial) * scale
__global__ void reduce_stage2_kernel(const float* partial, int np, float scale, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += blockDim.x) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// gradient of out = scale * sum term(a, b):  da = gout * scale * dterm/da  (db = -da for diffs)
__global__ void reduce_backward_kernel(const float* a, const float* b, float c, long n, int mode, float scale,
                                       const float* gout, float* da, float* db) {
  const float gs = gout[0] * scale;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float bv = b ? b[i] : 0.f;
    const float g = gs * red_dterm(mode, a[i], bv, c);
    if (da) da[i] = g;
    if (db) db[i] = mode == RED_ABS_DIFF_LRELU ? gs * red_dterm(mode, bv, a[i], c) : -g;
  }
}

// ---- multi-tensor loss reductions: out[slot] = sum_items scale_item * sum_i term(a_i, b_i | c) ---------
// The item table travels BY VALUE in the kernel arguments (<= 4 KiB), so a captured hipGraph holds it
// inside the kernel node: no device table, no host staging buffer that a later step could overwrite.
struct RedItems {
  pwg_red_item it[PWG_RED_MAX_ITEMS];
  int chunk_start[PWG_RED_MAX_ITEMS + 1];  // first chunk of item i (prefix sums), [n_items] = total chunks
  int n_items;
};
constexpr int RED_CHUNK = 8192;  // elements per workgroup

__device__ __forceinline__ int red_find_item(const RedItems& t, int chunk) {
  int i = 0;
  while (i + 1 < t.n_items && t.chunk_start[i + 1] <= chunk) ++i;
  return i;
}

// stage 1: partial[chunk] = scale_item * sum over the chunk (fixed element order per thread, fixed tree)
__global__ void __launch_bounds__(256) multi_reduce_stage1_kernel(const RedItems t, float* partial) {
  __shared__ float red[4];
  const int item = red_find_item(t, blockIdx.x);
  const pwg_red_item it = t.it[item];
  const long base = (long)(blockIdx.x - t.chunk_start[item]) * RED_CHUNK;
  const long end = base + RED_CHUNK < it.n ? base + RED_CHUNK : it.n;
  float s = 0.f;
  for (long i = base + threadIdx.x; i < end; i += 256) s += red_term(it.mode, it.a[i], it.b ? it.b[i] : 0.f, it.c);
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s * it.scale;
}
// stage 2: one workgroup per output slot sums the partials of its items in chunk order
__global__ void __launch_bounds__(256) multi_reduce_stage2_kernel(const RedItems t, const float* partial, float* out,
                                                                  int accumulate) {
  __shared__ float red[4];
  const int slot = blockIdx.x;
  float s = 0.f;
  for (int i = 0; i < t.n_items; ++i) {
    if (t.it[i].slot != slot) continue;
    for (int c = t.chunk_start[i] + threadIdx.x; c < t.chunk_start[i + 1]; c += 256) s += partial[c];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[slot] = accumulate ? out[slot] + s : s;
}
// backward: da = gout[slot] * scale * dterm/da, db = -da (diff modes)
__global__ void __launch_bounds__(256) multi_reduce_backward_kernel(const RedItems t, const float* gout) {
  const int item = red_find_item(t, blockIdx.x);
  const pwg_red_item it = t.it[item];
  if (!it.da && !it.db) return;
  const long base = (long)(blockIdx.x - t.chunk_start[item]) * RED_CHUNK;
  const long end = base + RED_CHUNK < it.n ? base + RED_CHUNK : it.n;
  const float gs = gout[it.slot] * it.scale;
  for (long i = base + threadIdx.x; i < end; i += 256) {
    const float bv = it.b ? it.b[i] : 0.f;
    const float g = gs * red_dterm(it.mode, it.a[i], bv, it.c);
    if (it.da) it.da[i] = g;
    // (diff modes: d/db = -d/da; the pre-activation form swaps the operands so that b's own derivative is used)
    if (it.db) it.db[i] = it.mode == RED_ABS_DIFF_LRELU ? gs * red_dterm(it.mode, bv, it.a[i], it.c) : -g;
  }
}

// ---- fused multi-tensor Adam / RAdam ------------------------------------------------------
struct OptChunk {  // one 64 Ki-element slice of one parameter tensor
  float* p;
  const float* g;
  float* m;
  float* v;
  float* vmax;  // amsgrad only (else NULL)
  int n;
  int pad_;
};

// torch.optim.Adam (non-fused, weight_decay L2, optional amsgrad) semantics:
//   g += wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
//   denom = sqrt(max? v) / sqrt(1 - b2^t) + eps;  p -= lr / (1 - b1^t) * m / denom
__global__ void adam_multi_kernel(const OptChunk* chunks, float lr, float b1, float b2, float eps, float wd,
                                  float bias_c1, float sqrt_bias_c2, float grad_scale) {
  const OptChunk c = chunks[blockIdx.x];
  const float step = lr / bias_c1;
  for (int i = threadIdx.x; i < c.n; i += blockDim.x) {
    float p = c.p[i];
    float g = c.g[i] * grad_scale;
    // wd > 0: torch.optim.Adam's L2 form (g += wd p);  wd < 0: torch.optim.AdamW's decoupled decay of |wd|
    // (p *= 1 - lr |wd| before the update, no gradient term)
    if (wd > 0.f) g += wd * p;
    else if (wd < 0.f) p *= (1.f + lr * wd);
    const float m = b1 * c.m[i] + (1.f - b1) * g;
    const float v = b2 * c.v[i] + (1.f - b2) * g * g;
    c.m[i] = m;
    c.v[i] = v;
    float vv = v;
    if (c.vmax) {
      vv = fmaxf(c.vmax[i], v);
      c.vmax[i] = vv;
    }
    const float denom = sqrtf(vv) / sqrt_bias_c2 + eps;
    c.p[i] = p - step * (m / denom);
  }
}

// RAdam as in the reference's optimizers/radam.py:27-99 (LiyuanLucasLiu):
//   v = b2 v + (1-b2) g^2 ; m = b1 m + (1-b1) g ; if wd: p -= wd*lr*p
//   N_sma >= 5:  p -= step_size * m / (sqrt(v) + eps)      else:  p -= step_size * m
// step_size (incl. the rectification term) is computed on the host per step.
__global__ void radam_multi_kernel(const OptChunk* chunks, float lr, float b1, float b2, float eps, float wd,
                                   float step_size, int rectified, float grad_scale) {
  const OptChunk c = chunks[blockIdx.x];
  for (int i = threadIdx.x; i < c.n; i += blockDim.x) {
    float p = c.p[i];
    const float g = c.g[i] * grad_scale;
    const float v = b2 * c.v[i] + (1.f - b2) * g * g;
    const float m = b1 * c.m[i] + (1.f - b1) * g;
    c.m[i] = m;
    c.v[i] = v;
    if (wd != 0.f) p += -wd * lr * p;
    if (rectified)
      p += -step_size * (m / (sqrtf(v) + eps));
    else
      p += -step_size * m;
    c.p[i] = p;
  }
}

// Variants reading their scalars from DEVICE memory, so that the launch can live inside a captured
// hipGraph while the host refreshes lr / bias corrections between replays:
//   hyper = [lr, beta1, beta2, eps, weight_decay, step_size, aux, grad_scale]
//   Adam : step_size = lr / (1 - beta1^t), aux = sqrt(1 - beta2^t)
//   RAdam: step_size as radam.py:63-86 (includes lr), aux = 1 if rectified (N_sma >= 5) else 0
__global__ void adam_multi_dev_kernel(const OptChunk* chunks, const float* hyper) {
  const OptChunk c = chunks[blockIdx.x];
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], step = hyper[5],
              sqrt_bias_c2 = hyper[6], grad_scale = hyper[7];
  for (int i = threadIdx.x; i < c.n; i += blockDim.x) {
    float p = c.p[i];
    float g = c.g[i] * grad_scale;
    if (wd > 0.f) g += wd * p;                // Adam: L2
    else if (wd < 0.f) p *= (1.f + lr * wd);  // AdamW: decoupled decay of |wd| (see adam_multi_kernel)
    const float m = b1 * c.m[i] + (1.f - b1) * g;
    const float v = b2 * c.v[i] + (1.f - b2) * g * g;
    c.m[i] = m;
    c.v[i] = v;
    float vv = v;
    if (c.vmax) {
      vv = fmaxf(c.vmax[i], v);
      c.vmax[i] = vv;
    }
    const float denom = sqrtf(vv) / sqrt_bias_c2 + eps;
    c.p[i] = p - step * (m / denom);
  }
}

__global__ void radam_multi_dev_kernel(const OptChunk* chunks, const float* hyper) {
  const OptChunk c = chunks[blockIdx.x];
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], step_size = hyper[5],
              grad_scale = hyper[7];
  const bool rectified = hyper[6] != 0.f;
  for (int i = threadIdx.x; i < c.n; i += blockDim.x) {
    float p = c.p[i];
    const float g = c.g[i] * grad_scale;
    const float v = b2 * c.v[i] + (1.f - b2) * g * g;
    const float m = b1 * c.m[i] + (1.f - b1) * g;
    c.m[i] = m;
    c.v[i] = v;
    if (wd != 0.f) p += -wd * lr * p;
    if (rectified)
      p += -step_size * (m / (sqrtf(v) + eps));
    else
      p += -step_size * m;
    c.p[i] = p;
  }
}

// sum of squares of many tensors (grad-norm clipping): partial per chunk, then one block
__global__ void sqsum_multi_kernel(const OptChunk* chunks, float* partial) {
  __shared__ float red[4];
  const OptChunk c = chunks[blockIdx.x];
  float s = 0.f;
  for (int i = threadIdx.x; i < c.n; i += blockDim.x) {
    const float g = c.g[i];
    s += g * g;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// out[0] = total_norm ; out[1] = clip coefficient = min(1, max_norm / (total_norm + 1e-6))
__global__ void clip_coef_kernel(const float* partial, int np, float max_norm, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += blockDim.x) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    out[0] = norm;
    const float c = max_norm / (norm + 1e-6f);
    out[1] = c < 1.f ? c : 1.f;
  }
}
__global__ void scale_grads_multi_kernel(const OptChunk* chunks, const float* coef) {
  const OptChunk c = chunks[blockIdx.x];
  const float k = coef[1];
  if (k >= 1.f) return;
  float* g = const_cast<float*>(c.g);
  for (int i = threadIdx.x; i < c.n; i += blockDim.x) g[i] *= k;
}

}  // namespace pwg

using namespace pwg;

// out[0] = scale * sum_i term(a_i, b_i)   mode: 0 |a-b|, 1 (a-b)^2, 2 a^2, 3 (a-c)^2, 4 a
// workspace: >= 512 floats.  Deterministic (fixed grid, fixed order).
extern "C" int pwg_reduce_forward(const float* a, const float* b, float c, int64_t n, int32_t mode, float scale,
                                  float* out, float* workspace, void* stream_) {
  PWG_REQUIRE(a && out && workspace, PWG_ERR_NULL, "reduce_forward: NULL pointer");
  PWG_REQUIRE((mode != RED_ABS_DIFF && mode != RED_SQ_DIFF && mode != RED_ABS_DIFF_LRELU) || b, PWG_ERR_NULL,
              "reduce_forward: mode needs b");
  PWG_REQUIRE(n > 0 && mode >= 0 && mode < RED_NUM_MODES, PWG_ERR_BAD_SHAPE, "reduce_forward: bad arguments");
  hipStream_t stream = (hipStream_t)stream_;
  long blocks = (n + 1023) / 1024;
  if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
  ProfScope prof(stream, "reduce_stage1_kernel", 0, 4.0 * n * (b ? 2 : 1));
  hipLaunchKernelGGL(reduce_stage1_kernel, dim3((int)blocks), dim3(256), 0, stream, a, b, c, (long)n, mode, workspace);
  hipLaunchKernelGGL(reduce_stage2_kernel, dim3(1), dim3(256), 0, stream, workspace, (int)blocks, scale, out);
  PWG_CHECK_LAUNCH("reduce_forward");
  return PWG_OK;
}

extern "C" int pwg_reduce_backward(const float* a, const float* b, float c, int64_t n, int32_t mode, float scale,
                                   const float* gout, float* da, float* db, void* stream) {
  PWG_REQUIRE(a && gout && (da || db), PWG_ERR_NULL, "reduce_backward: NULL pointer");
  PWG_REQUIRE(n > 0 && mode >= 0 && mode < RED_NUM_MODES, PWG_ERR_BAD_SHAPE, "reduce_backward: bad arguments");
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_backward_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, (long)n,
                     mode, scale, gout, da, db);
  PWG_CHECK_LAUNCH("reduce_backward");
  return PWG_OK;
}

// chunks: device array of n_chunks pwg_opt_chunk records (see header)
extern "C" int pwg_adam_step(const void* chunks, int32_t n_chunks, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int32_t step, float grad_scale, void* stream) {
  PWG_REQUIRE(chunks, PWG_ERR_NULL, "adam_step: NULL chunk table");
  PWG_REQUIRE(n_chunks > 0 && step >= 1, PWG_ERR_BAD_SHAPE, "adam_step: bad arguments");
  const double c1 = 1.0 - pow((double)beta1, step), c2 = 1.0 - pow((double)beta2, step);
  ProfScope prof((hipStream_t)stream, "adam_multi_kernel", 0, 0);  // (bytes: 28 per element; the chunk lengths are only in the device table)
  hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream,
                     (const OptChunk*)chunks, lr, beta1, beta2, eps, weight_decay, (float)c1, (float)sqrt(c2),
                     grad_scale);
  PWG_CHECK_LAUNCH("adam_step");
  return PWG_OK;
}

extern "C" int pwg_radam_step(const void* chunks, int32_t n_chunks, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int32_t step, float grad_scale, void* stream) {
  PWG_REQUIRE(chunks, PWG_ERR_NULL, "radam_step: NULL chunk table");
  PWG_REQUIRE(n_chunks > 0 && step >= 1, PWG_ERR_BAD_SHAPE, "radam_step: bad arguments");
  // optimizers/radam.py:63-86
  const double beta2_t = pow((double)beta2, step);
  const double n_sma_max = 2.0 / (1.0 - beta2) - 1.0;
  const double n_sma = n_sma_max - 2.0 * step * beta2_t / (1.0 - beta2_t);
  double step_size;
  int rectified = n_sma >= 5.0;
  if (rectified)
    step_size = lr * sqrt((1.0 - beta2_t) * (n_sma - 4.0) / (n_sma_max - 4.0) * (n_sma - 2.0) / n_sma * n_sma_max /
                          (n_sma_max - 2.0)) /
                (1.0 - pow((double)beta1, step));
  else
    step_size = lr / (1.0 - pow((double)beta1, step));
  hipLaunchKernelGGL(radam_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream,
                     (const OptChunk*)chunks, lr, beta1, beta2, eps, weight_decay, (float)step_size, rectified,
                     grad_scale);
  PWG_CHECK_LAUNCH("radam_step");
  return PWG_OK;
}

// torch.nn.utils.clip_grad_norm_ (L2) over a chunk table: out[0] = total norm, out[1] = coefficient
// applied; workspace >= n_chunks floats.
extern "C" int pwg_clip_grad_norm(const void* chunks, int32_t n_chunks, float max_norm, float* out, float* workspace,
                                  void* stream_) {
  PWG_REQUIRE(chunks && out && workspace, PWG_ERR_NULL, "clip_grad_norm: NULL pointer");
  PWG_REQUIRE(n_chunks > 0 && max_norm > 0.f, PWG_ERR_BAD_SHAPE, "clip_grad_norm: bad arguments");
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(sqsum_multi_kernel, dim3(n_chunks), dim3(256), 0, stream, (const OptChunk*)chunks, workspace);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, workspace, n_chunks, max_norm, out);
  hipLaunchKernelGGL(scale_grads_multi_kernel, dim3(n_chunks), dim3(256), 0, stream, (const OptChunk*)chunks, out);
  PWG_CHECK_LAUNCH("clip_grad_norm");
  return PWG_OK;
}

// hyper: 8 floats in device memory (layout above the *_dev kernels); the host refreshes it per step
extern "C" int pwg_adam_step_dev(const void* chunks, int32_t n_chunks, const float* hyper, void* stream) {
  PWG_REQUIRE(chunks && hyper, PWG_ERR_NULL, "adam_step_dev: NULL pointer");
  PWG_REQUIRE(n_chunks > 0, PWG_ERR_BAD_SHAPE, "adam_step_dev: bad arguments");
  ProfScope prof((hipStream_t)stream, "adam_multi_kernel", 0, 0);  // (bytes: 28 per element; the chunk lengths are only in the device table)
  hipLaunchKernelGGL(adam_multi_dev_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream,
                     (const OptChunk*)chunks, hyper);
  PWG_CHECK_LAUNCH("adam_step_dev");
  return PWG_OK;
}

extern "C" int pwg_radam_step_dev(const void* chunks, int32_t n_chunks, const float* hyper, void* stream) {
  PWG_REQUIRE(chunks && hyper, PWG_ERR_NULL, "radam_step_dev: NULL pointer");
  PWG_REQUIRE(n_chunks > 0, PWG_ERR_BAD_SHAPE, "radam_step_dev: bad arguments");
  hipLaunchKernelGGL(radam_multi_dev_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream,
                     (const OptChunk*)chunks, hyper);
  PWG_CHECK_LAUNCH("radam_step_dev");
  return PWG_OK;
}

// ---- multi-tensor reductions ---------------------------------------------------------------------
static int red_build(const pwg_red_item* items, int32_t n_items, int32_t n_slots, RedItems* t, const char* what) {
  PWG_REQUIRE(items, PWG_ERR_NULL, "%s: NULL item table", what);
  PWG_REQUIRE(n_items > 0 && n_items <= PWG_RED_MAX_ITEMS && n_slots > 0, PWG_ERR_BAD_SHAPE,
              "%s: need 1..%d items (got %d) and >= 1 slot", what, PWG_RED_MAX_ITEMS, n_items);
  long chunks = 0;
  for (int i = 0; i < n_items; ++i) {
    const pwg_red_item& it = items[i];
    PWG_REQUIRE(it.a, PWG_ERR_NULL, "%s: item %d has no operand", what, i);
    PWG_REQUIRE(it.n > 0 && it.mode >= 0 && it.mode < RED_NUM_MODES && it.slot >= 0 && it.slot < n_slots,
                PWG_ERR_BAD_SHAPE, "%s: item %d: bad n / mode / slot", what, i);
    PWG_REQUIRE((it.mode != RED_ABS_DIFF && it.mode != RED_SQ_DIFF && it.mode != RED_ABS_DIFF_LRELU) || it.b, PWG_ERR_NULL,
                "%s: item %d: mode %d needs a second operand", what, i, it.mode);
    t->it[i] = it;
    t->chunk_start[i] = (int)chunks;
    chunks += (it.n + RED_CHUNK - 1) / RED_CHUNK;
    PWG_REQUIRE(chunks < (1L << 30), PWG_ERR_BAD_SHAPE, "%s: too many elements", what);
  }
  t->chunk_start[n_items] = (int)chunks;
  t->n_items = n_items;
  return PWG_OK;
}

extern "C" size_t pwg_multi_reduce_workspace_floats(const pwg_red_item* items, int32_t n_items) {
  if (!items || n_items <= 0) return 0;
  size_t chunks = 0;
  for (int i = 0; i < n_items; ++i) chunks += (size_t)((items[i].n + RED_CHUNK - 1) / RED_CHUNK);
  return chunks;
}

extern "C" int pwg_multi_reduce_forward(const pwg_red_item* items, int32_t n_items, int32_t n_slots, float* out,
                                        int32_t accumulate, float* workspace, void* stream_) {
  RedItems t;
  const int rc = red_build(items, n_items, n_slots, &t, "multi_reduce_forward");
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(out && workspace, PWG_ERR_NULL, "multi_reduce_forward: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  double bytes = 0;
  for (int i = 0; i < n_items; ++i) bytes += 4.0 * items[i].n * (items[i].b ? 2 : 1);
  ProfScope prof(stream, "multi_reduce_kernel", 0, bytes);
  hipLaunchKernelGGL(multi_reduce_stage1_kernel, dim3(t.chunk_start[n_items]), dim3(256), 0, stream, t, workspace);
  hipLaunchKernelGGL(multi_reduce_stage2_kernel, dim3(n_slots), dim3(256), 0, stream, t, (const float*)workspace, out,
                     accumulate);
  PWG_CHECK_LAUNCH("multi_reduce_forward");
  return PWG_OK;
}

extern "C" int pwg_multi_reduce_backward(const pwg_red_item* items, int32_t n_items, int32_t n_slots,
                                         const float* gout, void* stream_) {
  RedItems t;
  const int rc = red_build(items, n_items, n_slots, &t, "multi_reduce_backward");
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(gout, PWG_ERR_NULL, "multi_reduce_backward: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  double bytes = 0;
  for (int i = 0; i < n_items; ++i)
    bytes += 4.0 * items[i].n * ((items[i].b ? 2 : 1) + (items[i].da ? 1 : 0) + (items[i].db ? 1 : 0));
  ProfScope prof(stream, "multi_reduce_backward_kernel", 0, bytes);
  hipLaunchKernelGGL(multi_reduce_backward_kernel, dim3(t.chunk_start[n_items]), dim3(256), 0, stream, t, gout);
  PWG_CHECK_LAUNCH("multi_reduce_backward");
  return PWG_OK;
}
