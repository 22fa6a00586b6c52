// elementwise.hip -- HBM-bound helper kernels around the convolutions: activation backward,
// weight/spectral-norm reparametrisation (forward pieces are in conv1d.hip), pooling, explicit
// padding, STFT framing / magnitude / log.  All are single-pass, coalesced along time.
#include "common.h"

namespace pwg {

static inline int grid_for(long n, int block = 256, int max_blocks = 4096) {
  long b = (n + block - 1) / block;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

#define GRID_STRIDE(i, n) \
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// dx = dy * act'(.) * scale, with act' expressed through the activation OUTPUT y
//   tanh: 1 - y^2;  leaky_relu (slope > 0): y > 0 ? 1 : slope;  relu: y > 0;  none: 1
__global__ void act_backward_kernel(const float* dy, const float* y, float* dx, long n, int act, float slope,
                                    float scale) {
  GRID_STRIDE(i, n) {
    float g = dy[i] * scale;
    if (act == PWG_ACT_TANH) {
      const float t = y[i];
      g *= (1.f - t * t);
    } else if (act == PWG_ACT_LEAKY_RELU) {
      g *= (y[i] > 0.f ? 1.f : slope);
    } else if (act == PWG_ACT_RELU) {
      g *= (y[i] > 0.f ? 1.f : 0.f);
    }
    dx[i] = g;
  }
}

__device__ __forceinline__ float block_sum_256(float s, float* red) {
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

// old-style weight_norm (dim 0) backward; one workgroup per dim-0 slice:
//   w = g v / |v|   =>   dg = <dw, v> / |v| ;  dv = (g/|v|) (dw - v <dw,v> / |v|^2)
__global__ void weight_norm_backward_kernel(const float* dw, const float* v, const float* g, float* dv, float* dg,
                                            int inner) {
  __shared__ float red[4];
  const long base = (long)blockIdx.x * inner;
  float svv = 0.f, sdv = 0.f;
  for (int i = threadIdx.x; i < inner; i += blockDim.x) {
    const float a = v[base + i], b = dw[base + i];
    svv += a * a;
    sdv += a * b;
  }
  svv = block_sum_256(svv, red);
  sdv = block_sum_256(sdv, red);
  const float norm = sqrtf(svv);
  const float gg = g[blockIdx.x];
  if (threadIdx.x == 0) dg[blockIdx.x] = sdv / norm;
  const float c1 = gg / norm, c2 = sdv / svv;
  for (int i = threadIdx.x; i < inner; i += blockDim.x) dv[base + i] = c1 * (dw[base + i] - v[base + i] * c2);
}

// AvgPool1d over rows of length t_in
__global__ void avg_pool1d_fwd_kernel(const float* x, float* y, long rows, int t_in, int t_out, int k, int s, int p,
                                      int count_include_pad) {
  const long n = rows * t_out;
  GRID_STRIDE(i, n) {
    const long r = i / t_out;
    const int o = (int)(i - r * t_out);
    const int start = o * s - p;
    int lo = start < 0 ? 0 : start;
    int hi = start + k > t_in ? t_in : start + k;
    float acc = 0.f;
    const float* xr = x + r * t_in;
    for (int t = lo; t < hi; ++t) acc += xr[t];
    // torch: with count_include_pad the divisor counts padded zeros but not positions past t_in + p
    int div;
    if (count_include_pad) {
      int e = start + k;
      if (e > t_in + p) e = t_in + p;
      div = e - start;
    } else {
      div = hi - lo;
    }
    y[i] = acc / (float)div;
  }
}

__global__ void avg_pool1d_bwd_kernel(const float* dy, float* dx, long rows, int t_in, int t_out, int k, int s, int p,
                                      int count_include_pad) {
  const long n = rows * t_in;
  GRID_STRIDE(i, n) {
    const long r = i / t_in;
    const int t = (int)(i - r * t_in);
    // outputs o with o*s - p <= t < o*s - p + k
    int o_hi = (t + p) / s;
    int o_lo = (t + p - k + s) / s;  // ceil((t + p - k + 1) / s)
    if (t + p - k + 1 <= 0) o_lo = 0;
    if (o_hi > t_out - 1) o_hi = t_out - 1;
    float acc = 0.f;
    const float* gr = dy + r * t_out;
    for (int o = o_lo; o <= o_hi; ++o) {
      const int start = o * s - p;
      int div;
      if (count_include_pad) {
        int e = start + k;
        if (e > t_in + p) e = t_in + p;
        div = e - start;
      } else {
        const int lo = start < 0 ? 0 : start;
        const int hi = start + k > t_in ? t_in : start + k;
        div = hi - lo;
      }
      acc += gr[o] / (float)div;
    }
    dx[i] = acc;
  }
}

__device__ __forceinline__ int pad_src(int j, int t_in, int mode) {  // j: index in the un-padded frame
  if (j >= 0 && j < t_in) return j;
  if (mode == PWG_PAD_REFLECT) return j < 0 ? -j : 2 * (t_in - 1) - j;
  if (mode == PWG_PAD_REPLICATE) return j < 0 ? 0 : t_in - 1;
  return -1;
}

// explicit padding: y[r][i] = x[r][src(i - pl)]
__global__ void pad1d_fwd_kernel(const float* x, float* y, long rows, int t_in, int pl, int pr, int mode) {
  const int t_out = t_in + pl + pr;
  const long n = rows * t_out;
  GRID_STRIDE(i, n) {
    const long r = i / t_out;
    const int o = (int)(i - r * t_out);
    const int j = pad_src(o - pl, t_in, mode);
    y[i] = j >= 0 ? x[r * t_in + j] : 0.f;
  }
}

// dx[r][t] = dy[r][t + pl] + reflected / replicated contributions (gather form, no atomics)
__global__ void pad1d_bwd_kernel(const float* dy, float* dx, long rows, int t_in, int pl, int pr, int mode) {
  const int t_out = t_in + pl + pr;
  const long n = rows * t_in;
  GRID_STRIDE(i, n) {
    const long r = i / t_in;
    const int t = (int)(i - r * t_in);
    const float* g = dy + r * t_out;
    float acc = g[t + pl];
    if (mode == PWG_PAD_REFLECT) {
      if (t >= 1 && t <= pl) acc += g[pl - t];                                  // left mirror of x[t]
      if (t <= t_in - 2 && t >= t_in - 1 - pr) acc += g[pl + 2 * (t_in - 1) - t];  // right mirror
    } else if (mode == PWG_PAD_REPLICATE) {
      if (t == 0)
        for (int o = 0; o < pl; ++o) acc += g[o];
      if (t == t_in - 1)
        for (int o = 0; o < pr; ++o) acc += g[pl + t_in + o];
    }
    dx[i] = acc;
  }
}

// STFT framing as a channel fold: y[b][c][n] = xr[b][n*hop + c - pad], c < hop, n < nn, where xr is x
// extended by reflection (torch.stft center=True pads n_fft/2; a window shorter than n_fft just
// shifts the frame start, which the caller folds into `pad`).  Indices further than one
// reflection away from the signal read 0 (they only meet zero filter taps).
__global__ void frame_fold_fwd_kernel(const float* x, float* y, int batch, int t, int pad, int hop, int nn) {
  const long n = (long)batch * hop * nn;
  GRID_STRIDE(i, n) {
    long r = i / nn;
    const int col = (int)(i - r * nn);
    const int c = (int)(r % hop);
    const int b = (int)(r / hop);
    const int j = col * hop + c - pad;  // index in the un-padded signal
    float v = 0.f;
    if (j > -t && j < 2 * t - 1) v = x[(long)b * t + pad_src(j, t, PWG_PAD_REFLECT)];
    y[i] = v;
  }
}

__global__ void frame_fold_bwd_kernel(const float* dy, float* dx, int batch, int t, int pad, int hop, int nn) {
  const long n = (long)batch * t;
  GRID_STRIDE(i, n) {
    const int b = (int)(i / t);
    const int tt = (int)(i - (long)b * t);
    const float* g = dy + (long)b * hop * nn;
    float acc = 0.f;
    // folded positions p = j + pad whose source is x[tt]: j = tt, j = -tt (tt >= 1), j = 2(t-1) - tt (tt <= t-2)
    int js[3];
    int np = 0;
    js[np++] = tt;
    if (tt >= 1) js[np++] = -tt;
    if (tt <= t - 2) js[np++] = 2 * (t - 1) - tt;
    for (int q = 0; q < np; ++q) {
      const int p = js[q] + pad;
      if (p < 0) continue;
      const int col = p / hop, c = p - col * hop;
      if (col < nn) acc += g[(long)c * nn + col];
    }
    dx[i] = acc;
  }
}

// spec: (B, 2*bins, frames) with rows [0,bins) = real, [bins, 2 bins) = imaginary
// mag = sqrt(max(re^2 + im^2, eps))      (losses/stft_loss.py:36-40, losses/mel_loss.py:101-104)
__global__ void stft_mag_fwd_kernel(const float* spec, float* mag, int batch, int bins, int frames, float eps) {
  const long n = (long)batch * bins * frames;
  const long plane = (long)bins * frames;
  GRID_STRIDE(i, n) {
    const long b = i / plane;
    const long r = i - b * plane;
    const float re = spec[b * 2 * plane + r], im = spec[b * 2 * plane + plane + r];
    const float p = re * re + im * im;
    mag[i] = sqrtf(p > eps ? p : eps);
  }
}

__global__ void stft_mag_bwd_kernel(const float* spec, const float* mag, const float* dmag, float* dspec, int batch,
                                    int bins, int frames, float eps) {
  const long n = (long)batch * bins * frames;
  const long plane = (long)bins * frames;
  GRID_STRIDE(i, n) {
    const long b = i / plane;
    const long r = i - b * plane;
    const float re = spec[b * 2 * plane + r], im = spec[b * 2 * plane + plane + r];
    const float p = re * re + im * im;
    float gr = 0.f, gi = 0.f;
    if (p > eps) {  // clamp passes no gradient below the floor
      const float s = dmag[i] / mag[i];
      gr = s * re;
      gi = s * im;
    }
    dspec[b * 2 * plane + r] = gr;
    dspec[b * 2 * plane + plane + r] = gi;
  }
}

// y = log(max(x, eps)) / log_div     (log_div = 1, ln 2 or ln 10)
__global__ void log_clamp_fwd_kernel(const float* x, float* y, long n, float eps, float log_div) {
  GRID_STRIDE(i, n) {
    const float v = x[i] > eps ? x[i] : eps;
    y[i] = logf(v) / log_div;
  }
}

__global__ void log_clamp_bwd_kernel(const float* x, const float* dy, float* dx, long n, float eps, float log_div) {
  GRID_STRIDE(i, n) { dx[i] = x[i] > eps ? dy[i] / (x[i] * log_div) : 0.f; }
}

// ---- spectral norm (torch.nn.utils.spectral_norm, dim 0, 1 power iteration) --------------
// parts[rs][c] = sum over the rs-th slice of rows of W[r][c] u[r]   (coalesced over columns; a
// workgroup = 64 columns x 4 row lanes; the slices are summed in a fixed order by
// normalize_kernel: deterministic).  SN_SPLITS row slices keep >= 256 workgroups in flight for the
// 1024 x 5120 layer instead of 20 column blocks walking 1024 rows serially.
constexpr int SN_SPLITS = 32;
__global__ __launch_bounds__(256) void matvec_t_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                       float* __restrict__ parts, int rows, int cols, int rows_per_split) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_split;
  const int r1 = min(rows, r0 + rows_per_split);
  float acc = 0.f;
  if (c < cols)
    for (int r = r0 + rl; r < r1; r += 4) acc += w[(long)r * cols + c] * u[r];
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < cols) parts[(long)blockIdx.y * cols + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// s[r] = sum_c W[r][c] v[c]        (one workgroup per row)
__global__ void matvec_kernel(const float* w, const float* v, float* s, int cols) {
  __shared__ float red[4];
  const long base = (long)blockIdx.x * cols;
  float acc = 0.f;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) acc += w[base + c] * v[c];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) s[blockIdx.x] = acc;
}
// out = in / max(|in|, eps)   and optionally dot = <a, b>   (single workgroup)
// parts[0][i] = sum_p parts[p][i]   (fixed order)
__global__ void sum_parts_kernel(float* parts, int n, int nparts) {
  // (round 6: all the loads first -- the additions keep their order; 32 dependent load + add rounds were 8 us per launch,
  // 28 launches per HiFi-GAN V1 step on the spectrally normalised discriminator's branch)
  GRID_STRIDE(i, n) {
    float vals[SN_SPLITS];
#pragma unroll
    for (int p = 0; p < SN_SPLITS; ++p) vals[p] = p < nparts ? parts[(long)p * n + i] : 0.f;
    float t = vals[0];
#pragma unroll
    for (int p = 1; p < SN_SPLITS; ++p)
      if (p < nparts) t += vals[p];
    parts[i] = t;
  }
}
__global__ void normalize_kernel(const float* in, float* out, int n, float eps) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += in[i] * in[i];
  s = block_sum_256(s, red);
  const float d = fmaxf(sqrtf(s), eps);
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = in[i] / d;
}
__global__ void dot_kernel(const float* a, const float* b, float* out, int n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += a[i] * b[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s;
}
// Round 6: the power iteration in fewer dependent launches (8 + two host-side clones of u / v -> 4 or 5 per layer and
// forward; 32 layer-forwards per HiFi-GAN V1 training step, a serial chain of ~5 us launches on the first scale
// discriminator's branch: profiles/r06_graph_gaps.txt).  Same arithmetic, same summation orders, bit-identical results.
// s[r] = sum_c W[r][c] * (raw[c] / max(|raw|, eps)): matvec_kernel with normalize_kernel folded in -- every workgroup
// recomputes the norm of `raw` (cols floats, from L2) with normalize_kernel's strided sums and tree; workgroup 0 also
// writes the normalised vector to v and to v_saved (the copy the backward pass keeps).
__global__ void matvec_normalized_kernel(const float* w, const float* raw, float* v, float* v_saved, float* s, int cols,
                                         float eps) {
  __shared__ float red[4];
  float q = 0.f;
  for (int i = threadIdx.x; i < cols; i += blockDim.x) q += raw[i] * raw[i];
  q = block_sum_256(q, red);
  const float d = fmaxf(sqrtf(q), eps);
  const long base = (long)blockIdx.x * cols;
  float acc = 0.f;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    const float vc = raw[c] / d;
    if (blockIdx.x == 0) {
      v[c] = vc;
      v_saved[c] = vc;
    }
    acc += w[base + c] * vc;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) s[blockIdx.x] = acc;
}
// u = t / max(|t|, eps) (also to u_saved), sigma = <u, t> with t = W v: normalize_kernel + dot_kernel in one single-workgroup
// launch; the second matvec of the unfused chain recomputed exactly this t.
__global__ void sn_finalize_kernel(const float* t, float* u, float* u_saved, float* sigma, int n, float eps) {
  __shared__ float red[4];
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) q += t[i] * t[i];
  q = block_sum_256(q, red);
  const float d = fmaxf(sqrtf(q), eps);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float ui = t[i] / d;
    u[i] = ui;
    u_saved[i] = ui;
    s += ui * t[i];
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) sigma[0] = s;
}
// dW_orig = dW / sigma - (<dW, W_orig> / sigma^2) u v^T ; dot2[0] must hold <dW, W_orig>
// (round 6: `nb` > 0: dot_part holds the nb per-block shares of <dW, W_orig> and every workgroup adds them itself, in
// dot_finish_kernel's order -- one launch less per layer and backward pass; nb == 0: dot_part[0] is the finished sum)
__global__ void spectral_norm_bwd_kernel(const float* dw, const float* u, const float* v, const float* sigma,
                                         const float* dot_part, int nb, float* dwo, int rows, int cols) {
  __shared__ float red[4];
  const long n = (long)rows * cols;
  const float sg = sigma[0];
  float dot = dot_part[0];
  if (nb > 0) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) s += dot_part[i];
    dot = block_sum_256(s, red);
  }
  const float coef = dot / (sg * sg);
  GRID_STRIDE(i, n) {
    const int r = (int)(i / cols);
    const int c = (int)(i - (long)r * cols);
    dwo[i] = dw[i] / sg - coef * u[r] * v[c];
  }
}
// part[block] = this block's share of <a, b>; dot_finish_kernel adds the shares in block order (no atomics)
__global__ void dot_big_kernel(const float* a, const float* b, float* part, long n) {
  __shared__ float red[4];
  float s = 0.f;
  GRID_STRIDE(i, n) s += a[i] * b[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void dot_finish_kernel(const float* part, float* out, int nb) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s;
}
// w = w_orig / sigma
__global__ void div_scalar_kernel(const float* x, const float* s, float* y, long n) {
  const float d = s[0];
  GRID_STRIDE(i, n) y[i] = x[i] / d;
}


// ---- WaveNet gate: out = tanh(z[:, :C]) * sigmoid(z[:, C:])   (layers/residual_block.py:120-132)
__global__ void gate_fwd_kernel(const float* z, float* out, int batch, int c, long t) {
  const long n = (long)batch * c * t;
  const long plane = (long)c * t;
  GRID_STRIDE(i, n) {
    const long b = i / plane;
    const long r = i - b * plane;
    const float a = z[b * 2 * plane + r], g = z[b * 2 * plane + plane + r];
    out[i] = tanhf(a) * (1.f / (1.f + expf(-g)));
  }
}
__global__ void gate_bwd_kernel(const float* z, const float* dout, float* dz, int batch, int c, long t) {
  const long n = (long)batch * c * t;
  const long plane = (long)c * t;
  GRID_STRIDE(i, n) {
    const long b = i / plane;
    const long r = i - b * plane;
    const float a = z[b * 2 * plane + r], g = z[b * 2 * plane + plane + r];
    const float th = tanhf(a), sg = 1.f / (1.f + expf(-g));
    const float d = dout[i];
    dz[b * 2 * plane + r] = d * sg * (1.f - th * th);
    dz[b * 2 * plane + plane + r] = d * th * sg * (1.f - sg);
  }
}

// ---- PWG mel upsampler stage: nearest stretch x s along time fused with the (F, 2s+1) smoothing
// conv over (mel channel, time) (layers/upsample.py:43-45,88-103; F = freq_axis_kernel_size, zero padding (F-1)/2 on
// the channel axis):  y[b][c][t] = sum_f sum_j w[f][j] * x[b][c + f - pf][(t + j - pad) / s]
__global__ void stretch_conv_fwd_kernel(const float* x, const float* w, float* y, long rows, int t_in, int s, int k,
                                        int pad, int channels, int fk, int act, float slope) {
  const int t_out = t_in * s;
  const long n = rows * t_out;
  const int pf = (fk - 1) / 2;
  GRID_STRIDE(i, n) {
    const long r = i / t_out;
    const int t = (int)(i - r * t_out);
    const int c = (int)(r % channels);
    float acc = 0.f;
    for (int f = 0; f < fk; ++f) {
      const int cc = c + f - pf;
      if (cc < 0 || cc >= channels) continue;
      const float* xr = x + (r + f - pf) * t_in;
      const float* wf = w + f * k;
      for (int j = 0; j < k; ++j) {
        const int u = t + j - pad;
        if (u >= 0 && u < t_out) acc += wf[j] * xr[u / s];
      }
    }
    y[i] = apply_act(acc, act, slope);  // the optional nonlinearity after each stage (layers/upsample.py:105-110)
  }
}
// Round 6: the same sum for the recipes' geometry (no channel-axis taps, k = 2 s + 1, pad = s, s in {2, 4}, t_out % 4 == 0):
// a thread owns 4 CONSECUTIVE outputs, loads the <= 4 input samples they touch once (the generic kernel issues one
// load and one integer division per tap and stores 4 B per thread: 0.5 TB/s on a 21 MB write, profiles/r05_hbm_helpers.txt)
// and stores 16 B.  Same products in the same tap order (bit-identical): a padded tap adds w * 0.
template <int S>
__global__ __launch_bounds__(256) void stretch_conv_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ y, long rows, int t_in, int act,
                                                                float slope) {
  static_assert(4 % S == 0, "4 outputs per thread start on an input sample");
  constexpr int K = 2 * S + 1, NQ = (3 + 2 * S) / S + 1;
  const int t4 = t_in * S / 4;  // 16-B pieces per row
  const long n = rows * t4;
  float wr[K];
#pragma unroll
  for (int j = 0; j < K; ++j) wr[j] = w[j];
  GRID_STRIDE(i, n) {
    const long r = i / t4;
    const int p = (int)(i - r * t4);
    const int q0 = p * 4 / S - 1;  // input sample of u = t0 - S
    const float* xr = x + r * t_in;
    float xq[NQ];
#pragma unroll
    for (int m = 0; m < NQ; ++m) {
      const int q = q0 + m;
      xq[m] = (q >= 0 && q < t_in) ? xr[q] : 0.f;
    }
    float o4[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) acc = __builtin_fmaf(wr[j], xq[(o + j) / S], acc);  // (explicit: the generic kernel's
      // `acc += w * x` is one v_fmac; written the same way here hipcc shares the products between outputs as v_mul + v_add)
      o4[o] = apply_act(acc, act, slope);
    }
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
  }
}
__global__ void stretch_conv_bwd_data_kernel(const float* dy, const float* w, float* dx, long rows, int t_in, int s,
                                             int k, int pad, int channels, int fk) {
  const int t_out = t_in * s;
  const long n = rows * t_in;
  const int pf = (fk - 1) / 2;
  GRID_STRIDE(i, n) {
    const long r = i / t_in;
    const int q = (int)(i - r * t_in);
    const int c = (int)(r % channels);
    float acc = 0.f;
    for (int f = 0; f < fk; ++f) {
      const int co = c - f + pf;  // the output channel that read x[c] through tap row f
      if (co < 0 || co >= channels) continue;
      const float* g = dy + (r - f + pf) * t_out;
      const float* wf = w + f * k;
      for (int u = q * s; u < q * s + s; ++u)
        for (int j = 0; j < k; ++j) {
          const int t = u - j + pad;
          if (t >= 0 && t < t_out) acc += wf[j] * g[t];
        }
    }
    dx[i] = acc;
  }
}
// part[(f k + j) * gridDim.x + block] = this block's share of sum_{r,t} dy[r][t] * xs[r + f - pf][t + j - pad];
// sum_partials_kernel adds the blocks' shares in a fixed order (deterministic: no atomics)
__global__ void stretch_conv_bwd_weight_kernel(const float* dy, const float* x, float* part, long rows, int t_in, int s,
                                               int k, int pad, int channels, int fk) {
  __shared__ float red[4];
  const int t_out = t_in * s;
  const long n = rows * t_out;
  const int pf = (fk - 1) / 2;
  for (int f = 0; f < fk; ++f)
    for (int j = 0; j < k; ++j) {
      float acc = 0.f;
      GRID_STRIDE(i, n) {
        const long r = i / t_out;
        const int t = (int)(i - r * t_out);
        const int cc = (int)(r % channels) + f - pf;
        const int u = t + j - pad;
        if (u >= 0 && u < t_out && cc >= 0 && cc < channels) acc += dy[i] * x[(r + f - pf) * t_in + u / s];
      }
      acc = block_sum_256(acc, red);
      if (threadIdx.x == 0) part[(long)(f * k + j) * gridDim.x + blockIdx.x] = acc;
    }
}
// out[o] = sum_i part[o * nb + i], every launch in the same order (thread-strided partial sums, fixed tree)
__global__ void sum_partials_kernel(const float* part, float* out, int nb) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[(long)blockIdx.x * nb + i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}


// y = ((a + b) + c) / div   (MRF: cs = b0 + b1 + b2; c = cs / num_blocks, models/hifigan.py:186-190)
__global__ void add3_div_kernel(const float* a, const float* b, const float* c, float* y, long n, float div) {
  GRID_STRIDE(i, n) {
    float v = a[i] + b[i];
    if (c) v += c[i];
    y[i] = v / div;
  }
}

// ---- StyleMelGAN pieces (layers/tade_res_block.py) ------------------------------------------------
// InstanceNorm1d (affine = False, biased variance, eps): one workgroup per (batch, channel) row.
__global__ __launch_bounds__(256) void instance_norm_fwd_kernel(const float* x, float* y, float* mean, float* rstd,
                                                                int t, float eps) {
  __shared__ float red[4];
  const float* xr = x + (long)blockIdx.x * t;
  float s = 0.f;
  for (int i = threadIdx.x; i < t; i += 256) s += xr[i];
  const float mu = block_sum_256(s, red) / t;
  __syncthreads();
  float v = 0.f;
  for (int i = threadIdx.x; i < t; i += 256) {
    const float d = xr[i] - mu;
    v += d * d;
  }
  const float rs = rsqrtf(block_sum_256(v, red) / t + eps);
  float* yr = y + (long)blockIdx.x * t;
  for (int i = threadIdx.x; i < t; i += 256) yr[i] = (xr[i] - mu) * rs;
  if (threadIdx.x == 0) {
    mean[blockIdx.x] = mu;
    rstd[blockIdx.x] = rs;
  }
}
// dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat)),  xhat = y
__global__ __launch_bounds__(256) void instance_norm_bwd_kernel(const float* dy, const float* y, const float* rstd,
                                                                float* dx, int t) {
  __shared__ float red[4];
  const long base = (long)blockIdx.x * t;
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < t; i += 256) {
    a += dy[base + i];
    b += dy[base + i] * y[base + i];
  }
  const float ma = block_sum_256(a, red) / t;
  __syncthreads();
  const float mb = block_sum_256(b, red) / t;
  const float rs = rstd[blockIdx.x];
  for (int i = threadIdx.x; i < t; i += 256) dx[base + i] = rs * (dy[base + i] - ma - y[base + i] * mb);
}
// nearest upsampling along time (+ optional addend of the upsampled shape): y[r][t] = x[r][t / s] (+ add[r][t])
__global__ void upsample_nearest_fwd_kernel(const float* x, const float* add, float* y, long rows, int t_in, int s) {
  const int t_out = t_in * s;
  const long n = rows * t_out;
  GRID_STRIDE(i, n) {
    const long r = i / t_out;
    const int t = (int)(i - r * t_out);
    float v = x[r * t_in + t / s];
    if (add) v += add[i];
    y[i] = v;
  }
}
__global__ void upsample_nearest_bwd_kernel(const float* dy, float* dx, long rows, int t_in, int s) {
  const long n = rows * t_in;
  GRID_STRIDE(i, n) {
    const long r = i / t_in;
    const int q = (int)(i - r * t_in);
    const float* g = dy + r * (long)t_in * s + (long)q * s;
    float acc = 0.f;
    for (int j = 0; j < s; ++j) acc += g[j];
    dx[i] = acc;
  }
}
// TADE modulation (tade_res_block.py:66-73): y[b][c][t] = cg[b][c][t] * xn[b][c][t/s] + cg[b][C+c][t]
__global__ void tade_modulate_fwd_kernel(const float* xn, const float* cg, float* y, int batch, int channels, int t_in,
                                         int s) {
  const int t_out = t_in * s;
  const long n = (long)batch * channels * t_out;
  GRID_STRIDE(i, n) {
    const int t = (int)(i % t_out);
    const long r = i / t_out;  // b * C + c
    const int c = (int)(r % channels);
    const long b = r / channels;
    const long g1 = ((b * 2 * channels + c) * (long)t_out) + t;
    y[i] = cg[g1] * xn[r * t_in + t / s] + cg[g1 + (long)channels * t_out];
  }
}
// dcg (B, 2C, T*s) and dxn (B, C, T): dcg1 = dy * up(xn), dcg2 = dy, dxn[q] = sum_j dy[qs+j] * cg1[qs+j]
__global__ void tade_modulate_bwd_cg_kernel(const float* dy, const float* xn, float* dcg, int batch, int channels,
                                            int t_in, int s) {
  const int t_out = t_in * s;
  const long n = (long)batch * channels * t_out;
  GRID_STRIDE(i, n) {
    const int t = (int)(i % t_out);
    const long r = i / t_out;
    const int c = (int)(r % channels);
    const long b = r / channels;
    const long g1 = ((b * 2 * channels + c) * (long)t_out) + t;
    const float g = dy[i];
    dcg[g1] = g * xn[r * t_in + t / s];
    dcg[g1 + (long)channels * t_out] = g;
  }
}
__global__ void tade_modulate_bwd_x_kernel(const float* dy, const float* cg, float* dxn, int batch, int channels,
                                           int t_in, int s) {
  const int t_out = t_in * s;
  const long n = (long)batch * channels * t_in;
  GRID_STRIDE(i, n) {
    const int q = (int)(i % t_in);
    const long r = i / t_in;
    const int c = (int)(r % channels);
    const long b = r / channels;
    const float* g = dy + r * (long)t_out + (long)q * s;
    const float* m = cg + ((b * 2 * channels + c) * (long)t_out) + (long)q * s;
    float acc = 0.f;
    for (int j = 0; j < s; ++j) acc += g[j] * m[j];
    dxn[i] = acc;
  }
}
// gated activation of TADEResBlock (tade_res_block.py:151-158): z (B, 2C, T) -> y (B, C, T)
//   softmax: y = softmax_c(z[:, :C]) * tanh(z[:, C:])      sigmoid: y = sigmoid(z[:, :C]) * tanh(z[:, C:])
// one thread per (b, t) column, coalesced along t
__global__ void softmax_gate_fwd_kernel(const float* z, float* y, int batch, int channels, long t, int use_softmax) {
  const long n = (long)batch * t;
  GRID_STRIDE(i, n) {
    const long b = i / t;
    const long tt = i - b * t;
    const float* za = z + (b * 2 * channels) * t + tt;
    const float* zb = za + (long)channels * t;
    float* yo = y + (b * channels) * t + tt;
    if (use_softmax) {
      float mx = -INFINITY;
      for (int c = 0; c < channels; ++c) mx = fmaxf(mx, za[(long)c * t]);
      float den = 0.f;
      for (int c = 0; c < channels; ++c) den += expf(za[(long)c * t] - mx);
      for (int c = 0; c < channels; ++c) yo[(long)c * t] = expf(za[(long)c * t] - mx) / den * tanhf(zb[(long)c * t]);
    } else {
      for (int c = 0; c < channels; ++c)
        yo[(long)c * t] = 1.f / (1.f + expf(-za[(long)c * t])) * tanhf(zb[(long)c * t]);
    }
  }
}
__global__ void softmax_gate_bwd_kernel(const float* z, const float* dy, float* dz, int batch, int channels, long t,
                                        int use_softmax) {
  const long n = (long)batch * t;
  GRID_STRIDE(i, n) {
    const long b = i / t;
    const long tt = i - b * t;
    const float* za = z + (b * 2 * channels) * t + tt;
    const float* zb = za + (long)channels * t;
    const float* g = dy + (b * channels) * t + tt;
    float* da = dz + (b * 2 * channels) * t + tt;
    float* db = da + (long)channels * t;
    if (use_softmax) {
      float mx = -INFINITY;
      for (int c = 0; c < channels; ++c) mx = fmaxf(mx, za[(long)c * t]);
      float den = 0.f;
      for (int c = 0; c < channels; ++c) den += expf(za[(long)c * t] - mx);
      float dot = 0.f;  // sum_c dp_c * p_c with dp = dy * tanh(zb)
      for (int c = 0; c < channels; ++c) {
        const float p = expf(za[(long)c * t] - mx) / den;
        dot += g[(long)c * t] * tanhf(zb[(long)c * t]) * p;
      }
      for (int c = 0; c < channels; ++c) {
        const float p = expf(za[(long)c * t] - mx) / den;
        const float th = tanhf(zb[(long)c * t]);
        const float gy = g[(long)c * t];
        da[(long)c * t] = p * (gy * th - dot);
        db[(long)c * t] = gy * p * (1.f - th * th);
      }
    } else {
      for (int c = 0; c < channels; ++c) {
        const float sg = 1.f / (1.f + expf(-za[(long)c * t]));
        const float th = tanhf(zb[(long)c * t]);
        const float gy = g[(long)c * t];
        da[(long)c * t] = gy * th * sg * (1.f - sg);
        db[(long)c * t] = gy * sg * (1.f - th * th);
      }
    }
  }
}

// ---- UHiFiGAN pieces (models/uhifigan.py) ---------------------------------------------------------
// y[b][c_off + c][t] = x[b][c][t] for c < c_src (y has c_dst channels): the two halves of torch.cat(dim=1);
// reverse = true copies the slice of y back into x (the backward of the concatenation)
__global__ void copy_channels_kernel(float* x, float* y, int batch, int c_src, int c_dst, int c_off, long t,
                                     int reverse) {
  const long n = (long)batch * c_src * t;
  GRID_STRIDE(i, n) {
    const long tt = i % t;
    const long r = i / t;
    const int c = (int)(r % c_src);
    const long b = r / c_src;
    const long j = (b * c_dst + c_off + c) * t + tt;
    if (reverse) x[i] = y[j];
    else y[j] = x[i];
  }
}
// torch.nn.Dropout (training): keep with probability 1 - p, scale kept values by 1 / (1 - p).  The mask
// comes from a counter-based hash of (seed, element index), so backward regenerates it from the seed.
__device__ __forceinline__ unsigned hash_u32(unsigned long long v) {
  v ^= v >> 33;
  v *= 0xff51afd7ed558ccdULL;
  v ^= v >> 33;
  v *= 0xc4ceb9fe1a85ec53ULL;
  v ^= v >> 33;
  return (unsigned)v;
}
__global__ void dropout_kernel(const float* x, float* y, long n, float p, float scale, unsigned long long seed,
                               const unsigned long long* seed_dev) {
  if (seed_dev) seed += *seed_dev;  // device-resident part of the seed (advances between hipGraph replays)
  const unsigned thr = (unsigned)((double)p * 4294967296.0);  // drop when hash < thr
  GRID_STRIDE(i, n) y[i] = hash_u32(seed * 0x9E3779B97F4A7C15ULL + (unsigned long long)i) >= thr ? x[i] * scale : 0.f;
}

// pcm[i] = (int16) rint(clamp(x[i], -1, 1) * 32767)   (the PCM_16 write of bin/decode.py:235-243)
__global__ void wave_to_pcm16_kernel(const float* x, short* pcm, long n) {
  GRID_STRIDE(i, n) {
    const float v = fminf(fmaxf(x[i], -1.f), 1.f) * 32767.f;
    pcm[i] = (short)__float2int_rn(v);
  }
}

// y[b][c][t] = (x[b][t][c] - mean[c]) / scale[c]: feature normalisation fused with the (T', C) -> (C, T')
// transpose that `inference` performs (models/hifigan.py:262-266; bin/normalize.py:238-270)
__global__ void normalize_transpose_kernel(const float* x, const float* mean, const float* scale, float* y, int batch,
                                           int frames, int channels) {
  const long n = (long)batch * frames * channels;
  GRID_STRIDE(i, n) {
    const int t = (int)(i % frames);
    const long r = i / frames;
    const int c = (int)(r % channels);
    const long b = r / channels;
    float v = x[(b * frames + t) * channels + c];
    if (mean) v = (v - mean[c]) / scale[c];
    y[i] = v;
  }
}

// Random-crop batch assembly from a device-resident corpus (the Collater of bin/train.py:646-896
// without the host round trip):
//   y[b][0][i]  = audio[audio_off[utt[b]] + min(start[b]*hop + i, audio_len[utt[b]] - 1)]   (edge pad,
//                 as Collater._adjust_length pads a short waveform)
//   c[b][ch][f] = mel[(mel_off[utt[b]] + start[b] - acw + f) * channels + ch]                 (transpose)
__global__ void gather_crop_kernel(const float* audio, const long* audio_off, const long* audio_len, const float* mel,
                                   const long* mel_off, const int* utt, const int* start, float* y, float* c,
                                   int batch, int steps, int hop, int frames_ctx, int acw, int channels) {
  const long ny = (long)batch * steps;
  const long nc = (long)batch * channels * frames_ctx;
  GRID_STRIDE(i, ny + nc) {
    if (i < ny) {
      const int b = (int)(i / steps);
      const long k = (long)start[b] * hop + (i - (long)b * steps);
      const int u = utt[b];
      y[i] = audio[audio_off[u] + (k < audio_len[u] ? k : audio_len[u] - 1)];
    } else {
      const long j = i - ny;
      const int f = (int)(j % frames_ctx);
      const long r = j / frames_ctx;
      const int ch = (int)(r % channels);
      const int b = (int)(r / channels);
      c[j] = mel[(mel_off[utt[b]] + start[b] - acw + f) * channels + ch];
    }
  }
}

}  // namespace pwg

using namespace pwg;

#define LAUNCH1D(kern, n, stream, ...)                                                         \
  do {                                                                                         \
    hipLaunchKernelGGL(kern, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__); \
    PWG_CHECK_LAUNCH(#kern);                                                                   \
  } while (0)

extern "C" int pwg_act_backward(const float* dy, const float* y, float* dx, int64_t n, int32_t act, float slope,
                                float scale, void* stream) {
  PWG_REQUIRE(dy && dx && (y || act == PWG_ACT_NONE), PWG_ERR_NULL, "act_backward: NULL pointer");
  PWG_REQUIRE(n > 0, PWG_ERR_BAD_SHAPE, "act_backward: n must be positive");
  ProfScope prof((hipStream_t)stream, "act_backward_kernel", 0, 12.0 * n);
  LAUNCH1D(act_backward_kernel, n, stream, dy, y, dx, (long)n, act, slope, scale);
  return PWG_OK;
}

extern "C" int pwg_weight_norm_backward(const float* dw, const float* v, const float* g, float* dv, float* dg,
                                        int32_t n0, int32_t inner, void* stream) {
  PWG_REQUIRE(dw && v && g && dv && dg, PWG_ERR_NULL, "weight_norm_backward: NULL pointer");
  PWG_REQUIRE(n0 > 0 && inner > 0, PWG_ERR_BAD_SHAPE, "weight_norm_backward: bad shape");
  ProfScope prof((hipStream_t)stream, "weight_norm_backward_kernel", 0, 16.0 * n0 * inner);
  hipLaunchKernelGGL(weight_norm_backward_kernel, dim3(n0), dim3(256), 0, (hipStream_t)stream, dw, v, g, dv, dg, inner);
  PWG_CHECK_LAUNCH("weight_norm_backward");
  return PWG_OK;
}

extern "C" int pwg_avg_pool1d_forward(const float* x, float* y, int64_t rows, int32_t t_in, int32_t t_out,
                                      int32_t kernel, int32_t stride, int32_t pad, int32_t count_include_pad,
                                      void* stream) {
  PWG_REQUIRE(x && y, PWG_ERR_NULL, "avg_pool1d_forward: NULL pointer");
  PWG_REQUIRE(rows > 0 && t_in > 0 && t_out > 0 && kernel > 0 && stride > 0 && pad >= 0 && 2 * pad <= kernel,
              PWG_ERR_BAD_SHAPE, "avg_pool1d: bad geometry");
  const long n = rows * t_out;
  ProfScope prof((hipStream_t)stream, "avg_pool1d_fwd_kernel", 0, 4.0 * rows * (t_in + t_out));
  LAUNCH1D(avg_pool1d_fwd_kernel, n, stream, x, y, (long)rows, t_in, t_out, kernel, stride, pad, count_include_pad);
  return PWG_OK;
}

extern "C" int pwg_avg_pool1d_backward(const float* dy, float* dx, int64_t rows, int32_t t_in, int32_t t_out,
                                       int32_t kernel, int32_t stride, int32_t pad, int32_t count_include_pad,
                                       void* stream) {
  PWG_REQUIRE(dy && dx, PWG_ERR_NULL, "avg_pool1d_backward: NULL pointer");
  PWG_REQUIRE(rows > 0 && t_in > 0 && t_out > 0 && kernel > 0 && stride > 0 && pad >= 0, PWG_ERR_BAD_SHAPE,
              "avg_pool1d: bad geometry");
  const long n = rows * t_in;
  ProfScope prof((hipStream_t)stream, "avg_pool1d_bwd_kernel", 0, 4.0 * rows * (t_in + t_out));
  LAUNCH1D(avg_pool1d_bwd_kernel, n, stream, dy, dx, (long)rows, t_in, t_out, kernel, stride, pad, count_include_pad);
  return PWG_OK;
}

extern "C" int pwg_pad1d_forward(const float* x, float* y, int64_t rows, int32_t t_in, int32_t pad_left,
                                 int32_t pad_right, int32_t mode, void* stream) {
  PWG_REQUIRE(x && y, PWG_ERR_NULL, "pad1d_forward: NULL pointer");
  PWG_REQUIRE(rows > 0 && t_in > 0 && pad_left >= 0 && pad_right >= 0, PWG_ERR_BAD_SHAPE, "pad1d: bad geometry");
  PWG_REQUIRE(mode != PWG_PAD_REFLECT || (pad_left < t_in && pad_right < t_in), PWG_ERR_BAD_SHAPE,
              "pad1d: reflect padding (%d,%d) must be smaller than the input length %d", pad_left, pad_right, t_in);
  const long n = rows * (t_in + pad_left + pad_right);
  LAUNCH1D(pad1d_fwd_kernel, n, stream, x, y, (long)rows, t_in, pad_left, pad_right, mode);
  return PWG_OK;
}

extern "C" int pwg_pad1d_backward(const float* dy, float* dx, int64_t rows, int32_t t_in, int32_t pad_left,
                                  int32_t pad_right, int32_t mode, void* stream) {
  PWG_REQUIRE(dy && dx, PWG_ERR_NULL, "pad1d_backward: NULL pointer");
  PWG_REQUIRE(rows > 0 && t_in > 0 && pad_left >= 0 && pad_right >= 0, PWG_ERR_BAD_SHAPE, "pad1d: bad geometry");
  const long n = rows * t_in;
  LAUNCH1D(pad1d_bwd_kernel, n, stream, dy, dx, (long)rows, t_in, pad_left, pad_right, mode);
  return PWG_OK;
}

extern "C" int pwg_frame_fold_forward(const float* x, float* y, int32_t batch, int32_t t, int32_t pad, int32_t hop,
                                      int32_t n_cols, void* stream) {
  PWG_REQUIRE(x && y, PWG_ERR_NULL, "frame_fold_forward: NULL pointer");
  // pad < 0 (torch.stft(center=False) with a window shorter than n_fft): the fold starts at x[-pad]
  PWG_REQUIRE(batch > 0 && t > pad && pad > -t && hop > 0 && n_cols > 0, PWG_ERR_BAD_SHAPE,
              "frame_fold: bad geometry (T=%d pad=%d hop=%d)", t, pad, hop);
  const long n = (long)batch * hop * n_cols;
  ProfScope prof((hipStream_t)stream, "frame_fold_fwd_kernel", 0, 4.0 * ((double)batch * t + n));
  LAUNCH1D(frame_fold_fwd_kernel, n, stream, x, y, batch, t, pad, hop, n_cols);
  return PWG_OK;
}

extern "C" int pwg_frame_fold_backward(const float* dy, float* dx, int32_t batch, int32_t t, int32_t pad,
                                       int32_t hop, int32_t n_cols, void* stream) {
  PWG_REQUIRE(dy && dx, PWG_ERR_NULL, "frame_fold_backward: NULL pointer");
  PWG_REQUIRE(batch > 0 && t > pad && pad > -t && hop > 0 && n_cols > 0, PWG_ERR_BAD_SHAPE, "frame_fold: bad geometry");
  const long n = (long)batch * t;
  LAUNCH1D(frame_fold_bwd_kernel, n, stream, dy, dx, batch, t, pad, hop, n_cols);
  return PWG_OK;
}

extern "C" int pwg_stft_mag_forward(const float* spec, float* mag, int32_t batch, int32_t bins, int32_t frames,
                                    float eps, void* stream) {
  PWG_REQUIRE(spec && mag, PWG_ERR_NULL, "stft_mag_forward: NULL pointer");
  PWG_REQUIRE(batch > 0 && bins > 0 && frames > 0, PWG_ERR_BAD_SHAPE, "stft_mag: bad shape");
  const long n = (long)batch * bins * frames;
  ProfScope prof((hipStream_t)stream, "stft_mag_fwd_kernel", 0, 12.0 * n);
  LAUNCH1D(stft_mag_fwd_kernel, n, stream, spec, mag, batch, bins, frames, eps);
  return PWG_OK;
}

extern "C" int pwg_stft_mag_backward(const float* spec, const float* mag, const float* dmag, float* dspec,
                                     int32_t batch, int32_t bins, int32_t frames, float eps, void* stream) {
  PWG_REQUIRE(spec && mag && dmag && dspec, PWG_ERR_NULL, "stft_mag_backward: NULL pointer");
  PWG_REQUIRE(batch > 0 && bins > 0 && frames > 0, PWG_ERR_BAD_SHAPE, "stft_mag: bad shape");
  const long n = (long)batch * bins * frames;
  LAUNCH1D(stft_mag_bwd_kernel, n, stream, spec, mag, dmag, dspec, batch, bins, frames, eps);
  return PWG_OK;
}

extern "C" int pwg_log_clamp_forward(const float* x, float* y, int64_t n, float eps, float log_div, void* stream) {
  PWG_REQUIRE(x && y, PWG_ERR_NULL, "log_clamp_forward: NULL pointer");
  PWG_REQUIRE(n > 0 && log_div > 0.f, PWG_ERR_BAD_SHAPE, "log_clamp: bad arguments");
  LAUNCH1D(log_clamp_fwd_kernel, n, stream, x, y, (long)n, eps, log_div);
  return PWG_OK;
}

extern "C" int pwg_log_clamp_backward(const float* x, const float* dy, float* dx, int64_t n, float eps, float log_div,
                                      void* stream) {
  PWG_REQUIRE(x && dy && dx, PWG_ERR_NULL, "log_clamp_backward: NULL pointer");
  PWG_REQUIRE(n > 0 && log_div > 0.f, PWG_ERR_BAD_SHAPE, "log_clamp: bad arguments");
  LAUNCH1D(log_clamp_bwd_kernel, n, stream, x, dy, dx, (long)n, eps, log_div);
  return PWG_OK;
}

// u (rows), v (cols) are updated in place when do_iter != 0 (training-mode forward of
// torch.nn.utils.spectral_norm: models/hifigan.py:613-621); sigma[0] = u^T W v; w = w_orig / sigma.
// tmp: workspace of max(rows, 32 * cols) floats (row-sliced partial sums of W^T u).
extern "C" int pwg_spectral_norm_forward(const float* w_orig, float* u, float* v, float* sigma, float* w,
                                         float* tmp, int32_t rows, int32_t cols, int32_t do_iter, float eps,
                                         void* stream_) {
  PWG_REQUIRE(w_orig && u && v && sigma && w && tmp, PWG_ERR_NULL, "spectral_norm_forward: NULL pointer");
  PWG_REQUIRE(rows > 0 && cols > 0, PWG_ERR_BAD_SHAPE, "spectral_norm: bad shape");
  hipStream_t stream = (hipStream_t)stream_;
  if (do_iter) {
    const int splits = rows >= 4 * SN_SPLITS ? SN_SPLITS : 1;
    const int rps = (rows + splits - 1) / splits;
    hipLaunchKernelGGL(matvec_t_kernel, dim3((cols + 63) / 64, splits), dim3(256), 0, stream, w_orig, u, tmp, rows,
                       cols, rps);
    if (splits > 1) hipLaunchKernelGGL(sum_parts_kernel, dim3(grid_for(cols)), dim3(256), 0, stream, tmp, cols, splits);
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, stream, tmp, v, cols, eps);
    hipLaunchKernelGGL(matvec_kernel, dim3(rows), dim3(256), 0, stream, w_orig, v, tmp, cols);
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, stream, tmp, u, rows, eps);
  }
  hipLaunchKernelGGL(matvec_kernel, dim3(rows), dim3(256), 0, stream, w_orig, v, tmp, cols);
  hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(256), 0, stream, u, tmp, sigma, rows);
  const long n = (long)rows * cols;
  hipLaunchKernelGGL(div_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, stream, w_orig, sigma, w, n);
  PWG_CHECK_LAUNCH("spectral_norm_forward");
  return PWG_OK;
}

// The same, additionally leaving the u / v of THIS forward in u_saved / v_saved (what torch's hook clones for the backward
// pass: later forwards update u and v in place).  do_iter == 0: nothing is updated, the copies are plain device copies.
extern "C" int pwg_spectral_norm_forward_saved(const float* w_orig, float* u, float* v, float* sigma, float* w, float* tmp,
                                               float* u_saved, float* v_saved, int32_t rows, int32_t cols, int32_t do_iter,
                                               float eps, void* stream_) {
  PWG_REQUIRE(w_orig && u && v && sigma && w && tmp && u_saved && v_saved, PWG_ERR_NULL,
              "spectral_norm_forward_saved: NULL pointer");
  PWG_REQUIRE(rows > 0 && cols > 0, PWG_ERR_BAD_SHAPE, "spectral_norm: bad shape");
  hipStream_t stream = (hipStream_t)stream_;
  static const bool fused = !(getenv("PWG_SN_FUSED") && atoi(getenv("PWG_SN_FUSED")) == 0);
  const int splits = rows >= 4 * SN_SPLITS ? SN_SPLITS : 1;
  // W v goes behind the (summed) W^T u in the workspace: needs cols + rows <= max(rows, 32 * cols) floats
  const long ws = rows > 32L * cols ? rows : 32L * cols;
  if (!do_iter || !fused || (long)cols + rows > ws) {
    const int rc = pwg_spectral_norm_forward(w_orig, u, v, sigma, w, tmp, rows, cols, do_iter, eps, stream_);
    if (rc != PWG_OK) return rc;
    if (u_saved != u) (void)hipMemcpyAsync(u_saved, u, sizeof(float) * rows, hipMemcpyDeviceToDevice, stream);
    if (v_saved != v) (void)hipMemcpyAsync(v_saved, v, sizeof(float) * cols, hipMemcpyDeviceToDevice, stream);
    PWG_CHECK_LAUNCH("spectral_norm_forward_saved");
    return PWG_OK;
  }
  const int rps = (rows + splits - 1) / splits;
  hipLaunchKernelGGL(matvec_t_kernel, dim3((cols + 63) / 64, splits), dim3(256), 0, stream, w_orig, u, tmp, rows, cols, rps);
  if (splits > 1) hipLaunchKernelGGL(sum_parts_kernel, dim3(grid_for(cols)), dim3(256), 0, stream, tmp, cols, splits);
  float* t = tmp + cols;
  hipLaunchKernelGGL(matvec_normalized_kernel, dim3(rows), dim3(256), 0, stream, w_orig, tmp, v, v_saved, t, cols, eps);
  hipLaunchKernelGGL(sn_finalize_kernel, dim3(1), dim3(256), 0, stream, t, u, u_saved, sigma, rows, eps);
  const long n = (long)rows * cols;
  hipLaunchKernelGGL(div_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, stream, w_orig, sigma, w, n);
  PWG_CHECK_LAUNCH("spectral_norm_forward_saved");
  return PWG_OK;
}

// dw_orig = dw / sigma - (<dw, w_orig> / sigma^2) u v^T ; scratch: PWG_SPECTRAL_NORM_SCRATCH_FLOATS floats
// ([0] = the dot product, [1 ..] = per-block shares)
extern "C" int pwg_spectral_norm_backward(const float* dw, const float* w_orig, const float* u, const float* v,
                                          const float* sigma, float* dw_orig, float* scratch, int32_t rows,
                                          int32_t cols, void* stream_) {
  PWG_REQUIRE(dw && w_orig && u && v && sigma && dw_orig && scratch, PWG_ERR_NULL, "spectral_norm_backward: NULL pointer");
  PWG_REQUIRE(rows > 0 && cols > 0, PWG_ERR_BAD_SHAPE, "spectral_norm: bad shape");
  hipStream_t stream = (hipStream_t)stream_;
  const long n = (long)rows * cols;
  const int nb = grid_for(n, 256, PWG_SPECTRAL_NORM_SCRATCH_FLOATS - 1);
  hipLaunchKernelGGL(dot_big_kernel, dim3(nb), dim3(256), 0, stream, dw, w_orig, scratch + 1, n);
  hipLaunchKernelGGL(spectral_norm_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dw, u, v, sigma, scratch + 1, nb,
                     dw_orig, rows, cols);
  PWG_CHECK_LAUNCH("spectral_norm_backward");
  return PWG_OK;
}

extern "C" int pwg_gate_forward(const float* z, float* out, int32_t batch, int32_t channels, int64_t t, void* stream) {
  PWG_REQUIRE(z && out, PWG_ERR_NULL, "gate_forward: NULL pointer");
  PWG_REQUIRE(batch > 0 && channels > 0 && t > 0, PWG_ERR_BAD_SHAPE, "gate: bad shape");
  const long n = (long)batch * channels * t;
  ProfScope prof((hipStream_t)stream, "gate_fwd_kernel", 0, 12.0 * n);
  LAUNCH1D(gate_fwd_kernel, n, stream, z, out, batch, channels, (long)t);
  return PWG_OK;
}

extern "C" int pwg_gate_backward(const float* z, const float* dout, float* dz, int32_t batch, int32_t channels,
                                 int64_t t, void* stream) {
  PWG_REQUIRE(z && dout && dz, PWG_ERR_NULL, "gate_backward: NULL pointer");
  PWG_REQUIRE(batch > 0 && channels > 0 && t > 0, PWG_ERR_BAD_SHAPE, "gate: bad shape");
  const long n = (long)batch * channels * t;
  LAUNCH1D(gate_bwd_kernel, n, stream, z, dout, dz, batch, channels, (long)t);
  return PWG_OK;
}

static int stretch_conv_check(int64_t rows, int32_t t_in, int32_t scale, int32_t kernel, int32_t pad_left,
                              int32_t channels, int32_t freq_kernel) {
  PWG_REQUIRE(rows > 0 && t_in > 0 && scale > 0 && kernel > 0 && kernel <= 64 && pad_left >= 0 && pad_left < kernel,
              PWG_ERR_BAD_SHAPE, "stretch_conv: bad geometry");
  PWG_REQUIRE(channels > 0 && rows % channels == 0 && freq_kernel >= 1 && freq_kernel <= 15 && (freq_kernel & 1),
              PWG_ERR_BAD_SHAPE, "stretch_conv: rows %lld not a multiple of channels %d, or freq kernel %d not odd in 1..15",
              (long long)rows, channels, freq_kernel);
  return PWG_OK;
}

extern "C" int pwg_stretch_conv_forward(const float* x, const float* w, float* y, int64_t rows, int32_t t_in,
                                        int32_t scale, int32_t kernel, int32_t pad_left, int32_t channels,
                                        int32_t freq_kernel, int32_t act, float slope, void* stream) {
  PWG_REQUIRE(x && w && y, PWG_ERR_NULL, "stretch_conv_forward: NULL pointer");
  PWG_REQUIRE(act == PWG_ACT_NONE || act == PWG_ACT_LEAKY_RELU || act == PWG_ACT_RELU || act == PWG_ACT_TANH,
              PWG_ERR_UNSUPPORTED, "stretch_conv_forward: activation %d", act);
  const int rc = stretch_conv_check(rows, t_in, scale, kernel, pad_left, channels, freq_kernel);
  if (rc != PWG_OK) return rc;
  const long n = rows * (long)t_in * scale;
  ProfScope prof((hipStream_t)stream, "stretch_conv_fwd_kernel", 2.0 * n * kernel * freq_kernel,
                 4.0 * (rows * (double)t_in + n));
  static const bool fast4 = !(getenv("PWG_STRETCH_FAST") && atoi(getenv("PWG_STRETCH_FAST")) == 0);
  if (fast4 && freq_kernel == 1 && kernel == 2 * scale + 1 && pad_left == scale && (scale == 2 || scale == 4) &&
      ((long)t_in * scale) % 4 == 0 && (((uintptr_t)y) & 15) == 0) {
    const long n4 = n / 4;
    if (scale == 4) LAUNCH1D(stretch_conv_fwd4_kernel<4>, n4, stream, x, w, y, (long)rows, t_in, act, slope);
    else LAUNCH1D(stretch_conv_fwd4_kernel<2>, n4, stream, x, w, y, (long)rows, t_in, act, slope);
    return PWG_OK;
  }
  LAUNCH1D(stretch_conv_fwd_kernel, n, stream, x, w, y, (long)rows, t_in, scale, kernel, pad_left, channels, freq_kernel,
           act, slope);
  return PWG_OK;
}

extern "C" size_t pwg_stretch_conv_backward_workspace_floats(int32_t kernel, int32_t freq_kernel) {
  return (size_t)PWG_STRETCH_CONV_WGRAD_BLOCKS * (size_t)kernel * (size_t)freq_kernel;
}

extern "C" int pwg_stretch_conv_backward(const float* dy, const float* x, const float* w, float* dx, float* dw,
                                         int64_t rows, int32_t t_in, int32_t scale, int32_t kernel, int32_t pad_left,
                                         int32_t channels, int32_t freq_kernel, float* workspace,
                                         size_t workspace_floats, void* stream) {
  PWG_REQUIRE(dy && w && (dx || dw), PWG_ERR_NULL, "stretch_conv_backward: NULL pointer");
  const int rc = stretch_conv_check(rows, t_in, scale, kernel, pad_left, channels, freq_kernel);
  if (rc != PWG_OK) return rc;
  if (dx) {
    const long n = rows * (long)t_in;
    LAUNCH1D(stretch_conv_bwd_data_kernel, n, stream, dy, w, dx, (long)rows, t_in, scale, kernel, pad_left, channels,
             freq_kernel);
  }
  if (dw) {
    PWG_REQUIRE(x, PWG_ERR_NULL, "stretch_conv_backward: x needed for dw");
    const size_t need = pwg_stretch_conv_backward_workspace_floats(kernel, freq_kernel);
    PWG_REQUIRE(workspace && workspace_floats >= need, PWG_ERR_WORKSPACE,
                "stretch_conv_backward: workspace of %zu floats needed, %zu given", need, workspace_floats);
    const long n = rows * (long)t_in * scale;
    const int nb = grid_for(n, 256, PWG_STRETCH_CONV_WGRAD_BLOCKS);
    hipLaunchKernelGGL(stretch_conv_bwd_weight_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, x, workspace,
                       (long)rows, t_in, scale, kernel, pad_left, channels, freq_kernel);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(kernel * freq_kernel), dim3(256), 0, (hipStream_t)stream, workspace, dw,
                       nb);
    PWG_CHECK_LAUNCH("stretch_conv_bwd_weight");
  }
  return PWG_OK;
}

extern "C" int pwg_add3_div(const float* a, const float* b, const float* c, float* y, int64_t n, float div,
                            void* stream) {
  PWG_REQUIRE(a && b && y, PWG_ERR_NULL, "add3_div: NULL pointer");
  PWG_REQUIRE(n > 0 && div != 0.f, PWG_ERR_BAD_SHAPE, "add3_div: bad arguments");
  ProfScope prof((hipStream_t)stream, "add3_div_kernel", 0, 16.0 * n);
  LAUNCH1D(add3_div_kernel, n, stream, a, b, c, y, (long)n, div);
  return PWG_OK;
}

extern "C" int pwg_wave_to_pcm16(const float* x, int16_t* pcm, int64_t n, void* stream) {
  PWG_REQUIRE(x && pcm, PWG_ERR_NULL, "wave_to_pcm16: NULL pointer");
  PWG_REQUIRE(n > 0, PWG_ERR_BAD_SHAPE, "wave_to_pcm16: empty");
  ProfScope prof((hipStream_t)stream, "wave_to_pcm16_kernel", 0, 6.0 * n);
  LAUNCH1D(wave_to_pcm16_kernel, n, stream, x, (short*)pcm, (long)n);
  return PWG_OK;
}

extern "C" int pwg_normalize_transpose(const float* x, const float* mean, const float* scale, float* y,
                                       int32_t batch, int32_t frames, int32_t channels, void* stream) {
  PWG_REQUIRE(x && y && ((mean == nullptr) == (scale == nullptr)), PWG_ERR_NULL, "normalize_transpose: NULL pointer");
  PWG_REQUIRE(batch > 0 && frames > 0 && channels > 0, PWG_ERR_BAD_SHAPE, "normalize_transpose: bad shape");
  const long n = (long)batch * frames * channels;
  ProfScope prof((hipStream_t)stream, "normalize_transpose_kernel", 0, 8.0 * n);
  LAUNCH1D(normalize_transpose_kernel, n, stream, x, mean, scale, y, batch, frames, channels);
  return PWG_OK;
}

extern "C" int pwg_gather_crop(const float* audio, const int64_t* audio_off, const int64_t* audio_len, const float* mel,
                               const int64_t* mel_off, const int32_t* utt, const int32_t* start, float* y, float* c,
                               int32_t batch, int32_t steps, int32_t hop, int32_t frames_ctx, int32_t acw,
                               int32_t channels, void* stream) {
  PWG_REQUIRE(audio && audio_off && audio_len && mel && mel_off && utt && start && y && c, PWG_ERR_NULL,
              "gather_crop: NULL pointer");
  PWG_REQUIRE(batch > 0 && steps > 0 && hop > 0 && frames_ctx > 0 && acw >= 0 && channels > 0, PWG_ERR_BAD_SHAPE,
              "gather_crop: bad shape");
  const long n = (long)batch * steps + (long)batch * channels * frames_ctx;
  ProfScope prof((hipStream_t)stream, "gather_crop_kernel", 0, 8.0 * n);
  LAUNCH1D(gather_crop_kernel, n, stream, audio, (const long*)audio_off, (const long*)audio_len, mel,
           (const long*)mel_off, utt, start, y, c, batch, steps, hop, frames_ctx, acw, channels);
  return PWG_OK;
}

// ---- StyleMelGAN pieces ---------------------------------------------------------------------------
extern "C" int pwg_instance_norm_forward(const float* x, float* y, float* mean, float* rstd, int64_t rows, int32_t t,
                                         float eps, void* stream) {
  PWG_REQUIRE(x && y && mean && rstd, PWG_ERR_NULL, "instance_norm_forward: NULL pointer");
  PWG_REQUIRE(rows > 0 && t > 0, PWG_ERR_BAD_SHAPE, "instance_norm: bad shape");
  ProfScope prof((hipStream_t)stream, "instance_norm_fwd_kernel", 0, 12.0 * rows * t);
  hipLaunchKernelGGL(instance_norm_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, y, mean, rstd,
                     t, eps);
  PWG_CHECK_LAUNCH("instance_norm_forward");
  return PWG_OK;
}

extern "C" int pwg_instance_norm_backward(const float* dy, const float* y, const float* rstd, float* dx, int64_t rows,
                                          int32_t t, void* stream) {
  PWG_REQUIRE(dy && y && rstd && dx, PWG_ERR_NULL, "instance_norm_backward: NULL pointer");
  PWG_REQUIRE(rows > 0 && t > 0, PWG_ERR_BAD_SHAPE, "instance_norm: bad shape");
  hipLaunchKernelGGL(instance_norm_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dy, y, rstd, dx,
                     t);
  PWG_CHECK_LAUNCH("instance_norm_backward");
  return PWG_OK;
}

extern "C" int pwg_upsample_nearest_forward(const float* x, const float* add, float* y, int64_t rows, int32_t t_in,
                                            int32_t scale, void* stream) {
  PWG_REQUIRE(x && y, PWG_ERR_NULL, "upsample_nearest_forward: NULL pointer");
  PWG_REQUIRE(rows > 0 && t_in > 0 && scale > 0, PWG_ERR_BAD_SHAPE, "upsample_nearest: bad shape");
  const long n = rows * (long)t_in * scale;
  LAUNCH1D(upsample_nearest_fwd_kernel, n, stream, x, add, y, (long)rows, t_in, scale);
  return PWG_OK;
}

extern "C" int pwg_upsample_nearest_backward(const float* dy, float* dx, int64_t rows, int32_t t_in, int32_t scale,
                                             void* stream) {
  PWG_REQUIRE(dy && dx, PWG_ERR_NULL, "upsample_nearest_backward: NULL pointer");
  PWG_REQUIRE(rows > 0 && t_in > 0 && scale > 0, PWG_ERR_BAD_SHAPE, "upsample_nearest: bad shape");
  LAUNCH1D(upsample_nearest_bwd_kernel, rows * (long)t_in, stream, dy, dx, (long)rows, t_in, scale);
  return PWG_OK;
}

extern "C" int pwg_tade_modulate_forward(const float* xn, const float* cg, float* y, int32_t batch, int32_t channels,
                                         int32_t t_in, int32_t scale, void* stream) {
  PWG_REQUIRE(xn && cg && y, PWG_ERR_NULL, "tade_modulate_forward: NULL pointer");
  PWG_REQUIRE(batch > 0 && channels > 0 && t_in > 0 && scale > 0, PWG_ERR_BAD_SHAPE, "tade_modulate: bad shape");
  const long n = (long)batch * channels * t_in * scale;
  LAUNCH1D(tade_modulate_fwd_kernel, n, stream, xn, cg, y, batch, channels, t_in, scale);
  return PWG_OK;
}

extern "C" int pwg_tade_modulate_backward(const float* dy, const float* xn, const float* cg, float* dxn, float* dcg,
                                          int32_t batch, int32_t channels, int32_t t_in, int32_t scale, void* stream) {
  PWG_REQUIRE(dy && xn && cg && (dxn || dcg), PWG_ERR_NULL, "tade_modulate_backward: NULL pointer");
  PWG_REQUIRE(batch > 0 && channels > 0 && t_in > 0 && scale > 0, PWG_ERR_BAD_SHAPE, "tade_modulate: bad shape");
  if (dcg) {
    const long n = (long)batch * channels * t_in * scale;
    LAUNCH1D(tade_modulate_bwd_cg_kernel, n, stream, dy, xn, dcg, batch, channels, t_in, scale);
  }
  if (dxn) {
    const long n = (long)batch * channels * t_in;
    LAUNCH1D(tade_modulate_bwd_x_kernel, n, stream, dy, cg, dxn, batch, channels, t_in, scale);
  }
  return PWG_OK;
}

extern "C" int pwg_softmax_gate_forward(const float* z, float* y, int32_t batch, int32_t channels, int64_t t,
                                        int32_t use_softmax, void* stream) {
  PWG_REQUIRE(z && y, PWG_ERR_NULL, "softmax_gate_forward: NULL pointer");
  PWG_REQUIRE(batch > 0 && channels > 0 && t > 0, PWG_ERR_BAD_SHAPE, "softmax_gate: bad shape");
  LAUNCH1D(softmax_gate_fwd_kernel, (long)batch * t, stream, z, y, batch, channels, (long)t, use_softmax);
  return PWG_OK;
}

extern "C" int pwg_softmax_gate_backward(const float* z, const float* dy, float* dz, int32_t batch, int32_t channels,
                                         int64_t t, int32_t use_softmax, void* stream) {
  PWG_REQUIRE(z && dy && dz, PWG_ERR_NULL, "softmax_gate_backward: NULL pointer");
  PWG_REQUIRE(batch > 0 && channels > 0 && t > 0, PWG_ERR_BAD_SHAPE, "softmax_gate: bad shape");
  LAUNCH1D(softmax_gate_bwd_kernel, (long)batch * t, stream, z, dy, dz, batch, channels, (long)t, use_softmax);
  return PWG_OK;
}

// ---- UHiFiGAN pieces ------------------------------------------------------------------------------
extern "C" int pwg_copy_channels(float* x, float* y, int32_t batch, int32_t c_src, int32_t c_dst, int32_t c_off,
                                 int64_t t, int32_t reverse, void* stream) {
  PWG_REQUIRE(x && y, PWG_ERR_NULL, "copy_channels: NULL pointer");
  PWG_REQUIRE(batch > 0 && c_src > 0 && c_off >= 0 && c_off + c_src <= c_dst && t > 0, PWG_ERR_BAD_SHAPE,
              "copy_channels: bad shape");
  LAUNCH1D(copy_channels_kernel, (long)batch * c_src * t, stream, x, y, batch, c_src, c_dst, c_off, (long)t, reverse);
  return PWG_OK;
}

extern "C" int pwg_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev,
                           void* stream) {
  PWG_REQUIRE(x && y, PWG_ERR_NULL, "dropout: NULL pointer");
  PWG_REQUIRE(n > 0 && p >= 0.f && p < 1.f, PWG_ERR_BAD_SHAPE, "dropout: bad arguments");
  LAUNCH1D(dropout_kernel, (long)n, stream, x, y, (long)n, p, 1.f / (1.f - p), (unsigned long long)seed,
           (const unsigned long long*)seed_dev);
  return PWG_OK;
}
