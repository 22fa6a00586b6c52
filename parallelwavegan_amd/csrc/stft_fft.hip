// stft_fft.hip -- single-resolution STFT loss of a (predicted, target) pair through a radix-2 FFT in LDS (round 4).
//
// Replaces torch.stft + clamp / sqrt (reference losses/stft_loss.py:16-40), SpectralConvergenceLoss (:61) and
// LogSTFTMagnitudeLoss (:82) for the power-of-two FFT sizes of MultiResolutionSTFTLoss (512 / 1024 / 2048); the
// sub-band sizes 171 / 384 / 683 keep the dense DFT-on-MFMA kernel of stft_loss.hip.  The dense form spends
// 2 * 2 * bins * win MACs per frame and signal (n_fft = 2048, win = 1200: 4.9 MFLOP) where the FFT needs
// 5 * N * log2 N = 0.11 MFLOP: the multi-band MelGAN step spent 4.1 ms in the dense kernels.
//
// One workgroup transforms one frame of both signals at a time: each signal's windowed frame xw[n] (window zero-padded to
// n_fft, centred, as torch.stft pads it; frames cut from the reflect-padded signal, center=True) is packed as the
// complex sequence z[m] = xw[2m] + i xw[2m+1] of n_fft / 2 points, both sequences run through the SAME Stockham autosort
// radix-2 passes between two LDS buffers (natural order out, no bit reversal; twiddles from a table computed in float64
// on the host), and X[k] = E[k] + W^k O[k] with E / O = (Z[k] +- conj Z[M-k]) / 2 (/ i) for k = 0 .. N/2 -- the same
// instruction sequence for x and y, so identical signals give bit-identical spectra and the loss of (x, x) is exactly 0,
// as with torch.stft.  Then magnitudes with the reference's clamp, logs, and the three running sums in registers.  No frame, spectrum, magnitude or
// log tensor exists in HBM.  Deterministic: fixed frame -> workgroup and bin -> thread assignment, fixed reduction trees.
//
// Backward (w.r.t. the predicted signal): the same transform is recomputed, G[k] = dL/dRe X[k] + i dL/dIm X[k] formed
// per bin (chain rule as stft_loss.hip), and the adjoint of the real DFT  d xw[n] = Re sum_{k <= N/2} G[k] e^{+2 pi i k n / N}
// is the SAME FFT applied to conj(G) (zero above N/2).  The windowed frame gradients go to HBM once, (B, frames, win),
// and stft_fft_gather_kernel overlap-adds them -- a gather per output sample over its <= ceil(win / hop) frames and the
// <= 2 reflected images of the padding, in a fixed order: no atomics.
#include "common.h"

namespace pwg {

struct StftFftArgs {
  const float* x;         // predicted (B, T)
  const float* y;         // target (B, T)
  const float* window;    // win floats
  const float2* twiddle;  // n_fft / 2 entries (cos, -sin)(2 pi t / n_fft)
  int batch, t, hop, win, off, frames;
  float eps;
  float* partial;         // forward: (gridDim.x, 4) partial sums [S_d, S_y, S_l, 0]
  const float* sums;      // backward: the forward's [S_d, S_y, S_l, sc, mag]
  const float* g2;        // backward: upstream gradients of (sc, mag), device
  float inv_n;            // 1 / (B * bins * frames)
  float* dframes;         // backward: (B, frames, win) windowed frame gradients
};

// Stockham autosort radix-2, decimation in frequency, of NSIG independent sequences of M = 2^LOGM points stored back to
// back: log2 M passes src -> dst.  Pass with stride s: butterfly i = p * s + q of a sequence reads src[i], src[i + M/2] and
// writes dst[2 p s + q], dst[2 p s + q + s] with twiddle W_M^(p s) = tw[TWS * p s] (tw = the n_fft-point table).
// Returns the buffer holding the (naturally ordered) results.
template <int LOGM, int NSIG, int TWS>
__device__ __forceinline__ float2* fft_lds(float2* src, float2* dst, const float2* __restrict__ tw, int tid) {
  constexpr int M = 1 << LOGM, H = M / 2;
#pragma unroll 1
  for (int s = 1; s < M; s <<= 1) {
    for (int i = tid; i < NSIG * H; i += 256) {
      const int sig = i >> (LOGM - 1), ii = i & (H - 1);
      const int q = ii & (s - 1), ps = ii - q;
      const float2* sp = src + sig * M;
      float2* dp = dst + sig * M;
      const float2 a = sp[ii], b = sp[ii + H], w = tw[TWS * ps];
      const float dr = a.x - b.x, di = a.y - b.y;
      dp[2 * ps + q] = make_float2(a.x + b.x, a.y + b.y);
      dp[2 * ps + q + s] = make_float2(dr * w.x - di * w.y, dr * w.y + di * w.x);
    }
    __syncthreads();
    float2* t = src;
    src = dst;
    dst = t;
  }
  return src;
}

// buf[0 .. N/2) = even / odd packing of the windowed frame f of x, buf[N/2 .. N) = the same of y
template <int LOGN>
__device__ __forceinline__ void load_frame_pair(float2* buf, const float* __restrict__ xb, const float* __restrict__ yb,
                                                const float* __restrict__ window, int f, int hop, int win, int off,
                                                int t_len, int tid) {
  constexpr int N = 1 << LOGN, H = N / 2;
  for (int m = tid; m < N; m += 256) {
    const float* sb = m < H ? xb : yb;
    const int mm = m & (H - 1);
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int n = 2 * mm + e, j = n - off;
      v[e] = 0.f;
      if (j >= 0 && j < win) {
        int t = f * hop + n - H;  // center = True: the signal is reflect-padded by n_fft / 2
        if (t < 0) t = -t;
        if (t >= t_len) t = 2 * (t_len - 1) - t;
        v[e] = window[j] * sb[t];
      }
    }
    buf[m] = make_float2(v[0], v[1]);
  }
}

// X[k] of a real frame from the M-point transform Z of its even / odd packing, k = 0 .. M (W = W_N^k, (-1, 0) at k = M)
__device__ __forceinline__ float2 rfft_bin(const float2* __restrict__ z, int k, int M, float2 w) {
  const float2 zk = z[k & (M - 1)], zn = z[(M - k) & (M - 1)];
  const float er = 0.5f * (zk.x + zn.x), ei = 0.5f * (zk.y - zn.y);
  const float orr = 0.5f * (zk.y + zn.y), oi = -0.5f * (zk.x - zn.x);
  return make_float2(er + (w.x * orr - w.y * oi), ei + (w.x * oi + w.y * orr));
}

template <int LOGN, bool BWD>
__global__ __launch_bounds__(256) void stft_fft_kernel(StftFftArgs a) {
  constexpr int N = 1 << LOGN, H = N / 2;
  extern __shared__ float2 sm2[];
  float2* buf0 = sm2;
  float2* buf1 = sm2 + N;
  float2* tw = sm2 + 2 * N;
  __shared__ float red[3][4];
  const int tid = threadIdx.x;
  for (int i = tid; i < H; i += 256) tw[i] = a.twiddle[i];
  float sd = 0.f, sy = 0.f, sl = 0.f;
  float gd = 0.f, gl = 0.f;
  if (BWD) {
    // sc = sqrt(S_d) / sqrt(S_y), mag = S_l / n; zero subgradient at S_d = 0 as torch.norm (see stft_loss.hip)
    const float s_d = a.sums[0], s_y = a.sums[1];
    gd = s_d > 0.f ? a.g2[0] / (2.f * sqrtf(s_d) * sqrtf(s_y)) : 0.f;
    gl = a.g2[1] * a.inv_n;
  }
  const int units = a.batch * a.frames;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int b = u / a.frames, f = u - b * a.frames;
    const float* xb = a.x + (long)b * a.t;
    const float* yb = a.y + (long)b * a.t;
    __syncthreads();  // the previous frame's readers are done (first time: the twiddles are staged)
    load_frame_pair<LOGN>(buf0, xb, yb, a.window, f, a.hop, a.win, a.off, a.t, tid);
    __syncthreads();
    float2* r = fft_lds<LOGN - 1, 2, 2>(buf0, buf1, tw, tid);
    float2* o = (r == buf0) ? buf1 : buf0;
    for (int k = tid; k <= H; k += 256) {
      const float2 w = k < H ? tw[k] : make_float2(-1.f, 0.f);
      const float2 xk = rfft_bin(r, k, H, w), yk = rfft_bin(r + H, k, H, w);
      const float xr = xk.x, xi = xk.y, yr = yk.x, yi = yk.y;
      const float p = xr * xr + xi * xi;
      const float mx = sqrtf(fmaxf(p, a.eps));
      const float my = sqrtf(fmaxf(yr * yr + yi * yi, a.eps));
      if (!BWD) {
        const float d = my - mx;
        sd += d * d;
        sy += my * my;
        sl += fabsf(logf(my) - logf(mx));
      } else {
        const float dl = logf(my) - logf(mx);
        // d/d|X| of  gd * (|Y| - |X|)^2  +  gl * |log|Y| - log|X||
        float dm = -2.f * gd * (my - mx);
        dm -= gl * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f)) / mx;
        // |X| = sqrt(clamp(p, eps)): the clamp passes the gradient where p >= eps
        const float s = p >= a.eps ? dm / mx : 0.f;
        o[k] = make_float2(s * xr, -s * xi);  // conj(G[k])
      }
    }
    if (BWD) {
      for (int k = H + 1 + tid; k < N; k += 256) o[k] = make_float2(0.f, 0.f);
      __syncthreads();
      float2* g = fft_lds<LOGN, 1, 1>(o, r, tw, tid);  // conj of the inverse-direction transform: same real part
      float* df = a.dframes + (long)u * a.win;
      for (int j = tid; j < a.win; j += 256) df[j] = a.window[j] * g[j + a.off].x;
    }
  }
  if (!BWD) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sd += __shfl_down(sd, o, 64);
      sy += __shfl_down(sy, o, 64);
      sl += __shfl_down(sl, o, 64);
    }
    __syncthreads();
    if ((tid & 63) == 0) {
      red[0][tid >> 6] = sd;
      red[1][tid >> 6] = sy;
      red[2][tid >> 6] = sl;
    }
    __syncthreads();
    if (tid == 0) {
      float* p = a.partial + (long)blockIdx.x * 4;
      p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      p[2] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
      p[3] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Mel-spectrogram loss (reference losses/mel_loss.py:95-110, :150-165) on the same transform: |X|, |Y| of a frame go to
// LDS, threads 0 .. 2 n_mels - 1 contract them with their filter -- only over the filter's support (Slaney triangles:
// 3 .. 60 of the 513 / 1025 bins; the bins outside hold exact zeros in the dense matrix) -- clamp, log and accumulate
// |log mel x - log mel y| / log_div.  Backward: the same quantities are recomputed, d mel_x goes to LDS, each bin gathers
// d|X|[k] = sum_j fb[j][k] d mel_x[j] over the few filters that cover it, and G = d|X| / |X| * X returns through the same
// inverse-direction FFT, windowing and gather as the STFT loss.  No spectrum, magnitude or mel tensor in HBM.
// ---------------------------------------------------------------------------------------------------------------
struct MelFftArgs {
  StftFftArgs s;
  const float* fb;       // filterbank [mel][bins_pad] (bin fastest)
  const int* mel_range;  // per mel: first / last bin of its support (last < first: empty)
  const int* bin_range;  // per bin: first / last mel whose support contains it
  int n_mels, bins_pad;
  float log_div;
  const float* gout;     // backward: d loss / d (sum |log mel x - log mel y|), device scalar
};
constexpr int MEL_FFT_MAX_MELS = 128;

template <int LOGN, bool BWD>
__global__ __launch_bounds__(256) void mel_fft_kernel(MelFftArgs m) {
  const StftFftArgs& a = m.s;
  constexpr int N = 1 << LOGN, H = N / 2;
  extern __shared__ float2 sm2[];
  float2* buf0 = sm2;
  float2* buf1 = sm2 + N;
  float2* tw = sm2 + 2 * N;
  __shared__ float red[4];
  __shared__ float dmel[MEL_FFT_MAX_MELS];
  const int tid = threadIdx.x;
  for (int i = tid; i < H; i += 256) tw[i] = a.twiddle[i];
  float acc = 0.f;
  const float gscale = BWD ? m.gout[0] / m.log_div : 0.f;
  const int units = a.batch * a.frames;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int b = u / a.frames, f = u - b * a.frames;
    __syncthreads();  // the previous frame's readers are done (first time: the twiddles are staged)
    load_frame_pair<LOGN>(buf0, a.x + (long)b * a.t, a.y + (long)b * a.t, a.window, f, a.hop, a.win, a.off, a.t, tid);
    __syncthreads();
    float2* r = fft_lds<LOGN - 1, 2, 2>(buf0, buf1, tw, tid);
    float2* o = (r == buf0) ? buf1 : buf0;
    float* mag = reinterpret_cast<float*>(o);  // [0 .. H] = |X|, [H + 1 .. 2 H + 1] = |Y|  (2 N floats available)
    for (int k = tid; k <= H; k += 256) {
      const float2 w = k < H ? tw[k] : make_float2(-1.f, 0.f);
      const float2 xk = rfft_bin(r, k, H, w), yk = rfft_bin(r + H, k, H, w);
      mag[k] = sqrtf(fmaxf(xk.x * xk.x + xk.y * xk.y, a.eps));
      mag[H + 1 + k] = sqrtf(fmaxf(yk.x * yk.x + yk.y * yk.y, a.eps));
    }
    __syncthreads();
    float mel_v = 0.f;
    if (tid < 2 * m.n_mels) {
      const int sig = tid >= m.n_mels, j = tid - sig * m.n_mels;
      const int lo = m.mel_range[2 * j], hi = m.mel_range[2 * j + 1];
      const float* fr = m.fb + (long)j * m.bins_pad;
      const float* mg = mag + sig * (H + 1);
      for (int k = lo; k <= hi; ++k) mel_v += fr[k] * mg[k];
    }
    // pair the two signals' mels of filter j: thread j (x) needs thread n_mels + j's value (y) -> through LDS
    __syncthreads();  // (all reads of mag are done: the scratch is reused below)
    if (tid < 2 * m.n_mels) mag[tid] = mel_v;
    __syncthreads();
    if (tid < m.n_mels) {
      const float vx = mag[tid], vy = mag[m.n_mels + tid];
      const float lx = logf(fmaxf(vx, a.eps)), ly = logf(fmaxf(vy, a.eps));
      if (!BWD) {
        acc += fabsf(lx - ly) / m.log_div;
      } else {
        // d/d mel_x of |log(clamp(mel_x)) - log(clamp(mel_y))| / log_div; the clamp passes where mel_x >= eps
        dmel[tid] = vx >= a.eps ? (lx > ly ? 1.f : (lx < ly ? -1.f : 0.f)) * gscale / vx : 0.f;
      }
    }
    if (BWD) {
      __syncthreads();
      for (int k = tid; k <= H; k += 256) {
        const int jlo = m.bin_range[2 * k], jhi = m.bin_range[2 * k + 1];
        float dm = 0.f;
        for (int j = jlo; j <= jhi; ++j) dm += m.fb[(long)j * m.bins_pad + k] * dmel[j];
        const float2 w = k < H ? tw[k] : make_float2(-1.f, 0.f);
        const float2 xk = rfft_bin(r, k, H, w);
        const float p = xk.x * xk.x + xk.y * xk.y;
        const float sc = p >= a.eps ? dm / sqrtf(fmaxf(p, a.eps)) : 0.f;
        o[k] = make_float2(sc * xk.x, -sc * xk.y);  // conj(G[k])
      }
      for (int k = H + 1 + tid; k < N; k += 256) o[k] = make_float2(0.f, 0.f);
      __syncthreads();
      float2* g = fft_lds<LOGN, 1, 1>(o, r, tw, tid);
      float* df = a.dframes + (long)u * a.win;
      for (int j = tid; j < a.win; j += 256) df[j] = a.window[j] * g[j + a.off].x;
    }
  }
  if (!BWD) {
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) a.partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// total = sum over workgroups (fixed order) of partial[.]
__global__ __launch_bounds__(256) void mel_fft_finish_kernel(const float* partial, int units, float* total) {
  __shared__ float red[4];
  float s = 0.f;
  for (int u = threadIdx.x; u < units; u += 256) s += partial[u];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) total[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// sums[j] = sum over workgroups (fixed order) of partial[.][j]; sums[3] = sqrt(S_d) / sqrt(S_y), sums[4] = S_l / n
__global__ __launch_bounds__(256) void stft_fft_finish_kernel(const float* partial, int units, float* sums, float inv_n) {
  __shared__ float red[3][4];
  float s[3] = {0.f, 0.f, 0.f};
  for (int u = threadIdx.x; u < units; u += 256) {
    s[0] += partial[(long)u * 4 + 0];
    s[1] += partial[(long)u * 4 + 1];
    s[2] += partial[(long)u * 4 + 2];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_down(s[j], o, 64);
    if ((threadIdx.x & 63) == 0) red[j][threadIdx.x >> 6] = s[j];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) t[j] = sums[j] = (red[j][0] + red[j][1]) + (red[j][2] + red[j][3]);
    sums[3] = sqrtf(t[0]) / sqrtf(t[1]);
    sums[4] = t[2] * inv_n;
  }
}

// dx[b][t] = sum over the padded positions p that the reflect padding maps to t (left image, centre, right image, in
// that order) and over the frames f whose window covers p (ascending) of dframes[b][f][p - f * hop - off]
__global__ __launch_bounds__(256) void stft_fft_gather_kernel(const float* __restrict__ dframes, float* __restrict__ dx,
                                                              int batch, int t_len, int n_fft, int hop, int win, int off,
                                                              int frames) {
  const long total = (long)batch * t_len;
  const int half = n_fft / 2;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int b = (int)(e / t_len), t = (int)(e - (long)b * t_len);
    const float* df = dframes + (long)b * frames * win;
    int cand[3];
    cand[0] = (t >= 1 && t <= half) ? half - t : -1;
    cand[1] = t + half;
    cand[2] = (t <= t_len - 2 && t >= t_len - 1 - half) ? half + 2 * (t_len - 1) - t : -1;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int p = cand[c];
      if (p < 0) continue;
      const int hi = p - off;            // n = p - f*hop in [off, off + win)  <=>  0 <= hi - f*hop < win
      if (hi < 0) continue;
      int f_hi = hi / hop;
      int f_lo = hi - win + 1 <= 0 ? 0 : (hi - win + 1 + hop - 1) / hop;
      if (f_hi > frames - 1) f_hi = frames - 1;
      for (int f = f_lo; f <= f_hi; ++f) acc += df[(long)f * win + (hi - f * hop)];
    }
    dx[e] = acc;
  }
}

static int log2_exact(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return (1 << l) == n ? l : -1;
}

static int fft_grid(int units, int n_fft) {
  const int per_cu = n_fft >= 2048 ? 4 : 8;  // LDS: (2 N + N / 2) float2 = 40 KB at N = 2048
  const int cap = 256 * per_cu;
  return units < cap ? units : cap;
}

}  // namespace pwg

using namespace pwg;

extern "C" int pwg_stft_fft_supported(int32_t n_fft, int32_t win, int32_t hop) {
  const int l = log2_exact(n_fft);
  return l >= 8 && l <= 11 && win > 0 && win <= n_fft && hop > 0;
}

extern "C" size_t pwg_stft_fft_workspace_floats(int32_t batch, int32_t frames, int32_t n_fft) {
  if (batch <= 0 || frames <= 0) return 0;
  return (size_t)fft_grid(batch * frames, n_fft) * 4;
}

static int fill_fft_args(StftFftArgs* a, const float* x, const float* y, const float* window, const float* twiddle,
                         int batch, int t, int n_fft, int hop, int win, float eps) {
  PWG_REQUIRE(x && y && window && twiddle, PWG_ERR_NULL, "stft_fft: NULL pointer");
  PWG_REQUIRE(pwg_stft_fft_supported(n_fft, win, hop), PWG_ERR_UNSUPPORTED,
              "stft_fft: n_fft %d (power of two in 256 .. 2048), win %d, hop %d unsupported", n_fft, win, hop);
  PWG_REQUIRE(batch > 0 && t > n_fft / 2, PWG_ERR_BAD_SHAPE,
              "stft_fft: reflect padding needs T > n_fft / 2 (B=%d T=%d n_fft=%d)", batch, t, n_fft);
  a->x = x;
  a->y = y;
  a->window = window;
  a->twiddle = reinterpret_cast<const float2*>(twiddle);
  a->batch = batch;
  a->t = t;
  a->hop = hop;
  a->win = win;
  a->off = (n_fft - win) / 2;
  a->frames = 1 + t / hop;  // torch.stft(center=True): 1 + (T + 2 (n_fft / 2) - n_fft) / hop
  a->eps = eps;
  a->partial = nullptr;
  a->sums = nullptr;
  a->g2 = nullptr;
  a->dframes = nullptr;
  a->inv_n = 1.f / ((float)batch * (float)(n_fft / 2 + 1) * (float)a->frames);
  return PWG_OK;
}

template <bool BWD>
static int launch_fft(const StftFftArgs& a, int n_fft, hipStream_t stream) {
  const int grid = fft_grid(a.batch * a.frames, n_fft);
  const size_t lds = (size_t)(2 * n_fft + n_fft / 2) * sizeof(float2);
  void (*kern)(StftFftArgs) = nullptr;
  switch (log2_exact(n_fft)) {
    case 8: kern = stft_fft_kernel<8, BWD>; break;
    case 9: kern = stft_fft_kernel<9, BWD>; break;
    case 10: kern = stft_fft_kernel<10, BWD>; break;
    default: kern = stft_fft_kernel<11, BWD>; break;
  }
  const double units = (double)a.batch * a.frames;
  const double nlog = 5.0 * n_fft * log2_exact(n_fft);
  ProfScope prof(stream, BWD ? "stft_fft_bwd_kernel" : "stft_fft_fwd_kernel", units * nlog * (BWD ? 2.0 : 1.0),
                 4.0 * (2.0 * a.batch * a.t + (BWD ? units * a.win : 0.0)));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  PWG_CHECK_LAUNCH(BWD ? "stft_fft_backward" : "stft_fft_forward");
  return PWG_OK;
}

extern "C" int pwg_stft_fft_loss_forward(const float* x, const float* y, const float* window, const float* twiddle,
                                         int32_t batch, int32_t t, int32_t n_fft, int32_t hop, int32_t win, float eps,
                                         float* workspace, float* sums, void* stream_) {
  StftFftArgs a;
  int rc = fill_fft_args(&a, x, y, window, twiddle, batch, t, n_fft, hop, win, eps);
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(workspace && sums, PWG_ERR_NULL, "stft_fft_loss_forward: NULL workspace / sums");
  hipStream_t stream = (hipStream_t)stream_;
  a.partial = workspace;
  rc = launch_fft<false>(a, n_fft, stream);
  if (rc != PWG_OK) return rc;
  hipLaunchKernelGGL(stft_fft_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace,
                     fft_grid(a.batch * a.frames, n_fft), sums, a.inv_n);
  PWG_CHECK_LAUNCH("stft_fft_finish");
  return PWG_OK;
}

extern "C" int pwg_stft_fft_loss_backward(const float* x, const float* y, const float* window, const float* twiddle,
                                          int32_t batch, int32_t t, int32_t n_fft, int32_t hop, int32_t win, float eps,
                                          const float* sums, const float* g2, float* dframes, float* dx, void* stream_) {
  StftFftArgs a;
  int rc = fill_fft_args(&a, x, y, window, twiddle, batch, t, n_fft, hop, win, eps);
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(sums && g2 && dframes && dx, PWG_ERR_NULL, "stft_fft_loss_backward: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  a.sums = sums;
  a.g2 = g2;
  a.dframes = dframes;
  rc = launch_fft<true>(a, n_fft, stream);
  if (rc != PWG_OK) return rc;
  const long total = (long)batch * t;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  ProfScope prof(stream, "stft_fft_gather_kernel", 0, 4.0 * ((double)batch * a.frames * win + (double)total));
  hipLaunchKernelGGL(stft_fft_gather_kernel, dim3((int)blocks), dim3(256), 0, stream, (const float*)dframes, dx, batch, t,
                     n_fft, hop, win, a.off, a.frames);
  PWG_CHECK_LAUNCH("stft_fft_gather");
  return PWG_OK;
}


// ---- mel loss through the FFT
static int fill_mel_fft(MelFftArgs* m, const float* x, const float* y, const float* window, const float* twiddle,
                        const float* fb, const int32_t* mel_range, const int32_t* bin_range, int batch, int t, int n_fft,
                        int hop, int win, int n_mels, int bins_pad, float eps, float log_div) {
  const int rc = fill_fft_args(&m->s, x, y, window, twiddle, batch, t, n_fft, hop, win, eps);
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(fb && mel_range && bin_range, PWG_ERR_NULL, "mel_fft: NULL filterbank / range table");
  PWG_REQUIRE(n_mels > 0 && n_mels <= MEL_FFT_MAX_MELS && bins_pad >= n_fft / 2 + 1 && log_div > 0.f, PWG_ERR_UNSUPPORTED,
              "mel_fft: n_mels %d (<= %d), bins_pad %d, log_div %g", n_mels, MEL_FFT_MAX_MELS, bins_pad, (double)log_div);
  m->fb = fb;
  m->mel_range = mel_range;
  m->bin_range = bin_range;
  m->n_mels = n_mels;
  m->bins_pad = bins_pad;
  m->log_div = log_div;
  m->gout = nullptr;
  return PWG_OK;
}

template <bool BWD>
static int launch_mel_fft(const MelFftArgs& m, int n_fft, hipStream_t stream) {
  const StftFftArgs& a = m.s;
  const int grid = fft_grid(a.batch * a.frames, n_fft);
  const size_t lds = (size_t)(2 * n_fft + n_fft / 2) * sizeof(float2);
  void (*kern)(MelFftArgs) = nullptr;
  switch (log2_exact(n_fft)) {
    case 8: kern = mel_fft_kernel<8, BWD>; break;
    case 9: kern = mel_fft_kernel<9, BWD>; break;
    case 10: kern = mel_fft_kernel<10, BWD>; break;
    default: kern = mel_fft_kernel<11, BWD>; break;
  }
  const double units = (double)a.batch * a.frames;
  ProfScope prof(stream, BWD ? "mel_fft_bwd_kernel" : "mel_fft_fwd_kernel",
                 units * 5.0 * n_fft * log2_exact(n_fft) * (BWD ? 2.0 : 1.0),
                 4.0 * (2.0 * a.batch * a.t + (BWD ? units * a.win : 0.0)));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, m);
  PWG_CHECK_LAUNCH(BWD ? "mel_fft_backward" : "mel_fft_forward");
  return PWG_OK;
}

extern "C" int pwg_mel_fft_loss_forward(const float* x, const float* y, const float* window, const float* twiddle,
                                        const float* fb, const int32_t* mel_range, const int32_t* bin_range, int32_t batch,
                                        int32_t t, int32_t n_fft, int32_t hop, int32_t win, int32_t n_mels, int32_t bins_pad,
                                        float eps, float log_div, float* workspace, float* total, void* stream_) {
  MelFftArgs m;
  int rc = fill_mel_fft(&m, x, y, window, twiddle, fb, mel_range, bin_range, batch, t, n_fft, hop, win, n_mels, bins_pad,
                        eps, log_div);
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(workspace && total, PWG_ERR_NULL, "mel_fft_loss_forward: NULL workspace / total");
  hipStream_t stream = (hipStream_t)stream_;
  m.s.partial = workspace;
  rc = launch_mel_fft<false>(m, n_fft, stream);
  if (rc != PWG_OK) return rc;
  hipLaunchKernelGGL(mel_fft_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace,
                     fft_grid(m.s.batch * m.s.frames, n_fft), total);
  PWG_CHECK_LAUNCH("mel_fft_finish");
  return PWG_OK;
}

extern "C" int pwg_mel_fft_loss_backward(const float* x, const float* y, const float* window, const float* twiddle,
                                         const float* fb, const int32_t* mel_range, const int32_t* bin_range, int32_t batch,
                                         int32_t t, int32_t n_fft, int32_t hop, int32_t win, int32_t n_mels, int32_t bins_pad,
                                         float eps, float log_div, const float* gout, float* dframes, float* dx,
                                         void* stream_) {
  MelFftArgs m;
  int rc = fill_mel_fft(&m, x, y, window, twiddle, fb, mel_range, bin_range, batch, t, n_fft, hop, win, n_mels, bins_pad,
                        eps, log_div);
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(gout && dframes && dx, PWG_ERR_NULL, "mel_fft_loss_backward: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  m.gout = gout;
  m.s.dframes = dframes;
  rc = launch_mel_fft<true>(m, n_fft, stream);
  if (rc != PWG_OK) return rc;
  const long total = (long)batch * t;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  ProfScope prof(stream, "stft_fft_gather_kernel", 0, 4.0 * ((double)batch * m.s.frames * win + (double)total));
  hipLaunchKernelGGL(stft_fft_gather_kernel, dim3((int)blocks), dim3(256), 0, stream, (const float*)dframes, dx, batch, t,
                     n_fft, hop, win, m.s.off, m.s.frames);
  PWG_CHECK_LAUNCH("stft_fft_gather");
  return PWG_OK;
}
