// stft_loss.hip -- fused single-resolution STFT loss of a (predicted, target) signal pair on gfx950:
//
//   frames (folded signal) -> windowed DFT on fp32 MFMA -> sqrt(max(re^2 + im^2, eps)) -> log ->
//   partial sums of  (|Y| - |X|)^2,  |Y|^2,  |log|Y| - log|X||
//
// replacing torch.stft + clamp/sqrt (losses/stft_loss.py:16-40), SpectralConvergenceLoss (:61) and
// LogSTFTMagnitudeLoss (:82) of the reference.  Neither the (B, 2*bins, frames) spectra nor the magnitudes
// or their logarithms exist in HBM: a wave owns a 32 bins x 32 frames tile of BOTH signals (4 accumulator
// tiles: re/im of x and y), forms the magnitudes in registers and emits three partial sums.
//
// The DFT is the K-tap convolution formulation of losses/stft.py: signals folded to (B, hop, n_cols),
// sample j = tap * hop + c of frame f is folded[c][f + tap], so for a fixed contraction index the 32
// frames of a tile are 32 consecutive floats (coalesced) and the basis image [tap][c][m] gives 32
// consecutive rows.  Operands come straight from global memory / L2 (the basis is 2.4-9.8 MB, the folded
// signals a few MB): the whole loss is < 1 % of a training step's FLOPs and nowhere near any roofline;
// what is bought here is the removal of ~20 launches and of every intermediate tensor per resolution.
//
// Backward (gradient w.r.t. the predicted signal only; the target is a constant in training): the same
// tile loop recomputes re/im of x and |Y|, applies the chain rule with the upstream gradients of the three
// sums and writes d(re), d(im) as the (B, 2*bins, frames) operand of the existing data-gradient
// convolution (transposed DFT) -- the only spectrum-shaped tensor of the whole loss.
#include "common.h"

#include <stdlib.h>

namespace pwg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct StftLossArgs {
  const float* fx;     // folded predicted signal (B, hop, n_cols)
  const float* fy;     // folded target signal
  const float* basis;  // [tap][hop][m_pad], m_pad = ngroups * 64; per 64-row group: 32 cos rows | 32 -sin rows
  int batch, hop, n_cols, taps, bins, frames, ngroups, ftiles, m_pad;
  float eps;
  float* partial;      // forward: (units, 4) partial sums [S_d, S_y, S_l, 0]
  const float* g2;     // backward: upstream gradients of (sc, mag) = sums[3..4], device
  const float* sums;   // backward: the forward's five outputs [S_d, S_y, S_l, sc, mag]
  float inv_n;         // 1 / (B * bins * frames)
  float* dspec;        // backward: (B, 2*bins, frames)
};

template <bool BWD>
__global__ __launch_bounds__(256) void stft_loss_kernel(StftLossArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int units = a.batch * a.ngroups * a.ftiles;
  int unit = blockIdx.x * 4 + wave;
  if (unit >= units) {  // (wave-uniform: no block-level synchronisation in this kernel)
    // surplus wave of the last workgroup: its slot of the partial sums is read by the finish kernel too
    if (!BWD && lane < 4) a.partial[(long)(blockIdx.x * 4 + wave) * 4 + lane] = 0.f;
    return;
  }
  const int ft = unit % a.ftiles;
  unit /= a.ftiles;
  const int g = unit % a.ngroups;
  const int b = unit / a.ngroups;

  const int col = ft * 32 + l31;  // frame of this lane's B column
  const float* bre = a.basis + g * 64 + l31;
  const long sig = (long)b * a.hop * a.n_cols;
  f32x16 rx, ix, ry, iy;
#pragma unroll
  for (int r = 0; r < 16; ++r) rx[r] = ix[r] = ry[r] = iy[r] = 0.f;

  for (int tap = 0; tap < a.taps; ++tap) {
    const int cc = col + tap;
    const bool col_ok = cc < a.n_cols;
    const float* px = a.fx + sig + (col_ok ? cc : 0);
    const float* py = a.fy + sig + (col_ok ? cc : 0);
    const float* pa = bre + (long)tap * a.hop * a.m_pad;
#pragma unroll 4
    for (int c0 = 0; c0 < a.hop; c0 += 2) {
      const int c = c0 + lhi;  // contraction index of this lane's k slot
      const bool ok = c < a.hop;
      const float are = ok ? pa[(long)c * a.m_pad] : 0.f;
      const float aim = ok ? pa[(long)c * a.m_pad + 32] : 0.f;
      const float bx = (ok && col_ok) ? px[(long)c * a.n_cols] : 0.f;
      const float by = (ok && col_ok) ? py[(long)c * a.n_cols] : 0.f;
      rx = __builtin_amdgcn_mfma_f32_32x32x2f32(are, bx, rx, 0, 0, 0);
      ix = __builtin_amdgcn_mfma_f32_32x32x2f32(aim, bx, ix, 0, 0, 0);
      ry = __builtin_amdgcn_mfma_f32_32x32x2f32(are, by, ry, 0, 0, 0);
      iy = __builtin_amdgcn_mfma_f32_32x32x2f32(aim, by, iy, 0, 0, 0);
    }
  }

  // D layout: col = lane & 31 (frame), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (bin within the group)
  const bool f_ok = col < a.frames;
  if (!BWD) {
    float sd = 0.f, sy = 0.f, sl = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bin = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (f_ok && bin < a.bins) {
        const float mx = sqrtf(fmaxf(rx[r] * rx[r] + ix[r] * ix[r], a.eps));
        const float my = sqrtf(fmaxf(ry[r] * ry[r] + iy[r] * iy[r], a.eps));
        const float d = my - mx;
        sd += d * d;
        sy += my * my;
        sl += fabsf(logf(my) - logf(mx));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {  // fixed butterfly: deterministic
      sd += __shfl_xor(sd, o, 64);
      sy += __shfl_xor(sy, o, 64);
      sl += __shfl_xor(sl, o, 64);
    }
    if (lane == 0) {
      float* p = a.partial + (long)(blockIdx.x * 4 + wave) * 4;
      p[0] = sd;
      p[1] = sy;
      p[2] = sl;
      p[3] = 0.f;
    }
  } else {
    // sc = sqrt(S_d) / sqrt(S_y), mag = S_l / n:  d sc / d S_d = 1 / (2 sqrt(S_d) sqrt(S_y)), and 0 at S_d = 0 --
    // the zero subgradient torch.norm(p="fro") returns there (x == y on every bin; sqrt's own backward would
    // hand inf * 0 = NaN to every bin)
    const float s_d = a.sums[0], s_y = a.sums[1];
    const float gd = s_d > 0.f ? a.g2[0] / (2.f * sqrtf(s_d) * sqrtf(s_y)) : 0.f;
    const float gl = a.g2[1] * a.inv_n;
    float* dre = a.dspec + ((long)b * 2 * a.bins) * a.frames + col;
    float* dim = dre + (long)a.bins * a.frames;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bin = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (f_ok && bin < a.bins) {
        const float p = rx[r] * rx[r] + ix[r] * ix[r];
        const float mx = sqrtf(fmaxf(p, a.eps));
        const float my = sqrtf(fmaxf(ry[r] * ry[r] + iy[r] * iy[r], a.eps));
        const float dl = logf(my) - logf(mx);
        // d/d|X| of  gd * (|Y| - |X|)^2  +  gl * |log|Y| - log|X||
        float dm = -2.f * gd * (my - mx);
        dm -= gl * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f)) / mx;
        // |X| = sqrt(clamp(p, eps)): the clamp passes the gradient where p >= eps
        const float s = p >= a.eps ? dm / mx : 0.f;
        dre[(long)bin * a.frames] = s * rx[r];
        dim[(long)bin * a.frames] = s * ix[r];
      }
    }
  }
}

// sums[j] = sum_u partial[u][j]  (one workgroup, fixed order); sums[3] = spectral convergence
// sqrt(S_d) / sqrt(S_y) (losses/stft_loss.py:61), sums[4] = log-magnitude L1 mean S_l / n (:82)
__global__ __launch_bounds__(256) void stft_loss_finish_kernel(const float* partial, int units, float* sums, float inv_n) {
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


// ---------------------------------------------------------------------------------------------------
// Mel-spectrogram loss (losses/mel_loss.py:95-110,150-165): the same DFT tiles, then the 32-bin slice of
// the mel filterbank contraction on MFMA straight from the magnitude registers (the D layout of |X| IS a
// valid B-operand order once the contraction index is taken as (r, lane >> 5) -> bin: the sum over bins is
// order-free), giving per bin group a partial (mels_pad x 32 frames) tile for x and y.  mel_finish_kernel adds the
// bin groups in order, clamps, takes the log and accumulates |log mel(x) - log mel(y)|.
// ---------------------------------------------------------------------------------------------------
struct MelLossArgs {
  StftLossArgs s;
  const float* mel_t;   // filterbank [bin (ngroups * 32)][mels_pad]  (mel index fastest; zero padded)
  const float* mel_b;   // filterbank [mels_pad][bin (ngroups * 32)]  (bin fastest; for the backward contraction)
  int n_mels, mels_pad; // mels_pad = 32 * ceil(n_mels / 32)
  float* pmel;          // forward: (B, ngroups, 2, mels_pad, frames) partial mels (x then y)
  const float* mel_x;   // backward: saved mel(x), mel(y) (B, n_mels, frames) BEFORE clamp / log
  const float* mel_y;
  float log_div;        // 1, ln 2 or ln 10
  const float* gout;    // backward: d loss / d (sum |log mel x - log mel y|), device scalar
};

template <bool WITH_Y>
__device__ __forceinline__ void dft_tile(const StftLossArgs& a, int b, int g, int col, int l31, int lhi, f32x16& rx,
                                         f32x16& ix, f32x16& ry, f32x16& iy) {
  const float* bre = a.basis + g * 64 + l31;
  const long sig = (long)b * a.hop * a.n_cols;
  for (int tap = 0; tap < a.taps; ++tap) {
    const int cc = col + tap;
    const bool col_ok = cc < a.n_cols;
    const float* px = a.fx + sig + (col_ok ? cc : 0);
    const float* py = a.fy + sig + (col_ok ? cc : 0);
    const float* pa = bre + (long)tap * a.hop * a.m_pad;
#pragma unroll 4
    for (int c0 = 0; c0 < a.hop; c0 += 2) {
      const int c = c0 + lhi;
      const bool ok = c < a.hop;
      const float are = ok ? pa[(long)c * a.m_pad] : 0.f;
      const float aim = ok ? pa[(long)c * a.m_pad + 32] : 0.f;
      const float bx = (ok && col_ok) ? px[(long)c * a.n_cols] : 0.f;
      rx = __builtin_amdgcn_mfma_f32_32x32x2f32(are, bx, rx, 0, 0, 0);
      ix = __builtin_amdgcn_mfma_f32_32x32x2f32(aim, bx, ix, 0, 0, 0);
      if (WITH_Y) {
        const float by = (ok && col_ok) ? py[(long)c * a.n_cols] : 0.f;
        ry = __builtin_amdgcn_mfma_f32_32x32x2f32(are, by, ry, 0, 0, 0);
        iy = __builtin_amdgcn_mfma_f32_32x32x2f32(aim, by, iy, 0, 0, 0);
      }
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void mel_loss_kernel(MelLossArgs m) {
  const StftLossArgs& a = m.s;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int units = a.batch * a.ngroups * a.ftiles;
  int unit = blockIdx.x * 4 + wave;
  if (unit >= units) return;
  const int ft = unit % a.ftiles;
  unit /= a.ftiles;
  const int g = unit % a.ngroups;
  const int b = unit / a.ngroups;
  const int col = ft * 32 + l31;
  const bool f_ok = col < a.frames;
  f32x16 rx, ix, ry, iy;
#pragma unroll
  for (int r = 0; r < 16; ++r) rx[r] = ix[r] = ry[r] = iy[r] = 0.f;
  dft_tile<!BWD>(a, b, g, col, l31, lhi, rx, ix, ry, iy);
  const int mtiles = m.mels_pad / 32;
  if (!BWD) {
    // magnitudes in place (rows of bins >= `bins` have a zero basis: sqrt(eps) there, but their filterbank rows are 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      rx[r] = sqrtf(fmaxf(rx[r] * rx[r] + ix[r] * ix[r], a.eps));
      ry[r] = sqrtf(fmaxf(ry[r] * ry[r] + iy[r] * iy[r], a.eps));
    }
    for (int mt = 0; mt < mtiles; ++mt) {
      f32x16 mx, my;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx[r] = my[r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // contraction step r: k slot `lhi` <-> bin g*32 + rowmap(r, lhi); A = filterbank[mel l31 of this tile][that bin]
        const int bin = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const float w = m.mel_t[(long)bin * m.mels_pad + mt * 32 + l31];
        mx = __builtin_amdgcn_mfma_f32_32x32x2f32(w, rx[r], mx, 0, 0, 0);
        my = __builtin_amdgcn_mfma_f32_32x32x2f32(w, ry[r], my, 0, 0, 0);
      }
      if (f_ok) {
        float* px = m.pmel + ((((long)b * a.ngroups + g) * 2 + 0) * m.mels_pad + mt * 32) * a.frames + col;
        float* py = px + (long)m.mels_pad * a.frames;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = (r & 3) + 8 * (r >> 2) + 4 * lhi;
          px[(long)j * a.frames] = mx[r];
          py[(long)j * a.frames] = my[r];
        }
      }
    }
  } else {
    // d|X|[bin][f] = sum_j filterbank[j][bin] * dmel[j][f],  dmel from the saved (pre-clamp) mels
    f32x16 dm;
#pragma unroll
    for (int r = 0; r < 16; ++r) dm[r] = 0.f;
    const float gscale = m.gout[0] / m.log_div;
    const float* mxp = m.mel_x + (long)b * m.n_mels * a.frames + (f_ok ? col : 0);
    const float* myp = m.mel_y + (long)b * m.n_mels * a.frames + (f_ok ? col : 0);
    const float* wb = m.mel_b + g * 32 + l31;  // A: row = bin l31 of this group, k = mel index
    for (int j0 = 0; j0 < m.mels_pad; j0 += 2) {
      const int j = j0 + lhi;
      float d = 0.f;
      if (f_ok && j < m.n_mels) {
        const float vx = mxp[(long)j * a.frames], vy = myp[(long)j * a.frames];
        const float lx = logf(fmaxf(vx, a.eps)), ly = logf(fmaxf(vy, a.eps));
        // d/d mel_x of |log(clamp(mel_x)) - log(clamp(mel_y))| / log_div; the clamp passes where mel_x >= eps
        if (vx >= a.eps) d = (lx > ly ? 1.f : (lx < ly ? -1.f : 0.f)) * gscale / vx;
      }
      const float w = wb[(long)j * (a.ngroups * 32)];
      dm = __builtin_amdgcn_mfma_f32_32x32x2f32(w, d, dm, 0, 0, 0);
    }
    float* dre = a.dspec + ((long)b * 2 * a.bins) * a.frames + col;
    float* dim = dre + (long)a.bins * a.frames;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bin = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (f_ok && bin < a.bins) {
        const float p = rx[r] * rx[r] + ix[r] * ix[r];
        const float mag = sqrtf(fmaxf(p, a.eps));
        const float sc = p >= a.eps ? dm[r] / mag : 0.f;
        dre[(long)bin * a.frames] = sc * rx[r];
        dim[(long)bin * a.frames] = sc * ix[r];
      }
    }
  }
}

// mel = sum over bin groups (in order) of the partial tiles; saves mel(x), mel(y) (pre-clamp) for the backward
// pass; partial[block] = sum |log(clamp(mel x)) - log(clamp(mel y))| / log_div over the block's elements
__global__ __launch_bounds__(256) void mel_finish_kernel(const float* pmel, int batch, int ngroups, int n_mels,
                                                         int mels_pad, int frames, float eps, float log_div,
                                                         float* mel_x, float* mel_y, float* partial) {
  __shared__ float red[4];
  const long n = (long)batch * n_mels * frames;
  float s = 0.f;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256L) {
    const int f = (int)(e % frames);
    long r = e / frames;
    const int j = (int)(r % n_mels);
    const int b = (int)(r / n_mels);
    float vx = 0.f, vy = 0.f;
    for (int g = 0; g < ngroups; ++g) {
      const float* p = pmel + ((((long)b * ngroups + g) * 2) * mels_pad + j) * frames + f;
      vx += p[0];
      vy += p[(long)mels_pad * frames];
    }
    mel_x[e] = vx;
    mel_y[e] = vy;
    s += fabsf(logf(fmaxf(vx, eps)) - logf(fmaxf(vy, eps))) / log_div;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sum_small_kernel(const float* partial, int n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

static int fill(StftLossArgs* a, const float* fx, const float* fy, const float* basis, int batch, int hop, int n_cols,
                int taps, int bins, int frames, float eps, const char* what) {
  PWG_REQUIRE(fx && fy && basis, PWG_ERR_NULL, "%s: NULL pointer", what);
  PWG_REQUIRE(batch > 0 && hop > 0 && taps > 0 && bins > 0 && frames > 0 && n_cols >= frames + taps - 1,
              PWG_ERR_BAD_SHAPE, "%s: bad geometry (B=%d hop=%d taps=%d bins=%d frames=%d n_cols=%d)", what, batch, hop,
              taps, bins, frames, n_cols);
  a->fx = fx;
  a->fy = fy;
  a->basis = basis;
  a->batch = batch;
  a->hop = hop;
  a->n_cols = n_cols;
  a->taps = taps;
  a->bins = bins;
  a->frames = frames;
  a->ngroups = ceil_div(bins, 32);
  a->ftiles = ceil_div(frames, 32);
  a->m_pad = a->ngroups * 64;
  a->eps = eps;
  a->partial = nullptr;
  a->g2 = nullptr;
  a->sums = nullptr;
  a->inv_n = (float)(1.0 / ((double)batch * bins * frames));
  a->dspec = nullptr;
  return PWG_OK;
}

}  // namespace pwg

using namespace pwg;

extern "C" size_t pwg_stft_loss_workspace_floats(int32_t batch, int32_t bins, int32_t frames) {
  if (batch <= 0 || bins <= 0 || frames <= 0) return 0;
  const size_t units = (size_t)batch * ceil_div(bins, 32) * ceil_div(frames, 32);
  return 4 * ((units + 3) / 4) * 4;  // whole workgroups of 4 wave tiles
}

extern "C" int pwg_stft_loss_forward(const float* fx, const float* fy, const float* basis, int32_t batch, int32_t hop,
                                     int32_t n_cols, int32_t taps, int32_t bins, int32_t frames, float eps,
                                     float* workspace, float* sums, void* stream_) {
  StftLossArgs a;
  const int rc = fill(&a, fx, fy, basis, batch, hop, n_cols, taps, bins, frames, eps, "stft_loss_forward");
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(workspace && sums, PWG_ERR_NULL, "stft_loss_forward: NULL pointer");
  a.partial = workspace;
  hipStream_t stream = (hipStream_t)stream_;
  const int units = a.batch * a.ngroups * a.ftiles;
  const int blocks = ceil_div(units, 4);
  // algorithmic: 2 signals x 2 (re, im) x bins x frames x (taps * hop) MACs; bytes: the two folded signals
  const double flops = 2.0 * 4.0 * a.ngroups * 32.0 * a.ftiles * 32.0 * batch * (double)taps * hop;
  const double bytes = 2.0 * 4.0 * (double)batch * hop * n_cols;
  // (no memset of the workspace: every wave of every workgroup writes its slot, surplus waves write zeros)
  {
    ProfScope prof(stream, "stft_loss_fwd_kernel", flops, bytes);
    hipLaunchKernelGGL(stft_loss_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(stft_loss_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, blocks * 4, sums,
                       a.inv_n);
  }
  PWG_CHECK_LAUNCH("stft_loss_forward");
  return PWG_OK;
}

extern "C" int pwg_stft_loss_backward(const float* fx, const float* fy, const float* basis, int32_t batch, int32_t hop,
                                      int32_t n_cols, int32_t taps, int32_t bins, int32_t frames, float eps,
                                      const float* sums, const float* g2, float* dspec, void* stream_) {
  StftLossArgs a;
  const int rc = fill(&a, fx, fy, basis, batch, hop, n_cols, taps, bins, frames, eps, "stft_loss_backward");
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(sums && g2 && dspec, PWG_ERR_NULL, "stft_loss_backward: NULL pointer");
  a.g2 = g2;
  a.sums = sums;
  a.dspec = dspec;
  hipStream_t stream = (hipStream_t)stream_;
  const int units = a.batch * a.ngroups * a.ftiles;
  const double flops = 2.0 * 4.0 * a.ngroups * 32.0 * a.ftiles * 32.0 * batch * (double)taps * hop;
  const double bytes = 2.0 * 4.0 * (double)batch * hop * n_cols + 4.0 * 2.0 * (double)batch * bins * frames;
  ProfScope prof(stream, "stft_loss_bwd_kernel", flops, bytes);
  hipLaunchKernelGGL(stft_loss_kernel<true>, dim3(ceil_div(units, 4)), dim3(256), 0, stream, a);
  PWG_CHECK_LAUNCH("stft_loss_backward");
  return PWG_OK;
}

// ---- mel-spectrogram loss -----------------------------------------------------------------------------
static const int MEL_FINISH_BLOCKS = 128;

extern "C" size_t pwg_mel_loss_workspace_floats(int32_t batch, int32_t bins, int32_t frames, int32_t n_mels) {
  if (batch <= 0 || bins <= 0 || frames <= 0 || n_mels <= 0) return 0;
  const size_t mels_pad = 32 * (size_t)ceil_div(n_mels, 32);
  return (size_t)batch * ceil_div(bins, 32) * 2 * mels_pad * frames + MEL_FINISH_BLOCKS;
}

extern "C" int pwg_mel_loss_forward(const float* fx, const float* fy, const float* basis, const float* mel_t,
                                    int32_t batch, int32_t hop, int32_t n_cols, int32_t taps, int32_t bins,
                                    int32_t frames, int32_t n_mels, float eps, float log_div, float* workspace,
                                    float* mel_x, float* mel_y, float* sum, void* stream_) {
  MelLossArgs m;
  const int rc = fill(&m.s, fx, fy, basis, batch, hop, n_cols, taps, bins, frames, eps, "mel_loss_forward");
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(mel_t && workspace && mel_x && mel_y && sum, PWG_ERR_NULL, "mel_loss_forward: NULL pointer");
  PWG_REQUIRE(n_mels > 0 && log_div > 0.f, PWG_ERR_BAD_SHAPE, "mel_loss_forward: bad arguments");
  m.mel_t = mel_t;
  m.mel_b = nullptr;
  m.n_mels = n_mels;
  m.mels_pad = 32 * ceil_div(n_mels, 32);
  m.pmel = workspace;
  m.mel_x = m.mel_y = nullptr;
  m.log_div = log_div;
  m.gout = nullptr;
  hipStream_t stream = (hipStream_t)stream_;
  const int units = m.s.batch * m.s.ngroups * m.s.ftiles;
  const size_t pmel_floats = (size_t)batch * m.s.ngroups * 2 * m.mels_pad * frames;
  float* partial = workspace + pmel_floats;
  const double flops = 2.0 * 4.0 * m.s.ngroups * 32.0 * m.s.ftiles * 32.0 * batch * (double)taps * hop;
  ProfScope prof(stream, "mel_loss_fwd_kernel", flops, 2.0 * 4.0 * (double)batch * hop * n_cols);
  hipLaunchKernelGGL(mel_loss_kernel<false>, dim3(ceil_div(units, 4)), dim3(256), 0, stream, m);
  hipLaunchKernelGGL(mel_finish_kernel, dim3(MEL_FINISH_BLOCKS), dim3(256), 0, stream, (const float*)workspace, batch,
                     m.s.ngroups, n_mels, m.mels_pad, frames, eps, log_div, mel_x, mel_y, partial);
  hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, MEL_FINISH_BLOCKS, sum);
  PWG_CHECK_LAUNCH("mel_loss_forward");
  return PWG_OK;
}

extern "C" int pwg_mel_loss_backward(const float* fx, const float* basis, const float* mel_b, const float* mel_x,
                                     const float* mel_y, int32_t batch, int32_t hop, int32_t n_cols, int32_t taps,
                                     int32_t bins, int32_t frames, int32_t n_mels, float eps, float log_div,
                                     const float* gout, float* dspec, void* stream_) {
  MelLossArgs m;
  const int rc = fill(&m.s, fx, fx, basis, batch, hop, n_cols, taps, bins, frames, eps, "mel_loss_backward");
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(mel_b && mel_x && mel_y && gout && dspec, PWG_ERR_NULL, "mel_loss_backward: NULL pointer");
  PWG_REQUIRE(n_mels > 0 && log_div > 0.f, PWG_ERR_BAD_SHAPE, "mel_loss_backward: bad arguments");
  m.s.dspec = dspec;
  m.mel_t = nullptr;
  m.mel_b = mel_b;
  m.n_mels = n_mels;
  m.mels_pad = 32 * ceil_div(n_mels, 32);
  m.pmel = nullptr;
  m.mel_x = mel_x;
  m.mel_y = mel_y;
  m.log_div = log_div;
  m.gout = gout;
  hipStream_t stream = (hipStream_t)stream_;
  const int units = m.s.batch * m.s.ngroups * m.s.ftiles;
  const double flops = 2.0 * 2.0 * m.s.ngroups * 32.0 * m.s.ftiles * 32.0 * batch * (double)taps * hop;
  ProfScope prof(stream, "mel_loss_bwd_kernel", flops, 4.0 * ((double)batch * hop * n_cols + 2.0 * batch * bins * frames));
  hipLaunchKernelGGL(mel_loss_kernel<true>, dim3(ceil_div(units, 4)), dim3(256), 0, stream, m);
  PWG_CHECK_LAUNCH("mel_loss_backward");
  return PWG_OK;
}
