/*
 * pwg_oracle.c -- plain-C restatement of the ATen primitives the hot path bottoms out in.
 *
 * TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference has no native code; at the
 * call sites cited below it calls torch (third-party, torch>=1.8, setup.py:27).  These loops
 * restate the published semantics of those ops (PyTorch docs: Conv1d / ConvTranspose1d /
 * AvgPool1d / LeakyReLU / stft) in double-accumulated C so that the torch-level oracle and
 * the HIP kernels can be checked independently of any torch build.
 *
 *   conv1d            F.conv1d            layers/residual_block.py:190-220, models/hifigan.py:75-81
 *   conv_transpose1d  F.conv_transpose1d  models/hifigan.py:99-107, layers/pqmf.py:146
 *   avg_pool1d        F.avg_pool1d        models/hifigan.py:773-775
 *   stft_mag          torch.stft + sqrt(clamp) losses/stft_loss.py:30-40
 */
#include <math.h>
#include <stddef.h>

static float lrelu(float v, float slope, int on) { return (on && v < 0.f) ? v * slope : v; }

/* x (B,Cin,T) w (Cout,Cin/g,K) y (B,Cout,Tout); zero padding `pad` both sides */
void pwgo_conv1d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int T,
                 int K, int stride, int dil, int pad, int groups, int pre_lrelu, float slope) {
  const int Tout = (T + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const int cig = Cin / groups, cog = Cout / groups;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      const int g = co / cog;
      for (int t = 0; t < Tout; ++t) {
        double acc = bias ? bias[co] : 0.0;
        for (int ci = 0; ci < cig; ++ci)
          for (int k = 0; k < K; ++k) {
            const int ti = t * stride + k * dil - pad;
            if (ti < 0 || ti >= T) continue;
            acc += (double)w[((size_t)co * cig + ci) * K + k] *
                   lrelu(x[((size_t)b * Cin + g * cig + ci) * T + ti], slope, pre_lrelu);
          }
        y[((size_t)b * Cout + co) * Tout + t] = (float)acc;
      }
    }
}

/* x (B,Cin,T) w (Cin,Cout,K) y (B,Cout,Tout), Tout = (T-1)*s - 2*pad + K + out_pad */
void pwgo_conv_transpose1d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                           int T, int K, int stride, int pad, int out_pad, int pre_lrelu, float slope) {
  const int Tout = (T - 1) * stride - 2 * pad + K + out_pad;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co)
      for (int u = 0; u < Tout; ++u) {
        double acc = bias ? bias[co] : 0.0;
        for (int k = 0; k < K; ++k) {
          const int num = u + pad - k;
          if (num < 0 || num % stride) continue;
          const int q = num / stride;
          if (q >= T) continue;
          for (int ci = 0; ci < Cin; ++ci)
            acc += (double)w[((size_t)ci * Cout + co) * K + k] * lrelu(x[((size_t)b * Cin + ci) * T + q], slope, pre_lrelu);
        }
        y[((size_t)b * Cout + co) * Tout + u] = (float)acc;
      }
}

/* count_include_pad semantics of torch.nn.AvgPool1d (ceil_mode = False) */
void pwgo_avg_pool1d(const float* x, float* y, int rows, int T, int K, int stride, int pad, int count_include_pad) {
  const int Tout = (T + 2 * pad - K) / stride + 1;
  for (int r = 0; r < rows; ++r)
    for (int o = 0; o < Tout; ++o) {
      const int start = o * stride - pad;
      int end = start + K;
      if (end > T + pad) end = T + pad;
      const int pool = end - start;
      const int lo = start < 0 ? 0 : start, hi = end > T ? T : end;
      double acc = 0.0;
      for (int t = lo; t < hi; ++t) acc += x[(size_t)r * T + t];
      y[(size_t)r * Tout + o] = (float)(acc / (count_include_pad ? pool : (hi - lo)));
    }
}

/* |STFT| of one row: center=True (reflect pad n_fft/2), periodic Hann of win_length centred in
 * the n_fft frame, one-sided; mag[(f*bins)+k] = sqrt(max(re^2+im^2, eps)); frames = 1 + (T+2*(n/2)-n)/hop */
void pwgo_stft_mag(const float* x, float* mag, int T, int n_fft, int hop, int win, float eps) {
  const int bins = n_fft / 2 + 1, padn = n_fft / 2, off = (n_fft - win) / 2;
  const int frames = 1 + (T + 2 * padn - n_fft) / hop;
  const double two_pi = 6.283185307179586476925286766559;
  for (int f = 0; f < frames; ++f)
    for (int k = 0; k < bins; ++k) {
      double re = 0.0, im = 0.0;
      for (int n = 0; n < win; ++n) {
        int j = f * hop + off + n - padn;
        if (j < 0) j = -j;
        if (j >= T) j = 2 * (T - 1) - j;
        const double wv = 0.5 - 0.5 * cos(two_pi * n / win);
        const double ph = two_pi * k * (double)(n + off) / n_fft;
        re += wv * x[j] * cos(ph);
        im -= wv * x[j] * sin(ph);
      }
      double p = re * re + im * im;
      if (p < eps) p = eps;
      mag[(size_t)f * bins + k] = (float)sqrt(p);
    }
}
