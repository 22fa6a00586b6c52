// gconv.hip -- grouped strided convolutions with FEW channels per group on v_mfma_f32_16x16x4_f32.
//
// The scale discriminators' k = 41 layers (models/melgan.py:318-336: groups = in_chs / 4, stride 4, i.e. 4 input and
// 8..16 output channels per group; models/hifigan.py:516-540: 128 -> 256 channels in 16 groups, stride 2) are
// "(16 x 4) x (4 x T)" contractions per tap.  On the 32 x 32 x 2 tile of conv1d.hip a group fills 1/8 .. 1/4 of the
// matrix instruction and the weight-gradient kernel 1/64 of its accumulators (1.7 - 3.5 TFLOP/s, r02); a 16-out x
// 4-in group IS one v_mfma_f32_16x16x4_f32 per tap (same 64 FLOP/clk/SIMD as the 32 x 32 x 2 form, exact fp32).
//
// One wave owns one group.  All of a group's weights live in that wave's registers (41 taps x 1-2 VGPRs), its input
// rows stream through a wave-private, double-buffered LDS tile filled by LDS-DMA (no workgroup barrier anywhere: only
// the issuing wave reads the tile, its own vmcnt(0) orders the reads behind the DMA), every operand address is one
// VGPR base + an immediate.
//
//   forward  : D[co][t]        = sum_{tap, ci} W[co][ci][tap] * act(X[ci][t*S + tap - pad])           K-dim = ci
//   data grad: D[(ci, r)][q]   = sum_{j, co}  W[co][ci][j*S + r] * G[co][q - j]   -> dX[ci][q*S + r - pad]   K-dim = co
//   weight gr: D[co][(ci,tap)] = sum_{b, t}   G[co][t] * act(X[ci][t*S + tap - pad])                   K-dim = t
//
// The weight operands are read from the packed images of conv1d.hip ([group][tap][ci pad 16][m pad 128] and its
// polyphase dual), so the parameter caches of the Python layer stay as they are.
#include "common.h"

#include <stdint.h>
#include <stdlib.h>

namespace pwg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct GcArgs {
  const float* x;      // forward / weight gradient: input (B, groups * CIG, t_in); data gradient: unused
  const float* g;      // data / weight gradient: output gradient (B, groups * cog, t_out)
  const float* wp;     // packed weight image (forward image, or the dual's image for the data gradient)
  const float* bias;   // forward
  const float* mask;   // data gradient: forward input (pre-activation derivative), may be NULL
  const float* accum;  // data gradient: added to the result, may be NULL
  float* y;            // forward: (B, groups * cog, t_out); data gradient: dX (B, groups * CIG, t_in)
  float* slabs;        // weight gradient: [slice][groups][16][NPAD + 1] partial sums
  int batch, groups, cog, t_in, t_out, pad;
  int passes_per_wave, chunks, total_waves;
  int pre_act_on, post_act;
  float pre_slope, post_slope, out_mul, mask_slope;
  int slices, steps_total;  // weight gradient: reduction slices per group, 64-column steps over (b, t)
  int vec_ok;  // data gradient, stride 4: 16-B loads / stores of four consecutive samples are legal
};

__device__ __forceinline__ float act_in(float v, float slope) { return __builtin_fmaxf(v, v * slope); }

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int CIG, int K, int S, int NT>
__global__ __launch_bounds__(256, 2) void gconv_fwd_kernel(GcArgs a) {
  constexpr int KS = CIG / 4;                   // MFMA k-steps per tap
  constexpr int COLS = 16 * NT;                 // output columns per pass
  constexpr int L = COLS * S + K - S;           // staged samples per input row and pass
  constexpr int NP = (L + 63) / 64;             // 64-lane DMA pieces per row
  constexpr int RS = NP * 64 + 1;               // odd row stride: the 4 channel rows of an operand read hit different banks
  constexpr int BUF = CIG * RS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, kk = lane >> 4;
  int wid = blockIdx.x * 4 + wave;
  if (wid >= a.total_waves) return;  // (wave-uniform; no workgroup-level synchronisation in this kernel)
  const int g = wid % a.groups;
  wid /= a.groups;
  const int chunk = wid % a.chunks;
  const int b = wid / a.chunks;
  float* xs = smem + wave * (2 * BUF);

  const long x_item = (long)a.groups * CIG * a.t_in;
  __amdgpu_buffer_rsrc_t x_rs = uniform_buffer_rsrc(a.x + (long)b * x_item, (unsigned)(x_item * 4));
  auto issue = [&](int col0, float* buf) {
    const int f0 = col0 * S - a.pad;
#pragma unroll
    for (int r = 0; r < CIG; ++r) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int f = f0 + p * 64 + lane;
        const unsigned off = (f >= 0 && f < a.t_in) ? (unsigned)((g * CIG + r) * a.t_in + f) * 4u : 0xFFFFFFFCu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(buf + r * RS + p * 64), 4, off, 0, 0, 0);
      }
    }
  };
  const int col_first = chunk * a.passes_per_wave * COLS;
  issue(col_first, xs);

  // all weights of the group: A[m = co][k = ci] per tap, from the packed image [g][tap][ci (16)][m (128)]
  float w[K][KS];
  {
    const float* wg = a.wp + (long)g * K * 16 * 128 + n;
#pragma unroll
    for (int tap = 0; tap < K; ++tap)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) w[tap][ks] = wg[(tap * 16 + ks * 4 + kk) * 128];
  }
  float bs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = 4 * kk + r;
    bs[r] = (a.bias && co < a.cog) ? a.bias[g * a.cog + co] : 0.f;
  }
  const long y_item = (long)a.groups * a.cog * a.t_out;
  float* yb = a.y + (long)b * y_item + (long)g * a.cog * a.t_out;

  for (int p = 0; p < a.passes_per_wave; ++p) {
    const int col0 = col_first + p * COLS;
    if (col0 >= a.t_out) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own DMA: the tile of pass p has landed
    float* buf = xs + (p & 1) * BUF;
    if (p + 1 < a.passes_per_wave && col0 + COLS < a.t_out) issue(col0 + COLS, xs + ((p + 1) & 1) * BUF);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xl = buf + kk * RS + n * S;  // + ks * 4 * RS + tile * 16 * S + tap   (immediates)
#pragma unroll
    for (int tap = 0; tap < K; ++tap) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float v = xl[ks * 4 * RS + t * 16 * S + tap];
          if (a.pre_act_on) v = act_in(v, a.pre_slope);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][ks], v, acc[t], 0, 0, 0);
        }
      }
    }
    // D layout: col = lane & 15 (time), row = 4 * (lane >> 4) + reg (output channel)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tt = col0 + t * 16 + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * kk + r;
        if (tt < a.t_out && co < a.cog) {
          float v = acc[t][r] + bs[r];
          if (a.out_mul != 1.0f) v *= a.out_mul;
          yb[(long)co * a.t_out + tt] = apply_act(v, a.post_act, a.post_slope);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// data gradient (the polyphase transposed convolution): rows m = ci * S + r (r fastest), CIG * S == 16
// ---------------------------------------------------------------------------------------------------------------
template <int CIG, int K, int S, int COGS, int NT>
__global__ __launch_bounds__(256, 2) void gconv_dgrad_kernel(GcArgs a) {
  static_assert((CIG * S) % 16 == 0, "whole 16-row MFMA tiles of (input channel, phase) rows");
  constexpr int MT = CIG * S / 16;              // row tiles: rows m = ci * S + r, tile mt holds m in [16 mt, 16 mt + 16)
  constexpr int J = (K + S - 1) / S;            // taps of a phase convolution
  constexpr int COLS = 16 * NT;                 // q columns per pass (COLS * S output samples per channel)
  constexpr int L = COLS + J - 1;
  constexpr int NP = (L + 63) / 64;
  constexpr int RS = NP * 64 + 1;
  constexpr int ROWS = 4 * COGS;                // staged gradient rows (>= cog)
  constexpr int BUF = ROWS * RS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, kk = lane >> 4;
  int wid = blockIdx.x * 4 + wave;
  if (wid >= a.total_waves) return;
  const int g = wid % a.groups;
  wid /= a.groups;
  const int chunk = wid % a.chunks;
  const int b = wid / a.chunks;
  float* gs = smem + wave * (2 * BUF);

  const long g_item = (long)a.groups * a.cog * a.t_out;
  __amdgpu_buffer_rsrc_t g_rs = uniform_buffer_rsrc(a.g + (long)b * g_item, (unsigned)(g_item * 4));
  auto issue = [&](int q0, float* buf) {
    const int f0 = q0 - (J - 1);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int f = f0 + p * 64 + lane;
        const unsigned off = (r < a.cog && f >= 0 && f < a.t_out) ? (unsigned)((g * a.cog + r) * a.t_out + f) * 4u : 0xFFFFFFFCu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_t)(buf + r * RS + p * 64), 4, off, 0, 0, 0);
      }
    }
  };
  const int n_cols = (a.t_in - 1 + a.pad) / S + 1;  // q range covering every output sample u = q * S + r - pad < t_in
  const int q_first = chunk * a.passes_per_wave * COLS;
  issue(q_first, gs);

  // A[m = (ci, r)][k = co] per phase tap, from the dual's packed image [g][tap'][co (16)][m' = r * CIG + ci (128)];
  // tap' multiplies G[q - (J - 1) + tap']
  float w[MT][J][COGS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = 16 * mt + n;
    const int ci = m / S, r = m % S;
    const float* wg = a.wp + (long)g * J * 16 * 128 + r * CIG + ci;
#pragma unroll
    for (int tap = 0; tap < J; ++tap)
#pragma unroll
      for (int ks = 0; ks < COGS; ++ks) w[mt][tap][ks] = wg[(tap * 16 + ks * 4 + kk) * 128];
  }
  const long x_item = (long)a.groups * CIG * a.t_in;
  const long xb = (long)b * x_item + (long)g * CIG * a.t_in;

  for (int p = 0; p < a.passes_per_wave; ++p) {
    const int q0 = q_first + p * COLS;
    if (q0 >= n_cols) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* buf = gs + (p & 1) * BUF;
    if (p + 1 < a.passes_per_wave && q0 + COLS < n_cols) issue(q0 + COLS, gs + ((p + 1) & 1) * BUF);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* gl = buf + kk * RS + n;  // + ks * 4 * RS + tile * 16 + tap'
#pragma unroll
    for (int tap = 0; tap < J; ++tap)
#pragma unroll
      for (int ks = 0; ks < COGS; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float bv = gl[ks * 4 * RS + t * 16 + tap];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[mt][tap][ks], bv, acc[mt][t], 0, 0, 0);
        }
    // D rows 16 * mt + 4 * kk + reg = (ci, r): the lane holds 4 consecutive rows of column q
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int q = q0 + t * 16 + n;
      if (S == 4) {
        // S = 4: ci = 4 * mt + kk, r = reg -> four consecutive samples u0 .. u0 + 3 of one channel row (16-B aligned
        // when pad % 4 == 0, t_in % 4 == 0 and the tensors are: a.vec_ok, decided by the host)
        const int u0 = q * 4 - a.pad;
        const long o = xb + (long)(4 * mt + kk) * a.t_in + u0;
        if (q < n_cols && u0 >= 0 && u0 + 3 < a.t_in && a.vec_ok) {
          f32x4 v = acc[mt][t];
          if (a.mask) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(a.mask + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= (m[r] > 0.f ? 1.f : a.mask_slope);
          }
          if (a.accum) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(a.accum + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += c[r];
          }
          *reinterpret_cast<f32x4*>(a.y + o) = v;
          continue;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * mt + 4 * kk + r;
        const int ci = m / S, ph = m % S;
        const int u = q * S + ph - a.pad;
        if (q < n_cols && u >= 0 && u < a.t_in) {
          const long o = xb + (long)ci * a.t_in + u;
          float v = acc[mt][t][r];
          if (a.mask) v *= (a.mask[o] > 0.f ? 1.f : a.mask_slope);
          if (a.accum) v += a.accum[o];
          a.y[o] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: rows = co, columns n' = ci * K + tap (NTW tiles of 16), reduction over (b, t) in steps of 4
// ---------------------------------------------------------------------------------------------------------------
template <int CIG, int K, int S>
__global__ __launch_bounds__(256, 2) void gconv_wgrad_kernel(GcArgs a) {
  constexpr int NCOL = CIG * K;                 // real columns (torch layout of a weight row: (ci, tap))
  constexpr int NTW = (NCOL + 15) / 16;
  constexpr int TT = 64;                        // reduction columns (output time steps) per pass
  constexpr int LX = TT * S + K - S;
  constexpr int NPX = (LX + 63) / 64;
  constexpr int RSX = NPX * 64 + 1;
  constexpr int RSG = TT + 2;                   // (n * RSG + kk) mod 32 distinct over a 32-lane group
  constexpr int BUF = CIG * RSX + 16 * RSG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, kk = lane >> 4;
  int wid = blockIdx.x * 4 + wave;
  if (wid >= a.total_waves) return;
  const int g = wid % a.groups;
  const int slice = wid / a.groups;
  float* ls = smem + wave * (2 * BUF);

  const long x_item = (long)a.groups * CIG * a.t_in;
  const long g_item = (long)a.groups * a.cog * a.t_out;
  __amdgpu_buffer_rsrc_t x_rs = uniform_buffer_rsrc(a.x, (unsigned)(x_item * a.batch * 4));
  __amdgpu_buffer_rsrc_t g_rs = uniform_buffer_rsrc(a.g, (unsigned)(g_item * a.batch * 4));
  const int steps_per_item = (a.t_out + TT - 1) / TT;
  // this slice's range of (item, 64-column step) pairs
  const int per = (a.steps_total + a.slices - 1) / a.slices;
  const int s_begin = slice * per;
  const int s_end = min(s_begin + per, a.steps_total);

  auto issue = [&](int s, float* buf) {
    const int b = s / steps_per_item;
    const int t0 = (s - b * steps_per_item) * TT;
    float* xs = buf;
    float* gsm = buf + CIG * RSX;
    const int f0 = t0 * S - a.pad;
#pragma unroll
    for (int r = 0; r < CIG; ++r) {
#pragma unroll
      for (int p = 0; p < NPX; ++p) {
        const int f = f0 + p * 64 + lane;
        const unsigned off = (f >= 0 && f < a.t_in) ? (unsigned)(b * x_item + (long)(g * CIG + r) * a.t_in + f) * 4u : 0xFFFFFFFCu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * RSX + p * 64), 4, off, 0, 0, 0);
      }
    }
    // G tile: 16 rows x 64 columns (rows >= cog and columns >= t_out are zero)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + lane;
      const unsigned off = (r < a.cog && t < a.t_out) ? (unsigned)(b * g_item + (long)(g * a.cog + r) * a.t_out + t) * 4u : 0xFFFFFFFCu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_t)(gsm + r * RSG), 4, off, 0, 0, 0);
    }
  };

  f32x4 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane constant part of the B operand address of column tile t: n' = t * 16 + n -> ci * RSX + tap, + kk * S
  int boff[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    int np = t * 16 + n;
    if (np > NCOL - 1) np = NCOL - 1;  // surplus columns of the last tile repeat the last real one (dropped below)
    boff[t] = (np / K) * RSX + (np % K) + kk * S;
  }
  float bsum = 0.f;
  if (s_begin < s_end) issue(s_begin, ls);
  for (int s = s_begin; s < s_end; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* buf = ls + ((s - s_begin) & 1) * BUF;
    if (s + 1 < s_end) issue(s + 1, ls + ((s - s_begin + 1) & 1) * BUF);
    const float* xs = buf;
    const float* gl = buf + CIG * RSX + n * RSG + kk;  // A[m = co = n][k = kk] = G[co][t0 + 4 * step + kk]
#pragma unroll 4
    for (int st = 0; st < TT / 4; ++st) {
      const float av = gl[4 * st];
      bsum += av;
      const float* xl = xs + 4 * st * S;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        float v = xl[boff[t]];
        if (a.pre_act_on) v = act_in(v, a.pre_slope);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v, acc[t], 0, 0, 0);
      }
    }
  }
  // partial sums of this slice: slabs[slice][g][co (16)][NTW * 16 + 1]; D col = lane & 15 (n'), row = 4 * kk + reg (co)
  constexpr int ROWW = NTW * 16 + 1;
  float* slab = a.slabs + ((long)slice * a.groups + g) * 16 * ROWW;
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[(4 * kk + r) * ROWW + t * 16 + n] = acc[t][r];
  // fused bias gradient: row sums of G; lanes (n, kk = 0..3) hold the four k slots of row n
  bsum += __shfl_xor(bsum, 16, 64);
  bsum += __shfl_xor(bsum, 32, 64);
  if (kk == 0) slab[n * ROWW + NTW * 16] = bsum;
}

// dw[(g * cog + co) * NCOL + n'] = sum over slices (in order); db[g * cog + co] likewise from the extra column
__global__ __launch_bounds__(256) void gconv_wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                 float* __restrict__ db, int groups, int cog, int ncol,
                                                                 int roww, int slices) {
  __shared__ float part[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long elems = (long)groups * cog * (ncol + 1);
  const long e = (long)blockIdx.x * 32 + el;
  float s = 0.f;
  long src = 0;
  int g = 0, co = 0, c = 0;
  if (e < elems) {
    c = (int)(e % (ncol + 1));
    long r = e / (ncol + 1);
    co = (int)(r % cog);
    g = (int)(r / cog);
    src = ((long)g * 16 + co) * roww + (c < ncol ? c : roww - 1);
    const long stride = (long)groups * 16 * roww;
    for (int j = sl; j < slices; j += 8) s += slabs[(long)j * stride + src];
  }
  part[sl][el] = s;
  __syncthreads();
  if (sl == 0 && e < elems) {
    float t = part[0][el];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += part[q][el];
    if (c < ncol) dw[((long)g * cog + co) * ncol + c] = t;
    else if (db) db[g * cog + co] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static bool gconv_geometry_ok(const pwg_conv1d_desc* d) {
  if (d->transposed || d->width != 1 || d->dilation != 1 || d->pad_mode != PWG_PAD_ZERO || d->kernel != 41) return false;
  if (d->groups < 2 || d->c_in % d->groups || d->c_out % d->groups) return false;
  const int cig = d->c_in / d->groups, cog = d->c_out / d->groups;
  if (!((cig == 4 && d->stride == 4) || (cig == 8 && (d->stride == 2 || d->stride == 4)))) return false;
  if (cog != 8 && cog != 16) return false;
  if ((long)d->batch * d->c_in * d->t_in * 4 >= 0xFFFFFFF0L || (long)d->batch * d->c_out * d->t_out * 4 >= 0xFFFFFFF0L) return false;
  static const bool off = getenv("PWG_NO_GCONV") != nullptr;  // (A/B switch for tools/bench_gconv.py)
  return !off;
}

bool gconv_forward_applicable(const pwg_conv1d_desc* d, const float* add1, const float* add2) {
  if (!gconv_geometry_ok(d) || add1 || add2 || d->out_div != 1.0f) return false;
  if (d->pre_act == PWG_ACT_TANH) return false;
  if (d->pre_act == PWG_ACT_LEAKY_RELU && !(d->pre_slope >= 0.f && d->pre_slope <= 1.f)) return false;
  return true;
}

template <typename KernT>
static int raise_lds(KernT kern, size_t lds, const char* what) {
  if (lds > 64 * 1024 && !lds_limit_is_set(reinterpret_cast<const void*>(kern), lds)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    PWG_REQUIRE(e == hipSuccess, PWG_ERR_LAUNCH, "%s: cannot raise the LDS limit to %zu: %s", what, lds, hipGetErrorString(e));
  }
  return PWG_OK;
}

// passes per wave such that the launch has about 16 waves per CU (4096) while a wave amortises its weight load
static int plan_passes(int per_item_passes, long items) {
  int ppw = 1;
  while (ppw < 8 && (long)((per_item_passes + 2 * ppw - 1) / (2 * ppw)) * items >= 4096) ppw *= 2;
  return ppw;
}

int gconv_forward(const pwg_conv1d_desc* d, const float* x, const float* wp, const float* bias, float* y, hipStream_t stream) {
  PWG_REQUIRE(x && wp && y, PWG_ERR_NULL, "conv1d_forward: NULL pointer");
  const int cig = d->c_in / d->groups;
  GcArgs a = {};
  a.x = x;
  a.wp = wp;
  a.bias = bias;
  a.y = y;
  a.batch = d->batch;
  a.groups = d->groups;
  a.cog = d->c_out / d->groups;
  a.t_in = d->t_in;
  a.t_out = d->t_out;
  a.pad = d->pad_left;
  a.pre_act_on = d->pre_act != PWG_ACT_NONE;
  a.pre_slope = d->pre_act == PWG_ACT_LEAKY_RELU ? d->pre_slope : 0.f;
  a.post_act = d->post_act;
  a.post_slope = d->post_slope;
  a.out_mul = d->out_mul;
  constexpr int NT = 4;
  const int passes = ceil_div(d->t_out, 16 * NT);
  a.passes_per_wave = plan_passes(passes, (long)d->batch * d->groups);
  a.chunks = ceil_div(passes, a.passes_per_wave);
  a.total_waves = d->batch * d->groups * a.chunks;
  const double out_elems = (double)d->batch * d->c_out * d->t_out;
  const double flops = 2.0 * out_elems * cig * d->kernel;
  const double bytes = 4.0 * ((double)d->batch * d->c_in * d->t_in + out_elems + (double)d->c_out * cig * d->kernel);
  ProfScope prof(stream, prof_shape_name("gconv_fwd_kernel", "B%d Cin%d Cout%d Tin%d Tout%d k%d s%d g%d", d->batch, d->c_in,
                                         d->c_out, d->t_in, d->t_out, d->kernel, d->stride, d->groups), flops, bytes);
  dim3 grid(ceil_div(a.total_waves, 4)), block(256);
  if (cig == 4) {
    constexpr int RS = ((16 * NT * 4 + 41 - 4 + 63) / 64) * 64 + 1;
    const size_t lds = (size_t)4 * 2 * 4 * RS * sizeof(float);
    auto kern = gconv_fwd_kernel<4, 41, 4, NT>;
    if (int rc = raise_lds(kern, lds, "gconv_forward")) return rc;
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  } else if (d->stride == 2) {
    constexpr int RS = ((16 * NT * 2 + 41 - 2 + 63) / 64) * 64 + 1;
    const size_t lds = (size_t)4 * 2 * 8 * RS * sizeof(float);
    auto kern = gconv_fwd_kernel<8, 41, 2, NT>;
    if (int rc = raise_lds(kern, lds, "gconv_forward")) return rc;
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  } else {
    constexpr int RS = ((16 * NT * 4 + 41 - 4 + 63) / 64) * 64 + 1;
    const size_t lds = (size_t)4 * 2 * 8 * RS * sizeof(float);
    auto kern = gconv_fwd_kernel<8, 41, 4, NT>;
    if (int rc = raise_lds(kern, lds, "gconv_forward")) return rc;
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  }
  PWG_CHECK_LAUNCH("gconv_forward");
  return PWG_OK;
}

// `d` is the FORWARD descriptor of the layer (not the dual)
bool gconv_dgrad_applicable(const pwg_conv1d_desc* d) { return gconv_geometry_ok(d); }

int gconv_backward_data(const pwg_conv1d_desc* d, const float* dy, const float* wp_bwd, const float* x_fwd, const float* accum,
                        float* dx, hipStream_t stream) {
  PWG_REQUIRE(dy && wp_bwd && dx, PWG_ERR_NULL, "conv1d_backward_data: NULL pointer");
  const int cig = d->c_in / d->groups, cog = d->c_out / d->groups;
  GcArgs a = {};
  a.g = dy;
  a.wp = wp_bwd;
  a.y = dx;
  a.accum = accum;
  if (d->pre_act != PWG_ACT_NONE) {
    a.mask = x_fwd;
    a.mask_slope = d->pre_act == PWG_ACT_LEAKY_RELU ? d->pre_slope : 0.f;
  }
  a.batch = d->batch;
  a.groups = d->groups;
  a.cog = cog;
  a.t_in = d->t_in;
  a.t_out = d->t_out;
  a.pad = d->pad_left;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  a.vec_ok = d->stride == 4 && d->pad_left % 4 == 0 && d->t_in % 4 == 0 && al16(dx) && al16(accum) && al16(a.mask);
  constexpr int NT = 4;
  const int n_cols = (d->t_in - 1 + d->pad_left) / d->stride + 1;
  const int passes = ceil_div(n_cols, 16 * NT);
  a.passes_per_wave = plan_passes(passes, (long)d->batch * d->groups);
  a.chunks = ceil_div(passes, a.passes_per_wave);
  a.total_waves = d->batch * d->groups * a.chunks;
  const double flops = 2.0 * (double)d->batch * d->c_out * d->t_out * cig * d->kernel;
  const double bytes = 4.0 * ((double)d->batch * d->c_in * d->t_in * (1 + (accum != nullptr) + (a.mask != nullptr)) +
                              (double)d->batch * d->c_out * d->t_out);
  ProfScope prof(stream, prof_shape_name("gconv_dgrad_kernel", "B%d Cin%d Cout%d Tin%d Tout%d k%d s%d g%d", d->batch, d->c_in,
                                         d->c_out, d->t_in, d->t_out, d->kernel, d->stride, d->groups), flops, bytes);
  dim3 grid(ceil_div(a.total_waves, 4)), block(256);
  constexpr int RS = ((16 * NT + 11 - 1 + 63) / 64) * 64 + 1;  // J = 11 (stride 4) or 21 (stride 2): see below
  if (cig == 4) {
    if (cog == 16) {
      const size_t lds = (size_t)4 * 2 * 16 * RS * sizeof(float);
      auto kern = gconv_dgrad_kernel<4, 41, 4, 4, NT>;
      if (int rc = raise_lds(kern, lds, "gconv_backward_data")) return rc;
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    } else {
      const size_t lds = (size_t)4 * 2 * 8 * RS * sizeof(float);
      auto kern = gconv_dgrad_kernel<4, 41, 4, 2, NT>;
      if (int rc = raise_lds(kern, lds, "gconv_backward_data")) return rc;
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    }
  } else if (d->stride == 4) {
    if (cog == 16) {
      const size_t lds = (size_t)4 * 2 * 16 * RS * sizeof(float);
      auto kern = gconv_dgrad_kernel<8, 41, 4, 4, NT>;
      if (int rc = raise_lds(kern, lds, "gconv_backward_data")) return rc;
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    } else {
      const size_t lds = (size_t)4 * 2 * 8 * RS * sizeof(float);
      auto kern = gconv_dgrad_kernel<8, 41, 4, 2, NT>;
      if (int rc = raise_lds(kern, lds, "gconv_backward_data")) return rc;
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    }
  } else {
    constexpr int RS2 = ((16 * NT + 21 - 1 + 63) / 64) * 64 + 1;
    if (cog == 16) {
      const size_t lds = (size_t)4 * 2 * 16 * RS2 * sizeof(float);
      auto kern = gconv_dgrad_kernel<8, 41, 2, 4, NT>;
      if (int rc = raise_lds(kern, lds, "gconv_backward_data")) return rc;
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    } else {
      const size_t lds = (size_t)4 * 2 * 8 * RS2 * sizeof(float);
      auto kern = gconv_dgrad_kernel<8, 41, 2, 2, NT>;
      if (int rc = raise_lds(kern, lds, "gconv_backward_data")) return rc;
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    }
  }
  PWG_CHECK_LAUNCH("gconv_backward_data");
  return PWG_OK;
}

bool gconv_wgrad_applicable(const pwg_conv1d_desc* d) {
  if (!gconv_geometry_ok(d)) return false;
  const float slope = d->pre_act == PWG_ACT_LEAKY_RELU ? d->pre_slope : (d->pre_act == PWG_ACT_RELU ? 0.f : 1.f);
  return d->pre_act != PWG_ACT_TANH && slope >= 0.f && slope <= 1.f;
}

static int gconv_wgrad_slices(const pwg_conv1d_desc* d) {
  const int steps = d->batch * ceil_div(d->t_out, 64);
  int slices = 4096 / d->groups;          // ~16 waves per CU
  if (slices > steps / 2) slices = steps / 2;  // at least two 64-column steps per slice
  if (slices < 1) slices = 1;
  return slices;
}

size_t gconv_wgrad_workspace_floats(const pwg_conv1d_desc* d) {
  const int cig = d->c_in / d->groups;
  const int roww = ceil_div(cig * d->kernel, 16) * 16 + 1;
  // slabs + room for the summed gradient (weight-norm finish)
  return (size_t)gconv_wgrad_slices(d) * d->groups * 16 * roww + (size_t)d->c_out * cig * d->kernel + d->c_out;
}

// dw: torch layout (c_out, cig, k); db may be NULL.  dw == NULL is not supported here (bias-only calls use the general path).
int gconv_backward_weight(const pwg_conv1d_desc* d, const float* x, const float* dy, float* dw, float* db, float* workspace,
                          size_t ws_floats, hipStream_t stream) {
  PWG_REQUIRE(x && dy && dw, PWG_ERR_NULL, "conv1d_backward_weight: NULL pointer");
  const int cig = d->c_in / d->groups, cog = d->c_out / d->groups;
  const int ncol = cig * d->kernel;
  const int roww = ceil_div(ncol, 16) * 16 + 1;
  GcArgs a = {};
  a.x = x;
  a.g = dy;
  a.batch = d->batch;
  a.groups = d->groups;
  a.cog = cog;
  a.t_in = d->t_in;
  a.t_out = d->t_out;
  a.pad = d->pad_left;
  a.pre_act_on = d->pre_act != PWG_ACT_NONE;
  a.pre_slope = d->pre_act == PWG_ACT_LEAKY_RELU ? d->pre_slope : 0.f;
  a.slices = gconv_wgrad_slices(d);
  a.steps_total = d->batch * ceil_div(d->t_out, 64);
  a.total_waves = a.slices * d->groups;
  const size_t need = (size_t)a.slices * d->groups * 16 * roww;
  PWG_REQUIRE(workspace && ws_floats >= need, PWG_ERR_WORKSPACE, "conv1d_backward_weight: workspace of %zu floats needed, %zu given",
              need, ws_floats);
  a.slabs = workspace;
  const double flops = 2.0 * (double)d->batch * d->c_out * d->t_out * cig * d->kernel;
  const double bytes = 4.0 * ((double)d->batch * d->c_in * d->t_in + (double)d->batch * d->c_out * d->t_out + (double)d->c_out * ncol);
  dim3 grid(ceil_div(a.total_waves, 4)), block(256);
  {
    ProfScope prof(stream, prof_shape_name("gconv_wgrad_kernel", "B%d Cin%d Cout%d Tin%d Tout%d k%d s%d g%d slices%d", d->batch,
                                           d->c_in, d->c_out, d->t_in, d->t_out, d->kernel, d->stride, d->groups, a.slices),
                   flops, bytes);
    if (cig == 4) {
      constexpr int RSX = ((64 * 4 + 41 - 4 + 63) / 64) * 64 + 1;
      const size_t lds = (size_t)4 * 2 * (4 * RSX + 16 * 66) * sizeof(float);
      auto kern = gconv_wgrad_kernel<4, 41, 4>;
      if (int rc = raise_lds(kern, lds, "gconv_backward_weight")) return rc;
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    } else if (d->stride == 2) {
      constexpr int RSX = ((64 * 2 + 41 - 2 + 63) / 64) * 64 + 1;
      const size_t lds = (size_t)4 * 2 * (8 * RSX + 16 * 66) * sizeof(float);
      auto kern = gconv_wgrad_kernel<8, 41, 2>;
      if (int rc = raise_lds(kern, lds, "gconv_backward_weight")) return rc;
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    } else {
      constexpr int RSX = ((64 * 4 + 41 - 4 + 63) / 64) * 64 + 1;
      const size_t lds = (size_t)4 * 2 * (8 * RSX + 16 * 66) * sizeof(float);
      auto kern = gconv_wgrad_kernel<8, 41, 4>;
      if (int rc = raise_lds(kern, lds, "gconv_backward_weight")) return rc;
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    }
    PWG_CHECK_LAUNCH("gconv_backward_weight");
  }
  {
    const long elems = (long)d->c_out * (ncol + 1);
    ProfScope prof(stream, "gconv_wgrad_reduce_kernel", 0, 4.0 * ((double)need + (double)elems));
    hipLaunchKernelGGL(gconv_wgrad_reduce_kernel, dim3((unsigned)((elems + 31) / 32)), dim3(256), 0, stream,
                       (const float*)workspace, dw, db, d->groups, cog, ncol, roww, a.slices);
    PWG_CHECK_LAUNCH("gconv_wgrad_reduce");
  }
  return PWG_OK;
}

}  // namespace pwg
