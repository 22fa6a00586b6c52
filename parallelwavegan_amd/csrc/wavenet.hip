// wavenet.hip -- one gated residual layer of the Parallel WaveGAN generator as ONE launch:
//
//     z      = conv_{k=3, dilation d}(x) + b  +  conv1x1_aux(c)            (128 rows; K = 3 * 64 + 80 = 272)
//     g      = tanh(z[:64]) * sigmoid(z[64:])
//     skips' = (conv1x1_skip(g) + b_s + skips) * skip_mul
//     x'     = (conv1x1_out(g)  + b_o + x) * out_mul                       (out_mul = sqrt(0.5))
//
// (/root/reference/parallel_wavegan/layers/residual_block.py:102-140; the running skip sum and its final
// sqrt(1/layers) are models/parallel_wavegan.py:164-169).  Un-fused this is five launches per layer -- aux 1x1,
// dilated conv, gate, skip 1x1, out 1x1 -- that move 2624 B per sample through HBM and run the two K = 64, M = 64
// 1x1 convolutions at 31-44 TFLOP/s; fused, a layer reads x, c and the skip sum once and writes x' and the skip sum
// once: 1344 B per sample, arithmetic intensity 64 flop/B, i.e. MFMA-bound.
//
// A workgroup (4 waves) owns 64 columns of one item.  Its LDS tile holds ALL operand rows of those columns: the
// three tap windows of x (x[n - d], x[n], x[n + d]: with d up to 512 they do not overlap, so each is its own 64 x 64
// window -- at small d the overlapping bytes come from L2) and the 80 aux rows, 272 x 64 floats = 68 KB, filled by
// one LDS-DMA pass (two workgroups per CU: one's load / store phases run under the other's matrix phases).  The
// reduction loop has no barrier and no DMA: B operands are LDS reads at base + immediate, A operands (weights) stream
// from L2 through a pre-swizzled image whose 16-B records are four v_mfma_f32_32x32x2_f32 A operands of a lane.
// Wave (h, n) computes rows [32 h, 32 h + 32) of BOTH halves of z for column block n, so tanh * sigmoid happens
// between two accumulator tiles of the same lane; g goes to LDS (over the first tap window) and is the B operand of
// the two 1x1 convolutions (one K = 64 contraction with 128 rows); the residual x comes from the resident centre
// window.  Training additionally stores z and g (the backward pass reads them).
#include "common.h"

#include <stdint.h>
#include <stdlib.h>

namespace pwg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int WN_R = 64;     // residual channels
constexpr int WN_G = 128;    // gate channels
constexpr int WN_S = 64;     // skip channels
constexpr int WN_K = 3;      // taps
constexpr int WN_COLS = 64;  // columns per workgroup

struct WnArgs {
  const float* x;       // (B, 64, T)
  const float* c;       // (B, AUX, T)
  const float* skips;   // (B, 64, T) running skip sum, may be NULL
  const float* w1;      // image [KP1 / 4][4 row tiles][64 lanes][4]
  const float* w2;      // image [KP2 / 4][4 row tiles][64 lanes][4]
  const float* b_dil;   // (128), may be NULL
  const float* b_skip;  // (64), may be NULL
  const float* b_out;   // (64), may be NULL
  float* x_out;         // (B, 64, T)
  float* skips_out;     // (B, 64, T) (may alias skips)
  float* z_out;         // (B, 128, T) or NULL
  float* g_out;         // (B, 64, T) or NULL
  int T, dil, pad;      // pad: samples left of the first tap (d, or 2 d for the causal form)
  float out_mul, skip_mul;
};

template <int AUX>
__global__ __launch_bounds__(256, 2) void wavenet_layer_kernel(WnArgs a) {
  constexpr int ROWS = WN_K * WN_R + AUX;  // 272 operand rows
  constexpr int NQ1 = ROWS / 8;            // 16-B weight records per lane and row tile in phase 1 (4 k-steps each)
  constexpr int NQ2 = WN_R / 8;            // phase 2: K = 64 rows of g
  static_assert(ROWS % 8 == 0, "aux channels must be a multiple of 8");
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [ROWS][64]; rows 0..63 are overwritten by g
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave >> 1, cn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * WN_COLS;
  const int T = a.T;

  // ---- stage the operand tile: window `tap` of x starts at sample n0 + tap * d - pad
  {
    __amdgpu_buffer_rsrc_t x_rs = uniform_buffer_rsrc(a.x + (long)b * WN_R * T, (unsigned)(WN_R * T) * 4u);
    __amdgpu_buffer_rsrc_t c_rs = uniform_buffer_rsrc(a.c + (long)b * AUX * T, (unsigned)(AUX * T) * 4u);
    for (int q = wave; q < ROWS / 4; q += 4) {  // 4 rows (1 KiB of LDS) per wave instruction
      const int r0 = 4 * q;
      const bool is_x = r0 < WN_K * WN_R;
      const int tap = r0 / WN_R;
      const int f0 = is_x ? n0 + tap * a.dil - a.pad : n0;  // first sample of the window
      const int ch0 = is_x ? r0 - tap * WN_R : r0 - WN_K * WN_R;
      if (__builtin_amdgcn_readfirstlane((f0 >= 0 && f0 + WN_COLS <= T) ? 1 : 0)) {
        const unsigned off = (unsigned)((ch0 + (lane >> 4)) * T + f0 + 4 * (lane & 15)) * 4u;
        if (is_x) __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(tile + r0 * WN_COLS), 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rs, (lds_ptr_t)(tile + r0 * WN_COLS), 16, off, 0, 0, 0);
      } else {
        // window partly outside the sequence: per-sample range check, zeros outside (= the zero padding)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int f = f0 + lane;
          const unsigned off = (f >= 0 && f < T) ? (unsigned)((ch0 + rr) * T + f) * 4u : 0xFFFFFFFCu;
          if (is_x) __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(tile + (r0 + rr) * WN_COLS), 4, off, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rs, (lds_ptr_t)(tile + (r0 + rr) * WN_COLS), 4, off, 0, 0, 0);
        }
      }
    }
  }

  // biases of this lane's accumulator rows: row = 8 * (r >> 2) + 4 * lhi + (r & 3) of the wave's 32-row block
  f32x16 bt, bs, bsk, bo;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = h * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
    bt[r] = a.b_dil ? a.b_dil[row] : 0.f;
    bs[r] = a.b_dil ? a.b_dil[WN_R + row] : 0.f;
    bsk[r] = a.b_skip ? a.b_skip[row] : 0.f;
    bo[r] = a.b_out ? a.b_out[row] : 0.f;
  }

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- phase 1: z rows [32 h, +32) (tanh half) and [64 + 32 h, +32) (sigmoid half) over K = ROWS
  {
    const float4* wa = reinterpret_cast<const float4*>(a.w1) + (h * 64 + lane);        // row tile h
    const float4* wb = reinterpret_cast<const float4*>(a.w1) + ((2 + h) * 64 + lane);  // row tile 2 + h
    float4 A[3][2];
    A[0][0] = wa[0];
    A[0][1] = wb[0];
    A[1][0] = wa[4 * 64];
    A[1][1] = wb[4 * 64];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the DMA pass (and its first weight records)
    __syncthreads();
    const float* bl = tile + lhi * WN_COLS + cn * 32 + l31;  // + (8 q + 2 j) * 64
    float B0[4], B1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) B0[j] = bl[(2 * j) * WN_COLS];
#pragma unroll
    for (int q = 0; q < NQ1; ++q) {
      const int qn = q + 2 < NQ1 ? q + 2 : NQ1 - 1;
      A[(q + 2) % 3][0] = wa[(long)qn * 4 * 64];
      A[(q + 2) % 3][1] = wb[(long)qn * 4 * 64];
      float(&Bc)[4] = (q & 1) ? B1 : B0;
      float(&Bn)[4] = (q & 1) ? B0 : B1;
      const int q1 = q + 1 < NQ1 ? q + 1 : q;
#pragma unroll
      for (int j = 0; j < 4; ++j) Bn[j] = bl[(8 * q1 + 2 * j) * WN_COLS];
      __builtin_amdgcn_sched_barrier(0);
      const float4 a0 = A[q % 3][0], a1 = A[q % 3][1];
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], Bc[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], Bc[j], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- gate in registers; z / g to HBM (training), g to LDS rows 0..63 (every wave is done with the first window)
  const int n = n0 + cn * 32 + l31;
  const bool n_ok = n < T;
  __syncthreads();
  {
    const long zb = (long)b * WN_G * T + n;
    const long gb = (long)b * WN_R * T + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = h * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
      const float zt = acc[0][r] + bt[r], zs = acc[1][r] + bs[r];
      const float g = tanhf(zt) * (1.f / (1.f + expf(-zs)));
      tile[row * WN_COLS + cn * 32 + l31] = g;
      if (a.z_out && n_ok) {
        a.z_out[zb + (long)row * T] = zt;
        a.z_out[zb + (long)(WN_R + row) * T] = zs;
      }
      if (a.g_out && n_ok) a.g_out[gb + (long)row * T] = g;
      acc[0][r] = 0.f;
      acc[1][r] = 0.f;
    }
  }
  __syncthreads();

  // ---- phase 2: skip rows [32 h, +32) and out rows [32 h, +32) over K = 64 rows of g
  {
    const float4* wa = reinterpret_cast<const float4*>(a.w2) + (h * 64 + lane);
    const float4* wb = reinterpret_cast<const float4*>(a.w2) + ((2 + h) * 64 + lane);
    const float* bl = tile + lhi * WN_COLS + cn * 32 + l31;
    float4 A0[NQ2], A1[NQ2];
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
      A0[q] = wa[q * 4 * 64];
      A1[q] = wb[q * 4 * 64];
    }
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
      float Bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) Bv[j] = bl[(8 * q + 2 * j) * WN_COLS];
      const float av0[4] = {A0[q].x, A0[q].y, A0[q].z, A0[q].w}, av1[4] = {A1[q].x, A1[q].y, A1[q].z, A1[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], Bv[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], Bv[j], acc[1], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: D layout col = lane & 31 (time), rows as above
  if (n_ok) {
    const long ob = (long)b * WN_R * T + n;
    float sk[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = h * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
      sk[r] = a.skips ? a.skips[ob + (long)row * T] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = h * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
      const float xc = tile[(WN_R + row) * WN_COLS + cn * 32 + l31];  // centre window = x[n] (non-causal: tap 1)
      float s = acc[0][r] + bsk[r] + sk[r];
      if (a.skip_mul != 1.0f) s *= a.skip_mul;
      a.skips_out[ob + (long)row * T] = s;
      a.x_out[ob + (long)row * T] = (acc[1][r] + bo[r] + xc) * a.out_mul;
    }
  }
}

// ---- weight images ----------------------------------------------------------------------------------------------
// phase 1: rows 0..127 of [w_dil | w_aux] over kc = tap * 64 + ci (kc < 192) or 192 + aux channel;
// phase 2: rows 0..63 = w_skip, 64..127 = w_out over kc = gate-output channel.
// record (q, tile, lane)[j]: row = tile * 32 + (lane & 31), kc = 2 * (4 q + j) + (lane >> 5)
__global__ void wavenet_pack_kernel(const float* w_dil, const float* s_dil, const float* w_aux, const float* s_aux,
                                    const float* w_skip, const float* s_skip, const float* w_out, const float* s_out,
                                    float* out, int aux) {
  const int rows1 = WN_K * WN_R + aux;
  const int n1 = rows1 * WN_G, n2 = WN_R * (WN_S + WN_R);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += gridDim.x * blockDim.x) {
    const bool p2 = i >= n1;
    const int e = p2 ? i - n1 : i;
    const int j = e & 3, lane = (e >> 2) & 63, tl = (e >> 8) & 3, q = e >> 10;
    const int row = tl * 32 + (lane & 31);
    const int kc = 2 * (4 * q + j) + (lane >> 5);
    float v;
    if (!p2) {
      if (kc < WN_K * WN_R) {
        const int tap = kc / WN_R, ci = kc % WN_R;
        v = w_dil[((long)row * WN_R + ci) * WN_K + tap] * (s_dil ? s_dil[row] : 1.f);
      } else {
        v = w_aux[(long)row * aux + (kc - WN_K * WN_R)] * (s_aux ? s_aux[row] : 1.f);
      }
    } else if (row < WN_S) {
      v = w_skip[(long)row * WN_R + kc] * (s_skip ? s_skip[row] : 1.f);
    } else {
      v = w_out[(long)(row - WN_S) * WN_R + kc] * (s_out ? s_out[row - WN_S] : 1.f);
    }
    out[i] = v;
  }
}

static bool wavenet_ok(const pwg_wavenet_desc* d) {
  if (!d) return false;
  if (d->residual_channels != WN_R || d->gate_channels != WN_G || d->skip_channels != WN_S || d->kernel != WN_K) return false;
  if (d->aux_channels != 80) return false;  // (the one instantiation; other widths: template + a case below)
  if (d->batch < 1 || d->batch > 65535 || d->t < 1 || d->dilation < 1) return false;
  if ((long)WN_G * d->t * 4 >= (1L << 32)) return false;
  return true;
}

}  // namespace pwg

using namespace pwg;

extern "C" {

int pwg_wavenet_layer_supported(const pwg_wavenet_desc* d) { return wavenet_ok(d) ? 1 : 0; }

size_t pwg_wavenet_packed_weight_floats(const pwg_wavenet_desc* d) {
  if (!wavenet_ok(d)) return 0;
  return (size_t)(WN_K * WN_R + d->aux_channels) * WN_G + (size_t)WN_R * (WN_S + WN_R);
}

int pwg_wavenet_pack_weights(const pwg_wavenet_desc* d, const float* w_dil, const float* scale_dil, const float* w_aux,
                             const float* scale_aux, const float* w_skip, const float* scale_skip, const float* w_out,
                             const float* scale_out, float* packed, void* stream) {
  PWG_REQUIRE(wavenet_ok(d), PWG_ERR_UNSUPPORTED, "wavenet_pack_weights: unsupported layer geometry");
  PWG_REQUIRE(w_dil && w_aux && w_skip && w_out && packed, PWG_ERR_NULL, "wavenet_pack_weights: NULL pointer");
  const int total = (int)pwg_wavenet_packed_weight_floats(d);
  hipLaunchKernelGGL(wavenet_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w_dil, scale_dil,
                     w_aux, scale_aux, w_skip, scale_skip, w_out, scale_out, packed, d->aux_channels);
  PWG_CHECK_LAUNCH("wavenet_pack_weights");
  return PWG_OK;
}

int pwg_wavenet_layer_forward(const pwg_wavenet_desc* d, const float* x, const float* c, const float* skips,
                              const float* packed, const float* b_dil, const float* b_skip, const float* b_out,
                              float* x_out, float* skips_out, float* z_out, float* g_out, void* stream_) {
  PWG_REQUIRE(wavenet_ok(d), PWG_ERR_UNSUPPORTED, "wavenet_layer_forward: unsupported layer geometry");
  PWG_REQUIRE(x && c && packed && x_out && skips_out, PWG_ERR_NULL, "wavenet_layer_forward: NULL pointer");
  PWG_REQUIRE(x != x_out, PWG_ERR_BAD_SHAPE, "wavenet_layer_forward: x_out must not alias x (tiles read their neighbours' samples)");
  hipStream_t stream = (hipStream_t)stream_;
  WnArgs a;
  a.x = x;
  a.c = c;
  a.skips = skips;
  a.w1 = packed;
  a.w2 = packed + (size_t)(WN_K * WN_R + d->aux_channels) * WN_G;
  a.b_dil = b_dil;
  a.b_skip = b_skip;
  a.b_out = b_out;
  a.x_out = x_out;
  a.skips_out = skips_out;
  a.z_out = z_out;
  a.g_out = g_out;
  a.T = d->t;
  a.dil = d->dilation;
  a.pad = d->causal ? 2 * d->dilation : d->dilation;
  a.out_mul = d->out_mul;
  a.skip_mul = d->skip_mul;
  PWG_REQUIRE(!d->causal, PWG_ERR_UNSUPPORTED, "wavenet_layer_forward: the causal form is not built (residual = last tap window)");
  constexpr int AUX = 80;
  const size_t lds = (size_t)(WN_K * WN_R + AUX) * WN_COLS * sizeof(float);
  void (*kern)(WnArgs) = wavenet_layer_kernel<AUX>;
  if (lds > 64 * 1024 && !lds_limit_is_set(reinterpret_cast<const void*>(kern), lds)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    PWG_REQUIRE(e == hipSuccess, PWG_ERR_LAUNCH, "wavenet_layer_forward: cannot raise the LDS limit to %zu: %s", lds, hipGetErrorString(e));
  }
  const double samples = (double)d->batch * d->t;
  const double flops = 2.0 * samples * (WN_G * (double)(WN_K * WN_R + AUX) + (double)(WN_S + WN_R) * WN_R);
  const double bytes = 4.0 * (samples * (WN_R * 4 + AUX + (skips ? WN_S : 0) + (z_out ? WN_G : 0) + (g_out ? WN_R : 0))) +
                       4.0 * (double)pwg_wavenet_packed_weight_floats(d);
  maybe_poison_lds(stream);
  {
    ProfScope prof(stream, prof_shape_name("wavenet_layer_kernel", "B%d T%d d%d train%d", d->batch, d->t, d->dilation, z_out != nullptr),
                   flops, bytes);
    hipLaunchKernelGGL(kern, dim3(ceil_div(d->t, WN_COLS), d->batch), dim3(256), lds, stream, a);
  }
  PWG_CHECK_LAUNCH("wavenet_layer_forward");
  return PWG_OK;
}

}  // extern "C"
