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
#include <type_traits>
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
  int vec_ok;  // x_out / skips / skips_out are 16-B aligned
  int dbg;  // timing experiments only (PWG_WN_DBG): 1 = phase-1 weights loaded once, 2 = no tanh / exp, 4 = no epilogue
            // loads / stores, 8 = no operand DMA
};

// tanh(t) * sigmoid(s) on the hardware exp2 / rcp (1 ulp each): sigmoid(v) = 1 / (1 + 2^(-v log2 e)),
// tanh(t) = 2 sigmoid(2 t) - 1; saturates correctly (2^inf = inf -> rcp = 0).  Absolute error ~1e-7, against
// 8 % of the kernel for libm's tanhf + expf (profiles/r03_wavenet_ablation.txt).
__device__ __forceinline__ float gate_fast(float t, float s) {
  const float L2E = 1.4426950408889634f;
  const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-s * L2E));
  const float th = 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.f * L2E * t)) - 1.f;
  return th * sg;
}

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
    for (int q = wave; q < ((a.dbg & 8) ? 0 : ROWS / 4); q += 4) {  // 4 rows (1 KiB of LDS) per wave instruction
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
      const int qn = (a.dbg & 1) ? 0 : (q + 2 < NQ1 ? q + 2 : NQ1 - 1);
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
  __syncthreads();
  float* scr = tile + 2 * WN_R * WN_COLS + wave * (2 * 32 * 36);  // wave-private scratch in the dead rows 128..
  const int trow = lane >> 3, tcol = (lane & 7) * 4;
  const int nq = n0 + cn * 32 + tcol;
  const bool vec = ((T & 3) == 0) && a.vec_ok;
  {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = 8 * (r >> 2) + 4 * lhi + (r & 3);
      const float zt = acc[0][r] + bt[r], zs = acc[1][r] + bs[r];
      const float g = (a.dbg & 2) ? zt * zs : gate_fast(zt, zs);
      tile[(h * 32 + rl) * WN_COLS + cn * 32 + l31] = g;
      if (a.z_out) {  // (training: the gate input goes out through the transposition scratch, 16-B stores)
        scr[rl * 36 + l31] = zt;
        scr[32 * 36 + rl * 36 + l31] = zs;
      }
      acc[0][r] = 0.f;
      acc[1][r] = 0.f;
    }
    if (a.z_out) {
      const long zb = (long)b * WN_G * T + nq;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int rl = ps * 8 + trow;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const float4 v = *reinterpret_cast<const float4*>(scr + half * 32 * 36 + rl * 36 + tcol);
          const long o = zb + (long)(half * WN_R + h * 32 + rl) * T;
          if (vec) {
            if (nq < T) *reinterpret_cast<float4*>(a.z_out + o) = v;
          } else {
            const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (nq + e < T) a.z_out[o + e] = e4[e];
          }
        }
      }
    }
  }
  __syncthreads();
  if (a.g_out) {
    // the gate output tile is row-major in LDS: 64 rows x 16 float4, 4 per lane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = i * 256 + tid;
      const int row = idx >> 4, c4 = (idx & 15) * 4;
      const float4 v = *reinterpret_cast<const float4*>(tile + row * WN_COLS + c4);
      const long o = ((long)b * WN_R + row) * T + n0 + c4;
      if (vec) {
        if (n0 + c4 < T) *reinterpret_cast<float4*>(a.g_out + o) = v;
      } else {
        const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + c4 + e < T) a.g_out[o + e] = e4[e];
      }
    }
  }

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
  if (a.dbg & 4) {
    if (acc[0][0] == 12345.678f && acc[1][3] == 1.f) a.x_out[0] = 1.f;
    return;
  }
  // Each wave transposes its two 32 x 32 result tiles through a private scratch (rows 128.. of the operand tile: the
  // last tap window and the aux rows are dead after phase 1) so that a lane owns 4 consecutive samples of a row:
  // 16-B loads of the skip sum, 16-B stores of both outputs (the D layout gives a lane ONE sample of 16 rows, i.e.
  // 48 dword accesses per lane: 14 % of the kernel).  The arithmetic per element is unchanged.
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rl = 8 * (r >> 2) + 4 * lhi + (r & 3);
    const float xc = tile[(WN_R + h * 32 + rl) * WN_COLS + cn * 32 + l31];  // centre window = x[n]
    scr[rl * 36 + l31] = acc[0][r] + bsk[r];
    scr[32 * 36 + rl * 36 + l31] = (acc[1][r] + bo[r] + xc) * a.out_mul;
  }
  // (wave-private scratch: no workgroup barrier, the wave's own LDS accesses are ordered)
  const long ob = (long)b * WN_R * T + nq;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int rl = ps * 8 + trow;
    const long o = ob + (long)(h * 32 + rl) * T;
    float4 sv = *reinterpret_cast<const float4*>(scr + rl * 36 + tcol);
    const float4 ov = *reinterpret_cast<const float4*>(scr + 32 * 36 + rl * 36 + tcol);
    if (vec) {
      if (nq < T) {  // (T % 4 == 0: a float4 is inside the sequence or outside it)
        if (a.skips) {
          const float4 k = *reinterpret_cast<const float4*>(a.skips + o);
          sv.x += k.x; sv.y += k.y; sv.z += k.z; sv.w += k.w;
        }
        if (a.skip_mul != 1.0f) { sv.x *= a.skip_mul; sv.y *= a.skip_mul; sv.z *= a.skip_mul; sv.w *= a.skip_mul; }
        *reinterpret_cast<float4*>(a.skips_out + o) = sv;
        *reinterpret_cast<float4*>(a.x_out + o) = ov;
      }
    } else {
      const float se[4] = {sv.x, sv.y, sv.z, sv.w}, oe[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (nq + e < T) {
          float v = se[e] + (a.skips ? a.skips[o + e] : 0.f);
          if (a.skip_mul != 1.0f) v *= a.skip_mul;
          a.skips_out[o + e] = v;
          a.x_out[o + e] = oe[e];
        }
      }
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


// =====================================================================================================================
// backward, data path.  Two launches per layer instead of six (scale of dx_out, two 1x1 data gradients, gate backward,
// dilated data gradient, aux data gradient):
//
//   wavenet_gate_bwd_kernel : dg = Wo^T (out_mul dx_out) + Ws^T (skip_mul ds_out)   (K = 128, 64 rows)
//                             dz = [dg * sg * (1 - th^2) ; dg * th * sg * (1 - sg)],  th = tanh(z[:64]), sg = sigmoid(z[64:])
//                             go = out_mul * dx_out  (the residual-path / out-conv weight-gradient operand)
//                             HBM-bound: 1792 B per sample for 16 KFLOP.
//   wavenet_dgrad_kernel    : dx[ci][n] = sum_{t, co} Wd[co][ci][t] dz[co][n - (t - 1) d] + go[ci][n]      (K = 384, 64 rows)
//                             dc[a][n]  = sum_co Wa[co][a] dz[co][n]                                       (K = 128, 80 rows)
//                             with the three dz windows resident in LDS (384 rows x 32 columns, three workgroups per CU).
// =====================================================================================================================
struct WnBwdArgs {
  const float* z;       // (B, 128, T)
  const float* dx_out;  // (B, 64, T) or NULL (the last layer's x_out is unused)
  const float* ds_out;  // (B, 64, T)
  const float* dz_in;   // dgrad kernel: (B, 128, T)
  const float* go_in;   // dgrad kernel: (B, 64, T) or NULL
  const float* w;       // image of this kernel
  float* dz;            // gate kernel out (B, 128, T)
  float* go;            // gate kernel out (B, 64, T) or NULL
  float* dx;            // dgrad kernel out (B, 64, T) or NULL
  float* dc;            // dgrad kernel out (B, AUX, T) or NULL
  const float* dc_in;   // dgrad kernel: gradient already accumulated for c by the later layers (added), or NULL
  int T, dil, aux;
  float out_mul;
  int vec_ok;
};

__global__ __launch_bounds__(256, 2) void wavenet_gate_bwd_kernel(WnBwdArgs a) {
  constexpr int NQ = 16;  // K = 128: [ds_out rows | dx_out rows]
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [128][64] + 4 x [32][36] scratch
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave >> 1, cn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * WN_COLS;
  const int T = a.T;
  {
    const float* any = a.ds_out;
    __amdgpu_buffer_rsrc_t s_rs = uniform_buffer_rsrc(a.ds_out + (long)b * WN_S * T, (unsigned)(WN_S * T) * 4u);
    __amdgpu_buffer_rsrc_t o_rs = uniform_buffer_rsrc((a.dx_out ? a.dx_out : any) + (long)b * WN_R * T,
                                                      a.dx_out ? (unsigned)(WN_R * T) * 4u : 0u);  // NULL: every load is out of range = 0
    for (int q = wave; q < 32; q += 4) {
      const int r0 = 4 * q;
      const bool is_s = r0 < WN_S;
      const int ch0 = is_s ? r0 : r0 - WN_S;
      if (__builtin_amdgcn_readfirstlane((n0 + WN_COLS <= T) ? 1 : 0)) {
        const unsigned off = (unsigned)((ch0 + (lane >> 4)) * T + n0 + 4 * (lane & 15)) * 4u;
        if (is_s) __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rs, (lds_ptr_t)(tile + r0 * WN_COLS), 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(o_rs, (lds_ptr_t)(tile + r0 * WN_COLS), 16, off, 0, 0, 0);
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int f = n0 + lane;
          const unsigned off = f < T ? (unsigned)((ch0 + rr) * T + f) * 4u : 0xFFFFFFFCu;
          if (is_s) __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rs, (lds_ptr_t)(tile + (r0 + rr) * WN_COLS), 4, off, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(o_rs, (lds_ptr_t)(tile + (r0 + rr) * WN_COLS), 4, off, 0, 0, 0);
        }
      }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float4* wa = reinterpret_cast<const float4*>(a.w) + (h * 64 + lane);  // image [q][2 tiles][lane][4]
  float4 A[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) A[q] = wa[q * 2 * 64];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const float* bl = tile + lhi * WN_COLS + cn * 32 + l31;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float av[4] = {A[q].x, A[q].y, A[q].z, A[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bl[(8 * q + 2 * j) * WN_COLS], acc, 0, 0, 0);
    }
  }
  // dg tile -> wave-private scratch -> a lane owns 4 consecutive samples of a row
  float* scr = tile + WN_G * WN_COLS + wave * (32 * 36);
#pragma unroll
  for (int r = 0; r < 16; ++r) scr[(8 * (r >> 2) + 4 * lhi + (r & 3)) * 36 + l31] = acc[r];
  const int trow = lane >> 3, tcol = (lane & 7) * 4;
  const int nq = n0 + cn * 32 + tcol;
  const bool vec = ((T & 3) == 0) && a.vec_ok;
  const float L2E = 1.4426950408889634f;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int rl = ps * 8 + trow;
    const int row = h * 32 + rl;
    const float4 dg4 = *reinterpret_cast<const float4*>(scr + rl * 36 + tcol);
    const long ot = ((long)b * WN_G + row) * T + nq, os = ot + (long)WN_R * T;
    float zt[4], zs[4];
    if (vec) {
      if (nq >= T) continue;
      const float4 t4 = *reinterpret_cast<const float4*>(a.z + ot), s4 = *reinterpret_cast<const float4*>(a.z + os);
      zt[0] = t4.x; zt[1] = t4.y; zt[2] = t4.z; zt[3] = t4.w;
      zs[0] = s4.x; zs[1] = s4.y; zs[2] = s4.z; zs[3] = s4.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        zt[e] = nq + e < T ? a.z[ot + e] : 0.f;
        zs[e] = nq + e < T ? a.z[os + e] : 0.f;
      }
    }
    const float dg[4] = {dg4.x, dg4.y, dg4.z, dg4.w};
    float dt[4], dsg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-zs[e] * L2E));
      const float th = 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.f * L2E * zt[e])) - 1.f;
      dt[e] = dg[e] * sg * (1.f - th * th);
      dsg[e] = dg[e] * th * sg * (1.f - sg);
    }
    if (vec) {
      *reinterpret_cast<float4*>(a.dz + ot) = make_float4(dt[0], dt[1], dt[2], dt[3]);
      *reinterpret_cast<float4*>(a.dz + os) = make_float4(dsg[0], dsg[1], dsg[2], dsg[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (nq + e < T) {
          a.dz[ot + e] = dt[e];
          a.dz[os + e] = dsg[e];
        }
    }
  }
  if (a.go && a.dx_out) {
    // go = out_mul * dx_out from the staged rows 64..127 (row-major in LDS): 64 rows x 16 float4, 4 per lane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = i * 256 + tid;
      const int row = idx >> 4, c4 = (idx & 15) * 4;
      float4 v = *reinterpret_cast<const float4*>(tile + (WN_S + row) * WN_COLS + c4);
      v.x *= a.out_mul; v.y *= a.out_mul; v.z *= a.out_mul; v.w *= a.out_mul;
      const long o = ((long)b * WN_R + row) * T + n0 + c4;
      if (vec) {
        if (n0 + c4 < T) *reinterpret_cast<float4*>(a.go + o) = v;
      } else {
        const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + c4 + e < T) a.go[o + e] = e4[e];
      }
    }
  }
}

constexpr int WD_COLS = 32;  // columns per workgroup of the data-gradient kernel

// D += A (image records `wa`, stride `rec_stride` float4 per record) x B (LDS rows starting at `bl`) over nq records
template <int NTL>
__device__ __forceinline__ void wn_contract(f32x16 (&acc)[NTL], const float4* const (&wa)[NTL], int rec_stride, int nq,
                                            const float* bl) {
  float4 A0[NTL], A1[NTL];
#pragma unroll
  for (int i = 0; i < NTL; ++i) A0[i] = wa[i][0];
  for (int q = 0; q < nq; q += 2) {
#pragma unroll
    for (int i = 0; i < NTL; ++i) A1[i] = wa[i][(long)(q + 1 < nq ? q + 1 : q) * rec_stride];
    {
      float Bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) Bv[j] = bl[(8 * q + 2 * j) * WD_COLS];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
          const float av = j == 0 ? A0[i].x : j == 1 ? A0[i].y : j == 2 ? A0[i].z : A0[i].w;
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Bv[j], acc[i], 0, 0, 0);
        }
    }
    if (q + 1 < nq) {
#pragma unroll
      for (int i = 0; i < NTL; ++i) A0[i] = wa[i][(long)(q + 2 < nq ? q + 2 : q + 1) * rec_stride];
      float Bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) Bv[j] = bl[(8 * (q + 1) + 2 * j) * WD_COLS];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
          const float av = j == 0 ? A1[i].x : j == 1 ? A1[i].y : j == 2 ? A1[i].z : A1[i].w;
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Bv[j], acc[i], 0, 0, 0);
        }
    }
  }
}

__global__ __launch_bounds__(256, 2) void wavenet_dgrad_kernel(WnBwdArgs a) {
  constexpr int ROWS = WN_K * WN_G;  // 384: window t holds dz[:, n + (1 - t) d]
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [ROWS][32]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * WD_COLS;
  const int T = a.T;
  {
    __amdgpu_buffer_rsrc_t z_rs = uniform_buffer_rsrc(a.dz_in + (long)b * WN_G * T, (unsigned)(WN_G * T) * 4u);
    for (int q = wave; q < ROWS / 8; q += 4) {  // 8 rows of 32 floats (1 KiB) per wave instruction
      const int r0 = 8 * q;
      const int t = r0 / WN_G;
      const int f0 = n0 + (1 - t) * a.dil;
      const int ch0 = r0 - t * WN_G;
      if (__builtin_amdgcn_readfirstlane((f0 >= 0 && f0 + WD_COLS <= T) ? 1 : 0)) {
        const unsigned off = (unsigned)((ch0 + (lane >> 3)) * T + f0 + 4 * (lane & 7)) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(z_rs, (lds_ptr_t)(tile + r0 * WD_COLS), 16, off, 0, 0, 0);
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {  // 2 rows per instruction on the per-sample path
          const int f = f0 + l31;
          const unsigned off = (f >= 0 && f < T) ? (unsigned)((ch0 + 2 * rr + lhi) * T + f) * 4u : 0xFFFFFFFCu;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(z_rs, (lds_ptr_t)(tile + (r0 + 2 * rr) * WD_COLS), 4, off, 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int n = n0 + l31;
  const bool n_ok = n < T;
  const float4* w4 = reinterpret_cast<const float4*>(a.w);
  const float* bl = tile + lhi * WD_COLS + l31;
  if (wave < 2) {
    // dx rows [32 wave, +32) over all three windows (K = 384): image [48][2 tiles][lane][4]
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
    const float4* const wa[1] = {w4 + wave * 64 + lane};
    if (a.dx) wn_contract<1>(acc, wa, 2 * 64, ROWS / 8, bl);
    if (a.dx && n_ok) {
      const long ob = (long)b * WN_R * T + n;
      float gv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
        gv[r] = a.go_in ? a.go_in[ob + (long)row * T] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
        a.dx[ob + (long)row * T] = acc[0][r] + gv[r];
      }
    }
  } else if (a.dc) {
    // dc rows over the centre window (K = 128, rows 128..255 of the tile): image [16][3 tiles][lane][4] behind the dx image
    const float4* wc = w4 + (ROWS / 8) * 2 * 64;
    const float* blc = bl + WN_G * WD_COLS;
    const long ob = (long)b * a.aux * T + n;
    // the gradient already accumulated by the later layers (dc_in) is fetched BEFORE the contraction (its latency
    // hides under the MFMAs; it may alias dc: every element is read and written by the same lane)
    if (wave == 2) {
      f32x16 acc[2];
      float pv[2][16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[i][r] = 0.f;
          const int row = i * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
          pv[i][r] = (a.dc_in && n_ok && row < a.aux) ? a.dc_in[ob + (long)row * T] : 0.f;
        }
      const float4* const wa[2] = {wc + lane, wc + 64 + lane};
      wn_contract<2>(acc, wa, 3 * 64, WN_G / 8, blc);
      if (n_ok) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
            if (row < a.aux) a.dc[ob + (long)row * T] = acc[i][r] + pv[i][r];
          }
      }
    } else {
      f32x16 acc[1];
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[0][r] = 0.f;
        const int row = 64 + 8 * (r >> 2) + 4 * lhi + (r & 3);
        pv[r] = (a.dc_in && n_ok && row < a.aux) ? a.dc_in[ob + (long)row * T] : 0.f;
      }
      const float4* const wa[1] = {wc + 2 * 64 + lane};
      wn_contract<1>(acc, wa, 3 * 64, WN_G / 8, blc);
      if (n_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 64 + 8 * (r >> 2) + 4 * lhi + (r & 3);
          if (row < a.aux) a.dc[ob + (long)row * T] = acc[0][r] + pv[r];
        }
      }
    }
  }
}

// backward images, one buffer: [gate image 16 x 2 tiles][dx image 48 x 2 tiles][dc image 16 x 3 tiles] (x 256 floats)
//   gate: row m = gate-output channel, kc < 64: w_skip[kc][m] * skip_mul, else w_out[kc - 64][m] * out_mul
//   dx  : row m = residual channel ci, kc = t * 128 + co: w_dil[co][ci][t]
//   dc  : row m = aux channel (zero for m >= aux), kc = co: w_aux[co][m]
__global__ void wavenet_pack_bwd_kernel(const float* w_dil, const float* s_dil, const float* w_aux, const float* s_aux,
                                        const float* w_skip, const float* s_skip, const float* w_out, const float* s_out,
                                        float* out, int aux, float out_mul, float skip_mul) {
  const int n_gate = 16 * 2 * 256, n_dx = 48 * 2 * 256, n_dc = 16 * 3 * 256;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_gate + n_dx + n_dc; i += gridDim.x * blockDim.x) {
    const int part = i < n_gate ? 0 : (i < n_gate + n_dx ? 1 : 2);
    const int e = part == 0 ? i : (part == 1 ? i - n_gate : i - n_gate - n_dx);
    const int tiles = part == 2 ? 3 : 2;
    const int j = e & 3, lane = (e >> 2) & 63;
    const int tl = (e >> 8) % tiles, q = (e >> 8) / tiles;
    const int m = tl * 32 + (lane & 31);
    const int kc = 2 * (4 * q + j) + (lane >> 5);
    float v = 0.f;
    if (part == 0) {
      v = kc < WN_S ? w_skip[(long)kc * WN_R + m] * (s_skip ? s_skip[kc] : 1.f) * skip_mul
                    : w_out[(long)(kc - WN_S) * WN_R + m] * (s_out ? s_out[kc - WN_S] : 1.f) * out_mul;
    } else if (part == 1) {
      const int t = kc / WN_G, co = kc % WN_G;
      v = w_dil[((long)co * WN_R + m) * WN_K + t] * (s_dil ? s_dil[co] : 1.f);
    } else if (m < aux) {
      v = w_aux[(long)kc * aux + m] * (s_aux ? s_aux[kc] : 1.f);
    }
    out[i] = v;
  }
}


// =====================================================================================================================
// backward, weight path.  The layer's four weight gradients are two "(128 rows) x (N rows) over time" contractions:
//
//   [dW_dil | dW_aux | db_dil] = dz   (128 x T)  x  [x(n - d); x(n); x(n + d); c(n)]  (272 x T)^T     (+ row sums of dz)
//   [dW_skip ; dW_out | db]    = [gs; go] (128)  x  g (64 x T)^T
//
// As separate k = 1 / k = 3 launches of the general weight-gradient kernel they are LDS-DMA-issue bound (64 dword-DMA
// instructions per 64 MFMAs: 34 - 38 TFLOP/s); here a workgroup (8 waves) stages BOTH operands of a 32-column chunk
// through registers -- 16-B global loads, written to LDS time-major ([column][row], odd row stride), so that the
// MFMA operand reads "32 rows at one column" are 32 consecutive words -- and owns the whole 128 x N output: 36 (or 8)
// accumulator tiles.  Two LDS buffers, ONE barrier per chunk: while the matrix phase runs on chunk c, the registers
// holding chunk c + 1 (loads issued a whole matrix phase earlier) are written to the other buffer in the second half
// of the MFMA loop, then the loads of chunk c + 2 are issued.  Operand rows are assigned to the 64-row load slots on
// the host (every source tensor starts at a multiple of 64 rows), so a slot's base pointer, shift and item stride are
// wave-uniform (scalar registers) and a thread keeps ONE row offset for all of them (211 -> 154 vector registers for
// N = 272 in the single-instantiation form; 232 with the two branch-free matrix phases below).  The skip+out
// contraction (N = 64: few MFMAs per byte) runs three workgroups per CU to keep more loads in flight.  Variants
// measured and dropped (profiles/r04_wavenet_wgrad_variants.txt): 12 waves x 3 tiles with register-prefetched LDS
// operands (137 us), the same with the loads interleaved into the matrix phase (132 us: lane-dependent control flow in
// the load code makes the compiler wait with vmcnt(0) after every load).  Slices of the (item, chunk) range write private
// slabs; one reduce kernel sums them in order and scatters to the torch layouts (deterministic).
// =====================================================================================================================
constexpr int WW_COLS = 32;
constexpr int WW_SLOTS = 7;  // 64-row load slots: (128 + 272) / 64 rounded up
struct WwArgs {
  const float* src[WW_SLOTS];  // row 0 of the slot in item 0 of its tensor; NULL = zeros
  int rows[WW_SLOTS];          // valid rows of the slot (<= 64)
  int shift[WW_SLOTS];         // the slot's rows are read at column n + shift
  int item[WW_SLOTS];          // floats between consecutive items of that tensor
  float* slabs;                // [slices][128][NROWS + 1]
  int T, chunks_per_item, chunks_total, chunks_per_slice;
  int maxshift;  // largest |shift| of the slots
};

template <int NROWS>
__global__ __launch_bounds__(512, NROWS > 64 ? 1 : 3) void wavenet_wgrad_kernel(WwArgs a) {
  constexpr int ROWS = WN_G + NROWS;            // staged rows: M operand first, then the N operand
  constexpr int RS2 = ROWS + 1;                 // words between consecutive columns (odd)
  constexpr int NLD = (ROWS + 63) / 64;         // 16-B loads per thread and chunk (one per slot)
  constexpr int NCT = (NROWS + 31) / 32;        // column tiles of the output
  constexpr int NTL = NCT > 2 ? 5 : 1;          // tiles per wave (N = 272: waves 0-3 own 5, waves 4-7 own 4)
  constexpr int BUF = WW_COLS * RS2;            // floats per LDS buffer
  constexpr int STEPS = WW_COLS / 2;
  static_assert(NLD <= WW_SLOTS, "slot table too small");
  extern __shared__ __attribute__((aligned(16))) float tile[];  // 2 x [32][RS2] (+ 32 floats of slack)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = wave & 3, cg = wave >> 2;
  const int ct0 = NCT > 2 ? cg * 5 : cg;
  const int ntl = NCT > 2 ? (cg == 0 ? 5 : NCT - 5) : 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int T = a.T;
  const int c_begin = blockIdx.x * a.chunks_per_slice;
  const int c_end = min(c_begin + a.chunks_per_slice, a.chunks_total);
  const int lrow = tid >> 3, q4 = 4 * (tid & 7);  // this thread's row inside every slot, its 4 columns of the chunk
  const int lane_off = lrow * T;

  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  float4 stage[NLD];
  // chunk c -> registers.  A slot's tensor, shift and item stride are wave-uniform: scalar base + one lane offset
  auto load_chunk = [&](int c) {
    const int b = c / a.chunks_per_item;
    const int n0 = (c - b * a.chunks_per_item) * WW_COLS;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* p = a.src[i];
      const int col = n0 + a.shift[i] + q4;
      if (p != nullptr && lrow < a.rows[i] && n0 + q4 < T) {  // (columns past the item's end stay zero for BOTH operands)
        p += (long)b * a.item[i] + lane_off;
        if (col >= 0 && col + 3 < T) {
          const f4u u = *reinterpret_cast<const f4u*>(p + col);
          v = make_float4(u[0], u[1], u[2], u[3]);
        } else {
          if (col >= 0 && col < T) v.x = p[col];
          if (col + 1 >= 0 && col + 1 < T) v.y = p[col + 1];
          if (col + 2 >= 0 && col + 2 < T) v.z = p[col + 2];
          if (col + 3 >= 0 && col + 3 < T) v.w = p[col + 3];
        }
      }
      stage[i] = v;
    }
  };
  // slot i of the staged chunk -> LDS buffer ``dst`` ([column][row]); rows past ROWS (last slot) are not stored
  auto store_slot = [&](float* dst, int i) {
    const int row = 64 * i + lrow;
    if (64 * (i + 1) <= ROWS || row < ROWS) {
      float* d = dst + q4 * RS2 + row;
      d[0] = stage[i].x;
      d[RS2] = stage[i].y;
      d[2 * RS2] = stage[i].z;
      d[3 * RS2] = stage[i].w;
    }
  };

  f32x16 acc[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  // interior chunk (every column of every slot inside its item: n0 - maxshift >= 0, n0 + 32 + maxshift <= T): one
  // 16-B load per slot, no lane-dependent control flow; lanes past the last slot's rows read its last row (not stored)
  const int lane_off_last = (lrow < a.rows[NLD - 1] ? lrow : a.rows[NLD - 1] - 1) * T;
  auto load_chunk_interior = [&](int b, int n0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const float* p = a.src[i];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p != nullptr) {  // (wave-uniform)
        p += (long)b * a.item[i] + (i == NLD - 1 ? lane_off_last : lane_off);
        const f4u u = *reinterpret_cast<const f4u*>(p + (n0 + a.shift[i] + q4));
        v = make_float4(u[0], u[1], u[2], u[3]);
      }
      stage[i] = v;
    }
  };
  auto load_any = [&](int c) {
    const int b = c / a.chunks_per_item;
    const int n0 = (c - b * a.chunks_per_item) * WW_COLS;
    if (n0 - a.maxshift >= 0 && n0 + WW_COLS + a.maxshift <= T) load_chunk_interior(b, n0);
    else load_chunk(c);
  };
  // matrix phase of one chunk for a wave that owns NT tiles: branch-free; the LDS operands of step st + 1 are read
  // before the MFMAs of step st are issued; in its second half the next chunk (in registers since the previous phase;
  // stale values after the last chunk -- never read) goes to the other buffer
  auto matrix_phase = [&](auto ntc, const float* cur, float* nxt) {
    constexpr int NT = decltype(ntc)::value;
    constexpr int H = STEPS / 2;
    const float* al = cur + lhi * RS2 + rt * 32 + l31;
    const float* bl = cur + lhi * RS2 + WN_G + ct0 * 32 + l31;
    float av = al[0];
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = bl[t * 32];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      float an = 0.f;
      float bn[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bn[t] = 0.f;
      if (st + 1 < STEPS) {
        an = al[2 * (st + 1) * RS2];
#pragma unroll
        for (int t = 0; t < NT; ++t) bn[t] = bl[2 * (st + 1) * RS2 + t * 32];
      }
      __builtin_amdgcn_sched_barrier(0);  // (keep the reads of step st + 1 ahead of this step's MFMAs)
      bsum += av;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
      if (st >= H) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
          if (i * H / NLD == st - H) store_slot(nxt, i);
      }
      __builtin_amdgcn_sched_barrier(0);
      av = an;
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = bn[t];
    }
  };

  if (c_begin < c_end) {
    load_chunk(c_begin);
#pragma unroll
    for (int i = 0; i < NLD; ++i) store_slot(tile, i);
    if (c_begin + 1 < c_end) load_any(c_begin + 1);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int par = (c - c_begin) & 1;
    const float* cur = tile + par * BUF;
    float* nxt = tile + (par ^ 1) * BUF;
    if (NTL == 1) {
      matrix_phase(std::integral_constant<int, 1>{}, cur, nxt);
    } else if (ntl == NTL) {
      matrix_phase(std::integral_constant<int, NTL>{}, cur, nxt);
    } else {
      matrix_phase(std::integral_constant<int, (NTL > 1 ? NTL - 1 : 1)>{}, cur, nxt);
    }
    if (c + 2 < c_end) load_any(c + 2);  // in flight under the next chunk's matrix phase
    __syncthreads();                     // chunk c + 1 is in LDS; every wave is done reading chunk c
  }
  // slab of this slice: [128][NROWS + 1]; D layout col = lane & 31, row = 8 * (r >> 2) + 4 * lhi + (r & 3)
  float* slab = a.slabs + (long)blockIdx.x * WN_G * (NROWS + 1);
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    if (t < ntl) {
      const int col = (ct0 + t) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = rt * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
        if (col < NROWS) slab[(long)m * (NROWS + 1) + col] = acc[t][r];
      }
    }
  }
  bsum += __shfl_xor(bsum, 32, 64);
  if (cg == 0 && lhi == 0) slab[(long)(rt * 32 + l31) * (NROWS + 1) + NROWS] = bsum;
}

// Finish: one workgroup per output row (128 rows of [dil | aux | bias] + 128 rows of [skip or out | bias]) sums the
// row over the slices (3 or 12 slice lanes per column, lane partials added in order: deterministic), then one wave per
// convolution turns the row into the parameter gradients: plain copy (torch layout), or the weight-norm backward
//   w = g v / |v|  =>  dg = <dw, v> / |v| ;  dv = (g / |v|) (dw - v <dw, v> / |v|^2)          (utils of torch.nn.utils.weight_norm)
struct WfArgs {
  const float* slab0;
  const float* slab1;
  int slices;   // of slab0
  int slices1;  // of slab1
  pwg_wavenet_param_grad conv[4];  // dil, aux, skip, out
};
constexpr int WF_N0 = WN_K * WN_R + 80 + 1;  // 273
constexpr int WF_N1 = WN_R + 1;              // 65
constexpr int WF_THREADS = 832;              // 13 waves >= 3 * 273 and >= 12 * 65

__device__ __forceinline__ void wf_finish_row(const pwg_wavenet_param_grad& cv, const float* tot, int n, int row, int lane,
                                              bool tap_major) {
  if (cv.dw == nullptr) return;
  float dwv[3], vv[3];
  float svv = 0.f, sdv = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = lane + 64 * k;  // torch index inside the row: (ci, tap) with tap fastest
    dwv[k] = vv[k] = 0.f;
    if (e < n) {
      dwv[k] = tot[tap_major ? (e % WN_K) * WN_R + e / WN_K : e];
      if (cv.v) {
        vv[k] = cv.v[(long)row * n + e];
        svv += vv[k] * vv[k];
        sdv += vv[k] * dwv[k];
      }
    }
  }
  float c1 = 1.f, c2 = 0.f;
  if (cv.v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      svv += __shfl_xor(svv, o, 64);
      sdv += __shfl_xor(sdv, o, 64);
    }
    const float norm = sqrtf(svv);
    if (lane == 0 && cv.dg) cv.dg[row] = sdv / norm;
    c1 = cv.g[row] / norm;
    c2 = sdv / svv;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = lane + 64 * k;
    if (e < n) cv.dw[(long)row * n + e] = c1 * (dwv[k] - vv[k] * c2);
  }
}

__global__ __launch_bounds__(WF_THREADS) void wavenet_wgrad_finish_kernel(WfArgs a) {
  __shared__ float part[3 * WF_N0];
  __shared__ float tot[WF_N0];
  const int layer = blockIdx.x >> 7, m = blockIdx.x & 127;
  const int cols = layer == 0 ? WF_N0 : WF_N1;
  const int SL = layer == 0 ? 3 : 12;
  const int nsl = layer == 0 ? a.slices : a.slices1;
  const int tid = threadIdx.x;
  const int sl = tid / cols, col = tid - sl * cols;
  const long sstride = (long)WN_G * cols;
  const float* src = (layer == 0 ? a.slab0 : a.slab1) + (long)m * cols + col;
  if (sl < SL) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four loads in flight
    int j = sl;
    for (; j + 3 * SL < nsl; j += 4 * SL) {
      s0 += src[(long)j * sstride];
      s1 += src[(long)(j + SL) * sstride];
      s2 += src[(long)(j + 2 * SL) * sstride];
      s3 += src[(long)(j + 3 * SL) * sstride];
    }
    for (; j < nsl; j += SL) s0 += src[(long)j * sstride];
    part[sl * cols + col] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (tid < cols) {
    float t = part[tid];
    for (int q = 1; q < SL; ++q) t += part[q * cols + tid];
    tot[tid] = t;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  if (layer == 0) {
    if (wave == 0) wf_finish_row(a.conv[0], tot, WN_K * WN_R, m, lane, true);
    if (wave == 1) wf_finish_row(a.conv[1], tot + WN_K * WN_R, 80, m, lane, false);
    if (tid == 128) {
      if (a.conv[0].db) a.conv[0].db[m] = tot[WF_N0 - 1];
      if (a.conv[1].db) a.conv[1].db[m] = tot[WF_N0 - 1];
    }
  } else {
    const pwg_wavenet_param_grad& cv = a.conv[m < WN_S ? 2 : 3];
    const int mm = m < WN_S ? m : m - WN_S;
    if (wave == 0) wf_finish_row(cv, tot, WN_R, mm, lane, false);
    if (tid == 64 && cv.db) cv.db[mm] = tot[WF_N1 - 1];
  }
}

static bool wavenet_ok(const pwg_wavenet_desc* d) {
  if (!d) return false;
  if (d->residual_channels != WN_R || d->gate_channels != WN_G || d->skip_channels != WN_S || d->kernel != WN_K) return false;
  if (d->aux_channels != 80) return false;  // (the one instantiation; other widths: template + a case below)
  if (d->causal) return false;              // (the fused kernels pad symmetrically; causal layers keep the un-fused path)
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
  static const int dbg = getenv("PWG_WN_DBG") ? atoi(getenv("PWG_WN_DBG")) : 0;
  a.dbg = dbg;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  a.vec_ok = al16(x_out) && al16(skips) && al16(skips_out) && al16(z_out) && al16(g_out);
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

size_t pwg_wavenet_packed_weight_bwd_floats(const pwg_wavenet_desc* d) {
  if (!wavenet_ok(d)) return 0;
  return (size_t)(16 * 2 + 48 * 2 + 16 * 3) * 256;
}

int pwg_wavenet_pack_weights_bwd(const pwg_wavenet_desc* d, const float* w_dil, const float* scale_dil, const float* w_aux,
                                 const float* scale_aux, const float* w_skip, const float* scale_skip, const float* w_out,
                                 const float* scale_out, float* packed, void* stream) {
  PWG_REQUIRE(wavenet_ok(d), PWG_ERR_UNSUPPORTED, "wavenet_pack_weights_bwd: unsupported layer geometry");
  PWG_REQUIRE(w_dil && w_aux && w_skip && w_out && packed, PWG_ERR_NULL, "wavenet_pack_weights_bwd: NULL pointer");
  const int total = (int)pwg_wavenet_packed_weight_bwd_floats(d);
  hipLaunchKernelGGL(wavenet_pack_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w_dil, scale_dil,
                     w_aux, scale_aux, w_skip, scale_skip, w_out, scale_out, packed, d->aux_channels, d->out_mul, d->skip_mul);
  PWG_CHECK_LAUNCH("wavenet_pack_weights_bwd");
  return PWG_OK;
}

int pwg_wavenet_gate_backward(const pwg_wavenet_desc* d, const float* z, const float* dx_out, const float* ds_out,
                              const float* packed_bwd, float* dz, float* go, void* stream_) {
  PWG_REQUIRE(wavenet_ok(d), PWG_ERR_UNSUPPORTED, "wavenet_gate_backward: unsupported layer geometry");
  PWG_REQUIRE(z && ds_out && packed_bwd && dz, PWG_ERR_NULL, "wavenet_gate_backward: NULL pointer");
  PWG_REQUIRE((dx_out == nullptr) == (go == nullptr), PWG_ERR_NULL, "wavenet_gate_backward: go goes with dx_out");
  hipStream_t stream = (hipStream_t)stream_;
  WnBwdArgs a = {};
  a.z = z;
  a.dx_out = dx_out;
  a.ds_out = ds_out;
  a.w = packed_bwd;
  a.dz = dz;
  a.go = go;
  a.T = d->t;
  a.dil = d->dilation;
  a.aux = d->aux_channels;
  a.out_mul = d->out_mul;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  a.vec_ok = al16(z) && al16(dz) && al16(go);
  const size_t lds = (size_t)(WN_G * WN_COLS + 4 * 32 * 36) * sizeof(float);
  const double samples = (double)d->batch * d->t;
  maybe_poison_lds(stream);
  {
    ProfScope prof(stream, "wavenet_gate_bwd_kernel", 2.0 * samples * WN_R * (WN_S + WN_R),
                   4.0 * samples * (WN_S + (dx_out ? 2 * WN_R : 0) + 2 * WN_G));
    hipLaunchKernelGGL(wavenet_gate_bwd_kernel, dim3(ceil_div(d->t, WN_COLS), d->batch), dim3(256), lds, stream, a);
  }
  PWG_CHECK_LAUNCH("wavenet_gate_backward");
  return PWG_OK;
}

int pwg_wavenet_data_backward(const pwg_wavenet_desc* d, const float* dz, const float* go, const float* packed_bwd,
                              const float* dc_accum, float* dx, float* dc, void* stream_) {
  PWG_REQUIRE(wavenet_ok(d), PWG_ERR_UNSUPPORTED, "wavenet_data_backward: unsupported layer geometry");
  PWG_REQUIRE(dz && packed_bwd && (dx || dc), PWG_ERR_NULL, "wavenet_data_backward: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  WnBwdArgs a = {};
  a.dz_in = dz;
  a.go_in = go;
  a.w = packed_bwd + 16 * 2 * 256;
  a.dx = dx;
  a.dc = dc;
  a.dc_in = dc ? dc_accum : nullptr;
  a.T = d->t;
  a.dil = d->dilation;
  a.aux = d->aux_channels;
  const size_t lds = (size_t)WN_K * WN_G * WD_COLS * sizeof(float);
  const double samples = (double)d->batch * d->t;
  maybe_poison_lds(stream);
  {
    ProfScope prof(stream, prof_shape_name("wavenet_dgrad_kernel", "B%d T%d d%d", d->batch, d->t, d->dilation),
                   2.0 * samples * WN_G * ((dx ? WN_K * WN_R : 0) + (dc ? d->aux_channels : 0)),
                   4.0 * samples * (WN_G + (dx ? WN_R : 0) + (go ? WN_R : 0) + (dc ? d->aux_channels : 0) + (a.dc_in ? d->aux_channels : 0)));
    hipLaunchKernelGGL(wavenet_dgrad_kernel, dim3(ceil_div(d->t, WD_COLS), d->batch), dim3(256), lds, stream, a);
  }
  PWG_CHECK_LAUNCH("wavenet_data_backward");
  return PWG_OK;
}

// ``per_cu`` workgroups (8 waves each) per CU: 1 for the dil+aux contraction (MFMA-bound, 100 KB of LDS), 3 for the
// skip+out contraction (few MFMAs per byte: more loads in flight)
static int ww_slices(const pwg_wavenet_desc* d, int* chunks_per_slice, int per_cu) {
  const int chunks = d->batch * ceil_div(d->t, WW_COLS);
  int per = ceil_div(chunks, 256 * per_cu);
  if (per < 2) per = chunks >= 2 ? 2 : 1;
  *chunks_per_slice = per;
  return ceil_div(chunks, per);
}

size_t pwg_wavenet_weight_backward_workspace_floats(const pwg_wavenet_desc* d) {
  if (!wavenet_ok(d)) return 0;
  int per;
  const int slices0 = ww_slices(d, &per, 1), slices1 = ww_slices(d, &per, 3);
  return (size_t)slices0 * WN_G * (size_t)(WN_K * WN_R + d->aux_channels + 1) + (size_t)slices1 * WN_G * (WN_R + 1);
}

int pwg_wavenet_weight_backward(const pwg_wavenet_desc* d, const float* dz, const float* x, const float* c, const float* gs,
                                const float* go, const float* g, const pwg_wavenet_param_grad* grads, float* workspace,
                                size_t workspace_floats, void* stream_) {
  PWG_REQUIRE(wavenet_ok(d), PWG_ERR_UNSUPPORTED, "wavenet_weight_backward: unsupported layer geometry");
  PWG_REQUIRE(dz && x && c && gs && g && grads && workspace, PWG_ERR_NULL, "wavenet_weight_backward: NULL pointer");
  PWG_REQUIRE(grads[0].dw && grads[1].dw && grads[2].dw, PWG_ERR_NULL, "wavenet_weight_backward: NULL weight gradient");
  PWG_REQUIRE((go == nullptr) == (grads[3].dw == nullptr), PWG_ERR_NULL, "wavenet_weight_backward: grads[3] goes with go");
  for (int i = 0; i < 4; ++i)
    PWG_REQUIRE((grads[i].v == nullptr) == (grads[i].g == nullptr) && (grads[i].v != nullptr || grads[i].dg == nullptr), PWG_ERR_NULL,
                "wavenet_weight_backward: grads[%d]: v, g (and dg) go together", i);
  PWG_REQUIRE(workspace_floats >= pwg_wavenet_weight_backward_workspace_floats(d), PWG_ERR_WORKSPACE,
              "wavenet_weight_backward: workspace too small");
  PWG_REQUIRE(!d->causal, PWG_ERR_UNSUPPORTED, "wavenet_weight_backward: the causal form is not built");
  hipStream_t stream = (hipStream_t)stream_;
  int per, per1;
  const int slices = ww_slices(d, &per, 1), slices1 = ww_slices(d, &per1, 3);
  constexpr int N0 = WN_K * WN_R + 80, N1 = WN_R;
  float* slab0 = workspace;
  float* slab1 = workspace + (size_t)slices * WN_G * (N0 + 1);
  WwArgs a = {};
  a.T = d->t;
  a.chunks_per_item = ceil_div(d->t, WW_COLS);
  a.chunks_total = d->batch * a.chunks_per_item;
  a.chunks_per_slice = per;
  a.maxshift = d->dilation;
  const double samples = (double)d->batch * d->t;
  // 64-row load slots of one operand tensor: rows [r0, r0 + rows) of ``p`` (B, total_rows, T), read at column n + shift
  auto put = [&](int& slot, const float* p, int rows, int total_rows, int shift) {
    for (int r0 = 0; r0 < rows; r0 += 64, ++slot) {
      a.src[slot] = p ? p + (size_t)r0 * d->t : nullptr;
      a.rows[slot] = rows - r0 < 64 ? rows - r0 : 64;
      a.shift[slot] = shift;
      a.item[slot] = total_rows * d->t;
    }
  };
  maybe_poison_lds(stream);
  {
    int slot = 0;
    put(slot, dz, WN_G, WN_G, 0);
    for (int t = 0; t < WN_K; ++t) put(slot, x, WN_R, WN_R, (t - 1) * d->dilation);
    put(slot, c, d->aux_channels, d->aux_channels, 0);
    a.slabs = slab0;
    const size_t lds = ((size_t)2 * WW_COLS * (WN_G + N0 + 1) + 32) * sizeof(float);
    void (*kern)(WwArgs) = wavenet_wgrad_kernel<N0>;
    if (!lds_limit_is_set(reinterpret_cast<const void*>(kern), lds)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      PWG_REQUIRE(e == hipSuccess, PWG_ERR_LAUNCH, "wavenet_weight_backward: cannot raise the LDS limit: %s", hipGetErrorString(e));
    }
    ProfScope prof(stream, "wavenet_wgrad_kernel<dil+aux>", 2.0 * samples * WN_G * N0, 4.0 * samples * (WN_G + WN_R + d->aux_channels));
    hipLaunchKernelGGL(kern, dim3(slices), dim3(512), lds, stream, a);
    PWG_CHECK_LAUNCH("wavenet_weight_backward");
  }
  {
    for (int i = 0; i < WW_SLOTS; ++i) { a.src[i] = nullptr; a.rows[i] = 0; a.shift[i] = 0; a.item[i] = 0; }
    a.maxshift = 0;
    int slot = 0;
    put(slot, gs, WN_S, WN_S, 0);
    if (go) put(slot, go, WN_R, WN_R, 0); else put(slot, nullptr, WN_R, WN_R, 0);
    put(slot, g, WN_R, WN_R, 0);
    a.slabs = slab1;
    a.chunks_per_slice = per1;
    const size_t lds = ((size_t)2 * WW_COLS * (WN_G + N1 + 1) + 32) * sizeof(float);
    static_assert(((size_t)2 * WW_COLS * (WN_G + N1 + 1) + 32) * sizeof(float) <= 64 * 1024, "skip+out tile fits the default LDS limit");
    ProfScope prof(stream, "wavenet_wgrad_kernel<skip+out>", 2.0 * samples * WN_G * N1, 4.0 * samples * (WN_G + WN_R));
    hipLaunchKernelGGL(wavenet_wgrad_kernel<N1>, dim3(slices1), dim3(512), lds, stream, a);
    PWG_CHECK_LAUNCH("wavenet_weight_backward");
  }
  {
    WfArgs f = {};
    f.slab0 = slab0;
    f.slab1 = slab1;
    f.slices = slices;
    f.slices1 = slices1;
    for (int i = 0; i < 4; ++i) f.conv[i] = grads[i];
    ProfScope prof(stream, "wavenet_wgrad_finish_kernel", 0, 4.0 * WN_G * ((double)slices * (N0 + 1) + (double)slices1 * (N1 + 1)));
    hipLaunchKernelGGL(wavenet_wgrad_finish_kernel, dim3(2 * WN_G), dim3(WF_THREADS), 0, stream, f);
    PWG_CHECK_LAUNCH("wavenet_wgrad_finish");
  }
  return PWG_OK;
}

}  // extern "C"
