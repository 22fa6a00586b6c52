// wgrad_k1.hip -- weight gradient of 1 x 1 convolutions with few channels on gfx950 (HBM-bound; see the kernel's header).
#include "common.h"

#include <stdlib.h>

namespace pwg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// Weight gradient of 1 x 1 convolutions with few channels (MelGAN's residual stacks: C = 48 / 96 per stack, two per
// stack and step): dW[o][i] = sum_{b,n} G[b][o][n] * act(X[b][i][n]).  24 flop per byte at C = 96, 12 at C = 48 -- at
// or below the ridge, so the bound is the ONE read of G and X.  On the general kernel above these layers ran at 0.9 -
// 1.5 TB/s (round 6, profiles/r06_wgrad_k1.txt): its 64 x 64 tile reads every operand row twice at C = 96 and pads 48
// to 64, and its rotation-swizzled tiles are staged by dword LDS-DMA pieces (32 DMA instructions per wave next to 32
// MFMAs per chunk).  Here one workgroup owns the WHOLE (padded) Co x Ci output: a chunk = 64 reduction columns of all
// Co + Ci rows, fetched with 16-B global loads into registers one chunk ahead (rows are contiguous along n), the
// activation applied and the bias row sums taken on the way, written to LDS with ds_write_b128 (row stride 68 floats:
// 16-B aligned, ds_read_b64 operand reads 2-way conflicted -- there are 2 * NB of them per 2 * NB^2 MFMAs); the four waves
// split the chunk's columns (16 each) and run all NB x NB accumulator blocks, so nothing is read twice (C <= 96).  A lane's b64
// read at column c + 2 * (lane / 32) feeds two MFMAs (reduction pairs {c, c + 2} and {c + 1, c + 3}; A and B use the
// same pairing).  At the end the four waves' partial blocks are added in wave order through the LDS and written as one
// tap-major slab (+ bias row) per workgroup; the usual finishers sum the slabs in order (deterministic, no atomics).
// ---------------------------------------------------------------------------
struct K1Args {
  const float* g;  // (B, co, n_cols)
  const float* x;  // (B, ci, n_cols)
  float* slabs;
  long slab_stride, slab_elems;
  int co, ci, n_cols, batch;
  int chunks_per_item, chunks_total, chunks_per_block;
  float slope_x;
  int write_bias;
  unsigned g_bytes, x_bytes;  // (tensors are below 4 GiB, checked by the host: 32-bit byte offsets)
};

// NBO x NBI: 32-row blocks of G (output channels) and of X (input channels) per workgroup; the four waves form a WK x WI
// grid: WK slices of the chunk's columns, WI slices of the X blocks.  KC = reduction columns per chunk.
//   <NB, NB, 4, 1, 64>: C <= 32 NB <= 96 -- every wave runs all NB x NB blocks on a quarter of the columns (see above);
//   <3, 6, 2, 2, 32>  : 96 < C <= 192 (MB-MelGAN's first stack) -- the whole 192 x 192 output is 36 blocks, too many
//                       for one workgroup's registers: blockIdx.y halves the output channels (X is then read twice:
//                       1.5 x the minimal traffic), a wave runs 3 x 3 blocks on half of the columns.
template <int NBO, int NBI, int WK, int WI, int KC>
__global__ __launch_bounds__(256, 2) void wgrad_k1_kernel(K1Args a) {
  static_assert(WK * WI == 4 && NBI % WI == 0, "four waves");
  constexpr int S = KC + 4;               // LDS row stride (floats): 16-B aligned rows, operand reads 2-way conflicted
  constexpr int GROWS = 32 * NBO, XROWS = 32 * NBI;
  constexpr int R4 = KC / 4;              // 16-B pieces per row and chunk
  constexpr int RPJ = 256 / R4;           // rows covered by one round of the 256 threads
  constexpr int NJG = GROWS / RPJ, NJX = XROWS / RPJ;  // rounds (16-B pieces per thread and chunk)
  static_assert(GROWS % RPJ == 0 && XROWS % RPJ == 0, "whole rounds");
  constexpr int KW = KC / WK;             // reduction columns per wave and chunk
  constexpr int QB = NBI / WI;            // X blocks per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gs = smem;
  float* xs = smem + GROWS * S;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / WI, wi = wave % WI;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int c4 = tid % R4, r0 = tid / R4;
  const int o_base = blockIdx.y * GROWS;  // first output channel of this workgroup

  // round j of this thread: row r0 + j * RPJ of the G tile / of the X tile (rows past co / ci and columns past the
  // item's end come back as zeros: out-of-range buffer offsets), so every row of the padded tiles is rewritten per chunk
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned OOB = 0xFFFFFFFCu;
  __amdgpu_buffer_rsrc_t g_rs = uniform_buffer_rsrc(a.g, a.g_bytes);
  __amdgpu_buffer_rsrc_t x_rs = uniform_buffer_rsrc(a.x, a.x_bytes);
  const unsigned row_step = (unsigned)(RPJ * a.n_cols) * 4u;
  const int c_begin = blockIdx.x * a.chunks_per_block;
  const int c_end = min(c_begin + a.chunks_per_block, a.chunks_total);
  u32x4 preg[NJG], prex[NJX];
  float bsum[NJG];
#pragma unroll
  for (int j = 0; j < NJG; ++j) bsum[j] = 0.f;

  auto fetch = [&](int b, int n0) {
    const int n = n0 + 4 * c4;
    const bool col_ok = n < a.n_cols;
    const unsigned gv = (unsigned)((b * a.co + o_base + r0) * a.n_cols + n) * 4u;
    const unsigned xv = (unsigned)((b * a.ci + r0) * a.n_cols + n) * 4u;
#pragma unroll
    for (int j = 0; j < NJG; ++j) {
      const int row = o_base + r0 + j * RPJ;
      preg[j] = __builtin_amdgcn_raw_buffer_load_b128(g_rs, (col_ok && row < a.co) ? gv + j * row_step : OOB, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      const int row = r0 + j * RPJ;
      prex[j] = __builtin_amdgcn_raw_buffer_load_b128(x_rs, (col_ok && row < a.ci) ? xv + j * row_step : OOB, 0, 0);
    }
  };
  auto stash = [&]() {
    // (native vector types throughout: with HIP's float4 -- a union of a vector and four scalars -- hipcc kept only the
    // .x lane of the activated X pieces, profiles/r06_wgrad_k1.txt)
#pragma unroll
    for (int j = 0; j < NJG; ++j) {
      const f32x4 v = __builtin_bit_cast(f32x4, preg[j]);
      bsum[j] += (v[0] + v[1]) + (v[2] + v[3]);
      *reinterpret_cast<f32x4*>(gs + (r0 + j * RPJ) * S + 4 * c4) = v;
    }
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      f32x4 v = __builtin_bit_cast(f32x4, prex[j]);
      v = __builtin_elementwise_max(v, v * a.slope_x);  // LeakyReLU for 0 <= slope <= 1 (1 = none, 0 = ReLU)
      *reinterpret_cast<f32x4*>(xs + (r0 + j * RPJ) * S + 4 * c4) = v;
    }
  };

  f32x16 acc[NBO][QB];
#pragma unroll
  for (int p = 0; p < NBO; ++p)
#pragma unroll
    for (int q = 0; q < QB; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.f;

  int b = c_begin / a.chunks_per_item;
  int n0 = (c_begin - b * a.chunks_per_item) * KC;
  if (c_begin < c_end) fetch(b, n0);
  const float* ga = gs + l31 * S + wk * KW + 2 * lhi;
  const float* xa = xs + (wi * QB * 32 + l31) * S + wk * KW + 2 * lhi;
  for (int c = c_begin; c < c_end; ++c) {
    __syncthreads();  // the previous chunk's operand reads are done
    stash();
    __syncthreads();
    if (c + 1 < c_end) {
      n0 += KC;
      if (n0 >= a.chunks_per_item * KC) {
        n0 = 0;
        ++b;
      }
      fetch(b, n0);
    }
#pragma unroll 1
    for (int kk = 0; kk < KW; kk += 4) {
      f32x2 av[NBO], bv[QB];
#pragma unroll
      for (int p = 0; p < NBO; ++p) av[p] = *reinterpret_cast<const f32x2*>(ga + p * 32 * S + kk);
#pragma unroll
      for (int q = 0; q < QB; ++q) bv[q] = *reinterpret_cast<const f32x2*>(xa + q * 32 * S + kk);
#pragma unroll
      for (int p = 0; p < NBO; ++p)
#pragma unroll
        for (int q = 0; q < QB; ++q) {
          acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p][0], bv[q][0], acc[p][q], 0, 0, 0);
          acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p][1], bv[q][1], acc[p][q], 0, 0, 0);
        }
    }
  }

  // ---- bias row: the R4 consecutive lanes of a row add their partial sums (fixed shuffle tree)
  float* slab = a.slabs + (long)blockIdx.x * a.slab_stride;
  if (a.write_bias) {
#pragma unroll
    for (int j = 0; j < NJG; ++j) {
      float s = bsum[j];
#pragma unroll
      for (int o = R4 / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const int row = o_base + r0 + j * RPJ;
      if (c4 == 0 && row < a.co) slab[a.slab_elems + row] = s;
    }
  }
  // ---- the WK column slices' partial blocks, added in slice order through the LDS (one block row at a time: (WK - 1)
  // x WI waves x QB x 1024 floats fit the operand tiles), then stored by the waves of slice 0: D layout
  // col = lane & 31 (-> i), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (-> o)
  static_assert((WK - 1) * WI * QB * 1024 <= (GROWS + XROWS) * S, "reduction scratch must fit the operand tiles");
#pragma unroll
  for (int p = 0; p < NBO; ++p) {
    __syncthreads();
    if (wk > 0) {
#pragma unroll
      for (int q = 0; q < QB; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) smem[(((wk - 1) * WI + wi) * QB + q) * 1024 + r * 64 + lane] = acc[p][q][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        const int i = (wi * QB + q) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[p][q][r];
#pragma unroll
          for (int w = 0; w < WK - 1; ++w) v += smem[((w * WI + wi) * QB + q) * 1024 + r * 64 + lane];
          const int o = o_base + p * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (o < a.co && i < a.ci) slab[(long)o * a.ci + i] = v;
        }
      }
    }
  }
}

}  // namespace pwg

using namespace pwg;

// 1 x 1 convolutions with <= 192 channels on either side and a long reduction (wgrad_k1_kernel): `d` flattened;
// PWG_WG_K1=0 disables the path (A/B against the general kernel)
bool pwg::k1_wgrad_applicable(const pwg_conv1d_desc* d) {
  static const bool on = !(getenv("PWG_WG_K1") && atoi(getenv("PWG_WG_K1")) == 0);
  return on && !d->transposed && d->groups == 1 && d->kernel == 1 && d->stride == 1 && d->width == 1 && d->pad_left == 0 &&
         d->t_in == d->t_out && d->pad_mode == PWG_PAD_ZERO && d->c_in >= 8 && d->c_out >= 8 && d->c_in <= 192 &&
         d->c_out <= 192 && (d->t_out & 3) == 0 && (long)d->batch * d->t_out >= 32768;
}
// One slab per workgroup: at least one workgroup per CU (C = 96 at the C4 batch with 135 workgroups: 63 us, half the
// chip idle), up to two when the slabs (written once, read once by the finisher) stay below a tenth of the operand bytes.
static const int K1_MIN_WGS = 256, K1_MAX_WGS = 512;
static bool k1_wide(const pwg_conv1d_desc* d) { return d->c_in > 96 || d->c_out > 96; }
static int k1_chunk_cols(const pwg_conv1d_desc* d) { return k1_wide(d) ? 32 : 64; }
int pwg::k1_wgrad_slabs(const pwg_conv1d_desc* d) {
  const int chunks = d->batch * ceil_div(d->t_out, k1_chunk_cols(d));
  const double in_bytes = 4.0 * d->batch * (double)d->t_out * (d->c_in + d->c_out);
  const double slab_bytes = 4.0 * ((double)d->c_out * d->c_in + d->c_out);
  int wgs = (int)(0.1 * in_bytes / slab_bytes);
  if (wgs > K1_MAX_WGS) wgs = K1_MAX_WGS;
  if (wgs < K1_MIN_WGS) wgs = K1_MIN_WGS;
  if (k1_wide(d)) wgs = K1_MIN_WGS / ceil_div(d->c_out, 96);  // (blockIdx.y halves the output channels; 148 KB per slab at C = 192)
  if (wgs > chunks) wgs = chunks;
  const int per = ceil_div(chunks, wgs);
  return ceil_div(chunks, per);
}


int pwg::k1_wgrad_launch(const pwg_conv1d_desc* d, const float* x, const float* dy, float* slabs, long slab_stride, int nslabs,
                         float slope_x, bool write_bias, hipStream_t stream) {
  K1Args k;
  k.g = dy;
  k.x = x;
  k.slabs = slabs;
  k.slab_elems = (long)d->c_out * d->c_in;
  k.slab_stride = slab_stride;
  k.co = d->c_out;
  k.ci = d->c_in;
  k.n_cols = d->t_out;
  k.batch = d->batch;
  const int kc = k1_chunk_cols(d);
  k.chunks_per_item = ceil_div(d->t_out, kc);
  k.chunks_total = d->batch * k.chunks_per_item;
  k.chunks_per_block = ceil_div(k.chunks_total, nslabs);
  k.slope_x = slope_x;
  k.write_bias = write_bias ? 1 : 0;
  k.g_bytes = (unsigned)((long)d->batch * d->c_out * d->t_out * 4);
  k.x_bytes = (unsigned)((long)d->batch * d->c_in * d->t_out * 4);
  PWG_REQUIRE(nslabs >= 2 && (long)nslabs * k.chunks_per_block >= k.chunks_total, PWG_ERR_UNSUPPORTED,
              "conv1d_backward_weight: 1 x 1 path with %d slabs for %d chunks", nslabs, k.chunks_total);
  const double cols = (double)d->batch * d->t_out;
  maybe_poison_lds(stream);
  ProfScope prof(stream, prof_shape_name("wgrad_k1_kernel", "B%d Co%d Ci%d cols%d slabs%d", d->batch, d->c_out, d->c_in,
                                         d->t_out, nslabs),
                 2.0 * cols * d->c_out * d->c_in, 4.0 * cols * (d->c_in + d->c_out));
  if (k1_wide(d)) {
    const size_t lds = (size_t)(96 + 192) * (32 + 4) * sizeof(float);
    hipLaunchKernelGGL((wgrad_k1_kernel<3, 6, 2, 2, 32>), dim3(nslabs, ceil_div(d->c_out, 96)), dim3(256), lds, stream, k);
  } else {
    const int nb = ceil_div(d->c_in > d->c_out ? d->c_in : d->c_out, 32);
    const size_t lds = (size_t)2 * 32 * nb * (64 + 4) * sizeof(float);
    if (nb == 1) hipLaunchKernelGGL((wgrad_k1_kernel<1, 1, 4, 1, 64>), dim3(nslabs), dim3(256), lds, stream, k);
    else if (nb == 2) hipLaunchKernelGGL((wgrad_k1_kernel<2, 2, 4, 1, 64>), dim3(nslabs), dim3(256), lds, stream, k);
    else hipLaunchKernelGGL((wgrad_k1_kernel<3, 3, 4, 1, 64>), dim3(nslabs), dim3(256), lds, stream, k);
  }
  PWG_CHECK_LAUNCH("wgrad_k1");
  return PWG_OK;
}
