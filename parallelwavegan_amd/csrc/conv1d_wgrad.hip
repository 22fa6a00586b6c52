// conv1d_wgrad.hip -- weight gradient of the conv1d family on gfx950 fp32 MFMA.
//
//   dW[o][i][k] = sum_{b,n}  G[b][o][n] * act(X[b][i][ xoff(n) + (k*dil - pad)*W ])
//
// with G the gradient w.r.t. the convolution output and X its input (roles are swapped by the
// host for ConvTranspose1d, whose weight gradient is the same expression with x and dy
// exchanged).  GEMM view per (group, tap): rows = o, cols = i, reduction = (b, n):
//   A[o][n]  <- G tile   (LDS, rotation-swizzled so that 32 rows hit 32 banks)
//   B[n][i]  <- X tile   (LDS, odd row stride; every tap is a shifted read of the same tile)
// One workgroup = 2x2 waves = 64 o x 64 i x TG taps (or 32 x 32 with the taps split over the waves),
// looping over its slice of the (b,n) range in chunks of 32/64/128 columns with LDS-DMA double
// buffering.  Every slice writes a private slab in torch weight layout (plus, for plain convolutions,
// the bias-gradient row sums of its G tiles); reduce_slabs_* adds the slabs in a fixed order:
// deterministic, no atomics.
#include "common.h"

#include <stdlib.h>

namespace pwg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct WgArgs {
  const float* g;   // "output gradient" role   (B, CO, n_cols)
  const float* x;   // "input" role             (B, CI, x_len)
  float* dw;        // slab base: (CO, CI/groups, K) torch layout per reduction slice
  int co_g, ci_g, groups;
  int k, k0_step;   // taps, taps per tap-group
  int stride, dil, pad, width;
  int n_cols;       // columns per batch item in G  (rows_out * width)
  int x_len;        // samples per channel row in X (rows_in * width)
  int batch;
  int chunks_per_item, chunks_total, chunks_per_block;
  int xs_stride;    // odd
  float* db;        // bias-gradient rows of the slabs (nullptr: not fused); same slab stride as dw
  long slab_elems;  // elements of dW
  long slab_stride; // floats between consecutive slabs (dW + fused bias row)
  float slope_g, slope_x;  // branch-free pre-activation slopes (1 = none)
  unsigned g_bytes, x_bytes;
  int rows_x4;    // MODE 4: stage the X rows with 16-B DMAs (one instruction per row)
  int rows_half;  // MODE 4 (row-aligned (k,1) chunks): G rows per lane half of a chunk (a chunk = 2 * rows_half rows x width)
  int tap_major;  // slab layout: 1 = [tap][o][i] (coalesced partial-sum stores, the finishers permute), 0 = torch (o, i, tap)
  int dbg;  // timing experiments only (PWG_WG_DBG env): 1 = no DMA after the first chunk, 2 = no waits / barriers, 4 = no stores
};

// TT = reduction columns per chunk (32 / 64 / 128): long sequences under a narrow tile want long
// chunks so that a barrier interval carries enough MFMA work.

// WIN = false: one X tile per chunk shared by all taps (taps are shifted reads).
// WIN = true : one 32-column window per tap (rotation-swizzled like the G tile); used when the taps
//              are far apart ((taps-1)*dilation >> 32, e.g. PWG dilations up to 512) so that the
//              shared tile would not fit the LDS.  Requires stride == 1 and width == 1.
// SMALL = false: workgroup tile 64 o x 64 i, 2x2 waves, every wave runs all TG taps of the block.
// SMALL = true : workgroup tile 32 o x 32 i (narrow layers: C = 32, grouped convs, 1-channel edge
//              layers); the 4 waves share the tiles and split the block's 4*TG taps between them.
// MODE 0: width 1, stride 1, no pre-activation; 1: width 1, stride 1, leaky-ReLU/ReLU applied to the
//      operands on the fly; 3: width 1, any stride; 2: width > 1 ((k,1) Conv2d of the period
//      discriminators).  ACT23 (modes 2 / 3 only): evaluate the activation formula there too -- round 6: the
//      discriminators hand both operands over already activated (slopes 1), and the 2 x (TG + 1) VALU operations per
//      reduction step of the always-on formula were a third of the strided layers' loop (profiles/r06_wgrad_strided.txt).
// MODE 4 (round 6): width > 1 with ROW-ALIGNED chunks.  Mode 2 walks the reduction columns n = h * W + w two at a time
//      (lane halves n, n + 1): with odd W the row wrap falls on different steps in different lanes, so every step
//      recomputes its X offset per lane (7 VALU) and adds it to every tap pointer (TG VALU) -- 31 VALU operations
//      per 5 MFMAs on the period discriminators' stride-3 layers, 43 TFLOP/s.  Here a chunk is 2 * rows_half whole rows
//      of G: the lower lane half walks the first rows_half rows, the upper half the others, both at the same (row,
//      column) step, so the walk is wave-uniform (scalar registers) and the G tile is two 32-column blocks, one per
//      half (rows_half * W <= 32), read exactly like the stride-1 modes' tile.
// STRIDE3 (mode 3 only): compile-time stride (0 = run time) -- the step offsets of the X reads fold into the LDS
//      instructions as in the stride-1 modes.
template <int TG, bool WIN, bool SMALL, int TT, int MODE, bool ACT23 = true, int STRIDE3 = 0>
__global__ __launch_bounds__(256, (TG > 7 ? 1 : 2)) void conv1d_wgrad_kernel(WgArgs a) {
  constexpr int BT = SMALL ? 32 : 64;          // tile rows (o) and columns (i)
  constexpr int TAPS_BLOCK = SMALL ? 4 * TG : TG;
  constexpr bool ROWS = MODE == 4;
  static_assert(!ROWS || (TT == 64 && !WIN), "row-aligned chunks: two 32-column G blocks, shared X tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int XS = a.xs_stride;
  const int buf_floats = BT * TT + (WIN ? TAPS_BLOCK * BT * TT : BT * XS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_o = SMALL ? 0 : wave >> 1, wave_i = SMALL ? 0 : wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int otiles = (a.co_g + BT - 1) / BT, itiles = (a.ci_g + BT - 1) / BT;
  int by = blockIdx.y;
  const int it = by % itiles;
  by /= itiles;
  const int ot = by % otiles;
  const int grp = by / otiles;
  const int o0 = ot * BT, i0 = it * BT;
  const int k0 = blockIdx.z * TAPS_BLOCK;
  const int ntaps_block = min(TAPS_BLOCK, a.k - k0);
  const int t0 = SMALL ? wave * TG : 0;                       // this wave's first tap within the block
  const int ntaps = max(0, min(TG, ntaps_block - t0));        // and how many it owns
  const int W = a.width;
  const int co_tot = a.co_g * a.groups, ci_tot = a.ci_g * a.groups;

  const int c_begin = blockIdx.x * a.chunks_per_block;
  const int c_end = min(c_begin + a.chunks_per_block, a.chunks_total);

  __amdgpu_buffer_rsrc_t g_rs = uniform_buffer_rsrc(a.g, a.g_bytes);
  __amdgpu_buffer_rsrc_t x_rs = uniform_buffer_rsrc(a.x, a.x_bytes);
  const unsigned OOB = 0xFFFFFFFCu;

  f32x16 acc[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // Per-lane, chunk-invariant pieces of the rotated-tile DMA addresses: piece q of a wave covers tile
  // rows (2j, 2j+1) of column block cb with jj = 4q + wave = cb * (BT/2) + j.
  constexpr int NQ = (TT / 32) * (BT / 2) / 4;
  int g_rel[NQ], x_rel[WIN ? NQ : 1];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int jj = 4 * q + wave;
    const int cb = jj / (BT / 2), j = jj - cb * (BT / 2);
    const int row = 2 * j + lhi;
    const int nrel = cb * 32 + ((l31 - row) & 31);
    g_rel[q] = row * a.n_cols + nrel;
    if (WIN) x_rel[q] = row * a.x_len + nrel;
  }
  const bool o_full = o0 + BT <= a.co_g, i_full = i0 + BT <= a.ci_g;

  // Stage chunk (b, n0) into buf.  Interior chunks (tile fully inside the tensors) take a path with
  // one VALU op per DMA instruction; edge chunks go through the fully predicated one.
  auto issue = [&](int b, int n0, float* buf) {
    float* gs = buf;
    float* xs = buf + BT * TT;
    // ---- G tile: [TT/32 column blocks][BT rows][32]; element (o, n) at column (n + o) & 31 of its block
    // (tensors are below 4 GiB, checked by the host: 32-bit element indices)
    const int g_base = (b * co_tot + grp * a.co_g + o0) * a.n_cols + n0;
    if (ROWS) {
      // two column blocks: block cb holds the rows_half * W columns of the chunk's rows [cb * rows_half, (cb + 1) * rows_half)
      const int hc = a.rows_half * a.width;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int jj = 4 * q + wave;
        const int cb = jj / (BT / 2), j = jj - cb * (BT / 2);
        const int row = 2 * j + lhi;
        const int nrel = (l31 - row) & 31;
        const int nc = cb * hc + nrel;
        const bool ok = o0 + row < a.co_g && nrel < hc && n0 + nc < a.n_cols;
        const unsigned off = ok ? (unsigned)(g_base + row * a.n_cols + nc) * 4u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_t)(gs + cb * (BT * 32) + j * 64), 4, off, 0, 0, 0);
      }
    } else if (o_full && n0 + TT <= a.n_cols) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int jj = 4 * q + wave;
        const int cb = jj / (BT / 2), j = jj - cb * (BT / 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_t)(gs + cb * (BT * 32) + j * 64), 4,
                                                 (unsigned)(g_base + g_rel[q]) * 4u, 0, 0, 0);
      }
    } else {
#pragma unroll 1
      for (int jj = wave; jj < (TT / 32) * (BT / 2); jj += 4) {
        const int cb = jj / (BT / 2), j = jj - cb * (BT / 2);
        const int row = 2 * j + lhi;
        const int nrel = cb * 32 + ((l31 - row) & 31);
        const bool ok = o0 + row < a.co_g && n0 + nrel < a.n_cols;
        const unsigned off = ok ? (unsigned)(g_base + row * a.n_cols + nrel) * 4u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_t)(gs + cb * (BT * 32) + j * 64), 4, off, 0, 0, 0);
      }
    }
    const int x_base = (b * ci_tot + grp * a.ci_g + i0) * a.x_len;
    if (WIN) {
      // ---- per-tap windows: [tap][column block][BT][32], element (i, n) at column (n + i) & 31
#pragma unroll 1
      for (int t = 0; t < ntaps_block; ++t) {
        const int fs = n0 + (k0 + t) * a.dil - a.pad;
        float* xt = xs + t * (BT * TT);
        if (i_full && fs >= 0 && fs + TT <= a.x_len) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int jj = 4 * q + wave;
            const int cb = jj / (BT / 2), j = jj - cb * (BT / 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xt + cb * (BT * 32) + j * 64), 4,
                                                     (unsigned)(x_base + fs + x_rel[WIN ? q : 0]) * 4u, 0, 0, 0);
          }
        } else {
#pragma unroll 1
          for (int jj = wave; jj < (TT / 32) * (BT / 2); jj += 4) {
            const int cb = jj / (BT / 2), j = jj - cb * (BT / 2);
            const int row = 2 * j + lhi;
            const int f = fs + cb * 32 + ((l31 - row) & 31);
            const bool ok = i0 + row < a.ci_g && (unsigned)f < (unsigned)a.x_len;
            const unsigned off = ok ? (unsigned)(x_base + row * a.x_len + f) * 4u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xt + cb * (BT * 32) + j * 64), 4, off, 0, 0, 0);
          }
        }
      }
      return;
    }
    // ---- X tile: BT rows, flat range [f0, f0 + L)
    const int h0 = n0 / W;
    int n_last = n0 + TT - 1;
    if (n_last > a.n_cols - 1) n_last = a.n_cols - 1;
    // (row-aligned chunks always stage -- and walk -- all 2 * rows_half rows: rows past the item meet G = 0)
    const int h1 = ROWS ? h0 + 2 * a.rows_half - 1 : n_last / W;
    const int f0 = (h0 * a.stride + k0 * a.dil - a.pad) * W;
    const int L = ((h1 - h0) * a.stride + (ntaps_block - 1) * a.dil + 1) * W;
    if (ROWS && a.rows_x4) {
      // Row-aligned chunks stage 154 .. 196 floats per X row: 3 - 4 dword pieces (256 B each), i.e. 48 - 64 DMA
      // instructions per wave and chunk next to 110 - 160 MFMAs -- issue-bound.  16-B pieces: ONE instruction per row
      // (lane l carries floats [4 l, 4 l + 4) of the row's window; the window start is moved down to a multiple of 4
      // so that no piece straddles the row's first sample; pieces wholly outside the row are out-of-range lanes that
      // land as 0.0; the one piece that may straddle the row's END is repaired before the barrier, see fix_tail).
      // A 4-byte-aligned LDS destination is legal for the 16-B form (tools/probes/glds_x3.hip), so the rows keep their
      // odd stride.
      const int sh = f0 & 3;  // (two's complement: also right for negative f0)
      const int f0a = f0 - sh;
      const int fl = f0a + 4 * lane;
      const bool lane_ok = 4 * lane < L + sh && fl >= 0 && fl < a.x_len;
      int rowoff = x_base + wave * a.x_len + fl;
      if (4 * lane < L + sh) {  // (lanes past the window must not store: their 16 B would land in the next row)
#pragma unroll 4
        for (int r = wave; r < BT; r += 4) {
          const unsigned off = (lane_ok && i0 + r < a.ci_g) ? (unsigned)rowoff * 4u : 0xFFFFFFF0u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS), 16, off, 0, 0, 0);
          rowoff += 4 * a.x_len;
        }
      }
      return;
    }
    const int Lr = (L + 63) & ~63;  // whole DMA pieces (the row stride XS covers them)
    // (full chunks only: the last chunk of an item stages fewer columns than the MFMA loop walks, and the
    // predicated path below zero-fills the rest -- stale LDS there meets G = 0, and 0 * NaN is NaN)
    if (i_full && f0 >= 0 && f0 + Lr <= a.x_len && (ROWS || n0 + TT <= a.n_cols)) {
#pragma unroll 1
      for (int e0 = 0; e0 < Lr; e0 += 64) {
        const int fl = f0 + e0 + lane;
        int rowbase = x_base + wave * a.x_len;
#pragma unroll 4
        for (int r = wave; r < BT; r += 4) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS + e0), 4,
                                                   (unsigned)(rowbase + fl) * 4u, 0, 0, 0);
          rowbase += 4 * a.x_len;
        }
      }
    } else {
#pragma unroll 1
      for (int e0 = 0; e0 < L; e0 += 64) {
        const int f = f0 + e0 + lane;
        const bool lane_ok = (unsigned)f < (unsigned)a.x_len && e0 + lane < L;
#pragma unroll 1
        for (int r = wave; r < BT; r += 4) {
          const int rowbase = x_base + r * a.x_len;
          const unsigned off = (lane_ok && i0 + r < a.ci_g) ? (unsigned)(rowbase + f) * 4u : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS + e0), 4, off, 0, 0, 0);
        }
      }
      // An edge chunk needs fewer columns than the MFMA loop reads (it always walks all TT columns): the
      // rest of the row is zero-filled.  Those columns meet G = 0 in the contraction, but 0 * stale LDS
      // content is NaN whenever that content happens to be a NaN/Inf bit pattern (seen as a rare
      // failure on a cold GPU; tools/probes/wgrad_stale_lds.py reproduces it).
#pragma unroll 1
      for (int e0 = (L + 63) & ~63; e0 < XS - 1; e0 += 64) {
#pragma unroll 1
        for (int r = wave; r < BT; r += 4)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS + e0), 4, OOB, 0, 0, 0);
      }
    }
  };

  // Every wave runs all TG accumulators unconditionally (a per-tap branch would put every MFMA in its
  // own basic block and serialise it behind its LDS read); accumulators beyond the wave's tap count
  // repeat its last tap and are dropped in the epilogue.
  int toff[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t) {
    const int tc = min(t, max(ntaps - 1, 0));
    toff[t] = tc * (WIN ? BT * TT : a.dil * W);
  }
  float bsum = 0.f;
  auto mac_chunk = [&](const float* grow, const float* xrow, int n0, int h0, int orow, int irow) {
    constexpr bool ACT = MODE == 1 || (MODE >= 2 && ACT23);
    constexpr bool S1 = MODE <= 1 && !WIN;  // stride 1, width 1: step offsets fold into the LDS instructions
    constexpr bool S3C = MODE == 3 && STRIDE3 > 0;  // compile-time stride: likewise
    constexpr int STEPS = TT / 2;
    const int rot0 = (lhi + orow) & 31, roti = (lhi + irow) & 31;
    const float* xt[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) xt[t] = xrow + toff[t] + (S1 ? lhi : (S3C ? lhi * STRIDE3 : 0));
    // MODE 2 state of the next load_ops call: column n = n0 + lhi, its in-row position w2 and tile offset xo2
    int w2 = 0, xo2 = 0;
    if (MODE == 2) {
      const int n = n0 + lhi;
      const int h = n / W;
      w2 = n - h * W;
      xo2 = (h - h0) * a.stride * W + w2;
    }
    // operands of one reduction step (2 columns): 1 G value + TG X values per lane
    auto load_ops = [&](int step, float& av, float(&bv)[TG]) {
      const int cb = step >> 4;  // 32-column block of columns (2*step, 2*step+1)
      av = grow[cb * (BT * 32) + ((2 * step + rot0) & 31)];
      int xo;
      if (WIN) {
        xo = cb * (BT * 32) + ((2 * step + roti) & 31);
      } else if (S1) {
        xo = 2 * step;
      } else if (S3C) {
        xo = 2 * step * STRIDE3;
      } else if (MODE == 3) {
        xo = (2 * step + lhi) * a.stride;
      } else {
        // (k,1) Conv2d: column n = h * W + w lives at (h - h0) * stride * W + w of the tile.  The calls
        // walk the steps in order, so (w, xo) are carried: +2 columns per step, one conditional wrap
        // (W >= 2) instead of an integer division per step.
        xo = xo2;
        w2 += 2;
        xo2 += 2;
        if (w2 >= W) {
          w2 -= W;
          xo2 += (a.stride - 1) * W;
        }
      }
#pragma unroll
      for (int t = 0; t < TG; ++t) bv[t] = xt[t][xo];
    };
    auto mma = [&](float av, float(&bv)[TG]) {
      bsum += av;  // fused bias gradient: row sums of G (only used where G is the raw output gradient)
      // pre-activation with 0 <= slope <= 1 (1 = identity, 0 = ReLU): max(v, slope * v), 2 VALU ops
      if (ACT) av = __builtin_fmaxf(av, av * a.slope_g);
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        float v = bv[t];
        if (ACT) v = __builtin_fmaxf(v, v * a.slope_x);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, v, acc[t], 0, 0, 0);
      }
    };
    // ping-pong: the next step's LDS reads are issued before this step's MFMAs
    float a0, a1, b0[TG], b1[TG];
    load_ops(0, a0, b0);
#pragma unroll 4
    for (int step = 0; step < STEPS; step += 2) {
      load_ops(step + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMA group they hide under
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load_ops(step + 2 < STEPS ? step + 2 : step, a0, b0);  // (last pair: harmless re-read)
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // MODE 4: one chunk = rows_half * W reduction steps; step (hr, w) pairs row hr of the lower half with row
  // rows_half + hr of the upper half, column w.  u = X-tile offset of the step (wave-uniform), the G column is the step
  // index inside the lane half's 32-column block.
  auto mac_rows = [&](const float* gs, const float* xrow, int orow, int irow) {
    constexpr bool ACT = ACT23;
    const int W_ = a.width;
    const int steps = a.rows_half * W_;
    const int row_skip = (a.stride - 1) * W_;
    const float* gb = gs + lhi * (BT * 32) + orow * 32;
    const int rot = orow & 31;
    const float* xt[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) xt[t] = xrow + toff[t] + lhi * (a.rows_half * a.stride * W_);
    int si = 0, u = 0, w = 0;  // the step the next load_ops call reads (wave-uniform)
    auto load_ops = [&](float& av, float(&bv)[TG]) {
      av = gb[(si + rot) & 31];
#pragma unroll
      for (int t = 0; t < TG; ++t) bv[t] = xt[t][u];
      if (si + 1 < steps) {  // (the call past the last step re-reads it: harmless, never used)
        ++si;
        ++u;
        if (++w == W_) {
          w = 0;
          u += row_skip;
        }
      }
    };
    auto mma = [&](float av, float(&bv)[TG]) {
      bsum += av;
      if (ACT) av = __builtin_fmaxf(av, av * a.slope_g);
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        float v = bv[t];
        if (ACT) v = __builtin_fmaxf(v, v * a.slope_x);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, v, acc[t], 0, 0, 0);
      }
    };
    float a0, a1, b0[TG], b1[TG];
    load_ops(a0, b0);
    int s = 0;
#pragma unroll 2
    for (; s + 2 <= steps; s += 2) {
      load_ops(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load_ops(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s < steps) mma(a0, b0);  // odd step count: the last step is in (a0, b0)
  };

  // (b, n0) of the chunk being computed and of the one being staged, carried without divisions
  const int chunk_cols = ROWS ? 2 * a.rows_half * a.width : TT;
  int cb_b = c_begin / a.chunks_per_item;
  int cb_n0 = (c_begin - cb_b * a.chunks_per_item) * chunk_cols;
  int nb_b = cb_b, nb_n0 = cb_n0;
  auto advance = [&](int& b, int& n0) {
    n0 += chunk_cols;
    if (n0 >= a.chunks_per_item * chunk_cols) {
      n0 = 0;
      ++b;
    }
  };
  if (c_begin < c_end) issue(nb_b, nb_n0, smem);
  for (int c = c_begin; c < c_end; ++c) {
    const int par = (c - c_begin) & 1;
    float* buf = smem + par * buf_floats;
    if (!(a.dbg & 2)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (ROWS && a.rows_x4 && (a.x_len & 3)) {
        // fix_tail: the 16-B piece that straddles the end of an X row carries the first samples of the next row where
        // the convolution's bottom padding (zeros) belongs; every wave overwrites them in the rows it staged itself
        // (its own DMAs have landed: vmcnt(0)), the barrier below publishes the stores.
        const int f0a = ((cb_n0 / W) * a.stride + k0 * a.dil - a.pad) * W & ~3;
        const int es = a.x_len - f0a;  // first float past the row, relative to the window
        const int rows_l = lane >> 2, q = lane & 3;
        const int r = wave + 4 * rows_l;  // 16 rows per wave (BT = 64) / 8 (BT = 32)
        if (es > 0 && es < a.xs_stride - 4 && (es & 3) && r < BT && q >= (es & 3))
          (buf + BT * TT)[r * XS + (es & ~3) + q] = 0.f;
      }
      __syncthreads();
    }
    if (c + 1 < c_end && !(a.dbg & 1)) {
      advance(nb_b, nb_n0);
      issue(nb_b, nb_n0, smem + (par ^ 1) * buf_floats);
    }
    const float* gs = buf;
    const float* xs = buf + BT * TT;
    const int n0 = cb_n0;
    advance(cb_b, cb_n0);
    const int h0 = n0 / W;
    int x_shift = 0;
    if (ROWS && a.rows_x4) x_shift = ((h0 * a.stride + k0 * a.dil - a.pad) * W) & 3;
    const int orow = wave_o * 32 + l31;
    const float* grow = gs + orow * 32;
    const int irow = wave_i * 32 + l31;
    const float* xrow = xs + irow * (WIN ? 32 : XS) + (WIN ? t0 * (BT * TT) : t0 * a.dil * W) + x_shift;
    if (ntaps > 0) {
      if constexpr (ROWS) mac_rows(gs, xrow, orow, irow);
      else mac_chunk(grow, xrow, n0, h0, orow, irow);
    }
  }

  // ---- epilogue: D layout col = lane&31 (-> i), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (-> o).
  // Every reduction slice owns a private slab (torch weight layout) that it fully overwrites with
  // plain stores; slabs are summed by reduce_slabs_kernel (no atomics, deterministic).
  float* slab = a.dw + (long)blockIdx.x * a.slab_stride;
  if (a.db != nullptr && it == 0 && blockIdx.z == 0) {
    // lanes l and l+32 hold the even/odd columns of row (wave_o*32 + l31); every wave of a narrow
    // tile sees the same G rows, the first one writes
    const float rs = bsum + __shfl_down(bsum, 32, 64);
    const int o = o0 + wave_o * 32 + l31;
    if ((SMALL ? wave == 0 : wave_i == 0) && lhi == 0 && o < a.co_g)
      a.db[(long)blockIdx.x * a.slab_stride + grp * a.co_g + o] = rs;
  }
  const int i = i0 + wave_i * 32 + l31;
  if (i < a.ci_g && !(a.dbg & 4)) {
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t < ntaps) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + wave_o * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (o < a.co_g) {
            // tap-major slabs: the 32 lanes of a half-wave (consecutive i) write one 128-B segment; in torch
            // layout they are k floats apart, every store instruction touching 64 different sectors (the
            // slab stores were 30-40 % of the kernel time on the C3 shapes, profiles/r02_wgrad_ablation.txt)
            const long e = a.tap_major ? ((long)(k0 + t0 + t) * co_tot + grp * a.co_g + o) * a.ci_g + i
                                       : (((long)(grp * a.co_g + o)) * a.ci_g + i) * a.k + k0 + t0 + t;
            slab[e] = acc[t][r];
          }
        }
      }
    }
  }
}

// dw[e] = sum_s slabs[s][e].  Few slabs: one thread per element.  Many slabs (narrow layers cut
// into hundreds of reduction slices): 32 elements x 8 slab lanes per workgroup, every lane sums its
// slabs (stride 8) in a fixed order, then the 8 partials are added in a fixed order (deterministic).
// Elements [0, dw_elems) of a slab go to dw, the rest (fused bias row) to db.
// Slabs are tap-major ([tap][o][i], `plane` = rows * ci_g elements per tap): element e of a slab goes to
// dw[(e % plane) * k + e / plane] (torch layout (o, i, tap)) -- read coalesced nslabs times, scattered once.
__device__ __forceinline__ long slab_to_torch(long e, long plane, int k) {
  const long tap = e / plane;
  return (e - tap * plane) * k + tap;
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ dw, float* __restrict__ db,
                                    long dw_elems, long elems, int nslabs, long plane, int k) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < elems; e += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < nslabs; ++j) s += slabs[(long)j * elems + e];
    if (e < dw_elems) dw[slab_to_torch(e, plane, k)] = s;
    else db[e - dw_elems] = s;
  }
}

__global__ __launch_bounds__(256) void reduce_slabs_wide_kernel(const float* __restrict__ slabs,
                                                                float* __restrict__ dw, float* __restrict__ db,
                                                                long dw_elems, long elems, int nslabs, long plane,
                                                                int k) {
  __shared__ float part[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long e = (long)blockIdx.x * 32 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < elems) {
    int j = sl;
    for (; j + 24 < nslabs; j += 32) {
      s0 += slabs[(long)j * elems + e];
      s1 += slabs[(long)(j + 8) * elems + e];
      s2 += slabs[(long)(j + 16) * elems + e];
      s3 += slabs[(long)(j + 24) * elems + e];
    }
    for (; j < nslabs; j += 8) s0 += slabs[(long)j * elems + e];
  }
  part[sl][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && e < elems) {
    float s = part[0][el];
#pragma unroll
    for (int q = 1; q < 8; ++q) s += part[q][el];
    if (e < dw_elems) dw[slab_to_torch(e, plane, k)] = s;
    else db[e - dw_elems] = s;
  }
}

// db[c] = sum over (item, time) of dy[b][c][:]: ONE workgroup of 1024 lanes per channel walks the items in order
// (lane-strided partial sums, fixed shuffle / LDS tree), so the result is bit-reproducible -- the round-4 kernel
// combined per-segment partial sums with fp32 atomics, one of the three sources of run-to-run drift of a training
// step.  Used only by ConvTranspose1d layers and bias-only calls (<= 128 KB of dy per channel at the training
// shapes); plain convolutions get their bias gradient from the weight-gradient kernel's fused row.
__global__ __launch_bounds__(1024) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int channels,
                                                         int n, int batch) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long plane = (long)channels * n;
  const float* pc = dy + (long)c * n;
  // the (item, time) pairs of this channel as ONE index space (row lengths are often < 1024 * 4 at the coarse layers:
  // a per-item loop would leave most lanes idle and serialise the items' latencies); four independent loads per trip
  if ((n & 3) == 0 && (((unsigned long long)dy) & 15) == 0) {  // rows start 16-B aligned
    const int n4 = n >> 2;
    const long tot = (long)batch * n4;
    auto ld = [&](long idx) {
      const long bb = idx / n4;
      const float4 a = *reinterpret_cast<const float4*>(pc + bb * plane + ((idx - bb * n4) << 2));
      return (a.x + a.y) + (a.z + a.w);
    };
    long i = threadIdx.x;
    for (; i + 3072 < tot; i += 4096) {
      const float a = ld(i), e = ld(i + 1024), f = ld(i + 2048), g = ld(i + 3072);
      s0 += a;
      s1 += e;
      s2 += f;
      s3 += g;
    }
    for (; i < tot; i += 1024) s0 += ld(i);
  } else {
    const long tot = (long)batch * n;
    auto ld = [&](long idx) {
      const long bb = idx / n;
      return pc[bb * plane + (idx - bb * n)];
    };
    long i = threadIdx.x;
    for (; i + 3072 < tot; i += 4096) {
      const float a = ld(i), e = ld(i + 1024), f = ld(i + 2048), g = ld(i + 3072);
      s0 += a;
      s1 += e;
      s2 += f;
      s3 += g;
    }
    for (; i < tot; i += 1024) s0 += ld(i);
  }
  float s = (s0 + s1) + (s2 + s3);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = red[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q];
    db[c] = t;
  }
}


// Fused finish of a weight-normalised layer: dw = sum of the reduction slabs (in slab order), then the
// old-style weight_norm (dim 0) backward of that row -- dg = <dw, v> / |v|, dv = (g / |v|)(dw - v <dw, v> / |v|^2)
// -- in ONE kernel: dw is never materialised and the separate reduce_slabs + weight_norm_backward launches
// (2 x ~170 per HiFi-GAN training step) disappear.  One workgroup per dim-0 slice; 32 element lanes x 8 slab
// lanes (every slab lane adds its slabs j = sl, sl + 8, ... in order, the 8 partials are added in order:
// deterministic).  Workgroups past the last row reduce the fused bias row.
struct WnFinish {
  const float* v;
  const float* g;
  float* dv;
  float* dg;
};

template <bool WIDE>
__global__ __launch_bounds__(256) void reduce_slabs_wn_kernel(const float* __restrict__ slabs, long slab_stride,
                                                              int nslabs, long slab_elems, const float* __restrict__ v,
                                                              const float* __restrict__ g, float* __restrict__ dv,
                                                              float* __restrict__ dg, float* __restrict__ db, int n0,
                                                              int inner, int nbias, int ci_g, int k) {
  // WIDE = false (few slabs): row[inner]; one thread per element adds the slabs in order.
  // WIDE = true (many slabs, short rows): part[8][inner] + 8 slab lanes per element, then the 8 partials in order.
  extern __shared__ float row[];
  __shared__ float red[2][4];
  if ((int)blockIdx.x >= n0) {
    const int c = ((int)blockIdx.x - n0) * 256 + threadIdx.x;
    if (c < nbias) {
      float s = 0.f;
      int j = 0;
      for (; j + 8 <= nslabs; j += 8) {  // (loads first, additions in slab order)
        float u[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) u[q] = slabs[(long)(j + q) * slab_stride + slab_elems + c];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += u[q];
      }
      for (; j < nslabs; ++j) s += slabs[(long)j * slab_stride + slab_elems + c];
      db[c] = s;
    }
    return;
  }
  // row o of dW: k segments of ci_g consecutive floats in a tap-major slab (segment `tap` starts at
  // (tap * n0 + o) * ci_g); gathered in slab order (coalesced) into row[] in torch order (i * k + tap)
  const long base = (long)blockIdx.x * inner;
  const long seg0 = (long)blockIdx.x * ci_g, seg_step = (long)n0 * ci_g;
  if (WIDE) {
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    float* part = row + inner;  // [8][inner], slab order
    for (int e = el; e < inner; e += 32) {
      const int tap = e / ci_g, i = e - tap * ci_g;
      const long src = seg0 + tap * seg_step + i;
      float s = 0.f;
      // (round 6: eight loads in flight per lane, added in the same order -- the dependent load + add rounds made this
      // finisher 62 - 121 us for MelGAN's short rows cut into 256 - 512 slabs, profiles/r06_wgrad_k1.txt)
      int j = sl;
      for (; j + 56 < nslabs; j += 64) {
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = slabs[(long)(j + 8 * q) * slab_stride + src];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += t[q];
      }
      for (; j < nslabs; j += 8) s += slabs[(long)j * slab_stride + src];
      part[sl * inner + e] = s;
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < inner; e += 256) {
    const int tap = e / ci_g, i = e - tap * ci_g;
    float t;
    if (WIDE) {
      const float* part = row + inner;
      t = part[e];
#pragma unroll
      for (int q = 1; q < 8; ++q) t += part[q * inner + e];
    } else {
      const long src = seg0 + tap * seg_step + i;
      t = 0.f;
      int j = 0;
      for (; j + 8 <= nslabs; j += 8) {  // (loads first, additions in slab order)
        float u[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) u[q] = slabs[(long)(j + q) * slab_stride + src];
#pragma unroll
        for (int q = 0; q < 8; ++q) t += u[q];
      }
      for (; j < nslabs; ++j) t += slabs[(long)j * slab_stride + src];
    }
    row[i * k + tap] = t;
  }
  __syncthreads();
  float svv = 0.f, sdv = 0.f;
  for (int e = threadIdx.x; e < inner; e += 256) {
    const float a = v[base + e];
    svv += a * a;
    sdv += a * row[e];
  }
  for (int o = 32; o > 0; o >>= 1) {
    svv += __shfl_down(svv, o, 64);
    sdv += __shfl_down(sdv, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = svv;
    red[1][threadIdx.x >> 6] = sdv;
  }
  __syncthreads();
  const float tvv = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  const float tdv = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float norm = sqrtf(tvv);
  if (threadIdx.x == 0) dg[blockIdx.x] = tdv / norm;
  const float c1 = g[blockIdx.x] / norm, c2 = tdv / tvv;
  for (int i = threadIdx.x; i < inner; i += 256) dv[base + i] = c1 * (row[i] - v[base + i] * c2);
}

// ---------------------------------------------------------------------------
// Weight gradient of single-input-channel convolutions (the discriminators' first layers, see conv1d_small_cin_kernel
// in conv1d.hip): dW[co][0][j] = sum_{b,t} dy[b][co][t] * act(x[b][t + j*d - pad]), db[co] = sum dy.  On the MFMA tile one
// of 32 operand columns carries data (round 3: 1.5 - 3 TFLOP/s, 1.15 ms of a C4 step, 0.65 ms of C3).  Here: one
// workgroup per (item, 1024-column chunk); the x window goes to LDS, each wave walks a quarter of the output channels,
// its lanes stride over the chunk's columns with k + 1 running sums in registers (dy read once, coalesced; x taps are
// shifted LDS reads), a wave reduction finishes each channel.  Every (item, chunk) writes a private tap-major slab
// [tap][co] + bias row -- the layout the slab finishers above expect -- summed afterwards in a fixed order.
// ---------------------------------------------------------------------------
constexpr int SIW_TILE = 1024;
constexpr int SIW_MAXK = 16;
constexpr int SIW_CO_PER_WG = 16;  // output channels per workgroup (grid.z walks the channel groups): 4 per wave
__global__ __launch_bounds__(256) void conv1d_small_cin_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     float* __restrict__ slabs, long slab_stride, int cout,
                                                                     int t_in, int t_out, int k, int dil, int pad,
                                                                     int chunks_per_item, float slope_x, int write_bias) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [SIW_TILE + halo (+ 4: see the dilation-1 loop)]
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int t0 = chunk * SIW_TILE;
  const int L = SIW_TILE + (k - 1) * dil;
  const float* xb = x + (long)b * t_in;
  for (int i = threadIdx.x; i < L; i += 256) {
    const int f = t0 - pad + i;
    float v = (f >= 0 && f < t_in) ? xb[f] : 0.f;
    xs[i] = __builtin_fmaxf(v, v * slope_x);  // LeakyReLU for 0 <= slope <= 1 (1 = none, 0 = ReLU), as the MFMA kernel
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = min(SIW_TILE, t_out - t0);
  float* slab = slabs + ((long)b * chunks_per_item + chunk) * slab_stride;
  const int co_end = min(cout, ((int)blockIdx.z + 1) * SIW_CO_PER_WG);
  if (dil == 1) {
    // Round 6 (0.2 of the streaming rate before, profiles/r06_wgrad_k1.txt): a lane owns 4 consecutive columns per round
    // (one 16-B load of dy per channel), holds their k + 3 window samples in registers (five 16-B LDS reads instead of
    // 4 * k dword reads) and feeds TWO output channels from them.
    for (int co = blockIdx.z * SIW_CO_PER_WG + wave; co < co_end; co += 8) {
      const bool two = co + 4 < co_end;
      const float* g0 = dy + ((long)b * cout + co) * t_out + t0;
      const float* g1 = g0 + (two ? 4 * (long)t_out : 0);
      float acc0[SIW_MAXK], acc1[SIW_MAXK], accb0 = 0.f, accb1 = 0.f;
#pragma unroll
      for (int j = 0; j < SIW_MAXK; ++j) acc0[j] = acc1[j] = 0.f;
      for (int t = 4 * lane; t < n; t += 256) {
        float v0[4], v1[4];
        if (t + 4 <= n) {
          float4 a, c;
          __builtin_memcpy(&a, g0 + t, 16);  // (rows of dy are only 4-B aligned: t_out is arbitrary)
          __builtin_memcpy(&c, g1 + t, 16);
          v0[0] = a.x, v0[1] = a.y, v0[2] = a.z, v0[3] = a.w;
          v1[0] = c.x, v1[1] = c.y, v1[2] = c.z, v1[3] = c.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            v0[u] = t + u < n ? g0[t + u] : 0.f;
            v1[u] = t + u < n ? g1[t + u] : 0.f;
          }
        }
        // window samples t .. t + k + 2, fetched four at a time (the launcher allocates 4 floats past the staged tile:
        // the last fetch of the last lane may run up to 3 floats over, into entries no tap uses)
        float xw[SIW_MAXK + 4];
#pragma unroll
        for (int q = 0; q < (SIW_MAXK + 4) / 4; ++q) {
          float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
          if (4 * q < k + 3) w = *reinterpret_cast<const float4*>(xs + t + 4 * q);
          xw[4 * q + 0] = w.x, xw[4 * q + 1] = w.y, xw[4 * q + 2] = w.z, xw[4 * q + 3] = w.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          accb0 += v0[u];
          accb1 += v1[u];
#pragma unroll
          for (int j = 0; j < SIW_MAXK; ++j)
            if (j < k) {
              acc0[j] = __builtin_fmaf(v0[u], xw[u + j], acc0[j]);
              acc1[j] = __builtin_fmaf(v1[u], xw[u + j], acc1[j]);
            }
        }
      }
#pragma unroll
      for (int j = 0; j < SIW_MAXK; ++j)
        for (int o = 32; o > 0; o >>= 1) {
          acc0[j] += __shfl_down(acc0[j], o, 64);
          acc1[j] += __shfl_down(acc1[j], o, 64);
        }
      for (int o = 32; o > 0; o >>= 1) {
        accb0 += __shfl_down(accb0, o, 64);
        accb1 += __shfl_down(accb1, o, 64);
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < SIW_MAXK; ++j)
          if (j < k) {
            slab[(long)j * cout + co] = acc0[j];
            if (two) slab[(long)j * cout + co + 4] = acc1[j];
          }
        if (write_bias) {
          slab[(long)k * cout + co] = accb0;
          if (two) slab[(long)k * cout + co + 4] = accb1;
        }
      }
    }
    return;
  }
  for (int co = blockIdx.z * SIW_CO_PER_WG + wave; co < co_end; co += 4) {
    const float* g = dy + ((long)b * cout + co) * t_out + t0;
    float acc[SIW_MAXK], accb = 0.f;
#pragma unroll
    for (int j = 0; j < SIW_MAXK; ++j) acc[j] = 0.f;
    // four independent 256-B loads per wave in flight (the chunk is 1024 columns: 4 rounds at most)
    for (int t = lane; t < n; t += 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (t + 64 * u < n) ? g[t + 64 * u] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        accb += v[u];
        const int tt = min(t + 64 * u, SIW_TILE - 1);  // (v = 0 past the end: any staged column will do)
#pragma unroll
        for (int j = 0; j < SIW_MAXK; ++j)
          if (j < k) acc[j] = __builtin_fmaf(v[u], xs[tt + j * dil], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < SIW_MAXK; ++j)
      for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_down(acc[j], o, 64);
    for (int o = 32; o > 0; o >>= 1) accb += __shfl_down(accb, o, 64);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < SIW_MAXK; ++j)
        if (j < k) slab[(long)j * cout + co] = acc[j];
      if (write_bias) slab[(long)k * cout + co] = accb;
    }
  }
}

struct WgPlan {
  bool small;      // 32x32 tile, waves split the taps
  bool win;        // per-tap windows
  int tg;          // accumulators (taps) per wave
  int taps_block;  // taps per workgroup
  int tt;          // reduction columns per chunk
  int xs_stride;
  size_t lds;
  int chunks_per_item, chunks_total;
  int tap_groups, tiles, splits;
  int rows_half;  // > 0: MODE 4 (row-aligned chunks of 2 * rows_half rows), tt = 64
  bool rows_x4;   // MODE 4: X rows staged with 16-B DMAs
};

static size_t wgrad_lds(bool small, bool win, int taps_block, int tt, int stride, int dil, int width, int k,
                        int* xs_stride) {
  const int bt = small ? 32 : 64;
  const int rows = (width == 1) ? tt : ((tt - 1) / width + 2);
  const int ntaps_max = k < taps_block ? k : taps_block;
  const int xs_len = ((rows - 1) * stride + (ntaps_max - 1) * dil + 1) * width;
  *xs_stride = round_up(xs_len, 64) + 1;  // whole DMA pieces per row + odd stride (bank spread)
  if (win) return 2 * (size_t)(bt * tt + taps_block * bt * tt) * sizeof(float);
  return 2 * (size_t)(bt * tt + bt * *xs_stride) * sizeof(float);
}

static WgPlan wgrad_plan(int co_g, int ci_g, int groups, int k, int stride, int dil, int width, int n_cols,
                         int batch) {
  WgPlan p;
  // tuning overrides (tools/bench_wgrad.py sweeps): PWG_WG_SMALL=1 forces the 32x32 tile, PWG_WG_TG the taps per
  // wave of the 64x64 tile, PWG_WG_TT caps the chunk length, PWG_WG_RES the resident workgroups per CU
  static const int env_small = getenv("PWG_WG_SMALL") ? atoi(getenv("PWG_WG_SMALL")) : 0;
  static const int env_tg = getenv("PWG_WG_TG") ? atoi(getenv("PWG_WG_TG")) : 0;
  static const int env_tt = getenv("PWG_WG_TT") ? atoi(getenv("PWG_WG_TT")) : 0;
  static const int env_res = getenv("PWG_WG_RES") ? atoi(getenv("PWG_WG_RES")) : 0;
  p.small = co_g <= 32 || ci_g <= 32 || env_small;
  // (second pass: a 64 x 64 tile whose X rows are too long for the LDS even with one tap per workgroup -- a
  // ConvTranspose1d with stride 11, as StyleMelGAN's noise upsampler has: 31 * 11 + 1 samples per row and chunk --
  // falls back to the 32 x 32 tile, half the rows per tile)
  for (int pass = 0; pass < 2; ++pass) {
    if (p.small) {
      const int per_wave = ceil_div(k, 4);
      // at most 4 accumulators per wave: with 6 or 11 (k = 41: all taps in one workgroup) the kernel drops
      // to one wave per SIMD and runs 1.5x slower than three tap groups of 16 taps (tools/bench_wgrad.py)
      p.tg = per_wave <= 1 ? 1 : per_wave <= 2 ? 2 : per_wave <= 3 ? 3 : 4;
      p.taps_block = 4 * p.tg;
    } else {
      // Taps per workgroup (tools/bench_wgrad.py sweep).  A tap group costs its MFMAs -- TG accumulators,
      // padded taps included, slower per MFMA when the registers allow only 2 workgroups per CU (TG >= 6)
      // -- plus the staging of its X tile, which is `stride` times wider for strided convolutions.  More
      // groups also mean fewer reduction slices per group (fewer slabs).  k = 11 -> 4 groups of 3;
      // k = 7 -> 7; k = 5 -> 5; k = 41 stride 4 -> 6 groups of 7.
      float best = 1e30f;
      p.tg = 1;
      for (int tg = 1; tg <= 7; ++tg) {  // ties go to the smaller group
        // far-apart taps (period 11 flattened: dilation 11) widen the shared X tile; past 80 KB only one
        // workgroup fits a CU, which costs more than re-staging the tile for a second tap group
        int xs;
        const size_t lds = wgrad_lds(false, false, tg, 32, stride, dil, width, k, &xs);
        const float per_mfma = tg <= 3 ? 1.0f : tg <= 5 ? 1.05f : 1.25f;
        const float cost = ceil_div(k, tg) * (tg * per_mfma + 0.5f * stride) * (lds > 80 * 1024 ? 1.5f : 1.0f);
        if (cost < best) {
          best = cost;
          p.tg = tg;
        }
      }
      if (env_tg > 0) p.tg = env_tg < 7 ? env_tg : 7;
      p.taps_block = p.tg;
    }
    // Many far-apart taps (e.g. k = 41 with dilation 5) can exceed the LDS even with per-tap windows:
    // fall back to fewer taps per workgroup (more tap groups) until the tiles fit.
    static const int small_tgs[] = {4, 3, 2, 1};
    for (;;) {
      const int ntaps_max = k < p.taps_block ? k : p.taps_block;
      p.win = stride == 1 && width == 1 && (ntaps_max - 1) * dil > 96;  // taps far apart: per-tap windows
      // longest chunk whose double-buffered tiles keep two workgroups per CU (<= 80 KB), at most ~n_cols
      p.tt = 32;
      // (round 6, measured and NOT kept: 64-column chunks for 1 x 1 convolutions on the 64 x 64 tile -- the MelGAN residual
      // stacks' pointwise layers, 22 - 50 TFLOP/s -- so that an X row's 64-float DMA piece is used whole: 0.763 / 0.430 /
      // 0.386 ms -> 0.846 / 0.534 / 0.488 ms for 96 / 48 / 192 channels at the C4 batch, gpurun_out/r06j: half as many
      // slices to fill the chip with)
      for (int tt = p.small ? 128 : 32; tt >= 32; tt >>= 1) {
        int xs;
        if (tt > 32 && tt > n_cols) continue;
        if (env_tt > 0 && tt > env_tt && tt > 32) continue;
        if (wgrad_lds(p.small, p.win, p.taps_block, tt, stride, dil, width, k, &xs) <= 80 * 1024) {
          p.tt = tt;
          break;
        }
      }
      p.lds = wgrad_lds(p.small, p.win, p.taps_block, p.tt, stride, dil, width, k, &p.xs_stride);
      if (p.lds <= 160 * 1024 || p.tg == 1) break;
      if (p.small) {
        int next = 1;
        for (int v : small_tgs)
          if (v < p.tg) {
            next = v;
            break;
          }
        p.tg = next;
        p.taps_block = 4 * p.tg;
      } else {
        p.tg -= 1;
        p.taps_block = p.tg;
      }
    }
    if (p.lds <= 160 * 1024 || p.small) break;
    p.small = true;
  }
  const int bt = p.small ? 32 : 64;
  // Round 6, MODE 4: (k,1) layers that stay in width mode (strided: the period discriminators' stride-3 layers, the
  // batch-folded stride-4 layers) walk row-aligned chunks when a row fits a 32-column G block.  rows_half: the even
  // split with the fewest wasted rows at the item's tail, larger chunks first.
  static const bool rows_on = !(getenv("PWG_WG_ROWS") && atoi(getenv("PWG_WG_ROWS")) == 0);
  p.rows_half = 0;
  p.rows_x4 = false;
  if (rows_on && width > 1 && width <= 32 && !p.win && n_cols % width == 0) {
    const int h_out = n_cols / width;
    const int ntaps_max = k < p.taps_block ? k : p.taps_block;
    float best = -1.f;
    static const bool rows_x4_on = !(getenv("PWG_WG_ROWS_X4") && atoi(getenv("PWG_WG_ROWS_X4")) == 0);
    for (int r = 32 / width; r >= 1; --r) {
      const int xs_len = ((2 * r - 1) * stride + (ntaps_max - 1) * dil + 1) * width;
      // 16-B pieces (one DMA instruction per row, <= 64 lanes): the window starts up to 3 floats early and ends on a
      // multiple of 4 -- an odd stride that covers it; dword pieces: whole 64-float pieces per row
      const bool x4 = rows_x4_on && xs_len + 6 <= 256;
      const int xs = x4 ? ((xs_len + 6) | 1) : round_up(xs_len, 64) + 1;
      const size_t lds = 2 * (size_t)(bt * 64 + bt * xs) * sizeof(float);
      if (lds > 160 * 1024) continue;
      const float useful = (float)h_out / (float)(ceil_div(h_out, 2 * r) * 2 * r);
      // (a chunk below ~24 columns carries too little work per barrier: small penalty)
      const float score = useful * (2 * r * width >= 24 ? 1.f : 0.85f) * (lds <= 80 * 1024 ? 1.05f : 1.f);
      if (score > best + 1e-6f) {
        best = score;
        p.rows_half = r;
        p.rows_x4 = x4;
        p.xs_stride = xs;
        p.lds = lds;
      }
    }
    if (p.rows_half > 0) {
      p.tt = 64;
      p.chunks_per_item = ceil_div(h_out, 2 * p.rows_half);
    }
  }
  if (p.rows_half == 0) p.chunks_per_item = ceil_div(n_cols, p.tt);
  p.chunks_total = p.chunks_per_item * batch;
  p.tap_groups = ceil_div(k, p.taps_block);
  p.tiles = ceil_div(co_g, bt) * ceil_div(ci_g, bt) * groups;
  // one wave of resident workgroups (register-limited: 2 per CU with 6-7 accumulators, else 3), all
  // with the same amount of work; at least 256 columns of reduction per workgroup
  // (with concurrent launches -- the parallel sub-discriminator branches of a captured step -- two resident
  // workgroups per CU per launch are enough, and every slice saved is a slab less to write and reduce)
  // (round 5: the strided (k,1) layers of the period discriminators stage an X tile `stride` x `width`-rows wide per
  // chunk -- 114 KB of LDS at 512 -> 1024, one workgroup per CU anyway --: fewer, longer slices win there even
  // stand-alone (tools/bench_wgrad.py, PWG_WG_RES=1 against 3: 128 -> 512 stride 3: 97.8 -> 66.9 us, 512 -> 1024:
  // 212 -> 195 us))
  const int res_default = (stride > 1 && width > 1) ? 1
                          : (!p.small && p.tg >= 6) ? 2 : (concurrency_hint() < 1.f ? 2 : 3);
  const int resident = 256 * (env_res > 0 ? env_res : res_default);
  int splits = resident / (p.tiles * p.tap_groups);
  const int chunk_cols = p.rows_half > 0 ? 2 * p.rows_half * width : p.tt;
  const int min_chunks = 256 / chunk_cols > 1 ? 256 / chunk_cols : 1;
  if (splits > p.chunks_total / min_chunks) splits = p.chunks_total / min_chunks;
  if (splits < 1) splits = 1;
  p.splits = ceil_div(p.chunks_total, ceil_div(p.chunks_total, splits));
  return p;
}

// Sum `splits` tap-major slabs (+ fused bias row) of a layer with `n0` rows, `ci_g` input channels per group and `k`
// taps into the torch-layout gradients; with `wn`, through the weight-norm backward.  (Shared by the MFMA kernel's
// launcher and the single-input-channel path.)
static int finish_wgrad_slabs(float* workspace, int splits, long slab_elems, long slab_stride, int n0, int ci_g, int k,
                              float* dw_out, float* db_out, const WnFinish* wn, hipStream_t stream) {
  // fused finisher: one workgroup per weight row, so it needs many rows or few slabs to fill the chip; layers
  // with few rows cut into many slabs (C <= 128 generator stages) keep the wide slab reduction (one workgroup
  // per 32 elements) followed by the row-wise weight-norm backward, into a spare slab of the workspace
  const long plane = (long)n0 * ci_g;  // elements per tap of a tap-major slab
  const int inner = ci_g * k;
  // (round 6, measured twice and NOT kept: the fused finisher also for short rows cut into many slabs -- MelGAN's 48 / 96 / 192-
  // channel stacks, one workgroup per row walking 128 - 512 slabs.  With dependent load + add rounds: 62 - 121 us per layer
  // instead of 13 - 16 us for the two launches; with eight loads in flight (as the kernel has them now): 8.6 - 25 us, better at
  // 128 / 192 rows, worse at 32 - 64, and the captured steps lose: C3 45.9 -> 46.4 ms, C5 41.9 -> 42.5 ms, C4 25.8 -> 25.9 ms,
  // profiles/r06_wgrad_k1.txt)
  const bool wn_fused = wn != nullptr && (splits < 16 || n0 >= 512);
  if (wn != nullptr && !wn_fused) {
    float* dw_tmp = workspace + (size_t)splits * slab_stride;
    {
      ProfScope prof(stream, "reduce_slabs_kernel", 0, 4.0 * slab_stride * (splits + 1));
      hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((slab_stride + 31) / 32)), dim3(256), 0, stream,
                         workspace, dw_tmp, db_out, slab_elems, slab_stride, splits, plane, k);
      PWG_CHECK_LAUNCH("reduce_slabs");
    }
    return pwg_weight_norm_backward(dw_tmp, wn->v, wn->g, wn->dv, wn->dg, n0, inner, stream);
  }
  if (wn != nullptr) {
    const int nbias = db_out ? n0 : 0;
    ProfScope prof(stream, "reduce_slabs_wn_kernel", 0, 4.0 * (slab_stride * (double)splits + 3.0 * slab_elems));
    const bool wide = splits >= 16 && (size_t)9 * inner * sizeof(float) <= 64 * 1024;  // (>= 512 rows of <= 1820 floats)
    if (wide)
      hipLaunchKernelGGL(reduce_slabs_wn_kernel<true>, dim3(n0 + ceil_div(nbias, 256)), dim3(256),
                         (size_t)9 * inner * sizeof(float), stream, (const float*)workspace, slab_stride, splits,
                         slab_elems, wn->v, wn->g, wn->dv, wn->dg, db_out, n0, inner, nbias, ci_g, k);
    else
      hipLaunchKernelGGL(reduce_slabs_wn_kernel<false>, dim3(n0 + ceil_div(nbias, 256)), dim3(256),
                         (size_t)inner * sizeof(float), stream, (const float*)workspace, slab_stride, splits,
                         slab_elems, wn->v, wn->g, wn->dv, wn->dg, db_out, n0, inner, nbias, ci_g, k);
    PWG_CHECK_LAUNCH("reduce_slabs_wn");
    return PWG_OK;
  }
  if (splits > 1) {
    ProfScope prof(stream, "reduce_slabs_kernel", 0, 4.0 * slab_stride * (splits + 1));
    if (splits >= 16) {
      hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((slab_stride + 31) / 32)), dim3(256), 0, stream,
                         workspace, dw_out, db_out, slab_elems, slab_stride, splits, plane, k);
    } else {
      long blocks = (slab_stride + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3((int)blocks), dim3(256), 0, stream, workspace, dw_out, db_out,
                         slab_elems, slab_stride, splits, plane, k);
    }
    PWG_CHECK_LAUNCH("reduce_slabs");
  }
  return PWG_OK;
}

template <int TG, bool SMALL, int TT, bool WIN, int MODE, bool ACT23 = true, int STRIDE3 = 0>
static int launch_wgrad_mode(WgArgs a, const WgPlan& p, float* dw_out, float* workspace, size_t ws_floats,
                        hipStream_t stream, double flops, double bytes, const WnFinish* wn) {
  a.xs_stride = p.xs_stride;
  a.chunks_per_item = p.chunks_per_item;
  a.chunks_total = p.chunks_total;
  const size_t lds = p.lds;
  PWG_REQUIRE(lds <= 160 * 1024, PWG_ERR_UNSUPPORTED, "conv1d_backward_weight: tile needs %zu B of LDS", lds);
  void (*kern)(WgArgs) = conv1d_wgrad_kernel<TG, WIN, SMALL, TT, MODE, ACT23, STRIDE3>;
  if (lds > 64 * 1024 && !lds_limit_is_set(reinterpret_cast<const void*>(kern), lds)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    PWG_REQUIRE(e == hipSuccess, PWG_ERR_LAUNCH, "conv1d_backward_weight: cannot raise LDS limit: %s",
                hipGetErrorString(e));
  }
  a.chunks_per_block = ceil_div(a.chunks_total, p.splits);
  a.slab_elems = (long)a.co_g * a.groups * a.ci_g * a.k;
  float* db_out = a.db;  // non-null: the bias gradient rides along (row sums of the G tiles)
  a.slab_stride = a.slab_elems + (db_out ? (long)a.co_g * a.groups : 0);
  a.tap_major = 0;
  if (p.splits == 1 && wn == nullptr) {
    a.dw = dw_out;  // single slice: write the gradients directly (torch layout)
  } else {
    a.tap_major = 1;
    PWG_REQUIRE(workspace && ws_floats >= (size_t)p.splits * a.slab_stride, PWG_ERR_WORKSPACE,
                "conv1d_backward_weight: workspace of %zu floats needed, %zu given",
                (size_t)p.splits * a.slab_stride, ws_floats);
    a.dw = workspace;
    if (db_out) a.db = workspace + a.slab_elems;
  }
  dim3 grid(p.splits, p.tiles, p.tap_groups);
  maybe_poison_lds(stream);
  {
    ProfScope prof(stream,
                   prof_shape_name("conv1d_wgrad_kernel", "B%d Co%d Ci%d k%d s%d d%d g%d W%d cols%d splits%d tiles%d tg%d small%d win%d tt%d rows%d",
                                   a.batch, a.co_g * a.groups, a.ci_g * a.groups, a.k, a.stride, a.dil, a.groups, a.width,
                                   a.n_cols, p.splits, p.tiles, p.tg, (int)p.small, (int)p.win, p.tt, 2 * p.rows_half),
                   flops, bytes);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
  }
  PWG_CHECK_LAUNCH("conv1d_backward_weight");
  return finish_wgrad_slabs(workspace, p.splits, a.slab_elems, a.slab_stride, a.co_g * a.groups, a.ci_g, a.k, dw_out,
                            db_out, wn, stream);
}

template <int TG, bool SMALL, int TT>
static int launch_wgrad(WgArgs a, const WgPlan& p, float* dw_out, float* workspace, size_t ws_floats,
                        hipStream_t stream, double flops, double bytes, const WnFinish* wn) {
  const bool act = (a.slope_g != 1.f || a.slope_x != 1.f) && !(a.dbg & 8);
#define WG_GO(WINV, MODEV) \
  return launch_wgrad_mode<TG, SMALL, TT, WINV, MODEV>(a, p, dw_out, workspace, ws_floats, stream, flops, bytes, wn)
#define WG_GO23(MODEV, ACTV, SV) \
  return launch_wgrad_mode<TG, SMALL, TT, false, MODEV, ACTV, SV>(a, p, dw_out, workspace, ws_floats, stream, flops, bytes, wn)
  static const bool fast23 = !(getenv("PWG_WG_FAST23") && atoi(getenv("PWG_WG_FAST23")) == 0);  // (0: the round-5 loops, A/B)
  if (a.width != 1) {  // per-tap windows need width == 1 and stride == 1 (wgrad_plan)
    if (act || !fast23) WG_GO23(2, true, 0);
    WG_GO23(2, false, 0);
  }
  if (a.stride != 1) {
    if (fast23 && !act) {  // the strides of the recipes' layers: scale discriminators 4 (2 in some recipes), upsamplers 8 / 5 / 3
      switch (a.stride) {
        case 2: WG_GO23(3, false, 2);
        case 3: WG_GO23(3, false, 3);
        case 4: WG_GO23(3, false, 4);
        case 8: WG_GO23(3, false, 8);
        default: WG_GO23(3, false, 0);
      }
    }
    if (fast23 && a.stride == 8) WG_GO23(3, true, 8);  // (ConvTranspose1d of the generators: activated input)
    if (fast23 && a.stride == 4) WG_GO23(3, true, 4);
    if (fast23 && a.stride == 2) WG_GO23(3, true, 2);
    WG_GO23(3, true, 0);
  }
  if (p.win) {
    if (act) WG_GO(true, 1);
    WG_GO(true, 0);
  }
  if (act) WG_GO(false, 1);
  WG_GO(false, 0);
#undef WG_GO
#undef WG_GO23
}

// MODE 4 only (row-aligned (k,1) chunks, TT = 64): kept apart so that the 64 x 64 tile gets no other TT = 64 instantiation
template <int TG, bool SMALL>
static int launch_wgrad_rows(WgArgs a, const WgPlan& p, float* dw_out, float* workspace, size_t ws_floats,
                             hipStream_t stream, double flops, double bytes, const WnFinish* wn) {
  const bool act = (a.slope_g != 1.f || a.slope_x != 1.f) && !(a.dbg & 8);
  a.rows_half = p.rows_half;
  a.rows_x4 = p.rows_x4 ? 1 : 0;
  if (act) return launch_wgrad_mode<TG, SMALL, 64, false, 4, true, 0>(a, p, dw_out, workspace, ws_floats, stream, flops, bytes, wn);
  return launch_wgrad_mode<TG, SMALL, 64, false, 4, false, 0>(a, p, dw_out, workspace, ws_floats, stream, flops, bytes, wn);
}

}  // namespace pwg

using namespace pwg;

// single-input-channel path (conv1d_small_cin_wgrad_kernel): `d` flattened; PWG_SMALL_CIN=0 disables it
static bool small_cin_wgrad_applicable(const pwg_conv1d_desc* d) {
  static const bool on = !(getenv("PWG_SMALL_CIN") && atoi(getenv("PWG_SMALL_CIN")) == 0);
  return on && !d->transposed && d->groups == 1 && d->c_in == 1 && d->width == 1 && d->stride == 1 &&
         d->pad_mode == PWG_PAD_ZERO && d->kernel <= SIW_MAXK && d->c_out >= 8 && d->c_out <= 1024 && d->t_out >= 2048 &&
         (size_t)(SIW_TILE + (d->kernel - 1) * d->dilation + 4) * sizeof(float) <= 64 * 1024;
}
static long small_cin_wgrad_slabs(const pwg_conv1d_desc* d) { return (long)d->batch * ceil_div(d->t_out, SIW_TILE); }

static void wgrad_roles(const pwg_conv1d_desc* d, int* co_g, int* ci_g, int* n_cols) {
  if (!d->transposed) {
    *co_g = d->c_out / d->groups;
    *ci_g = d->c_in / d->groups;
    *n_cols = d->t_out * d->width;
  } else {
    *co_g = d->c_in / d->groups;
    *ci_g = d->c_out / d->groups;
    *n_cols = d->t_in * d->width;
  }
}

extern "C" size_t pwg_conv1d_backward_weight_workspace_floats(const pwg_conv1d_desc* d_in) {
  if (!d_in || d_in->groups <= 0 || d_in->c_in % d_in->groups || d_in->c_out % d_in->groups) return 0;
  const pwg_conv1d_desc flat = flatten_width(*d_in);
  const pwg_conv1d_desc* d = &flat;
  int co_g, ci_g, n_cols;
  wgrad_roles(d, &co_g, &ci_g, &n_cols);
  const WgPlan p = wgrad_plan(co_g, ci_g, d->groups, d->kernel, d->stride, d->dilation, d->width, n_cols, d->batch);
  if (gconv_wgrad_applicable(d)) return gconv_wgrad_workspace_floats(d);
  if (small_cin_wgrad_applicable(d)) return (size_t)small_cin_wgrad_slabs(d) * ((size_t)d->c_out * (d->kernel + 1));
  if (k1_wgrad_applicable(d)) return (size_t)k1_wgrad_slabs(d) * ((size_t)d->c_out * (d->c_in + 1));
  // one slab per reduction slice: dW plus the fused bias row
  return p.splits > 1 ? (size_t)p.splits * ((size_t)co_g * d->groups * ci_g * d->kernel + (size_t)co_g * d->groups) : 0;
}

// wn != nullptr: finish the slabs with the fused weight-norm backward (pwg_conv1d_backward_weight_wn)
static int backward_weight_impl(const pwg_conv1d_desc* d_in, const float* x, const float* dy, float* dw, float* db,
                                float* workspace, size_t workspace_floats, void* stream_, const WnFinish* wn) {
  PWG_REQUIRE(d_in && x && dy, PWG_ERR_NULL, "conv1d_backward_weight: NULL pointer");
  const pwg_conv1d_desc flat = flatten_width(*d_in);
  const pwg_conv1d_desc* d = &flat;
  PWG_REQUIRE(d->c_in % d->groups == 0 && d->c_out % d->groups == 0 && d->groups > 0, PWG_ERR_BAD_SHAPE,
              "conv1d_backward_weight: bad groups");
  PWG_REQUIRE(d->pad_mode == PWG_PAD_ZERO, PWG_ERR_UNSUPPORTED,
              "conv1d_backward_weight: only zero padding (pad reflect/replicate inputs explicitly)");
  hipStream_t stream = (hipStream_t)stream_;
  const long y_elems = (long)d->batch * d->c_out * d->t_out * d->width;
  const long x_elems = (long)d->batch * d->c_in * d->t_in * d->width;
  PWG_REQUIRE(y_elems * 4 < 0xFFFFFFF0L && x_elems * 4 < 0xFFFFFFF0L, PWG_ERR_UNSUPPORTED,
              "conv1d_backward_weight: tensors above 4 GiB need batch splitting");
  // The bias gradient (row sums of dy) is fused into the weight-gradient kernel whenever dy plays the
  // G role there (plain convolutions); ConvTranspose1d and bias-only calls use the separate kernel.
  const bool fuse_bias = db && dw && !d->transposed;
  if (db && !fuse_bias) {
    const int n = d->t_out * d->width;
    ProfScope prof(stream, "bias_grad_kernel", 0, 4.0 * y_elems);
    hipLaunchKernelGGL(bias_grad_kernel, dim3(d->c_out), dim3(1024), 0, stream, dy, db, d->c_out, n, d->batch);
    PWG_CHECK_LAUNCH("bias_grad");
  }
  if (!dw) return PWG_OK;
  if (gconv_wgrad_applicable(d)) {
    // few channels per group: 16 x 16 x 4 MFMA kernel of gconv.hip (bias gradient fused when dy is the G operand)
    float* db_fused = fuse_bias ? db : nullptr;
    if (wn == nullptr) return gconv_backward_weight(d, x, dy, dw, db_fused, workspace, workspace_floats, stream);
    const size_t need = gconv_wgrad_workspace_floats(d);
    PWG_REQUIRE(workspace && workspace_floats >= need, PWG_ERR_WORKSPACE,
                "conv1d_backward_weight_wn: workspace of %zu floats needed, %zu given", need, workspace_floats);
    const size_t w_elems = (size_t)d->c_out * (d->c_in / d->groups) * d->kernel;
    float* dw_tmp = workspace + (need - w_elems - d->c_out);  // behind the slabs
    const int rc = gconv_backward_weight(d, x, dy, dw_tmp, db_fused, workspace, need - w_elems - d->c_out, stream);
    if (rc != PWG_OK) return rc;
    return pwg_weight_norm_backward(dw_tmp, wn->v, wn->g, wn->dv, wn->dg, d->c_out, (int)(w_elems / d->c_out), stream);
  }
  WgArgs a;
  const float slope = d->pre_act == PWG_ACT_LEAKY_RELU ? d->pre_slope : (d->pre_act == PWG_ACT_RELU ? 0.f : 1.f);
  PWG_REQUIRE(slope >= 0.f && slope <= 1.f, PWG_ERR_UNSUPPORTED,
              "conv1d_backward_weight: LeakyReLU slope %g outside [0, 1] (the operand activation is max(v, slope*v))",
              (double)slope);
  if (small_cin_wgrad_applicable(d)) {
    const long nslabs = small_cin_wgrad_slabs(d);
    const long slab_elems = (long)d->c_out * d->kernel, slab_stride = slab_elems + (db ? d->c_out : 0);
    const size_t need = (size_t)(nslabs + (wn ? 1 : 0)) * slab_stride;
    PWG_REQUIRE(workspace && workspace_floats >= need, PWG_ERR_WORKSPACE,
                "conv1d_backward_weight: workspace of %zu floats needed, %zu given", need, workspace_floats);
    PWG_REQUIRE(nslabs < (1L << 30), PWG_ERR_UNSUPPORTED, "conv1d_backward_weight: too many slabs");
    {
      ProfScope prof(stream, "conv1d_small_cin_wgrad_kernel", 2.0 * y_elems * d->kernel, 4.0 * ((double)x_elems + (double)y_elems));
      hipLaunchKernelGGL(conv1d_small_cin_wgrad_kernel,
                         dim3(ceil_div(d->t_out, SIW_TILE), d->batch, ceil_div(d->c_out, SIW_CO_PER_WG)), dim3(256),
                         (size_t)(SIW_TILE + (d->kernel - 1) * d->dilation + 4) * sizeof(float), stream, x, dy, workspace,
                         slab_stride, d->c_out, d->t_in, d->t_out, d->kernel, d->dilation, d->pad_left,
                         ceil_div(d->t_out, SIW_TILE), slope, db ? 1 : 0);
      PWG_CHECK_LAUNCH("conv1d_small_cin_wgrad");
    }
    return finish_wgrad_slabs(workspace, (int)nslabs, slab_elems, slab_stride, d->c_out, 1, d->kernel, dw, db, wn, stream);
  }
  if (k1_wgrad_applicable(d)) {
    // 1 x 1 layers with few channels: HBM-bound kernel of wgrad_k1.hip, one slab per workgroup
    const int nslabs = k1_wgrad_slabs(d);
    const long slab_elems = (long)d->c_out * d->c_in, slab_stride = slab_elems + (db ? d->c_out : 0);
    const size_t need = (size_t)(nslabs + (wn ? 1 : 0)) * slab_stride;
    PWG_REQUIRE(workspace && workspace_floats >= need, PWG_ERR_WORKSPACE,
                "conv1d_backward_weight: workspace of %zu floats needed, %zu given", need, workspace_floats);
    const int rc = k1_wgrad_launch(d, x, dy, workspace, slab_stride, nslabs, slope, db != nullptr, stream);
    if (rc != PWG_OK) return rc;
    return finish_wgrad_slabs(workspace, nslabs, slab_elems, slab_stride, d->c_out, d->c_in, 1, dw, db, wn, stream);
  }
  if (!d->transposed) {
    a.g = dy;
    a.x = x;
    a.co_g = d->c_out / d->groups;
    a.ci_g = d->c_in / d->groups;
    a.n_cols = d->t_out * d->width;
    a.x_len = d->t_in * d->width;
    a.slope_g = 1.f;
    a.slope_x = slope;
    a.g_bytes = (unsigned)(y_elems * 4);
    a.x_bytes = (unsigned)(x_elems * 4);
  } else {
    // ConvTranspose1d: dW[ci][co][k] = sum x[ci][q] * dy[co][q*s - p + k]: same kernel, roles swapped
    PWG_REQUIRE(d->dilation == 1 || d->stride == 1, PWG_ERR_UNSUPPORTED, "conv_transpose1d wgrad: dilation with stride");
    a.g = x;
    a.x = dy;
    a.co_g = d->c_in / d->groups;
    a.ci_g = d->c_out / d->groups;
    a.n_cols = d->t_in * d->width;
    a.x_len = d->t_out * d->width;
    a.slope_g = slope;
    a.slope_x = 1.f;
    a.g_bytes = (unsigned)(x_elems * 4);
    a.x_bytes = (unsigned)(y_elems * 4);
  }
  static const int dbg = getenv("PWG_WG_DBG") ? atoi(getenv("PWG_WG_DBG")) : 0;
  a.dbg = dbg;
  a.dw = nullptr;
  a.db = fuse_bias ? db : nullptr;
  a.groups = d->groups;
  a.k = d->kernel;
  a.stride = d->stride;
  a.dil = d->dilation;
  a.pad = d->pad_left;
  a.width = d->width;
  a.batch = d->batch;
  const double flops = 2.0 * d->batch * (double)a.n_cols * a.co_g * a.ci_g * d->groups * d->kernel;
  const double bytes = 4.0 * ((double)x_elems + (double)y_elems + (double)a.co_g * a.ci_g * d->groups * d->kernel);
  const WgPlan p = wgrad_plan(a.co_g, a.ci_g, d->groups, d->kernel, d->stride, d->dilation, d->width, a.n_cols,
                              d->batch);
#define WG_CASE(TGV, SM)                                                                             \
  switch (p.tt) {                                                                                    \
    case 128: return launch_wgrad<TGV, SM, 128>(a, p, dw, workspace, workspace_floats, stream, flops, bytes, wn); \
    case 64: return launch_wgrad<TGV, SM, 64>(a, p, dw, workspace, workspace_floats, stream, flops, bytes, wn);   \
    default: return launch_wgrad<TGV, SM, 32>(a, p, dw, workspace, workspace_floats, stream, flops, bytes, wn);   \
  }
  a.rows_half = 0;
  a.rows_x4 = 0;
  if (p.rows_half > 0) {
#define WG_ROWS(TGV, SM) return launch_wgrad_rows<TGV, SM>(a, p, dw, workspace, workspace_floats, stream, flops, bytes, wn)
    if (p.small) {
      switch (p.tg) {
        case 1: WG_ROWS(1, true);
        case 2: WG_ROWS(2, true);
        case 3: WG_ROWS(3, true);
        default: WG_ROWS(4, true);
      }
    }
    switch (p.tg) {
      case 1: WG_ROWS(1, false);
      case 2: WG_ROWS(2, false);
      case 3: WG_ROWS(3, false);
      case 4: WG_ROWS(4, false);
      case 5: WG_ROWS(5, false);
      case 6: WG_ROWS(6, false);
      default: WG_ROWS(7, false);
    }
#undef WG_ROWS
  }
  if (p.small) {
    switch (p.tg) {
      case 1: WG_CASE(1, true);
      case 2: WG_CASE(2, true);
      case 3: WG_CASE(3, true);
      default: WG_CASE(4, true);
    }
  }
#undef WG_CASE
#define WG_CASE(TGV) return launch_wgrad<TGV, false, 32>(a, p, dw, workspace, workspace_floats, stream, flops, bytes, wn)
  switch (p.tg) {  // 64x64 tiles always run 32-column chunks (LDS)
    case 1: WG_CASE(1);
    case 2: WG_CASE(2);
    case 3: WG_CASE(3);
    case 4: WG_CASE(4);
    case 5: WG_CASE(5);
    case 6: WG_CASE(6);
    default: WG_CASE(7);
  }
#undef WG_CASE
}

extern "C" int pwg_conv1d_backward_weight(const pwg_conv1d_desc* d, const float* x, const float* dy, float* dw,
                                          float* db, float* workspace, size_t workspace_floats, void* stream) {
  return backward_weight_impl(d, x, dy, dw, db, workspace, workspace_floats, stream, nullptr);
}

// Weight-normalised layers: dv, dg (and db) straight from the reduction slabs -- see reduce_slabs_wn_kernel.
// v (weight_v, torch layout) and g (weight_g, one value per dim-0 slice) are the forward's parameters.  The
// workspace is always needed here (>= 1 slab): query it with the function below.
extern "C" size_t pwg_conv1d_backward_weight_wn_workspace_floats(const pwg_conv1d_desc* d_in) {
  if (!d_in || d_in->groups <= 0 || d_in->c_in % d_in->groups || d_in->c_out % d_in->groups) return 0;
  const pwg_conv1d_desc flat = flatten_width(*d_in);
  const pwg_conv1d_desc* d = &flat;
  int co_g, ci_g, n_cols;
  wgrad_roles(d, &co_g, &ci_g, &n_cols);
  const WgPlan p = wgrad_plan(co_g, ci_g, d->groups, d->kernel, d->stride, d->dilation, d->width, n_cols, d->batch);
  if (gconv_wgrad_applicable(d)) return gconv_wgrad_workspace_floats(d);
  if (small_cin_wgrad_applicable(d)) return (size_t)(small_cin_wgrad_slabs(d) + 1) * ((size_t)d->c_out * (d->kernel + 1));
  if (k1_wgrad_applicable(d)) return (size_t)(k1_wgrad_slabs(d) + 1) * ((size_t)d->c_out * (d->c_in + 1));
  // (+1: room for the summed gradient when the two-kernel finish is used)
  return (size_t)(p.splits + 1) * ((size_t)co_g * d->groups * ci_g * d->kernel + (size_t)co_g * d->groups);
}

extern "C" int pwg_conv1d_backward_weight_wn(const pwg_conv1d_desc* d, const float* x, const float* dy, const float* v,
                                             const float* g, float* dv, float* dg, float* db, float* workspace,
                                             size_t workspace_floats, void* stream) {
  PWG_REQUIRE(d && v && g && dv && dg && workspace, PWG_ERR_NULL, "conv1d_backward_weight_wn: NULL pointer");
  int co_g, ci_g, n_cols;
  const pwg_conv1d_desc flat = flatten_width(*d);
  wgrad_roles(&flat, &co_g, &ci_g, &n_cols);
  // (the finishing kernel holds one weight row in dynamic LDS next to 32 B of static LDS: stay below the 64 KiB
  // default limit with room to spare; functional.py routes longer rows to the two-kernel finish)
  PWG_REQUIRE((size_t)ci_g * d->kernel * sizeof(float) + 256 <= 64 * 1024, PWG_ERR_UNSUPPORTED,
              "conv1d_backward_weight_wn: a weight row of %d floats exceeds the LDS row buffer", ci_g * d->kernel);
  PWG_REQUIRE(workspace_floats >= pwg_conv1d_backward_weight_wn_workspace_floats(d), PWG_ERR_WORKSPACE,
              "conv1d_backward_weight_wn: workspace too small");
  const WnFinish wn{v, g, dv, dg};
  // (dw argument: any non-NULL pointer selects the weight-gradient path; the slabs live in the workspace)
  return backward_weight_impl(d, x, dy, dv, db, workspace, workspace_floats, stream, &wn);
}
