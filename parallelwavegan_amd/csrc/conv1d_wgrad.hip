// conv1d_wgrad.hip -- weight gradient of the conv1d family on gfx950 fp32 MFMA.
//
//   dW[o][i][k] = sum_{b,n}  G[b][o][n] * act(X[b][i][ xoff(n) + (k*dil - pad)*W ])
//
// with G the gradient w.r.t. the convolution output and X its input (roles are swapped by the
// host for ConvTranspose1d, whose weight gradient is the same expression with x and dy
// exchanged).  GEMM view per (group, tap): rows = o, cols = i, reduction = (b, n):
//   A[o][n]  <- G tile   (LDS, rotation-swizzled so that 32 rows hit 32 banks)
//   B[n][i]  <- X tile   (LDS, odd row stride; every tap is a shifted read of the same tile)
// One workgroup = 2x2 waves = 64 o x 64 i x TG taps, looping over its slice of the (b,n) range in
// chunks of 32 columns with LDS-DMA double buffering; partial sums are combined with fp32
// atomics into the torch-layout gradient (the caller zeroes it).
#include "common.h"

namespace pwg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct WgArgs {
  const float* g;   // "output gradient" role   (B, CO, n_cols)
  const float* x;   // "input" role             (B, CI, x_len)
  float* dw;        // (CO, CI/groups, K) torch layout, accumulated atomically
  int co_g, ci_g, groups;
  int k, k0_step;   // taps, taps per tap-group
  int stride, dil, pad, width;
  int n_cols;       // columns per batch item in G  (rows_out * width)
  int x_len;        // samples per channel row in X (rows_in * width)
  int batch;
  int chunks_per_item, chunks_total, chunks_per_block;
  int xs_stride;    // odd
  long slab_elems;  // elements of one gradient slab (= the whole dW)
  float slope_g, slope_x;  // branch-free pre-activation slopes (1 = none)
  unsigned g_bytes, x_bytes;
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
template <int TG, bool WIN, bool SMALL, int TT>
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(WgArgs a) {
  constexpr int BT = SMALL ? 32 : 64;          // tile rows (o) and columns (i)
  constexpr int TAPS_BLOCK = SMALL ? 4 * TG : TG;
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

  auto issue = [&](int c, float* buf) {
    float* gs = buf;
    float* xs = buf + BT * TT;
    const int b = c / a.chunks_per_item;
    const int n0 = (c - b * a.chunks_per_item) * TT;
    // ---- G tile: [TT/32 column blocks][BT rows][32]; element (o, n) at column (n + o) & 31 of its block
    for (int jj = wave; jj < (TT / 32) * (BT / 2); jj += 4) {
      const int cb = jj / (BT / 2), j = jj - cb * (BT / 2);
      const int row = 2 * j + lhi;
      const int n = n0 + cb * 32 + ((l31 - row) & 31);
      const int o = o0 + row;
      unsigned off = OOB;
      if (o < a.co_g && n < a.n_cols)
        off = (unsigned)((((long)b * co_tot + grp * a.co_g + o) * a.n_cols + n) * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_t)(gs + cb * (BT * 32) + j * 64), 4, off, 0, 0, 0);
    }
    if (WIN) {
      // ---- per-tap windows: [tap][column block][BT][32], element (i, n) at column (n + i) & 31
      for (int t = 0; t < ntaps_block; ++t)
        for (int jj = wave; jj < (TT / 32) * (BT / 2); jj += 4) {
          const int cb = jj / (BT / 2), j = jj - cb * (BT / 2);
          const int row = 2 * j + lhi;
          const int i = i0 + row;
          const int f = n0 + cb * 32 + ((l31 - row) & 31) + (k0 + t) * a.dil - a.pad;
          unsigned off = OOB;
          if (i < a.ci_g && f >= 0 && f < a.x_len)
            off = (unsigned)((((long)b * ci_tot + grp * a.ci_g + i) * a.x_len + f) * 4);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + t * (BT * TT) + cb * (BT * 32) + j * 64), 4, off, 0, 0, 0);
        }
      return;
    }
    // ---- X tile: BT rows, flat range [f0, f0 + L)
    const int h0 = n0 / W;
    int n_last = n0 + TT - 1;
    if (n_last > a.n_cols - 1) n_last = a.n_cols - 1;
    const int h1 = n_last / W;
    const int f0 = (h0 * a.stride + k0 * a.dil - a.pad) * W;
    const int L = ((h1 - h0) * a.stride + (ntaps_block - 1) * a.dil + 1) * W;
    for (int r = wave; r < BT; r += 4) {
      const int i = i0 + r;
      const long rowbase = ((long)b * ci_tot + grp * a.ci_g + i) * a.x_len;
      for (int e0 = 0; e0 < L; e0 += 64) {
        const int f = f0 + e0 + lane;
        unsigned off = OOB;
        if (i < a.ci_g && f >= 0 && f < a.x_len && e0 + lane < L) off = (unsigned)((rowbase + f) * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS + e0), 4, off, 0, 0, 0);
      }
    }
  };

  if (c_begin < c_end) issue(c_begin, smem);
  for (int c = c_begin; c < c_end; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* buf = smem + ((c - c_begin) & 1) * buf_floats;
    if (c + 1 < c_end) issue(c + 1, smem + ((c + 1 - c_begin) & 1) * buf_floats);
    const float* gs = buf;
    const float* xs = buf + BT * TT;
    const int b = c / a.chunks_per_item;
    const int n0 = (c - b * a.chunks_per_item) * TT;
    const int h0 = n0 / W;
    const int orow = wave_o * 32 + l31;
    const float* grow = gs + orow * 32;
    const int irow = wave_i * 32 + l31;
    const float* xrow = xs + irow * (WIN ? 32 : XS) + (WIN ? t0 * (BT * TT) : t0 * a.dil * W);
#pragma unroll 4
    for (int step = 0; step < TT / 2; ++step) {
      const int nl = 2 * step + lhi;          // column within the chunk
      const int cb = nl >> 5, nb = nl & 31;   // 32-column block / column within it
      float av = grow[cb * (BT * 32) + ((nb + orow) & 31)];
      av = __builtin_fmaf(a.slope_g, __builtin_fminf(av, 0.f), __builtin_fmaxf(av, 0.f));
      int xo;
      if (WIN) {
        xo = cb * (BT * 32) + ((nb + irow) & 31);
      } else if (W == 1) {
        xo = nl * a.stride;
      } else {
        const int n = n0 + nl;
        const int h = n / W;
        xo = (h - h0) * a.stride * W + (n - h * W);
      }
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        if (t < ntaps) {
          float bv = WIN ? xrow[xo + t * (BT * TT)] : xrow[xo + t * a.dil * W];
          bv = __builtin_fmaf(a.slope_x, __builtin_fminf(bv, 0.f), __builtin_fmaxf(bv, 0.f));
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: D layout col = lane&31 (-> i), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (-> o).
  // Every reduction slice owns a private slab (torch weight layout) that it fully overwrites with
  // plain stores; slabs are summed by reduce_slabs_kernel (no atomics, deterministic).
  float* slab = a.dw + (long)blockIdx.x * a.slab_elems;
  const int i = i0 + wave_i * 32 + l31;
  if (i < a.ci_g) {
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t < ntaps) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + wave_o * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (o < a.co_g) slab[(((long)(grp * a.co_g + o)) * a.ci_g + i) * a.k + k0 + t0 + t] = acc[t][r];
        }
      }
    }
  }
}

// dw[e] = sum_s slabs[s][e]
__global__ void reduce_slabs_kernel(const float* slabs, float* dw, long elems, int nslabs) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < elems; e += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < nslabs; ++j) s += slabs[(long)j * elems + e];
    dw[e] = s;
  }
}

// db[c] += sum over one (item, 4096-sample segment) of dy[b][c][:]; grid (channels, items * segments);
// db is zeroed first (hipMemsetAsync) and the per-segment partial sums are combined with atomics.
constexpr int BG_SEG = 4096;
__global__ void bias_grad_kernel(const float* dy, float* db, int channels, int n, int segs) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const int b = blockIdx.y / segs, sg = blockIdx.y - b * segs;
  const float* p = dy + ((long)b * channels + c) * n;
  const int lo = sg * BG_SEG, hi = min(n, lo + BG_SEG);
  float s = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) s += p[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(db + c, red[0] + red[1] + red[2] + red[3]);
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
  p.small = co_g <= 32 || ci_g <= 32;
  if (p.small) {
    const int per_wave = ceil_div(k, 4);
    p.tg = per_wave <= 1 ? 1 : per_wave <= 2 ? 2 : per_wave <= 3 ? 3 : per_wave <= 4 ? 4 : per_wave <= 6 ? 6 : 11;
    p.taps_block = 4 * p.tg;
  } else {
    p.tg = k <= 4 ? 4 : (k <= 6 || k == 11 || k == 12) ? 6 : (k == 7 || k == 41 || k == 42 || k == 14) ? 7 : 8;
    p.taps_block = p.tg;
  }
  const int bt = p.small ? 32 : 64;
  const int ntaps_max = k < p.taps_block ? k : p.taps_block;
  p.win = stride == 1 && width == 1 && (ntaps_max - 1) * dil > 96;  // taps far apart: per-tap windows
  // longest chunk whose double-buffered tiles keep two workgroups per CU (<= 80 KB), at most ~n_cols
  p.tt = 32;
  for (int tt = 128; tt >= 32; tt >>= 1) {
    int xs;
    if (tt > 32 && tt > n_cols) continue;
    if (wgrad_lds(p.small, p.win, p.taps_block, tt, stride, dil, width, k, &xs) <= 80 * 1024) {
      p.tt = tt;
      break;
    }
  }
  p.lds = wgrad_lds(p.small, p.win, p.taps_block, p.tt, stride, dil, width, k, &p.xs_stride);
  p.chunks_per_item = ceil_div(n_cols, p.tt);
  p.chunks_total = p.chunks_per_item * batch;
  p.tap_groups = ceil_div(k, p.taps_block);
  p.tiles = ceil_div(co_g, bt) * ceil_div(ci_g, bt) * groups;
  // ~3 workgroups per CU, but at least 256 columns of reduction per workgroup
  int splits = ceil_div(768, p.tiles * p.tap_groups);
  const int min_chunks = 256 / p.tt > 1 ? 256 / p.tt : 1;
  if (splits > p.chunks_total / min_chunks) splits = p.chunks_total / min_chunks;
  if (splits < 1) splits = 1;
  p.splits = ceil_div(p.chunks_total, ceil_div(p.chunks_total, splits));
  return p;
}

template <int TG, bool SMALL, int TT>
static int launch_wgrad(WgArgs a, const WgPlan& p, float* dw_out, float* workspace, size_t ws_floats,
                        hipStream_t stream, double flops, double bytes) {
  a.xs_stride = p.xs_stride;
  a.chunks_per_item = p.chunks_per_item;
  a.chunks_total = p.chunks_total;
  const size_t lds = p.lds;
  PWG_REQUIRE(lds <= 160 * 1024, PWG_ERR_UNSUPPORTED, "conv1d_backward_weight: tile needs %zu B of LDS", lds);
  void (*kern)(WgArgs) = p.win ? conv1d_wgrad_kernel<TG, true, SMALL, TT> : conv1d_wgrad_kernel<TG, false, SMALL, TT>;
  if (lds > 64 * 1024 && !lds_limit_is_set(reinterpret_cast<const void*>(kern), lds)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    PWG_REQUIRE(e == hipSuccess, PWG_ERR_LAUNCH, "conv1d_backward_weight: cannot raise LDS limit: %s",
                hipGetErrorString(e));
  }
  a.chunks_per_block = ceil_div(a.chunks_total, p.splits);
  a.slab_elems = (long)a.co_g * a.groups * a.ci_g * a.k;
  if (p.splits == 1) {
    a.dw = dw_out;  // single slice: write the gradient directly
  } else {
    PWG_REQUIRE(workspace && ws_floats >= (size_t)p.splits * a.slab_elems, PWG_ERR_WORKSPACE,
                "conv1d_backward_weight: workspace of %zu floats needed, %zu given",
                (size_t)p.splits * a.slab_elems, ws_floats);
    a.dw = workspace;
  }
  dim3 grid(p.splits, p.tiles, p.tap_groups);
  {
    ProfScope prof(stream, "conv1d_wgrad_kernel", flops, bytes);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
  }
  PWG_CHECK_LAUNCH("conv1d_backward_weight");
  if (p.splits > 1) {
    long blocks = (a.slab_elems + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    ProfScope prof(stream, "reduce_slabs_kernel", 0, 4.0 * a.slab_elems * (p.splits + 1));
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((int)blocks), dim3(256), 0, stream, workspace, dw_out, a.slab_elems,
                       p.splits);
    PWG_CHECK_LAUNCH("reduce_slabs");
  }
  return PWG_OK;
}

}  // namespace pwg

using namespace pwg;

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

extern "C" size_t pwg_conv1d_backward_weight_workspace_floats(const pwg_conv1d_desc* d) {
  if (!d || d->groups <= 0 || d->c_in % d->groups || d->c_out % d->groups) return 0;
  int co_g, ci_g, n_cols;
  wgrad_roles(d, &co_g, &ci_g, &n_cols);
  const WgPlan p = wgrad_plan(co_g, ci_g, d->groups, d->kernel, d->stride, d->dilation, d->width, n_cols, d->batch);
  return p.splits > 1 ? (size_t)p.splits * co_g * d->groups * ci_g * d->kernel : 0;
}

extern "C" int pwg_conv1d_backward_weight(const pwg_conv1d_desc* d, const float* x, const float* dy,
                                          float* dw, float* db, float* workspace, size_t workspace_floats,
                                          void* stream_) {
  PWG_REQUIRE(d && x && dy, PWG_ERR_NULL, "conv1d_backward_weight: NULL pointer");
  PWG_REQUIRE(d->c_in % d->groups == 0 && d->c_out % d->groups == 0 && d->groups > 0, PWG_ERR_BAD_SHAPE,
              "conv1d_backward_weight: bad groups");
  PWG_REQUIRE(d->pad_mode == PWG_PAD_ZERO, PWG_ERR_UNSUPPORTED,
              "conv1d_backward_weight: only zero padding (pad reflect/replicate inputs explicitly)");
  hipStream_t stream = (hipStream_t)stream_;
  const long y_elems = (long)d->batch * d->c_out * d->t_out * d->width;
  const long x_elems = (long)d->batch * d->c_in * d->t_in * d->width;
  PWG_REQUIRE(y_elems * 4 < 0xFFFFFFF0L && x_elems * 4 < 0xFFFFFFF0L, PWG_ERR_UNSUPPORTED,
              "conv1d_backward_weight: tensors above 4 GiB need batch splitting");
  if (db) {
    const int n = d->t_out * d->width;
    const int segs = ceil_div(n, BG_SEG);
    (void)hipMemsetAsync(db, 0, sizeof(float) * d->c_out, stream);
    ProfScope prof(stream, "bias_grad_kernel", 0, 4.0 * y_elems);
    hipLaunchKernelGGL(bias_grad_kernel, dim3(d->c_out, d->batch * segs), dim3(256), 0, stream, dy, db, d->c_out, n,
                       segs);
    PWG_CHECK_LAUNCH("bias_grad");
  }
  if (!dw) return PWG_OK;
  WgArgs a;
  const float slope = d->pre_act == PWG_ACT_LEAKY_RELU ? d->pre_slope : (d->pre_act == PWG_ACT_RELU ? 0.f : 1.f);
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
  a.dw = nullptr;
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
    case 128: return launch_wgrad<TGV, SM, 128>(a, p, dw, workspace, workspace_floats, stream, flops, bytes); \
    case 64: return launch_wgrad<TGV, SM, 64>(a, p, dw, workspace, workspace_floats, stream, flops, bytes);   \
    default: return launch_wgrad<TGV, SM, 32>(a, p, dw, workspace, workspace_floats, stream, flops, bytes);   \
  }
  if (p.small) {
    switch (p.tg) {
      case 1: WG_CASE(1, true);
      case 2: WG_CASE(2, true);
      case 3: WG_CASE(3, true);
      case 4: WG_CASE(4, true);
      case 6: WG_CASE(6, true);
      default: WG_CASE(11, true);
    }
  }
  switch (p.tg) {
    case 4: WG_CASE(4, false);
    case 6: WG_CASE(6, false);
    case 7: WG_CASE(7, false);
    default: WG_CASE(8, false);
  }
#undef WG_CASE
}
