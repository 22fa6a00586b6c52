// conv1d.hip -- fp32 implicit-GEMM 1-D convolution family on gfx950 MFMA.
//
// One kernel template covers Conv1d (dilated/strided/grouped), ConvTranspose1d
// (polyphase: every output phase is a stride-1 conv with K/s taps; the phases
// are extra GEMM rows) and the (k,1) Conv2d of the period discriminator
// (rows of `width` samples are the time axis, the flat view is contiguous).
//
// GEMM view per (batch item, group):   D[m][n] = sum_{tap,ci} A[m][tap,ci] * B[tap,ci][n]
//   m = phase*Cout_g + co   (rows,   weights,   A operand)
//   n = output column       (cols,   time axis, B operand)
// The contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD,
// the fp32 matrix peak of 157 TFLOP/s); k-order = (ci-chunk, tap, ci), so the
// result equals an fmaf chain in that order (cdna_hip_programming.md s3).
//
// Data movement per workgroup (BM rows x BN columns, 4 waves):
//   x : (CK channels) x (BN*stride + halo) samples are read ONCE from HBM per
//       ci-chunk, coalesced along time, pre-activation (LeakyReLU) and padding
//       (zero/reflect/replicate) applied on the way into LDS, so each tap is a
//       shifted LDS read and not a re-read of HBM.
//   w : packed image [group][tap][ci][Mpad] (m fastest, Mpad % 128 == 0) -> LDS with 16-B loads;
//       lanes 0-31 read 32 consecutive rows m, lanes 32-63 the next ci:
//       conflict-free ds_read_b32 for both operands.
//   y : MFMA D layout has col = lane&31, i.e. 32 consecutive time samples per
//       half-wave -> coalesced 128-B stores; bias/residual/scale/tanh fused.
#include "common.h"

#include <stdint.h>
#include <stdlib.h>

namespace pwg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef PWG_PRIO
#define PWG_PRIO 0
#endif
constexpr int kPrio = PWG_PRIO;  // s_setprio by phase in conv1d_mfma_dma_kernel (tools/build_variant.py)

struct ConvArgs {
  const float* x;
  const float* wp;
  const float* bias;
  const float* add1;
  const float* add2;
  float* y;
  int cin_g;      // input channels per group
  int cin_pad;    // packed-weight ci extent (multiple of 16)
  int cout_g;     // real output channels per group
  int m_g;        // GEMM rows per group = phases * cout_g
  int m_pad;      // packed-weight m extent (multiple of 128)
  int t_in;       // input rows
  int t_out;      // output rows (real)
  int width;      // samples per row (1 for Conv1d)
  int k;          // taps of the (phase) convolution
  int stride;     // input rows advanced per output column-row
  int dil;
  int pad;        // left padding in rows
  int n_cols;     // GEMM columns per batch item = q_rows * width
  int out_stride; // output row u = q*out_stride + phase - out_off
  int out_off;
  int x_cstride;  // t_in * width
  int y_cstride;  // t_out * width
  long x_bstride;
  long y_bstride;
  int xs_stride;  // LDS row stride of the x tile (floats)
  int pad_mode, pre_act, post_act;
  float pre_slope, post_slope, out_mul, out_div;
  // backward-data only: multiply by d(pre_act)/dx evaluated at the forward input
  const float* mask_src;
  float mask_slope;
  int dbg;  // timing experiments only (PWG_DBG env): 1 = no DMA after chunk 0, 2 = no barriers, 4 = no epilogue
  // split-K (DMA kernel): the ci-chunks are cut into ksplit slices (blockIdx.z = item * ksplit + slice);
  // each slice stores its raw partial sums into its own y-shaped slab of `partial`, and
  // splitk_finish_kernel sums the slabs and applies the fused epilogue.
  int ksplit;
  float* partial;
  long slab_elems;
  int epi_vec;  // 1: y / add1 / add2 / mask are 16-B aligned with y_cstride % 4 == 0 -> LDS-transposed 16-B epilogue
  int item_major;  // logical tile order inside an XCD's run: 0 = row blocks of a column tile together (x window shared),
                   // 1 = the items of a (row block, reduction slice) together (weight chunk shared); choose_tile_order
};

template <int WM, int WN, int WAVES_M, int WAVES_N, int CK>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv1d_mfma_kernel(ConvArgs a) {
  constexpr int BM = 32 * WM * WAVES_M;
  constexpr int BN = 32 * WN * WAVES_N;
  constexpr int NWAVES = WAVES_M * WAVES_N;
  constexpr int NT = 64 * NWAVES;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int XS = a.xs_stride;
  float* xs = smem;            // [CK][XS]
  float* ws = smem + CK * XS;  // [k][CK][BM]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int l31 = lane & 31;
  const int lhi = lane >> 5;

  const int n0 = blockIdx.x * BN;
  const int mtiles = (a.m_g + BM - 1) / BM;
  const int g = blockIdx.y / mtiles;
  const int m0 = (blockIdx.y % mtiles) * BM;
  const int b = blockIdx.z;

  const int W = a.width;
  // rows of the input covered by this column tile
  const int h0 = n0 / W;
  int n_last = n0 + BN - 1;
  if (n_last > a.n_cols - 1) n_last = a.n_cols - 1;
  const int h1 = n_last / W;
  const int f0 = (h0 * a.stride - a.pad) * W;                                   // first flat input index
  const int L = ((h1 - h0) * a.stride + (a.k - 1) * a.dil + 1) * W;             // staged samples per channel
  const int tap_step = a.dil * W;
  const int in_len = a.t_in * W;

  // per-lane LDS column offsets of this wave's WN column sub-tiles
  int coff[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    int n = n0 + (wave_n * WN + ni) * 32 + l31;
    if (n > a.n_cols - 1) n = a.n_cols - 1;
    const int h = n / W;
    coff[ni] = (h - h0) * a.stride * W + (n - h * W);
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const float* xb = a.x + (long)b * a.x_bstride + (long)g * a.cin_g * a.x_cstride;
  const float* wg = a.wp + (long)g * a.k * a.cin_pad * a.m_pad;

  for (int ci0 = 0; ci0 < a.cin_g; ci0 += CK) {
    // ---- stage x chunk: CK channels x L samples (activation + padding fused)
    for (int r = wave; r < CK; r += NWAVES) {
      const int ci = ci0 + r;
      const float* xrow = xb + (long)ci * a.x_cstride;
      float* dst = xs + r * XS;
      const bool ch_ok = ci < a.cin_g;
      for (int i = lane; i < L; i += 64) {
        int f = f0 + i;
        float v = 0.f;
        if (ch_ok) {
          bool ok = (f >= 0) && (f < in_len);
          if (!ok && a.pad_mode != PWG_PAD_ZERO) {
            if (a.pad_mode == PWG_PAD_REFLECT) {
              f = f < 0 ? -f : 2 * (in_len - 1) - f;
            } else {
              f = f < 0 ? 0 : in_len - 1;
            }
            ok = (f >= 0) && (f < in_len);
          }
          if (ok) {
            v = xrow[f];
            if (a.pre_act == PWG_ACT_LEAKY_RELU)
              v = v > 0.f ? v : v * a.pre_slope;
            else if (a.pre_act == PWG_ACT_RELU)
              v = v > 0.f ? v : 0.f;
          }
        }
        dst[i] = v;
      }
    }
    // ---- stage w chunk: k x CK x BM floats, 16-B loads
    {
      constexpr int BM4 = BM / 4;
      const int total4 = a.k * CK * BM4;
      for (int idx = tid; idx < total4; idx += NT) {
        const int j4 = idx % BM4;
        const int rr = idx / BM4;  // tap*CK + r
        const int r = rr % CK;
        const int tap = rr / CK;
        const int m = m0 + j4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < a.m_pad)
          v = *reinterpret_cast<const float4*>(wg + ((long)tap * a.cin_pad + ci0 + r) * a.m_pad + m);
        *reinterpret_cast<float4*>(ws + rr * BM + j4 * 4) = v;
      }
    }
    __syncthreads();

    // ---- contraction over (tap, ci in chunk)
    for (int tap = 0; tap < a.k; ++tap) {
      const float* wt = ws + tap * CK * BM + wave_m * (WM * 32) + l31;
      const float* xt = xs + tap * tap_step;
#pragma unroll
      for (int kk = 0; kk < CK / 2; ++kk) {
        const int kr = 2 * kk + lhi;
        float av[WM], bv[WN];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) av[mi] = wt[kr * BM + mi * 32];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) bv[ni] = xt[kr * XS + coff[ni]];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const long ybase = (long)b * a.y_bstride;
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    const int n = n0 + (wave_n * WN + ni) * 32 + l31;
    if (n >= a.n_cols) continue;
    const int q = n / W;
    const int wcol = n - q * W;
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wave_m * WM + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m >= a.m_g) continue;
        const int phase = m / a.cout_g;
        const int co = m - phase * a.cout_g;
        const int u = q * a.out_stride + phase - a.out_off;
        if (u < 0 || u >= a.t_out) continue;
        const int cglob = g * a.cout_g + co;
        const long o = ybase + (long)cglob * a.y_cstride + (long)u * W + wcol;
        float v = acc[mi][ni][r];
        if (a.bias) v += a.bias[cglob];
        if (a.mask_src) v *= (a.mask_src[o] > 0.f ? 1.f : a.mask_slope);
        if (a.add1) v += a.add1[o];
        if (a.add2) v += a.add2[o];
        if (a.out_mul != 1.0f) v *= a.out_mul;
        if (a.out_div != 1.0f) v = v / a.out_div;
        v = apply_act(v, a.post_act, a.post_slope);
        a.y[o] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// v2: LDS-DMA staged, double-buffered variant (zero padding only).
//
//   * w chunk: global_load_lds_dwordx4 (1 KiB per wave-instruction, lane-linear LDS image
//     == the [tap][ci][BM] tile, no VGPR round trip)
//   * x chunk: buffer_load_dword ... lds through a per-channel-row raw buffer descriptor
//     whose num_records is the row length: samples left/right of the row (the implicit
//     zero padding) and channels past c_in are out of range and land as 0.0 in LDS
//     (probed on hardware: tools/probes/glds_oob.hip).  The pre-activation is applied
//     when the B operand is read from LDS (2-3 VALU ops per 64-cycle MFMA).
//   * chunk c+1 is in flight while chunk c is contracted: one vmcnt(0)+barrier per chunk.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// XCD-aware tile order: dispatch id -> logical tile (bx = column tile, by = group * mtiles + row block,
// bz = item * ksplit + reduction slice).  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs,
// each with its own 4 MB L2.  In grid order the row blocks of one column tile -- which stage the SAME x window -- sit
// gridDim.x ids apart, on different XCDs, and each pulls that window from HBM.  The remap hands every XCD one
// contiguous run of logical tiles, ordered
//   item_major = 0: row block fastest, then column tile, group, (item, slice): the row blocks of a column tile (and
//     the neighbouring column tiles, which share the halo) are resident on one XCD at the same time and meet in its L2;
//   item_major = 1: item fastest, then column tile, row block, group, slice -- for weight-heavy launches (few columns
//     per item under a long reduction: the 512 / 1024-channel discriminator layers at T = 9 .. 128), where the operand
//     worth sharing is the WEIGHT chunk: the workgroups that walk the same (row block, slice) of the weight image, one
//     per item and column tile, form one run on one XCD, so each XCD streams its 1/8 of the image from HBM instead of
//     all of it (choose_tile_order decides).
// A bijection for any grid size, so results do not depend on the dispatch assumption (tests/test_tile_order.py walks
// it on the host through pwg_debug_conv_tile_of_workgroup).
__host__ __device__ __forceinline__ void tile_of_workgroup(unsigned lin, unsigned gx, unsigned gy, unsigned gz,
                                                            unsigned mtiles, unsigned ksplit, int item_major, int& bx,
                                                            int& by, int& bz) {
  const unsigned total = gx * gy * gz;
  const unsigned per = total >> 3, rem = total & 7, xcd = lin & 7, seq = lin >> 3;
  lin = xcd < rem ? xcd * (per + 1) + seq : rem * (per + 1) + (xcd - rem) * per + seq;
  const unsigned ngroups = gy / mtiles;
  if (item_major) {
    const unsigned nb = gz / ksplit;
    const unsigned item = lin % nb;
    lin /= nb;
    bx = lin % gx;
    lin /= gx;
    const unsigned mi = lin % mtiles;
    lin /= mtiles;
    by = (lin % ngroups) * mtiles + mi;
    bz = item * ksplit + lin / ngroups;
  } else {
    const unsigned mi = lin % mtiles;
    lin /= mtiles;
    bx = lin % gx;
    lin /= gx;
    by = (lin % ngroups) * mtiles + mi;
    bz = lin / ngroups;
  }
}

// FAST = true: stride 1, width 1, halo (k-1)*dil <= 64.  The x-tile row stride is the compile-time
//   constant BN + 64, so every LDS operand address is (one VGPR base per tap) + immediate and the
//   two column sub-tiles of a wave come from one ds_read2_b32; the pre-activation is
//   ACT = 0 none / 1 LeakyReLU with 0 < slope < 1 as max(v, slope*v) (2 VALU) / 2 generic (3 VALU).
// FAST = false: any geometry (runtime row stride, per-lane column offsets), generic activation.
// (second launch bound: at least 2 waves per SIMD unless the wave owns 8 accumulator tiles -- hipcc's
// allocation for the 2x2-tile waves otherwise flips between 174 and 256 VGPRs on unrelated edits)
template <int WM, int WN, int WAVES_M, int WAVES_N, int CK, bool FAST, int ACT>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WM * WN >= 8 ? 1 : (WM * WN == 1 ? 4 : (WAVES_M * WAVES_N == 4 ? 3 : 2)))) void conv1d_mfma_dma_kernel(ConvArgs a) {
  constexpr bool SLIM = WM * WN >= 2 && WAVES_M * WAVES_N == 4;  // 3 waves per SIMD (168 VGPRs)
  constexpr int BM = 32 * WM * WAVES_M;
  constexpr int BN = 32 * WN * WAVES_N;
  constexpr int NWAVES = WAVES_M * WAVES_N;
  constexpr int ROWS_PER_PIECE = 256 / BM;  // w rows per 1 KiB DMA piece
  constexpr int L4 = BM / 4;                // lanes per w row
  constexpr int KS = CK / 2;                // MFMA k-steps per tap
  static_assert((CK % ROWS_PER_PIECE) == 0, "CK must cover whole DMA pieces");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int XS = FAST ? BN + 64 : a.xs_stride;
  const int buf_floats = CK * XS + a.k * CK * BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int l31 = lane & 31;
  const int lhi = lane >> 5;

  // XCD-aware tile order (tile_of_workgroup above; PWG_DBG bit 16 = grid order, for the A/B)
  const int mtiles = (a.m_g + BM - 1) / BM;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (!(a.dbg & 16))
    tile_of_workgroup(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x, gridDim.y, gridDim.z,
                      (unsigned)mtiles, (unsigned)a.ksplit, a.item_major, bx, by, bz);
  const int n0 = bx * BN;
  const int g = by / mtiles;
  const int m0 = (by % mtiles) * BM;
  const int b = bz / a.ksplit;
  const int ks = bz - b * a.ksplit;  // reduction slice of this workgroup

  const int W = a.width;
  const int h0 = n0 / W;
  int n_last = n0 + BN - 1;
  if (n_last > a.n_cols - 1) n_last = a.n_cols - 1;
  const int h1 = n_last / W;
  const int f0 = (h0 * a.stride - a.pad) * W;
  const int L = ((h1 - h0) * a.stride + (a.k - 1) * a.dil + 1) * W;
  const int tap_step = a.dil * W;
  int coff[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    int n = n0 + (wave_n * WN + ni) * 32 + l31;
    if (n > a.n_cols - 1) n = a.n_cols - 1;
    const int h = n / W;
    coff[ni] = (h - h0) * a.stride * W + (n - h * W);
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const float* xb = a.x + (long)b * a.x_bstride + (long)g * a.cin_g * a.x_cstride;
  const float* wg = a.wp + (long)g * a.k * a.cin_pad * a.m_pad + m0;
  const int npieces = a.k * CK / ROWS_PER_PIECE;
  const int w_row_in_piece = lane / L4;
  const int w_col = (lane % L4) * 4;

  // ONE buffer descriptor per workgroup, built from kernel arguments and blockIdx only so that the
  // compiler can prove it wave-uniform (a per-row descriptor made hipcc wrap every DMA in a
  // readfirstlane "waterfall" loop, cdna_hip_programming.md T20).  Padding / channel tails are
  // expressed per lane: an out-of-range lane gets the offset 0xFFFFFFFC, which is past num_records
  // and lands as 0.0 in LDS.
  __amdgpu_buffer_rsrc_t x_rs = uniform_buffer_rsrc(xb, (unsigned)(a.cin_g * a.x_cstride) * 4u);
  // Interior FAST tiles (the whole BN + 64 staged range inside the row): one 16-B-per-lane DMA per x row
  // instead of three 4-B ones.  `buffer_load_dwordx4 ... lds` is legal at 4-byte source alignment and its
  // range check is per dword at the END of the buffer but per access for negative offsets
  // (tools/probes/glds_x4.hip), so edge tiles keep the dword path below.
  // (generic path: the staged flat range [f0, f0 + L) rounded up to whole 16-B pieces; rows are XS apart,
  // XS a multiple of 64 floats there)
  const int lanes4 = FAST ? (BN + 64) / 4 : (L + 3) / 4;  // 16-B pieces per row
  const bool x4_ok = f0 >= 0 && f0 + 4 * lanes4 <= a.t_in * W;
  auto issue = [&](int ci0, float* buf) {
    float* xs = buf;
    float* ws = buf + CK * XS;
    if (__builtin_amdgcn_readfirstlane(x4_ok ? 1 : 0)) {
      const int LANES = lanes4;
      for (int r = wave; r < CK; r += NWAVES) {
        const int ci = ci0 + r;
        for (int l0 = 0; l0 < LANES; l0 += 64) {
          if (l0 + lane < LANES) {
            const unsigned off = ci < a.cin_g ? (unsigned)(ci * a.x_cstride + f0 + 4 * (l0 + lane)) * 4u : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS + 4 * l0), 16, off, 0, 0, 0);
          }
        }
      }
    } else
    for (int r = wave; r < CK; r += NWAVES) {
      const int ci = ci0 + r;
      const int rowoff = ci * a.x_cstride;
      for (int i0 = 0; i0 < L; i0 += 64) {
        const int f = f0 + i0 + lane;
        unsigned off = 0xFFFFFFFCu;
        if (ci < a.cin_g && f >= 0 && f < a.t_in * W) off = (unsigned)(rowoff + f) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS + i0), 4, off, 0, 0, 0);
      }
    }
    for (int p = wave; p < npieces; p += NWAVES) {
      const int rr = p * ROWS_PER_PIECE + w_row_in_piece;
      const int tap = rr / CK;
      const int r = rr - tap * CK;
      const float* src = wg + ((long)tap * a.cin_pad + ci0 + r) * a.m_pad + w_col;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(ws + p * 256), 16, 0, 0);
    }
  };

  const float slope_eff = a.pre_act == PWG_ACT_LEAKY_RELU ? a.pre_slope : (a.pre_act == PWG_ACT_RELU ? 0.f : 1.f);
  const int nchunks_all = (a.cin_g + CK - 1) / CK;
  const int per_slice = (nchunks_all + a.ksplit - 1) / a.ksplit;
  const int c_first = ks * per_slice;
  const int nchunks = max(0, min(per_slice, nchunks_all - c_first));
  if (nchunks > 0) issue(c_first * CK, smem);
  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(a.dbg & 2)) __syncthreads();
    float* buf = smem + (c & 1) * buf_floats;
    // (wave priority by phase, compile-time experiment PWG_PRIO -- the resident workgroups of a CU are at different
    // phases.  bit 0: matrix phase raised; bit 1: the DMA issue of the next chunk raised; bit 2: the epilogue raised)
    if constexpr (kPrio & 2) __builtin_amdgcn_s_setprio(1);
    if (c + 1 < nchunks && !(a.dbg & 1)) issue((c_first + c + 1) * CK, smem + ((c + 1) & 1) * buf_floats);
    if constexpr (kPrio & 2) __builtin_amdgcn_s_setprio(0);
    if constexpr (kPrio & 1) __builtin_amdgcn_s_setprio(1);
    const float* xs = buf;
    const float* wl = buf + CK * XS + wave_m * (WM * 32) + l31 + lhi * BM;  // + tap*CK*BM + 2*kk*BM + mi*32
    const float* xl = xs + lhi * XS + (FAST ? wave_n * (WN * 32) + l31 : 0);  // + tap*tap_step + 2*kk*XS + coff
    // operands of one tap live in registers; the next tap's LDS reads are issued before this
    // tap's MFMAs so that LDS latency hides under the 64-cycle matrix instructions
    auto load_ops = [&](int tap, float(&av)[KS][WM], float(&bv)[KS][WN]) {
      const float* wt = wl + tap * (CK * BM);
      const float* xt = xl + tap * tap_step;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) av[kk][mi] = wt[2 * kk * BM + mi * 32];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) bv[kk][ni] = FAST ? xt[2 * kk * XS + ni * 32] : xt[2 * kk * XS + coff[ni]];
      }
    };
    auto mma = [&](float(&av)[KS][WM], float(&bv)[KS][WN]) {
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        float bact[WN];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
          // branch-free pre-activation: v>0 ? v : v*slope  ==  max(v,0) + slope*min(v,0)  (exact:
          // one of the two terms is always 0); slope_eff = 1 (none) / slope (LeakyReLU) / 0 (ReLU)
          const float v = bv[kk][ni];
          if (ACT == 0)
            bact[ni] = v;
          else if (ACT == 1)
            bact[ni] = __builtin_fmaxf(v, v * slope_eff);  // LeakyReLU for 0 < slope < 1, exact
          else
            bact[ni] = __builtin_fmaf(slope_eff, __builtin_fminf(v, 0.f), __builtin_fmaxf(v, 0.f));
        }
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][mi], bact[ni], acc[mi][ni], 0, 0, 0);
      }
    };
    float a0[KS][WM], b0[KS][WN], a1[KS][WM], b1[KS][WN];
    load_ops(0, a0, b0);
    int tap = 0;
    for (; tap + 2 <= a.k; tap += 2) {
      load_ops(tap + 1, a1, b1);
      mma(a0, b0);
      if (tap + 2 < a.k) load_ops(tap + 2, a0, b0);
      mma(a1, b1);
    }
    if (tap < a.k) mma(a0, b0);
    if constexpr (kPrio & 1) __builtin_amdgcn_s_setprio(0);
  }
  if constexpr (kPrio & 4) __builtin_amdgcn_s_setprio(1);

  if (a.dbg & 4) {
    if (acc[0][0][0] == 12345.678f) a.y[0] = 1.f;
    return;
  }
  // ---- epilogue: D layout col = lane&31 (time), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel).
  // Per 32x32 accumulator tile: first ALL addend loads of the lane's 16 outputs are issued (they are
  // independent: y never aliases bias/add1/add2), then the arithmetic, then 16 stores -- memory
  // latency is paid once per tile instead of once per element.
  const long ybase = (long)b * a.y_bstride;
  const bool single_phase = a.m_g == a.cout_g;
  if (a.ksplit > 1) {
    // split-K slice: raw partial sums into this slice's y-shaped slab; bias / addends / activation
    // are applied by splitk_finish_kernel.  (Kept apart from the fused epilogue below so that its
    // register allocation -- 2 waves per SIMD for the 128x128 tiles -- is untouched.)
    float* __restrict__ slab = a.partial + (long)ks * a.slab_elems;
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
      const int n = n0 + (wave_n * WN + ni) * 32 + l31;
      const int q = n / W;
      const int wcol = n - q * W;
#pragma unroll
      for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wave_m * WM + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          int phase = 0, co = m;
          if (!single_phase) {
            phase = m / a.cout_g;
            co = m - phase * a.cout_g;
          }
          const int u = q * a.out_stride + phase - a.out_off;
          if (n < a.n_cols && m < a.m_g && u >= 0 && u < a.t_out)
            slab[ybase + (long)(g * a.cout_g + co) * a.y_cstride + (long)u * W + wcol] = acc[mi][ni][r];
        }
      }
    }
    return;
  }
  const float* __restrict__ bias_p = a.bias;
  const float* __restrict__ add1_p = a.add1;
  const float* __restrict__ add2_p = a.add2;
  const float* __restrict__ mask_p = a.mask_src;
  float* __restrict__ y_p = a.y;
  // ---- 16-B epilogue (plain Conv1d, wave tile fully inside the output).  The MFMA D layout gives a lane
  // ONE time sample of 16 rows, i.e. 16 dword stores (+16 dword loads per fused addend) per 32x32 tile:
  // the epilogue was issue-bound (ablation: 12.5 % of the generator forward).  Each wave transposes its
  // tile through a private 32 x 36 float LDS scratch (16 ds_write_b32 + 4 ds_read_b128) so that a lane owns
  // 4 consecutive time samples of a row: 4 dwordx4 stores / loads per tile, all loads of a tile issued
  // before its arithmetic.  Element-wise arithmetic and its order are those of the scalar path below.
  {
    const int m_w0 = m0 + wave_m * (WM * 32);
    const int n_w0 = n0 + wave_n * (WN * 32);
    const bool vec_ok = a.epi_vec && single_phase && W == 1 && a.out_off == 0 && a.out_stride == 1 && m_w0 + WM * 32 <= a.m_g && n_w0 + WN * 32 <= a.n_cols;
    if (__builtin_amdgcn_readfirstlane(vec_ok ? 1 : 0)) {
      // Scratch placement: the chunk buffer that the LAST chunk did not use is dead -- every wave passed the last
      // chunk's barrier, i.e. finished the chunk before it -- so when it is large enough the scratch lives there
      // and the launch needs no LDS beyond the two chunk buffers (128x128x8 at k = 7: 70 KB instead of 88 KB, two
      // workgroups per CU); small buffers keep a separate scratch behind them.
      const bool alias_scr = buf_floats >= NWAVES * (32 * 36) && nchunks >= 2;
      float* scratch = (alias_scr ? smem + (nchunks & 1) * buf_floats : smem + 2 * buf_floats) + wave * (32 * 36);
      const int trow = lane >> 3;        // + 8 * pass
      const int tcol = (lane & 7) * 4;
      const long tile_base = ybase + (long)(g * a.cout_g + m_w0) * a.y_cstride + n_w0;  // wave-uniform
      // the residual (add1) loads of ALL the wave's tiles are issued first (one exposure of the memory latency
      // instead of one per tile); add2 (last convolution of an MRF block) and the dgrad mask per tile
      // (three-waves-per-SIMD instantiations, 168 VGPRs: per tile instead -- the third resident workgroup hides it)
      constexpr bool PF_ALL = !SLIM;
      float4 a1v[PF_ALL ? WM : 1][PF_ALL ? WN : 1][4];
      if (PF_ALL) {
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
              const unsigned o = (unsigned)(mi * 32 + ps * 8 + trow) * (unsigned)a.y_cstride + (unsigned)(ni * 32 + tcol);
              if (add1_p) a1v[PF_ALL ? mi : 0][PF_ALL ? ni : 0][ps] = *reinterpret_cast<const float4*>(add1_p + tile_base + o);
            }
      }
#pragma unroll
      for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            scratch[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[mi][ni][r];
          float4 v[4], a2v[4], mkv[4];
          float bs[4];
          unsigned off[4];
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int row = mi * 32 + ps * 8 + trow;
            off[ps] = (unsigned)row * (unsigned)a.y_cstride + (unsigned)(ni * 32 + tcol);
            bs[ps] = bias_p ? bias_p[g * a.cout_g + m_w0 + row] : 0.f;
            if (!PF_ALL && add1_p) a1v[0][0][ps] = *reinterpret_cast<const float4*>(add1_p + tile_base + off[ps]);
            if (add2_p) a2v[ps] = *reinterpret_cast<const float4*>(add2_p + tile_base + off[ps]);
            if (mask_p) mkv[ps] = *reinterpret_cast<const float4*>(mask_p + tile_base + off[ps]);
          }
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) v[ps] = *reinterpret_cast<const float4*>(scratch + (ps * 8 + trow) * 36 + tcol);
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            float o[4] = {v[ps].x, v[ps].y, v[ps].z, v[ps].w};
            const float m4[4] = {mkv[ps].x, mkv[ps].y, mkv[ps].z, mkv[ps].w};
            const float4 a1t = a1v[PF_ALL ? mi : 0][PF_ALL ? ni : 0][ps];
            const float p4[4] = {a1t.x, a1t.y, a1t.z, a1t.w};
            const float q4[4] = {a2v[ps].x, a2v[ps].y, a2v[ps].z, a2v[ps].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = o[e] + bs[ps];
              if (mask_p) t *= (m4[e] > 0.f ? 1.f : a.mask_slope);
              if (add1_p) t += p4[e];
              if (add2_p) t += q4[e];
              if (a.out_mul != 1.0f) t *= a.out_mul;
              if (a.out_div != 1.0f) t = t / a.out_div;
              o[e] = apply_act(t, a.post_act, a.post_slope);
            }
            *reinterpret_cast<float4*>(y_p + tile_base + off[ps]) = make_float4(o[0], o[1], o[2], o[3]);
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    const int n = n0 + (wave_n * WN + ni) * 32 + l31;
    const bool n_ok = n < a.n_cols;
    const int q = n / W;
    const int wcol = n - q * W;
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
      // (RB outputs at a time: 8 for the single-tile waves that run at 4 waves per SIMD / 128 VGPRs)
      constexpr int RB = SLIM ? 4 : (WM * WN == 1 ? 8 : 16);
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += RB) {
        long off[RB];
        bool ok[RB];
        float bsv[RB], a1[RB], a2[RB], mk[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int r = r0 + j;
          const int m = m0 + (wave_m * WM + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          int phase = 0, co = m;
          if (!single_phase) {
            phase = m / a.cout_g;
            co = m - phase * a.cout_g;
          }
          const int u = q * a.out_stride + phase - a.out_off;
          const int cglob = g * a.cout_g + co;
          ok[j] = n_ok && m < a.m_g && u >= 0 && u < a.t_out;
          off[j] = ybase + (long)cglob * a.y_cstride + (long)u * W + wcol;
          bsv[j] = (bias_p && ok[j]) ? bias_p[cglob] : 0.f;
          a1[j] = (add1_p && ok[j]) ? add1_p[off[j]] : 0.f;
          a2[j] = (add2_p && ok[j]) ? add2_p[off[j]] : 0.f;
          mk[j] = (mask_p && ok[j]) ? mask_p[off[j]] : 1.f;
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          float v = acc[mi][ni][r0 + j] + bsv[j];
          if (mask_p) v *= (mk[j] > 0.f ? 1.f : a.mask_slope);
          v += a1[j];
          v += a2[j];
          if (a.out_mul != 1.0f) v *= a.out_mul;
          if (a.out_div != 1.0f) v = v / a.out_div;
          v = apply_act(v, a.post_act, a.post_slope);
          if (ok[j]) y_p[off[j]] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// split-K epilogue: y = post((sum_s partial[s] + bias) * mask' + add1 + add2) * out_mul / out_div)
// elementwise over the y layout (B, C_out, T_out*W); slabs are summed in slice order (deterministic).
// ---------------------------------------------------------------------------
struct FinishArgs {
  const float* partial;
  const float* bias;
  const float* add1;
  const float* add2;
  const float* mask_src;
  float* y;
  long elems, slab_elems;
  int nslabs, y_cstride, c_out;
  int post_act;
  float post_slope, out_mul, out_div, mask_slope;
};

__global__ __launch_bounds__(256) void splitk_finish_kernel(FinishArgs a) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < a.elems; e += (long)gridDim.x * 256L) {
    float v = a.partial[e];
    for (int s = 1; s < a.nslabs; ++s) v += a.partial[(long)s * a.slab_elems + e];
    if (a.bias) v += a.bias[(e / a.y_cstride) % a.c_out];
    if (a.mask_src) v *= (a.mask_src[e] > 0.f ? 1.f : a.mask_slope);
    if (a.add1) v += a.add1[e];
    if (a.add2) v += a.add2[e];
    if (a.out_mul != 1.0f) v *= a.out_mul;
    if (a.out_div != 1.0f) v = v / a.out_div;
    a.y[e] = apply_act(v, a.post_act, a.post_slope);
  }
}


// ---------------------------------------------------------------------------
// Few-output-channel convolutions (the generators' last layer: C -> 1, k = 7, at the full audio rate).
// On the MFMA tile a 1-row output wastes 31/32 of the matrix work (0.6 ms = 2.4 TFLOP/s for HiFi-GAN's
// 32 -> 1 layer at 3.3 M samples, although it is a 420 MB streaming read).  Here: one workgroup per
// (item, 1024-sample tile); the input rows go through LDS one channel at a time (pre-activation applied
// on the way), every thread accumulates 4 outputs (stride 256: conflict-free LDS reads) x <= 4 channels
// with plain fp32 FMAs in (ci, tap) order.  HBM-bound: one read of x, one write of y.
// ---------------------------------------------------------------------------
constexpr int SC_MAXW = 2048;  // cout * cin * k floats of weights in LDS
// OPT = outputs per thread (tile = 256 * OPT samples).  Round 4: the launch is a serial walk over the input channels with two
// barriers each, so few large tiles leave the chip idle -- PWG's last 64 -> 1 layer at B6 x 25600 was 150 workgroups and
// 116 us for a 39 MB read; the host picks the largest OPT that still gives >= 1024 workgroups.
template <int OPT>
__global__ __launch_bounds__(256) void conv1d_small_cout_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int cin, int cin_pad, int m_pad, int cout, int t_in,
                                                                int t_out, int k, int dil,
                                                                int pad, int pre_act, float pre_slope, int post_act,
                                                                float post_slope, float out_mul) {
  extern __shared__ float sm[];
  float* ws = sm;                      // [cout][cin][k]
  float* xs = sm + cout * cin * k;     // [tile + halo], tile = 256 * OPT samples
  const int b = blockIdx.y;
  constexpr int TILE = 256 * OPT;
  const int t0 = blockIdx.x * TILE;
  const int halo = (k - 1) * dil;
  const int L = TILE + halo;
  for (int i = threadIdx.x; i < cout * cin * k; i += 256) {  // from the packed image [tap][ci][m]
    const int tap = i % k, ci = (i / k) % cin, c = i / (k * cin);
    ws[i] = wp[((long)tap * cin_pad + ci) * m_pad + c];
  }
  float acc[OPT][4];
#pragma unroll
  for (int o = 0; o < OPT; ++o)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[o][c] = 0.f;
  const float* xb = x + (long)b * cin * t_in;
  for (int ci = 0; ci < cin; ++ci) {
    __syncthreads();  // previous channel consumed (and, first time, the weights staged)
    for (int i = threadIdx.x; i < L; i += 256) {
      const int f = t0 - pad + i;
      float v = (f >= 0 && f < t_in) ? xb[(long)ci * t_in + f] : 0.f;
      if (pre_act == PWG_ACT_LEAKY_RELU)
        v = v > 0.f ? v : v * pre_slope;
      else if (pre_act == PWG_ACT_RELU)
        v = v > 0.f ? v : 0.f;
      xs[i] = v;
    }
    __syncthreads();
    for (int tap = 0; tap < k; ++tap) {
      float xv[OPT];
#pragma unroll
      for (int o = 0; o < OPT; ++o) xv[o] = xs[threadIdx.x + 256 * o + tap * dil];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < cout) {
          const float wv = ws[(c * cin + ci) * k + tap];
#pragma unroll
          for (int o = 0; o < OPT; ++o) acc[o][c] = __builtin_fmaf(wv, xv[o], acc[o][c]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < cout) {
      const float bv = bias ? bias[c] : 0.f;
#pragma unroll
      for (int o = 0; o < OPT; ++o) {
        const int t = t0 + threadIdx.x + 256 * o;
        if (t < t_out) {
          float v = acc[o][c] + bv;
          if (out_mul != 1.0f) v *= out_mul;
          y[((long)b * cout + c) * t_out + t] = apply_act(v, post_act, post_slope);
        }
      }
    }
  }
}

// Round 6: the same layer as a pure stream (dilation 1, "same" padding (k - 1) / 2 <= 4, 16-B aligned rows): no LDS
// round trip and no barrier per input channel -- every thread owns 4 consecutive outputs, loads the 12-sample window
// [t - 4, t + 8) of each input row with three 16-B loads (neighbouring threads' windows overlap: the vector L1 serves
// the re-reads, HBM sees every byte once), applies the pre-activation in registers and accumulates in the SAME
// (ci, tap) order with the same fmaf chain as the kernel above (bit-identical results).  HiFi-GAN's 32 -> 1 k = 7
// output layer at B16 x 204 800: 230 us = 1.8 TB/s (0.29 of the 6.3 TB/s streaming rate) with the LDS kernel
// (profiles/r05_hbm_helpers.txt); this kernel: profiles/r06_hbm_helpers.txt.
template <int K, int PAD, int COUT>
__global__ __launch_bounds__(256) void conv1d_small_cout_stream_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                                       int cin, int cin_pad, int m_pad, int t_in, int t_out,
                                                                       int pre_act, float pre_slope, int post_act,
                                                                       float post_slope, float out_mul) {
  static_assert(PAD <= 4 && K - 1 - PAD <= 4, "window [t - 4, t + 8)");
  extern __shared__ float ws[];  // [cin][K][COUT]
  for (int i = threadIdx.x; i < cin * K * COUT; i += 256) {  // from the packed image [tap][ci][m]
    const int c = i % COUT, tap = (i / COUT) % K, ci = i / (COUT * K);
    ws[i] = wp[((long)tap * cin_pad + ci) * m_pad + c];
  }
  __syncthreads();
  const int b = blockIdx.y;
  const int t = blockIdx.x * 1024 + 4 * (int)threadIdx.x;
  if (t >= t_out) return;
  const float* xb = x + (long)b * cin * t_in;
  float acc[4][COUT];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[o][c] = 0.f;
  const bool interior = t >= 4 && t + 8 <= t_in;
  // LeakyReLU / ReLU as one select per value; slope_eff: 1 = no activation
  const float slope_eff = pre_act == PWG_ACT_LEAKY_RELU ? pre_slope : (pre_act == PWG_ACT_RELU ? 0.f : 1.f);
#pragma unroll 4
  for (int ci = 0; ci < cin; ++ci) {
    const float* row = xb + (long)ci * t_in;
    float w12[12];
    if (interior) {
      const float4 a = *reinterpret_cast<const float4*>(row + t - 4);
      const float4 m = *reinterpret_cast<const float4*>(row + t);
      const float4 e = *reinterpret_cast<const float4*>(row + t + 4);
      w12[0] = a.x; w12[1] = a.y; w12[2] = a.z; w12[3] = a.w;
      w12[4] = m.x; w12[5] = m.y; w12[6] = m.z; w12[7] = m.w;
      w12[8] = e.x; w12[9] = e.y; w12[10] = e.z; w12[11] = e.w;
    } else {
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int f = t - 4 + q;
        w12[q] = (f >= 0 && f < t_in) ? row[f] : 0.f;
      }
    }
#pragma unroll
    for (int q = 4 - PAD; q < 4 + 3 + K - PAD; ++q) w12[q] = w12[q] > 0.f ? w12[q] : w12[q] * slope_eff;
    const float* wc = ws + ci * (K * COUT);
#pragma unroll
    for (int tap = 0; tap < K; ++tap)
#pragma unroll
      for (int c = 0; c < COUT; ++c) {
        const float wv = wc[tap * COUT + c];
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o][c] = __builtin_fmaf(wv, w12[4 + o + tap - PAD], acc[o][c]);
      }
  }
#pragma unroll
  for (int c = 0; c < COUT; ++c) {
    const float bv = bias ? bias[c] : 0.f;
    float r[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float v = acc[o][c] + bv;
      if (out_mul != 1.0f) v *= out_mul;
      r[o] = apply_act(v, post_act, post_slope);
    }
    *reinterpret_cast<float4*>(y + ((long)b * COUT + c) * t_out + t) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// ---------------------------------------------------------------------------
// Single-input-channel convolutions at the audio rate: the first layer of every discriminator (HiFi-GAN MSD 1 -> 128,
// k = 15; MelGAN 1 -> 16, k = 15; PWG 1 -> 64, k = 3 -- reference models/hifigan.py:516-528, models/melgan.py:318-327,
// models/parallel_wavegan.py:300-316).  On the MFMA tile Cin = 1 fills one of the contraction's channel slots: 3 - 9
// TFLOP/s for what is a pure streaming write of the output (round 3: 0.65 ms of a C3 step, 0.92 ms of C4).  Here: one
// workgroup per (item, 1024-sample tile); the input tile goes to LDS once, every thread keeps the k taps of its 4
// output columns in registers and walks the output channels (weights: LDS broadcast reads), plain fp32 FMAs in tap order.
// HBM-bound: one read of x, one write of y.
// ---------------------------------------------------------------------------
constexpr int SI_TILE = 1024;
constexpr int SI_MAXK = 16;
// grid (time tiles, items, channel groups): the output channels are cut into gridDim.z groups so that a launch with
// few (item, tile) pairs still fills the chip -- HiFi-GAN's scale discriminators at the training batch are 16 items x
// 8 tiles = 128 workgroups each walking 128 channels (round 4: 126 us = 0.5 TB/s; the write of y is the whole job).
// VEC: every thread owns 4 CONSECUTIVE columns and stores them as one 16-B piece (t_out % 4 == 0, y 16-B aligned);
// else the columns of a thread are 256 apart (4-B stores, each wave instruction one contiguous 256-B run).
// (round 6 A/B, profiles/r06_hbm_helpers.txt: non-temporal stores 28.7 -> 27.6 us at the C3 shape, 512 / 2048 / 256
// workgroups per launch instead of 1024: 40 / 33 / 65 us -- neither kept)
template <bool VEC>
__global__ __launch_bounds__(256) void conv1d_small_cin_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int cin_pad, int m_pad, int cout, int cg, int t_in, int t_out,
                                                               int k, int dil, int pad, int pre_act, float pre_slope,
                                                               int post_act, float post_slope, float out_mul) {
  extern __shared__ float sm[];
  float* ws = sm;            // [cg][k]
  float* bs = sm + cg * k;   // [cg]
  float* xs = bs + cg;       // [SI_TILE + halo]
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * SI_TILE;
  const int c0 = blockIdx.z * cg;
  const int nc = min(cg, cout - c0);
  const int L = SI_TILE + (k - 1) * dil;
  for (int i = threadIdx.x; i < nc * k; i += 256) {  // from the packed image [tap][ci][m], ci = 0
    const int tap = i % k, c = i / k;
    ws[i] = wp[(long)tap * cin_pad * m_pad + c0 + c];
  }
  for (int i = threadIdx.x; i < nc; i += 256) bs[i] = bias ? bias[c0 + i] : 0.f;
  const float* xb = x + (long)b * t_in;
  for (int i = threadIdx.x; i < L; i += 256) {
    const int f = t0 - pad + i;
    float v = (f >= 0 && f < t_in) ? xb[f] : 0.f;
    if (pre_act == PWG_ACT_LEAKY_RELU)
      v = v > 0.f ? v : v * pre_slope;
    else if (pre_act == PWG_ACT_RELU)
      v = v > 0.f ? v : 0.f;
    xs[i] = v;
  }
  __syncthreads();
  // local column of output o of this thread
  const int col0 = VEC ? 4 * (int)threadIdx.x : (int)threadIdx.x;
  constexpr int CSTEP = VEC ? 1 : 256;
  float xv[4][SI_MAXK];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int tap = 0; tap < SI_MAXK; ++tap) xv[o][tap] = tap < k ? xs[col0 + CSTEP * o + tap * dil] : 0.f;
  float* yb = y + ((long)b * cout + c0) * t_out;
  for (int c = 0; c < nc; ++c) {
    const float bv = bs[c];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < SI_MAXK; ++tap) {
      if (tap < k) {
        const float wv = ws[c * k + tap];
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = __builtin_fmaf(wv, xv[o][tap], acc[o]);
      }
    }
    float r[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float v = acc[o] + bv;
      if (out_mul != 1.0f) v *= out_mul;
      r[o] = apply_act(v, post_act, post_slope);
    }
    if (VEC) {
      const int t = t0 + col0;
      if (t < t_out) *reinterpret_cast<float4*>(yb + (long)c * t_out + t) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int t = t0 + col0 + 256 * o;
        if (t < t_out) yb[(long)c * t_out + t] = r[o];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// weight packing:  torch layout -> [group][tap][ci (pad 16)][m (pad 32)]
// ---------------------------------------------------------------------------
struct PackArgs {
  const float* w;
  const float* scale;
  float* wp;
  int groups, k_phase, cin_g, cin_pad, cout_g, m_g, m_pad;
  int kernel, stride, transposed;
};

__global__ void pack_weight_kernel(PackArgs a) {
  const long total = (long)a.groups * a.k_phase * a.cin_pad * a.m_pad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int m = i % a.m_pad;
    long r = i / a.m_pad;
    const int ci = r % a.cin_pad;
    r /= a.cin_pad;
    const int tap = r % a.k_phase;
    const int g = r / a.k_phase;
    float v = 0.f;
    if (m < a.m_g && ci < a.cin_g) {
      if (!a.transposed) {
        // w: (c_out, cin_g, kernel); dim0 = global out channel
        const int co = g * a.cout_g + m;
        v = a.w[((long)co * a.cin_g + ci) * a.kernel + tap];
        if (a.scale) v *= a.scale[co];
      } else {
        // w: (c_in, cout_g, kernel); phase conv tap' reads x[q + tap' - (J-1)]
        // and multiplies the original tap k = phase + (J-1-tap')*stride
        const int phase = m / a.cout_g;
        const int co = m - phase * a.cout_g;
        const int kk = phase + (a.k_phase - 1 - tap) * a.stride;
        if (kk < a.kernel) {
          const int cig = g * a.cin_g + ci;
          v = a.w[((long)cig * a.cout_g + co) * a.kernel + kk];
          if (a.scale) v *= a.scale[cig];
        }
      }
    }
    a.wp[i] = v;
  }
}

// one block per dim-0 slice: scale = g / ||v||
__global__ void weight_norm_scale_kernel(const float* v, const float* g, float* scale, int inner) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const float* p = v + (long)row * inner;
  float s = 0.f;
  for (int i = threadIdx.x; i < inner; i += blockDim.x) {
    const float t = p[i];
    s += t * t;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    scale[row] = g[row] / sqrtf(t);
  }
}

__global__ void scale_rows_kernel(const float* v, const float* scale, float* w, long total, int inner) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x)
    w[i] = v[i] * scale[i / inner];
}

// ---------------------------------------------------------------------------
// Weight bank (round 4): the weight preparation of a WHOLE model -- weight-norm row scales, packed forward images,
// packed data-gradient images of every convolution -- as two launches over a device table instead of one
// weight_norm_scale + pack + pack launch sequence per layer (HiFi-GAN V1 training step: 551 of them, 4.8 ms of
// kernel time, all of it 3-10 us kernels).  Same arithmetic as the per-layer kernels above, element for element.
// ---------------------------------------------------------------------------
struct BankRows {   // one per weight-normalised layer
  const float* v;
  const float* g;
  float* scale;
  int inner, row0;  // floats per dim-0 slice; first global row of this layer
};
struct BankImage {  // one per packed image
  PackArgs a;
  int block0;       // first workgroup of this image in the pack launch
  int col_tiles;    // > 0: plain convolution, 64 x 64 LDS transposition tiles, this many per row block;
                    // < 0: polyphase image, -col_tiles chunks per input-channel row; 0: element-wise packing
};
constexpr int BANK_ELEMS_PER_BLOCK = 2048;
constexpr int BANK_TT = 64;  // transposition tile

// last table entry whose `first` field (row0 / block0) is <= id  (entries are sorted by it, the first is 0)
template <typename T, typename F>
__device__ __forceinline__ int bank_find(const T* tab, int n, int id, F first) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first(tab[mid]) <= id) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void bank_scale_kernel(const BankRows* __restrict__ tab, int n) {
  __shared__ float red[4];
  const int li = bank_find(tab, n, (int)blockIdx.x, [](const BankRows& r) { return r.row0; });
  const BankRows r = tab[li];
  const int row = (int)blockIdx.x - r.row0;
  const float* p = r.v + (long)row * r.inner;
  float s = 0.f;
  for (int i = threadIdx.x; i < r.inner; i += blockDim.x) {
    const float t = p[i];
    s += t * t;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 4; ++i) t += red[i];
    r.scale[row] = r.g[row] / sqrtf(t);
  }
}

// Plain convolutions: the torch layout (co, ci, tap) has the contraction index fastest, the packed image the output
// channel.  Element-wise packing reads one 4-byte word per 64-byte sector (PMC, round 4: 1.7 GB fetched per launch for
// 0.26 GB written on the HiFi-GAN discriminator, profiles/r04_train_pmc_summary.json).  A 64 x 64 tile through LDS reads
// rows of 256 B and writes runs of 64 output channels; the zero padding of the image (ci >= cin_g, m >= m_g) is never
// touched -- the bank zero-fills its persistent buffers once.
__device__ __forceinline__ void bank_pack_tile(const PackArgs& a, int tile, int col_tiles) {
  __shared__ float t[BANK_TT][BANK_TT + 1];
  const int inner = a.cin_g * a.kernel;           // floats per output channel
  const int n0 = a.groups * a.cout_g;             // output channels
  const int r0 = (tile / col_tiles) * BANK_TT, c0 = (tile % col_tiles) * BANK_TT;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < BANK_TT; r += 4) {
    const int co = r0 + r, c = c0 + tx;
    float v = 0.f;
    if (co < n0 && c < inner) {
      v = a.w[(long)co * inner + c];
      if (a.scale) v *= a.scale[co];
    }
    t[r][tx] = v;
  }
  __syncthreads();
  const int co = r0 + tx;
  if (co < n0) {
    const int g = co / a.cout_g, m = co - g * a.cout_g;
    for (int cc = ty; cc < BANK_TT; cc += 4) {
      const int c = c0 + cc;
      if (c < inner) {
        const int ci = c / a.kernel, tap = c - ci * a.kernel;
        a.wp[(((long)g * a.k_phase + tap) * a.cin_pad + ci) * a.m_pad + m] = t[tx][cc];
      }
    }
  }
}

// Polyphase (transposed-convolution) images -- the data-gradient image of every plain convolution: the torch layout is
// (cig, co, kk) with the tap fastest, the image wants runs of co for a fixed (cig, kk).  One workgroup stages
// floor(1024 / kernel) channels x kernel taps of one input-channel row (a contiguous read) and writes one run per tap.
constexpr int BANK_CHUNK = 1024;
__device__ __forceinline__ void bank_pack_row_chunk(const PackArgs& a, int blk, int chunks) {
  __shared__ float t[BANK_CHUNK];
  const int cig = blk / chunks, chunk = blk - cig * chunks;
  const int co_per = BANK_CHUNK / a.kernel;
  const int co0 = chunk * co_per;
  const int nco = min(co_per, a.cout_g - co0);
  const int nel = nco * a.kernel;
  const float sc = a.scale ? a.scale[cig] : 1.f;
  const float* src = a.w + ((long)cig * a.cout_g + co0) * a.kernel;
  for (int e = threadIdx.x; e < nel; e += 256) t[e] = src[e] * sc;
  __syncthreads();
  const int g = cig / a.cin_g, ci = cig - g * a.cin_g;
  for (int e = threadIdx.x; e < nel; e += 256) {
    const int kk = e / nco, j = e - kk * nco;  // tap-major over the chunk: consecutive threads, consecutive channels
    const int phase = kk % a.stride, tap = a.k_phase - 1 - kk / a.stride;
    a.wp[(((long)g * a.k_phase + tap) * a.cin_pad + ci) * a.m_pad + phase * a.cout_g + co0 + j] = t[j * a.kernel + kk];
  }
}

__global__ __launch_bounds__(256) void bank_pack_kernel(const BankImage* __restrict__ tab, int n) {
  const int ii = bank_find(tab, n, (int)blockIdx.x, [](const BankImage& e) { return e.block0; });
  const PackArgs a = tab[ii].a;
  if (tab[ii].col_tiles > 0) {  // (uniform per workgroup)
    bank_pack_tile(a, (int)blockIdx.x - tab[ii].block0, tab[ii].col_tiles);
    return;
  }
  if (tab[ii].col_tiles < 0) {
    bank_pack_row_chunk(a, (int)blockIdx.x - tab[ii].block0, -tab[ii].col_tiles);
    return;
  }
  const long total = (long)a.groups * a.k_phase * a.cin_pad * a.m_pad;
  const long base = (long)((int)blockIdx.x - tab[ii].block0) * BANK_ELEMS_PER_BLOCK;
#pragma unroll 2
  for (int j = 0; j < BANK_ELEMS_PER_BLOCK / 256; ++j) {
    const long i = base + j * 256 + threadIdx.x;
    if (i >= total) break;
    const int m = i % a.m_pad;
    long r = i / a.m_pad;
    const int ci = r % a.cin_pad;
    r /= a.cin_pad;
    const int tap = r % a.k_phase;
    const int g = r / a.k_phase;
    float v = 0.f;
    if (m < a.m_g && ci < a.cin_g) {
      if (!a.transposed) {
        const int co = g * a.cout_g + m;
        v = a.w[((long)co * a.cin_g + ci) * a.kernel + tap];
        if (a.scale) v *= a.scale[co];
      } else {
        const int phase = m / a.cout_g;
        const int co = m - phase * a.cout_g;
        const int kk = phase + (a.k_phase - 1 - tap) * a.stride;
        if (kk < a.kernel) {
          const int cig = g * a.cin_g + ci;
          v = a.w[((long)cig * a.cout_g + co) * a.kernel + kk];
          if (a.scale) v *= a.scale[cig];
        }
      }
    }
    a.wp[i] = v;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct Geometry {
  int cin_g, cin_pad, cout_g, phases, k_phase, m_g, m_pad;
  int stride, dil, pad, n_cols, out_stride, out_off;
};

static int make_geometry(const pwg_conv1d_desc* d, Geometry* g) {
  PWG_REQUIRE(d != nullptr, PWG_ERR_NULL, "conv1d: NULL descriptor");
  PWG_REQUIRE(d->batch > 0 && d->c_in > 0 && d->c_out > 0 && d->t_in > 0 && d->t_out > 0,
              PWG_ERR_BAD_SHAPE, "conv1d: non-positive size (B=%d Cin=%d Cout=%d Tin=%d Tout=%d)",
              d->batch, d->c_in, d->c_out, d->t_in, d->t_out);
  PWG_REQUIRE(d->kernel > 0 && d->stride > 0 && d->dilation > 0 && d->groups > 0 && d->width > 0,
              PWG_ERR_BAD_SHAPE, "conv1d: kernel/stride/dilation/groups/width must be positive");
  PWG_REQUIRE(d->c_in % d->groups == 0 && d->c_out % d->groups == 0, PWG_ERR_BAD_SHAPE,
              "conv1d: channels (%d,%d) not divisible by groups %d", d->c_in, d->c_out, d->groups);
  PWG_REQUIRE(d->pad_left >= 0, PWG_ERR_BAD_SHAPE, "conv1d: negative padding");
  g->cin_g = d->c_in / d->groups;
  g->cout_g = d->c_out / d->groups;
  g->cin_pad = round_up(g->cin_g, 16);
  if (!d->transposed) {
    const int need = (d->t_out - 1) * d->stride + (d->kernel - 1) * d->dilation + 1;
    PWG_REQUIRE((d->t_out - 1) * d->stride - d->pad_left < d->t_in, PWG_ERR_BAD_SHAPE,
                "conv1d: t_out=%d starts past the end of t_in=%d", d->t_out, d->t_in);
    if (d->pad_mode == PWG_PAD_REFLECT) {
      const int right = need - d->pad_left - d->t_in;
      PWG_REQUIRE(d->width == 1 && d->pad_left < d->t_in && right < d->t_in, PWG_ERR_BAD_SHAPE,
                  "conv1d: reflect padding (%d,%d) must be smaller than t_in=%d and width==1",
                  d->pad_left, right, d->t_in);
    }
    if (d->pad_mode == PWG_PAD_REPLICATE)
      PWG_REQUIRE(d->width == 1, PWG_ERR_BAD_SHAPE, "conv1d: replicate padding needs width==1");
    g->phases = 1;
    g->k_phase = d->kernel;
    g->stride = d->stride;
    g->dil = d->dilation;
    g->pad = d->pad_left;
    g->n_cols = d->t_out * d->width;
    g->out_stride = 1;
    g->out_off = 0;
  } else {
    PWG_REQUIRE(d->pad_mode == PWG_PAD_ZERO, PWG_ERR_UNSUPPORTED, "conv_transpose1d: zero padding only");
    PWG_REQUIRE(d->dilation == 1 || d->stride == 1, PWG_ERR_UNSUPPORTED,
                "conv_transpose1d: dilation > 1 is supported for stride 1 only");
    // polyphase: output row u = q*s + r - pad; phase r is a stride-1 conv with J taps over x[q-m]
    g->phases = d->stride;
    g->k_phase = ceil_div(d->kernel, d->stride);
    g->stride = 1;
    g->dil = d->dilation;  // != 1 only when stride == 1
    g->pad = (g->k_phase - 1) * g->dil;
    const int q_rows = (d->t_out - 1 + d->pad_left) / d->stride + 1;  // covers every u < t_out
    g->n_cols = q_rows * d->width;
    g->out_stride = d->stride;
    g->out_off = d->pad_left;
    if (d->stride == 1 && g->pad >= d->pad_left) {
      // stride 1 (the data gradient of every stride-1 convolution): substitute q = u + pad_left, i.e. the
      // transposed convolution IS a plain one with padding (k-1)*dil - pad_left over the flipped taps (which
      // is how the backward image is packed).  Columns are then the output samples themselves: no
      // `pad_left` wasted leading columns, tile columns aligned with y (16-B epilogue, FAST row stride).
      g->pad -= d->pad_left;
      g->n_cols = d->t_out * d->width;
      g->out_off = 0;
    }
  }
  g->m_g = g->phases * g->cout_g;
  g->m_pad = round_up(g->m_g, 128);
  return PWG_OK;
}

// configurations that also have FAST instantiations (the ones the heuristic picks for the G stacks)
template <int WM, int WN, int WAVES_M, int WAVES_N, int CK>
struct HasFast {
  static constexpr bool value = (WM == 2 && WN == 2 && WAVES_M == 2 && WAVES_N == 2 && CK == 4) ||   // 128x128x4
                                (WM == 2 && WN == 1 && WAVES_M == 2 && WAVES_N == 2 && CK == 8) ||   // 128x64x8
                                (WM == 1 && WN == 1 && WAVES_M == 1 && WAVES_N == 4 && CK == 8) ||   // 32x128x8
                                (WM == 2 && WN == 2 && WAVES_M == 1 && WAVES_N == 4 && CK == 4) ||   // 64x256x4
                                (WM == 1 && WN == 2 && WAVES_M == 2 && WAVES_N == 2 && CK == 8) ||   // 64x128x8
                                (WM == 1 && WN == 1 && WAVES_M == 2 && WAVES_N == 2 && CK == 8) ||   // 64x64x8
                                (WM == 1 && WN == 1 && WAVES_M == 2 && WAVES_N == 2 && CK == 16) ||  // 64x64x16
                                (WM == 1 && WN == 1 && WAVES_M == 1 && WAVES_N == 4 && CK == 16) ||  // 32x128x16
                                (WM == 1 && WN == 1 && WAVES_M == 2 && WAVES_N == 2 && CK == 4) ||   // 64x64x4 (k = 41 groups)
                                (WM == 2 && WN == 2 && WAVES_M == 2 && WAVES_N == 4 && (CK == 4 || CK == 8));  // 128x256, 8 waves
};

template <int WM, int WN, int WAVES_M, int WAVES_N, int CK>
static void (*pick_dma_kernel(bool fast, int act))(ConvArgs) {
  if constexpr (HasFast<WM, WN, WAVES_M, WAVES_N, CK>::value) {
    if (fast) {
      if (act == 0) return conv1d_mfma_dma_kernel<WM, WN, WAVES_M, WAVES_N, CK, true, 0>;
      if (act == 1) return conv1d_mfma_dma_kernel<WM, WN, WAVES_M, WAVES_N, CK, true, 1>;
      return conv1d_mfma_dma_kernel<WM, WN, WAVES_M, WAVES_N, CK, true, 2>;
    }
  }
  return conv1d_mfma_dma_kernel<WM, WN, WAVES_M, WAVES_N, CK, false, 2>;
}

// LDS behind the two chunk buffers for the epilogue's transposition scratch: none for split-K launches (their
// epilogue stores raw partial sums, no transposition) and none when the scratch fits the dead chunk buffer (mirrors
// the kernel's `alias_scr`: unsplit launches have nchunks = ceil(cin_g / CK) in every workgroup)
static size_t scratch_bytes_needed(size_t buf_bytes, size_t scr_bytes, int cin_g, int ck, int ksplit) {
  if (ksplit > 1) return 0;
  return (buf_bytes >= scr_bytes && ceil_div(cin_g, ck) >= 2) ? 0 : scr_bytes;
}

// Logical tile order of a launch (ConvArgs::item_major, tile_of_workgroup).  What an XCD pulls into its L2 is the set
// of DISTINCT weight tiles (row block x group x reduction slice: k * Cin_slice * BM floats) and x windows (item x
// column tile x group x slice: Cin_slice * staged columns) its contiguous run of total / 8 logical tiles touches; the
// two orders are scored by those bytes and the item-major one is taken when it pulls less than 0.8 x the bytes of the
// x-window-major one (the order every layer ran with up to round 4 keeps the ties).  Examples at B = 16:
//   1024 -> 1024 k = 5, T = 32, 128 x 32 tiles, 4 slices:  21.3 MB per XCD -> 3.2 MB  (item-major)
//   128 -> 128 k = 11, T = 51200 (inference), 64 x 256:    55 MB per XCD vs 109 MB      (x-window-major)
// MEASURED (profiles/r05_tile_order_ab.txt, tools/bench_tile_order.py): outputs bit-identical, and NO change in time
// on any of the 29 weight-heavy shapes (1024 -> 1024 k = 5 at T = 9 / 17 / 32: 81 - 86 us either way, warm or with the
// Infinity Cache flushed) nor on the captured C3 / C5 steps (48.98 vs 49.10 ms, 45.71 vs 45.92 ms: run-to-run spread)
// -- where the weights come from (HBM, Infinity Cache, the XCD's own L2) does not bound these launches.  The
// x-window-major order therefore stays the default; PWG_TILE_ORDER=2 applies the model, = 1 forces item-major.
static int choose_tile_order(const Geometry& g, int width, int t_in, int bm, int bn, int gx, int mtiles, int groups,
                             int batch, int ksplit) {
  static const int mode = getenv("PWG_TILE_ORDER") ? atoi(getenv("PWG_TILE_ORDER")) : 0;
  if (mode == 1) return 1;
  if (mode != 2) return 0;
  if (batch < 2) return 0;
  const double total = (double)gx * mtiles * groups * batch * ksplit;
  const double run = total / 8.0 < 1.0 ? 1.0 : total / 8.0;
  auto cdiv = [](double a, double b) { double q = a / b; double f = (double)(long)q; return f < q ? f + 1.0 : f; };
  auto dmin = [](double a, double b) { return a < b ? a : b; };
  const double cin_slice = cdiv((double)g.cin_g, (double)ksplit);
  const int rows = (width == 1) ? bn : ((bn - 1) / width + 2);
  double xwin = (double)((rows - 1) * g.stride + (g.k_phase - 1) * g.dil + 1) * width;
  xwin = dmin(xwin, (double)t_in * width);
  const double wt = 4.0 * g.k_phase * cin_slice * bm;
  const double xt = 4.0 * cin_slice * xwin;
  const double gs = (double)groups * ksplit;
  // x-window-major: row block, column tile, group, slice, item
  const double w0 = dmin(run, mtiles) * dmin(gs, cdiv(run, (double)mtiles * gx));
  const double x0 = cdiv(run, mtiles);
  // item-major: item, column tile, row block, group, slice
  const double ig = (double)batch * gx;
  const double w1 = cdiv(run, ig);
  const double x1 = dmin(run, ig) * dmin(gs, cdiv(run, ig * mtiles));
  const double bytes0 = w0 * wt + x0 * xt, bytes1 = w1 * wt + x1 * xt;
  return bytes1 < 0.8 * bytes0 ? 1 : 0;
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int CK, bool DMA>
static int launch_conv(const ConvArgs& a0, const Geometry& g, int batch, int groups, hipStream_t stream) {
  constexpr int BM = 32 * WM * WAVES_M;
  constexpr int BN = 32 * WN * WAVES_N;
  ConvArgs a = a0;
  const int W = a.width;
  const int rows = (W == 1) ? BN : ((BN - 1) / W + 2);
  const int xs_len = ((rows - 1) * g.stride + (g.k_phase - 1) * g.dil + 1) * W;
  // DMA variant: whole 64-lane pieces per row; register variant: 16-B aligned weight tile
  a.xs_stride = DMA ? round_up(xs_len, 64) : round_up(xs_len, 4);
  const bool fast = DMA && HasFast<WM, WN, WAVES_M, WAVES_N, CK>::value && g.stride == 1 && W == 1 &&
                    (g.k_phase - 1) * g.dil <= 64;
  if (fast) a.xs_stride = BN + 64;
  const int act = a.pre_act == PWG_ACT_NONE ? 0
                  : (a.pre_act == PWG_ACT_LEAKY_RELU && a.pre_slope > 0.f && a.pre_slope < 1.f) ? 1 : 2;
  const size_t buf = ((size_t)CK * a.xs_stride + (size_t)g.k_phase * CK * BM) * sizeof(float);
  // DMA kernel: two chunk buffers + one 32 x 36 float transposition scratch per wave (16-B epilogue), which lives
  // in the dead chunk buffer when that is large enough (and the reduction has >= 2 chunks per slice)
  const size_t scr = (size_t)WAVES_M * WAVES_N * 32 * 36 * sizeof(float);
  const size_t lds = DMA ? 2 * buf + scratch_bytes_needed(buf, scr, g.cin_g, CK, a.ksplit) : buf;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  a.epi_vec = DMA && a.width == 1 && (a.y_cstride % 4) == 0 && al16(a.y) && al16(a.add1) && al16(a.add2) &&
              al16(a.mask_src) && (((size_t)a.y_bstride) % 4) == 0;
  static const bool no_vec = getenv("PWG_NO_VEC_EPILOGUE") != nullptr;
  if (no_vec) a.epi_vec = 0;
  PWG_REQUIRE(lds <= 160 * 1024, PWG_ERR_UNSUPPORTED,
              "conv1d: tile needs %zu B of LDS (k=%d stride=%d dil=%d)", lds, g.k_phase, g.stride, g.dil);
  void (*kern)(ConvArgs);
  if (DMA)
    kern = pick_dma_kernel<WM, WN, WAVES_M, WAVES_N, CK>(fast, act);
  else
    kern = conv1d_mfma_kernel<WM, WN, WAVES_M, WAVES_N, CK>;
  if (lds > 64 * 1024 && !lds_limit_is_set(reinterpret_cast<const void*>(kern), lds)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    PWG_REQUIRE(e == hipSuccess, PWG_ERR_LAUNCH, "conv1d: cannot raise LDS limit to %zu: %s", lds,
                hipGetErrorString(e));
  }
  if (!DMA) a.ksplit = 1;  // (the register-staged kernel has no split-K)
  dim3 grid(ceil_div(g.n_cols, BN), ceil_div(g.m_g, BM) * groups, batch * a.ksplit);
  dim3 block(64 * WAVES_M * WAVES_N);
  a.item_major = DMA ? choose_tile_order(g, W, a.t_in, BM, BN, (int)grid.x, ceil_div(g.m_g, BM), groups, batch, a.ksplit) : 0;
  // algorithmic work of this launch: 2*taps*Cin_g MACs per real output element, and one read of
  // x / one write of y / one read of each fused addend / one read of the packed weights
  const double out_elems = (double)batch * groups * g.cout_g * a.t_out * a.width;
  const double taps_eff = (double)g.k_phase * g.phases / g.out_stride;  // transposed: k/stride taps hit each output
  const double flops = 2.0 * out_elems * g.cin_g * taps_eff;
  const double bytes = 4.0 * ((double)batch * groups * g.cin_g * a.t_in * a.width +
                              out_elems * (1 + (a.add1 != nullptr) + (a.add2 != nullptr)) +
                              (double)groups * g.k_phase * g.cin_g * g.m_g);
  maybe_poison_lds(stream);
  {
    ProfScope prof(stream,
                   prof_shape_name(DMA ? "conv1d_mfma_dma_kernel" : "conv1d_mfma_kernel",
                                   "B%d Cin%d M%d(x%dph) Tin%d cols%d k%d s%d d%d g%d W%d tile%dx%dx%d split%d%s%s", batch,
                                   g.cin_g * groups, g.cout_g * groups, g.phases, a.t_in, g.n_cols, g.k_phase, g.stride,
                                   g.dil, groups, a.width, BM, BN, CK, a.ksplit, a.item_major ? " im" : "",
                                   a.mask_src ? " dgrad" : ""),
                   flops, bytes);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  }
  PWG_CHECK_LAUNCH("conv1d_forward");
  if (a.ksplit > 1) {
    FinishArgs f;
    f.partial = a.partial;
    f.bias = a.bias;
    f.add1 = a.add1;
    f.add2 = a.add2;
    f.mask_src = a.mask_src;
    f.y = a.y;
    f.elems = f.slab_elems = a.slab_elems;
    f.nslabs = a.ksplit;
    f.y_cstride = a.y_cstride;
    f.c_out = g.cout_g * groups;
    f.post_act = a.post_act;
    f.post_slope = a.post_slope;
    f.out_mul = a.out_mul;
    f.out_div = a.out_div;
    f.mask_slope = a.mask_slope;
    long blocks = (f.elems + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const int extra = (a.add1 != nullptr) + (a.add2 != nullptr) + (a.mask_src != nullptr);
    ProfScope prof(stream, "splitk_finish_kernel", 0, 4.0 * f.elems * (a.ksplit + 1 + extra));
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((int)blocks), dim3(256), 0, stream, f);
    PWG_CHECK_LAUNCH("splitk_finish");
  }
  return PWG_OK;
}

// Tile configurations (id -> instantiation); tools/bench_conv.py sweeps them.  (The 2x4-accumulator
// tiles of round 1 needed scratch and never won a sweep: removed.)
struct TileCfg {
  int bm, bn, ck;
  int wm, wn;  // 32x32 accumulator tiles per wave
  int waves;   // waves per workgroup (4, or 8 for the 128x256 tiles: one workgroup per CU)
};
#define PWG_CONV_CFGS(X)  \
  X(0, 2, 2, 2, 2, 8)     \
  X(1, 2, 2, 2, 2, 16)    \
  X(2, 2, 2, 2, 2, 4)     \
  X(3, 2, 2, 1, 4, 8)     \
  X(4, 2, 2, 1, 4, 16)    \
  X(5, 1, 2, 1, 4, 8)     \
  X(6, 1, 2, 1, 4, 16)    \
  X(7, 1, 4, 1, 4, 8)     \
  X(8, 1, 4, 1, 4, 16)    \
  X(9, 2, 1, 2, 2, 8)     \
  X(10, 1, 1, 1, 4, 8)    \
  X(11, 2, 2, 1, 4, 4)    \
  X(12, 1, 2, 2, 2, 8)    \
  X(13, 1, 1, 2, 2, 8)    \
  X(14, 1, 1, 4, 1, 8)    \
  X(15, 1, 1, 2, 2, 16)   \
  X(16, 1, 1, 1, 4, 16)   \
  X(17, 1, 1, 2, 2, 4)    \
  X(18, 2, 2, 2, 4, 4)    \
  X(19, 2, 2, 2, 4, 8)
static const int kNumCfgs = 20;

static TileCfg cfg_info(int id) {
  switch (id) {
#define X(ID, WM, WN, WVM, WVN, CK) \
  case ID:                          \
    return TileCfg{32 * WM * WVM, 32 * WN * WVN, CK, WM, WN, WVM * WVN};
    PWG_CONV_CFGS(X)
#undef X
  }
  return TileCfg{0, 0, 0, 0, 0, 0};
}

static size_t cfg_lds(int id, const Geometry& g, int W, bool dma) {
  TileCfg c = cfg_info(id);
  const int rows = (W == 1) ? c.bn : ((c.bn - 1) / W + 2);
  const int xs_len = ((rows - 1) * g.stride + (g.k_phase - 1) * g.dil + 1) * W;
  int xs = dma ? round_up(xs_len, 64) : round_up(xs_len, 4);
  if (dma && g.stride == 1 && W == 1 && (g.k_phase - 1) * g.dil <= 64) xs = c.bn + 64;  // FAST row stride (upper bound)
  const size_t buf = ((size_t)c.ck * xs + (size_t)g.k_phase * c.ck * c.bm) * sizeof(float);
  const size_t scr = (size_t)c.waves * 32 * 36 * sizeof(float);
  return dma ? 2 * buf + scratch_bytes_needed(buf, scr, g.cin_g, c.ck, 1) : buf;
}

static int launch_cfg(int id, bool dma, const ConvArgs& a, const Geometry& g, int batch, int groups,
                      hipStream_t stream) {
  switch (id) {
#define X(ID, WM, WN, WVM, WVN, CK)                                                        \
  case ID:                                                                                 \
    return dma ? launch_conv<WM, WN, WVM, WVN, CK, true>(a, g, batch, groups, stream)      \
               : launch_conv<WM, WN, WVM, WVN, CK, false>(a, g, batch, groups, stream);
    PWG_CONV_CFGS(X)
#undef X
  }
  set_error("conv1d: unknown tile config %d", id);
  return PWG_ERR_UNSUPPORTED;
}

// Heuristic from the tools/bench_conv.py sweep on MI355X (HiFi-GAN V1 problem set): the kernel is
// latency- rather than reuse-bound, so few taps want SMALL tiles (more resident workgroups per
// CU) and many taps a short ci-chunk (CK=4) under a 128x128 / 64x256 tile.  Training shapes add
// launches with few columns per item (T = 9..256): every candidate is scored by
//   (chip fill: workgroups up to 2 per CU) x (useful fraction of its column tiles) x (measured
//   relative speed of the tile shape)
// and the best one that fits the LDS wins.
struct Cand {
  int id;
  float speed;
};
// (pwg_set_concurrency_hint; PWG_CONV_FILL_SCALE overrides it for experiments)
static float g_fill_scale = 1.0f;
static const float g_fill_scale_env = getenv("PWG_CONV_FILL_SCALE") ? (float)atof(getenv("PWG_CONV_FILL_SCALE")) : 0.f;
float concurrency_hint() { return g_fill_scale_env > 0.f ? g_fill_scale_env : g_fill_scale; }

// ksplit (optional out): number of reduction slices.  A workgroup walks its ci-chunks serially and a
// chunk costs at least one DMA + barrier round trip (~2.7 us measured) however little MFMA work it
// carries; when the chosen tile leaves the SIMDs that idle (few workgroups, short chunks: the
// 1024-channel discriminator layers with T = 9..32, C -> 1 output convs, deep generator layers at
// training lengths) the reduction is cut into 2/4/8 slices so that 2-8 workgroups interleave per
// CU.  Slices are added only while the per-SIMD MFMA time of a chunk stays below the round trip.
static int choose_cfg(const Geometry& g, int W, int batch, int groups, bool dma, int* ksplit = nullptr) {
  const int m = g.m_g;
  const int k = g.k_phase;
  // ids: 9 = 128x64x8, 0 = 128x128x8, 12 = 64x128x8, 13 = 64x64x8, 14 = 128x32x8, 2 = 128x128x4, 15 = 64x64x16,
  // 16 = 32x128x16, 10 = 32x128x8, 11 = 64x256x4
  // 18 / 19 = 128x256x4 / x8 with 8 waves (one workgroup per CU, one barrier domain, half the weight DMA per MFMA):
  // tools/probes/conv_wreg.hip (loop only, no epilogue) ranked them first at every k, but on the real kernel they
  // LOSE 5-10 % to 128x128x4 at C = 256 and 3-5 % to 64x256x4 at C = 128 (profiles/r03_conv_sweep.txt): with one
  // workgroup per CU nothing overlaps a workgroup's epilogue / first-chunk latency.  Kept as sweepable
  // configurations (pwg_conv1d_forward_cfg), never chosen.
  // 11 = 64x256x4: the row blocks re-read x, still 3-7 % faster than 128x128x4 at 128 rows and 3-4 % at 256 rows
  // for k >= 7 (tools/bench_cfgs.py: 103 / 108 vs 99 / 105 TFLOP/s at k = 7 / 11, C = 256): its weight chunk -- the
  // bulk of a chunk's DMA instructions -- is half as long.
  // Tried in round 3 and NOT kept (both measured on the real kernel, tools/bench_conv.py / bench_cfgs.py):
  //   * a 4-wave 128x256 tile with 2 x 4 accumulator tiles per wave: hipcc spills 171 VGPRs at the 256-register
  //     cap that two workgroups per CU need (18 - 60 TFLOP/s);
  //   * a fifth "loader" wave per workgroup that issues every LDS-DMA of the next chunk while the four compute
  //     waves only contract: one wave cannot issue a chunk's 26 - 30 DMA instructions and wait for them in the time
  //     the others need for 24 - 88 MFMAs each: 49 - 94 TFLOP/s against 86 - 119 (gpurun_out/r3/conv_loader.txt).
  static const Cand big_few[] = {{9, 0.95f}, {0, 0.85f}, {12, 0.85f}, {13, 0.75f}, {14, 0.6f}, {2, 0.8f},
                                 {15, 0.8f}, {16, 0.82f}, {11, 0.97f}};
  static const Cand big_many[] = {{2, 1.0f}, {9, 0.9f}, {12, 0.85f}, {13, 0.75f}, {14, 0.6f}, {15, 0.8f}, {16, 0.82f},
                                  {11, 1.06f}};
  // 17 = 64x64x4: the only 64-row tile whose double-buffered weight chunk fits the LDS at k = 41 (grouped
  // scale-discriminator layers, 64 channels per group, T = 9..128 columns per item)
  // (16 / 6 = 32x128x16 / 32x256x16: eligible only for k <= 3, see the CK = 16 rule below; 64 -> 64 k = 3 at
  // T = 25600: 61.7 us against 67.4 us, 48 -> 48 k = 1 at T = 4096: 59.4 against 64.4, profiles/r04_k1_sweep.txt)
  static const Cand mid_few[] = {{10, 1.0f}, {11, 0.8f}, {13, 0.8f}, {17, 0.7f}, {16, 1.08f}, {6, 1.05f}};
  static const Cand mid_many[] = {{11, 1.0f}, {10, 0.9f}, {13, 0.8f}, {17, 0.85f}};
  static const Cand small_any[] = {{10, 1.0f}};
  // Round 4: 96 rows (multi-band MelGAN's middle stage and its residual stacks, models/melgan.py:130-178) are three
  // 32-row blocks: every 64 / 128-row tile computes a quarter of its rows for nothing AND the planner had no 32-row
  // tile with a long chunk on its list.  Forced sweep at B64 x T2048 (tools/bench_dsplit.py k1,
  // profiles/r04_k1_sweep.txt): k = 1: 32x256x16 53.7 us against 101.2 us for the 64x256x4 the old list chose;
  // k = 3 (dilation 3): 32x128x16 103 us against 166 us.  (PWG_ROWS32=0 restores the round-3 lists.)
  static const Cand rows32[] = {{6, 1.0f}, {16, 0.98f}, {5, 0.94f}, {10, 0.86f}, {12, 0.66f}, {9, 0.66f}, {11, 0.6f}};
  static const bool rows32_on = !(getenv("PWG_ROWS32") && atoi(getenv("PWG_ROWS32")) == 0);
  const Cand* cand;
  int ncand;
  if (rows32_on && m > 64 && m <= 128 && m % 64 != 0 && m % 32 == 0 && k <= 4) {
    cand = rows32;
    ncand = 7;
  } else if (m > 64) {
    cand = k <= 4 ? big_few : big_many;
    ncand = (k <= 4 ? 9 : 8) - ((m > 128 && k <= 4) ? 1 : 0);  // (few taps: 64x256x4 only for m <= 128)
  } else if (m > 32) {
    cand = k <= 4 ? mid_few : mid_many;  // (k = 7 at C = 64: 64x256x4 103 vs 32x128x8 95 TFLOP/s, profiles/r02_conv_sweep.txt)
    ncand = (k <= 4 && rows32_on) ? 6 : 4;
  } else {
    cand = small_any;
    ncand = 1;
  }
  // slices the latency rule allows for a tile: more workgroups per CU only while the per-SIMD MFMA
  // time of a chunk stays below the DMA + barrier round trip (6200 cycles ~ 2.7 us)
  auto max_split = [&](const TileCfg& c, long blocks) {
    if (!ksplit || !dma) return 1;
    const long per_cu = (blocks + 255) / 256;                       // = waves per SIMD (4-wave workgroups)
    const double mfma_chunk = 64.0 * c.wm * c.wn * k * (c.ck / 2);  // MFMA cycles per chunk per wave
    const int nchunks = ceil_div(g.cin_g, c.ck);
    int split = 1;
    // ... and only while the launch has fewer than 2 workgroups per CU: a grid that already fills the chip
    // loses by splitting (measured, tools/sweep_shapes.py: C=128 k=3 T=2048 at 512 workgroups 72 -> 50 TFLOP/s,
    // C=64 k=3 T=4096 at 1024 workgroups 58 -> 36)
    while (split < 8 && nchunks >= 8 * split && blocks * split < 512 && mfma_chunk * per_cu * (2 * split) <= 6200.0)
      split *= 2;
    // Round 4: LONG reductions over few columns -- the period discriminators' 512 / 1024-channel layers, 9..110
    // columns per item, K = Cin * k = 2560..5120 -- leave the big tiles (the only ones whose chunk hides its own DMA
    // round trip) with 128 workgroups on 256 CUs.  Slicing the reduction there is not about latency but about filling
    // the chip with efficient tiles: forced sweep (tools/bench_dsplit.py, profiles/r04_dsplit_sweep.txt) 128x128x4 split
    // 4 = 226 us against 274 us for the 32x128x16 tile the fill score used to pick (1024 -> 1024, k = 5), 133 against
    // 165 us for 512 -> 1024 stride 3.  Each slice keeps >= 32 chunks, so the slab traffic (one y-sized write + read
    // per slice) stays below 2 % of the launch's operand traffic.  (PWG_SPLIT_FILL=0 restores the round-3 rule.)
    static const bool split_fill = !(getenv("PWG_SPLIT_FILL") && atoi(getenv("PWG_SPLIT_FILL")) == 0);
    if (split_fill && c.wm * c.wn >= 2) {
      const long full = (long)(512.f * concurrency_hint());
      while (split < 4 && nchunks >= 32 * (2 * split) && blocks * split < full) split *= 2;
    }
    // Round 6: SINGLE-UTTERANCE launches (batch 1, width 1: bin/decode.py's regime) that leave half the chip (or more)
    // without a workgroup are cut until the slices fill it, whatever a chunk carries -- the latency rule above never
    // splits heavy chunks.  Measured on the batch-folded training layers first (tools/bench_dsplit.py fold,
    // profiles/r06_fold_split_sweep.txt: grouped k = 41 layers of 64 .. 128 workgroups 86 -> 32 .. 54 us stand-alone),
    // but NOT applied to them: inside the captured training step those launches share the chip with the other
    // sub-discriminators' branches, the idle CUs are not idle, and the extra slabs cost more than the shorter launch
    // gains (C5 44.04 ms with the rule against 43.73 ms without, C3 within the spread: profiles/r06_train_ab.txt); on
    // multi-item launches it made the planner trade tiles that fill the chip for split big tiles (C3: +115 finish
    // launches, +2.0 ms of splitk_finish per step) and changes summation orders the full-shape training bars are
    // calibrated on.  (PWG_SPLIT_UNDERFILL=0 restores the round-5 rule.)
    static const bool split_underfill = !(getenv("PWG_SPLIT_UNDERFILL") && atoi(getenv("PWG_SPLIT_UNDERFILL")) == 0);
    if (split_underfill && batch == 1 && W == 1) {
      const long full = (long)((k >= 32 ? 256.f : 512.f) * concurrency_hint());
      while (split < 16 && nchunks >= 2 * split && blocks * split * 2 <= full) split *= 2;
    }
    return split;
  };
  int best = -1, best_split = 1;
  float best_score = -1.f;
  for (int pass = 0; pass < 2 && best < 0; ++pass) {
    for (int i = 0; i < ncand; ++i) {
      const TileCfg c = cfg_info(cand[i].id);
      // first try to keep >= 2 workgroups per CU (the 8-wave tiles are one workgroup per CU by design)
      const size_t cap = (pass == 0 && c.waves < 8) ? 80 * 1024 : 160 * 1024;
      if (cfg_lds(cand[i].id, g, W, dma) > cap) continue;
      if (c.waves == 8 && (!dma || g.stride != 1 || W != 1 || (k - 1) * g.dil > 64)) continue;  // FAST geometry only
      // 16-channel chunks: long reductions -- or few taps (k <= 3: a 4 / 8-channel chunk carries 2 - 12 MFMAs per
      // wave against a 2.7 us DMA round trip; measured at 96 channels, see rows32 above)
      if (c.ck == 16 && (!dma || (g.cin_g < 256 && !(rows32_on && k <= 3 && g.cin_g >= 32)))) continue;
      const long ntiles = ceil_div(g.n_cols, c.bn);
      const long blocks = ntiles * ceil_div(m, c.bm) * groups * batch;
      const int split = max_split(c, blocks);
      // (very long filters -- k = 41 grouped layers -- carry >= 5000 MFMA cycles per chunk: one workgroup
      // per CU already hides the DMA round trip, so 256 workgroups count as a full chip there)
      // g_fill_scale: concurrency hint (pwg_conv1d_set_concurrency): inside a captured step the launches of the
      // parallel sub-discriminator branches share the chip, so one launch need not fill it alone
      const float full = ((k >= 32 || c.waves == 8) ? 256.f : 512.f) * concurrency_hint();
      const float fill = blocks * split >= full ? 1.f : (float)(blocks * split) / full;
      const float useful = (float)g.n_cols / (float)(ntiles * c.bn) * (float)m / (float)(ceil_div(m, c.bm) * c.bm);
      const float score = fill * useful * cand[i].speed * (split > 1 ? 0.9f : 1.f);
      if (score > best_score) {
        best_score = score;
        best = cand[i].id;
        best_split = split;
      }
    }
  }
  if (best < 0) {
    best = 10;
    best_split = 1;
  }
  if (ksplit) *ksplit = best_split;
  return best;
}

static int fill_args(const pwg_conv1d_desc* d, const Geometry& g, const float* x, const float* w_packed,
                     const float* bias, const float* add1, const float* add2, float* y, ConvArgs* out) {
  PWG_REQUIRE(x && w_packed && y, PWG_ERR_NULL, "conv1d_forward: NULL pointer");
  PWG_REQUIRE(d->pre_act == PWG_ACT_NONE || d->pre_act == PWG_ACT_LEAKY_RELU || d->pre_act == PWG_ACT_RELU,
              PWG_ERR_UNSUPPORTED, "conv1d: pre_act %d unsupported", d->pre_act);
  ConvArgs a;
  a.x = x;
  a.wp = w_packed;
  a.bias = bias;
  a.add1 = add1;
  a.add2 = add2;
  a.y = y;
  a.cin_g = g.cin_g;
  a.cin_pad = g.cin_pad;
  a.cout_g = g.cout_g;
  a.m_g = g.m_g;
  a.m_pad = g.m_pad;
  a.t_in = d->t_in;
  a.t_out = d->t_out;
  a.width = d->width;
  a.k = g.k_phase;
  a.stride = g.stride;
  a.dil = g.dil;
  a.pad = g.pad;
  a.n_cols = g.n_cols;
  a.out_stride = g.out_stride;
  a.out_off = g.out_off;
  a.x_cstride = d->t_in * d->width;
  a.y_cstride = d->t_out * d->width;
  a.x_bstride = (long)d->c_in * a.x_cstride;
  a.y_bstride = (long)d->c_out * a.y_cstride;
  a.xs_stride = 0;
  a.pad_mode = d->pad_mode;
  a.pre_act = d->pre_act;
  a.post_act = d->post_act;
  a.pre_slope = d->pre_slope;
  a.post_slope = d->post_slope;
  a.out_mul = d->out_mul;
  a.out_div = d->out_div;
  a.mask_src = nullptr;
  a.mask_slope = 0.f;
  static const int dbg = getenv("PWG_DBG") ? atoi(getenv("PWG_DBG")) : 0;
  a.dbg = dbg;
  a.ksplit = 1;
  a.epi_vec = 0;
  a.item_major = 0;
  a.partial = nullptr;
  a.slab_elems = (long)d->batch * d->c_out * d->t_out * d->width;
  *out = a;
  return PWG_OK;
}

}  // namespace pwg

using namespace pwg;

extern "C" size_t pwg_conv1d_packed_weight_floats(const pwg_conv1d_desc* d) {
  Geometry g;  // (the packed layout does not depend on width / dilation: no flattening needed)
  if (make_geometry(d, &g) != PWG_OK) return 0;
  return (size_t)d->groups * g.k_phase * g.cin_pad * g.m_pad;
}

extern "C" int pwg_conv1d_pack_weight(const pwg_conv1d_desc* d, const float* w, const float* scale,
                                      float* w_packed, void* stream) {
  Geometry g;
  int rc = make_geometry(d, &g);
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(w && w_packed, PWG_ERR_NULL, "pack_weight: NULL pointer");
  PackArgs a;
  a.w = w;
  a.scale = scale;
  a.wp = w_packed;
  a.groups = d->groups;
  a.k_phase = g.k_phase;
  a.cin_g = g.cin_g;
  a.cin_pad = g.cin_pad;
  a.cout_g = g.cout_g;
  a.m_g = g.m_g;
  a.m_pad = g.m_pad;
  a.kernel = d->kernel;
  a.stride = d->stride;
  a.transposed = d->transposed;
  const long total = (long)a.groups * a.k_phase * a.cin_pad * a.m_pad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  PWG_CHECK_LAUNCH("pack_weight");
  return PWG_OK;
}

// which launches of pwg_conv1d_forward leave the MFMA kernel (flattened descriptors): the streaming VALU kernels for a
// single input channel (conv1d_small_cin_kernel) and for <= 4 output channels over a long sequence
// (conv1d_small_cout_kernel)
static bool small_cin_applicable(const pwg_conv1d_desc* d, bool has_addends) {
  static const bool small_cin_on = !(getenv("PWG_SMALL_CIN") && atoi(getenv("PWG_SMALL_CIN")) == 0);
  return small_cin_on && !d->transposed && d->groups == 1 && d->c_in == 1 && d->width == 1 && d->stride == 1 &&
         d->pad_mode == PWG_PAD_ZERO && d->kernel <= SI_MAXK && d->c_out >= 8 && d->c_out <= 256 && !has_addends &&
         d->out_div == 1.0f && d->t_out >= 2048 &&
         (d->pre_act == PWG_ACT_NONE || d->pre_act == PWG_ACT_LEAKY_RELU || d->pre_act == PWG_ACT_RELU);
}
static bool small_cout_applicable(const pwg_conv1d_desc* d, bool has_addends) {
  return !d->transposed && d->groups == 1 && d->width == 1 && d->stride == 1 && d->pad_mode == PWG_PAD_ZERO &&
         d->c_out <= 4 && d->c_out * d->c_in * d->kernel <= SC_MAXW && !has_addends && d->out_div == 1.0f &&
         d->t_out >= 4096 && (d->kernel - 1) * d->dilation <= 1024 &&
         (d->pre_act == PWG_ACT_NONE || d->pre_act == PWG_ACT_LEAKY_RELU || d->pre_act == PWG_ACT_RELU);
}

// tile configuration, staging path and split-K factor of a (flattened) forward-form descriptor
struct ConvPlan {
  int id, ksplit;
  bool dma;
};
static ConvPlan plan_conv(const pwg_conv1d_desc* d, const Geometry& g) {
  ConvPlan p;
  p.dma = d->pad_mode == PWG_PAD_ZERO;  // reflect/replicate need index remapping: register path
  p.ksplit = 1;
  p.id = choose_cfg(g, d->width, d->batch, d->groups, p.dma, &p.ksplit);
  // tuning override (tools/bench_dshapes.py): PWG_FORCE_CFG=<id>[,<ksplit>], read per call
  if (const char* f = getenv("PWG_FORCE_CFG")) {
    int id = -1, ks = 1;
    if (sscanf(f, "%d,%d", &id, &ks) >= 1 && id >= 0 && id < kNumCfgs && p.dma && cfg_lds(id, g, d->width, true) <= 160 * 1024) {
      p.id = id;
      p.ksplit = ks < 1 ? 1 : (ks > 16 ? 16 : ks);
      if (ceil_div(g.cin_g, cfg_info(id).ck) < p.ksplit) p.ksplit = 1;
    }
  }
  if (p.dma && cfg_lds(p.id, g, d->width, true) > 160 * 1024) {  // very long filters (PQMF k=63): single buffer
    p.dma = false;
    p.ksplit = 1;
    p.id = choose_cfg(g, d->width, d->batch, d->groups, false);
  }
  return p;
}

static int run_conv(const pwg_conv1d_desc* d, const Geometry& g, ConvArgs& a, float* workspace, size_t ws_floats,
                    hipStream_t stream) {
  ConvPlan p = plan_conv(d, g);
  if (p.ksplit > 1) {
    const size_t need = (size_t)p.ksplit * a.slab_elems;
    if (workspace && ws_floats >= need) {
      a.ksplit = p.ksplit;
      a.partial = workspace;
    } else {
      // no (or too small a) workspace: run unsplit with the best unsplit tile
      p.id = choose_cfg(g, d->width, d->batch, d->groups, p.dma);
    }
  }
  static const bool trace = getenv("PWG_TRACE_CFG") != nullptr;
  if (trace)
    fprintf(stderr, "[pwg] conv B=%d Cin=%d Cout=%d Tin=%d Tout=%d k=%d s=%d d=%d g=%d tr=%d -> cfg %d split %d dma %d\n",
            d->batch, d->c_in, d->c_out, d->t_in, d->t_out, d->kernel, d->stride, d->dilation, d->groups,
            d->transposed, p.id, a.ksplit, (int)p.dma);
  return launch_cfg(p.id, p.dma, a, g, d->batch, d->groups, stream);
}

extern "C" size_t pwg_conv1d_forward_workspace_floats(const pwg_conv1d_desc* d_in) {
  if (!d_in) return 0;
  const pwg_conv1d_desc flat = flatten_width(*d_in);
  Geometry g;
  if (make_geometry(&flat, &g) != PWG_OK) return 0;
  const ConvPlan p = plan_conv(&flat, g);
  return p.ksplit > 1 ? (size_t)p.ksplit * flat.batch * flat.c_out * flat.t_out * flat.width : 0;
}

// Few output channels over a long sequence: the streaming VALU kernels (conv1d_small_cout_stream_kernel / conv1d_small_cout_kernel).
// `d`: a plain (non-transposed) flattened descriptor for which small_cout_applicable holds; `g` its geometry.
static int run_small_cout(const pwg_conv1d_desc* d, const Geometry& g, const float* x, const float* w_packed, const float* bias,
                          float* y, void* stream) {
  // few output channels over a long sequence: streaming VALU kernel (see conv1d_small_cout_kernel)
  {
    // round 6: the LDS-free stream for "same"-padded dilation-1 layers with 16-B aligned rows (PWG_SMALL_COUT_STREAM=0: off)
    static const bool stream_on = !(getenv("PWG_SMALL_COUT_STREAM") && atoi(getenv("PWG_SMALL_COUT_STREAM")) == 0);
    const int kk = d->kernel;
    void (*sk)(const float*, const float*, const float*, float*, int, int, int, int, int, int, float, int, float, float) = nullptr;
    if (stream_on && d->dilation == 1 && (kk & 1) && d->pad_left == (kk - 1) / 2 && d->t_out == d->t_in &&
        (d->t_in & 3) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (d->c_out == 1 || d->c_out == 4)) {
#define PWG_SCS(KV) (d->c_out == 1 ? conv1d_small_cout_stream_kernel<KV, (KV - 1) / 2, 1> : conv1d_small_cout_stream_kernel<KV, (KV - 1) / 2, 4>)
      if (kk == 1) sk = PWG_SCS(1);
      else if (kk == 3) sk = PWG_SCS(3);
      else if (kk == 5) sk = PWG_SCS(5);
      else if (kk == 7) sk = PWG_SCS(7);
#undef PWG_SCS
    }
    if (sk) {
      const double out_elems = (double)d->batch * d->c_out * d->t_out;
      ProfScope prof((hipStream_t)stream, "conv1d_small_cout_stream_kernel", 2.0 * out_elems * d->c_in * d->kernel,
                     4.0 * ((double)d->batch * d->c_in * d->t_in + out_elems));
      hipLaunchKernelGGL(sk, dim3(ceil_div(d->t_out, 1024), d->batch), dim3(256),
                         (size_t)d->c_out * d->c_in * d->kernel * sizeof(float), (hipStream_t)stream, x, w_packed, bias, y,
                         d->c_in, g.cin_pad, g.m_pad, d->t_in, d->t_out, d->pre_act, d->pre_slope, d->post_act,
                         d->post_slope, d->out_mul);
      PWG_CHECK_LAUNCH("conv1d_small_cout_stream");
      return PWG_OK;
    }
  }
  int opt = 4;
  while (opt > 1 && (long)ceil_div(d->t_out, 256 * opt) * d->batch < 1024) opt >>= 1;
  const int tile = 256 * opt;
  const size_t lds = ((size_t)d->c_out * d->c_in * d->kernel + tile + (d->kernel - 1) * d->dilation) * sizeof(float);
  const double out_elems = (double)d->batch * d->c_out * d->t_out;
  ProfScope prof((hipStream_t)stream, "conv1d_small_cout_kernel", 2.0 * out_elems * d->c_in * d->kernel,
                 4.0 * ((double)d->batch * d->c_in * d->t_in + out_elems));
  void (*sck)(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int, float, int,
              float, float) = opt == 4 ? conv1d_small_cout_kernel<4> : (opt == 2 ? conv1d_small_cout_kernel<2> : conv1d_small_cout_kernel<1>);
  hipLaunchKernelGGL(sck, dim3(ceil_div(d->t_out, tile), d->batch), dim3(256), lds,
                     (hipStream_t)stream, x, w_packed, bias, y, d->c_in, g.cin_pad, g.m_pad, d->c_out, d->t_in, d->t_out,
                     d->kernel, d->dilation, d->pad_left, d->pre_act, d->pre_slope, d->post_act, d->post_slope,
                     d->out_mul);
  PWG_CHECK_LAUNCH("conv1d_small_cout");
  return PWG_OK;
}

extern "C" int pwg_conv1d_forward(const pwg_conv1d_desc* d_in, const float* x, const float* w_packed,
                                  const float* bias, const float* add1, const float* add2, float* y,
                                  float* workspace, size_t workspace_floats, void* stream) {
  PWG_REQUIRE(d_in != nullptr, PWG_ERR_NULL, "conv1d: NULL descriptor");
  const pwg_conv1d_desc flat = flatten_width(*d_in);
  const pwg_conv1d_desc* d = &flat;
  Geometry g;
  int rc = make_geometry(d, &g);
  if (rc != PWG_OK) return rc;
  if (gconv_forward_applicable(d, add1, add2)) return gconv_forward(d, x, w_packed, bias, y, (hipStream_t)stream);
  if (small_cin_applicable(d, add1 || add2)) {
    ConvArgs chk;
    rc = fill_args(d, g, x, w_packed, bias, add1, add2, y, &chk);
    if (rc != PWG_OK) return rc;
    // channel groups: at least ~1024 workgroups per launch (4 per CU), never fewer than 8 channels per group
    const int pairs = ceil_div(d->t_out, SI_TILE) * d->batch;
    int ngroups = ceil_div(1024, pairs);
    if (ngroups > d->c_out / 8) ngroups = d->c_out / 8;
    if (ngroups < 1) ngroups = 1;
    const int cg = ceil_div(d->c_out, ngroups);
    ngroups = ceil_div(d->c_out, cg);
    const size_t lds = ((size_t)cg * (d->kernel + 1) + SI_TILE + (d->kernel - 1) * d->dilation) * sizeof(float);
    PWG_REQUIRE(lds <= 64 * 1024, PWG_ERR_UNSUPPORTED, "conv1d (single input channel): %zu B of LDS", lds);
    const double out_elems = (double)d->batch * d->c_out * d->t_out;
    const bool vec = d->t_out % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15u) == 0;
    ProfScope prof((hipStream_t)stream, "conv1d_small_cin_kernel", 2.0 * out_elems * d->kernel,
                   4.0 * ((double)d->batch * d->t_in + out_elems));
    hipLaunchKernelGGL(vec ? conv1d_small_cin_kernel<true> : conv1d_small_cin_kernel<false>,
                       dim3(ceil_div(d->t_out, SI_TILE), d->batch, ngroups), dim3(256), lds, (hipStream_t)stream, x, w_packed,
                       bias, y, g.cin_pad, g.m_pad, d->c_out, cg, d->t_in, d->t_out, d->kernel, d->dilation, d->pad_left,
                       d->pre_act, d->pre_slope, d->post_act, d->post_slope, d->out_mul);
    PWG_CHECK_LAUNCH("conv1d_small_cin");
    return PWG_OK;
  }
  if (small_cout_applicable(d, add1 || add2)) {
    // (the same argument checks as the MFMA path: fill_args validates pointers and the pre-activation)
    ConvArgs chk;
    rc = fill_args(d, g, x, w_packed, bias, add1, add2, y, &chk);
    if (rc != PWG_OK) return rc;
    return run_small_cout(d, g, x, w_packed, bias, y, stream);
  }
  ConvArgs a;
  rc = fill_args(d, g, x, w_packed, bias, add1, add2, y, &a);
  if (rc != PWG_OK) return rc;
  return run_conv(d, g, a, workspace, workspace_floats, (hipStream_t)stream);
}

// The data gradient of a convolution is the transposed convolution with the same torch-layout
// weight (and vice versa), so both directions reuse the forward kernels through the dual descriptor.
static void dual_desc(const pwg_conv1d_desc* d, pwg_conv1d_desc* o) {
  *o = *d;
  o->c_in = d->c_out;
  o->c_out = d->c_in;
  o->t_in = d->t_out;
  o->t_out = d->t_in;
  o->transposed = d->transposed ? 0 : 1;
  o->pad_mode = PWG_PAD_ZERO;
  o->pre_act = PWG_ACT_NONE;
  o->post_act = PWG_ACT_NONE;
  o->pre_slope = o->post_slope = 0.f;
  o->out_mul = o->out_div = 1.f;
}

extern "C" size_t pwg_conv1d_packed_weight_bwd_floats(const pwg_conv1d_desc* d) {
  if (!d) return 0;
  pwg_conv1d_desc dd;
  dual_desc(d, &dd);
  return pwg_conv1d_packed_weight_floats(&dd);
}

extern "C" int pwg_conv1d_pack_weight_bwd(const pwg_conv1d_desc* d, const float* w, const float* scale,
                                          float* w_packed_bwd, void* stream) {
  PWG_REQUIRE(d, PWG_ERR_NULL, "pack_weight_bwd: NULL descriptor");
  pwg_conv1d_desc dd;
  dual_desc(d, &dd);
  return pwg_conv1d_pack_weight(&dd, w, scale, w_packed_bwd, stream);
}

extern "C" size_t pwg_conv1d_backward_data_workspace_floats(const pwg_conv1d_desc* d) {
  if (!d) return 0;
  pwg_conv1d_desc dd;
  dual_desc(d, &dd);
  return pwg_conv1d_forward_workspace_floats(&dd);
}

extern "C" int pwg_conv1d_backward_data(const pwg_conv1d_desc* d, const float* dy, const float* w_packed_bwd,
                                        const float* x, const float* accum, float* dx, float* workspace,
                                        size_t workspace_floats, void* stream) {
  PWG_REQUIRE(d, PWG_ERR_NULL, "conv1d_backward_data: NULL descriptor");
  PWG_REQUIRE(d->pad_mode == PWG_PAD_ZERO, PWG_ERR_UNSUPPORTED,
              "conv1d_backward_data: only zero padding (pad reflect/replicate inputs explicitly)");
  PWG_REQUIRE(d->pre_act == PWG_ACT_NONE || x != nullptr, PWG_ERR_NULL,
              "conv1d_backward_data: the forward input is needed for the pre-activation derivative");
  if (gconv_dgrad_applicable(d)) return gconv_backward_data(d, dy, w_packed_bwd, x, accum, dx, (hipStream_t)stream);
  pwg_conv1d_desc dd;
  dual_desc(d, &dd);
  dd = flatten_width(dd);
  Geometry g;
  int rc = make_geometry(&dd, &g);
  if (rc != PWG_OK) return rc;
  // Round 6: the data gradient of a stride-1 layer with one (<= 4) input channel over a long sequence -- every discriminator's
  // first layer, when the discriminator's input needs a gradient (generator phase) -- is a plain few-OUTPUT-channel convolution
  // over the flipped taps (make_geometry: that is how the backward image is packed): on the MFMA tile one of 32 rows carries data
  // (2.5 TFLOP/s, 0.2 ms for MelGAN's 16 -> 1 k = 15 layer at B64 x 16384, 0.15 ms for HiFi-GAN's 128 -> 1); the streaming
  // kernels read dy once.  PWG_SMALL_COUT_DGRAD=0: the MFMA path.
  static const bool dgrad_stream = !(getenv("PWG_SMALL_COUT_DGRAD") && atoi(getenv("PWG_SMALL_COUT_DGRAD")) == 0);
  if (dgrad_stream && dd.stride == 1 && g.phases == 1 && g.out_off == 0 && accum == nullptr && d->pre_act == PWG_ACT_NONE) {
    pwg_conv1d_desc pd = dd;
    pd.transposed = 0;
    pd.pad_left = g.pad;
    // (the LDS kernel walks the input channels one by one, two barriers each: 0.29 ms for HiFi-GAN's 128 -> 1 k = 15 against 0.13 ms
    // on the MFMA tile -- only few channels take it; the LDS-free stream (k <= 7, "same" padding) has no such walk)
    const bool stream_form = pd.dilation == 1 && (pd.kernel & 1) && pd.kernel <= 7 && pd.pad_left == (pd.kernel - 1) / 2 &&
                             pd.t_out == pd.t_in && (pd.t_in & 3) == 0 && ((((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0 &&
                             (pd.c_out == 1 || pd.c_out == 4);
    if (small_cout_applicable(&pd, false) && (pd.c_in <= 32 || stream_form)) {
      PWG_REQUIRE(dy && w_packed_bwd && dx, PWG_ERR_NULL, "conv1d_backward_data: NULL pointer");
      Geometry pg;
      rc = make_geometry(&pd, &pg);
      if (rc != PWG_OK) return rc;
      PWG_REQUIRE(pg.cin_pad == g.cin_pad && pg.m_pad == g.m_pad && pg.k_phase == g.k_phase, PWG_ERR_UNSUPPORTED,
                  "conv1d_backward_data: plain form of the dual descriptor has another image layout");
      return run_small_cout(&pd, pg, dy, w_packed_bwd, nullptr, dx, stream);
    }
  }
  ConvArgs a;
  rc = fill_args(&dd, g, dy, w_packed_bwd, nullptr, accum, nullptr, dx, &a);
  if (rc != PWG_OK) return rc;
  if (d->pre_act != PWG_ACT_NONE) {
    a.mask_src = x;
    a.mask_slope = d->pre_act == PWG_ACT_LEAKY_RELU ? d->pre_slope : 0.f;
  }
  return run_conv(&dd, g, a, workspace, workspace_floats, (hipStream_t)stream);
}

extern "C" int pwg_conv1d_num_tile_configs(void) { return kNumCfgs; }

extern "C" int pwg_conv1d_plan(const pwg_conv1d_desc* d_in, int32_t has_addends, int32_t* out) {
  PWG_REQUIRE(d_in != nullptr && out != nullptr, PWG_ERR_NULL, "conv1d_plan: NULL pointer");
  const pwg_conv1d_desc flat = flatten_width(*d_in);
  const pwg_conv1d_desc* d = &flat;
  Geometry g;
  int rc = make_geometry(d, &g);
  if (rc != PWG_OK) return rc;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  const float* addend = has_addends ? reinterpret_cast<const float*>(out) : nullptr;  // (only tested against NULL)
  if (gconv_forward_applicable(d, addend, nullptr)) {
    out[0] = 1;
    return PWG_OK;
  }
  if (small_cin_applicable(d, has_addends != 0)) {
    out[0] = 2;
    return PWG_OK;
  }
  if (small_cout_applicable(d, has_addends != 0)) {
    out[0] = 3;
    return PWG_OK;
  }
  const ConvPlan p = plan_conv(d, g);
  const TileCfg c = cfg_info(p.id);
  const int gx = ceil_div(g.n_cols, c.bn), mtiles = ceil_div(g.m_g, c.bm);
  out[1] = p.id;
  out[2] = p.dma ? p.ksplit : 1;
  out[3] = p.dma ? 1 : 0;
  out[4] = p.dma ? choose_tile_order(g, d->width, d->t_in, c.bm, c.bn, gx, mtiles, d->groups, d->batch, out[2]) : 0;
  out[5] = gx;
  out[6] = mtiles * d->groups;
  out[7] = d->batch * out[2];
  return PWG_OK;
}

extern "C" int pwg_debug_conv_tile_of_workgroup(int32_t gx, int32_t gy, int32_t gz, int32_t mtiles, int32_t ksplit,
                                                int32_t item_major, int32_t workgroup, int32_t* out) {
  PWG_REQUIRE(out != nullptr, PWG_ERR_NULL, "tile_of_workgroup: NULL pointer");
  PWG_REQUIRE(gx > 0 && gy > 0 && gz > 0 && mtiles > 0 && ksplit > 0 && gy % mtiles == 0 && gz % ksplit == 0 &&
                  workgroup >= 0 && (long)workgroup < (long)gx * gy * gz,
              PWG_ERR_BAD_SHAPE, "tile_of_workgroup: inconsistent grid (%d, %d, %d) mtiles %d ksplit %d id %d", gx, gy,
              gz, mtiles, ksplit, workgroup);
  int bx, by, bz;
  tile_of_workgroup((unsigned)workgroup, (unsigned)gx, (unsigned)gy, (unsigned)gz, (unsigned)mtiles, (unsigned)ksplit,
                    item_major, bx, by, bz);
  out[0] = bx;
  out[1] = by;
  out[2] = bz;
  return PWG_OK;
}

extern "C" float pwg_set_concurrency_hint(float fill_scale) {
  const float was = g_fill_scale;
  if (fill_scale > 0.f && fill_scale <= 1.f) g_fill_scale = fill_scale;
  return was;
}

extern "C" int pwg_conv1d_forward_cfg(const pwg_conv1d_desc* d_in, const float* x, const float* w_packed,
                                      const float* bias, const float* add1, const float* add2, float* y,
                                      int32_t tile_config, int32_t use_dma, void* stream) {
  PWG_REQUIRE(d_in != nullptr, PWG_ERR_NULL, "conv1d: NULL descriptor");
  const pwg_conv1d_desc flat = flatten_width(*d_in);
  const pwg_conv1d_desc* d = &flat;
  Geometry g;
  int rc = make_geometry(d, &g);
  if (rc != PWG_OK) return rc;
  ConvArgs a;
  rc = fill_args(d, g, x, w_packed, bias, add1, add2, y, &a);
  if (rc != PWG_OK) return rc;
  PWG_REQUIRE(tile_config >= 0 && tile_config < kNumCfgs, PWG_ERR_UNSUPPORTED, "conv1d: tile config %d out of range",
              tile_config);
  PWG_REQUIRE(!(use_dma && d->pad_mode != PWG_PAD_ZERO), PWG_ERR_UNSUPPORTED,
              "conv1d: the DMA path implements zero padding only");
  return launch_cfg(tile_config, use_dma != 0, a, g, d->batch, d->groups, (hipStream_t)stream);
}

extern "C" int pwg_weight_norm_scale(const float* v, const float* g, float* scale, int32_t n0,
                                     int32_t inner, void* stream) {
  PWG_REQUIRE(v && g && scale, PWG_ERR_NULL, "weight_norm_scale: NULL pointer");
  PWG_REQUIRE(n0 > 0 && inner > 0, PWG_ERR_BAD_SHAPE, "weight_norm_scale: bad shape (%d,%d)", n0, inner);
  hipLaunchKernelGGL(weight_norm_scale_kernel, dim3(n0), dim3(256), 0, (hipStream_t)stream, v, g, scale, inner);
  PWG_CHECK_LAUNCH("weight_norm_scale");
  return PWG_OK;
}

extern "C" int pwg_scale_rows(const float* v, const float* scale, float* w, int32_t n0, int32_t inner,
                              void* stream) {
  PWG_REQUIRE(v && scale && w, PWG_ERR_NULL, "scale_rows: NULL pointer");
  PWG_REQUIRE(n0 > 0 && inner > 0, PWG_ERR_BAD_SHAPE, "scale_rows: bad shape (%d,%d)", n0, inner);
  const long total = (long)n0 * inner;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, scale, w, total, inner);
  PWG_CHECK_LAUNCH("scale_rows");
  return PWG_OK;
}


// ---------------------------------------------------------------------------
// Weight bank: host side.  The table is built ONCE per set of parameter / image addresses (host memory, uploaded by
// the caller outside any stream capture) and then only referenced: a prepare call is two launches whose arguments
// never change, so it can be captured into a hipGraph and replayed.
// Table layout (bytes): [BankRows x n_rows_tab][BankImage x n_images], images ordered forward images first.
// info[0] = n_rows_tab, info[1] = total scale rows, info[2] = n forward images, info[3] = pack workgroups of the
// forward images, info[4] = n images, info[5] = pack workgroups of all images, info[6] = byte offset of the images.
// ---------------------------------------------------------------------------
static void fill_pack_args(const pwg_conv1d_desc* d, const Geometry& g, const float* w, const float* scale, float* wp,
                           PackArgs* a) {
  a->w = w;
  a->scale = scale;
  a->wp = wp;
  a->groups = d->groups;
  a->k_phase = g.k_phase;
  a->cin_g = g.cin_g;
  a->cin_pad = g.cin_pad;
  a->cout_g = g.cout_g;
  a->m_g = g.m_g;
  a->m_pad = g.m_pad;
  a->kernel = d->kernel;
  a->stride = d->stride;
  a->transposed = d->transposed;
}

extern "C" size_t pwg_weight_bank_table_bytes(int32_t n_items) {
  if (n_items <= 0) return 0;
  return (size_t)n_items * (sizeof(BankRows) + 2 * sizeof(BankImage));
}

extern "C" int pwg_weight_bank_build(const pwg_bank_item* items, int32_t n_items, void* table_host, size_t table_bytes,
                                     int32_t* info) {
  PWG_REQUIRE(items && table_host && info && n_items > 0, PWG_ERR_NULL, "weight_bank_build: NULL pointer / no items");
  PWG_REQUIRE(table_bytes >= pwg_weight_bank_table_bytes(n_items), PWG_ERR_WORKSPACE, "weight_bank_build: table too small");
  BankRows* rows = reinterpret_cast<BankRows*>(table_host);
  int n_rows_tab = 0, row0 = 0;
  for (int i = 0; i < n_items; ++i) {
    const pwg_bank_item& it = items[i];
    PWG_REQUIRE(it.w != nullptr, PWG_ERR_NULL, "weight_bank_build: item %d has no weight", i);
    if (it.g == nullptr) continue;
    PWG_REQUIRE(it.scale != nullptr, PWG_ERR_NULL, "weight_bank_build: item %d: weight-norm layer without a scale buffer", i);
    const pwg_conv1d_desc& d = it.desc;
    PWG_REQUIRE(d.groups > 0 && d.c_in % d.groups == 0 && d.c_out % d.groups == 0 && d.kernel > 0, PWG_ERR_BAD_SHAPE,
                "weight_bank_build: item %d: bad descriptor", i);
    // torch layouts: Conv (c_out, c_in/g, k): dim 0 = c_out; ConvTranspose (c_in, c_out/g, k): dim 0 = c_in
    const int n0 = d.transposed ? d.c_in : d.c_out;
    const int inner = (d.transposed ? d.c_out / d.groups : d.c_in / d.groups) * d.kernel;
    rows[n_rows_tab] = BankRows{it.w, it.g, it.scale, inner, row0};
    row0 += n0;
    ++n_rows_tab;
  }
  BankImage* imgs = reinterpret_cast<BankImage*>(rows + n_rows_tab);
  int n_img = 0, block0 = 0, n_fwd = 0, blocks_fwd = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < n_items; ++i) {
      const pwg_bank_item& it = items[i];
      float* out = pass == 0 ? it.fwd : it.bwd;
      if (!out) continue;
      pwg_conv1d_desc d = it.desc;
      if (pass == 1) dual_desc(&it.desc, &d);
      Geometry g;
      const int rc = make_geometry(&d, &g);
      if (rc != PWG_OK) return rc;
      BankImage& e = imgs[n_img];
      fill_pack_args(&d, g, it.w, it.g ? it.scale : nullptr, out, &e.a);
      e.block0 = block0;
      e.col_tiles = 0;
      // (the caller zero-fills the image buffers once: the tiled forms write the real elements only)
      static const bool tiled = !(getenv("PWG_BANK_TILED") && atoi(getenv("PWG_BANK_TILED")) == 0);
      if (d.transposed && tiled && e.a.kernel <= BANK_CHUNK / 4) {
        // polyphase image: one workgroup per (input-channel row, chunk of floor(1024 / k) output channels)
        const int chunks = ceil_div(e.a.cout_g, BANK_CHUNK / e.a.kernel);
        e.col_tiles = -chunks;
        block0 += e.a.groups * e.a.cin_g * chunks;
      } else if (!d.transposed && tiled) {
        e.col_tiles = ceil_div(e.a.cin_g * e.a.kernel, BANK_TT);
        block0 += ceil_div(e.a.groups * e.a.cout_g, BANK_TT) * e.col_tiles;
      } else {
        const long total = (long)e.a.groups * e.a.k_phase * e.a.cin_pad * e.a.m_pad;
        block0 += (int)((total + BANK_ELEMS_PER_BLOCK - 1) / BANK_ELEMS_PER_BLOCK);
      }
      ++n_img;
    }
    if (pass == 0) {
      n_fwd = n_img;
      blocks_fwd = block0;
    }
  }
  info[0] = n_rows_tab;
  info[1] = row0;
  info[2] = n_fwd;
  info[3] = blocks_fwd;
  info[4] = n_img;
  info[5] = block0;
  info[6] = (int32_t)((const char*)imgs - (const char*)table_host);
  return PWG_OK;
}

extern "C" int pwg_weight_bank_prepare(const void* table_dev, const int32_t* info, int32_t with_bwd, void* stream_) {
  PWG_REQUIRE(table_dev && info, PWG_ERR_NULL, "weight_bank_prepare: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const BankRows* rows = reinterpret_cast<const BankRows*>(table_dev);
  const BankImage* imgs = reinterpret_cast<const BankImage*>((const char*)table_dev + info[6]);
  if (info[0] > 0 && info[1] > 0) {
    ProfScope prof(stream, "bank_scale_kernel", 0, 0);
    hipLaunchKernelGGL(bank_scale_kernel, dim3(info[1]), dim3(256), 0, stream, rows, info[0]);
    PWG_CHECK_LAUNCH("bank_scale");
  }
  const int n_img = with_bwd ? info[4] : info[2];
  const int blocks = with_bwd ? info[5] : info[3];
  if (n_img > 0 && blocks > 0) {
    ProfScope prof(stream, "bank_pack_kernel", 0, 8.0 * (double)blocks * BANK_ELEMS_PER_BLOCK);
    hipLaunchKernelGGL(bank_pack_kernel, dim3(blocks), dim3(256), 0, stream, imgs, n_img);
    PWG_CHECK_LAUNCH("bank_pack");
  }
  return PWG_OK;
}
