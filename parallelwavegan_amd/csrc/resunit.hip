// resunit.hip -- one HiFi-GAN MRF residual unit as ONE launch (inference, C = 32 / 64):
//
//     y = ( x + conv_{k,1}( lrelu( conv_{k,d}( lrelu(x) ) + b1 ) ) + b2  [+ add2] ) [/ out_div]
//
// (/root/reference/parallel_wavegan/layers/residual_block.py:243-258: `xt = convs1[idx](x);
//  xt = convs2[idx](xt); x = xt + x`; without convs2 the unit is y = x + conv_{k,d}(lrelu(x)) + b1).
//
// Why a second convolution kernel: at C <= 64 the general implicit-GEMM kernel (conv1d.hip) is neither
// MFMA- nor HBM-bound -- a ci-chunk carries too little matrix work for its DMA + barrier round trip, and
// every convolution pays its own HBM read, write and epilogue.  Here a workgroup keeps ALL input channels
// of its column tile resident in LDS (C x (H + halo) floats, one 16-B LDS-DMA pass), so
//   * the reduction loop has no barrier and no DMA at all: B operands are shifted LDS reads at
//     (one VGPR base per tap) + immediate, A operands (weights) stream from L2 into registers through a
//     pre-swizzled image whose 256-B records ARE v_mfma_f32_32x32x2_f32 A operands (prefetched one tap ahead);
//   * the intermediate h = lrelu(conv1 + b1) never leaves the CU: it is written in MFMA D layout straight
//     into a second LDS tile (zero outside [0,T) = conv2's zero padding) and is conv2's B operand;
//   * the residual comes from the resident raw x tile, the result is transposed through the same tile and
//     leaves as row-contiguous 16-B stores.
// HBM traffic of a unit drops from 5 tensor passes (x, h, h, x, y) to 2; the price is the recomputed halo:
// a tile of H = 256 (C = 32) / 128 (C = 64) h-columns yields H - (k-1) outputs.
//
// Work split: 4 waves = (C/32 row blocks) x (4 / (C/32)) column groups, each wave 32 rows x 64 columns
// (two accumulator tiles).  LDS: C=32: 40 KB x + 32 KB h; C=64: 46 KB x + 32 KB h -> 2 workgroups per CU,
// so one workgroup's load / store phases overlap the other's MFMA phases.
#include "common.h"

#include <stdint.h>
#include <stdlib.h>

namespace pwg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct ResUnitArgs {
  const float* x;
  const float* w1;  // packed [tap][C/32][C/8][64 lanes][4]
  const float* b1;
  const float* w2;  // nullptr: single convolution
  const float* b2;
  const float* add2;
  float* y;
  int T;
  int k;
  int d1;
  int bn_out;  // outputs per tile (multiple of 4)
  int hla;     // x-tile column of output slot 0 (left halo rounded up to a multiple of 4)
  int sh;      // hla - (true left halo)
  int p2;      // (k-1)/2 in pair mode, 0 otherwise
  int xw4;     // 16-B pieces staged per x row
  float slope1, slope2, out_div;
  int dbg;      // timing experiments (PWG_RU_DBG): 1 = no LeakyReLU on the B operands, 2 = weights loaded once (no per-tap A loads),
                // 4 = no x-tile DMA, 8 = no global stores
};

template <int C>
struct ResUnitCfg {
  static constexpr int MB = C / 32;       // 32-row blocks = waves along M
  static constexpr int WAVES_N = 4 / MB;  // waves along the columns
  static constexpr int H = WAVES_N * 64;  // h-columns / output slots per workgroup
  static constexpr int XS = C == 32 ? 320 : 184;  // x-tile row stride (floats)
  static constexpr int HS = H;                     // h-tile row stride
  static constexpr int CP = C / 2;                 // channel pairs = MFMA k-steps per tap
};

// D += W (*) B over (tap, channel pair).  wl: this wave's row block of the packed image + lane;
// bl: LDS lane base (row lhi, first column of the wave); RS: LDS row stride; tap_step: columns per tap.
template <int C, int RS, bool ACT>
__device__ __forceinline__ void resunit_contract(const float* __restrict__ wl, const float* bl, int tap_step, int k,
                                                 float slope, f32x16 (&acc)[2], bool reload_a = true) {
  using Cfg = ResUnitCfg<C>;
  constexpr int CP = Cfg::CP;
  constexpr int G = 8;        // channel pairs per operand group (LDS reads of group g+1 fly under group g's MFMAs)
  constexpr int NG = CP / G;  // 2 or 4 (even: the B ping-pong phase is the same at every tap start)
  constexpr int TAP_W = Cfg::MB * CP * 64;  // floats per tap of the packed image
  float A0[CP], A1[CP], B0[G][2], B1[G][2];
  // one global_load_dwordx4 per 4 channel pairs (image [tap][row block][cp/4][lane][4]): with a dword per channel
  // pair the loop ran at 118 TFLOP/s beside 2 workgroups per CU, with 16-B loads at 144 (tools/probes/mfma_loop.hip)
  auto load_a = [&](float(&A)[CP], int tap) {
    const float4* p4 = reinterpret_cast<const float4*>(wl + (long)tap * TAP_W);
#pragma unroll
    for (int q = 0; q < CP / 4; ++q) {
      const float4 v = p4[q * 64];
      A[4 * q] = v.x;
      A[4 * q + 1] = v.y;
      A[4 * q + 2] = v.z;
      A[4 * q + 3] = v.w;
    }
  };
  auto load_b = [&](float(&B)[G][2], int tap, int g) {
    const float* p = bl + tap * tap_step;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      B[i][0] = p[2 * (g * G + i) * RS];
      B[i][1] = p[2 * (g * G + i) * RS + 32];
    }
  };
  auto mma = [&](const float* A, float(&B)[G][2]) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        float v = B[i][ni];
        if (ACT) v = __builtin_fmaxf(v, v * slope);  // LeakyReLU, 0 < slope < 1 (exact)
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i], v, acc[ni], 0, 0, 0);
      }
    }
  };
  // one tap: NG operand groups; the group after the last one is group 0 of the next tap (for the last
  // tap that is a harmless read past the taps: never fed to an MFMA)
  auto tap_body = [&](float(&A)[CP], int tap) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // (sched_barrier: hipcc otherwise sinks the next group's LDS reads to their first use)
      if (g & 1) {
        if (g + 1 < NG) load_b(B0, tap, g + 1); else load_b(B0, tap + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(&A[g * G], B1);
      } else {
        if (g + 1 < NG) load_b(B1, tap, g + 1); else load_b(B1, tap + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(&A[g * G], B0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  load_a(A0, 0);
  load_b(B0, 0, 0);
  int tap = 0;
  for (; tap + 2 <= k; tap += 2) {
    if (reload_a || tap == 0) load_a(A1, tap + 1);
    __builtin_amdgcn_sched_barrier(0);
    tap_body(A0, tap);
    if (reload_a) load_a(A0, tap + 2 < k ? tap + 2 : k - 1);
    __builtin_amdgcn_sched_barrier(0);
    tap_body(A1, tap + 1);
  }
  if (tap < k) tap_body(A0, tap);
}

template <int C>
__global__ __launch_bounds__(256, 2) void resunit_kernel(ResUnitArgs a) {
  using Cfg = ResUnitCfg<C>;
  constexpr int WAVES_N = Cfg::WAVES_N, XS = Cfg::XS, HS = Cfg::HS, CP = Cfg::CP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;           // [C][XS]  raw x, later the result tile
  float* hs = smem + C * XS;  // [C][HS]  lrelu(conv1 + b1), zero outside the sequence

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int l31 = lane & 31;
  const int lhi = lane >> 5;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * a.bn_out;  // first output sample of this tile
  const int f0 = t0 - a.hla;             // sample of x-tile column 0
  const int T = a.T;
  const bool pair = a.w2 != nullptr;

  // ---- stage the x tile: C rows x xw4 16-B pieces, LDS-DMA (zero outside the row = implicit padding)
  const float* xb = a.x + (long)b * C * T;
  __amdgpu_buffer_rsrc_t x_rs = uniform_buffer_rsrc(xb, (unsigned)(C * T) * 4u);
  const int XW = a.xw4 * 4;
  if (a.dbg & 4) {
  } else if (__builtin_amdgcn_readfirstlane((f0 >= 0 && f0 + XW <= T) ? 1 : 0)) {
    for (int r = wave; r < C; r += 4)
      for (int l0 = 0; l0 < a.xw4; l0 += 64)
        if (l0 + lane < a.xw4) {
          const unsigned off = (unsigned)(r * T + f0 + 4 * (l0 + lane)) * 4u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS + 4 * l0), 16, off, 0, 0, 0);
        }
  } else {
    // first / last tiles of a sequence: per-dword range check (a 16-B access that starts left of the
    // buffer is dropped whole, tools/probes/glds_x4.hip)
    for (int r = wave; r < C; r += 4)
      for (int i0 = 0; i0 < XW; i0 += 64)
        if (i0 + lane < XW) {
          const int f = f0 + i0 + lane;
          const unsigned off = (f >= 0 && f < T) ? (unsigned)(r * T + f) * 4u : 0xFFFFFFFCu;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS + i0), 4, off, 0, 0, 0);
        }
  }

  // biases of this lane's 16 accumulator rows (row = 8*(r>>2) + 4*lhi + (r&3) of the wave's block)
  f32x16 bias1, bias2;
  {
    const float* bp1 = a.b1 ? a.b1 : a.x;  // (any readable address: the value is discarded)
    const float* bp2 = (pair && a.b2) ? a.b2 : a.x;
    const bool has1 = a.b1 != nullptr, has2 = pair && a.b2 != nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = wave_m * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
      const float v1 = bp1[c], v2 = bp2[c];
      bias1[r] = has1 ? v1 : 0.f;
      bias2[r] = has2 ? v2 : 0.f;
    }
  }

  f32x16 acc[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- phase 1: conv_{k,d} over lrelu(x); h column m (output slot m in single mode) reads x-tile
  // column m + tap*d + sh
  const float* w1l = a.w1 + (long)wave_m * (CP * 64) + lane * 4;
  if (a.dbg & 1)
    resunit_contract<C, XS, false>(w1l, xs + lhi * XS + wave_n * 64 + l31 + a.sh, a.d1, a.k, a.slope1, acc, !(a.dbg & 2));
  else
    resunit_contract<C, XS, true>(w1l, xs + lhi * XS + wave_n * 64 + l31 + a.sh, a.d1, a.k, a.slope1, acc, !(a.dbg & 2));

  if (pair) {
    // h = lrelu(acc + b1) (conv2's pre-activation applied at production), 0 outside [0, T)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int m = wave_n * 64 + ni * 32 + l31;
      const int th = t0 - a.p2 + m;
      const bool inside = th >= 0 && th < T;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = wave_m * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
        float v = acc[ni][r] + bias1[r];
        v = __builtin_fmaxf(v, v * a.slope2);
        hs[c * HS + m] = inside ? v : 0.f;
        acc[ni][r] = 0.f;
      }
    }
    __syncthreads();
    // ---- phase 2: conv_{k,1} over h; output slot n reads h column n + tap
    const float* w2l = a.w2 + (long)wave_m * (CP * 64) + lane * 4;
    resunit_contract<C, HS, false>(w2l, hs + lhi * HS + wave_n * 64 + l31, 1, a.k, 1.f, acc, !(a.dbg & 2));
  } else {
    __syncthreads();  // the other waves may still be reading x columns this wave is about to overwrite
  }

  // ---- result = acc + bias + x, transposed through the x tile (slot n lives at column n + hla)
  const f32x16 bias_out = pair ? bias2 : bias1;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = wave_n * 64 + ni * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = wave_m * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
      float* p = xs + c * XS + a.hla + n;
      *p = (acc[ni][r] + bias_out[r]) + *p;
    }
  }
  __syncthreads();

  // ---- row-contiguous 16-B stores (T % 4 == 0: a float4 is inside the sequence or outside it)
  const int nv = a.bn_out >> 2;
  const bool do_div = a.out_div != 1.0f;
  for (int c = wave; c < C; c += 4) {
    const long row = ((long)b * C + c) * T;
    for (int q = lane; q < nv; q += 64) {
      const int t = t0 + 4 * q;
      if (t >= T || (a.dbg & 8)) continue;
      float4 v = *reinterpret_cast<const float4*>(xs + c * XS + a.hla + 4 * q);
      if (a.add2) {
        const float4 u = *reinterpret_cast<const float4*>(a.add2 + row + t);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (do_div) {
        v.x = v.x / a.out_div; v.y = v.y / a.out_div; v.z = v.z / a.out_div; v.w = v.w / a.out_div;
      }
      *reinterpret_cast<float4*>(a.y + row + t) = v;
    }
  }
}

// packed image: [tap][row block][channel pair / 4][lane][4]; element j of a lane's 16 B is the A operand of channel
// pair cp = 4 * (cp / 4) + j: lane -> (row = lane & 31, channel = 2 * cp + (lane >> 5)) of v_mfma_f32_32x32x2_f32
__global__ void resunit_pack_kernel(const float* w, const float* scale, float* out, int C, int k) {
  const int total = k * C * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 3;
    const int lane = (i >> 2) & 63;
    int r = i >> 8;
    const int q = r % (C / 8);
    r /= (C / 8);
    const int mb = r % (C / 32);
    const int tap = r / (C / 32);
    const int cp = 4 * q + j;
    const int m = mb * 32 + (lane & 31);
    const int ci = 2 * cp + (lane >> 5);
    float v = w[((long)m * C + ci) * k + tap];
    if (scale) v *= scale[m];
    out[i] = v;
  }
}

struct ResUnitGeom {
  int H, XS, hla, sh, p2, bn_out, xw4, tiles;
  size_t lds;
};

static bool resunit_geometry(const pwg_resunit_desc* d, ResUnitGeom* g) {
  const int C = d->channels;
  if (C != 32 && C != 64) return false;
  if (d->kernel < 1 || (d->kernel & 1) == 0 || d->dilation < 1 || d->batch < 1 || d->t < 4 || (d->t & 3)) return false;
  if (d->batch > 65535) return false;
  if (!(d->slope1 > 0.f && d->slope1 < 1.f)) return false;
  if (d->has_conv2 && !(d->slope2 > 0.f && d->slope2 < 1.f)) return false;
  if ((long)C * d->t * 4 >= (1L << 32)) return false;  // one buffer descriptor per item
  g->H = C == 32 ? ResUnitCfg<32>::H : ResUnitCfg<64>::H;
  g->XS = C == 32 ? ResUnitCfg<32>::XS : ResUnitCfg<64>::XS;
  const int k = d->kernel;
  const int p1 = (k - 1) / 2 * d->dilation;
  g->p2 = d->has_conv2 ? (k - 1) / 2 : 0;
  const int hl = p1 + g->p2;
  g->hla = (hl + 3) & ~3;
  g->sh = g->hla - hl;
  g->bn_out = d->has_conv2 ? ((g->H - (k - 1)) & ~3) : g->H;
  const int need = g->H + (k - 1) * d->dilation + g->sh;
  g->xw4 = (need + 3) / 4;
  if (g->bn_out < 64 || 4 * g->xw4 > g->XS || g->hla + g->H > g->XS) return false;
  g->tiles = ceil_div(d->t, g->bn_out);
  g->lds = ((size_t)C * g->XS + (d->has_conv2 ? (size_t)C * g->H : 0)) * sizeof(float);
  return true;
}

}  // namespace pwg

using namespace pwg;

extern "C" {

int pwg_resunit_supported(const pwg_resunit_desc* d) {
  ResUnitGeom g;
  return d != nullptr && resunit_geometry(d, &g) ? 1 : 0;
}

// measured on MI355X (tools/bench_resunit.py, B16 x 800 frames): the one-launch unit wins x1.10-1.41 at C = 32
// and x1.17 at C = 64, k = 3, tied at C = 64, k = 7 in round 2 (with 2.5x less HBM traffic) and loses 7 % at C = 64,
// k = 11, where the recomputed halo (128 h-columns for 116 outputs, in both phases) outweighs the saved passes
int pwg_resunit_profitable(const pwg_resunit_desc* d) {
  if (!pwg_resunit_supported(d)) return 0;
  if (!d->has_conv2) return 1;
  // round 3: with three resident workgroups per CU the two general launches are 3 % faster than the fused unit at
  // C = 64, k = 7 in isolation (5052 vs 5227 us per block, tools/bench_resunit.py) and 0.4 % on the whole forward
  // (PWG_RU_K64=3, same-box A/B) at 2.5x the HBM traffic of those units: within box-to-box variation, rule unchanged
  static const int k64 = getenv("PWG_RU_K64") ? atoi(getenv("PWG_RU_K64")) : 7;  // largest fused kernel size at C = 64
  return (d->channels == 32 || d->kernel <= k64) ? 1 : 0;
}

size_t pwg_resunit_packed_weight_floats(int32_t channels, int32_t kernel) {
  return (size_t)kernel * channels * channels;
}

int pwg_resunit_pack_weight(int32_t channels, int32_t kernel, const float* w, const float* scale, float* w_packed,
                            void* stream) {
  PWG_REQUIRE(w && w_packed, PWG_ERR_NULL, "resunit_pack_weight: null pointer");
  PWG_REQUIRE((channels == 32 || channels == 64) && kernel >= 1, PWG_ERR_UNSUPPORTED,
              "resunit_pack_weight: channels=%d kernel=%d", channels, kernel);
  const int total = kernel * channels * channels;
  hipLaunchKernelGGL(resunit_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, scale,
                     w_packed, channels, kernel);
  PWG_CHECK_LAUNCH("resunit_pack_weight");
  return PWG_OK;
}

int pwg_resunit_forward(const pwg_resunit_desc* d, const float* x, const float* w1_packed, const float* b1,
                        const float* w2_packed, const float* b2, const float* add2, float* y, void* stream_) {
  PWG_REQUIRE(d && x && w1_packed && y, PWG_ERR_NULL, "resunit_forward: null pointer");
  PWG_REQUIRE(x != y, PWG_ERR_BAD_SHAPE, "resunit_forward: y must not alias x (tiles read their neighbours' halo)");
  ResUnitGeom g;
  PWG_REQUIRE(resunit_geometry(d, &g), PWG_ERR_UNSUPPORTED,
              "resunit_forward: unsupported unit (C=%d k=%d d=%d T=%d): use pwg_conv1d_forward", d->channels,
              d->kernel, d->dilation, d->t);
  PWG_REQUIRE((d->has_conv2 != 0) == (w2_packed != nullptr), PWG_ERR_BAD_SHAPE,
              "resunit_forward: has_conv2=%d but w2_packed is %s", d->has_conv2, w2_packed ? "given" : "null");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  PWG_REQUIRE(al16(x) && al16(y) && al16(add2), PWG_ERR_BAD_SHAPE, "resunit_forward: x / y / add2 must be 16-B aligned");
  hipStream_t stream = (hipStream_t)stream_;
  ResUnitArgs a;
  a.x = x; a.w1 = w1_packed; a.b1 = b1; a.w2 = w2_packed; a.b2 = b2; a.add2 = add2; a.y = y;
  a.T = d->t; a.k = d->kernel; a.d1 = d->dilation;
  a.bn_out = g.bn_out; a.hla = g.hla; a.sh = g.sh; a.p2 = g.p2; a.xw4 = g.xw4;
  a.slope1 = d->slope1; a.slope2 = d->slope2; a.out_div = d->out_div;
  static const int dbg = getenv("PWG_RU_DBG") ? atoi(getenv("PWG_RU_DBG")) : 0;
  a.dbg = dbg;
  void (*kern)(ResUnitArgs) = d->channels == 32 ? resunit_kernel<32> : resunit_kernel<64>;
  if (g.lds > 64 * 1024 && !lds_limit_is_set(reinterpret_cast<const void*>(kern), g.lds)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    PWG_REQUIRE(e == hipSuccess, PWG_ERR_LAUNCH, "resunit_forward: cannot raise LDS limit to %zu: %s", g.lds,
                hipGetErrorString(e));
  }
  const double C = d->channels;
  const double elems = (double)d->batch * C * d->t;
  const double flops = 2.0 * elems * C * d->kernel * (d->has_conv2 ? 2 : 1);
  const double bytes = 4.0 * (elems * (2 + (add2 != nullptr)) + (d->has_conv2 ? 2 : 1) * C * C * d->kernel);
  maybe_poison_lds(stream);
  {
    ProfScope prof(stream,
                   prof_shape_name("resunit_kernel", "resunit_kernel B%d C%d T%d k%d d%d pair%d", d->batch, d->channels,
                                   d->t, d->kernel, d->dilation, d->has_conv2),
                   flops, bytes);
    hipLaunchKernelGGL(kern, dim3(g.tiles, d->batch), dim3(256), g.lds, stream, a);
  }
  PWG_CHECK_LAUNCH("resunit_forward");
  return PWG_OK;
}

}  // extern "C"
