// pqmf.hip -- pseudo-QMF filterbank as two HBM-bound polyphase kernels (no matrix unit: 63 taps x K sub-bands is
// ~63 FMAs per sample against 8 B of traffic per sample; the launch is bounded by HBM, not by arithmetic).
//
// Reference (parallel_wavegan/layers/pqmf.py:120-149): analysis = pad(taps/2) -> full-rate FIR 1 -> K channels ->
// one-hot stride-K "pick" convolution (output floor(T / K) samples per band, ANY T); synthesis = one-hot zero-stuffing
// transposed convolution (x K) -> pad(taps/2) -> FIR K -> 1.  Only the kept samples are computed here:
//   down:  y[b][k][i] = sum_j  h[k][j]              * x[b][i K + j - pad]            (analysis; adjoint of `up`)
//   up:    x[b][t]    = sum_k sum_i g[k][t + pad - i K] * y[b][k][i]                 (synthesis; adjoint of `down`)
// The time window of a workgroup is staged in LDS in POLYPHASE order (plane r holds the samples with index = r mod K),
// so that the 63 reads of a lane are consecutive across the wave (no bank conflicts); the filter taps sit in LDS too,
// transposed so that the K taps a step needs are one broadcast read, and every lane owns 4 outputs per sub-band: one
// tap read feeds 4 K FMAs.  Backward passes reuse the two kernels with the roles of the filters swapped.
#include "common.h"

namespace pwg {

constexpr int PQMF_NI = 4;                 // outputs (down) / input positions (up) per lane: a filter tap read from LDS
constexpr int PQMF_TI = 256 * PQMF_NI;     //   feeds NI x K FMAs (round-5 first version: one scalar load per FMA, 0.9 TB/s)
constexpr int PQMF_MAXL = 256;             // filter length limit (taps + 1)

// 63 taps x K bands per K samples is 63 FMAs per 8 bytes of traffic: at 6.3 TB/s the launch needs ~50 TFMA/s, more than
// the 39 T scalar-fp32 FMAs/s of the vector units -- the accumulators of two bands ride in one v_pk_fma_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int K>
__device__ __forceinline__ void pqmf_fma_row(float (&acc)[K], const float (&c)[K], float v) {
  constexpr int P = K / 2;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    f32x2 a = {acc[2 * p], acc[2 * p + 1]};
    const f32x2 cc = {c[2 * p], c[2 * p + 1]};
    const f32x2 vv = {v, v};
    a = __builtin_elementwise_fma(cc, vv, a);
    acc[2 * p] = a.x;
    acc[2 * p + 1] = a.y;
  }
  if (K & 1) acc[K - 1] = fmaf(c[K - 1], v, acc[K - 1]);
}

// grid (ceil(n_out / TI), B); block 256; lane `tid` owns outputs i0 + tid + 256 n, n < NI (consecutive lanes read
// consecutive LDS words).  LDS: K planes of (TI + qmax) samples, then the filter transposed to [j][k] (K taps of one j
// contiguous: one broadcast read), zero past the last tap.
template <int K>
__global__ __launch_bounds__(256) void pqmf_down_kernel(const float* __restrict__ x, const float* __restrict__ h,
                                                        float* __restrict__ y, int t_in, int n_out, int len, int pad) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int qmax = (len + K - 1) / K;  // polyphase taps per plane
  const int pl = PQMF_TI + qmax;       // plane length
  float* ht = lds + K * pl;            // [qmax * K][K]
  const int i0 = blockIdx.x * PQMF_TI;
  const long b = blockIdx.y;
  const float* xb = x + b * (long)t_in;
  const int t0 = i0 * K - pad;  // window sample u = t0 + a K + r lives at plane r, position a
  // (eight independent loads in flight per lane: the trip count is a run-time value, so the compiler would otherwise
  // wait for every load before issuing the next one)
  for (int base = threadIdx.x; base < pl * K; base += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + base + 256 * u;
      v[u] = (base + 256 * u < pl * K && t >= 0 && t < t_in) ? xb[t] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + 256 * u;
      const int a = idx / K, r = idx - a * K;
      if (idx < pl * K) lds[r * pl + a] = v[u];
    }
  }
  for (int idx = threadIdx.x; idx < qmax * K * K; idx += 256) {
    const int j = idx / K, k = idx - j * K;
    ht[idx] = j < len ? h[k * len + j] : 0.f;
  }
  __syncthreads();
  float acc[PQMF_NI][K];
#pragma unroll
  for (int n = 0; n < PQMF_NI; ++n)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[n][k] = 0.f;
  const int i = threadIdx.x;
  for (int q = 0; q < qmax; ++q) {
#pragma unroll
    for (int r = 0; r < K; ++r) {
      float hv[K], xv[PQMF_NI];
#pragma unroll
      for (int k = 0; k < K; ++k) hv[k] = ht[(q * K + r) * K + k];
#pragma unroll
      for (int n = 0; n < PQMF_NI; ++n) xv[n] = lds[r * pl + i + 256 * n + q];
#pragma unroll
      for (int n = 0; n < PQMF_NI; ++n) pqmf_fma_row<K>(acc[n], hv, xv[n]);
    }
  }
#pragma unroll
  for (int n = 0; n < PQMF_NI; ++n) {
    const int io = i0 + i + 256 * n;
    if (io < n_out) {
#pragma unroll
      for (int k = 0; k < K; ++k) y[(b * K + k) * (long)n_out + io] = acc[n][k];
    }
  }
}

// grid (ceil(ceil(t_out / K) / TI), B); block 256.  Lane `tid` owns the positions q0 + tid + 256 n and produces
// x[qK .. qK + K) of each.  d = i - q runs over [dlo, dhi] = [ceil((pad - L + 1) / K), floor((K - 1 + pad) / K)];
// the taps g[k][r + pad - d K] are laid out in LDS as [k][d - dlo][r], zero where the index leaves the filter.
template <int K>
__global__ __launch_bounds__(256) void pqmf_up_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                      float* __restrict__ x, int n_in, int t_out, int len, int pad,
                                                      int dlo, int dhi) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int span = dhi - dlo + 1;
  const int pl = PQMF_TI + span;  // per-band window: i in [q0 + dlo, q0 + TI + dhi)
  float* ct = lds + K * pl;       // [K][span][K]
  const int q0 = blockIdx.x * PQMF_TI;
  const long b = blockIdx.y;
  for (int base = threadIdx.x; base < pl * K; base += 256 * 8) {  // (eight loads in flight per lane, as in `down`)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + 256 * u;
      const int k = idx / pl, a = idx - k * pl;
      const int i = q0 + dlo + a;
      v[u] = (idx < pl * K && i >= 0 && i < n_in) ? y[(b * K + k) * (long)n_in + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + 256 * u < pl * K) lds[base + 256 * u] = v[u];
  }
  for (int idx = threadIdx.x; idx < K * span * K; idx += 256) {
    const int r = idx % K, a = (idx / K) % span, k = idx / (K * span);
    const int m = r + pad - (dlo + a) * K;
    ct[idx] = (m >= 0 && m < len) ? g[k * len + m] : 0.f;
  }
  __syncthreads();
  float acc[PQMF_NI][K];
#pragma unroll
  for (int n = 0; n < PQMF_NI; ++n)
#pragma unroll
    for (int r = 0; r < K; ++r) acc[n][r] = 0.f;
  const int q = threadIdx.x;
  for (int k = 0; k < K; ++k) {
    for (int a = 0; a < span; ++a) {
      float cv[K], yv[PQMF_NI];
#pragma unroll
      for (int r = 0; r < K; ++r) cv[r] = ct[(k * span + a) * K + r];
#pragma unroll
      for (int n = 0; n < PQMF_NI; ++n) yv[n] = lds[k * pl + q + 256 * n + a];
#pragma unroll
      for (int n = 0; n < PQMF_NI; ++n) pqmf_fma_row<K>(acc[n], cv, yv[n]);
    }
  }
  __syncthreads();  // the y window is dead: reuse it to turn K strided stores per lane into coalesced ones
#pragma unroll
  for (int n = 0; n < PQMF_NI; ++n)
#pragma unroll
    for (int r = 0; r < K; ++r) lds[(q + 256 * n) * K + r] = acc[n][r];
  __syncthreads();
  float* xb = x + b * (long)t_out;
  const long tbase = (long)q0 * K;
  for (int idx = threadIdx.x; idx < PQMF_TI * K; idx += 256) {
    const long t = tbase + idx;
    if (t < t_out) xb[t] = lds[idx];
  }
}

// ---- register-window variants (the common geometries): a lane owns 4 CONSECUTIVE outputs, so the samples its taps need
// are 4 + QM - 1 consecutive words of each polyphase plane -- read ONCE as 16-B pieces into registers (conflict-free:
// consecutive lanes read consecutive 16-B pieces) instead of one 4-B LDS read per tap and output; the tap loop is fully
// unrolled (QM = polyphase taps per plane, rounded up; the filter image is zero past the last tap).  LDS instructions
// per lane: ~K (QM / 4 + 1) + K QM instead of 5 K QM -- the first version was LDS-issue bound (1.9 TB/s at 26 MB).
template <int K, int QM>
__global__ __launch_bounds__(256) void pqmf_down_reg_kernel(const float* __restrict__ x, const float* __restrict__ h,
                                                            float* __restrict__ y, int t_in, int n_out, int len, int pad,
                                                            int vec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PL = PQMF_TI + QM;  // plane length (multiple of 4)
  float* ht = lds + K * PL;         // [QM * K][K], zero past `len`
  const int i0 = blockIdx.x * PQMF_TI;
  const long b = blockIdx.y;
  const float* xb = x + b * (long)t_in;
  const int t0 = i0 * K - pad;
  for (int base = threadIdx.x; base < PL * K; base += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + base + 256 * u;
      v[u] = (base + 256 * u < PL * K && t >= 0 && t < t_in) ? xb[t] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + 256 * u;
      const int a = idx / K, r = idx - a * K;
      if (idx < PL * K) lds[r * PL + a] = v[u];
    }
  }
  for (int idx = threadIdx.x; idx < QM * K * K; idx += 256) {
    const int j = idx / K, k = idx - j * K;
    ht[idx] = j < len ? h[k * len + j] : 0.f;
  }
  __syncthreads();
  float acc[4][K];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[n][k] = 0.f;
  // plane by plane, four taps per trip of a ROLLED loop (a fully unrolled tap loop let hipcc hoist every LDS read of
  // the kernel to the top: 326 VGPRs at K = 4, one workgroup per CU): a trip needs 8 consecutive window words -- the
  // upper four become the next trip's lower four -- and K taps per step as one broadcast read
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const float* pw = lds + r * PL + 4 * threadIdx.x;
    float4 cur = *reinterpret_cast<const float4*>(pw);
#pragma unroll 1
    for (int qb = 0; qb < QM / 4; ++qb) {
      // (the last piece of the last lanes may reach past the plane: those words only meet window positions
      // >= 4 + QM - 1, i.e. taps >= QM, which are zero in the filter image)
      const float4 nxt = *reinterpret_cast<const float4*>(pw + 4 * qb + 4);
      const float w[8] = {cur.x, cur.y, cur.z, cur.w, nxt.x, nxt.y, nxt.z, nxt.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float hv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) hv[k] = ht[((4 * qb + j) * K + r) * K + k];
#pragma unroll
        for (int n = 0; n < 4; ++n) pqmf_fma_row<K>(acc[n], hv, w[n + j]);
      }
      cur = nxt;
    }
  }
  const int io = i0 + 4 * threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float* yr = y + (b * K + k) * (long)n_out;
    if (vec && io + 3 < n_out) {
      *reinterpret_cast<float4*>(yr + io) = make_float4(acc[0][k], acc[1][k], acc[2][k], acc[3][k]);
    } else {
#pragma unroll
      for (int n = 0; n < 4; ++n)
        if (io + n < n_out) yr[io + n] = acc[n][k];
    }
  }
}

// SP = span of d = i - q, rounded up to a multiple of 4; a lane owns the 4 consecutive positions q0 + 4 tid + n.
template <int K, int SP>
__global__ __launch_bounds__(256) void pqmf_up_reg_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                          float* __restrict__ x, int n_in, int t_out, int len, int pad,
                                                          int dlo, int span, int vec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PL = PQMF_TI + SP;
  constexpr int NW = 4 + SP;
  float* ct = lds + K * PL;  // [K][SP][K], zero where the tap index leaves the filter or d > dhi
  const int q0 = blockIdx.x * PQMF_TI;
  const long b = blockIdx.y;
  for (int base = threadIdx.x; base < PL * K; base += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + 256 * u;
      const int k = idx / PL, a = idx - k * PL;
      const int i = q0 + dlo + a;
      v[u] = (idx < PL * K && i >= 0 && i < n_in) ? y[(b * K + k) * (long)n_in + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + 256 * u < PL * K) lds[base + 256 * u] = v[u];
  }
  for (int idx = threadIdx.x; idx < K * SP * K; idx += 256) {
    const int r = idx % K, a = (idx / K) % SP, k = idx / (K * SP);
    const int m = r + pad - (dlo + a) * K;
    ct[idx] = (a < span && m >= 0 && m < len) ? g[k * len + m] : 0.f;
  }
  __syncthreads();
  float acc[4][K];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < K; ++r) acc[n][r] = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float yw[NW];
#pragma unroll
    for (int m = 0; m < NW / 4; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(lds + k * PL + 4 * threadIdx.x + 4 * m);
      yw[4 * m] = v.x;
      yw[4 * m + 1] = v.y;
      yw[4 * m + 2] = v.z;
      yw[4 * m + 3] = v.w;
    }
#pragma unroll
    for (int a = 0; a < SP; ++a) {
      float cv[K];
#pragma unroll
      for (int r = 0; r < K; ++r) cv[r] = ct[(k * SP + a) * K + r];
#pragma unroll
      for (int n = 0; n < 4; ++n) pqmf_fma_row<K>(acc[n], cv, yw[n + a]);
    }
  }
  __syncthreads();  // the window is dead: 4 K consecutive outputs per lane -> LDS -> coalesced 16-B stores
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < K; ++r) lds[(4 * threadIdx.x + n) * K + r] = acc[n][r];
  __syncthreads();
  float* xb = x + b * (long)t_out;
  const long tbase = (long)q0 * K;
  if (vec) {
    for (int idx = 4 * threadIdx.x; idx < PQMF_TI * K; idx += 1024) {
      const long t = tbase + idx;
      if (t + 3 < t_out) {
        *reinterpret_cast<float4*>(xb + t) = *reinterpret_cast<const float4*>(lds + idx);
      } else {
        for (int e = 0; e < 4; ++e)
          if (t + e < t_out) xb[t + e] = lds[idx + e];
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < PQMF_TI * K; idx += 256) {
      const long t = tbase + idx;
      if (t < t_out) xb[t] = lds[idx];
    }
  }
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int K, int QM>
static int launch_down_reg(const float* x, const float* h, float* y, int batch, int t_in, int n_out, int len, int pad,
                           hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)K * (PQMF_TI + QM) + (size_t)QM * K * K + 16);
  const int vec = (n_out % 4 == 0 && al16(y)) ? 1 : 0;
  hipLaunchKernelGGL((pqmf_down_reg_kernel<K, QM>), dim3(ceil_div(n_out, PQMF_TI), batch), dim3(256), lds, s, x, h, y, t_in,
                     n_out, len, pad, vec);
  PWG_CHECK_LAUNCH("pqmf_down_kernel");
  return PWG_OK;
}
template <int K, int SP>
static int launch_up_reg(const float* y, const float* g, float* x, int batch, int n_in, int t_out, int len, int pad, int dlo,
                         int span, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)K * (PQMF_TI + SP) + (size_t)K * SP * K + 16);
  const int vec = (t_out % 4 == 0 && al16(x)) ? 1 : 0;
  const int nq = ceil_div(t_out, K);
  hipLaunchKernelGGL((pqmf_up_reg_kernel<K, SP>), dim3(ceil_div(nq, PQMF_TI), batch), dim3(256), lds, s, y, g, x, n_in, t_out,
                     len, pad, dlo, span, vec);
  PWG_CHECK_LAUNCH("pqmf_up_kernel");
  return PWG_OK;
}

template <int K>
static int launch_down(const float* x, const float* h, float* y, int batch, int t_in, int n_out, int len, int pad,
                       hipStream_t s) {
  const int qmax = (len + K - 1) / K;
  // register-window variant when the window of a lane fits the register file: K (4 + QM) words
  if constexpr (K >= 2) {
    if (qmax <= 8) return launch_down_reg<K, 8>(x, h, y, batch, t_in, n_out, len, pad, s);
    if (qmax <= 16) return launch_down_reg<K, 16>(x, h, y, batch, t_in, n_out, len, pad, s);
    if (qmax <= 32) return launch_down_reg<K, 32>(x, h, y, batch, t_in, n_out, len, pad, s);
  }
  const size_t lds = sizeof(float) * ((size_t)K * (PQMF_TI + qmax) + (size_t)qmax * K * K);
  hipLaunchKernelGGL(pqmf_down_kernel<K>, dim3(ceil_div(n_out, PQMF_TI), batch), dim3(256), lds, s, x, h, y, t_in, n_out,
                     len, pad);
  PWG_CHECK_LAUNCH("pqmf_down_kernel");
  return PWG_OK;
}
template <int K>
static int launch_up(const float* y, const float* g, float* x, int batch, int n_in, int t_out, int len, int pad,
                     hipStream_t s) {
  // floor division towards -inf for the (negative) lower bound
  const int lo_num = pad - len + 1;
  const int dlo = lo_num >= 0 ? (lo_num + K - 1) / K : -((-lo_num) / K);
  const int dhi = (K - 1 + pad) / K;
  const int span = dhi - dlo + 1;
  if constexpr (K >= 2) {
    if (span <= 12) return launch_up_reg<K, 12>(y, g, x, batch, n_in, t_out, len, pad, dlo, span, s);
    if (span <= 20) return launch_up_reg<K, 20>(y, g, x, batch, n_in, t_out, len, pad, dlo, span, s);
    if (span <= 36) return launch_up_reg<K, 36>(y, g, x, batch, n_in, t_out, len, pad, dlo, span, s);
  }
  const size_t lds = sizeof(float) * ((size_t)K * (PQMF_TI + span) + (size_t)K * span * K);  // (window >= K * TI outputs)
  const int nq = ceil_div(t_out, K);
  hipLaunchKernelGGL(pqmf_up_kernel<K>, dim3(ceil_div(nq, PQMF_TI), batch), dim3(256), lds, s, y, g, x, n_in, t_out, len,
                     pad, dlo, dhi);
  PWG_CHECK_LAUNCH("pqmf_up_kernel");
  return PWG_OK;
}

}  // namespace pwg

using namespace pwg;

#define PQMF_DISPATCH(fn, ...)                 \
  switch (subbands) {                          \
    case 1: return fn<1>(__VA_ARGS__);         \
    case 2: return fn<2>(__VA_ARGS__);         \
    case 3: return fn<3>(__VA_ARGS__);         \
    case 4: return fn<4>(__VA_ARGS__);         \
    case 5: return fn<5>(__VA_ARGS__);         \
    case 6: return fn<6>(__VA_ARGS__);         \
    case 7: return fn<7>(__VA_ARGS__);         \
    default: return fn<8>(__VA_ARGS__);        \
  }

static int pqmf_check(const char* what, const void* a, const void* f, const void* o, int32_t batch, int64_t t, int64_t n,
                      int32_t subbands, int32_t len, int32_t pad) {
  PWG_REQUIRE(a && f && o, PWG_ERR_NULL, "%s: NULL pointer", what);
  PWG_REQUIRE(batch > 0 && batch <= 65535 && t > 0 && n > 0 && t < (1LL << 31) && n * subbands < (1LL << 31),
              PWG_ERR_BAD_SHAPE, "%s: bad geometry (B=%d T=%lld n=%lld)", what, batch, (long long)t, (long long)n);
  PWG_REQUIRE(subbands >= 1 && subbands <= 8, PWG_ERR_UNSUPPORTED, "%s: 1 <= subbands <= 8 (got %d)", what, subbands);
  PWG_REQUIRE(len >= 1 && len <= PQMF_MAXL && pad >= 0 && pad < len, PWG_ERR_BAD_SHAPE,
              "%s: filter length %d (<= %d) / pad %d", what, len, PQMF_MAXL, pad);
  return PWG_OK;
}

extern "C" int pwg_pqmf_down(const float* x, const float* h, float* y, int32_t batch, int64_t t_in, int64_t n_out,
                             int32_t subbands, int32_t len, int32_t pad, void* stream) {
  const int rc = pqmf_check("pqmf_down", x, h, y, batch, t_in, n_out, subbands, len, pad);
  if (rc != PWG_OK) return rc;
  // every output needs i K - pad <= T - 1 + pad at its LAST tap at most: i K + (len - 1) - pad may pass the end (zero
  // padding), but an output whose FIRST tap is past the end would be a caller error
  PWG_REQUIRE((n_out - 1) * subbands - pad < t_in, PWG_ERR_BAD_SHAPE, "pqmf_down: n_out %lld reaches past T %lld",
              (long long)n_out, (long long)t_in);
  ProfScope prof((hipStream_t)stream, "pqmf_down_kernel", 2.0 * batch * (double)n_out * subbands * len,
                 4.0 * batch * ((double)t_in + (double)n_out * subbands));
  PQMF_DISPATCH(launch_down, x, h, y, batch, (int)t_in, (int)n_out, len, pad, (hipStream_t)stream);
}

extern "C" int pwg_pqmf_up(const float* y, const float* g, float* x, int32_t batch, int64_t n_in, int64_t t_out,
                           int32_t subbands, int32_t len, int32_t pad, void* stream) {
  const int rc = pqmf_check("pqmf_up", y, g, x, batch, t_out, n_in, subbands, len, pad);
  if (rc != PWG_OK) return rc;
  ProfScope prof((hipStream_t)stream, "pqmf_up_kernel", 2.0 * batch * (double)t_out * len,
                 4.0 * batch * ((double)t_out + (double)n_in * subbands));
  PQMF_DISPATCH(launch_up, y, g, x, batch, (int)n_in, (int)t_out, len, pad, (hipStream_t)stream);
}
