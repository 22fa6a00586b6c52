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

template <int K>
static int launch_down(const float* x, const float* h, float* y, int batch, int t_in, int n_out, int len, int pad,
                       hipStream_t s) {
  const int qmax = (len + K - 1) / K;
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
