// Probe (prepared at the end of round 2, first run planned for round 3): would the general conv kernel gain from
// streaming its weights from L2 straight into registers (16-B A-operand records, the layout that took the
// residual-unit loop from 118 to 144 TFLOP/s in mfma_loop.hip) instead of staging them through LDS with LDS-DMA?
//
// Two stripped-down replicas of the FAST 128x128x4 tile of csrc/conv1d.hip (2x2 waves, each 64 rows x 64 columns,
// 4 input channels per chunk, stride 1, x rows staged by one 16-B LDS-DMA instruction per row, double buffered, one
// vmcnt(0) + barrier per chunk), no epilogue beyond storing the accumulators:
//   variant L: weights [tap][ci][128] per chunk through global_load_lds_dwordx4 (1 KiB pieces) + ds_read_b32
//   variant R: weights as [chunk][tap][wave_m][lane][4] records = {A(kk0,mi0), A(kk0,mi1), A(kk1,mi0), A(kk1,mi1)}
//              of one lane: ONE global_load_dwordx4 per tap and wave, prefetched one tap ahead; LDS holds x only
// on C_in = C_out = 128, T = 51200, B = 16 (HiFi-GAN V1 stage 2 at the bench shape) for k = 3 / 7 / 11.
// Prints TFLOP/s of both and max|L - R| over the outputs (same math, same summation order).
// build: hipcc --offload-arch=gfx950 -O3 -w -fno-honor-nans -mno-amdgpu-ieee -o conv_wreg.bin conv_wreg.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int C = 128, BM = 128, BN = 128, CK = 4, XS = BN + 64;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// x: (B, C, T); y: (B, C, T) (only the interior tiles are launched: no padding logic); k taps, dilation 1
template <bool WREG>
__global__ __launch_bounds__(256, 2) void conv_probe(const float* x, const float* wl /* [k][C][C] m fastest */,
                                                     const float* wr /* [C/CK][k][2][64][4] */, float* y, int T, int k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int buf_floats = CK * XS + (WREG ? 0 : k * CK * BM);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave >> 1, wave_n = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.y;
  const int n0 = (blockIdx.x + 1) * BN;  // interior tiles: [n0 - pad, n0 + BN + pad) inside the row
  const int pad = (k - 1) / 2;
  const float* xb = x + (long)b * C * T;
  __amdgpu_buffer_rsrc_t x_rs = rsrc(xb, (unsigned)(C * T) * 4u);

  f32x16 acc[2][2];
  for (int mi = 0; mi < 2; ++mi)
    for (int ni = 0; ni < 2; ++ni)
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  auto issue = [&](int c, float* buf) {
    float* xs = buf;
    // one 16-B LDS-DMA instruction per x row: (BN + 64) floats = 48 lanes
    for (int r = wave; r < CK; r += 4)
      if (lane < XS / 4) {
        const unsigned off = (unsigned)((c * CK + r) * T + n0 - pad + 4 * lane) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS), 16, off, 0, 0, 0);
      }
    if (!WREG) {
      float* ws = buf + CK * XS;  // [tap][ci][BM]
      const int npieces = k * CK * BM / 256;
      for (int p = wave; p < npieces; p += 4) {
        const int rr = 2 * p + (lane >> 5);  // one piece = 256 floats = two (tap, ci) rows of 128; this lane's row
        const int tap = rr / CK, ci = rr - tap * CK;
        const float* src = wl + ((long)tap * C + c * CK + ci) * C + (lane & 31) * 4;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(ws + p * 256), 16, 0, 0);
      }
    }
  };

  const int nchunks = C / CK;
  issue(0, smem);
  float4 a_cur, a_nxt;
  const float4* wr4 = reinterpret_cast<const float4*>(wr) + wave_m * 64 + lane;  // + (chunk * k + tap) * 128
  if (WREG) a_cur = wr4[0];
  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* buf = smem + (c & 1) * buf_floats;
    if (c + 1 < nchunks) issue(c + 1, smem + ((c + 1) & 1) * buf_floats);
    const float* xl = buf + lhi * XS + wave_n * 64 + l31;
    const float* wlds = buf + CK * XS + wave_m * 64 + l31 + lhi * BM;
    for (int tap = 0; tap < k; ++tap) {
      float av[2][2], bv[2][2];
      if (WREG) {
        // next record: next tap of this chunk, or tap 0 of the next chunk (the last one re-reads: harmless)
        const int nxt = (c * k + tap + 1 < nchunks * k) ? c * k + tap + 1 : c * k + tap;
        a_nxt = wr4[(long)nxt * 128];
        av[0][0] = a_cur.x; av[0][1] = a_cur.y; av[1][0] = a_cur.z; av[1][1] = a_cur.w;
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) av[kk][mi] = wlds[(tap * CK + 2 * kk) * BM + mi * 32];
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bv[kk][ni] = xl[2 * kk * XS + tap + ni * 32];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][mi], bv[kk][ni], acc[mi][ni], 0, 0, 0);
      if (WREG) a_cur = a_nxt;
    }
  }
  // plain D-layout stores (the probe measures the loop, not the epilogue)
  for (int mi = 0; mi < 2; ++mi)
    for (int ni = 0; ni < 2; ++ni)
      for (int r = 0; r < 16; ++r) {
        const int m = wave_m * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const int n = n0 + wave_n * 64 + ni * 32 + l31;
        y[((long)b * C + m) * T + n] = acc[mi][ni][r];
      }
}

int main() {
  const int B = 16, T = 51200;
  std::vector<float> hx((size_t)B * C * T);
  unsigned h = 1u;
  for (auto& v : hx) {
    h = h * 1664525u + 1013904223u;
    v = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
  }
  float *x, *yl, *yr, *wl, *wr;
  (void)hipMalloc(&x, hx.size() * 4);
  (void)hipMalloc(&yl, hx.size() * 4);
  (void)hipMalloc(&yr, hx.size() * 4);
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(yl, 0, hx.size() * 4);
  (void)hipMemset(yr, 0, hx.size() * 4);
  for (int k : {3, 7, 11}) {
    std::vector<float> hw((size_t)k * C * C), hwl(hw.size()), hwr(hw.size());
    for (auto& v : hw) {  // hw[m][ci][tap]
      h = h * 1664525u + 1013904223u;
      v = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 27));
    }
    for (int tap = 0; tap < k; ++tap)
      for (int ci = 0; ci < C; ++ci)
        for (int m = 0; m < C; ++m) hwl[((size_t)tap * C + ci) * C + m] = hw[((size_t)m * C + ci) * k + tap];
    for (int c = 0; c < C / CK; ++c)
      for (int tap = 0; tap < k; ++tap)
        for (int wm = 0; wm < 2; ++wm)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
              const int kk = j >> 1, mi = j & 1;
              const int m = wm * 64 + mi * 32 + (lane & 31), ci = c * CK + 2 * kk + (lane >> 5);
              hwr[((((size_t)c * k + tap) * 2 + wm) * 64 + lane) * 4 + j] = hw[((size_t)m * C + ci) * k + tap];
            }
    (void)hipMalloc(&wl, hw.size() * 4);
    (void)hipMalloc(&wr, hw.size() * 4);
    (void)hipMemcpy(wl, hwl.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(wr, hwr.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const dim3 grid(T / BN - 2, B);
    const double flops = 2.0 * grid.x * BN * (double)B * C * C * k;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms[2];
    for (int v = 0; v < 2; ++v) {
      const size_t lds = 2 * (CK * XS + (v ? 0 : k * CK * BM)) * sizeof(float);
      auto kern = v ? conv_probe<true> : conv_probe<false>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, x, wl, wr, v ? yr : yl, T, k);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms[v], e0, e1);
      }
    }
    std::vector<float> a(hx.size()), bvec(hx.size());
    (void)hipMemcpy(a.data(), yl, a.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(bvec.data(), yr, a.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      md = fmax(md, fabs((double)a[i] - bvec[i]));
      mx = fmax(mx, fabs((double)a[i]));
    }
    printf("k=%2d  LDS weights %7.1f us %6.1f TFLOP/s | register weights %7.1f us %6.1f TFLOP/s | max|L-R| %.2e (max|y| %.2e)\n", k,
           ms[0] * 1e3, flops / (ms[0] * 1e-3) / 1e12, ms[1] * 1e3, flops / (ms[1] * 1e-3) / 1e12, md, mx);
    (void)hipFree(wl);
    (void)hipFree(wr);
  }
  return 0;
}
