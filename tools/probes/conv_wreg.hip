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

constexpr int C = 128, BM = 128;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// x: (B, C, T); y: (B, C, T) (only the interior tiles are launched: no padding logic); k taps, dilation 1
// CK: input channels per chunk (per barrier); WAVES_N: column groups of 64 (2 -> 128x128 tile, 4 waves; 4 -> 128x256, 8 waves)
template <bool WREG, int CK, int WAVES_N>
__global__ __launch_bounds__(128 * WAVES_N, 2) void conv_probe(const float* x, const float* wl /* [k][C][C] m fastest */,
                                                     const float* wr /* [C/CK][k][2][64][4] */, float* y, int T, int k) {
  constexpr int BN = 64 * WAVES_N, XS = BN + 64, NW = 2 * WAVES_N, KS = CK / 2;
  static_assert(!WREG || CK == 4, "register-weight records hold 4 channels");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int buf_floats = CK * XS + (WREG ? 0 : k * CK * BM);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N, l31 = lane & 31, lhi = lane >> 5;
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
    // 16-B LDS-DMA per x row: (BN + 64) floats = 48 lanes (BN = 128) / 64 + 16 lanes (BN = 256)
    for (int r = wave; r < CK; r += NW)
      for (int l0 = 0; l0 < XS / 4; l0 += 64)
        if (l0 + lane < XS / 4) {
          const unsigned off = (unsigned)((c * CK + r) * T + n0 - pad + 4 * (l0 + lane)) * 4u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xs + r * XS + 4 * l0), 16, off, 0, 0, 0);
        }
    if (!WREG) {
      float* ws = buf + CK * XS;  // [tap][ci][BM]
      const int npieces = k * CK * BM / 256;
      for (int p = wave; p < npieces; p += NW) {
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
      float av[KS][2], bv[KS][2];
      if (WREG) {
        // next record: next tap of this chunk, or tap 0 of the next chunk (the last one re-reads: harmless)
        const int nxt = (c * k + tap + 1 < nchunks * k) ? c * k + tap + 1 : c * k + tap;
        a_nxt = wr4[(long)nxt * 128];
        av[0][0] = a_cur.x; av[0][1] = a_cur.y; av[KS - 1][0] = a_cur.z; av[KS - 1][1] = a_cur.w;
      } else {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) av[kk][mi] = wlds[(tap * CK + 2 * kk) * BM + mi * 32];
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bv[kk][ni] = xl[2 * kk * XS + tap + ni * 32];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
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
    for (int c = 0; c < C / 4; ++c)  // register-weight records: 4 channels per chunk
      for (int tap = 0; tap < k; ++tap)
        for (int wm = 0; wm < 2; ++wm)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
              const int kk = j >> 1, mi = j & 1;
              const int m = wm * 64 + mi * 32 + (lane & 31), ci = c * 4 + 2 * kk + (lane >> 5);
              hwr[((((size_t)c * k + tap) * 2 + wm) * 64 + lane) * 4 + j] = hw[((size_t)m * C + ci) * k + tap];
            }
    (void)hipMalloc(&wl, hw.size() * 4);
    (void)hipMalloc(&wr, hw.size() * 4);
    (void)hipMemcpy(wl, hwl.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(wr, hwr.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    struct Variant {
      const char* name;
      void (*kern)(const float*, const float*, const float*, float*, int, int);
      int ck, bn;
      bool wreg;
    };
    const Variant variants[] = {
        {"LDS weights 128x128x4", conv_probe<false, 4, 2>, 4, 128, false},
        {"reg weights 128x128x4", conv_probe<true, 4, 2>, 4, 128, true},
        {"LDS weights 128x128x8", conv_probe<false, 8, 2>, 8, 128, false},
        {"LDS weights 128x128x16", conv_probe<false, 16, 2>, 16, 128, false},
        {"LDS weights 128x256x4 (8 waves)", conv_probe<false, 4, 4>, 4, 256, false},
        {"LDS weights 128x256x8 (8 waves)", conv_probe<false, 8, 4>, 8, 256, false},
    };
    float ms[2] = {0, 0};
    for (const Variant& v : variants) {
      const size_t lds = 2 * ((size_t)v.ck * (v.bn + 64) + (v.wreg ? 0 : (size_t)k * v.ck * BM)) * sizeof(float);
      if (lds > 160 * 1024) continue;
      const dim3 grid(T / v.bn - 2, B);
      const double flops = 2.0 * grid.x * v.bn * (double)B * C * C * k;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(v.kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      float t = 0, t8 = 0;
      // second timing with the LDS allocation inflated so that exactly 8 waves share a CU (2 per SIMD), the
      // occupancy of the real kernel's 174-VGPR waves: 2 workgroups of 4 waves or 1 workgroup of 8
      const size_t lds8 = v.bn == 128 ? 80 * 1024 : 159 * 1024;
      for (int pass = 0; pass < 2; ++pass) {
        const size_t l = pass ? (lds8 > lds ? lds8 : lds) : lds;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(v.kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);
        for (int rep = 0; rep < 3; ++rep) {
          (void)hipEventRecord(e0, 0);
          hipLaunchKernelGGL(v.kern, grid, dim3(2 * v.bn), l, 0, x, wl, wr, v.wreg ? yr : yl, T, k);
          (void)hipEventRecord(e1, 0);
          (void)hipDeviceSynchronize();
          (void)hipEventElapsedTime(pass ? &t8 : &t, e0, e1);
        }
      }
      printf("k=%2d  %-34s %7.1f us %6.1f TFLOP/s (LDS %zu KB) | at 8 waves per CU %7.1f us %6.1f TFLOP/s\n", k, v.name, t * 1e3,
             flops / (t * 1e-3) / 1e12, lds / 1024, t8 * 1e3, flops / (t8 * 1e-3) / 1e12);
      if (&v == &variants[0]) ms[0] = t;
      if (&v == &variants[1]) ms[1] = t;
    }
    // compare the first two variants (same tile, same summation order)
    (void)hipMemset(yl, 0, hx.size() * 4);
    (void)hipMemset(yr, 0, hx.size() * 4);
    {
      const dim3 grid(T / 128 - 2, B);
      const size_t l0 = 2 * ((size_t)4 * 192 + (size_t)k * 4 * BM) * sizeof(float), l1 = 2 * (size_t)4 * 192 * sizeof(float);
      hipLaunchKernelGGL((conv_probe<false, 4, 2>), grid, dim3(256), l0, 0, x, wl, wr, yl, T, k);
      hipLaunchKernelGGL((conv_probe<true, 4, 2>), grid, dim3(256), l1, 0, x, wl, wr, yr, T, k);
      (void)hipDeviceSynchronize();
    }
    std::vector<float> a(hx.size()), bvec(hx.size());
    (void)hipMemcpy(a.data(), yl, a.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(bvec.data(), yr, a.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      md = fmax(md, fabs((double)a[i] - bvec[i]));
      mx = fmax(mx, fabs((double)a[i]));
    }
    printf("k=%2d  max|LDS - reg| over the outputs %.2e (max|y| %.2e)\n", k, md, mx);
    (void)hipFree(wl);
    (void)hipFree(wr);
  }
  return 0;
}
