// Probe: issue cost of v_mfma_f32_32x32x2_f32 on gfx950 -- dependent chain vs independent accumulators,
// with and without LDS operand reads, for 1 / 2 / 3 waves per SIMD.  Prints cycles per MFMA per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ void k(float* out, long long* cyc, int iters) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a = 1.0f + lane * 0.001f, b = 0.5f;
  const float* pa = sm + lane;
  const float* pb = sm + 2048 + lane;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (LDS) {
        av[u] = pa[u * 64 + (it & 7) * 64];
        bv[u] = pb[u * 64 + (it & 7) * 64];
      } else {
        av[u] = a;
        bv[u] = b;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc[u % NACC], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, bool LDS>
void run(const char* name, int threads) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4);
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NACC, LDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mf = (double)iters * 8;
  const double tf = 256.0 * (threads / 64) * mf * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-28s waves/SIMD=%d  clock64 ticks/MFMA/wave=%7.1f  time/MFMA/wave=%6.1f ns  chip %6.1f TF\n", name, threads / 256,
         (double)c / mf, ms * 1e6 / mf, tf);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int threads : {256, 512, 768}) {
    run<1, false>("dependent chain, regs", threads);
    run<2, false>("2 accumulators, regs", threads);
    run<4, false>("4 accumulators, regs", threads);
    run<1, true>("dependent chain, LDS ops", threads);
    run<4, true>("4 accumulators, LDS ops", threads);
  }
  return 0;
}
