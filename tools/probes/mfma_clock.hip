// Probe: SUSTAINED fp32 MFMA rate and shader clock of an MI355X.  Back-to-back launches (~0.25 s each, ~5 s in
// total) of a pure v_mfma_f32_32x32x2_f32 loop on every CU (2 workgroups x 4 waves per CU, operands in
// registers); per launch: TFLOP/s from HIP events and the shader clock = clock64() cycles / wall_clock64()
// (100 MHz) ticks of one wave.  Tells how much of the gap between a kernel's TFLOP/s and the nominal
// 157.3 TFLOP/s (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz) is clock, not kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_clock.bin mfma_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// RANDOM = false: constant operands (few toggling bits); true: 8 + 8 pseudo-random operand registers per lane,
// zero-mean so that the accumulators stay finite -- the data-dependent power of a real convolution
template <bool RANDOM>
__global__ __launch_bounds__(256, 2) void k(float* out, long long* stamps, int iters) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float av[8], bv[8];
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  for (int u = 0; u < 8; ++u) {
    h = h * 1664525u + 1013904223u;
    av[u] = RANDOM ? ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 1.0f + lane * 0.001f;
    h = h * 1664525u + 1013904223u;
    bv[u] = RANDOM ? ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 0.5f;
  }
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it += 8) {  // (static register indices: 64 MFMAs per trip)
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[(u + s) & 7], acc[u & 3], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    stamps[0] = c1 - c0;
    stamps[1] = w1 - w0;
  }
}

template <bool RANDOM>
void run(const char* what);

int main() {
  run<false>("constant operands");
  run<true>("pseudo-random operands");
  return 0;
}

template <bool RANDOM>
void run(const char* what) {
  printf("== %s\n", what);
  float* out;
  long long* stamps;
  hipMalloc(&out, 512 * 256 * 4);
  hipMalloc(&stamps, 16);
  const int iters = 500000;  // 4e6 MFMAs per wave, 2 waves per SIMD: ~5.1e8 cycles ~ 0.25 s
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double t_total = 0;
  for (int rep = 0; rep < 12; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<RANDOM>, dim3(512), dim3(256), 0, 0, out, stamps, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long st[2];
    hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost);
    const double flops = 512.0 * 4 * iters * 8.0 * (2.0 * 32 * 32 * 2);
    t_total += ms;
    printf("t=%6.2f s  %7.1f ms  %6.1f TFLOP/s  shader clock %7.1f MHz (cycles/MFMA/SIMD %.1f)\n", t_total * 1e-3, ms,
           flops / (ms * 1e-3) / 1e12, (double)st[0] / ((double)st[1] / 100.0), (double)st[0] / (iters * 8.0 * 2));
  }
  hipFree(out);
  hipFree(stamps);
}
