// Probe: what does the reduction loop of csrc/resunit.hip cost beside its MFMAs?  The loop (copied from
// resunit_contract, C = 32: per tap 16 channel pairs x 2 column tiles) runs back to back on every CU with no
// prologue / epilogue / barrier, in variants that drop one ingredient each:
//   full      : A operands from global/L2 (prefetched one tap ahead), B operands from LDS, LeakyReLU on B
//   -act      : no LeakyReLU
//   -act -A   : weights loaded once
//   -act -A -B: operands loaded once (pure MFMA issue in the same control flow)
// for 1 / 2 / 3 workgroups (of 4 waves) per CU.  Prints chip-wide TFLOP/s.
// build: hipcc --offload-arch=gfx950 -O3 -w -fno-honor-nans -mno-amdgpu-ieee -o mfma_loop.bin mfma_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// AMODE: 0 = A from global, dword per channel pair (the compiler waits per use); 1 = same + ONE s_waitcnt per tap;
// 2 = A from LDS (wl is an LDS pointer); 3 = A from global, dwordx4 per 4 channel pairs (image [tap][cp/4][lane][4])
template <int RS, bool ACT, int AMODE = 0>
__device__ __forceinline__ void contract(const float* __restrict__ wl, const float* bl, int tap_step, int k, float slope,
                                         f32x16 (&acc)[2], bool reload_a, bool reload_b) {
  constexpr int CP = 16, G = 8, NG = CP / G, TAP_W = CP * 64;
  float A0[CP], A1[CP], B0[G][2], B1[G][2];
  auto load_a = [&](float(&A)[CP], int tap) {
    const float* p = wl + (long)tap * TAP_W;
    if (AMODE == 3) {
      const float4* p4 = reinterpret_cast<const float4*>(wl + (long)tap * TAP_W);  // wl = base + lane * 4
#pragma unroll
      for (int q = 0; q < CP / 4; ++q) {
        const float4 v = p4[q * 64];
        A[4 * q] = v.x; A[4 * q + 1] = v.y; A[4 * q + 2] = v.z; A[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int cp = 0; cp < CP; ++cp) A[cp] = p[cp * 64];
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
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        float v = B[i][ni];
        if (ACT) v = __builtin_fmaxf(v, v * slope);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i], v, acc[ni], 0, 0, 0);
      }
  };
  auto tap_body = [&](float(&A)[CP], int tap) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g & 1) {
        if (reload_b) { if (g + 1 < NG) load_b(B0, tap, g + 1); else load_b(B0, tap + 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        mma(&A[g * G], B1);
      } else {
        if (reload_b) { if (g + 1 < NG) load_b(B1, tap, g + 1); else load_b(B1, tap + 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        mma(&A[g * G], B0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  load_a(A0, 0);
  load_a(A1, 1);
  load_b(B0, 0, 0);
  load_b(B1, 0, 1);
  int tap = 0;
  for (; tap + 2 <= k; tap += 2) {
    if (reload_a) load_a(A1, tap + 1);
    if (AMODE == 1) __builtin_amdgcn_s_waitcnt(0x4F70);  // vmcnt(16): everything but the 16 loads just issued
    __builtin_amdgcn_sched_barrier(0);
    tap_body(A0, tap);
    if (reload_a) load_a(A0, tap + 2 < k ? tap + 2 : k - 1);
    if (AMODE == 1) __builtin_amdgcn_s_waitcnt(0x4F70);
    __builtin_amdgcn_sched_barrier(0);
    tap_body(A1, tap + 1);
  }
  if (tap < k) tap_body(A0, tap);
}

template <int MODE, int WPC>
__global__ __launch_bounds__(256, WPC) void k(const float* w, float* out, int reps, int taps) {
  constexpr int XS = 320;
  extern __shared__ float xs[];
  unsigned h = threadIdx.x * 2654435761u + 99u;
  for (int i = threadIdx.x; i < 32 * XS; i += 256) {
    h = h * 1664525u + 1013904223u;
    xs[i] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  f32x16 acc[2];
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float* wl = w + lane;
  const float* bl = xs + lhi * XS + wave * 64 + l31;
  float* ws = xs + 32 * XS;  // MODE 5: the weights staged in LDS
  if (MODE == 5) {
    for (int i = threadIdx.x; i < (taps + 1) * 16 * 64; i += 256) ws[i] = w[i];
    __syncthreads();
  }
  for (int rep = 0; rep < reps; ++rep) {
    if (MODE == 0) contract<XS, true>(wl, bl, 1, taps, 0.1f, acc, true, true);
    if (MODE == 1) contract<XS, false>(wl, bl, 1, taps, 0.1f, acc, true, true);
    if (MODE == 2) contract<XS, false>(wl, bl, 1, taps, 0.1f, acc, false, true);
    if (MODE == 3) contract<XS, false>(wl, bl, 1, taps, 0.1f, acc, false, false);
    if (MODE == 4) contract<XS, false, 1>(wl, bl, 1, taps, 0.1f, acc, true, true);
    if (MODE == 5) contract<XS, false, 2>(ws + lane, bl, 1, taps, 0.1f, acc, true, true);
    if (MODE == 6) contract<XS, false, 3>(w + lane * 4, bl, 1, taps, 0.1f, acc, true, true);
    if (MODE == 7) contract<XS, true, 3>(w + lane * 4, bl, 1, taps, 0.1f, acc, true, true);
    if (MODE == 8) contract<XS, true, 1>(wl, bl, 1, taps, 0.1f, acc, true, true);
  }
  float s = 0.f;
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int WPC>
void run(const char* name, const float* w, float* out, int taps) {
  const int reps = 4400 / taps, blocks = 256 * WPC;
  const size_t lds = 32 * 320 * 4 + (MODE == 5 ? (taps + 1) * 16 * 64 * 4 : 0);
  if (lds * WPC > 160 * 1024) return;
  void (*kern)(const float*, float*, int, int) = k<MODE, WPC>;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, w, out, reps, taps);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * reps * taps * 16 * 2 * (2.0 * 32 * 32 * 2);
  printf("taps %2d  %-16s %d WG/CU: %7.2f ms  %6.1f TFLOP/s\n", taps, name, WPC, best, flops / (best * 1e-3) / 1e12);
}

int main() {
  float *w, *out;
  std::vector<float> hw(12 * 16 * 64);
  unsigned h = 7u;
  for (auto& v : hw) {
    h = h * 1664525u + 1013904223u;
    v = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 26));
  }
  (void)hipMalloc(&w, hw.size() * 4);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 768 * 256 * 4);
#define ALL(WPC, T)                               \
  run<0, WPC>("full", w, out, T);                 \
  run<8, WPC>("full 1wait", w, out, T);           \
  run<7, WPC>("full Ax4", w, out, T);             \
  run<1, WPC>("-act", w, out, T);                 \
  run<4, WPC>("-act 1wait", w, out, T);           \
  run<6, WPC>("-act Ax4", w, out, T);             \
  run<5, WPC>("-act A-LDS", w, out, T);           \
  run<2, WPC>("-act -A", w, out, T);              \
  run<3, WPC>("-act -A -B", w, out, T);
  ALL(1, 11) ALL(2, 11) ALL(3, 11) ALL(2, 5) ALL(3, 5)
  return 0;
}
