// Probe (round 6): `buffer_load_dwordx3 ... lds` (12 B per lane LDS-DMA, gfx950) and LDS destination alignment.
//   Q1: does size 12 work through __builtin_amdgcn_raw_ptr_buffer_load_lds, with a 4-byte-aligned global source?
//   Q2: is the LDS image lane-linear at 12 B per lane (768 B per wave instruction)?
//   Q3: may the LDS base of a 12-B / 16-B DMA be only 4-byte aligned (base + 4, + 8)?
//   Q4: out-of-range lanes (offset past num_records) land as 0.0 per dword?
//   Q5: bank conflicts of ds_read_b32 over rows laid out with an odd number of 12-B granules (timed loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int BYTES>
__global__ void k(const float* __restrict__ x, float* y, int nfloats, int shift, int lds_off) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) smem[i] = -777.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nfloats * 4, 0x00020000);
  constexpr int FL = BYTES / 4;
  int voff = (lane * FL + shift) * 4;  // lane i -> floats [FL*i + shift, ...)
  if (lane >= 60) voff = 0x7FFFFFF0;   // out of range lanes
  if (threadIdx.x < 64) {
    if constexpr (BYTES == 12) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + lds_off), 12, voff, 0, 0, 0);
    else if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + lds_off), 16, voff, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + lds_off), 4, voff, 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) y[i] = smem[i];
}
// timed: 32 lanes read rows r = lane & 31 at a common column from a tile whose row stride is `rs` floats
__global__ void bank(float* out, int rs, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 64 * 40; i += blockDim.x) smem[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float* p = smem + (lane & 31) * rs + (lane >> 5);
  float acc = 0.f;
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) acc += p[2 * c];
    asm volatile("" ::: "memory");
  }
  long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 16);
}
template <int BYTES>
static void run(const float* dx, float* dy, int n) {
  constexpr int FL = BYTES / 4;
  for (int lds_off = 0; lds_off < 4; ++lds_off)
    for (int shift = 0; shift < 2; ++shift) {
      hipLaunchKernelGGL(k<BYTES>, dim3(1), dim3(256), 8192, 0, dx, dy, n, shift, lds_off);
      std::vector<float> hy(512);
      if (hipMemcpy(hy.data(), dy, 512 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return; }
      int good = 0, bad = 0, oob0 = 0, oobx = 0, untouched_bad = 0;
      for (int t = 0; t < 512; ++t) {
        const int rel = t - lds_off;
        if (rel < 0 || rel >= 64 * FL) { if (hy[t] != -777.f) ++untouched_bad; continue; }
        const int lane = rel / FL;
        if (lane >= 60) { if (hy[t] == 0.f) ++oob0; else ++oobx; continue; }
        if (hy[t] == (float)(rel + shift + 1)) ++good; else ++bad;
      }
      printf("size %2d B  lds base +%d floats  src shift %d: lane-linear ok %d wrong %d | OOB lanes zero %d nonzero %d | bytes outside touched %d  first: %g %g %g %g %g\n",
             BYTES, lds_off, shift, good, bad, oob0, oobx, untouched_bad, hy[0], hy[1], hy[2], hy[3], hy[4]);
    }
}
int main() {
  const int n = 4096;
  std::vector<float> hx(n);
  for (int i = 0; i < n; ++i) hx[i] = i + 1;
  float *dx, *dy;
  hipMalloc(&dx, n * 4 + 64); hipMalloc(&dy, 65536);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  run<12>(dx, dy, n);
  run<16>(dx, dy, n);
  run<4>(dx, dy, n);
  for (int rs : {32, 33, 36, 39, 40, 35, 37}) {
    hipLaunchKernelGGL(bank, dim3(1), dim3(64), 64 * 40 * 4, 0, dy, rs, 2000);
    float t;
    hipMemcpy(&t, dy, 4, hipMemcpyDeviceToHost);
    printf("ds_read_b32, 32 rows x common column, row stride %d floats: %.2f cycles per read (one wave)\n", rs, t);
  }
  return 0;
}
