// Probe: does `buffer_load_dword ... lds` write 0 into LDS for out-of-range lanes (raw buffer, stride 0)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ x, float* y, int rowlen, int shift) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int lane = threadIdx.x & 63;
  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 512; i += blockDim.x) smem[i] = -777.f;
  __syncthreads();
  const float* row = x + (long)blockIdx.x * rowlen;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, rowlen * 4, 0x00020000);
  int voff = (lane + wave * 64 - shift) * 4;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 64), 4, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  y[blockIdx.x * 256 + threadIdx.x] = smem[threadIdx.x];
}
int main() {
  const int rowlen = 200, rows = 3, shift = 17;
  std::vector<float> hx(rows * rowlen);
  for (int i = 0; i < rows * rowlen; ++i) hx[i] = 1000.f * (i / rowlen) + (i % rowlen) + 1;
  float *dx, *dy;
  hipMalloc(&dx, hx.size() * 4); hipMalloc(&dy, rows * 256 * 4);
  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(rows), dim3(256), 2048, 0, dx, dy, rowlen, shift);
  std::vector<float> hy(rows * 256);
  hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < rows; ++r)
    for (int t = 0; t < 256; ++t) {
      int f = t - shift;
      float want = (f >= 0 && f < rowlen) ? 1000.f * r + f + 1 : 0.f;
      if (hy[r * 256 + t] != want) { if (bad < 10) printf("row %d t %d got %f want %f\n", r, t, hy[r*256+t], want); ++bad; }
    }
  printf("glds_oob: %s (%d mismatches)\n", bad ? "FAIL" : "PASS: OOB lanes land as 0.0 in LDS", bad);
  return bad != 0;
}
