// Probe: `buffer_load_dwordx4 ... lds` (16 B per lane, LDS-DMA) with a global address that is only
// 4-byte aligned, and with lanes that are partly / wholly out of range of the raw buffer.
//   Q1: is a dword-aligned (not 16-B aligned) source address legal and correct?
//   Q2: range check per dword or per 16-B access?  (what lands in LDS for a lane straddling the end)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ x, float* y, int rowlen, int shift) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 512; i += blockDim.x) smem[i] = -777.f;
  __syncthreads();
  const float* row = x + (long)blockIdx.x * rowlen;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, rowlen * 4, 0x00020000);
  int voff = (lane * 4 - shift) * 4;  // lane i -> floats [4i - shift, 4i - shift + 4)
  if (threadIdx.x < 64)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  y[blockIdx.x * 256 + threadIdx.x] = smem[threadIdx.x];
}
int main() {
  const int rowlen = 203, rows = 2;
  std::vector<float> hx(rows * rowlen);
  for (int i = 0; i < rows * rowlen; ++i) hx[i] = 1000.f * (i / rowlen) + (i % rowlen) + 1;
  float *dx, *dy;
  hipMalloc(&dx, hx.size() * 4 + 64); hipMalloc(&dy, rows * 256 * 4);
  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 7; ++shift) {
    hipLaunchKernelGGL(k, dim3(rows), dim3(256), 2048, 0, dx, dy, rowlen, shift);
    std::vector<float> hy(rows * 256);
    if (hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("shift %d: launch failed\n", shift); return 1; }
    int bad_in = 0, zero_oob = 0, other_oob = 0, partial_lane_zeroed = 0;
    for (int r = 0; r < rows; ++r)
      for (int t = 0; t < 256; ++t) {
        int f = t - shift;
        bool in = f >= 0 && f < rowlen;
        float got = hy[r * 256 + t];
        if (in) {
          if (got != 1000.f * r + f + 1) {
            ++bad_in;
            int lane = t / 4;  // does this lane straddle a boundary?
            int f0 = lane * 4 - shift;
            if ((f0 < 0 || f0 + 3 >= rowlen) && got == 0.f) ++partial_lane_zeroed;
          }
        } else {
          if (got == 0.f) ++zero_oob; else ++other_oob;
        }
      }
    printf("shift %d (src %s16B-aligned): in-range wrong %d (of which straddling lanes zeroed: %d), OOB dwords zero %d / nonzero %d\n",
           shift, shift % 4 ? "NOT " : "", bad_in, partial_lane_zeroed, zero_oob, other_oob);
  }
  return 0;
}
