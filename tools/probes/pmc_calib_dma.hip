// pmc_calib_dma.hip -- known-byte-count streams in the access patterns of the product kernels, for calibrating
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 for wide streaming reads;
// "calibrate on a known byte count in your own access pattern"):
//   dma4   : buffer_load_dword   ... lds (4 B per lane)   + 4-B  global stores
//   dma16  : buffer_load_dwordx4 ... lds (16 B per lane)  + 16-B global stores
//   gld16  : global_load_dwordx4 (16 B per lane, to VGPRs) + 16-B global stores
// Each kernel copies `n` floats once: reads n * 4 bytes, writes n * 4 bytes.
//   hipcc --offload-arch=gfx950 -O2 -o pmc_calib_dma.bin pmc_calib_dma.hip && ./pmc_calib_dma.bin [MiB] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// one workgroup copies 16 KiB per iteration through LDS
template <int W>
__global__ __launch_bounds__(256) void dma_copy(const float* src, float* dst, long n) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (long base = (long)blockIdx.x * 4096; base < n; base += (long)gridDim.x * 4096) {
    __amdgpu_buffer_rsrc_t rs = rsrc(src + base, 4096 * 4);
    if (W == 16) {
      for (int p = wave; p < 16; p += 4)  // 16 pieces of 1 KiB
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + p * 256), 16, (unsigned)(p * 256 + lane * 4) * 4u, 0, 0, 0);
    } else {
      for (int p = wave; p < 64; p += 4)  // 64 pieces of 256 B
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + p * 64), 4, (unsigned)(p * 64 + lane) * 4u, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (W == 16) {
      for (int i = threadIdx.x; i < 1024; i += 256) reinterpret_cast<float4*>(dst + base)[i] = reinterpret_cast<const float4*>(lds)[i];
    } else {
      for (int i = threadIdx.x; i < 4096; i += 256) dst[base + i] = lds[i];
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void gld16_copy(const float4* src, float4* dst, long n4) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) dst[i] = src[i];
}

int main(int argc, char** argv) {
  const long mib = argc > 1 ? atol(argv[1]) : 1024;
  const int reps = argc > 2 ? atoi(argv[2]) : 3;
  const long n = mib * 262144L;
  float *a, *b;
  CK(hipMalloc(&a, n * 4));
  CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 1, n * 4));
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(dma_copy<4>, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(dma_copy<16>, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(gld16_copy, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4);
  }
  CK(hipDeviceSynchronize());
  printf("copied %ld MiB x %d per kernel (reads = writes = %ld bytes per launch)\n", mib, reps, n * 4);
  return 0;
}
