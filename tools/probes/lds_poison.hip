// Probe helper: fill the LDS of every CU with quiet-NaN bit patterns (to expose "0 * stale LDS" hazards).
#include <hip/hip_runtime.h>
__global__ void poison(float* sink) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4 - 64; i += blockDim.x) lds[i] = __int_as_float(0x7fc00000);
  __syncthreads();
  if (threadIdx.x == 0 && lds[17] == 1.0f) sink[0] = 1.f;
}
extern "C" int lds_poison(float* sink, void* stream) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(poison), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipLaunchKernelGGL(poison, dim3(1024), dim3(256), 160 * 1024 - 256, (hipStream_t)stream, sink);
  return (int)hipGetLastError();
}
