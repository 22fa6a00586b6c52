// graph_memset.hip -- does a hipMemsetAsync captured into a hipGraph (a memset NODE) always take effect, in order,
// between the kernel nodes around it?  Mirrors the round-2 STFT-loss forward: memset(partial) -> kernel that
// writes all but the last two slots -> one-workgroup finish kernel that sums ALL slots -> (later in the same
// graph) a kernel that reuses the buffer for something else.  Replayed many times; a device-side counter
// records every replay whose sum is not the expected value.
//   hipcc --offload-arch=gfx950 -O2 -o graph_memset.bin graph_memset.hip && ./graph_memset.bin [replays] [pad_kernels]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void write_kernel(float* partial, int units) {  // slot u < units: 1.0 in column 0..2
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= units) return;
  if ((threadIdx.x & 63) == 0) {
    partial[u * 4 + 0] = 1.f; partial[u * 4 + 1] = 2.f; partial[u * 4 + 2] = 3.f; partial[u * 4 + 3] = 0.f;
  }
}
__global__ void finish_kernel(const float* partial, int slots, float* sums) {
  __shared__ float red[3][4];
  float s[3] = {0.f, 0.f, 0.f};
  for (int u = threadIdx.x; u < slots; u += 256) { s[0] += partial[u * 4]; s[1] += partial[u * 4 + 1]; s[2] += partial[u * 4 + 2]; }
  for (int j = 0; j < 3; ++j) {
    for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_down(s[j], o, 64);
    if ((threadIdx.x & 63) == 0) red[j][threadIdx.x >> 6] = s[j];
  }
  __syncthreads();
  if (threadIdx.x < 3) sums[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ void check_kernel(const float* sums, int units, unsigned* bad, float* first_bad) {
  if (threadIdx.x == 0) {
    const bool ok = sums[0] == 1.f * units && sums[1] == 2.f * units && sums[2] == 3.f * units;
    if (!ok) { if (atomicAdd(bad, 1u) == 0) { first_bad[0] = sums[0]; first_bad[1] = sums[1]; first_bad[2] = sums[2]; } }
  }
}
__global__ void garbage_kernel(float* p, long n, float v) {  // the buffer's "next tenant" within the same graph
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) p[i] = v * (float)(i + 1);
}
__global__ void busy_kernel(float* p, long n) {  // unrelated work before / after (timing noise)
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) p[i] = p[i] * 1.0001f + 1.f;
}

int main(int argc, char** argv) {
  const int replays = argc > 1 ? atoi(argv[1]) : 20000;
  const int pad = argc > 2 ? atoi(argv[2]) : 4;
  const int units = 714, blocks = (units + 3) / 4, slots = blocks * 4;  // C2, n_fft 1024: 714 units -> 716 slots
  float *partial, *sums, *first_bad, *noise;
  unsigned* bad;
  const long noise_n = 8L << 20;
  CK(hipMalloc(&partial, slots * 4 * sizeof(float)));
  CK(hipMalloc(&sums, 3 * sizeof(float)));
  CK(hipMalloc(&first_bad, 3 * sizeof(float)));
  CK(hipMalloc(&bad, sizeof(unsigned)));
  CK(hipMalloc(&noise, noise_n * sizeof(float)));
  CK(hipMemset(bad, 0, sizeof(unsigned)));
  CK(hipMemset(noise, 0, noise_n * sizeof(float)));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  float* pinned;
  CK(hipHostMalloc(&pinned, slots * 4 * sizeof(float), 0));
  for (int i = 0; i < slots * 4; ++i) pinned[i] = -1.0e26f * (float)(i + 1);
  // 0: memset node, next tenant written by a kernel; 1: zero-fill kernel instead of the memset node;
  // 2: memset node, next tenant written by an H2D memcpy NODE from pinned memory (the optimizer / clip chunk tables
  //    of a captured training step are uploaded like that); 3: zero-fill kernel + H2D memcpy node
  for (int variant = 0; variant < 4; ++variant) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < pad; ++i) hipLaunchKernelGGL(busy_kernel, dim3(2048), dim3(256), 0, s, noise, noise_n);
    if ((variant & 1) == 0) CK(hipMemsetAsync(partial, 0, (size_t)blocks * 16 * sizeof(float), s));
    else hipLaunchKernelGGL(garbage_kernel, dim3(4), dim3(256), 0, s, partial, (long)slots * 4, 0.f);
    hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(256), 0, s, partial, units);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, s, (const float*)partial, slots, sums);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, s, (const float*)sums, units, bad, first_bad);
    for (int i = 0; i < pad; ++i) hipLaunchKernelGGL(busy_kernel, dim3(2048), dim3(256), 0, s, noise, noise_n);
    if (variant < 2) hipLaunchKernelGGL(garbage_kernel, dim3(4), dim3(256), 0, s, partial, (long)slots * 4, -1.0e24f);
    else CK(hipMemcpyAsync(partial, pinned, slots * 4 * sizeof(float), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(256), 0, s, partial, (long)slots * 4);  // (a reader of the next tenant)
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipMemset(bad, 0, sizeof(unsigned)));
    for (int r = 0; r < replays; ++r) {
      CK(hipGraphLaunch(ge, s));
      if ((r & 1023) == 1023) CK(hipStreamSynchronize(s));  // sometimes the host is ahead, sometimes not
    }
    CK(hipStreamSynchronize(s));
    unsigned nbad;
    float fb[3];
    CK(hipMemcpy(&nbad, bad, sizeof(unsigned), hipMemcpyDeviceToHost));
    CK(hipMemcpy(fb, first_bad, sizeof(fb), hipMemcpyDeviceToHost));
    static const char* names[4] = {"memset node, kernel tenant", "zero-fill kernel, kernel tenant", "memset node, H2D-memcpy-node tenant",
                                   "zero-fill kernel, H2D-memcpy-node tenant"};
    printf("variant %d (%s): %u bad of %d replays", variant, names[variant], nbad, replays);
    if (nbad) printf("  first bad sums = %g %g %g (expected %d %d %d)", fb[0], fb[1], fb[2], units, 2 * units, 3 * units);
    printf("\n");
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
