// What does hipLaunchCooperativeKernel cost next to a plain launch?  (the fused residual + LayerNorm GEMM
// needs its 248 tiles co-resident; a cooperative launch would guarantee that and delete the stand-by LayerNorm launch.)
// Alternates a 248-block, 768-thread, 156-KiB-LDS kernel that runs ~20 us with a short plain kernel, both ways.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/coop_probe tools/coop_probe.hip && gpurun_out/coop_probe
// Tuning tool, not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(768) void big_kernel(float* out, int spin) {
  extern __shared__ float lds[];
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  lds[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[1];
}
__global__ void small_kernel(float* out) { out[blockIdx.x * blockDim.x + threadIdx.x] += 1.f; }

int main() {
  float* buf;
  CK(hipMalloc(&buf, 1 << 20));
  CK(hipMemset(buf, 0, 1 << 20));
  const size_t shmem = 156 * 1024;
  CK(hipFuncSetAttribute((const void*)big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  int spin = 6000;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      const int n = 200;
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < n; ++i) {
        if (mode == 0) {
          hipLaunchKernelGGL(big_kernel, dim3(248), dim3(768), shmem, s, buf, spin);
        } else {
          void* args[] = {(void*)&buf, (void*)&spin};
          CK(hipLaunchCooperativeKernel((const void*)big_kernel, dim3(248), dim3(768), args, (unsigned)shmem, s));
        }
        hipLaunchKernelGGL(small_kernel, dim3(256), dim3(256), 0, s, buf);
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s launch + small kernel: %.2f us per pair\n", mode ? "cooperative" : "plain      ", ms * 1000.f / n);
    }
  }
  return 0;
}
