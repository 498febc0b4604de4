// What streaming rate can a kernel reach on this part?  LayerNorm backward moves 160 MB in 33 us (4.8 TB/s), the forward 60 MB in
// 14.5 us (4.2 TB/s); the guide says ~6.3 TB/s achievable.  Read / write / copy of fp32 rows at in-step sizes (40 - 160 MB, cold:
// the buffer set rotates through 2 GiB) with loads in flight per lane x blocks per CU swept.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/bw_probe tools/bw_probe.hip && gpurun_out/bw_probe
// Tuning tool, not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int U, int MODE>   // MODE 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16, uint32_t* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (MODE != 1) v[u] = j < n16 ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      else v[u] = make_float4((float)j, 1.f, 2.f, 3.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (MODE == 0) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      else if (j < n16) dst[j] = v[u];
    }
  }
  if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = 1;
}

template <int U, int MODE>
float run(const char* buf, char* out, size_t bytes, int blocks, uint32_t* sink, size_t pool) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  size_t off = 0;
  for (int rep = 0; rep < 6; ++rep) {
    off = (off + bytes + (64 << 20)) % (pool - bytes);   // a fresh (cold) piece of the pool every time
    off &= ~(size_t)255;
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<U, MODE>), dim3(blocks), dim3(256), 0, 0, (const float4*)(buf + off), (float4*)(out + off), bytes / 16, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t pool = 3ull << 30;
  char *a, *b; uint32_t* sink;
  CK(hipMalloc(&a, pool)); CK(hipMalloc(&b, pool)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, pool)); CK(hipMemset(b, 0, pool));
  CK(hipDeviceSynchronize());
  const size_t sizes[] = {40ull << 20, 80ull << 20, 160ull << 20, 1024ull << 20};
  const int bpc[] = {2, 4, 8, 16};
  printf("%-6s %-9s %-7s %10s %10s %10s   (TB/s; copy counts read + write)\n", "mode", "MiB", "blk/CU", "U=1", "U=2", "U=4");
  for (int mode = 0; mode < 3; ++mode)
    for (size_t sz : sizes)
      for (int k : bpc) {
        const int blocks = 256 * k;
        float t1, t2, t4;
        if (mode == 0) { t1 = run<1, 0>(a, b, sz, blocks, sink, pool); t2 = run<2, 0>(a, b, sz, blocks, sink, pool); t4 = run<4, 0>(a, b, sz, blocks, sink, pool); }
        else if (mode == 1) { t1 = run<1, 1>(a, b, sz, blocks, sink, pool); t2 = run<2, 1>(a, b, sz, blocks, sink, pool); t4 = run<4, 1>(a, b, sz, blocks, sink, pool); }
        else { t1 = run<1, 2>(a, b, sz, blocks, sink, pool); t2 = run<2, 2>(a, b, sz, blocks, sink, pool); t4 = run<4, 2>(a, b, sz, blocks, sink, pool); }
        const double f = (mode == 2 ? 2.0 : 1.0) * sz / 1e9;
        printf("%-6s %-9zu %-7d %10.2f %10.2f %10.2f\n", mode == 0 ? "read" : (mode == 1 ? "write" : "copy"), sz >> 20, k, f / t1, f / t2, f / t4);
      }
  return 0;
}
