#include <hip/hip_runtime.h>
#include <stdint.h>
struct P4 { uint32_t x, y, z, w; };
__device__ __forceinline__ P4 philA(uint64_t seed, uint32_t site, uint64_t ctr) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = site, c3 = 0x7149u;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return P4{c0, c1, c2, c3};
}
__device__ __forceinline__ P4 philB(uint64_t seed, uint32_t site, uint64_t ctr) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = site, c3 = 0x7149u;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const uint64_t q0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    const uint64_t q1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    uint32_t hi0 = (uint32_t)(q0 >> 32), lo0 = (uint32_t)q0, hi1 = (uint32_t)(q1 >> 32), lo1 = (uint32_t)q1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return P4{c0, c1, c2, c3};
}

template <int V>
__global__ void loopk(uint32_t* o, uint64_t seed, int iters) {
  uint64_t ctr = threadIdx.x + blockIdx.x * 256ull;
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    P4 r = V ? philB(seed, 3, ctr) : philA(seed, 3, ctr);
    acc ^= r.x ^ r.y ^ r.z ^ r.w;
    ctr += 977 + (acc & 1);
  }
  o[threadIdx.x + blockIdx.x * 256] = acc;
}
#include <cstdio>
int main() {
  uint32_t* o; hipMalloc(&o, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 2; ++v) for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    if (v) hipLaunchKernelGGL(loopk<1>, dim3(4096), dim3(256), 0, 0, o, 12345ull, 256);
    else hipLaunchKernelGGL(loopk<0>, dim3(4096), dim3(256), 0, 0, o, 12345ull, 256);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d: %.3f ms for %.0f M philox -> %.2f G philox/s\n", v, ms, 4096.0 * 256 * 256 / 1e6, 4096.0 * 256 * 256 / ms / 1e6);
  }
  return 0;
}
