// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, per access pattern the product
// kernels use ("calibrate FETCH_SIZE on the prefetch pattern and on the DMA stream separately").
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/pmc_calib tools/pmc_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -o p --output-format csv -- gpurun_out/pmc_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/w -o p --output-format csv -- gpurun_out/pmc_calib
//   python tools/pmc_calib.py out  ->  bytes the counter reports per byte the kernel really moved
// Every kernel walks its own 1 GiB region of a 6 GiB buffer once (no reuse: nothing can come from the 256 MiB Infinity Cache
// or from L2), so the true traffic is the region size (or, for the line-touch kernels, lines x 128 B at most).
// Tuning tool, not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr size_t REGION = 1ull << 30;

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// (a) wide coalesced streaming read: 16 B per lane, consecutive lanes consecutive 16-B chunks
__global__ __launch_bounds__(256) void calib_read16(const uint4* __restrict__ src, size_t n16, uint32_t* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

// (b) the GEMM kernels' global -> LDS DMA: global_load_lds_dwordx4, 8 lanes per 128-B row (a stage copy)
__global__ __launch_bounds__(256) void calib_dma16(const char* __restrict__ src, size_t bytes, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) char lds[256 * 16];
  const uint32_t l0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds) + (threadIdx.x >> 6) * 1024);
  const size_t per_block_iter = 256 * 16;
  for (size_t off = (size_t)blockIdx.x * per_block_iter; off < bytes; off += (size_t)gridDim.x * per_block_iter) {
    const char* p = src + off + (size_t)threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off\n\t" ::"s"(l0), "v"(p) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (*reinterpret_cast<volatile uint32_t*>(lds) == 0x12345678u) *sink = 1;
}

// (c) the consumer waves' L2 prefetch: ONE global_load_dword per 128-byte line, result unused
__global__ __launch_bounds__(256) void calib_touch_line(const char* __restrict__ src, size_t lines, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < lines; i += (size_t)gridDim.x * blockDim.x)
    acc ^= *reinterpret_cast<const uint32_t*>(src + i * 128);
  if (acc == 0x12345678u) *sink = 1;
}
// (c2) the same touch followed, in a second launch, by the full DMA read of the same lines: is the line fetched twice?
//      (calib_touch_line on region X, then calib_dma16 on region X; the region is 64 MiB so that it stays in the Infinity
//      Cache / partly in L2 between the two launches, as a stage's lines do between the prefetch and the DMA four stages later)

// (d) one dword per 64-byte half line (does a touch fetch 64 or 128 bytes?)
__global__ __launch_bounds__(256) void calib_touch_half(const char* __restrict__ src, size_t halves, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < halves; i += (size_t)gridDim.x * blockDim.x)
    acc ^= *reinterpret_cast<const uint32_t*>(src + i * 64);
  if (acc == 0x12345678u) *sink = 1;
}

// (e) writes: 16 B per lane plain stores / nontemporal stores / 4 B per lane stores
__global__ __launch_bounds__(256) void calib_write16(uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void calib_write16_nt(uint4* __restrict__ dst, size_t n16) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    u4 v = {(uint32_t)i, 1, 2, 3};
    __builtin_nontemporal_store(v, reinterpret_cast<u4*>(dst) + i);
  }
}
__global__ __launch_bounds__(256) void calib_write4(uint32_t* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint32_t)i;
}
// (f) 8-byte atomic-free read-modify-write of fp32 (the "+=" epilogues): 16 B read + 16 B write per lane
__global__ __launch_bounds__(256) void calib_rmw16(float4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = dst[i];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    dst[i] = v;
  }
}

int main() {
  char* buf = nullptr;
  uint32_t* sink = nullptr;
  const size_t total = 8 * REGION;
  CK(hipMalloc(&buf, total));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 1, total));
  CK(hipDeviceSynchronize());
  const dim3 grid(256 * 8), block(256);
  size_t r = 0;
  hipLaunchKernelGGL(calib_read16, grid, block, 0, 0, (const uint4*)(buf + r * REGION), REGION / 16, sink); r++;
  hipLaunchKernelGGL(calib_dma16, grid, block, 0, 0, buf + r * REGION, REGION, sink); r++;
  hipLaunchKernelGGL(calib_touch_line, grid, block, 0, 0, buf + r * REGION, REGION / 128, sink); r++;
  hipLaunchKernelGGL(calib_touch_half, grid, block, 0, 0, buf + r * REGION, REGION / 64, sink); r++;
  hipLaunchKernelGGL(calib_write16, grid, block, 0, 0, (uint4*)(buf + r * REGION), REGION / 16); r++;
  hipLaunchKernelGGL(calib_write16_nt, grid, block, 0, 0, (uint4*)(buf + r * REGION), REGION / 16); r++;
  hipLaunchKernelGGL(calib_write4, grid, block, 0, 0, (uint32_t*)(buf + r * REGION), REGION / 4); r++;
  hipLaunchKernelGGL(calib_rmw16, grid, block, 0, 0, (float4*)(buf + r * REGION), REGION / 16); r++;
  CK(hipDeviceSynchronize());
  // (c2) touch then DMA of the SAME 64 MiB, eight times over fresh 64-MiB pieces of region 2's neighbourhood (region 0: cold
  // again after 7 GiB of other traffic): the pair's FETCH_SIZE against 64 MiB tells whether the touched line is fetched again
  const size_t piece = 64ull << 20;
  for (int i = 0; i < 8; ++i) {
    hipLaunchKernelGGL(calib_touch_line, grid, block, 0, 0, buf + i * piece, piece / 128, sink);
    hipLaunchKernelGGL(calib_dma16, grid, block, 0, 0, buf + i * piece, piece, sink);
  }
  CK(hipDeviceSynchronize());
  printf("pmc_calib: done (regions of %zu MiB, pairs of %zu MiB)\n", REGION >> 20, piece >> 20);
  return 0;
}
