// Microbenchmark: how fast can a CU pull GEMM-stage-shaped data (rows of 128 B, 8 lanes per row) out of the
// cache hierarchy, by path:  (1) global_load_dwordx4 -> VGPR,  (2) global_load_lds_dwordx4 (LDS-DMA),
// (3) global_load_dwordx4 -> VGPR -> ds_write_b128.   Reported as bytes / clock / CU at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/fill_bw tools/fill_bw.hip && gpurun_out/fill_bw
// Tuning tool, not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ROWS = 288;          // rows of one stage (160 + 128)
constexpr int NI = ROWS / 8 / 4;   // wave-instructions per wave per stage (8 rows per instruction, 4 waves) = 9

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// mode 0: VGPR loads; 1: LDS-DMA; 2: VGPR loads + ds_write
template <int MODE>
__global__ __launch_bounds__(256) void fill_kernel(const char* __restrict__ src, int ld, int ksteps, int iters,
                                                   int blocks_share, uint32_t* __restrict__ sink, int pattern) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 x 36 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (size_t)(blocks_share ? 0 : blockIdx.x) * ROWS * ld;
  uint32_t off[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = wave * NI + i;
    // pattern 0: 8 lanes per 128-B row (stage copy); pattern 1: MFMA-fragment order, lane -> row lane%32, two 16-B chunks
    off[i] = pattern == 0 ? (uint32_t)((q * 8 + (lane >> 3)) * ld + (lane & 7) * 16)
                          : (uint32_t)(((q >> 2) * 32 + (lane & 31)) * ld + ((q & 3) * 2 + (lane >> 5)) * 16);
  }
  uint4 acc = make_uint4(0, 0, 0, 0);
  int kt = 0, buf = 0;
  for (int it = 0; it < iters; ++it) {
    const char* p = base + (size_t)kt * 128;
    char* l = lds + buf * (ROWS * 128) + wave * NI * 1024;
    if (MODE == 1) {
      const uint32_t lb = __builtin_amdgcn_readfirstlane(lds_addr_of(l));
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        uint32_t keep;
        const char* g = p + off[i];
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lb + i * 1024) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");  // previous stage landed, this one in flight
    } else {
      uint4 v[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const uint4*>(p + off[i]);
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<uint4*>(l + i * 1024 + lane * 16) = v[i];
      } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
      }
    }
    kt = kt + 1 == ksteps ? 0 : kt + 1;
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 0) acc = *reinterpret_cast<const uint4*>(lds + threadIdx.x * 16);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
int run(const char* name, const char* src, int ld, int ksteps, int share, int nblocks, uint32_t* sink, int pattern = 0) {
  const int iters = 4000;
  const size_t shmem = 2 * ROWS * 128;
  CK(hipFuncSetAttribute((const void*)fill_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fill_kernel<MODE>, dim3(nblocks), dim3(256), shmem, 0, src, ld, ksteps, 200, share, sink, pattern);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(fill_kernel<MODE>, dim3(nblocks), dim3(256), shmem, 0, src, ld, ksteps, iters, share, sink, pattern);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)nblocks * iters * ROWS * 128;
  const int cus = nblocks < 256 ? nblocks : 256;
  printf("  %-22s %7.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", name, ms * 1e3, bytes / ms / 1e9,
         bytes / (ms * 1e-3) / 2.4e9 / cus);
  return 0;
}

int main() {
  const int ld = 2048;  // K = 1024 bf16
  const size_t total = (size_t)512 * ROWS * ld;
  char* src; uint32_t* sink;
  CK(hipMalloc(&src, total)); CK(hipMemset(src, 1, total)); CK(hipMalloc(&sink, 4));
  struct { const char* what; int ksteps, share, nblocks; } cfg[] = {
      {"L1-resident (one 36 KiB stage per block, re-read), 2 blocks/CU", 1, 0, 512},
      {"L2-resident (all blocks walk the same 576 KiB panel), 2 blocks/CU", 16, 1, 512},
      {"private panels (512 x 576 KiB = 302 MB, MALL/HBM), 2 blocks/CU", 16, 0, 512},
      {"L1-resident, 1 block/CU", 1, 0, 256},
      {"L2-resident, 1 block/CU", 16, 1, 256},
  };
  for (auto& c : cfg) {
    printf("%s\n", c.what);
    if (run<0>("global->VGPR", src, ld, c.ksteps, c.share, c.nblocks, sink)) return 1;
    if (run<1>("LDS-DMA", src, ld, c.ksteps, c.share, c.nblocks, sink)) return 1;
    if (run<2>("global->VGPR->ds_write", src, ld, c.ksteps, c.share, c.nblocks, sink)) return 1;
    if (run<0>("global->VGPR fragment", src, ld, c.ksteps, c.share, c.nblocks, sink, 1)) return 1;
  }
  return 0;
}
