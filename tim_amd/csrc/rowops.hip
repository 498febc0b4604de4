// HBM-bound row kernels of the TIM path: casts/transposes of operand copies,
// LayerNorm forward/backward (nn.LayerNorm at transformers.py:98-99,108-110,
// encodings.py:25,145,152, tim.py:73), bias-gradient column sums, sequence
// assembly (encodings.py:190-250) and the K=2 first layer of the time MLP.
#include <stdlib.h>

#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// weights: fp32 master -> operand dtype, plain or transposed, zero padded
// ---------------------------------------------------------------------------
template <typename T>
__global__ void cast_weight_kernel(const float* __restrict__ src, int rows, int cols, T* __restrict__ dst,
                                   int ld) {
  const int r = blockIdx.y;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ld; c += gridDim.x * blockDim.x)
    dst[(size_t)r * ld + c] = OpT<T>::from_f(c < cols ? src[(size_t)r * cols + c] : 0.f);
}

// dst[c, r] = src[r, c] through a 32x33 LDS tile; dst has ld >= rows, zero padded.  A block walks a
// panel of TP_ROWS rows x 32 columns; with colsum != nullptr it also accumulates
// colsum[c] += sum_r src[r, c] (bias gradients) with one atomic per column per panel.
constexpr int TP_ROWS = 256;
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_kernel(const TS* __restrict__ src, int rows, int cols, int lds_,
                                                        TD* __restrict__ dst, int ld, float* __restrict__ colsum) {
  __shared__ float tile[32][33];
  __shared__ float part[8][32];
  const int c0 = blockIdx.x * 32, rp0 = blockIdx.y * TP_ROWS;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  float acc = 0.f;
  for (int r0 = rp0; r0 < rp0 + TP_ROWS && r0 < ld; r0 += 32) {
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, c = c0 + tx;
      const float v = (r < rows && c < cols) ? OpT<TS>::to_f(src[(size_t)r * lds_ + c]) : 0.f;
      tile[i][tx] = v;
      acc += v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;  // dst row = c, dst col = r
      if (c < cols && r < ld) dst[(size_t)c * ld + r] = OpT<TD>::from_f(tile[tx][i]);
    }
    __syncthreads();
  }
  if (colsum) {
    part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c0 + tx < cols) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += part[k][tx];
      atomicAdd(colsum + c0 + tx, t);
    }
  }
}

// fp32 master weight [rows, cols] -> operand copies W[rows, ldp] and W^T[cols, ldt] in one pass
template <typename T>
__global__ __launch_bounds__(256) void cast_weight_both_kernel(const float* __restrict__ src, int rows, int cols,
                                                               T* __restrict__ plain, int ldp, T* __restrict__ tr,
                                                               int ldt) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    const float v = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    tile[i][tx] = v;
    if (r < rows && c < ldp) plain[(size_t)r * ldp + c] = OpT<T>::from_f(v);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < ldt) tr[(size_t)c * ldt + r] = OpT<T>::from_f(tile[tx][i]);
  }
}

// The same for up to CAST_MAX weights in ONE launch (the per-step refresh of every operand copy after an optimizer
// step): 64 x 64 tiles, float4 loads, 8-byte bf16x4 stores in both orientations (the transposed copy goes through LDS).
constexpr int CAST_MAX = 32;
struct CastBatch {
  int n;
  int blk0[CAST_MAX + 1];  // first block of item i
  TimCastItem it[CAST_MAX];
};

template <typename T>
__global__ __launch_bounds__(256) void cast_weights_kernel(CastBatch cb) {
  __shared__ float tile[64][65];
  int w = 0;
  while (w + 1 < cb.n && (int)blockIdx.x >= cb.blk0[w + 1]) ++w;
  const TimCastItem it = cb.it[w];
  const int rows = it.rows, cols = it.cols, ldp = it.ldp, ldt = it.ldt;
  const int tiles_x = (max(ldp, cols) + 63) / 64;
  const int lb = blockIdx.x - cb.blk0[w];
  const int c0 = (lb % tiles_x) * 64, r0 = (lb / tiles_x) * 64;
  const float* __restrict__ src = it.src;
  T* __restrict__ plain = (T*)it.plain;
  T* __restrict__ tr = (T*)it.tr;
  // 16-bit copies leave in 16-byte stores where alignment allows (round 5: 8 columns / 8 rows per lane; the leading dimensions
  // are multiples of 64 elements): a lane owns 8 consecutive columns of a row on the way in, 8 consecutive rows of a column on
  // the way out
  constexpr bool W16 = sizeof(T) == 2;
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;      // 8 column octets x 32 rows per pass
  const bool vec = (cols & 3) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + ty + 32 * i, c = c0 + tx * 8;
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = 0.f;
    if (r < rows) {
      if (vec && c + 7 < cols) {
        const float4 a = *reinterpret_cast<const float4*>(src + (size_t)r * cols + c);
        const float4 b = *reinterpret_cast<const float4*>(src + (size_t)r * cols + c + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (c + u < cols) v[u] = src[(size_t)r * cols + c + u];
      }
      if (c < ldp) {   // ldp % 64 == 0: whole octets
        if constexpr (W16) {
          typedef T v8_t __attribute__((ext_vector_type(8)));
          v8_t pk;
#pragma unroll
          for (int u = 0; u < 8; ++u) pk[u] = OpT<T>::from_f(v[u]);
          *reinterpret_cast<v8_t*>(plain + (size_t)r * ldp + c) = pk;
        } else {
          store4<T>(plain + (size_t)r * ldp + c, v[0], v[1], v[2], v[3]);
          store4<T>(plain + (size_t)r * ldp + c + 4, v[4], v[5], v[6], v[7]);
        }
      }
    }
    float* t = &tile[ty + 32 * i][tx * 8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = v[u];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lc = ty + 32 * i, lr = tx * 8;
    const int c = c0 + lc, r = r0 + lr;
    if (c < cols && r < ldt) {    // ldt % 64 == 0: whole octets
      if constexpr (W16) {
        typedef T v8_t __attribute__((ext_vector_type(8)));
        v8_t pk;
#pragma unroll
        for (int u = 0; u < 8; ++u) pk[u] = OpT<T>::from_f(tile[lr + u][lc]);
        *reinterpret_cast<v8_t*>(tr + (size_t)c * ldt + r) = pk;
      } else {
        store4<T>(tr + (size_t)c * ldt + r, tile[lr][lc], tile[lr + 1][lc], tile[lr + 2][lc], tile[lr + 3][lc]);
        store4<T>(tr + (size_t)c * ldt + r + 4, tile[lr + 4][lc], tile[lr + 5][lc], tile[lr + 6][lc], tile[lr + 7][lc]);
      }
    }
  }
}

// column sums of per-block partials: out[c] += sum_b part[b][c]   (deterministic second stage of LN backward)
__global__ void partial_colsum_kernel(const float* __restrict__ part, int nblk, int cols, float* __restrict__ o0,
                                      float* __restrict__ o1) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 2 * cols) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += part[(size_t)b * 2 * cols + c];
    s1 += part[(size_t)(b + 1) * 2 * cols + c];
    s2 += part[(size_t)(b + 2) * 2 * cols + c];
    s3 += part[(size_t)(b + 3) * 2 * cols + c];
  }
  for (; b < b1; ++b) s0 += part[(size_t)b * 2 * cols + c];
  float* o = c < cols ? o0 : o1;
  if (o) atomicAdd(o + (c < cols ? c : c - cols), (s0 + s1) + (s2 + s3));
}

// the same for several LayerNorm backward launches at once (blockIdx.z = set): one launch per training step instead of one per
// LayerNorm
constexpr int LNS_MAX = 16;
struct LnSets { float* dg[LNS_MAX]; float* db[LNS_MAX]; };
__global__ void partial_colsum_sets_kernel(const float* __restrict__ part, long long set_stride, int nblk, int cols, LnSets ls) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 2 * cols) return;
  const float* p = part + (long long)blockIdx.z * set_stride;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += p[(size_t)b * 2 * cols + c];
    s1 += p[(size_t)(b + 1) * 2 * cols + c];
    s2 += p[(size_t)(b + 2) * 2 * cols + c];
    s3 += p[(size_t)(b + 3) * 2 * cols + c];
  }
  for (; b < b1; ++b) s0 += p[(size_t)b * 2 * cols + c];
  float* o = c < cols ? ls.dg[blockIdx.z] : ls.db[blockIdx.z];
  if (o) atomicAdd(o + (c < cols ? c : c - cols), (s0 + s1) + (s2 + s3));
}

// dW[i] += sum_z slab[z][i]   (split-K weight-gradient partials)
__global__ void slab_reduce_kernel(const float* __restrict__ slab, long long n, int nslab, float* __restrict__ dW) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    float4 a = *reinterpret_cast<const float4*>(dW + i);
    for (int z = 0; z < nslab; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(slab + (long long)z * n + i);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(dW + i) = a;
  }
}

// fp32 rows -> T rows with optional dropout and zero padding
template <typename T>
__global__ void cast_rows_kernel(const float* __restrict__ src, int rows, int cols, int lds_, T* __restrict__ dst,
                                 int ld, uint32_t thr, float scale, TimSeed seed, uint32_t site,
                                 const float* __restrict__ vscale) {
  const int r = blockIdx.y;
  const float vs = vscale ? *vscale : 1.f;   // factor on the values (gradient scale of the fp16 mode)
  const int colsq = (cols + 3) >> 2;
  const bool vec = (lds_ & 3) == 0 && ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)dst) & 15) == 0);   // (ld % 4 == 0 always)
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q * 4 < ld; q += gridDim.x * blockDim.x) {
    float k[4] = {1.f, 1.f, 1.f, 1.f};
    if (thr != 0u && q < colsq) drop_mask4(seed, site, (uint64_t)r * colsq + q, thr, scale, k[0], k[1], k[2], k[3]);
    if (vec && q * 4 + 3 < cols) {   // 16-byte load, one 8- / 16-byte store
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * lds_ + q * 4);
      store4<T>(dst + (size_t)r * ld + q * 4, v.x * k[0] * vs, v.y * k[1] * vs, v.z * k[2] * vs, v.w * k[3] * vs);
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = q * 4 + j;
      if (c < ld) dst[(size_t)r * ld + c] = OpT<T>::from_f(c < cols ? src[(size_t)r * lds_ + c] * k[j] * vs : 0.f);
    }
  }
}

// two matrices with the same row count in one launch (blockIdx.z): the two modality embedders' inputs - same arithmetic and the same
// mask indexing as cast_rows_kernel (row * quads-per-row + quad, per site), contiguous fp32 rows
struct CastPair { const float* src[2]; void* dst[2]; int cols[2], ld[2]; uint32_t site[2]; };
template <typename T>
__global__ void cast_rows_pair_kernel(CastPair cp, int rows, uint32_t thr, float scale, TimSeed seed) {
  const int z = blockIdx.z, r = blockIdx.y;
  const float* __restrict__ src = cp.src[z];
  T* __restrict__ dst = (T*)cp.dst[z];
  const int cols = cp.cols[z], ld = cp.ld[z], colsq = (cols + 3) >> 2;
  const bool vec = (cols & 3) == 0 && ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)dst) & 15) == 0);
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q * 4 < ld; q += gridDim.x * blockDim.x) {
    float k[4] = {1.f, 1.f, 1.f, 1.f};
    if (thr != 0u && q < colsq) drop_mask4(seed, cp.site[z], (uint64_t)r * colsq + q, thr, scale, k[0], k[1], k[2], k[3]);
    if (vec && q * 4 + 3 < cols) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * cols + q * 4);
      store4<T>(dst + (size_t)r * ld + q * 4, v.x * k[0], v.y * k[1], v.z * k[2], v.w * k[3]);
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = q * 4 + j;
      if (c < ld) dst[(size_t)r * ld + c] = OpT<T>::from_f(c < cols ? src[(size_t)r * cols + c] * k[j] : 0.f);
    }
  }
}

// d_x[r, c] = g[r, c] * mask  (backward of the feature dropout on raw inputs)
__global__ void drop_bwd_rows_kernel(const float* __restrict__ g, int rows, int cols, int ldg, float* __restrict__ dx,
                                     int ldx, uint32_t thr, float scale, TimSeed seed, uint32_t site) {
  const int r = blockIdx.y;
  const int colsq = (cols + 3) >> 2;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < colsq; q += gridDim.x * blockDim.x) {
    float k[4] = {1.f, 1.f, 1.f, 1.f};
    if (thr != 0u) drop_mask4(seed, site, (uint64_t)r * colsq + q, thr, scale, k[0], k[1], k[2], k[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = q * 4 + j;
      if (c < cols) dx[(size_t)r * ldx + c] = g[(size_t)r * ldg + c] * k[j];
    }
  }
}

// out[c] += sum_r src[r, c]
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ src, int rows, int cols, int ld, float* __restrict__ out,
                              int rows_per_block) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += OpT<T>::to_f(src[(size_t)r * ld + c]);
  atomicAdd(out + c, s);
}

__global__ void dropout_mask_kernel(TimSeed seed, uint32_t site, uint32_t thr, int rows, int cols, uint8_t* out) {
  // element (r, c) has linear index r * cols + c; one thread per 8 consecutive elements = one Philox call (common.h)
  const size_t total = (size_t)rows * cols, groups = (total + 7) >> 3;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < groups; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t bits = drop_bits8(seed, site, i, thr);
    for (int e = 0; e < 8; ++e)
      if (8 * i + e < total) out[8 * i + e] = (bits >> e) & 1u;
  }
}

// ---------------------------------------------------------------------------
// LayerNorm.  One wave per row; the row lives in registers (cols <= 64*4*NV).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float act_f(int act, float v) {
  return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? gelu_f(v) : v);
}
__device__ __forceinline__ float act_grad_f(int act, float v) {
  return act == 1 ? (v > 0.f ? 1.f : 0.f) : (act == 2 ? gelu_grad_f(v) : 1.f);
}

constexpr int LN_MAXV_MAX = 8;  // float4 per lane: cols <= 2048 (kernels are instantiated for 1, 2, 4, 8)

// A second parameter set for the rows from `row` on (round 5): the two modality embedders' LayerNorms (encodings.py:21-26; same
// width, same activation, different gamma / beta) run as ONE launch over their stacked rows, forward and backward.  row =
// 0x7fffffff: one parameter set.  The backward picks per BLOCK (the caller stacks the halves at a multiple of the block's rows).
struct LnSplit { int row; const float* w2; const float* b2; float* dg2; float* db2; };

template <typename T, int LN_MAXV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ y, int rows, int cols, int ldy,
                                                     int act, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ xf, int ldx,
                                                     T* __restrict__ xt, int ldt, float* __restrict__ stats,
                                                     uint32_t* __restrict__ mbits, int mwords, uint32_t mthr,
                                                     TimSeed mseed, uint32_t msite, LnSplit sp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (row >= sp.row) { w = sp.w2; b = sp.b2; }
  if (mbits) {   // dropout keep-bits for the GEMM that consumes this row (common.h: drop_bits32): VALU work under the row's loads
    for (int wd = lane; wd < mwords; wd += 64)
      mbits[(size_t)row * mwords + wd] = drop_bits32(mseed, msite, ((uint64_t)row * mwords + wd) * 8, mthr);
  }
  const int nv = (cols + 255) >> 8;  // float4 slots per lane (cols % 4 == 0)
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < cols) {
      float4 t = *reinterpret_cast<const float4*>(y + (size_t)row * ldy + c);
      t.x = act_f(act, t.x); t.y = act_f(act, t.y); t.z = act_f(act, t.z); t.w = act_f(act, t.w);
      v[i] = t;
      s += (t.x + t.y) + (t.z + t.w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < cols) {
      const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  }
  const float var = wave_sum(q) / (float)cols;
  const float rstd = rsqrtf(var + 1e-5f);
  if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < cols) {
      const float4 g = *reinterpret_cast<const float4*>(w + c);
      const float4 be = *reinterpret_cast<const float4*>(b + c);
      const float o0 = (v[i].x - mean) * rstd * g.x + be.x, o1 = (v[i].y - mean) * rstd * g.y + be.y;
      const float o2 = (v[i].z - mean) * rstd * g.z + be.z, o3 = (v[i].w - mean) * rstd * g.w + be.w;
      if (xf) store4<float>(xf + (size_t)row * ldx + c, o0, o1, o2, o3);
      if (xt) store4<T>(xt + (size_t)row * ldt + c, o0, o1, o2, o3);
    }
  }
}

// Forward for cols = NS * 512 (the encoder widths): a lane owns 8 consecutive columns per 512-column slot - the operand copy
// leaves in 16-byte stores (1 KiB per wave instruction instead of 512 B) - and gamma / beta are requested together with the row,
// not after the reductions (one exposed L2 latency less per row).  Cold 9920 x 1024 (tools/ln_time.py): 18.9 -> 14.4 us.
template <typename T, int NS>
__global__ __launch_bounds__(256) void ln_fwd8_kernel(const float* __restrict__ y, int rows, int ldy, int act,
                                                      const float* __restrict__ w, const float* __restrict__ b,
                                                      float* __restrict__ xf, int ldx, T* __restrict__ xt, int ldt,
                                                      float* __restrict__ stats, uint32_t* __restrict__ mbits, int mwords,
                                                      uint32_t mthr, TimSeed mseed, uint32_t msite, const uint32_t* __restrict__ run_if,
                                                      LnSplit sp) {
  // the stand-by launch behind a GEMM that already normalised its rows (gemm_nt_ldln_kernel): run_if = that kernel's control
  // words {epoch, done, time-out word of even launches, of odd launches}; the launch in front of this one has advanced the epoch
  if (run_if && run_if[2 + ((run_if[0] - 1u) & 1u)] == 0u) return;
  constexpr int cols = NS * 512;
  const int lane = threadIdx.x & 63;
  // (a stand-by launch is a small grid that walks the rows; the normal launch has a block per four rows: one trip)
  for (int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < rows; row += gridDim.x * (blockDim.x >> 6)) {
  float4 v[NS][2], gw[NS][2], gb[NS][2];
  const float* yr = y + (size_t)row * ldy;
  const float* wr = row >= sp.row ? sp.w2 : w;
  const float* br = row >= sp.row ? sp.b2 : b;
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) v[i][h] = *reinterpret_cast<const float4*>(yr + i * 512 + lane * 8 + h * 4);
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      gw[i][h] = *reinterpret_cast<const float4*>(wr + i * 512 + lane * 8 + h * 4);
      gb[i][h] = *reinterpret_cast<const float4*>(br + i * 512 + lane * 8 + h * 4);
    }
  if (mbits) {   // dropout keep-bits for the GEMM that consumes this row: VALU work under the row's loads
    for (int wd = lane; wd < mwords; wd += 64)
      mbits[(size_t)row * mwords + wd] = drop_bits32(mseed, msite, ((uint64_t)row * mwords + wd) * 8, mthr);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 t = v[i][h];
      t.x = act_f(act, t.x); t.y = act_f(act, t.y); t.z = act_f(act, t.z); t.w = act_f(act, t.w);
      v[i][h] = t;
      s += (t.x + t.y) + (t.z + t.w);
    }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float a0 = v[i][h].x - mean, a1 = v[i][h].y - mean, a2 = v[i][h].z - mean, a3 = v[i][h].w - mean;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  const float var = wave_sum(q) / (float)cols;
  const float rstd = rsqrtf(var + 1e-5f);
  if (lane == 0 && stats) *reinterpret_cast<float2*>(stats + 2 * row) = make_float2(mean, rstd);
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    float o[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4 t = v[i][h], g = gw[i][h], be = gb[i][h];
      o[4 * h] = (t.x - mean) * rstd * g.x + be.x; o[4 * h + 1] = (t.y - mean) * rstd * g.y + be.y;
      o[4 * h + 2] = (t.z - mean) * rstd * g.z + be.z; o[4 * h + 3] = (t.w - mean) * rstd * g.w + be.w;
    }
    const int c = i * 512 + lane * 8;
    if (xf) {
      store4<float>(xf + (size_t)row * ldx + c, o[0], o[1], o[2], o[3]);
      store4<float>(xf + (size_t)row * ldx + c + 4, o[4], o[5], o[6], o[7]);
    }
    if (xt) {
      if constexpr (sizeof(T) == 2) {
        typedef T v8_t __attribute__((ext_vector_type(8)));
        v8_t pk;
#pragma unroll
        for (int u = 0; u < 8; ++u) pk[u] = OpT<T>::from_f(o[u]);
        *reinterpret_cast<v8_t*>(xt + (size_t)row * ldt + c) = pk;
      } else {
        store4<T>(xt + (size_t)row * ldt + c, o[0], o[1], o[2], o[3]);
        store4<T>(xt + (size_t)row * ldt + c + 4, o[4], o[5], o[6], o[7]);
      }
    }
  }
  }
}

// Backward.  A block owns ROWS_PB consecutive rows (one wave walks rows wave, wave+4, ...) and
// reduces dgamma/dbeta over its rows in registers, then LDS across its 4 waves, then one atomic
// per column per block.
// ACT0: the LayerNorm input is not an activation (act == 0, the encoder layers): no gelu' / relu' factors and their registers
// S16 (round 4, fp16 mode inside the encoder stack): bit 0 - `dx` is a T matrix holding the gradient times the gradient scale
// (what the previous LayerNorm-backward wrote with bit 1), bit 1 - `dyf` is written as such a T matrix (unmasked, times
// t_scale) instead of fp32: the residual part of the gradient stream travels 16-bit like its branch parts do - 120 instead
// of 160 MB per launch at C2a.  Its rounding (11 bits under the same scale as the operand copies) moves the parameter
// gradients by <= 1.5e-3 of their largest element (oracle with the stream rounded at every LayerNorm: cos 0.9999996).
template <typename T, int LN_MAXV, bool ACT0 = false, int S16 = 0>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dx, int lddx,
                                                     const float* __restrict__ y, int ldy,
                                                     const float* __restrict__ stats, int rows, int cols, int act,
                                                     const float* __restrict__ w, float* __restrict__ dyf, int lddy,
                                                     T* __restrict__ dyt, int ldt, uint32_t thr, float scale,
                                                     TimSeed seed, uint32_t site, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int rows_pb,
                                                     float* __restrict__ partial, const float* __restrict__ t_scale,
                                                     const T* __restrict__ addt, int ldadd, const float* __restrict__ add_scale,
                                                     LnSplit sp, int pair_sw) {
  extern __shared__ float red[];  // [4][2][cols]
  if (blockIdx.x * rows_pb >= sp.row) { w = sp.w2; dgamma = sp.dg2; dbeta = sp.db2; }
  const float ts = t_scale ? *t_scale : 1.f;   // factor on the operand-dtype copy (gradient scale of the fp16 mode)
  // optional second addend of the incoming gradient, in the operand dtype: dx_eff = dx + add_scale * addt.  The input-gradient
  // GEMM in front of this LayerNorm then stores its (scaled) 16-bit product instead of reading the fp32 stream and writing
  // the sum back (20 MB instead of 80 per launch at C2a)
  const float as = ((addt || (S16 & 1) != 0) && add_scale) ? *add_scale : 1.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = (cols + 255) >> 8;
  float4 ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) { ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); }
  const int r0 = blockIdx.x * rows_pb, r1 = min(rows, r0 + rows_pb);
  // gamma stays in registers; the loads of row + 4 are issued before the arithmetic of the current row so that every
  // wave always has a full row of y and dx in flight (the kernel is a pure HBM stream: 2 fp32 reads, 1.5 writes)
  float4 ww[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    ww[i] = (i < nv && c < cols) ? *reinterpret_cast<const float4*>(w + c) : make_float4(0, 0, 0, 0);
  }
  constexpr bool IN16 = (S16 & 1) != 0, OUT16 = (S16 & 2) != 0;
  float4 tn[LN_MAXV], dn[LN_MAXV];
  typedef T t4_t __attribute__((ext_vector_type(4)));
  t4_t an[LN_MAXV], dn16[LN_MAXV];
  float2 stn = make_float2(0.f, 0.f);
  auto fetch = [&](int row) {
    stn = *reinterpret_cast<const float2*>(stats + 2 * row);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < cols) {
        {   // y: a saved forward activation, cold and read once - nontemporal (-0.3 % of the step; dx / the 16-bit addend, fresh
            // from the previous kernel, and the outputs are better left to the cache policy: +0.5 ... +1.3 % as nontemporal)
          typedef float f4_t __attribute__((ext_vector_type(4)));
          const f4_t q0_ = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(y + (size_t)row * ldy + c));
          tn[i] = make_float4(q0_[0], q0_[1], q0_[2], q0_[3]);
        }
        if constexpr (IN16) dn16[i] = *reinterpret_cast<const t4_t*>(reinterpret_cast<const T*>(dx) + (size_t)row * lddx + c);
        else dn[i] = *reinterpret_cast<const float4*>(dx + (size_t)row * lddx + c);
        if (addt) an[i] = *reinterpret_cast<const t4_t*>(addt + (size_t)row * ldadd + c);
      }
    }
  };
  if (r0 + wave < r1) fetch(r0 + wave);
  float kq[4] = {1.f, 1.f, 1.f, 1.f};
  // (pairs of slots exist when the row is an even number of full 256-column slots; the library-wide switch rides in the sign of rows_pb)
  const bool pair_ok = (cols % 512) == 0 && nv <= LN_MAXV && pair_sw;
#pragma unroll 1
  for (int row = r0 + wave; row < r1; row += 4) {
    const float mean = stn.x, rstd = stn.y;
    float4 xh[LN_MAXV], d[LN_MAXV], ga[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      xh[i] = tn[i];
      if constexpr (IN16) {
        d[i] = make_float4(OpT<T>::to_f(dn16[i][0]) * as, OpT<T>::to_f(dn16[i][1]) * as, OpT<T>::to_f(dn16[i][2]) * as,
                           OpT<T>::to_f(dn16[i][3]) * as);
      } else {
        d[i] = dn[i];
      }
      if (addt) {
        d[i].x = fmaf(OpT<T>::to_f(an[i][0]), as, d[i].x); d[i].y = fmaf(OpT<T>::to_f(an[i][1]), as, d[i].y);
        d[i].z = fmaf(OpT<T>::to_f(an[i][2]), as, d[i].z); d[i].w = fmaf(OpT<T>::to_f(an[i][3]), as, d[i].w);
      }
    }
    if (row + 4 < r1) fetch(row + 4);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < cols) {
        const float4 t = xh[i];
        if (!ACT0 && act != 0) ga[i] = make_float4(act_grad_f(act, t.x), act_grad_f(act, t.y), act_grad_f(act, t.z),
                                          act_grad_f(act, t.w));
        float4 h;
        if (ACT0) {
          h.x = (t.x - mean) * rstd; h.y = (t.y - mean) * rstd; h.z = (t.z - mean) * rstd; h.w = (t.w - mean) * rstd;
        } else {
          h.x = (act_f(act, t.x) - mean) * rstd; h.y = (act_f(act, t.y) - mean) * rstd;
          h.z = (act_f(act, t.z) - mean) * rstd; h.w = (act_f(act, t.w) - mean) * rstd;
        }
        xh[i] = h;
        ag[i].x += d[i].x * h.x; ag[i].y += d[i].y * h.y; ag[i].z += d[i].z * h.z; ag[i].w += d[i].w * h.w;
        ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
        d[i].x *= ww[i].x; d[i].y *= ww[i].y; d[i].z *= ww[i].z; d[i].w *= ww[i].w;
        s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
        s2 += (d[i].x * h.x + d[i].y * h.y) + (d[i].z * h.z + d[i].w * h.w);
      }
    }
    s1 = wave_sum(s1) / (float)cols;
    s2 = wave_sum(s2) / (float)cols;
    // Dropout keep factors of the operand copy.  A lane owns 4 consecutive columns per slot, a Philox counter covers 8: lanes 2 k and
    // 2 k + 1 would draw the same counter in every slot.  Round 6 (PAIRED: an even number of full slots): the even lane draws slot
    // i's counter, the odd lane slot i + 1's, and the two trade the halves the other one owns (two DPP moves) - half the Philox
    // calls at the end of the row's dependent chain.  Same counters, same words, same bits (TIMHIP_LN_PAIR=0: every lane draws).
    const bool paired = LN_MAXV >= 2 && pair_ok && thr != 0u && dyt != nullptr;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      float kp[4] = {ts, ts, ts, ts};
      if constexpr (LN_MAXV >= 2) {
        if (paired && (i & 1) == 0) {
          const int isel = i + (lane & 1);
          const Philox4 r = philox4x32_7(seed, site, ((uint64_t)row * cols + (uint64_t)((isel * 64 + (lane & ~1)) * 4)) >> 3);
          const bool odd = (lane & 1) != 0;
          const uint32_t sa = odd ? r.x : r.z, sb = odd ? r.y : r.w;       // what the partner owns of my counter
          const uint32_t pa = (uint32_t)__builtin_amdgcn_mov_dpp((int)sa, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
          const uint32_t pb = (uint32_t)__builtin_amdgcn_mov_dpp((int)sb, 0xB1, 0xF, 0xF, true);
          // slot i: even lane its own (x, y), odd lane the even lane's (z, w); slot i + 1: even lane the odd lane's (x, y), odd its own (z, w)
          drop_mask4_words(odd ? pa : r.x, odd ? pb : r.y, thr, scale, kp[0], kp[1], kp[2], kp[3]);
          drop_mask4_words(odd ? r.z : pa, odd ? r.w : pb, thr, scale, kq[0], kq[1], kq[2], kq[3]);
#pragma unroll
          for (int u = 0; u < 4; ++u) { kp[u] *= ts; kq[u] *= ts; }
        } else if (paired) {
#pragma unroll
          for (int u = 0; u < 4; ++u) kp[u] = kq[u];
        }
      }
      if (i < nv && c < cols) {
        float o0 = rstd * (d[i].x - s1 - xh[i].x * s2), o1 = rstd * (d[i].y - s1 - xh[i].y * s2);
        float o2 = rstd * (d[i].z - s1 - xh[i].z * s2), o3 = rstd * (d[i].w - s1 - xh[i].w * s2);
        if (!ACT0 && act != 0) { o0 *= ga[i].x; o1 *= ga[i].y; o2 *= ga[i].z; o3 *= ga[i].w; }
        if constexpr (OUT16) {
          if (dyf) store4<T>(reinterpret_cast<T*>(dyf) + (size_t)row * lddy + c, o0 * ts, o1 * ts, o2 * ts, o3 * ts);
        } else {
          if (dyf) store4<float>(dyf + (size_t)row * lddy + c, o0, o1, o2, o3);
        }
        if (dyt) {
          float k0 = kp[0], k1 = kp[1], k2 = kp[2], k3 = kp[3];
          if (thr != 0u && !paired) {
            drop_mask4(seed, site, ((uint64_t)row * cols + c) >> 2, thr, scale, k0, k1, k2, k3);
            k0 *= ts; k1 *= ts; k2 *= ts; k3 *= ts;
          }
          store4<T>(dyt + (size_t)row * ldt + c, o0 * k0, o1 * k1, o2 * k2, o3 * k3);
        }
      }
    }
  }
  // block reduction of the column sums
  float* rg = red + (size_t)wave * 2 * cols;
  float* rb = rg + cols;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < cols) {
      *reinterpret_cast<float4*>(rg + c) = ag[i];
      *reinterpret_cast<float4*>(rb + c) = ab[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { sg += red[(size_t)k * 2 * cols + c]; sb += red[(size_t)k * 2 * cols + cols + c]; }
    if (partial) {  // deterministic two-stage reduction: [block][gamma | beta][cols]
      partial[(size_t)blockIdx.x * 2 * cols + c] = sg;
      partial[(size_t)blockIdx.x * 2 * cols + cols + c] = sb;
    } else {
      if (dgamma) atomicAdd(dgamma + c, sg);
      if (dbeta) atomicAdd(dbeta + c, sb);
    }
  }
}

// ---------------------------------------------------------------------------
// time MLP layer 1 (K = 2: an outer product, tim.py:67)  h[r, j] = relu(t0 w[j,0] + t1 w[j,1] + b[j])
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void time_l1_fwd_kernel(const float* __restrict__ times, int rows, int d,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          T* __restrict__ h, int ld, int rows_pb, int split) {
  // split != 0 (timhip_time_l1_fwd_split3): the row is written as the three 16-bit column blocks [hi | lo | hi] of the fp32 value
  // (block width ld, row stride 3 ld) - what timhip_split3_many (mode 0) made of this kernel's fp32 output in a second launch
  const int rs = split ? 3 * ld : ld;
  // a thread owns 4 consecutive columns (its weights stay in registers) and walks the block's rows: 16-byte stores, a few
  // hundred blocks (one block per row and 4-byte stores took 15.6 us for the 8000 x 512 rows of C2a)
  const int r0 = blockIdx.x * rows_pb, r1 = min(rows, r0 + rows_pb);
  if ((ld & 3) == 0) {
    const int nq = ld >> 2, rl_n = max(1, 256 / nq);          // row lanes per pass
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq;
    if (threadIdx.x >= nq * rl_n && nq <= 256) return;
    for (int q0 = q; q0 < nq; q0 += 256) {
      float w0[4], w1[4], bb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = 4 * q0 + u;
        w0[u] = j < d ? w[2 * j] : 0.f; w1[u] = j < d ? w[2 * j + 1] : 0.f; bb[u] = j < d ? b[j] : 0.f;
      }
      for (int r = r0 + (nq <= 256 ? rl : 0); r < r1; r += (nq <= 256 ? rl_n : 1)) {
        const float t0 = times[2 * r], t1 = times[2 * r + 1];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = 4 * q0 + u < d ? fmaxf(fmaf(t0, w0[u], fmaf(t1, w1[u], bb[u])), 0.f) : 0.f;
        T* dst = h + (size_t)r * rs + 4 * q0;
        store4<T>(dst, v[0], v[1], v[2], v[3]);
        if (split) {
          float lo[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) lo[u] = v[u] - OpT<T>::to_f(OpT<T>::from_f(v[u]));
          store4<T>(dst + ld, lo[0], lo[1], lo[2], lo[3]);
          store4<T>(dst + 2 * ld, v[0], v[1], v[2], v[3]);
        }
      }
    }
    return;
  }
  for (int r = r0; r < r1; ++r) {
    const float t0 = times[2 * r], t1 = times[2 * r + 1];
    for (int j = threadIdx.x; j < ld; j += blockDim.x) {
      float v = 0.f;
      if (j < d) v = fmaxf(fmaf(t0, w[2 * j], fmaf(t1, w[2 * j + 1], b[j])), 0.f);
      const T hi = OpT<T>::from_f(v);
      h[(size_t)r * rs + j] = hi;
      if (split) {
        h[(size_t)r * rs + ld + j] = OpT<T>::from_f(v - OpT<T>::to_f(hi));
        h[(size_t)r * rs + 2 * ld + j] = hi;
      }
    }
  }
}
// dh: gradient w.r.t. the post-relu h (T, relu mask already applied by the dgrad epilogue).
// dw[j,0] += sum_r dh t0 ; dw[j,1] += sum_r dh t1 ; db[j] += sum_r dh ; dt[r,:] = sum_j dh w[j,:]
template <typename T>
__global__ __launch_bounds__(256) void time_l1_bwd_kernel(const float* __restrict__ times, int rows, int d,
                                                          const float* __restrict__ w, const T* __restrict__ dh,
                                                          int ld, float* __restrict__ dw, float* __restrict__ db,
                                                          float* __restrict__ dt, int rows_pb,
                                                          const float* __restrict__ out_scale) {
  const float os = out_scale ? *out_scale : 1.f;   // factor on everything written (1 / gradient scale of the fp16 mode)
  float chk = 0.f;
  // one block: rows [r0, r1).  dw/db: a thread owns 4 consecutive columns (one vector load per row) and every RL-th row;
  // the row lanes are combined through LDS before the block's atomics (3 per column).  per-row dt via wave reduction
  const int r0 = blockIdx.x * rows_pb, r1 = min(rows, r0 + rows_pb);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((d & 3) == 0 && (ld & 3) == 0) {
    __shared__ float4 red[3][256];
    const int nq = d >> 2;
    const int RL = nq >= 256 ? 1 : (nq > 128 ? 1 : (nq > 64 ? 2 : 4));
    const int W = 256 / RL;
    const int ql = threadIdx.x % W, rl = threadIdx.x / W;
    for (int q0 = 0; q0 < nq; q0 += W) {
      const int q = q0 + ql;
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
      if (q < nq) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += RL) {
          float g0, g1, g2, g3;
          load4<T>(dh + (size_t)r * ld + 4 * q, g0, g1, g2, g3);
          const float t0 = times[2 * r], t1 = times[2 * r + 1];
          a0.x += g0 * t0; a0.y += g1 * t0; a0.z += g2 * t0; a0.w += g3 * t0;
          a1.x += g0 * t1; a1.y += g1 * t1; a1.z += g2 * t1; a1.w += g3 * t1;
          a2.x += g0; a2.y += g1; a2.z += g2; a2.w += g3;
        }
      }
      if (RL > 1) {
        red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2;
        __syncthreads();
        if (rl == 0) {
          for (int o = 1; o < RL; ++o) {
            const float4 b0 = red[0][o * W + ql], b1 = red[1][o * W + ql], b2 = red[2][o * W + ql];
            a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
            a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
            a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
          }
        }
        __syncthreads();
      }
      if (rl == 0 && q < nq) {
        const float v0[4] = {a0.x, a0.y, a0.z, a0.w}, v1[4] = {a1.x, a1.y, a1.z, a1.w}, v2[4] = {a2.x, a2.y, a2.z, a2.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = 4 * q + u;
          atomicAdd(dw + 2 * j, v0[u] * os); atomicAdd(dw + 2 * j + 1, v1[u] * os); atomicAdd(db + j, v2[u] * os);
          nf_note(chk, v0[u]); nf_note(chk, v1[u]); nf_note(chk, v2[u]);
        }
      }
    }
  } else {
    for (int j = threadIdx.x; j < d; j += 256) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      for (int r = r0; r < r1; ++r) {
        const float g = OpT<T>::to_f(dh[(size_t)r * ld + j]);
        a0 += g * times[2 * r]; a1 += g * times[2 * r + 1]; a2 += g;
      }
      atomicAdd(dw + 2 * j, a0 * os); atomicAdd(dw + 2 * j + 1, a1 * os); atomicAdd(db + j, a2 * os);
      nf_note(chk, a0); nf_note(chk, a1); nf_note(chk, a2);
    }
  }
  nf_commit(out_scale, chk);
  if (dt) {
    for (int r = r0 + wave; r < r1; r += 4) {
      float s0 = 0.f, s1 = 0.f;
      for (int j = lane; j < d; j += 64) {
        const float g = OpT<T>::to_f(dh[(size_t)r * ld + j]);
        s0 += g * w[2 * j]; s1 += g * w[2 * j + 1];
      }
      s0 = wave_sum(s0); s1 = wave_sum(s1);
      if (lane == 0) { dt[2 * r] = s0 * os; dt[2 * r + 1] = s1 * os; }
    }
  }
}

// ---------------------------------------------------------------------------
// sequence assembly (encodings.py:190-250), batch-first, no transposes
// ---------------------------------------------------------------------------
// the CLS token vectors ([d] each) and modality vectors ([2 d] each) by pointer: they are separate parameters, and gathering
// them into one buffer first cost two concatenation launches per forward (and a copy back per backward)
constexpr int SV_CLS = 8, SV_MOD = 4;
struct SeqVecs { const float* cls[SV_CLS]; const float* mod[SV_MOD]; };
struct SeqVecGrads { float* cls[SV_CLS]; float* mod[SV_MOD]; };
template <typename T>
__global__ void assemble_fwd_kernel(const TimSeqRow* __restrict__ rows, int B, int S, int d,
                                    const float* __restrict__ e0, const float* __restrict__ e1, int n_e_rows,
                                    SeqVecs sv, const float* __restrict__ te, int Trows,
                                    uint32_t thr, float scale, TimSeed seed,
                                    uint32_t site, float* __restrict__ x, T* __restrict__ xt) {
  const int bs = blockIdx.x;  // b*S + s
  const int b = bs / S, s = bs % S;
  const TimSeqRow r = rows[s];
  const int E = 2 * d;
  const float* left = r.kind == 1 ? sv.cls[r.src & (SV_CLS - 1)]
                                  : (r.kind == 0 ? e0 : e1) + ((size_t)b * n_e_rows + r.src) * d;
  const float* right = te + ((size_t)b * Trows + r.te_row) * d;
  const float* mv = r.mod >= 0 ? sv.mod[r.mod & (SV_MOD - 1)] : nullptr;
  for (int c = threadIdx.x * 4; c < E; c += blockDim.x * 4) {
    float4 v = c < d ? *reinterpret_cast<const float4*>(left + c) : *reinterpret_cast<const float4*>(right + (c - d));
    if (mv) { const float4 m4 = *reinterpret_cast<const float4*>(mv + c); v.x += m4.x; v.y += m4.y; v.z += m4.z; v.w += m4.w; }
    if (thr != 0u) {
      float k0, k1, k2, k3;
      drop_mask4(seed, site, ((uint64_t)bs * E + c) >> 2, thr, scale, k0, k1, k2, k3);
      v.x *= k0; v.y *= k1; v.z *= k2; v.w *= k3;
    }
    store4<float>(x + (size_t)bs * E + c, v.x, v.y, v.z, v.w);
    store4<T>(xt + (size_t)bs * E + c, v.x, v.y, v.z, v.w);
  }
}

// backward.  Kernel 1: block (s, y) owns 256 columns of token row s for ALL windows: 4 row lanes x 64 column quads, each row
// lane walks every 4th window (16-byte loads), writes d_e (feature rows) and keeps the cls / modality sums in registers; the
// row lanes are combined through LDS and the block issues ONE atomic per column (device-scope float atomics are resolved
// at the memory side on this part - an earlier version with 8 window groups per row spent most of its 110 us in them).
// Kernel 2: d_te[b, t, :] = sum over the token rows that read time row t (fixed order, no atomics).
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const TimSeqRow* __restrict__ rows, int B, int S, int d,
                                                           const float* __restrict__ dx, int n_e_rows, uint32_t thr,
                                                           float scale, TimSeed seed, uint32_t site,
                                                           float* __restrict__ d_e0, float* __restrict__ d_e1,
                                                           SeqVecGrads sg, int G) {
  // A block walks G consecutive token rows and carries the modality / cls sums across rows that add into the same vector
  // (detection: 399 query rows share one cls vector and one modality vector - one atomic per row and column took 75 us at
  // S = 499, the adds being resolved one at a time at the memory side); the atomics go out when the target changes.
  __shared__ float4 red[4][64];
  const int E = 2 * d;
  const int q = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int s_lo = blockIdx.x * G, s_hi = min(S, s_lo + G);
  for (int c0 = blockIdx.y * 256; c0 < E; c0 += gridDim.y * 256) {
    const int c = c0 + q * 4;
    float4 msum = make_float4(0.f, 0.f, 0.f, 0.f), csum = msum;
    int mtgt = -1, ctgt = -1;
    auto flush = [&](float* vec, float4& v) {
      if (vec) {
        float* p = vec + c;
        atomicAdd(p, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
      }
      v = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto mvec = [&](int t) -> float* { return t >= 0 ? sg.mod[t & (SV_MOD - 1)] : nullptr; };
    auto cvec = [&](int t) -> float* { return t >= 0 ? sg.cls[t & (SV_CLS - 1)] : nullptr; };
    for (int s = s_lo; s < s_hi; ++s) {
      const TimSeqRow r = rows[s];
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < E) {
#pragma unroll 4
        for (int b = rl; b < B; b += 4) {
          const size_t bs = (size_t)b * S + s;
          float4 g = *reinterpret_cast<const float4*>(dx + bs * E + c);
          if (thr != 0u) {
            float k0, k1, k2, k3;
            drop_mask4(seed, site, (bs * E + c) >> 2, thr, scale, k0, k1, k2, k3);
            g.x *= k0; g.y *= k1; g.z *= k2; g.w *= k3;
          }
          acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
          if (c < d && r.kind != 1) {
            float* d_e = r.kind == 0 ? d_e0 : d_e1;
            if (d_e) store4<float>(d_e + ((size_t)b * n_e_rows + r.src) * d + c, g.x, g.y, g.z, g.w);
          }
        }
      }
      red[rl][q] = acc;
      __syncthreads();
      if (rl == 0 && c < E) {
        const float4 a1 = red[1][q], a2 = red[2][q], a3 = red[3][q];
        acc.x += a1.x + a2.x + a3.x; acc.y += a1.y + a2.y + a3.y; acc.z += a1.z + a2.z + a3.z; acc.w += a1.w + a2.w + a3.w;
        const int mt = (r.mod >= 0 && mvec(r.mod)) ? r.mod : -1;
        if (mt != mtgt) { flush(mvec(mtgt), msum); mtgt = mt; }
        if (mt >= 0) { msum.x += acc.x; msum.y += acc.y; msum.z += acc.z; msum.w += acc.w; }
        const int ct = (c < d && r.kind == 1 && cvec(r.src)) ? r.src : -1;
        if (ct != ctgt) { flush(cvec(ctgt), csum); ctgt = ct; }
        if (ct >= 0) { csum.x += acc.x; csum.y += acc.y; csum.z += acc.z; csum.w += acc.w; }
      }
      __syncthreads();
    }
    if (rl == 0 && c < E) { flush(mvec(mtgt), msum); flush(cvec(ctgt), csum); }
  }
}

__global__ __launch_bounds__(128) void assemble_bwd_te_kernel(const TimSeqRow* __restrict__ rows, int B, int S, int d,
                                                              const float* __restrict__ dx, int Trows, uint32_t thr,
                                                              float scale, TimSeed seed, uint32_t site,
                                                              float* __restrict__ d_te) {
  // block (t, y): time row t for the y-th share of the windows.  The token rows that read time row t (1 for features and audio
  // queries, 3 for visual queries) are found ONCE per block, by a ballot scan of the table (one thread walking its 155 entries
  // took ~5 us of dependent loads per block).
  const int t = blockIdx.x;
  const int E = 2 * d;
  __shared__ int readers[8];
  __shared__ int nread;
  if (threadIdx.x < 64) {   // wave 0 scans the table 64 rows at a time (ballot; the readers stay in ascending order)
    int n = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + (int)threadIdx.x;
      unsigned long long m = __ballot(s < S && rows[s].te_row == t);
      if (threadIdx.x == 0) {
        while (m) {
          const int bit = __ffsll((long long)m) - 1;
          if (n < 8) readers[n] = s0 + bit;
          ++n;
          m &= m - 1;
        }
      }
    }
    if (threadIdx.x == 0) nread = n;
  }
  __syncthreads();
  const int nr = nread;   // > 8 (no TIM layout does this): the list is not used, every row is tested again
  const int bper = (B + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * bper, b1 = min(B, b0 + bper);
  for (int b = b0; b < b1; ++b) {
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const int cnt = nr <= 8 ? nr : S;
      for (int i = 0; i < cnt; ++i) {
        const int sr = nr <= 8 ? readers[i] : i;
        if (nr > 8 && rows[sr].te_row != t) continue;
        const size_t bs = (size_t)b * S + sr;
        float4 g = *reinterpret_cast<const float4*>(dx + bs * E + d + c);
        if (thr != 0u) {
          float k0, k1, k2, k3;
          drop_mask4(seed, site, (bs * E + d + c) >> 2, thr, scale, k0, k1, k2, k3);
          g.x *= k0; g.y *= k1; g.z *= k2; g.w *= k3;
        }
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
      }
      store4<float>(d_te + ((size_t)b * Trows + t) * d + c, acc.x, acc.y, acc.z, acc.w);
    }
  }
}

template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ xt, int B, int S, int E, int s0, int n, T* __restrict__ out) {
  const int i = blockIdx.x;  // b*n + j
  const int b = i / n, j = i % n;
  const T* src = xt + ((size_t)b * S + s0 + j) * E;
  T* dst = out + (size_t)i * E;
  for (int c = threadIdx.x * 4; c < E; c += blockDim.x * 4) {
    float a0, a1, a2, a3;
    load4<T>(src + c, a0, a1, a2, a3);
    store4<T>(dst + c, a0, a1, a2, a3);
  }
}
__global__ void scatter_rows_add_kernel(const float* __restrict__ d_rows, int B, int S, int E, int s0, int n,
                                        float* __restrict__ dx) {
  const int i = blockIdx.x;
  const int b = i / n, j = i % n;
  const float* src = d_rows + (size_t)i * E;
  float* dst = dx + ((size_t)b * S + s0 + j) * E;
  for (int c = threadIdx.x * 4; c < E; c += blockDim.x * 4) {
    const float4 a = *reinterpret_cast<const float4*>(src + c);
    float4 o = *reinterpret_cast<float4*>(dst + c);
    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    *reinterpret_cast<float4*>(dst + c) = o;
  }
}

// ---- the same row moves for several token ranges in one launch (the classification heads: 4 ranges per direction) ----
constexpr int RR_MAX = 6;
struct RowRanges {
  const void* src[RR_MAX]; void* dst[RR_MAX];   // per range: gathered / per-head buffer on one side, the [B,S,E] stream on the other
  int s0[RR_MAX], n[RR_MAX], joff[RR_MAX + 1];  // token range, prefix of n
  int count;
};
template <typename T>
__global__ void gather_ranges_kernel(const T* __restrict__ xt, int B, int S, int E, RowRanges rr) {
  const int per = rr.joff[rr.count];
  const int b = blockIdx.x / per, jg = blockIdx.x % per;
  int r = 0;
#pragma unroll
  for (int k = 1; k < RR_MAX; ++k)
    if (k < rr.count && jg >= rr.joff[k]) r = k;
  const int j = jg - rr.joff[r];
  const T* src = xt + ((size_t)b * S + rr.s0[r] + j) * E;
  T* dst = (T*)rr.dst[r] + ((size_t)b * rr.n[r] + j) * E;
  for (int c = threadIdx.x * 4; c < E; c += blockDim.x * 4) {
    float a0, a1, a2, a3;
    load4<T>(src + c, a0, a1, a2, a3);
    store4<T>(dst + c, a0, a1, a2, a3);
  }
}
// the gathered fp32 rows written straight as split operands [hi | lo | hi] (three 16-bit column blocks of width E, row stride 3 E):
// gather_ranges + split3 (mode 0) of the classification heads' fp16 path in one launch, without the fp32 row buffers
template <typename T>
__global__ void gather_split3_ranges_kernel(const float* __restrict__ x, int B, int S, int E, RowRanges rr) {
  const int per = rr.joff[rr.count];
  const int b = blockIdx.x / per, jg = blockIdx.x % per;
  int r = 0;
#pragma unroll
  for (int k = 1; k < RR_MAX; ++k)
    if (k < rr.count && jg >= rr.joff[k]) r = k;
  const int j = jg - rr.joff[r];
  const float* src = x + ((size_t)b * S + rr.s0[r] + j) * E;
  T* dst = (T*)rr.dst[r] + ((size_t)b * rr.n[r] + j) * 3 * E;
  for (int c = threadIdx.x * 4; c < E; c += blockDim.x * 4) {
    const float4 v = *reinterpret_cast<const float4*>(src + c);
    const float f[4] = {v.x, v.y, v.z, v.w};
    float lo[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) lo[u] = f[u] - OpT<T>::to_f(OpT<T>::from_f(f[u]));
    store4<T>(dst + c, f[0], f[1], f[2], f[3]);
    store4<T>(dst + E + c, lo[0], lo[1], lo[2], lo[3]);
    store4<T>(dst + 2 * E + c, f[0], f[1], f[2], f[3]);
  }
}
__global__ void scatter_ranges_add_kernel(int B, int S, int E, float* __restrict__ dx, RowRanges rr) {
  const int per = rr.joff[rr.count];
  const int b = blockIdx.x / per, jg = blockIdx.x % per;
  int r = 0;
#pragma unroll
  for (int k = 1; k < RR_MAX; ++k)
    if (k < rr.count && jg >= rr.joff[k]) r = k;
  const int j = jg - rr.joff[r];
  const float* src = (const float*)rr.src[r] + ((size_t)b * rr.n[r] + j) * E;
  float* dst = dx + ((size_t)b * S + rr.s0[r] + j) * E;
  for (int c = threadIdx.x * 4; c < E; c += blockDim.x * 4) {
    const float4 a = *reinterpret_cast<const float4*>(src + c);
    float4 o = *reinterpret_cast<float4*>(dst + c);
    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    *reinterpret_cast<float4*>(dst + c) = o;
  }
}
// The gradient stream entering the encoder stack, written in ONE pass: token row (b, s) <- the `feats` cotangent (feature rows;
// zero without one), the row of the head whose token range covers s (query rows; ranges disjoint), or zero.  Replaces a strided
// zero fill of the query rows (29 us at C2a), a copy of the feature rows (14 us) and the heads' read-modify-write scatter (8 us).
__global__ __launch_bounds__(256) void dx_init_kernel(int B, int S, int F, int E, const float* __restrict__ feats,
                                                      float* __restrict__ dx, RowRanges rr) {
  const int b = blockIdx.x / S, s = blockIdx.x % S;
  const float* src = nullptr;
  int nslab = 1;
  size_t sstride = 0;
  if (s < F) {
    if (feats) src = feats + ((size_t)b * F + s) * E;
  } else {
#pragma unroll
    for (int k = 0; k < RR_MAX; ++k)
      if (k < rr.count && s >= rr.s0[k] && s < rr.s0[k] + rr.n[k]) {
        src = (const float*)rr.src[k] + ((size_t)b * rr.n[k] + (s - rr.s0[k])) * E;
        nslab = rr.joff[k];                      // (dx_init: joff[k] = number of [B n, E] slabs of range k to add up, >= 1)
        sstride = (size_t)B * rr.n[k] * E;
      }
  }
  float* dst = dx + ((size_t)b * S + s) * E;
  for (int c = threadIdx.x * 4; c < E; c += blockDim.x * 4) {
    float4 v = src ? *reinterpret_cast<const float4*>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 1; z < nslab; ++z) {
      const float4 w = *reinterpret_cast<const float4*>(src + z * sstride + c);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    *reinterpret_cast<float4*>(dst + c) = v;
  }
}
// dst[r, c] = T(scale * g[r, c] * y[r, c] * (1 - y[r, c])) for c < cols, 0 up to ld: the backward of the regression heads' sigmoid
// (det head.py:95-163: 2 outputs per query) straight into the zero-padded operand rows of the gradient GEMMs
template <typename T>
__global__ void sigmoid_bwd_rows_kernel(const float* __restrict__ g, const float* __restrict__ y, int rows, int cols,
                                        T* __restrict__ dst, int ld, const float* __restrict__ vscale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * ld) return;
  const int r = (int)(i / ld), c = (int)(i % ld);
  float v = 0.f;
  if (c < cols) {
    const float yy = y[(size_t)r * cols + c];
    v = g[(size_t)r * cols + c] * yy * (1.f - yy) * (vscale ? *vscale : 1.f);
  }
  dst[i] = OpT<T>::from_f(v);
}

// fp32 [rows, cols] -> T [rows, ld] (zero padded) for several matrices in one launch (blockIdx.z = matrix)
struct CastMany {
  const float* src[RR_MAX]; void* dst[RR_MAX];
  int rows[RR_MAX], cols[RR_MAX], ld[RR_MAX];
  const float* vscale;   // device scalar or NULL: factor on the values
};
template <typename T>
__global__ void cast_many_kernel(CastMany cm) {
  const int i = blockIdx.z;
  const int r = blockIdx.y;
  if (r >= cm.rows[i]) return;
  const float vs = cm.vscale ? *cm.vscale : 1.f;
  const float* src = cm.src[i] + (size_t)r * cm.cols[i];
  T* dst = (T*)cm.dst[i] + (size_t)r * cm.ld[i];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cm.ld[i]; c += gridDim.x * blockDim.x)
    dst[c] = OpT<T>::from_f(c < cm.cols[i] ? src[c] * vs : 0.f);
}

// ---------------------------------------------------------------------------
// Split operands (timhip_split3_many): an fp32 matrix as THREE 16-bit column blocks so that a plain 16-bit GEMM over the
// tripled contraction length computes the product to ~22 bits:  x = hi + lo, hi = T(x), lo = T(x - hi);
//   activations (mode 0): [hi | lo | hi]      weights (mode 1): [hi | hi | lo]
//   sum over the three blocks = x_hi w_hi + x_lo w_hi + x_hi w_lo   (the lo * lo term is below 2^-22 relative).
// The fp16 mode uses it at the two small sites that dominate its error budget (time MLP, classification heads).
// ---------------------------------------------------------------------------
struct Split3Many {
  const float* src[RR_MAX]; void* dst[RR_MAX];
  int rows[RR_MAX], cols[RR_MAX], lds[RR_MAX], ldd[RR_MAX];   // ldd = 3 * block width (block width = cols rounded up to 64)
  int mode, relu;
};
template <typename T>
__global__ __launch_bounds__(256) void split3_kernel(Split3Many sm) {
  // one thread per 4 consecutive columns of a row (block width is a multiple of 64): a 16-byte load, three 8-byte stores
  const int i = blockIdx.z;
  const int cols = sm.cols[i], cp = sm.ldd[i] / 3, qpr = cp >> 2;
  const long long nq = (long long)sm.rows[i] * qpr;
  const bool vec = (sm.lds[i] & 3) == 0 && (((uintptr_t)sm.src[i]) & 15) == 0;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(q / qpr), c = (int)(q % qpr) * 4;
    const float* src = sm.src[i] + (size_t)r * sm.lds[i] + c;
    float v[4];
    if (vec && c + 3 < cols) {
      const float4 f = *reinterpret_cast<const float4*>(src);
      v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = c + u < cols ? src[u] : 0.f;
    }
    float lo[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (sm.relu) v[u] = fmaxf(v[u], 0.f);
      const float h = OpT<T>::to_f(OpT<T>::from_f(v[u]));
      lo[u] = v[u] - h;
    }
    T* dst = (T*)sm.dst[i] + (size_t)r * sm.ldd[i] + c;
    store4<T>(dst, v[0], v[1], v[2], v[3]);
    if (sm.mode == 0) {
      store4<T>(dst + cp, lo[0], lo[1], lo[2], lo[3]);
      store4<T>(dst + 2 * cp, v[0], v[1], v[2], v[3]);
    } else {
      store4<T>(dst + cp, v[0], v[1], v[2], v[3]);
      store4<T>(dst + 2 * cp, lo[0], lo[1], lo[2], lo[3]);
    }
  }
}

// ---------------------------------------------------------------------------
// Data-parallel gradient exchange, step 3 of tim_amd/dp.py: rank r holds chunk r of every rank's bucket (recv [W][per], wire
// dtype) after the all-to-all; out[i] = T(scale * sum_w float(recv[w][i])) - fp32 accumulation whatever travelled on the wire.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dp_reduce_kernel(const T* __restrict__ recv, int W, long long per, float scale,
                                                        T* __restrict__ out) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < per; i += (long long)gridDim.x * blockDim.x * 4) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int w = 0; w < W; ++w) {
      float v0, v1, v2, v3;
      load4<T>(recv + (long long)w * per + i, v0, v1, v2, v3);
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    store4<T>(out + i, a0 * scale, a1 * scale, a2 * scale, a3 * scale);
  }
}

// ---------------------------------------------------------------------------
// Gradient scale of the fp16 mode (timhip_grad_scale): S = the power of two that brings the largest |cotangent| to
// `target`; out = {S, 1/S, scratch, scratch}.  One launch: every block folds its maximum into out[2] (float bits compare
// like unsigned integers for non-negative values), the last block to arrive (ticket in out[3]) writes S and 1/S and
// clears the scratch words for the next call.
// ---------------------------------------------------------------------------
constexpr int GS_MAX = 8;
struct GsList { const float* p[GS_MAX]; long long n[GS_MAX]; int count; };
__global__ __launch_bounds__(256) void grad_scale_kernel(GsList gl, float target, float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < GS_MAX; ++i) {   // (static indexing of the kernel-argument arrays)
    if (i >= gl.count) break;
    const float* __restrict__ p = gl.p[i];
    const long long n = gl.n[i], n4 = n >> 2;
    const bool al = (((uintptr_t)p) & 15) == 0;
    if (al) {
      // a block takes chunks of 1024 float4 (16 KiB): every lane has four 16-byte loads in flight whatever the tensor's size
      const long long nchunk = n4 >> 10;
      for (long long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        const float4* q = reinterpret_cast<const float4*>(p) + (ch << 10) + threadIdx.x;
        const float4 a = q[0], b = q[256], c = q[512], d = q[768];
        const float ma = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
        const float mb = fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)));
        const float mc = fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w)));
        const float md = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
        m = fmaxf(m, fmaxf(fmaxf(ma, mb), fmaxf(mc, md)));
      }
      for (long long q = (nchunk << 10) + t0; q < n4; q += stride) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
    }
    for (long long j = (al ? 4 * n4 : 0) + t0; j < n; j += stride) m = fmaxf(m, fabsf(p[j]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    unsigned* w = reinterpret_cast<unsigned*>(out);
    if (!(m <= 3.0e38f)) m = 3.0e38f;   // inf / nan cotangents: smallest scale
    atomicMax(w + 2, __float_as_uint(m));
    __threadfence();
    const unsigned ticket = atomicAdd(w + 3, 1u);
    if (ticket == gridDim.x - 1) {
      const float amax = __uint_as_float(atomicMax(w + 2, 0u));
      float S = 1.f;
      if (amax > 0.f) {
        int e = (int)floorf(log2f(target / amax));
        e = e < -40 ? -40 : (e > 40 ? 40 : e);
        S = exp2f((float)e);
      }
      out[0] = S; out[1] = 1.f / S;
      atomicExch(w + 2, 0u); atomicExch(w + 3, 0u);
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

int tim_transpose(int precision, const void* src, int rows, int cols, int lds_, void* dst, int ld, float* colsum,
                  hipStream_t s) {
  if (!src || !dst || rows <= 0 || cols <= 0 || ld < rows) return TIMHIP_EINVAL;
  dim3 grid((cols + 31) / 32, (ld + TP_ROWS - 1) / TP_ROWS);
  DISPATCH_T(precision, hipLaunchKernelGGL((transpose_kernel<T, T>), grid, dim3(256), 0, s, (const T*)src, rows,
                                           cols, lds_, (T*)dst, ld, colsum));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int tim_slab_reduce(const float* slab, long long n, int nslab, float* dW, hipStream_t s) {
  if (!slab || !dW || n <= 0 || (n & 3)) return TIMHIP_EINVAL;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slab, n, nslab, dW);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int tim_colsum(int precision, const void* src, int rows, int cols, int ld, float* out, hipStream_t s) {
  if (!src || !out || rows <= 0 || cols <= 0) return TIMHIP_EINVAL;
  const int rpb = 64;
  dim3 grid((cols + 255) / 256, (rows + rpb - 1) / rpb);
  DISPATCH_T(precision, hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(256), 0, s, (const T*)src, rows, cols, ld,
                                           out, rpb));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int tim_layernorm_fwd(int precision, const float* y, int rows, int cols, int ldy, int act, const float* w,
                      const float* b, float* xf, int ldx, void* xt, int ldt, float* stats, hipStream_t s,
                      uint8_t* mask_out, int mask_cols, float mask_p, uint64_t mask_seed, uint32_t mask_site,
                      const uint32_t* run_if, int split_row, const float* w2, const float* b2) {
  if (!y || !w || !b || rows <= 0) return TIMHIP_EINVAL;
  // (bench.py's non-GEMM brackets: fp32 rows read; fp32 and / or operand-type rows written)
  TimGemmScope timing((double)rows * cols * (4 + (xf ? 4 : 0) + (xt ? (f32_storage(precision) ? 4 : 2) : 0)) + (mask_out ? (double)rows * mask_cols / 8 : 0.0), s, 2);
  LnSplit sp{0x7fffffff, nullptr, nullptr, nullptr, nullptr};
  if (split_row > 0 && split_row < rows) {
    if (!w2 || !b2 || (((uintptr_t)w2 | (uintptr_t)b2) & 15)) return TIMHIP_EINVAL;
    sp.row = split_row; sp.w2 = w2; sp.b2 = b2;
  }
  if (cols % 4 || cols > 256 * LN_MAXV_MAX || ldy % 4 || (xf && ldx % 4) || (xt && ldt % 4)) return TIMHIP_EUNSUPPORTED;
  dim3 grid((rows + 3) / 4);
  uint32_t* mbits = nullptr;
  int mwords = 0;
  uint32_t mthr = 0u;
  if (mask_out && mask_p > 0.f) {
    if (mask_cols % 32 || ((uintptr_t)mask_out & 3)) return TIMHIP_EUNSUPPORTED;
    mbits = (uint32_t*)mask_out; mwords = mask_cols / 32; mthr = drop_threshold(mask_p);
  }
#define LN_FWD(NV) hipLaunchKernelGGL((ln_fwd_kernel<T, NV>), grid, dim3(256), 0, s, y, rows, cols, ldy, act, w, b, \
                                   xf, ldx, (T*)xt, ldt, stats, mbits, mwords, mthr, TimSeed(mask_seed), mask_site, sp)
  const int nv = (cols + 255) / 256;
  // encoder widths (512 / 1024 / 2048 columns, 16-byte aligned rows): the 8-columns-per-lane kernel
  const bool al8 = (ldy % 4) == 0 && (!xt || ldt % 8 == 0) && (!xf || ldx % 4 == 0) &&
                   ((((uintptr_t)y | (uintptr_t)xt | (uintptr_t)xf | (uintptr_t)w | (uintptr_t)b) & 15) == 0);
  // (the kernel walks rows grid-stride: a block of 4 waves takes `frpb` rows, one per wave and trip)
  const int frpb = tim_knobs().ln_fwd_rpb >= 4 ? tim_knobs().ln_fwd_rpb / 4 * 4 : 4;
  const dim3 grid8 = run_if ? dim3(min((int)grid.x, 256)) : dim3((rows + frpb - 1) / frpb);
#define LN_FWD8(NS) hipLaunchKernelGGL((ln_fwd8_kernel<T, NS>), grid8, dim3(256), 0, s, y, rows, ldy, act, w, b, xf, ldx, (T*)xt, ldt, \
                                     stats, mbits, mwords, mthr, TimSeed(mask_seed), mask_site, run_if, sp)
  if (run_if && !(al8 && (cols == 512 || cols == 1024 || cols == 2048))) return TIMHIP_EUNSUPPORTED;   // (only the 8-column kernel has the switch)
  if (al8 && (cols == 512 || cols == 1024 || cols == 2048)) {
    DISPATCH_T(precision, if (cols == 512) LN_FWD8(1); else if (cols == 1024) LN_FWD8(2); else LN_FWD8(4));
  } else {
    DISPATCH_T(precision, if (nv <= 1) LN_FWD(1); else if (nv <= 2) LN_FWD(2); else if (nv <= 4) LN_FWD(4); else LN_FWD(8));
  }
#undef LN_FWD8
#undef LN_FWD
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

// 16 rows per block, more when that would exceed the 768 co-resident blocks (3 per CU at the 154 VGPRs of the encoder's
// act == 0 kernel): one balanced round.  Round 6: FEWER for small row counts - 1240 rows (C2a at 8 windows per GPU, the reference
// recipe on 8 GPUs) made 78 blocks of 16 rows on 256 CUs, 12.3 us for 18 MB; 4 / 8 / 12 rows per block keep about 500 blocks
static int ln_bwd_rows_per_block(int rows) {
  if (tim_knobs().ln_rpb >= 4) return tim_knobs().ln_rpb / 4 * 4;   // (A/B knob)
  int rpb = 16;
  if (rows > 16 * 768) rpb = (((rows + 767) / 768) + 3) / 4 * 4;
  else if (rows < 16 * 512) { rpb = (((rows + 511) / 512) + 3) / 4 * 4; if (rpb < 4) rpb = 4; }
  else {
    // 512 - 768 blocks of 16 rows all run at once, 2 on some CUs and 3 on others: 9920 rows = 620 blocks = 2.42 per CU, and the launch
    // ends with the CUs that hold 3.  20 rows per block = 496 blocks = 1.94 per CU: LayerNorm's in-step brackets 678 -> 640 us per
    // step, the step -0.4 % (profiles/r06_s_ln_rpb_step_ab.txt; 40 rows = one block per CU: brackets -13 us, step +0.8 %).  Pick the
    // fullest rounds of 256 among 16 / 20 / 24 rows (ties: the fewest rows).
    double best = 0.0;
    for (int c = 16; c <= 24; c += 4) {
      const int blocks = (rows + c - 1) / c, rounds = (blocks + 255) / 256;
      const double fill = (double)blocks / (rounds * 256.0);
      if (fill > best + 0.02) { best = fill; rpb = c; }
    }
  }
  return rpb;
}

// (per-block partial sums [gamma | beta][cols] of one launch without atomics: as many rows as the launch has blocks)
size_t tim_layernorm_bwd_ws(int rows, int cols) {
  const int rpb = ln_bwd_rows_per_block(rows);
  return (size_t)((rows + rpb - 1) / rpb) * 2 * cols * sizeof(float);
}

int tim_layernorm_bwd(int precision, const float* dx, int lddx, const float* y, int ldy, const float* stats,
                      int rows, int cols, int act, const float* w, float* dyf, int lddy, void* dyt, int ldt,
                      float p_drop, uint64_t seed, uint32_t site, float* dgamma, float* dbeta, float* partial_ws,
                      hipStream_t s, bool defer_colsum, const float* t_scale, const void* addt, int ldadd,
                      const float* add_scale, int stream16, int split_row, const float* w2, float* dgamma2, float* dbeta2) {
  if (!dx || !y || !stats || !w || rows <= 0) return TIMHIP_EINVAL;
  // (gradient rows + pre-norm rows read, the 16-bit addend if any; fp32 and / or operand-type gradient rows written)
  TimGemmScope timing((double)rows * cols * (8 + (addt ? 2 : 0) + (dyf ? 4 : 0) + (dyt ? (f32_storage(precision) ? 4 : 2) : 0)), s, 2);
  LnSplit sp{0x7fffffff, nullptr, nullptr, nullptr, nullptr};
  if (stream16 && (act != 0 || !h16_storage(precision) || !add_scale || !t_scale)) return TIMHIP_EUNSUPPORTED;
  if (cols % 4 || cols > 256 * LN_MAXV_MAX || ldy % 4 || lddx % 4 || (dyf && lddy % 4) || (dyt && ldt % 4) || (addt && ldadd % 4))
    return TIMHIP_EUNSUPPORTED;
  int rpb = ln_bwd_rows_per_block(rows);
  // launches that end in atomics on dgamma / dbeta (no partials: the time MLP's and the embedders' LayerNorms): A/B knob of their own
  if (!partial_ws && tim_knobs().ln_rpb_small >= 4) rpb = tim_knobs().ln_rpb_small / 4 * 4;
  if (split_row > 0 && split_row < rows) {
    // a block takes its parameter set from its first row: the halves must meet at a multiple of the block's rows (no partials).
    // The balanced-round choice above grows with the row count (20, 24, 28 ... rows per block past 12288 rows) and the knobs
    // change it: where it does not divide the split row, fall back to the largest multiple of 4 up to 16 that does (round-5
    // advisor finding: B >= 160 windows of 50 feature tokens used to be refused here, in the BACKWARD of a forward that ran)
    if (!w2 || partial_ws) return TIMHIP_EINVAL;
    if (split_row % rpb) {
      rpb = 0;
      for (int c = 16; c >= 4 && !rpb; c -= 4) if (split_row % c == 0) rpb = c;
      if (!rpb) return TIMHIP_EINVAL;
    }
    sp.row = split_row; sp.w2 = w2; sp.dg2 = dgamma2; sp.db2 = dbeta2;
  }
  dim3 grid((rows + rpb - 1) / rpb);
  const size_t shmem = (size_t)4 * 2 * cols * sizeof(float);
  const uint32_t thr = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int pair_sw = tim_knobs().ln_pair;
#define LN_BWD(NV) hipLaunchKernelGGL((ln_bwd_kernel<T, NV>), grid, dim3(256), shmem, s, dx, lddx, y, ldy, stats, rows, \
                                   cols, act, w, dyf, lddy, (T*)dyt, ldt, thr, scale, seed, site, dgamma, dbeta, rpb, partial_ws, t_scale, \
                                   (const T*)addt, ldadd, add_scale, sp, pair_sw)
  const int nv = (cols + 255) / 256;
#define LN_BWD0(NV) hipLaunchKernelGGL((ln_bwd_kernel<T, NV, true>), grid, dim3(256), shmem, s, dx, lddx, y, ldy, stats, rows, \
                                   cols, act, w, dyf, lddy, (T*)dyt, ldt, thr, scale, seed, site, dgamma, dbeta, rpb, partial_ws, t_scale, \
                                   (const T*)addt, ldadd, add_scale, sp, pair_sw)
#define LN_BWD16(NV, SV) hipLaunchKernelGGL((ln_bwd_kernel<HT, NV, true, SV>), grid, dim3(256), shmem, s, dx, lddx, y, ldy, stats, rows, \
                                   cols, act, w, dyf, lddy, (HT*)dyt, ldt, thr, scale, seed, site, dgamma, dbeta, rpb, partial_ws, t_scale, \
                                   (const HT*)addt, ldadd, add_scale, sp, pair_sw)
#define LN_BWD16_NV(SV) do { if (nv <= 1) LN_BWD16(1, SV); else if (nv <= 2) LN_BWD16(2, SV); else if (nv <= 4) LN_BWD16(4, SV); else LN_BWD16(8, SV); } while (0)
  if (stream16) {
    DISPATCH_H16(precision, { if (stream16 == 1) LN_BWD16_NV(1); else if (stream16 == 2) LN_BWD16_NV(2); else LN_BWD16_NV(3); });
  } else if (act == 0 && nv == 4) {
    DISPATCH_T(precision, LN_BWD0(4));
  } else {
    DISPATCH_T(precision, if (nv <= 1) LN_BWD(1); else if (nv <= 2) LN_BWD(2); else if (nv <= 4) LN_BWD(4); else LN_BWD(8));
  }
#undef LN_BWD16_NV
#undef LN_BWD16
#undef LN_BWD0
#undef LN_BWD
  TIM_CHECK_LAUNCH();
  if (partial_ws && (dgamma || dbeta) && !defer_colsum) {
    hipLaunchKernelGGL(partial_colsum_kernel, dim3((2 * cols + 255) / 256, 32), dim3(256), 0, s, partial_ws,
                       (int)grid.x, cols, dgamma, dbeta);
    TIM_CHECK_LAUNCH();
  }
  return TIMHIP_OK;
}

int tim_layernorm_bwd_blocks(int rows) {
  const int rpb = ln_bwd_rows_per_block(rows);
  return (rows + rpb - 1) / rpb;
}

extern "C" {

int timhip_cast_weight(int precision, const float* src, int rows, int cols, void* dst, int ld, int transpose,
                       void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!src || !dst || rows <= 0 || cols <= 0 || ld % 64) return TIMHIP_EINVAL;
  if (!transpose) {
    if (ld < cols) return TIMHIP_EINVAL;
    dim3 grid((ld + 255) / 256, rows);
    DISPATCH_T(precision, hipLaunchKernelGGL(cast_weight_kernel<T>, grid, dim3(256), 0, s, src, rows, cols, (T*)dst, ld));
  } else {
    if (ld < rows) return TIMHIP_EINVAL;
    dim3 grid((cols + 31) / 32, (ld + TP_ROWS - 1) / TP_ROWS);
    DISPATCH_T(precision, hipLaunchKernelGGL((transpose_kernel<float, T>), grid, dim3(256), 0, s, src, rows, cols,
                                             cols, (T*)dst, ld, (float*)nullptr));
  }
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_cast_weight_both(int precision, const float* src, int rows, int cols, void* plain, int ldp, void* tr, int ldt,
                            void* stream) {
  if (!src || !plain || !tr || rows <= 0 || cols <= 0 || ldp % 64 || ldt % 64 || ldp < cols || ldt < rows)
    return TIMHIP_EINVAL;
  dim3 grid((max(ldp, cols) + 31) / 32, (max(ldt, rows) + 31) / 32);
  DISPATCH_T(precision, hipLaunchKernelGGL(cast_weight_both_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, src,
                                           rows, cols, (T*)plain, ldp, (T*)tr, ldt));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_cast_weights(int precision, const TimCastItem* items, int n, void* stream) {
  if (!items || n < 0) return TIMHIP_EINVAL;
  for (int i0 = 0; i0 < n; i0 += CAST_MAX) {
    CastBatch cb;
    cb.n = n - i0 < CAST_MAX ? n - i0 : CAST_MAX;
    int blocks = 0;
    for (int i = 0; i < cb.n; ++i) {
      const TimCastItem& it = items[i0 + i];
      if (!it.src || !it.plain || !it.tr || it.rows <= 0 || it.cols <= 0 || it.ldp % 64 || it.ldt % 64 ||
          it.ldp < it.cols || it.ldt < it.rows)
        return TIMHIP_EINVAL;
      if ((((uintptr_t)it.src | (uintptr_t)it.plain | (uintptr_t)it.tr) & 15) != 0) return TIMHIP_EALIGN;
      cb.it[i] = it;
      cb.blk0[i] = blocks;
      blocks += ((max(it.ldp, it.cols) + 63) / 64) * ((max(it.ldt, it.rows) + 63) / 64);
    }
    cb.blk0[cb.n] = blocks;
    DISPATCH_T(precision, hipLaunchKernelGGL(cast_weights_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cb));
    TIM_CHECK_LAUNCH();
  }
  return TIMHIP_OK;
}

int timhip_transpose(int precision, const void* src, int rows, int cols, int lds_, void* dst, int ld, void* stream) {
  return tim_transpose(precision, src, rows, cols, lds_, dst, ld, nullptr, (hipStream_t)stream);
}

int timhip_colsum(int precision, const void* src, int rows, int cols, int ld, float* out, void* stream) {
  return tim_colsum(precision, src, rows, cols, ld, out, (hipStream_t)stream);
}

int timhip_cast_rows(int precision, const float* src, int rows, int cols, int lds_, void* dst, int ld, float p_drop,
                     uint64_t seed, uint32_t site, const float* vscale, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || ld < cols || ld % 4) return TIMHIP_EINVAL;
  const uint32_t thr = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  dim3 grid((ld / 4 + 255) / 256, rows);
  DISPATCH_T(precision, hipLaunchKernelGGL(cast_rows_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, src, rows,
                                           cols, lds_, (T*)dst, ld, thr, scale, seed, site, vscale));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_dropout_rows_bwd(const float* g, int rows, int cols, int ldg, float* dx, int ldx, float p_drop,
                            uint64_t seed, uint32_t site, void* stream) {
  if (!g || !dx || rows <= 0 || cols <= 0) return TIMHIP_EINVAL;
  const uint32_t thr = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  dim3 grid(((cols + 3) / 4 + 255) / 256, rows);
  hipLaunchKernelGGL(drop_bwd_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, rows, cols, ldg, dx, ldx, thr,
                     scale, seed, site);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_dropout_mask(uint64_t seed, uint32_t site, float p, int rows, int cols, uint8_t* out, void* stream) {
  if (!out || rows <= 0 || cols <= 0) return TIMHIP_EINVAL;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, seed, site,
                     drop_threshold(p), rows, cols, out);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_layernorm_fwd(int precision, const float* y, int rows, int cols, int ldy, int act, const float* w,
                         const float* b, float* x_f32, int ldx, void* x_T, int ldt, float* stats, void* stream) {
  return tim_layernorm_fwd(precision, y, rows, cols, ldy, act, w, b, x_f32, ldx, x_T, ldt, stats, (hipStream_t)stream);
}

int timhip_layernorm_bwd(int precision, const float* dx, int lddx, const float* y, int ldy, const float* stats,
                         int rows, int cols, int act, const float* w, float* dy_f32, int lddy, void* dy_T, int ldt,
                         float p_drop, uint64_t seed, uint32_t site, float* dgamma, float* dbeta, const float* t_scale,
                         void* stream) {
  return tim_layernorm_bwd(precision, dx, lddx, y, ldy, stats, rows, cols, act, w, dy_f32, lddy, dy_T, ldt, p_drop,
                           seed, site, dgamma, dbeta, nullptr, (hipStream_t)stream, false, t_scale);
}

int timhip_layernorm_fwd2(int precision, const float* y, int rows, int cols, int ldy, int act, const float* w, const float* b,
                          int split_row, const float* w2, const float* b2, float* x_f32, int ldx, void* x_T, int ldt, float* stats,
                          void* stream) {
  return tim_layernorm_fwd(precision, y, rows, cols, ldy, act, w, b, x_f32, ldx, x_T, ldt, stats, (hipStream_t)stream, nullptr, 0, 0.f,
                           0, 0, nullptr, split_row, w2, b2);
}

int timhip_layernorm_bwd2(int precision, const float* dx, int lddx, const float* y, int ldy, const float* stats, int rows, int cols,
                          int act, const float* w, int split_row, const float* w2, float* dy_f32, int lddy, void* dy_T, int ldt,
                          float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, const float* t_scale, void* stream) {
  return tim_layernorm_bwd(precision, dx, lddx, y, ldy, stats, rows, cols, act, w, dy_f32, lddy, dy_T, ldt, 0.f, 0, 0, dgamma, dbeta,
                           nullptr, (hipStream_t)stream, false, t_scale, nullptr, 0, nullptr, 0, split_row, w2, dgamma2, dbeta2);
}

// two feature matrices (the two modality embedders' inputs: different widths, different dropout sites) cast in one launch
int timhip_cast_rows_pair(int precision, const float* const* src, const int* cols, void* const* dst, const int* ld, int rows,
                          float p_drop, uint64_t seed, const uint32_t* sites, void* stream) {
  if (!src || !cols || !dst || !ld || !sites || rows <= 0) return TIMHIP_EINVAL;
  CastPair cp;
  int maxq = 0;
  for (int i = 0; i < 2; ++i) {
    if (!src[i] || !dst[i] || cols[i] <= 0 || ld[i] < cols[i] || ld[i] % 4) return TIMHIP_EINVAL;
    cp.src[i] = src[i]; cp.dst[i] = dst[i]; cp.cols[i] = cols[i]; cp.ld[i] = ld[i]; cp.site[i] = sites[i];
    maxq = ld[i] / 4 > maxq ? ld[i] / 4 : maxq;
  }
  const uint32_t thr = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  dim3 grid((maxq + 255) / 256, rows, 2);
  DISPATCH_T(precision, hipLaunchKernelGGL(cast_rows_pair_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, cp, rows, thr, scale,
                                           TimSeed(seed)));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_time_l1_fwd(int precision, const float* times, int rows, int d, const float* w, const float* b, void* h,
                       int ld, void* stream) {
  if (!times || !w || !b || !h || rows <= 0 || ld < d) return TIMHIP_EINVAL;
  const int rpb = rows >= 4096 ? 16 : (rows >= 512 ? 4 : 1);
  dim3 grid((rows + rpb - 1) / rpb);
  DISPATCH_T(precision, hipLaunchKernelGGL(time_l1_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, times,
                                           rows, d, w, b, (T*)h, ld, rpb, 0));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_time_l1_fwd_split3(int precision, const float* times, int rows, int d, const float* w, const float* b, void* h3,
                              int ld, void* stream) {
  if (!times || !w || !b || !h3 || rows <= 0 || ld < d || (ld % 64) || !h16_storage(precision)) return TIMHIP_EINVAL;
  const int rpb = rows >= 4096 ? 16 : (rows >= 512 ? 4 : 1);
  dim3 grid((rows + rpb - 1) / rpb);
  DISPATCH_H16(precision, hipLaunchKernelGGL(time_l1_fwd_kernel<HT>, grid, dim3(256), 0, (hipStream_t)stream, times,
                                             rows, d, w, b, (HT*)h3, ld, rpb, 1));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_time_l1_bwd(int precision, const float* times, int rows, int d, const float* w, const void* dh, int ld,
                       float* dw, float* db, float* dt, const float* out_scale, void* stream) {
  if (!times || !w || !dh || !dw || !db || rows <= 0) return TIMHIP_EINVAL;
  // every block ends with 3*d memory-side atomics onto the same addresses, and they are what the launch costs: 8000 rows at
  // 8 / 16 / 32 / 64 / 128 / 256 / 512 rows per block: 199 / 99 / 52 / 29 / 20 / 21 / 33 us
  const int rpb = rows >= 2048 ? 128 : 32;   // (C1: 3840 rows ran 120 blocks of 32 rows: 26 us)
  dim3 grid((rows + rpb - 1) / rpb);
  DISPATCH_T(precision, hipLaunchKernelGGL(time_l1_bwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, times,
                                           rows, d, w, (const T*)dh, ld, dw, db, dt, rpb, out_scale));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

static int assemble_fwd_launch(int precision, const TimSeqRow* rows, int B, int S, int d, const float* e0, const float* e1,
                               int n_e_rows, const SeqVecs& sv, const float* te, int T_, float p_seq_drop, uint64_t seed,
                               uint32_t site, float* x, void* x_T, void* stream) {
  if (!rows || !te || !x || !x_T || B <= 0 || S <= 0 || d % 4) return TIMHIP_EINVAL;
  const uint32_t thr = p_seq_drop > 0.f ? drop_threshold(p_seq_drop) : 0u;
  const float scale = p_seq_drop > 0.f ? 1.f / (1.f - p_seq_drop) : 1.f;
  DISPATCH_T(precision, hipLaunchKernelGGL(assemble_fwd_kernel<T>, dim3(B * S), dim3(256), 0, (hipStream_t)stream,
                                           rows, B, S, d, e0, e1, n_e_rows, sv, te, T_, thr, scale, seed, site, x,
                                           (T*)x_T));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_assemble_fwd(int precision, const TimSeqRow* rows, int B, int S, int d, const float* e0, const float* e1,
                        int n_e_rows, const float* cls, const float* te, int T_, const float* mod, float p_seq_drop,
                        uint64_t seed, uint32_t site, float* x, void* x_T, void* stream) {
  SeqVecs sv;   // contiguous [ncls, d] / [nmod, 2 d] buffers: the table points into them
  for (int i = 0; i < SV_CLS; ++i) sv.cls[i] = cls ? cls + (size_t)i * d : nullptr;
  for (int i = 0; i < SV_MOD; ++i) sv.mod[i] = mod ? mod + (size_t)i * 2 * d : nullptr;
  return assemble_fwd_launch(precision, rows, B, S, d, e0, e1, n_e_rows, sv, te, T_, p_seq_drop, seed, site, x, x_T, stream);
}

int timhip_assemble_fwd_p(int precision, const TimSeqRow* rows, int B, int S, int d, const float* e0, const float* e1,
                          int n_e_rows, const float* const* cls, int ncls, const float* te, int T_, const float* const* mod,
                          int nmod, float p_seq_drop, uint64_t seed, uint32_t site, float* x, void* x_T, void* stream) {
  if (ncls < 0 || ncls > SV_CLS || nmod < 0 || nmod > SV_MOD || (ncls && !cls) || (nmod && !mod)) return TIMHIP_EINVAL;
  SeqVecs sv;
  for (int i = 0; i < SV_CLS; ++i) sv.cls[i] = i < ncls ? cls[i] : nullptr;
  for (int i = 0; i < SV_MOD; ++i) sv.mod[i] = i < nmod ? mod[i] : nullptr;
  for (int i = 0; i < ncls; ++i) if (!cls[i] || ((uintptr_t)cls[i] & 15)) return TIMHIP_EALIGN;
  for (int i = 0; i < nmod; ++i) if (!mod[i] || ((uintptr_t)mod[i] & 15)) return TIMHIP_EALIGN;
  return assemble_fwd_launch(precision, rows, B, S, d, e0, e1, n_e_rows, sv, te, T_, p_seq_drop, seed, site, x, x_T, stream);
}

static int assemble_bwd_launch(const TimSeqRow* rows, int B, int S, int d, const float* dx, int n_e_rows, int T_,
                               float p_seq_drop, uint64_t seed, uint32_t site, float* d_e0, float* d_e1, const SeqVecGrads& sg,
                               float* d_te, void* stream) {
  if (!rows || !dx || B <= 0 || S <= 0 || d % 4) return TIMHIP_EINVAL;
  const uint32_t thr = p_seq_drop > 0.f ? drop_threshold(p_seq_drop) : 0u;
  const float scale = p_seq_drop > 0.f ? 1.f / (1.f - p_seq_drop) : 1.f;
  const int G = S >= 256 ? (S + 127) / 128 : 1;   // token rows per block: about 128 row groups, one atomic per group, vector and column
  hipLaunchKernelGGL(assemble_bwd_kernel, dim3((S + G - 1) / G, (2 * d + 255) / 256), dim3(256), 0, (hipStream_t)stream, rows, B, S, d, dx,
                     n_e_rows, thr, scale, seed, site, d_e0, d_e1, sg, G);
  TIM_CHECK_LAUNCH();
  if (d_te) {
    hipLaunchKernelGGL(assemble_bwd_te_kernel, dim3(T_, B >= 32 ? 16 : (B >= 8 ? 4 : 1)), dim3(128), 0, (hipStream_t)stream, rows, B, S, d, dx, T_,
                       thr, scale, seed, site, d_te);
    TIM_CHECK_LAUNCH();
  }
  return TIMHIP_OK;
}

int timhip_assemble_bwd(const TimSeqRow* rows, int B, int S, int d, const float* dx, int n_e_rows, int T_,
                        float p_seq_drop, uint64_t seed, uint32_t site, float* d_e0, float* d_e1, float* d_cls,
                        float* d_te, float* d_mod, void* stream) {
  SeqVecGrads sg;
  for (int i = 0; i < SV_CLS; ++i) sg.cls[i] = d_cls ? d_cls + (size_t)i * d : nullptr;
  for (int i = 0; i < SV_MOD; ++i) sg.mod[i] = d_mod ? d_mod + (size_t)i * 2 * d : nullptr;
  return assemble_bwd_launch(rows, B, S, d, dx, n_e_rows, T_, p_seq_drop, seed, site, d_e0, d_e1, sg, d_te, stream);
}

int timhip_assemble_bwd_p(const TimSeqRow* rows, int B, int S, int d, const float* dx, int n_e_rows, int T_,
                          float p_seq_drop, uint64_t seed, uint32_t site, float* d_e0, float* d_e1, float* const* d_cls, int ncls,
                          float* d_te, float* const* d_mod, int nmod, void* stream) {
  if (ncls < 0 || ncls > SV_CLS || nmod < 0 || nmod > SV_MOD || (ncls && !d_cls) || (nmod && !d_mod)) return TIMHIP_EINVAL;
  SeqVecGrads sg;
  for (int i = 0; i < SV_CLS; ++i) sg.cls[i] = i < ncls ? d_cls[i] : nullptr;
  for (int i = 0; i < SV_MOD; ++i) sg.mod[i] = i < nmod ? d_mod[i] : nullptr;
  return assemble_bwd_launch(rows, B, S, d, dx, n_e_rows, T_, p_seq_drop, seed, site, d_e0, d_e1, sg, d_te, stream);
}

int timhip_gather_rows(int precision, const void* x_T, int B, int S, int E, int s0, int n, void* rows_T, void* stream) {
  if (!x_T || !rows_T || n <= 0 || E % 4) return TIMHIP_EINVAL;
  DISPATCH_T(precision, hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(B * n), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)x_T, B, S, E, s0, n, (T*)rows_T));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

static int fill_ranges(RowRanges& rr, int count, const int* s0, const int* n) {
  if (count < 1 || count > RR_MAX || !s0 || !n) return TIMHIP_EINVAL;
  rr.count = count;
  rr.joff[0] = 0;
  for (int i = 0; i < RR_MAX; ++i) {
    rr.s0[i] = i < count ? s0[i] : 0;
    rr.n[i] = i < count ? n[i] : 0;
    if (i < count && n[i] <= 0) return TIMHIP_EINVAL;
    rr.joff[i + 1] = rr.joff[i] + rr.n[i];
    rr.src[i] = nullptr; rr.dst[i] = nullptr;
  }
  return TIMHIP_OK;
}

int timhip_gather_ranges(int precision, const void* x_T, int B, int S, int E, int count, const int* s0, const int* n,
                         void* const* rows_T, void* stream) {
  if (!x_T || !rows_T || B <= 0 || E % 4) return TIMHIP_EINVAL;
  RowRanges rr;
  int rc = fill_ranges(rr, count, s0, n);
  if (rc) return rc;
  for (int i = 0; i < count; ++i) { if (!rows_T[i]) return TIMHIP_EINVAL; rr.dst[i] = rows_T[i]; }
  DISPATCH_T(precision, hipLaunchKernelGGL(gather_ranges_kernel<T>, dim3(B * rr.joff[count]), dim3(256), 0,
                                           (hipStream_t)stream, (const T*)x_T, B, S, E, rr));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_gather_split3_ranges(int precision, const float* x, int B, int S, int E, int count, const int* s0, const int* n,
                                void* const* rows3_T, void* stream) {
  if (!x || !rows3_T || B <= 0 || E % 64 || !h16_storage(precision)) return TIMHIP_EINVAL;
  RowRanges rr;
  int rc = fill_ranges(rr, count, s0, n);
  if (rc) return rc;
  for (int i = 0; i < count; ++i) { if (!rows3_T[i] || ((uintptr_t)rows3_T[i] & 7)) return TIMHIP_EINVAL; rr.dst[i] = rows3_T[i]; }
  DISPATCH_H16(precision, hipLaunchKernelGGL(gather_split3_ranges_kernel<HT>, dim3(B * rr.joff[count]), dim3(256), 0,
                                             (hipStream_t)stream, x, B, S, E, rr));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_scatter_ranges_add(int B, int S, int E, int count, const int* s0, const int* n, const float* const* d_rows,
                              float* dx, void* stream) {
  if (!d_rows || !dx || B <= 0 || E % 4) return TIMHIP_EINVAL;
  RowRanges rr;
  int rc = fill_ranges(rr, count, s0, n);
  if (rc) return rc;
  for (int i = 0; i < count; ++i) { if (!d_rows[i]) return TIMHIP_EINVAL; rr.src[i] = d_rows[i]; }
  hipLaunchKernelGGL(scatter_ranges_add_kernel, dim3(B * rr.joff[count]), dim3(256), 0, (hipStream_t)stream, B, S, E, dx, rr);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_sigmoid_bwd_rows(int precision, const float* grad_out, const float* y, int rows, int cols, void* dst, int ld,
                            const float* scale, void* stream) {
  if (!grad_out || !y || !dst || rows <= 0 || cols <= 0 || ld < cols) return TIMHIP_EINVAL;
  const long long n = (long long)rows * ld;
  DISPATCH_T(precision, hipLaunchKernelGGL(sigmoid_bwd_rows_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                           (hipStream_t)stream, grad_out, y, rows, cols, (T*)dst, ld, scale));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_dx_init(int B, int S, int F, int E, const float* feats_cot, int count, const int* s0, const int* n,
                   const float* const* d_rows, float* dx, void* stream) {
  return timhip_dx_init_slabs(B, S, F, E, feats_cot, count, s0, n, d_rows, nullptr, dx, stream);
}

int timhip_dx_init_slabs(int B, int S, int F, int E, const float* feats_cot, int count, const int* s0, const int* n,
                         const float* const* d_rows, const int* nslab, float* dx, void* stream) {
  if (!dx || B <= 0 || S <= 0 || F < 0 || F > S || E % 4 || count < 0 || count > RR_MAX) return TIMHIP_EINVAL;
  RowRanges rr;
  rr.count = 0; rr.joff[0] = 0;
  for (int i = 0; i < RR_MAX; ++i) { rr.s0[i] = 0; rr.n[i] = 0; rr.joff[i + 1] = 0; rr.src[i] = nullptr; rr.dst[i] = nullptr; }
  if (count > 0) {
    if (!s0 || !n || !d_rows) return TIMHIP_EINVAL;
    for (int i = 0; i < count; ++i) {
      if (!d_rows[i] || n[i] <= 0 || s0[i] < F || s0[i] + n[i] > S) return TIMHIP_EINVAL;
      for (int j = 0; j < i; ++j)
        if (s0[i] < s0[j] + n[j] && s0[j] < s0[i] + n[i]) return TIMHIP_EINVAL;   // overlapping ranges: the caller adds instead
      rr.s0[i] = s0[i]; rr.n[i] = n[i]; rr.src[i] = d_rows[i];
      rr.joff[i] = nslab ? nslab[i] : 1;
      if (rr.joff[i] < 1 || rr.joff[i] > 16) return TIMHIP_EINVAL;
    }
    rr.count = count;
  }
  hipLaunchKernelGGL(dx_init_kernel, dim3(B * S), dim3(256), 0, (hipStream_t)stream, B, S, F, E, feats_cot, dx, rr);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_cast_rows_many(int precision, int count, const float* const* src, const int* rows, const int* cols,
                          void* const* dst, const int* ld, const float* scale, void* stream) {
  if (count < 1 || count > RR_MAX || !src || !rows || !cols || !dst || !ld) return TIMHIP_EINVAL;
  CastMany cm;
  cm.vscale = scale;
  int maxr = 0, maxld = 0;
  for (int i = 0; i < RR_MAX; ++i) {
    const bool on = i < count;
    if (on && (!src[i] || !dst[i] || rows[i] <= 0 || cols[i] <= 0 || ld[i] < cols[i])) return TIMHIP_EINVAL;
    cm.src[i] = on ? src[i] : nullptr; cm.dst[i] = on ? dst[i] : nullptr;
    cm.rows[i] = on ? rows[i] : 0; cm.cols[i] = on ? cols[i] : 0; cm.ld[i] = on ? ld[i] : 0;
    if (on && rows[i] > maxr) maxr = rows[i];
    if (on && ld[i] > maxld) maxld = ld[i];
  }
  dim3 grid((maxld + 255) / 256 > 8 ? 8 : (maxld + 255) / 256, maxr, count);
  DISPATCH_T(precision, hipLaunchKernelGGL(cast_many_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, cm));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_ln_partials_reduce(const float* partials, int nsets, int rows, int cols, float* const* dgamma,
                              float* const* dbeta, void* stream) {
  if (!partials || nsets < 1 || nsets > LNS_MAX || rows <= 0 || cols <= 0 || !dgamma || !dbeta) return TIMHIP_EINVAL;
  LnSets ls;
  for (int i = 0; i < LNS_MAX; ++i) { ls.dg[i] = i < nsets ? dgamma[i] : nullptr; ls.db[i] = i < nsets ? dbeta[i] : nullptr; }
  const int nblk = tim_layernorm_bwd_blocks(rows);
  const long long stride = (long long)(tim_layernorm_bwd_ws(rows, cols) / sizeof(float));
  hipLaunchKernelGGL(partial_colsum_sets_kernel, dim3((2 * cols + 255) / 256, 32, nsets), dim3(256), 0, (hipStream_t)stream,
                     partials, stride, nblk, cols, ls);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_scatter_rows_add(const float* d_rows, int B, int S, int E, int s0, int n, float* dx, void* stream) {
  if (!d_rows || !dx || n <= 0 || E % 4) return TIMHIP_EINVAL;
  hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(B * n), dim3(256), 0, (hipStream_t)stream, d_rows, B, S, E, s0, n, dx);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_split3_many(int precision, int count, const float* const* src, const int* rows, const int* cols, const int* lds_,
                       void* const* dst, const int* ldd, int mode, int relu, void* stream) {
  if (!h16_storage(precision)) return TIMHIP_EUNSUPPORTED;
  if (count < 1 || count > RR_MAX || !src || !rows || !cols || !lds_ || !dst || !ldd || (mode != 0 && mode != 1)) return TIMHIP_EINVAL;
  Split3Many sm;
  sm.mode = mode; sm.relu = relu ? 1 : 0;
  int maxr = 0, maxc = 0;
  for (int i = 0; i < RR_MAX; ++i) {
    const bool on = i < count;
    if (on && (!src[i] || !dst[i] || rows[i] <= 0 || cols[i] <= 0 || lds_[i] < cols[i] || ldd[i] % 192 || ldd[i] / 3 < cols[i]))
      return TIMHIP_EINVAL;
    sm.src[i] = on ? src[i] : nullptr; sm.dst[i] = on ? dst[i] : nullptr;
    sm.rows[i] = on ? rows[i] : 0; sm.cols[i] = on ? cols[i] : 0; sm.lds[i] = on ? lds_[i] : 0; sm.ldd[i] = on ? ldd[i] : 0;
    if (on && rows[i] > maxr) maxr = rows[i];
    if (on && ldd[i] / 3 > maxc) maxc = ldd[i] / 3;
  }
  long long blocks = ((long long)maxr * (maxc / 4) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
  for (int i = 0; i < count; ++i)
    if ((ldd[i] & 3) || (((uintptr_t)dst[i]) & 7)) return TIMHIP_EALIGN;
  dim3 grid((unsigned)blocks, 1, count);
  DISPATCH_H16(precision, hipLaunchKernelGGL(split3_kernel<HT>, grid, dim3(256), 0, (hipStream_t)stream, sm));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_dp_reduce(int wire_bf16, const void* recv, int world, long long per, float scale, void* out, void* stream) {
  if (!recv || !out || world < 1 || per <= 0 || (per & 3)) return TIMHIP_EINVAL;
  if ((((uintptr_t)recv | (uintptr_t)out) & 15) != 0) return TIMHIP_EALIGN;
  long long blocks = (per / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  if (wire_bf16)
    hipLaunchKernelGGL(dp_reduce_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)recv, world,
                       per, scale, (bf16_t*)out);
  else
    hipLaunchKernelGGL(dp_reduce_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)recv, world,
                       per, scale, (float*)out);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_grad_scale(const float* const* cot, const long long* counts, int n, float target, float* out, void* stream) {
  if (!cot || !counts || n < 1 || n > GS_MAX || !out || !(target > 0.f)) return TIMHIP_EINVAL;
  GsList gl;
  gl.count = n;
  long long total = 0;
  for (int i = 0; i < GS_MAX; ++i) {
    gl.p[i] = i < n ? cot[i] : nullptr; gl.n[i] = i < n ? counts[i] : 0;
    if (i < n && (!cot[i] || counts[i] < 0)) return TIMHIP_EINVAL;
    total += gl.n[i];
  }
  long long blocks = (total / 4 + 1023) / 1024 / 2;   // ~two 16-KiB chunks per block
  // every block ends with two atomics on the same two words and they, not the bytes, set the launch time: 42 MB of
  // cotangents with at most 4096 / 1024 / 512 / 256 / 128 / 64 blocks: 30 / 27 / 21 / 16 / 14 / 16 us
  blocks = blocks < 1 ? 1 : (blocks > 128 ? 128 : blocks);
  hipLaunchKernelGGL(grad_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gl, target, out);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

}  // extern "C"
