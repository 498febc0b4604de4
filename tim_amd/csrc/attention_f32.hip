// Structured attention in exact fp32 arithmetic on the f32 matrix cores (v_mfma_f32_32x32x2_f32) - the attention of the
// `fp32` and `bf16x3` precision modes (activations stored in fp32).  Same math and the same dropout stream as the bf16
// kernels of attention_mfma.hip / attention_bwd2.hip (transformers.py:92-111 under TIM's mask: keys = the F feature tokens
// plus, for a query row, the row itself), same three-kernel structure:
//   attn_fwd_f32       : lane = token row; S^T = K Q^T per 32-key block, register softmax, O^T = V^T P^T.
//   attn_bwd_rows_f32  : recomputes S^T and dP^T = V dO^T, forms dS, writes dQ (+ self terms) and hands dS and the dropped
//                        probabilities P~ to the key-side kernel through an fp32 scratch [B*H][S][FP].
//   attn_bwd_keys_f32  : dK = dS^T Q, dV = P~^T dO for the feature keys (contraction over the token rows).
// MFMA operand convention (k = 2 per instruction, k slot = lane >> 5): the contraction order is a free permutation, so
//   * head-dim contractions give slot g the half [g*DH/2, (g+1)*DH/2): a lane's Q / dO operand values are 64 CONTIGUOUS
//     floats of its row (16-byte global loads), and the K / V operand is a 16-byte LDS read per four instructions;
//   * key contractions use key(step (q,t), g) = 32 jb + 8q + 4g + t, which is exactly the accumulator register 4q + t of the
//     S^T tile in the same lane: probabilities feed the second product straight from the accumulators.
// One LDS tile [FP][DH] fp32 (16-byte chunks XOR-swizzled with the row) holds K, then V (then K again in the backward).
#include "common.h"

namespace {

struct AttnArgsF {
  int S, F, E, H, LP;
  float scale;
  uint32_t thr; float dscale; TimSeed seed; uint32_t site;
};

__device__ __forceinline__ void keep4f(const AttnArgsF& a, uint64_t rowbase, int key, float& k0, float& k1, float& k2,
                                       float& k3) {
  drop_mask4(a.seed, a.site, (rowbase + (uint64_t)key) >> 2, a.thr, a.dscale, k0, k1, k2, k3);
}
__device__ __forceinline__ float keep1f(const AttnArgsF& a, uint64_t rowbase, int key) {
  float k[4];
  drop_mask4(a.seed, a.site, (rowbase + (uint64_t)key) >> 2, a.thr, a.dscale, k[0], k[1], k[2], k[3]);
  const int c = (int)((rowbase + (uint64_t)key) & 3);
  return c == 0 ? k[0] : (c == 1 ? k[1] : (c == 2 ? k[2] : k[3]));
}

// byte offset of element (row, col) of the [rows][DH] fp32 tile
template <int DH>
__device__ __forceinline__ int toff(int row, int col) {
  constexpr int NCH = DH / 4;
  return row * (DH * 4) + ((((col >> 2) ^ row) & (NCH - 1)) << 4) + ((col & 3) << 2);
}

// stage rows [0, nvalid) of a [.., ld] fp32 matrix (zero rows up to nrows) into the swizzled tile
template <int DH>
__device__ __forceinline__ void stage_f32(char* tile, const float* src, size_t ld, int nrows, int nvalid, int tid, int nthreads) {
  constexpr int NCH = DH / 4, UN = 4;
  for (int i0 = tid; i0 < nrows * NCH; i0 += nthreads * UN) {
    float4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = i0 + u * nthreads, row = idx / NCH, c = idx % NCH;
      v[u] = (idx < nrows * NCH && row < nvalid) ? *reinterpret_cast<const float4*>(src + (size_t)row * ld + c * 4)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = i0 + u * nthreads;
      if (idx < nrows * NCH) *reinterpret_cast<float4*>(tile + toff<DH>(idx / NCH, (idx % NCH) * 4)) = v[u];
    }
  }
}

// acc[jb][key][row] += sum_dh T[32 jb + key][dh] * x[row][dh] for all key blocks; the lane's half row x (slot g) is streamed
// from global memory four floats at a time, so no register copy of the row is kept
template <int DH, int NJB>
__device__ __forceinline__ void dot_all(f32x16_t (&acc)[NJB], const char* tile, int li, int g, const float* xrow) {
  const float* xp = xrow + g * (DH / 2);
#pragma unroll 4
  for (int j = 0; j < DH / 8; ++j) {
    const float4 x = *reinterpret_cast<const float4*>(xp + 4 * j);
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
      const float4 t = *reinterpret_cast<const float4*>(tile + toff<DH>(jb * 32 + li, g * (DH / 2) + 4 * j));
      acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(t.x, x.x, acc[jb], 0, 0, 0);
      acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(t.y, x.y, acc[jb], 0, 0, 0);
      acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(t.z, x.z, acc[jb], 0, 0, 0);
      acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(t.w, x.w, acc[jb], 0, 0, 0);
    }
  }
}
// sum over the lane's half (slot g) of x[row][dh] * y[row][dh]; add the partner lane's half with a shuffle
template <int DH>
__device__ __forceinline__ float dot_rows(const float* x, const float* y, int g) {
  float t = 0.f;
#pragma unroll 4
  for (int j = 0; j < DH / 8; ++j) {
    const float4 a = *reinterpret_cast<const float4*>(x + g * (DH / 2) + 4 * j);
    const float4 b = *reinterpret_cast<const float4*>(y + g * (DH / 2) + 4 * j);
    t += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  return t + __shfl_xor(t, 32, 64);
}

// out[dh][row] += sum_key T[key][32 db + dh] * w[row][key] over the keys of block jb, w = an S^T-layout accumulator tile
template <int DH>
__device__ __forceinline__ void mix_tile(f32x16_t& out, const char* tile, int jb, int db, int li, int g, const f32x16_t& w) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = jb * 32 + 8 * (r >> 2) + 4 * g + (r & 3);
    const float t = *reinterpret_cast<const float*>(tile + toff<DH>(key, 32 * db + li));
    out = __builtin_amdgcn_mfma_f32_32x32x2f32(t, w[r], out, 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int DH, int NJB>
__global__ __launch_bounds__(512) void attn_fwd_f32(const float* __restrict__ qkv, float* __restrict__ o,
                                                    float* __restrict__ lse, AttnArgsF a) {
  constexpr int FP = NJB * 32, NDB = DH / 32;
  extern __shared__ __attribute__((aligned(16))) char tile[];
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int S = a.S, F = a.F, E = a.E;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int li = lane & 31, g = lane >> 5;
  const size_t ld = (size_t)3 * E;
  const float* base = qkv + (size_t)b * S * ld + (size_t)h * DH;
  const int nrb = (S + 31) >> 5, npass = (nrb + nwaves - 1) / nwaves;
  for (int pass = 0; pass < npass; ++pass) {     // every wave takes part in every barrier
    const int rb = pass * nwaves + wave;
    const bool work = rb < nrb;
    const int row = rb * 32 + li;
    const bool valid = work && row < S;
    const int rowc = valid ? row : S - 1;
    const bool isq = rowc >= F;
    const float* qp = base + (size_t)rowc * ld;
    if (pass > 0) __syncthreads();               // everybody is done with the V tile of the previous pass
    stage_f32<DH>(tile, base + E, ld, FP, F, tid, blockDim.x);
    __syncthreads();
    f32x16_t sc[NJB];
    float sself = -INFINITY;
    {
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[jb][r] = 0.f;
      dot_all<DH, NJB>(sc, tile, li, g, qp);
      const float t = dot_rows<DH>(qp, qp + E, g);     // shuffles run in every lane
      if (isq) sself = t;
    }
    {
      constexpr int jb = NJB - 1;   // only the last key block can hold padded keys
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g >= F) sc[jb][r] = -INFINITY;
    }
    float mx = sself;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[jb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf((sc[jb][r] - mx) * a.scale);
        sc[jb][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float pself_un = isq ? expf((sself - mx) * a.scale) : 0.f;
    sum += pself_un;
    const float inv = 1.f / sum;
    if (valid && g == 0) lse[((size_t)b * a.H + h) * S + row] = mx * a.scale + logf(sum);
    const uint64_t rowbase = (((uint64_t)b * a.H + h) * S + rowc) * (uint64_t)a.LP;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float k0 = 1.f, k1 = 1.f, k2 = 1.f, k3 = 1.f;
        if (a.thr != 0u) keep4f(a, rowbase, jb * 32 + 8 * q + 4 * g, k0, k1, k2, k3);
        sc[jb][4 * q] *= inv * k0; sc[jb][4 * q + 1] *= inv * k1; sc[jb][4 * q + 2] *= inv * k2; sc[jb][4 * q + 3] *= inv * k3;
      }
    float pself = pself_un * inv;
    if (isq && a.thr != 0u) pself *= keep1f(a, rowbase, F);

    __syncthreads();                             // K tile no longer needed
    stage_f32<DH>(tile, base + 2 * E, ld, FP, F, tid, blockDim.x);
    __syncthreads();
    float* op = o + ((size_t)b * S + rowc) * E + (size_t)h * DH;
#pragma unroll 1
    for (int db = 0; db < NDB; ++db) {
      f32x16_t oa;
#pragma unroll
      for (int r = 0; r < 16; ++r) oa[r] = 0.f;
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) mix_tile<DH>(oa, tile, jb, db, li, g, sc[jb]);
      if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dh = 32 * db + 8 * q + 4 * g;
          float v0 = oa[4 * q], v1 = oa[4 * q + 1], v2 = oa[4 * q + 2], v3 = oa[4 * q + 3];
          if (isq) {
            const float4 sv = *reinterpret_cast<const float4*>(qp + 2 * E + dh);
            v0 = fmaf(pself, sv.x, v0); v1 = fmaf(pself, sv.y, v1); v2 = fmaf(pself, sv.z, v2); v3 = fmaf(pself, sv.w, v3);
          }
          *reinterpret_cast<float4*>(op + dh) = make_float4(v0, v1, v2, v3);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int DH, int NJB>
__global__ __launch_bounds__(512) void attn_bwd_rows_f32(const float* __restrict__ qkv, const float* __restrict__ o,
                                                         const float* __restrict__ lse, const float* __restrict__ d_o,
                                                         float* __restrict__ dqkv, float* __restrict__ dS_scr,
                                                         float* __restrict__ Pt_scr, AttnArgsF a) {
  constexpr int FP = NJB * 32, NDB = DH / 32;
  extern __shared__ __attribute__((aligned(16))) char tile[];
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int S = a.S, F = a.F, E = a.E;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int li = lane & 31, g = lane >> 5;
  const size_t ld = (size_t)3 * E;
  const float* base = qkv + (size_t)b * S * ld + (size_t)h * DH;
  float* dbase = dqkv + (size_t)b * S * ld + (size_t)h * DH;
  const float* dobase = d_o + (size_t)b * S * E + (size_t)h * DH;
  const float* obase = o + (size_t)b * S * E + (size_t)h * DH;
  const float* lsebase = lse + ((size_t)b * a.H + h) * S;
  float* dSs = dS_scr + (size_t)blockIdx.x * S * FP;
  float* Pts = Pt_scr + (size_t)blockIdx.x * S * FP;
  const int nrb = (S + 31) >> 5, npass = (nrb + nwaves - 1) / nwaves;
  for (int pass = 0; pass < npass; ++pass) {
    const int rb = pass * nwaves + wave;
    const bool work = rb < nrb;
    const int row = rb * 32 + li;
    const bool valid = work && row < S;
    const int rowc = valid ? row : S - 1;
    const bool isq = rowc >= F;
    const float* qp = base + (size_t)rowc * ld;
    const float* dop = dobase + (size_t)rowc * E;
    const float* op = obase + (size_t)rowc * E;
    const float l = lsebase[rowc];
    const uint64_t rowbase = (((uint64_t)b * a.H + h) * S + rowc) * (uint64_t)a.LP;
    // ---- K tile: S^T
    if (pass > 0) __syncthreads();
    stage_f32<DH>(tile, base + E, ld, FP, F, tid, blockDim.x);
    __syncthreads();
    f32x16_t sc[NJB], dp[NJB];
    float ds_self = 0.f, pt_self = 0.f, delta;
    {
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[jb][r] = 0.f;
      dot_all<DH, NJB>(sc, tile, li, g, qp);
      ds_self = dot_rows<DH>(qp, qp + E, g);             // raw self score (used by query rows only)
    }
    // ---- V tile: dP^T
    __syncthreads();
    stage_f32<DH>(tile, base + 2 * E, ld, FP, F, tid, blockDim.x);
    __syncthreads();
    {
      delta = dot_rows<DH>(dop, op, g);
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[jb][r] = 0.f;
      dot_all<DH, NJB>(dp, tile, li, g, dop);
      pt_self = dot_rows<DH>(dop, qp + 2 * E, g);
    }
    if (isq) {
      const float p = expf(ds_self * a.scale - l);
      const float keep = a.thr != 0u ? keep1f(a, rowbase, F) : 1.f;
      ds_self = p * (pt_self * keep - delta) * a.scale;
      pt_self = p * keep;
    } else {
      ds_self = 0.f; pt_self = 0.f;
    }
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float k[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.thr != 0u) keep4f(a, rowbase, jb * 32 + 8 * q + 4 * g, k[0], k[1], k[2], k[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = 4 * q + t;
          const int key = jb * 32 + 8 * q + 4 * g + t;
          const float p = key < F ? expf(sc[jb][r] * a.scale - l) : 0.f;
          sc[jb][r] = p * (dp[jb][r] * k[t] - delta) * a.scale;   // dS
          dp[jb][r] = p * k[t];                                    // P~
        }
        if (valid) {
          const size_t so = (size_t)row * FP + jb * 32 + 8 * q + 4 * g;
          *reinterpret_cast<float4*>(dSs + so) = make_float4(sc[jb][4 * q], sc[jb][4 * q + 1], sc[jb][4 * q + 2], sc[jb][4 * q + 3]);
          *reinterpret_cast<float4*>(Pts + so) = make_float4(dp[jb][4 * q], dp[jb][4 * q + 1], dp[jb][4 * q + 2], dp[jb][4 * q + 3]);
        }
      }
    // ---- K tile again: dQ^T = K^T dS^T
    __syncthreads();
    stage_f32<DH>(tile, base + E, ld, FP, F, tid, blockDim.x);
    __syncthreads();
    float* dq = dbase + (size_t)row * ld;
#pragma unroll 1
    for (int db = 0; db < NDB; ++db) {
      f32x16_t qa;
#pragma unroll
      for (int r = 0; r < 16; ++r) qa[r] = 0.f;
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) mix_tile<DH>(qa, tile, jb, db, li, g, sc[jb]);
      if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dh = 32 * db + 8 * q + 4 * g;
          float v0 = qa[4 * q], v1 = qa[4 * q + 1], v2 = qa[4 * q + 2], v3 = qa[4 * q + 3];
          if (isq) {
            const float4 kk = *reinterpret_cast<const float4*>(qp + E + dh);
            const float4 qq = *reinterpret_cast<const float4*>(qp + dh);
            const float4 dd = *reinterpret_cast<const float4*>(dop + dh);
            v0 = fmaf(ds_self, kk.x, v0); v1 = fmaf(ds_self, kk.y, v1); v2 = fmaf(ds_self, kk.z, v2); v3 = fmaf(ds_self, kk.w, v3);
            // a query token's own key / value receive the self term only
            *reinterpret_cast<float4*>(dq + E + dh) = make_float4(ds_self * qq.x, ds_self * qq.y, ds_self * qq.z, ds_self * qq.w);
            *reinterpret_cast<float4*>(dq + 2 * E + dh) = make_float4(pt_self * dd.x, pt_self * dd.y, pt_self * dd.z, pt_self * dd.w);
          }
          *reinterpret_cast<float4*>(dq + dh) = make_float4(v0, v1, v2, v3);
        }
      }
    }
  }
}

// dK / dV of the feature keys: out[key][dh] = sum_row Y[row][key] X[row][dh]
//   blockIdx.y = 0: Y = dS, X = Q -> dK ;  blockIdx.y = 1: Y = P~, X = dO -> dV ;  blockIdx.z = (window, head);
//   blockIdx.x = 128-key tile.  4 waves, each a 64 (dh) x 64 (key) part of the 128 x 128 output.
__global__ __launch_bounds__(256) void attn_bwd_keys_f32(const float* __restrict__ dS_scr, const float* __restrict__ Pt_scr,
                                                         const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                         float* __restrict__ dqkv, int S, int F, int FP, int E, int H,
                                                         int DH) {
  constexpr int WT = 128, WM = 32;
  __shared__ __attribute__((aligned(16))) float sY[WM][WT + 4];
  __shared__ __attribute__((aligned(16))) float sX[WM][WT + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 1, wn = wave & 1, li = lane & 31, g = lane >> 5;
  const int bh = blockIdx.z, b = bh / H, h = bh % H, prod = blockIdx.y;
  const float* Y = (prod ? Pt_scr : dS_scr) + (size_t)bh * S * FP;
  const float* X = prod ? d_o + (size_t)b * S * E + (size_t)h * DH : qkv + (size_t)b * S * 3 * E + (size_t)h * DH;
  const int ldx = prod ? E : 3 * E;
  float* out = dqkv + (size_t)b * S * 3 * E + (prod ? 2 * E : E) + (size_t)h * DH;
  const int ldo = 3 * E, n0 = blockIdx.x * WT;
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int m0 = 0; m0 < S; m0 += WM) {
    __syncthreads();
    for (int idx = tid; idx < WM * (WT / 4); idx += 256) {
      const int r = idx / (WT / 4), c = (idx % (WT / 4)) * 4, row = m0 + r;
      float4 y = make_float4(0.f, 0.f, 0.f, 0.f), x = y;
      if (row < S) {
        if (n0 + c < FP) y = *reinterpret_cast<const float4*>(Y + (size_t)row * FP + n0 + c);
        if (c < DH) x = *reinterpret_cast<const float4*>(X + (size_t)row * ldx + c);
      }
      *reinterpret_cast<float4*>(&sY[r][c]) = y;
      *reinterpret_cast<float4*>(&sX[r][c]) = x;
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < WM / 2; ++s) {
      const int r = 2 * s + g;
      float xa[2], yb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) xa[i] = sX[r][wk * 64 + i * 32 + li];
#pragma unroll
      for (int j = 0; j < 2; ++j) yb[j] = sY[r][wn * 64 + j * 32 + li];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i], yb[j], acc[i][j], 0, 0, 0);
    }
  }
  // D[i = dh][j = key]: lane owns one key row of dK / dV, 4 consecutive dh per quad
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + li;
    if (n >= F) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = wk * 64 + i * 32 + 8 * q + 4 * g;
        if (k + 3 < DH)
          *reinterpret_cast<float4*>(out + (size_t)n * ldo + k) =
              make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
      }
  }
}

AttnArgsF make_args_f(const TimDesc& d) {
  AttnArgsF a;
  a.S = d.S; a.F = d.F; a.E = d.E; a.H = d.H; a.LP = round_up(d.F + 1, 8);
  a.scale = 1.f / sqrtf((float)(d.E / d.H));
  a.thr = d.p_drop > 0.f ? drop_threshold(d.p_drop) : 0u;
  a.dscale = d.p_drop > 0.f ? 1.f / (1.f - d.p_drop) : 1.f;
  a.seed = d.seed; a.site = layer_site(d.layer, SITE_L_ATTN);
  return a;
}

static inline int waves_for(int S) { const int n = (S + 31) / 32; return n < 1 ? 1 : (n > 8 ? 8 : n); }

template <int DH, int NJB>
int launch_fwd(const TimDesc& d, const void* qkv, void* o, float* lse, hipStream_t s) {
  const size_t lds = (size_t)NJB * 32 * DH * 4;
  (void)hipFuncSetAttribute((const void*)attn_fwd_f32<DH, NJB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_fwd_f32<DH, NJB>), dim3(d.B * d.H), dim3(64 * waves_for(d.S)), lds, s, (const float*)qkv,
                     (float*)o, lse, make_args_f(d));
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}

template <int DH, int NJB>
int launch_bwd(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, void* ws,
               hipStream_t s) {
  const int FP = NJB * 32;
  float* dS = (float*)ws;
  float* Pt = dS + (size_t)d.B * d.H * d.S * FP;
  const size_t lds = (size_t)FP * DH * 4;
  (void)hipFuncSetAttribute((const void*)attn_bwd_rows_f32<DH, NJB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_bwd_rows_f32<DH, NJB>), dim3(d.B * d.H), dim3(64 * waves_for(d.S)), lds, s, (const float*)qkv,
                     (const float*)o, lse, (const float*)d_o, (float*)dqkv, dS, Pt, make_args_f(d));
  if (hipGetLastError() != hipSuccess) return TIMHIP_ELAUNCH;
  hipLaunchKernelGGL(attn_bwd_keys_f32, dim3((d.F + 127) / 128, 2, d.B * d.H), dim3(256), 0, s, dS, Pt, (const float*)qkv,
                     (const float*)d_o, (float*)dqkv, d.S, d.F, FP, d.E, d.H, DH);
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}

}  // namespace

size_t tim_attention_f32_bwd_ws(const TimDesc& d) {
  return (size_t)2 * d.B * d.H * d.S * round_up(d.F, 32) * sizeof(float);
}

#define F32_DISPATCH(CALL)                                                                                             \
  const int DHv = d.E / d.H, NJBv = (d.F + 31) / 32;                                                                   \
  if (!f32_storage(d.precision) || (d.E % 4) != 0 || d.B * d.H > 65535) return TIMHIP_EUNSUPPORTED;                    \
  if (DHv == 128) {                                                                                                    \
    switch (NJBv) { case 1: CALL(128, 1); case 2: CALL(128, 2); case 3: CALL(128, 3); case 4: CALL(128, 4); case 5: CALL(128, 5); default: break; } \
  } else if (DHv == 64) {                                                                                              \
    switch (NJBv) { case 1: CALL(64, 1); case 2: CALL(64, 2); case 4: CALL(64, 4); default: break; }                   \
  } else if (DHv == 32) {                                                                                              \
    switch (NJBv) { case 1: CALL(32, 1); case 2: CALL(32, 2); default: break; }                                        \
  }                                                                                                                    \
  return TIMHIP_EUNSUPPORTED;

int tim_attention_fwd_f32(const TimDesc& d, const void* qkv, void* o, float* lse, hipStream_t s) {
#define FWD(DHc, NJBc) return launch_fwd<DHc, NJBc>(d, qkv, o, lse, s)
  F32_DISPATCH(FWD)
#undef FWD
}

int tim_attention_bwd_f32(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv,
                          void* ws, size_t ws_bytes, hipStream_t s) {
  if (!ws || ws_bytes < tim_attention_f32_bwd_ws(d)) return TIMHIP_EUNSUPPORTED;
#define BWD(DHc, NJBc) return launch_bwd<DHc, NJBc>(d, qkv, o, lse, d_o, dqkv, ws, s)
  F32_DISPATCH(BWD)
#undef BWD
}
