// Structured self-attention of the TIM encoder (tim.py:161-166 mask +
// nn.MultiheadAttention need_weights math path, transformers.py:102):
// token i attends to the F feature tokens of its window and, if it is a query
// token (i >= F), additionally to itself.  The S x S score matrix and the dense
// [B*H, S, S] mask of the reference are never materialised.
//
// This file holds the fp32-arithmetic kernels (operand storage T = fp32 or
// bf16): they are the TIMHIP_PREC_FP32 path and the reference point for the
// MFMA kernels in attention_mfma.hip.
#include "common.h"

namespace {

struct AttnArgs {
  int S, F, E, H, Dh, LP;  // LP = round_up(F + 1, 8): row pitch of the probability dropout stream
  float scale;
  uint32_t thr; float dscale; TimSeed seed; uint32_t site;
};

__device__ __forceinline__ float attn_keep(const AttnArgs& a, int b, int h, int row, int j) {
  if (a.thr == 0u) return 1.f;
  const uint64_t base = (((uint64_t)b * a.H + h) * a.S + row) * (uint64_t)a.LP + (uint64_t)j;
  return drop_mask1(a.seed, a.site, base, a.thr, a.dscale);
}

constexpr int KPL = 3;  // keys per lane: F <= 192

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_simple(const T* __restrict__ qkv, T* __restrict__ o,
                                                       float* __restrict__ lse, AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int Dh = a.Dh, F = a.F, S = a.S, E = a.E;
  const int KS = Dh + (sizeof(T) == 2 ? 2 : 1);
  T* sK = reinterpret_cast<T*>(smem);
  T* sV = sK + (size_t)F * KS;
  float* sQ = reinterpret_cast<float*>(smem + align_up((size_t)2 * F * KS * sizeof(T), 16));
  float* sP = sQ + 4 * Dh;
  const int PP = F + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t ld = (size_t)3 * E;
  const T* base = qkv + (size_t)b * S * ld + (size_t)h * Dh;
  for (int idx = tid; idx < F * Dh; idx += 256) {
    const int j = idx / Dh, c = idx % Dh;
    sK[j * KS + c] = base[(size_t)j * ld + E + c];
    sV[j * KS + c] = base[(size_t)j * ld + 2 * E + c];
  }
  __syncthreads();
  for (int r0 = 0; r0 < S; r0 += 4) {
    const int row = r0 + wave;
    const bool active = row < S;
    const T* qp = base + (size_t)(active ? row : 0) * ld;
    for (int c = lane; c < Dh; c += 64) sQ[wave * Dh + c] = OpT<T>::to_f(qp[c]) * a.scale;
    __syncthreads();
    float sc[KPL];
    float mx = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      const int j = lane + 64 * kk;
      float s = -INFINITY;
      if (j < F) {
        s = 0.f;
        for (int c = 0; c < Dh; ++c) s = fmaf(sQ[wave * Dh + c], OpT<T>::to_f(sK[j * KS + c]), s);
      }
      sc[kk] = s;
      mx = fmaxf(mx, s);
    }
    const bool isq = row >= F;
    float sself = -INFINITY;
    if (isq && active) {
      float t = 0.f;
      for (int c = lane; c < Dh; c += 64) t = fmaf(sQ[wave * Dh + c], OpT<T>::to_f(qp[E + c]), t);
      sself = wave_sum(t);
    }
    mx = fmaxf(wave_max(mx), sself);
    float sum = 0.f;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      sc[kk] = (lane + 64 * kk < F) ? __expf(sc[kk] - mx) : 0.f;
      sum += sc[kk];
    }
    const float pself_un = isq ? __expf(sself - mx) : 0.f;
    sum = wave_sum(sum) + pself_un;
    const float inv = 1.f / sum;
    if (active && lane == 0) lse[((size_t)b * a.H + h) * S + row] = mx + __logf(sum);
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      const int j = lane + 64 * kk;
      if (j < F) sP[wave * PP + j] = sc[kk] * inv * (active ? attn_keep(a, b, h, row, j) : 0.f);
    }
    const float pself = (isq && active) ? pself_un * inv * attn_keep(a, b, h, row, F) : 0.f;
    __syncthreads();
    if (active) {
      for (int c = lane; c < Dh; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < F; ++j) acc = fmaf(sP[wave * PP + j], OpT<T>::to_f(sV[j * KS + c]), acc);
        if (isq) acc = fmaf(pself, OpT<T>::to_f(qp[2 * E + c]), acc);
        o[((size_t)b * S + row) * E + (size_t)h * Dh + c] = OpT<T>::from_f(acc);
      }
    }
    __syncthreads();
  }
}

// Backward, one block per (window, head):
//   phase A (rows over waves): recompute p from lse, dp = dO V^T, ds = p (dp - delta), dq = ds K;
//            the self terms of query rows give their own dk / dv directly.
//   phase B (feature keys over waves): dk_j = sum_i ds_ij q_i, dv_j = sum_i p~_ij dO_i.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_simple(const T* __restrict__ qkv, const T* __restrict__ o,
                                                       const float* __restrict__ lse, const T* __restrict__ d_o,
                                                       T* __restrict__ dqkv, float* __restrict__ ws, AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int Dh = a.Dh, F = a.F, S = a.S, E = a.E;
  const int KS = Dh + (sizeof(T) == 2 ? 2 : 1);
  T* sK = reinterpret_cast<T*>(smem);
  T* sV = sK + (size_t)F * KS;
  float* sQ = reinterpret_cast<float*>(smem + align_up((size_t)2 * F * KS * sizeof(T), 16));
  float* sD = sQ + 4 * Dh;      // dO rows
  float* sP = sD + 4 * Dh;      // ds rows
  const int PP = F + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t ld = (size_t)3 * E;
  const T* base = qkv + (size_t)b * S * ld + (size_t)h * Dh;
  T* dbase = dqkv + (size_t)b * S * ld + (size_t)h * Dh;
  float* DS = ws + (size_t)blockIdx.x * 2 * S * PP;  // [S][PP] ds*scale
  float* PT = DS + (size_t)S * PP;                   // [S][PP] dropped probabilities
  for (int idx = tid; idx < F * Dh; idx += 256) {
    const int j = idx / Dh, c = idx % Dh;
    sK[j * KS + c] = base[(size_t)j * ld + E + c];
    sV[j * KS + c] = base[(size_t)j * ld + 2 * E + c];
  }
  __syncthreads();
  for (int r0 = 0; r0 < S; r0 += 4) {
    const int row = r0 + wave;
    const bool active = row < S;
    const int rr = active ? row : 0;
    const T* qp = base + (size_t)rr * ld;
    const T* dop = d_o + ((size_t)b * S + rr) * E + (size_t)h * Dh;
    const T* op = o + ((size_t)b * S + rr) * E + (size_t)h * Dh;
    float dl = 0.f;
    for (int c = lane; c < Dh; c += 64) {
      const float dv = OpT<T>::to_f(dop[c]);
      sQ[wave * Dh + c] = OpT<T>::to_f(qp[c]) * a.scale;
      sD[wave * Dh + c] = dv;
      dl = fmaf(dv, OpT<T>::to_f(op[c]), dl);
    }
    const float delta = wave_sum(dl);
    __syncthreads();
    const float l = lse[((size_t)b * a.H + h) * S + rr];
    const bool isq = row >= F;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
      const int j = lane + 64 * kk;
      if (j < F) {
        float s = 0.f, dp = 0.f;
        for (int c = 0; c < Dh; ++c) {
          s = fmaf(sQ[wave * Dh + c], OpT<T>::to_f(sK[j * KS + c]), s);
          dp = fmaf(sD[wave * Dh + c], OpT<T>::to_f(sV[j * KS + c]), dp);
        }
        const float p = __expf(s - l);
        const float keep = attn_keep(a, b, h, rr, j);
        const float ds = p * (dp * keep - delta) * a.scale;
        sP[wave * PP + j] = ds;
        if (active) { DS[(size_t)row * PP + j] = ds; PT[(size_t)row * PP + j] = p * keep; }
      }
    }
    float ds_self = 0.f, pt_self = 0.f;
    if (isq && active) {
      float t = 0.f, u = 0.f;
      for (int c = lane; c < Dh; c += 64) {
        t = fmaf(sQ[wave * Dh + c], OpT<T>::to_f(qp[E + c]), t);
        u = fmaf(sD[wave * Dh + c], OpT<T>::to_f(qp[2 * E + c]), u);
      }
      t = wave_sum(t); u = wave_sum(u);
      const float p = __expf(t - l);
      const float keep = attn_keep(a, b, h, row, F);
      ds_self = p * (u * keep - delta) * a.scale;
      pt_self = p * keep;
    }
    __syncthreads();
    if (active) {
      for (int c = lane; c < Dh; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < F; ++j) acc = fmaf(sP[wave * PP + j], OpT<T>::to_f(sK[j * KS + c]), acc);
        if (isq) {
          acc = fmaf(ds_self, OpT<T>::to_f(qp[E + c]), acc);
          // self-only gradients of a query token's own key / value
          dbase[(size_t)row * ld + E + c] = OpT<T>::from_f(ds_self * OpT<T>::to_f(qp[c]));
          dbase[(size_t)row * ld + 2 * E + c] = OpT<T>::from_f(pt_self * sD[wave * Dh + c]);
        }
        dbase[(size_t)row * ld + c] = OpT<T>::from_f(acc);
      }
    }
    __syncthreads();
  }
  // phase B
  __syncthreads();
  for (int j = wave; j < F; j += 4) {
    for (int c = lane; c < Dh; c += 64) {
      float ak = 0.f, av = 0.f;
      for (int i = 0; i < S; ++i) {
        const float ds = DS[(size_t)i * PP + j], pt = PT[(size_t)i * PP + j];
        ak = fmaf(ds, OpT<T>::to_f(base[(size_t)i * ld + c]), ak);
        av = fmaf(pt, OpT<T>::to_f(d_o[((size_t)b * S + i) * E + (size_t)h * Dh + c]), av);
      }
      dbase[(size_t)j * ld + E + c] = OpT<T>::from_f(ak);
      dbase[(size_t)j * ld + 2 * E + c] = OpT<T>::from_f(av);
    }
  }
}

AttnArgs make_args(const TimDesc& d) {
  AttnArgs a;
  a.S = d.S; a.F = d.F; a.E = d.E; a.H = d.H; a.Dh = d.E / d.H; a.LP = round_up(d.F + 1, 8);
  a.scale = 1.f / sqrtf((float)a.Dh);
  a.thr = d.p_drop > 0.f ? drop_threshold(d.p_drop) : 0u;
  a.dscale = d.p_drop > 0.f ? 1.f / (1.f - d.p_drop) : 1.f;
  a.seed = d.seed; a.site = layer_site(d.layer, SITE_L_ATTN);
  return a;
}

size_t simple_lds(const TimDesc& d, int nrowbuf) {
  const int Dh = d.E / d.H;
  const size_t ts = opsize(d.precision);
  const int KS = Dh + (ts == 2 ? 2 : 1);
  return align_up((size_t)2 * d.F * KS * ts, 16) + (size_t)nrowbuf * 4 * Dh * 4 + (size_t)4 * (d.F + 4) * 4;
}

int check_desc(const TimDesc& d) {
  if (d.B <= 0 || d.S <= 0 || d.F <= 0 || d.F > d.S || d.H <= 0 || d.E % d.H) return TIMHIP_EINVAL;
  if (d.F > 64 * KPL) return TIMHIP_EUNSUPPORTED;
  return TIMHIP_OK;
}

}  // namespace

int tim_attention_fwd_mfma(const TimDesc& d, const void* qkv, void* o, float* lse, hipStream_t s, const unsigned long long* kbits);
int tim_attention_bwd_mfma(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o,
                           void* dqkv, hipStream_t s);
int tim_attention_bwd2_mfma(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o,
                            void* dqkv, void* ws, size_t ws_bytes, hipStream_t s, const unsigned long long* kbits);
size_t tim_attention_bwd2_ws(const TimDesc& d);
// attention_f32.hip: exact-fp32 MFMA kernels for the fp32 / bf16x3 modes
int tim_attention_fwd_f32(const TimDesc& d, const void* qkv, void* o, float* lse, hipStream_t s);
int tim_attention_bwd_f32(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv,
                          void* ws, size_t ws_bytes, hipStream_t s);
size_t tim_attention_f32_bwd_ws(const TimDesc& d);

int tim_attention_fwd(const TimDesc& d, const void* qkv, void* o, float* lse, hipStream_t s, const unsigned long long* kbits) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!qkv || !o || !lse) return TIMHIP_EINVAL;
  // (bench.py's non-GEMM brackets: qkv read, o written, in the operand type)
  TimGemmScope timing((double)d.B * d.S * d.E * 4 * (h16_storage(d.precision) ? 2 : 4), s, 1);
  if (h16_storage(d.precision) && !(d.reserved & 1)) {  // reserved bit 0: force the fp32-arithmetic kernels
    rc = tim_attention_fwd_mfma(d, qkv, o, lse, s, kbits);
    if (rc != TIMHIP_EUNSUPPORTED) return rc;
  }
  if (f32_storage(d.precision) && !(d.reserved & 1)) {   // fp32 / bf16x3: f32 matrix cores
    rc = tim_attention_fwd_f32(d, qkv, o, lse, s);
    if (rc != TIMHIP_EUNSUPPORTED) return rc;
  }
  const size_t lds = simple_lds(d, 1);
  if (lds > 160 * 1024) return TIMHIP_EUNSUPPORTED;
  AttnArgs a = make_args(d);
  DISPATCH_T(d.precision,
    (void)hipFuncSetAttribute((const void*)attn_fwd_simple<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_fwd_simple<T>, dim3(d.B * d.H), dim3(256), lds, s, (const T*)qkv, (T*)o, lse, a));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

size_t tim_attention_bwd_ws(const TimDesc& d) {
  const size_t simple = (size_t)d.B * d.H * 2 * d.S * (d.F + 4) * sizeof(float);
  const size_t two = tim_attention_bwd2_ws(d);
  const size_t f32 = f32_storage(d.precision) ? tim_attention_f32_bwd_ws(d) : 0;
  const size_t m = simple > two ? simple : two;
  return m > f32 ? m : f32;
}

int tim_attention_bwd(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o,
                      void* dqkv, void* ws, size_t ws_bytes, hipStream_t s, const unsigned long long* kbits) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!qkv || !o || !lse || !d_o || !dqkv || !ws) return TIMHIP_EINVAL;
  // (qkv, o, dO read; dqkv written)
  TimGemmScope timing((double)d.B * d.S * d.E * 8 * (h16_storage(d.precision) ? 2 : 4), s, 1);
  if (h16_storage(d.precision) && !(d.reserved & 1)) {
    if (!(d.reserved & 2)) {  // reserved bit 1: force the single-kernel MFMA backward
      rc = tim_attention_bwd2_mfma(d, qkv, o, lse, d_o, dqkv, ws, ws_bytes, s, kbits);
      if (rc != TIMHIP_EUNSUPPORTED) return rc;
    }
    rc = tim_attention_bwd_mfma(d, qkv, o, lse, d_o, dqkv, s);
    if (rc != TIMHIP_EUNSUPPORTED) return rc;
  }
  if (f32_storage(d.precision) && !(d.reserved & 1)) {
    rc = tim_attention_bwd_f32(d, qkv, o, lse, d_o, dqkv, ws, ws_bytes, s);
    if (rc != TIMHIP_EUNSUPPORTED) return rc;
  }
  if (ws_bytes < tim_attention_bwd_ws(d)) return TIMHIP_EWORKSPACE;
  const size_t lds = simple_lds(d, 2);
  if (lds > 160 * 1024) return TIMHIP_EUNSUPPORTED;
  AttnArgs a = make_args(d);
  DISPATCH_T(d.precision,
    (void)hipFuncSetAttribute((const void*)attn_bwd_simple<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_bwd_simple<T>, dim3(d.B * d.H), dim3(256), lds, s, (const T*)qkv, (const T*)o, lse, (const T*)d_o,
                       (T*)dqkv, (float*)ws, a));
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

extern "C" {
int timhip_attention_fwd(const TimDesc* d, const void* qkv, void* o, float* lse, void* stream) {
  if (!d) return TIMHIP_EINVAL;
  return tim_attention_fwd(*d, qkv, o, lse, (hipStream_t)stream);
}
int timhip_attention_bwd(const TimDesc* d, const void* qkv, const void* o, const float* lse, const void* d_o,
                         void* dqkv, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d) return TIMHIP_EINVAL;
  return tim_attention_bwd(*d, qkv, o, lse, d_o, dqkv, workspace, workspace_bytes, (hipStream_t)stream);
}
size_t timhip_attention_bwd_workspace_bytes(const TimDesc* d) { return d ? tim_attention_bwd_ws(*d) : 0; }
}
