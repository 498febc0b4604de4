// Fused GEMM epilogues and tile-order helpers shared by the NT GEMM kernels (gemm.hip: 2-blocks-per-CU kernel, fp32 and
// split-operand kernels; gemm_pp.hip: the one-block-per-CU ping-pong kernel).
#pragma once
#include <utility>
#include "common.h"

namespace {


// ---------------------------------------------------------------------------
// epilogues
// ---------------------------------------------------------------------------
struct EpiDev {
  void* out0; void* out1;
  const float* bias; const float* res; const void* aux;
  int ld0, ld1, ldres, ldaux;
  uint32_t thr; float scale; uint32_t site; TimSeed seed;
  const uint8_t* mask; int ldmask;   // precomputed keep-bits of the dropout site (row stride in bytes), or NULL: draw here
  // DROP_RES_F32 only: the residual is LayerNorm(res) - res holds the PRE-norm rows, ln_stats their (mean, rstd) pairs, ln_w /
  // ln_b the affine parameters - so the normalised fp32 rows never have to exist in memory (NULL: res is used as it is)
  const float* ln_stats; const float* ln_w; const float* ln_b;
  const float* acc_scale;  // device scalar or NULL: factor on the accumulators (1 / gradient scale where fp16 gradient operands end)
  int vec;  // all leading dims % 4 == 0 and pointers 16 B aligned
  int vec8; // the operand-dtype outputs / aux of this epilogue also allow 8-element (16-byte) accesses
  long long slab_stride;  // EPI_STORE_F32 with split-K: split z writes out0 + z*slab_stride (elements)
  int pair;     // dropout + residual epilogue of the one-block-per-CU kernels: lane pairs share their Philox calls (TIMHIP_EPI_PAIR, default 1)
  int a_wrap;   // 0, or the number of 64-deep contraction steps after which the A operand repeats (TimEpi.a_wrap_k / 64): the
                // product of an activation matrix with a weight matrix split into [hi | lo] column blocks reads A twice
};

// Streaming stores for everything an epilogue writes (round 3): the outputs are read next by another kernel, never by this one,
// and a 160 x 256 tile's 80-160 KiB per CU would otherwise displace the operand panels the co-running tiles re-read from the
// XCD's L2 (in the step, every output nontemporal against plain stores: 5.306 -> 5.257 ms, NT GEMMs 836 -> 852 TFLOP/s).
// Which outputs (TIMHIP_NT_MASK, A/B builds: bit 0 the 16-bit main output, 1 the second 16-bit output (gelu' x mask / the
// pre-activation), 2 the fp32 outputs), step in ms, three interleaved runs each: only bit 1: 5.182; bits 0 + 1: **5.134**; 0 + 2:
// 5.187; 1 + 2: 5.204; all: 5.152; none: +0.9 % on all.  The fp32 sums (read back by the LayerNorm that follows and as the next
// residual) are better left to the cache policy; the 16-bit streams are not.
#ifndef TIMHIP_NT_MASK
#define TIMHIP_NT_MASK 3
#endif
template <typename T>
__device__ __forceinline__ void nt_store4(T* p, float a, float b, float c, float d) {
  if constexpr ((sizeof(T) == 4 && !(TIMHIP_NT_MASK & 4)) || (sizeof(T) == 2 && !(TIMHIP_NT_MASK & 1))) {
    store4<T>(p, a, b, c, d);
  } else if constexpr (sizeof(T) == 4) {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    f4_t v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<f4_t*>(p));
  } else {
    vec4<T> v;
    v[0] = (T)a; v[1] = (T)b; v[2] = (T)c; v[3] = (T)d;
    __builtin_nontemporal_store(v, reinterpret_cast<vec4<T>*>(p));
  }
}

template <int EPI, typename T>
__device__ __forceinline__ void epi_one(const EpiDev& e, int m, int n, int N, float v, float mask) {
  size_t i0 = (size_t)m * e.ld0 + n;
  if (e.bias && EPI != TIMHIP_EPI_ADD_F32 && EPI != TIMHIP_EPI_DGELU_T && EPI != TIMHIP_EPI_DRELU_T &&
      EPI != TIMHIP_EPI_ATOMIC_F32 && EPI != TIMHIP_EPI_DRELU_F32IN_T && EPI != TIMHIP_EPI_MULAUX_T)
    v += e.bias[n];
  if (EPI == TIMHIP_EPI_STORE_T) {
    ((T*)e.out0)[i0] = OpT<T>::from_f(v);
  } else if (EPI == TIMHIP_EPI_RELU_T) {
    ((T*)e.out0)[i0] = OpT<T>::from_f(fmaxf(v, 0.f));
  } else if (EPI == TIMHIP_EPI_STORE_F32) {
    ((float*)e.out0)[i0] = v;
  } else if (EPI == TIMHIP_EPI_GELU_DROP_T2) {
    ((T*)e.out1)[(size_t)m * e.ld1 + n] = OpT<T>::from_f(v);
    ((T*)e.out0)[i0] = OpT<T>::from_f(gelu_f(v) * mask);
  } else if (EPI == TIMHIP_EPI_GELU_DROP_G2) {
    float gl, dg;
    gelu_both_f(v, gl, dg);
    ((T*)e.out1)[(size_t)m * e.ld1 + n] = OpT<T>::from_f(dg * mask);
    ((T*)e.out0)[i0] = OpT<T>::from_f(gl * mask);
  } else if (EPI == TIMHIP_EPI_MULAUX_T) {
    ((T*)e.out0)[i0] = OpT<T>::from_f(v * OpT<T>::to_f(((const T*)e.aux)[(size_t)m * e.ldaux + n]));
  } else if (EPI == TIMHIP_EPI_DROP_RES_F32) {
    float r = e.res[(size_t)m * e.ldres + n];
    if (e.ln_stats) r = (r - e.ln_stats[2 * m]) * e.ln_stats[2 * m + 1] * e.ln_w[n] + e.ln_b[n];
    ((float*)e.out0)[i0] = r + v * mask;
  } else if (EPI == TIMHIP_EPI_ADD_F32) {
    ((float*)e.out0)[i0] = v + (e.res ? e.res[(size_t)m * e.ldres + n] : 0.f);
  } else if (EPI == TIMHIP_EPI_DGELU_T) {
    float u = OpT<T>::to_f(((const T*)e.aux)[(size_t)m * e.ldaux + n]);
    ((T*)e.out0)[i0] = OpT<T>::from_f(v * mask * gelu_grad_f(u));
  } else if (EPI == TIMHIP_EPI_DRELU_T) {
    float h = OpT<T>::to_f(((const T*)e.aux)[(size_t)m * e.ldaux + n]);
    ((T*)e.out0)[i0] = OpT<T>::from_f(h > 0.f ? v : 0.f);
  } else if (EPI == TIMHIP_EPI_DRELU_F32IN_T) {
    float h = ((const float*)e.aux)[(size_t)m * e.ldaux + n];
    ((T*)e.out0)[i0] = OpT<T>::from_f(h > 0.f ? v : 0.f);
  } else if (EPI == TIMHIP_EPI_ATOMIC_F32) {
    atomicAdd(((float*)e.out0) + i0, v);
  } else if (EPI == TIMHIP_EPI_SIGMOID_F32) {
    ((float*)e.out0)[i0] = 1.f / (1.f + __expf(-v));
  } else if (EPI == TIMHIP_EPI_RELU_SPLIT3_T) {
    const float r = fmaxf(v, 0.f);
    const T hi = OpT<T>::from_f(r);
    ((T*)e.out0)[i0] = hi;
    ((T*)e.out0)[i0 + e.ld1] = OpT<T>::from_f(r - OpT<T>::to_f(hi));
    ((T*)e.out0)[i0 + 2 * (size_t)e.ld1] = hi;
  }
}

constexpr bool epi_uses_dropout(int EPI) {
  return EPI == TIMHIP_EPI_GELU_DROP_T2 || EPI == TIMHIP_EPI_DROP_RES_F32 || EPI == TIMHIP_EPI_DGELU_T ||
         EPI == TIMHIP_EPI_GELU_DROP_G2;
}

// 4 consecutive columns n..n+3 of row m
template <int EPI, typename T>
__device__ __forceinline__ void epi_quad(const EpiDev& e, int m, int n, int N, float v0, float v1,
                                         float v2, float v3, bool has_pre = false,
                                         float4 pre = make_float4(0.f, 0.f, 0.f, 0.f), bool has_b = false,
                                         float4 pb = make_float4(0.f, 0.f, 0.f, 0.f), bool has_ln = false,
                                         float2 pst = make_float2(0.f, 1.f),
                                         float4 pg = make_float4(1.f, 1.f, 1.f, 1.f),
                                         float4 pbe = make_float4(0.f, 0.f, 0.f, 0.f), bool has_k = false,
                                         float4 kpre = make_float4(1.f, 1.f, 1.f, 1.f)) {
  float k0 = 1.f, k1 = 1.f, k2 = 1.f, k3 = 1.f;
  if (epi_uses_dropout(EPI) && e.thr != 0u) {
    // element index m*N + n, N % 4 == 0 wherever dropout is applied
    if (has_k) {   // keep factors drawn by the caller (lane pairs sharing the Philox calls of two quads: pp_epilogue)
      k0 = kpre.x; k1 = kpre.y; k2 = kpre.z; k3 = kpre.w;
    } else if (e.mask)
      drop_mask4_bits((uint32_t)e.mask[(size_t)m * e.ldmask + (n >> 3)] >> (n & 4), e.scale, k0, k1, k2, k3);
    else
      drop_mask4(e.seed, e.site, ((uint64_t)m * (uint64_t)N + (uint64_t)n) >> 2, e.thr, e.scale, k0, k1, k2, k3);
  }
  if (e.vec && n + 3 < N) {
    size_t i0 = (size_t)m * e.ld0 + n;
    if (EPI != TIMHIP_EPI_ADD_F32 && EPI != TIMHIP_EPI_DGELU_T && EPI != TIMHIP_EPI_DRELU_T &&
        EPI != TIMHIP_EPI_ATOMIC_F32 && EPI != TIMHIP_EPI_DRELU_F32IN_T && EPI != TIMHIP_EPI_MULAUX_T && e.bias) {
      float4 b = pb;   // this lane's bias columns are the same for every row: fetched once by the caller, or here
      if (!has_b) b = *reinterpret_cast<const float4*>(e.bias + n);
      v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
    }
    if (EPI == TIMHIP_EPI_STORE_T) {
      nt_store4<T>((T*)e.out0 + i0, v0, v1, v2, v3);
    } else if (EPI == TIMHIP_EPI_RELU_T) {
      nt_store4<T>((T*)e.out0 + i0, fmaxf(v0, 0.f), fmaxf(v1, 0.f), fmaxf(v2, 0.f), fmaxf(v3, 0.f));
    } else if (EPI == TIMHIP_EPI_STORE_F32) {
      nt_store4<float>((float*)e.out0 + i0, v0, v1, v2, v3);
    } else if (EPI == TIMHIP_EPI_GELU_DROP_T2) {
      nt_store4<T>((T*)e.out1 + (size_t)m * e.ld1 + n, v0, v1, v2, v3);
      nt_store4<T>((T*)e.out0 + i0, gelu_f(v0) * k0, gelu_f(v1) * k1, gelu_f(v2) * k2, gelu_f(v3) * k3);
    } else if (EPI == TIMHIP_EPI_GELU_DROP_G2) {
      float g0, g1, g2, g3, d0, d1, d2, d3;
      gelu_both_f(v0, g0, d0); gelu_both_f(v1, g1, d1); gelu_both_f(v2, g2, d2); gelu_both_f(v3, g3, d3);
      nt_store4<T>((T*)e.out1 + (size_t)m * e.ld1 + n, d0 * k0, d1 * k1, d2 * k2, d3 * k3);
      nt_store4<T>((T*)e.out0 + i0, g0 * k0, g1 * k1, g2 * k2, g3 * k3);
    } else if (EPI == TIMHIP_EPI_MULAUX_T) {
      float u0, u1, u2, u3;
      load4<T>((const T*)e.aux + (size_t)m * e.ldaux + n, u0, u1, u2, u3);
      nt_store4<T>((T*)e.out0 + i0, v0 * u0, v1 * u1, v2 * u2, v3 * u3);
    } else if (EPI == TIMHIP_EPI_DROP_RES_F32) {
      float4 r = pre;   // fetched by the caller ahead of the stores, or here
      if (!has_pre) r = *reinterpret_cast<const float4*>(e.res + (size_t)m * e.ldres + n);
      if (e.ln_stats) {   // residual = LayerNorm of the fetched pre-norm values
        float2 st = pst;
        float4 g = pg, be = pbe;
        if (!has_ln) {
          st = *reinterpret_cast<const float2*>(e.ln_stats + 2 * (size_t)m);
          g = *reinterpret_cast<const float4*>(e.ln_w + n);
          be = *reinterpret_cast<const float4*>(e.ln_b + n);
        }
        r.x = (r.x - st.x) * st.y * g.x + be.x; r.y = (r.y - st.x) * st.y * g.y + be.y;
        r.z = (r.z - st.x) * st.y * g.z + be.z; r.w = (r.w - st.x) * st.y * g.w + be.w;
      }
      nt_store4<float>((float*)e.out0 + i0, r.x + v0 * k0, r.y + v1 * k1, r.z + v2 * k2, r.w + v3 * k3);
    } else if (EPI == TIMHIP_EPI_ADD_F32) {
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_pre) r = pre;
      else if (e.res) r = *reinterpret_cast<const float4*>(e.res + (size_t)m * e.ldres + n);
      nt_store4<float>((float*)e.out0 + i0, r.x + v0, r.y + v1, r.z + v2, r.w + v3);
    } else if (EPI == TIMHIP_EPI_DGELU_T) {
      float u0, u1, u2, u3;
      load4<T>((const T*)e.aux + (size_t)m * e.ldaux + n, u0, u1, u2, u3);
      nt_store4<T>((T*)e.out0 + i0, v0 * k0 * gelu_grad_f(u0), v1 * k1 * gelu_grad_f(u1),
                v2 * k2 * gelu_grad_f(u2), v3 * k3 * gelu_grad_f(u3));
    } else if (EPI == TIMHIP_EPI_DRELU_T) {
      float u0, u1, u2, u3;
      load4<T>((const T*)e.aux + (size_t)m * e.ldaux + n, u0, u1, u2, u3);
      nt_store4<T>((T*)e.out0 + i0, u0 > 0.f ? v0 : 0.f, u1 > 0.f ? v1 : 0.f, u2 > 0.f ? v2 : 0.f,
                u3 > 0.f ? v3 : 0.f);
    } else if (EPI == TIMHIP_EPI_DRELU_F32IN_T) {
      float4 u = *reinterpret_cast<const float4*>((const float*)e.aux + (size_t)m * e.ldaux + n);
      nt_store4<T>((T*)e.out0 + i0, u.x > 0.f ? v0 : 0.f, u.y > 0.f ? v1 : 0.f, u.z > 0.f ? v2 : 0.f,
                u.w > 0.f ? v3 : 0.f);
    } else if (EPI == TIMHIP_EPI_ATOMIC_F32) {
      float* p = (float*)e.out0 + i0;
      atomicAdd(p, v0); atomicAdd(p + 1, v1); atomicAdd(p + 2, v2); atomicAdd(p + 3, v3);
    } else if (EPI == TIMHIP_EPI_SIGMOID_F32) {
      nt_store4<float>((float*)e.out0 + i0, 1.f / (1.f + __expf(-v0)), 1.f / (1.f + __expf(-v1)),
                    1.f / (1.f + __expf(-v2)), 1.f / (1.f + __expf(-v3)));
    } else if (EPI == TIMHIP_EPI_RELU_SPLIT3_T) {
      const float r0 = fmaxf(v0, 0.f), r1 = fmaxf(v1, 0.f), r2 = fmaxf(v2, 0.f), r3 = fmaxf(v3, 0.f);
      T* o = (T*)e.out0 + i0;
      store4<T>(o, r0, r1, r2, r3);
      store4<T>(o + e.ld1, r0 - OpT<T>::to_f(OpT<T>::from_f(r0)), r1 - OpT<T>::to_f(OpT<T>::from_f(r1)),
                r2 - OpT<T>::to_f(OpT<T>::from_f(r2)), r3 - OpT<T>::to_f(OpT<T>::from_f(r3)));
      store4<T>(o + 2 * (size_t)e.ld1, r0, r1, r2, r3);
    }
  } else {
    if (n < N) epi_one<EPI, T>(e, m, n, N, v0, k0);
    if (n + 1 < N) epi_one<EPI, T>(e, m, n + 1, N, v1, k1);
    if (n + 2 < N) epi_one<EPI, T>(e, m, n + 2, N, v2, k2);
    if (n + 3 < N) epi_one<EPI, T>(e, m, n + 3, N, v3, k3);
  }
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>).  Used where an index into a register array
// (the accumulators) must be a constant even when the body is too large for the unroller's thresholds.
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr bool epi_has_oct(int EPI) {
  return EPI == TIMHIP_EPI_STORE_T || EPI == TIMHIP_EPI_RELU_T || EPI == TIMHIP_EPI_GELU_DROP_T2 ||
         EPI == TIMHIP_EPI_DGELU_T || EPI == TIMHIP_EPI_DRELU_T || EPI == TIMHIP_EPI_GELU_DROP_G2 ||
         EPI == TIMHIP_EPI_MULAUX_T;
}
template <typename HT, bool NT = true>
__device__ __forceinline__ void store8(HT* p, float4 lo, float4 hi) {
  vec8<HT> o;
  o[0] = (HT)lo.x; o[1] = (HT)lo.y; o[2] = (HT)lo.z; o[3] = (HT)lo.w;
  o[4] = (HT)hi.x; o[5] = (HT)hi.y; o[6] = (HT)hi.z; o[7] = (HT)hi.w;
  if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<vec8<HT>*>(p));
  else *reinterpret_cast<vec8<HT>*>(p) = o;
}
// the arithmetic of the bf16-output epilogues on one quad (v in/out; a = the quad's aux values; k = dropout factors)
template <int EPI>
__device__ __forceinline__ float4 epi_math4(float4 v, float4 a, float4 k) {
  if (EPI == TIMHIP_EPI_RELU_T) return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
  if (EPI == TIMHIP_EPI_GELU_DROP_T2)
    return make_float4(gelu_f(v.x) * k.x, gelu_f(v.y) * k.y, gelu_f(v.z) * k.z, gelu_f(v.w) * k.w);
  if (EPI == TIMHIP_EPI_DGELU_T)
    return make_float4(v.x * k.x * gelu_grad_f(a.x), v.y * k.y * gelu_grad_f(a.y), v.z * k.z * gelu_grad_f(a.z),
                       v.w * k.w * gelu_grad_f(a.w));
  if (EPI == TIMHIP_EPI_DRELU_T)
    return make_float4(a.x > 0.f ? v.x : 0.f, a.y > 0.f ? v.y : 0.f, a.z > 0.f ? v.z : 0.f, a.w > 0.f ? v.w : 0.f);
  if (EPI == TIMHIP_EPI_MULAUX_T) return make_float4(v.x * a.x, v.y * a.y, v.z * a.z, v.w * a.w);
  return v;
}
// 8 consecutive columns n..n+7 of row m for the epilogues that write bf16: ONE 16-byte store (and 16-byte aux load)
// per lane instead of two 8-byte ones.  Caller guarantees e.vec8 and n + 7 < N.
template <int EPI, typename HT>
__device__ __forceinline__ void epi_oct(const EpiDev& e, int m, int n, int N, float4 lo, float4 hi, uint32_t byte,
                                        bool has_pre = false, vec8<HT> pre = vec8<HT>{}, bool has_b = false,
                                        float4 pb0 = make_float4(0.f, 0.f, 0.f, 0.f),
                                        float4 pb1 = make_float4(0.f, 0.f, 0.f, 0.f)) {
  float4 klo = make_float4(1.f, 1.f, 1.f, 1.f), khi = klo;
  if (epi_uses_dropout(EPI) && e.thr != 0u) {
    if (e.mask) {   // byte = e.mask[m * ldmask + n / 8], fetched by the caller ahead of the stores (n % 8 == 0 here)
      drop_mask4_bits(byte, e.scale, klo.x, klo.y, klo.z, klo.w);
      drop_mask4_bits(byte >> 4, e.scale, khi.x, khi.y, khi.z, khi.w);
    } else {
      const uint64_t q = ((uint64_t)m * (uint64_t)N + (uint64_t)n) >> 2;
      drop_mask4(e.seed, e.site, q, e.thr, e.scale, klo.x, klo.y, klo.z, klo.w);
      drop_mask4(e.seed, e.site, q + 1, e.thr, e.scale, khi.x, khi.y, khi.z, khi.w);
    }
  }
  const size_t i0 = (size_t)m * e.ld0 + n;
  if ((EPI == TIMHIP_EPI_STORE_T || EPI == TIMHIP_EPI_RELU_T || EPI == TIMHIP_EPI_GELU_DROP_T2 ||
       EPI == TIMHIP_EPI_GELU_DROP_G2) && e.bias) {
    float4 b0 = pb0, b1 = pb1;
    if (!has_b) { b0 = *reinterpret_cast<const float4*>(e.bias + n); b1 = *reinterpret_cast<const float4*>(e.bias + n + 4); }
    lo.x += b0.x; lo.y += b0.y; lo.z += b0.z; lo.w += b0.w; hi.x += b1.x; hi.y += b1.y; hi.z += b1.z; hi.w += b1.w;
  }
  float4 alo = make_float4(0.f, 0.f, 0.f, 0.f), ahi = alo;
  if (EPI == TIMHIP_EPI_DGELU_T || EPI == TIMHIP_EPI_DRELU_T || EPI == TIMHIP_EPI_MULAUX_T) {
    vec8<HT> a = pre;
    if (!has_pre) a = *reinterpret_cast<const vec8<HT>*>((const HT*)e.aux + (size_t)m * e.ldaux + n);
    alo = make_float4((float)a[0], (float)a[1], (float)a[2], (float)a[3]);
    ahi = make_float4((float)a[4], (float)a[5], (float)a[6], (float)a[7]);
  }
  if (EPI == TIMHIP_EPI_GELU_DROP_G2) {
    float4 glo, ghi, dlo, dhi;
    gelu_both_f(lo.x, glo.x, dlo.x); gelu_both_f(lo.y, glo.y, dlo.y); gelu_both_f(lo.z, glo.z, dlo.z); gelu_both_f(lo.w, glo.w, dlo.w);
    gelu_both_f(hi.x, ghi.x, dhi.x); gelu_both_f(hi.y, ghi.y, dhi.y); gelu_both_f(hi.z, ghi.z, dhi.z); gelu_both_f(hi.w, ghi.w, dhi.w);
    store8<HT, (TIMHIP_NT_MASK & 2) != 0>((HT*)e.out1 + (size_t)m * e.ld1 + n, make_float4(dlo.x * klo.x, dlo.y * klo.y, dlo.z * klo.z, dlo.w * klo.w),
           make_float4(dhi.x * khi.x, dhi.y * khi.y, dhi.z * khi.z, dhi.w * khi.w));
    store8<HT, (TIMHIP_NT_MASK & 1) != 0>((HT*)e.out0 + i0, make_float4(glo.x * klo.x, glo.y * klo.y, glo.z * klo.z, glo.w * klo.w),
           make_float4(ghi.x * khi.x, ghi.y * khi.y, ghi.z * khi.z, ghi.w * khi.w));
    return;
  }
  if (EPI == TIMHIP_EPI_GELU_DROP_T2) store8<HT, (TIMHIP_NT_MASK & 2) != 0>((HT*)e.out1 + (size_t)m * e.ld1 + n, lo, hi);
  store8<HT, (TIMHIP_NT_MASK & 1) != 0>((HT*)e.out0 + i0, epi_math4<EPI>(lo, alo, klo), epi_math4<EPI>(hi, ahi, khi));
}

// XCD-aware tile order: block b runs on XCD b % 8 (observed); give each XCD a
// contiguous range of logical tiles so that the tiles_n tiles sharing one A
// row-panel hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr int BK = 64;  // K granularity required of the operands (leading dims are multiples of 64)

template <int BKT> __device__ __forceinline__ int kswz(int row);
template <> __device__ __forceinline__ int kswz<64>(int row) { return (row >> 1) & 7; }
template <> __device__ __forceinline__ int kswz<32>(int row) { return (row >> 2) & 3; }

}  // namespace
