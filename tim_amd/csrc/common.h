// Shared device/host helpers for libtimhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/timhip.h"

// 16-bit MFMA operand types: HT = bf16_t (TIMHIP_PREC_BF16) or f16_t (TIMHIP_PREC_F16).  Every 16-bit kernel is a template
// over H; the two differ in the MFMA opcode and the f32 <-> H conversions only (same fragment layouts, same LDS images).
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
template <typename T> using vec8 = T __attribute__((ext_vector_type(8)));
template <typename T> using vec4 = T __attribute__((ext_vector_type(4)));
typedef vec8<bf16_t> bf16x8_t;
typedef vec4<bf16_t> bf16x4_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

#define TIM_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return TIMHIP_ELAUNCH; \
  } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ----------------------------------------------------------------------------
// operand-type traits
// ----------------------------------------------------------------------------
template <typename T> struct OpT;
template <> struct OpT<float> {
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct OpT<bf16_t> {
  static __device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
  static __device__ __forceinline__ bf16_t from_f(float v) { return (bf16_t)v; }
};

template <> struct OpT<f16_t> {
  static __device__ __forceinline__ float to_f(f16_t v) { return (float)v; }
  static __device__ __forceinline__ f16_t from_f(float v) { return (f16_t)v; }
};

// v_mfma_f32_32x32x16_{bf16,f16}: D[32x32] += A[32x16] B[16x32], 8 operand values per lane, fp32 accumulate
template <typename HT> __device__ __forceinline__ f32x16_t mfma16(vec8<HT> a, vec8<HT> b, f32x16_t c);
template <> __device__ __forceinline__ f32x16_t mfma16<bf16_t>(vec8<bf16_t> a, vec8<bf16_t> b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16_t mfma16<f16_t>(vec8<f16_t> a, vec8<f16_t> b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// v_mfma_f32_16x16x32_{bf16,f16}: D[16x16] += A[16x32] B[32x16]; lane l holds A[l & 15][8 (l >> 4) .. +7], the same of B^T,
// and D[4 (l >> 4) + r][l & 15], r = 0..3
template <typename HT> __device__ __forceinline__ f32x4_t mfma16x16(vec8<HT> a, vec8<HT> b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mfma16x16<bf16_t>(vec8<bf16_t> a, vec8<bf16_t> b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma16x16<f16_t>(vec8<f16_t> a, vec8<f16_t> b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  bf16x4_t v;
  v[0] = (bf16_t)a; v[1] = (bf16_t)b; v[2] = (bf16_t)c; v[3] = (bf16_t)d;
  *reinterpret_cast<bf16x4_t*>(p) = v;
}
template <>
__device__ __forceinline__ void store4<f16_t>(f16_t* p, float a, float b, float c, float d) {
  vec4<f16_t> v;
  v[0] = (f16_t)a; v[1] = (f16_t)b; v[2] = (f16_t)c; v[3] = (f16_t)d;
  *reinterpret_cast<vec4<f16_t>*>(p) = v;
}
template <typename T>
__device__ __forceinline__ void load4(const T* p, float& a, float& b, float& c, float& d);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float& a, float& b, float& c, float& d) {
  float4 v = *reinterpret_cast<const float4*>(p);
  a = v.x; b = v.y; c = v.z; d = v.w;
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float& a, float& b, float& c, float& d) {
  bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
  a = (float)v[0]; b = (float)v[1]; c = (float)v[2]; d = (float)v[3];
}

template <>
__device__ __forceinline__ void load4<f16_t>(const f16_t* p, float& a, float& b, float& c, float& d) {
  vec4<f16_t> v = *reinterpret_cast<const vec4<f16_t>*>(p);
  a = (float)v[0]; b = (float)v[1]; c = (float)v[2]; d = (float)v[3];
}

// ----------------------------------------------------------------------------
// Philox4x32-7 counter RNG (7 rounds: the fewest that pass BigCrush, Salmon et al. 2011): dropout masks are a pure function of
// (seed, site, element index) so the backward regenerates them.
// ----------------------------------------------------------------------------
struct Philox4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
#ifdef __HIP_DEVICE_COMPILE__
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

constexpr int PHILOX_ROUNDS = 7;
__host__ __device__ __forceinline__ Philox4 philox4x32_7(uint64_t seed, uint32_t site, uint64_t ctr) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = site, c3 = 0x7149u;
#pragma unroll
  for (int r = 0; r < PHILOX_ROUNDS; ++r) {
    // one 64-bit product per multiplier: v_mad_u64_u32 yields the high and the low word together (separate mul_hi / mul_lo
    // are two quarter-rate instructions; tools/px/philox_mad.hip: 590 -> 780 G Philox/s, same bits)
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// Dropout seed as the kernels see it: the launch-time seed plus, when the caller registered one (timhip_dropout_salt), a
// 64-bit salt read from device memory at run time.  A step captured in a HIP graph bakes its launch arguments in; the salt
// is what lets every replay draw fresh masks (the host bumps the device word between replays, or inside the graph).
extern const unsigned long long* tim_salt_ptr;  // host variable holding a device pointer (api.hip)
struct TimSeed {
  uint64_t base;
  const unsigned long long* salt;
  __host__ __device__ TimSeed() : base(0), salt(nullptr) {}
  __host__ TimSeed(uint64_t s) : base(s), salt(tim_salt_ptr) {}
  __device__ __forceinline__ operator uint64_t() const { return salt ? base + (uint64_t)*salt : base; }
};

// keep-threshold: element kept iff rnd >= thr  (P[drop] = thr / 2^32 = p)
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// Dropout decisions are 16-bit draws: ONE Philox call serves the 8 consecutive elements 8*c .. 8*c+7 (element e of the group
// = bits 16*(e&1) .. +15 of word e>>1; kept iff that halfword >= thr >> 16).  Half the Philox work of 32-bit draws - it was
// ~70 % of the attention kernels' VALU instructions and 8 us of LayerNorm-1 - at a drop probability quantised to 1/65536
// (0.1 -> 0.09999).  Every kernel and timhip_dropout_mask derive their masks from these helpers, so forward, backward and the
// tests' oracle agree by construction.
//
// masks of the 4 consecutive elements whose linear index is 4*q .. 4*q+3 (one half of counter q >> 1)
__device__ __forceinline__ void drop_mask4(uint64_t seed, uint32_t site, uint64_t q, uint32_t thr,
                                           float scale, float& m0, float& m1, float& m2, float& m3) {
  const Philox4 r = philox4x32_7(seed, site, q >> 1);
  const uint32_t a = (q & 1) ? r.z : r.x, b = (q & 1) ? r.w : r.y, t = thr >> 16;
  m0 = (a & 0xffffu) >= t ? scale : 0.f;
  m1 = (a >> 16) >= t ? scale : 0.f;
  m2 = (b & 0xffffu) >= t ? scale : 0.f;
  m3 = (b >> 16) >= t ? scale : 0.f;
}
// the same four factors from two words of a counter somebody else drew (lane pairs of the attention kernels share calls)
__device__ __forceinline__ void drop_mask4_words(uint32_t a, uint32_t b, uint32_t thr, float scale, float& m0, float& m1,
                                                 float& m2, float& m3) {
  const uint32_t t = thr >> 16;
  m0 = (a & 0xffffu) >= t ? scale : 0.f;
  m1 = (a >> 16) >= t ? scale : 0.f;
  m2 = (b & 0xffffu) >= t ? scale : 0.f;
  m3 = (b >> 16) >= t ? scale : 0.f;
}
// factor of the single element with linear index idx
__device__ __forceinline__ float drop_mask1(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thr, float scale) {
  const Philox4 r = philox4x32_7(seed, site, idx >> 3);
  const int e = (int)(idx & 7), wsel = e >> 1;
  const uint32_t w = wsel == 0 ? r.x : (wsel == 1 ? r.y : (wsel == 2 ? r.z : r.w));
  return ((w >> (16 * (e & 1))) & 0xffffu) >= (thr >> 16) ? scale : 0.f;
}
// keep-bits of the 8 consecutive elements 8*c .. 8*c+7 (bit e = element 8*c + e kept)
__device__ __forceinline__ uint32_t drop_bits8(uint64_t seed, uint32_t site, uint64_t c, uint32_t thr) {
  const Philox4 r = philox4x32_7(seed, site, c);
  const uint32_t t = thr >> 16;
  return ((r.x & 0xffffu) >= t ? 1u : 0u) | ((r.x >> 16) >= t ? 2u : 0u) | ((r.y & 0xffffu) >= t ? 4u : 0u) |
         ((r.y >> 16) >= t ? 8u : 0u) | ((r.z & 0xffffu) >= t ? 16u : 0u) | ((r.z >> 16) >= t ? 32u : 0u) |
         ((r.w & 0xffffu) >= t ? 64u : 0u) | ((r.w >> 16) >= t ? 128u : 0u);
}

// keep-bits of the 32 consecutive elements 4*q0 .. 4*q0+31 (q0 a multiple of 2; bit i = element 4*q0 + i kept): the packed
// form of drop_mask4, produced ahead of time by a memory-bound kernel with idle VALU (LayerNorm forward) for a GEMM epilogue
// that would otherwise spend 15-18 us per launch drawing the same numbers while the matrix pipes wait
__device__ __forceinline__ uint32_t drop_bits32(uint64_t seed, uint32_t site, uint64_t q0, uint32_t thr) {
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) bits |= drop_bits8(seed, site, (q0 >> 1) + j, thr) << (8 * j);
  return bits;
}
// factors of 4 consecutive elements from their keep-bits (low 4 bits of `nib`)
__device__ __forceinline__ void drop_mask4_bits(uint32_t nib, float scale, float& m0, float& m1, float& m2, float& m3) {
  m0 = (nib & 1u) ? scale : 0.f;
  m1 = (nib & 2u) ? scale : 0.f;
  m2 = (nib & 4u) ? scale : 0.f;
  m3 = (nib & 8u) ? scale : 0.f;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute: the "already done" flag of a launcher is kept per
// device (a process that drives a second GPU would otherwise skip the call there and fail its launches)
struct PerDeviceOnce {
  bool done[32] = {};
  bool first() {   // true the first time it is asked on the current device
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 ? 0 : (dev > 31 ? 31 : dev);
    if (done[dev]) return false;
    done[dev] = true;
    return true;
  }
};

// dropout site ids (stream id = site; per-layer sites add 16*layer)
enum {
  SITE_FEAT_V = 1, SITE_FEAT_A = 2, SITE_SEQ = 3,
  SITE_L_BASE = 16, SITE_L_ATTN = 0, SITE_L_DROP1 = 1, SITE_L_FFN = 2, SITE_L_DROP2 = 3, SITE_L_STRIDE = 8
};
static inline uint32_t layer_site(int layer, int which) {
  return SITE_L_BASE + (uint32_t)layer * SITE_L_STRIDE + (uint32_t)which;
}

// ----------------------------------------------------------------------------
// math
// ----------------------------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7): 1 rcp + 1 exp + 5 fma instead of the
// ~40-instruction libm erff; the epilogues that apply GELU run once per output element and their
// VALU time is otherwise comparable to the MFMA main loop of a K=1024 tile.
// Returns erf(x/sqrt2) given x, and e = exp(-x*x/2) for reuse by the derivative.
__device__ __forceinline__ float erf_half_f(float x, float& e) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  e = __expf(-0.5f * x * x);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) {
  float e;
  return 0.5f * x * (1.0f + erf_half_f(x, e));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float kInvSqrt2Pi = 0.3989422804014327f;
  float e;
  const float er = erf_half_f(x, e);
  return fmaf(x * kInvSqrt2Pi, e, 0.5f * (1.0f + er));
}

// gelu(x) and gelu'(x) from one erf / exp evaluation
__device__ __forceinline__ void gelu_both_f(float x, float& gl, float& dg) {
  const float kInvSqrt2Pi = 0.3989422804014327f;
  float e;
  const float ph = 0.5f * (1.0f + erf_half_f(x, e));   // Phi(x)
  gl = x * ph;
  dg = fmaf(x * kInvSqrt2Pi, e, ph);
}

// A/B knobs of the launchers, read from the environment ONCE (first launch) instead of per launch - the step issues 144
// launches and the host is within 36 % of being the bottleneck.  timhip_reload_env() (test hook, include/timhip.h) re-reads
// them; the knobs marked (T) select kernels that exist only in a TUNING=1 build of the library and are ignored otherwise.
struct TimKnobs {
  int gemm_pp;        // TIMHIP_GEMM_PP      0: no one-block-per-CU NT kernels at all (the two-blocks-per-CU kernels of gemm.hip)
  int gemm_ld;        // TIMHIP_GEMM_LD      0: the 8-wave ping-pong kernel instead of loader waves + L2 prefetch
  int gemm_pf;        // TIMHIP_GEMM_PF      L2 prefetch distance in stages (default 4, 0: off)
  int gemm_pf_mode;   // TIMHIP_GEMM_PF_MODE 1: a tile touches its share of the XCD's lines, 2: all its lines
  int gemm_pf_mr;     // TIMHIP_GEMM_PF_MR   prefetch distance of multi-round shapes run one tile per block (default 0)
  int gemm_ldp;       // TIMHIP_GEMM_LDP     0: one tile per block where the default walks 2-4
  int gemm_ld1;       // TIMHIP_GEMM_LD1 (T) 1: one barrier per contraction step
  int gemm_pt;        // TIMHIP_GEMM_PT  (T) 1: 8-wave persistent-tile kernel
  int gemm_dg;        // TIMHIP_GEMM_DG  (T) 1: dual-group persistent kernel;  gemm_dg_offset: TIMHIP_GEMM_DG_OFFSET
  int gemm_dg_offset;
  int fuse_ln;        // TIMHIP_FUSE_LN  (T) 1: residual + LayerNorm inside the out-projection / linear2 epilogue
  int fuse_ln_spin;   // TIMHIP_FUSE_LN_SPIN
  int wgrad_pp;       // TIMHIP_WGRAD_PP     0: no one-block-per-CU weight-gradient grid
  int wgrad_ld;       // TIMHIP_WGRAD_LD     0: its 8-wave merged-phase form
  int wgrad_pf;       // TIMHIP_WGRAD_PF     its L2 prefetch distance (default 4)
  int wgrad_p8_ph;    // TIMHIP_WGRAD_P8_PH  phases per contraction step of the eight-phase weight-gradient kernel: 2 (32 MFMAs each, default) or 4 (16 each, as first written)
  int wgrad_p8;       // TIMHIP_WGRAD_P8     0: no eight-phase 256 x 256 weight-gradient grid (two layers per launch)
  int attn_waves;     // TIMHIP_ATTN_WAVES   waves per attention block (0: by shape)
  int attn_fused;     // TIMHIP_ATTN_FUSED   0: two-kernel attention backward
  int ln_rpb;         // TIMHIP_LN_RPB       rows per LayerNorm-backward block (0: by shape)
  int gemm_p8_ph;     // TIMHIP_GEMM_P8_PH   phases per contraction step of the eight-phase NT kernel: 2 (4 TM MFMAs each, default) or 4 (2 TM each, as first written)
  int gemm_small_w8;  // TIMHIP_GEMM_SMALL_W8   0: four waves instead of eight (2 x 4 of 32 x 32) on the small-problem GEMM's 64 x 128 tile where four stages are taken
  int gemm_small_nst; // TIMHIP_GEMM_SMALL_NST  LDS stages of the small-problem GEMM instances (0: by the block count; 1 / 2: two, as before round 6)
  int ln_fwd_rpb;     // TIMHIP_LN_FWD_RPB   rows per block of the 8-columns-per-lane LayerNorm forward (0: by the row count)
  int ln_rpb_small;   // TIMHIP_LN_RPB_SMALL rows per block of the LayerNorm-backward launches that end in atomics (0: as TIMHIP_LN_RPB)
  int gemm_tmw;       // TIMHIP_GEMM_TMW     5 / 4: force the 160- / 128-row tile of the loader-wave NT kernels (0: by shape)
  int attn_ks;        // TIMHIP_ATTN_KS      0: fused attention backward with the one-wave-per-row-block phase 1 (default 1: key-split)
  int gemm_pp_min;    // TIMHIP_GEMM_PP_MIN_TILES  fewest 160 x 256 tiles the one-block-per-CU NT kernels are used for (default 192)
  int attn_split_min; // TIMHIP_ATTN_SPLIT_MIN  attention forward, B * H < 128: fewest row blocks per workgroup when a (window, head) is split (default 4)
  int epi_pair;       // TIMHIP_EPI_PAIR     dropout + residual NT epilogue: lane pairs share the Philox calls of their keep factors (default 1)
  int ln_pair;        // TIMHIP_LN_PAIR      LayerNorm backward: lane pairs share the Philox calls of their dropout keep factors (default 1)
  int gemm_p8;        // TIMHIP_GEMM_P8      eight-phase 256 / 320 x 256 tiles for the multi-round NT shapes: 0 off, 1 by shape (default), 2 + the mulaux epilogue, 8 / 10 forced
};
const TimKnobs& tim_knobs();

// Non-finite watch of the fp16 gradient path.  `out_scale` (where a kernel takes one) points at word 1 of the block
// timhip_grad_scale writes: {S, 1/S, scratch, scratch, FLAG, 0, 0, 0}.  Every kernel that writes FINAL fp32 gradients
// through out_scale folds what it writes into `chk` (0 * v stays 0 unless v is inf / nan, then chk is nan for good) and ORs
// the flag word once per lane that saw one: what GradScaler's inf check (reference scripts/train.py:351,357-363) looks
// for, found where the values are produced instead of in a pass over 233 MB of gradients.
__device__ __forceinline__ void nf_note(float& chk, float v) { chk = fmaf(v, 0.f, chk); }
__device__ __forceinline__ void nf_commit(const float* out_scale, float chk) {
  if (out_scale != nullptr && chk != chk) atomicOr(reinterpret_cast<unsigned*>(const_cast<float*>(out_scale)) + 3, 1u);
}

// Wave-wide reductions without the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32: six dependent LDS round trips per
// reduction): four DPP row rotations leave every 16-lane row's result in all of its lanes, four v_readlane + three VALU ops
// combine the rows.  Every lane of the wave must be active (as for the shuffles these replace).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_f(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0x128>(v); v += dpp_f<0x124>(v); v += dpp_f<0x122>(v); v += dpp_f<0x121>(v);   // row_ror 8, 4, 2, 1
  return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0x128>(v)); v = fmaxf(v, dpp_f<0x124>(v)); v = fmaxf(v, dpp_f<0x122>(v)); v = fmaxf(v, dpp_f<0x121>(v));
  return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}

// ----------------------------------------------------------------------------
// LDS-DMA (global_load_lds_dwordx4) issued from inline asm: hipcc does not count it, so it inserts no
// conservative `s_waitcnt vmcnt(0)` in front of later LDS reads; the kernel waits with
// glds_wait<N>() itself before the barrier that publishes the stage.  LDS destination =
// wave-uniform base + lane*16.
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void glds16(const void* gptr, uint32_t lds_base_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gptr), "s"(lds_base_uniform)
      : "memory");
}
// make a wave-uniform pointer provably uniform (SGPR pair) for the "s" constraints below
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const void*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
// SGPR base (wave-uniform 64-bit address) + per-lane 32-bit byte offset: the per-step address update is
// one scalar add instead of a 64-bit vector add per load.
__device__ __forceinline__ void glds16_s(const void* sbase, uint32_t voff, uint32_t lds_base_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_base_uniform)
      : "memory");
}
// four loads to LDS base, base+1 KiB, +2 KiB, +3 KiB with one M0 save/restore
__device__ __forceinline__ void glds16_x4(const void* sbase, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3,
                                          uint32_t lds_base_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %3, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %4, %5\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(sbase), "s"(lds_base_uniform)
      : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void glds16_xn(const void* sbase_, const uint32_t (&off)[N], uint32_t lds_base_uniform) {
  const void* sbase = uniform_ptr(sbase_);
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int i = 0; i < N; i += 4) glds16_x4(sbase, off[i], off[i + 1], off[i + 2], off[i + 3], lds_base_uniform + i * 1024);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) glds16_s(sbase, off[i], lds_base_uniform + i * 1024);
  }
}
template <int N>
__device__ __forceinline__ void glds_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ----------------------------------------------------------------------------
// internal C++ launchers shared between translation units
// ----------------------------------------------------------------------------
// timing.hip: brackets a GEMM-family launch with HIP events while timhip_gemm_timing_start() is armed (bench.py roofline)
struct TimGemmScope {
  TimGemmScope(double flops, hipStream_t s, int family = 0);   // family 1: attention, 2: LayerNorm (work = algorithmic bytes)
  ~TimGemmScope();
  int slot;
  hipStream_t stream;
};

int tim_gemm_nt(int precision, int epi, const void* A, int lda, const void* B, int ldb, int M, int N,
                int K, const TimEpi& e, int splitk, hipStream_t s);
int tim_gemm_nt_group(int precision, int epi, const TimGemmItem* items, int n, hipStream_t s);
// gemm_pp.hip: the one-block-per-CU ping-pong kernel for the encoder-layer shapes (epi_dev: the caller's EpiDev)
bool tim_gemm_pp_wins(int M, int N, int K, int splitk);
int tim_gemm_p8_choice(int epi, int M, int N, int K);   // gemm_pp.hip: 8 / 10 = the eight-phase kernel's row tile for this launch, 0 = not taken
// gemm_pp.hip: out-projection / linear2 with the LayerNorm that follows fused into the epilogue (gemm_nt_ldln_kernel)
struct TimLnFuse {
  void* xt; int ldt; float* xf; int ldx; float* stats; const float* g; const float* b;            // LayerNorm outputs / parameters
  uint8_t* mask_out; int mask_cols; float mask_p; uint64_t mask_seed; uint32_t mask_site;           // keep-bits as tim_layernorm_fwd draws them
};
int tim_gemm_nt_pp_ln(int precision, const void* A, int lda, const void* B, int ldb, int M, int N, int K, const void* epi_dev,
                      const TimLnFuse& lf, const uint32_t** fail, hipStream_t s);
int tim_gemm_nt_fuse_ln(int precision, const void* A, int lda, const void* B, int ldb, int M, int N, int K, const TimEpi& te,
                        const TimLnFuse& lf, const uint32_t** fail, hipStream_t s);
int tim_gemm_nt_pp(int precision, int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K, const void* epi_dev,
                   hipStream_t s);
int tim_transpose(int precision, const void* src, int rows, int cols, int lds, void* dst, int ld,
                  float* colsum, hipStream_t s);
int tim_slab_reduce(const float* slab, long long n, int nslab, float* dW, hipStream_t s);
size_t tim_wgrad_tn_ws(int Nout, int Kout, int M);
// out_scale: NULL, or a device scalar the written gradients are multiplied by (TIMHIP_PREC_F16 stores its gradient operands
// multiplied by S and passes 1/S here, see timhip_grad_scale)
int tim_wgrad_tn_h16(int precision, const void* dY, int ldy, int Nout, const void* X, int ldx, int Kout, int M, float* dW,
                     float* db, void* ws, size_t ws_bytes, hipStream_t s, int accumulate = 1, const float* out_scale = nullptr);
size_t tim_wgrad_group_ws(const TimWgradItem* it, int n, int M);
int tim_wgrad_group_splits(const TimWgradItem* it, int n, int M);
int tim_wgrad_group_h16(int precision, const TimWgradItem* it, int n, int M, int accumulate, void* ws, size_t ws_bytes,
                        const float* out_scale, hipStream_t s);
// wgrad_pp.hip: the grouped weight gradients as one-block-per-CU ping-pong blocks (no split of the contraction, no workspace)
bool tim_wgrad_pp_wins(const TimWgradItem* it, int n, int M);
int tim_wgrad_group_pp(int precision, const TimWgradItem* it, int n, int M, int accumulate, const float* out_scale, hipStream_t s);
// its eight-phase form (256 x 256 tiles; wins for groups of whole rounds of 256 such tiles: two encoder layers at C2a)
bool tim_wgrad_p8_wins(const TimWgradItem* it, int n, int M);
int tim_wgrad_group_p8(int precision, const TimWgradItem* it, int n, int M, int accumulate, const float* out_scale, hipStream_t s);
int tim_colsum(int precision, const void* src, int rows, int cols, int ld, float* out, hipStream_t s);
// mask_out != NULL: additionally writes the dropout keep-bits of a [rows, mask_cols] site (1 bit per element, row stride
// mask_cols / 8 bytes, element index r * mask_cols + c as in the GEMM epilogues) - see drop_bits32
int tim_layernorm_fwd(int precision, const float* y, int rows, int cols, int ldy, int act,
                      const float* w, const float* b, float* x_f32, int ldx, void* x_T, int ldt,
                      float* stats, hipStream_t s, uint8_t* mask_out = nullptr, int mask_cols = 0, float mask_p = 0.f,
                      uint64_t mask_seed = 0, uint32_t mask_site = 0,
                      const uint32_t* run_if = nullptr,   // run_if: control words of gemm_nt_ldln_kernel; the launch does nothing unless a tile of the launch in front of it timed out
                      int split_row = 0, const float* w2 = nullptr, const float* b2 = nullptr);   // rows >= split_row (> 0) use w2 / b2
int tim_layernorm_bwd(int precision, const float* dx, int lddx, const float* y, int ldy,
                      const float* stats, int rows, int cols, int act, const float* w, float* dy_f32,
                      int lddy, void* dy_T, int ldt, float p_drop, uint64_t seed, uint32_t site,
                      float* dgamma, float* dbeta, float* partial_ws, hipStream_t s, bool defer_colsum = false,
                      const float* t_scale = nullptr,   // t_scale: device scalar multiplied into the operand-dtype copy dy_T
                      const void* add_T = nullptr, int ldadd = 0, const float* add_scale = nullptr,   // dx += add_scale * add_T
                      int stream16 = 0,   // bit 0: dx is a T matrix (times 1 / add_scale); bit 1: dy_f32 is written as a T matrix times t_scale
                      int split_row = 0, const float* w2 = nullptr, float* dgamma2 = nullptr, float* dbeta2 = nullptr);   // blocks from split_row on: second parameter set
size_t tim_layernorm_bwd_ws(int rows, int cols);
int tim_layernorm_bwd_blocks(int rows);   // partial rows one backward launch over `rows` rows writes
// kbits (round 6): the layer's attention keep-bits as tim_attn_keep_bits wrote them (nullptr: the kernels draw their own)
int tim_attention_fwd(const TimDesc& d, const void* qkv, void* o, float* lse, hipStream_t s, const unsigned long long* kbits = nullptr);
int tim_attention_bwd(const TimDesc& d, const void* qkv, const void* o, const float* lse,
                      const void* d_o, void* dqkv, void* ws, size_t ws_bytes, hipStream_t s, const unsigned long long* kbits = nullptr);
// attention_mfma.hip: keep-bits of layers first .. first + n - 1 (n <= 8) into out[i]: [B * H * S][2] 64-bit words
int tim_attn_keep_bits(const TimDesc& d, int first_layer, int n, unsigned long long* const* out, hipStream_t s);
size_t tim_attention_bwd_ws(const TimDesc& d);

// operand storage: bf16 for TIMHIP_PREC_BF16, fp16 for TIMHIP_PREC_F16, fp32 for TIMHIP_PREC_FP32 and TIMHIP_PREC_BF16X3
static inline bool h16_storage(int precision) { return precision == TIMHIP_PREC_BF16 || precision == TIMHIP_PREC_F16; }
static inline bool f32_storage(int precision) { return !h16_storage(precision); }
static inline bool valid_precision(int precision) { return precision >= TIMHIP_PREC_BF16 && precision <= TIMHIP_PREC_F16; }
// run the statement with HT = the 16-bit operand type of `precision` (which must satisfy h16_storage)
// run the statement with T = the operand STORAGE type of `precision` (float / bf16_t / f16_t)
#define DISPATCH_T(prec, ...)                                           \
  do {                                                                  \
    if (f32_storage(prec)) { using T = float; __VA_ARGS__; }            \
    else if ((prec) == TIMHIP_PREC_F16) { using T = f16_t; __VA_ARGS__; } \
    else { using T = bf16_t; __VA_ARGS__; }                             \
  } while (0)
#define DISPATCH_H16(precision, ...)                                   \
  do {                                                                 \
    if ((precision) == TIMHIP_PREC_F16) { using HT = f16_t; __VA_ARGS__; } \
    else { using HT = bf16_t; __VA_ARGS__; }                            \
  } while (0)
static inline size_t opsize(int precision) { return f32_storage(precision) ? 4 : 2; }
