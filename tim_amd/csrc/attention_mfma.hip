// MFMA (bf16 / fp16 operands: template parameter HT) structured attention for the TIM encoder, forward and backward.
//
// Math: see attention.hip (token i attends to the F feature tokens + itself; reference
// tim.py:161-166 mask over nn.MultiheadAttention, transformers.py:102).
//
// One workgroup (4 waves) per (window, head).  The F feature keys/values of the head
// (F <= 192, padded to NJB*32) live in LDS for the whole kernel; query rows are processed
// in blocks of 32 with the MFMA issued "swapped" (D = K_frag x Q_frag = S^T) so that a
// LANE OWNS ONE QUERY ROW: the softmax reduction is over that lane's registers plus one
// cross-half shuffle, the probabilities feed the P.V MFMA straight from registers (the MFMA
// contraction order is a free permutation, so P's accumulator registers 8a..8a+7 ARE a valid
// B operand), and V^T fragments come from LDS through ds_read_b64_tr_b16.
//
// LDS image of a [rows][DH] bf16 tile: row stride DH*2 bytes, 16-byte chunk index XOR g(row)
// with g chosen so that BOTH access patterns are bank-conflict free:
//   ds_read_b128 fragment reads (16 different rows, same chunk)   -> rows map to 16 distinct slots
//   ds_read_b64_tr_b16 reads (4 rows x 64 B)                      -> 16 distinct slots
#include <stdlib.h>

#include "common.h"
#include "mfma_tiles.h"

namespace {

struct AttnArgsM {
  int S, F, E, H, LP;
  float scale;
  uint32_t thr; float dscale; TimSeed seed; uint32_t site;
  int rsplit, rper;   // the 32-row blocks of a (window, head) are spread over rsplit workgroups of rper row blocks each (1: one workgroup)
  const unsigned long long* kbits;   // keep-bits drawn ahead of the layer (tim_attn_keep_bits; round 6), or nullptr
};

__device__ __forceinline__ void keep4(const AttnArgsM& a, uint64_t rowbase, int key, float& k0, float& k1, float& k2,
                                      float& k3) {
  drop_mask4(a.seed, a.site, (rowbase + (uint64_t)key) >> 2, a.thr, a.dscale, k0, k1, k2, k3);
}
__device__ __forceinline__ float keep1(const AttnArgsM& a, uint64_t rowbase, int key) {
  float k[4];
  drop_mask4(a.seed, a.site, (rowbase + (uint64_t)key) >> 2, a.thr, a.dscale, k[0], k[1], k[2], k[3]);
  const int c = (int)((rowbase + (uint64_t)key) & 3);
  return c == 0 ? k[0] : (c == 1 ? k[1] : (c == 2 ? k[2] : k[3]));
}

// keep factors of one lane pair's 16 keys kb .. kb+15 (kb a multiple of 16, rowbase of 8): lane g owns keys kb + 8t + 4g .. +3
// for t = 0, 1.  Counter t covers keys kb + 8t .. +7: lane g draws counter t = g and passes its partner (lane ^ 32) the half
// that lane owns - one Philox call and two exchanges per lane instead of two calls (common.h: 16-bit draws)
__device__ __forceinline__ void keep_pair(const AttnArgsM& a, uint64_t rowbase, int kb, int g, float (&k0)[4], float (&k1)[4]) {
  const Philox4 r = philox4x32_7(a.seed, a.site, ((rowbase + (uint64_t)kb) >> 3) + (uint64_t)g);
  // v_permlane32_swap (x, z) and (y, w): lane g = 0 ends with (own x, partner's x), lane g = 1 with (partner's z, own z) - the
  // words of counter 0 first and of counter 1 second in both lanes, no select
  const auto xz = __builtin_amdgcn_permlane32_swap(r.x, r.z, false, false);
  const auto yw = __builtin_amdgcn_permlane32_swap(r.y, r.w, false, false);
  drop_mask4_words(xz[0], yw[0], a.thr, a.dscale, k0[0], k0[1], k0[2], k0[3]);
  drop_mask4_words(xz[1], yw[1], a.thr, a.dscale, k1[0], k1[1], k1[2], k1[3]);
}

template <int C>
__device__ __forceinline__ float quad_bcast(float v) {  // value of lane (lane & ~3) + C within each quad
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), C * 0x55, 0xF, 0xF, true));
}
template <int C>
__device__ __forceinline__ float quad_pick(float k0, float k1, float k2, float k3, int tl) {
  const float b0 = quad_bcast<C>(k0), b1 = quad_bcast<C>(k1), b2 = quad_bcast<C>(k2), b3 = quad_bcast<C>(k3);
  return tl == 0 ? b0 : (tl == 1 ? b1 : (tl == 2 ? b2 : b3));
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// KB (round 6): the dropout keep-bits of the layer were drawn ahead of it (a.kbits, two 64-bit words per row: tim_attn_keep_bits) -
// a lane reads the word of its (row, key half g) with its q row and turns bits into factors (v_bfe_i32 + v_and per element)
// instead of eight Philox calls per row block.  Same bits, same results.
template <typename HT, int DH, int NJB, bool KB = false>
__global__ __launch_bounds__(512) void attn_fwd_mfma(const HT* __restrict__ qkv, HT* __restrict__ o,
                                                     float* __restrict__ lse, AttnArgsM a) {
  constexpr int FP = NJB * 32, NKK = DH / 16, NDB = DH / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + FP * DH * 2;
  // (window, head) = blockIdx.x / rsplit; a long sequence (detection: S = 499, B * H = 128) spreads its row blocks over rsplit
  // workgroups that each stage the K / V tile themselves (51 KB from L2) - 128 workgroups would leave half of the 256 CUs idle
  const int bh = blockIdx.x / a.rsplit, part = blockIdx.x - bh * a.rsplit;
  const int b = bh / a.H, h = bh % a.H;
  const int S = a.S, F = a.F, E = a.E;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t ld = (size_t)3 * E;
  const HT* base = qkv + (size_t)b * S * ld + (size_t)h * DH;
  const int li = lane & 31, g = lane >> 5;
  const int nrb = min((S + 31) >> 5, (part + 1) * a.rper);
  const int nwaves = blockDim.x >> 6;
  // (five key blocks - F > 128 - hold 80 score registers per lane: no room for operands in flight, the requests stay where
  //  they are consumed)
  constexpr bool PRE = NJB <= 4;
  // Operand prefetch (round 4).  The block used to run a chain of dependent memory latencies - K tile, V tile, then per row
  // block its q / self-k rows, its self-v rows at the store - with two workgroups per CU to hide them: 31.5 us for 82 MB.  Now
  // every load is issued as early as its registers allow: the first row block's q / self-k rows BEFORE the K / V staging
  // loads (one latency for all three), K and V staged as a pair, the self-v rows and the NEXT row block's q / self-k rows
  // right after the S^T products (under the softmax and the P V products).
  auto load_rows = [&](int rbx, vec8<HT> (&qf)[NKK], vec8<HT> (&kself)[NKK]) {
    const int rowx = min(rbx * 32 + li, S - 1);
    const HT* qx = base + (size_t)rowx * ld;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = *reinterpret_cast<const vec8<HT>*>(qx + kk * 16 + g * 8);
    if (!PRE || rowx >= F) {   // (feature tokens have no self term; without operand prefetch the plain unconditional form)
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) kself[kk] = *reinterpret_cast<const vec8<HT>*>(qx + E + kk * 16 + g * 8);
    }
  };
  // one 32-row block of queries; qf / kself: its q rows and (query tokens) own-key rows, already requested
  auto row_block = [&](int rb, vec8<HT> (&qf)[NKK], vec8<HT> (&kself)[NKK]) {
    const int row = rb * 32 + li;
    const bool valid = row < S;
    const int rowc = valid ? row : S - 1;
    const bool isq = rowc >= F;
    const HT* qp = base + (size_t)rowc * ld;
    unsigned long long kw = 0ull;   // KB: this lane's keep-bits, bit 4 c + t = key 8 c + 4 g + t (requested ahead of the products)
    if constexpr (KB) kw = a.kbits[((((size_t)b * a.H + h) * S + rowc) << 1) + g];

    // S^T = K Q^T : lane owns query row `row`, registers hold keys 32jb + (r&3) + 8(r>>2) + 4g
    f32x16_t sc[NJB];
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[jb][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const vec8<HT> kf = *reinterpret_cast<const vec8<HT>*>(sK + tile_off<DH>(jb * 32 + li, kk * 2 + g));
        sc[jb] = mfma16<HT>(kf, qf[kk], sc[jb]);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the K-fragment reads of later key blocks from being hoisted
    }
    // self score of query tokens (raw, unscaled like sc)
    float sself = -INFINITY;
    {
      float t = 0.f;
      if (isq) {
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) t += dot8(qf[kk], kself[kk]);
      }
      const float other = __shfl_xor(t, 32, 64);
      if (isq) sself = t + other;
    }
    // only the last key block can hold padded keys
    {
      constexpr int jb = NJB - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (key >= F) sc[jb][r] = -INFINITY;
      }
    }
    float mx = sself;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[jb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // p = exp(scale*(s - mx)) = exp2(c*s - c*mx)
    const float c2 = a.scale * 1.4426950408889634f;
    const float mc = mx * c2;
    float sum = 0.f;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sc[jb][r], c2, -mc));
        sc[jb][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float pself_un = isq ? __builtin_amdgcn_exp2f(fmaf(sself, c2, -mc)) : 0.f;
    sum += pself_un;
    const float inv = 1.f / sum;
    if (valid && g == 0) lse[((size_t)b * a.H + h) * S + row] = mx * a.scale + __logf(sum);
    const uint64_t rowbase = (((uint64_t)b * a.H + h) * S + rowc) * (uint64_t)a.LP;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        float ka[4] = {1.f, 1.f, 1.f, 1.f}, kb[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (KB) {
          const int w32 = (int)(uint32_t)(kw >> (32 * (jb >> 1)));   // keys of key blocks 2 (jb >> 1), + 1: one dword
          const int ds = __float_as_int(a.dscale);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            ka[t] = __int_as_float(__builtin_amdgcn_sbfe(w32, 16 * (jb & 1) + 8 * qp + t, 1) & ds);
            kb[t] = __int_as_float(__builtin_amdgcn_sbfe(w32, 16 * (jb & 1) + 8 * qp + 4 + t, 1) & ds);
          }
        } else if (a.thr != 0u) keep_pair(a, rowbase, jb * 32 + 16 * qp, g, ka, kb);   // (both lanes of a pair take this branch)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          sc[jb][8 * qp + t] *= inv * ka[t];
          sc[jb][8 * qp + 4 + t] *= inv * kb[t];
        }
      }
    float pself = pself_un * inv;
    if (isq && a.thr != 0u) pself *= keep1(a, rowbase, F);

    // O^T = V^T P^T.  P is packed to bf16 first (frees the fp32 score registers); the head dim is
    // processed in halves so only NDB/2 accumulator tiles are live at a time.
    vec8<HT> pf[NJB][2];
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
      pf[jb][0] = pack8<HT>(sc[jb], 0);
      pf[jb][1] = pack8<HT>(sc[jb], 1);
    }
    // the self-v rows of this block (used at the store) go out now - the score registers are free - under the P V products
    vec8<HT> vself[NKK];
    if (PRE && isq) {
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) vself[kk] = *reinterpret_cast<const vec8<HT>*>(qp + 2 * E + kk * 16 + g * 8);
    }
    constexpr int NH = NDB >= 2 ? 2 : 1, DBH = NDB / NH;
    HT* op = o + ((size_t)b * S + rowc) * E + (size_t)h * DH;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
      f32x16_t oa[DBH];
#pragma unroll
      for (int d2 = 0; d2 < DBH; ++d2)
#pragma unroll
        for (int r = 0; r < 16; ++r) oa[d2][r] = 0.f;
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
          for (int d2 = 0; d2 < DBH; ++d2) {
            const vec8<HT> vf = tr_frag<DH, HT>(sV, jb * 32 + 16 * aa, hh * DBH + d2, lane);
            oa[d2] = mfma16<HT>(vf, pf[jb][aa], oa[d2]);
            if (d2 == DBH - 1 && aa == 1) __builtin_amdgcn_sched_barrier(0);
          }
      // lanes l and l ^ 32 trade quads so that each stores 8 contiguous head-dim columns (16 bytes) per pair of quads;
      // the exchange is executed by every lane (shuffles), the store only by valid rows
#pragma unroll
      for (int d2 = 0; d2 < DBH; ++d2)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
          float v[8];
          pair_exchange(v, oa[d2][8 * p2], oa[d2][8 * p2 + 1], oa[d2][8 * p2 + 2], oa[d2][8 * p2 + 3], oa[d2][8 * p2 + 4],
                        oa[d2][8 * p2 + 5], oa[d2][8 * p2 + 6], oa[d2][8 * p2 + 7], g);
          const int dh = 32 * (hh * DBH + d2) + 16 * p2 + 8 * g;
          if (valid) {
            if (isq) {
              // columns dh .. dh + 7 = 16 (dh / 16) + 8 g: the chunk requested above (PRE), or read here
              const vec8<HT> sv = PRE ? vself[2 * (hh * DBH + d2) + p2] : *reinterpret_cast<const vec8<HT>*>(qp + 2 * E + dh);
#pragma unroll
              for (int u = 0; u < 8; ++u) v[u] = fmaf(pself, (float)sv[u], v[u]);
            }
            store8_h<HT>(op + dh, v);
          }
        }
    }
  };

  // The wave's first row block is peeled out of the loop: its operands were requested before the K / V staging, and keeping
  // them in registers of their own (not loop-carried) is what lets the compiler fit the kernel without spills.
  int rb = part * a.rper + wave;
  if constexpr (!PRE) {   // the plain loop: operands requested where they are consumed
    stage_tile<DH>(sK, base + E, ld, FP, F, tid, blockDim.x);
    stage_tile<DH>(sV, base + 2 * E, ld, FP, F, tid, blockDim.x);
    __syncthreads();
    for (; rb < nrb; rb += nwaves) {
      vec8<HT> q1[NKK], k1[NKK];
      load_rows(rb, q1, k1);
      row_block(rb, q1, k1);
    }
    return;
  }
  {
    vec8<HT> q0[NKK], k0[NKK];
    if (PRE && rb < nrb) load_rows(rb, q0, k0);
    if constexpr (PRE) {
      stage_tile_pair<DH>(sK, base + E, sV, base + 2 * E, ld, FP, F, tid, blockDim.x);
    } else {
      stage_tile<DH>(sK, base + E, ld, FP, F, tid, blockDim.x);
      stage_tile<DH>(sV, base + 2 * E, ld, FP, F, tid, blockDim.x);
    }
    __syncthreads();
    if (!PRE && rb < nrb) load_rows(rb, q0, k0);
    if (rb < nrb) row_block(rb, q0, k0);
  }
  for (rb += nwaves; rb < nrb; rb += nwaves) {
    vec8<HT> q1[NKK], k1[NKK];
    load_rows(rb, q1, k1);
    row_block(rb, q1, k1);
  }
}

// ---------------------------------------------------------------------------
// backward
//   phase 1 (lane = query row): S^T, dP^T = V dO^T, dS -> dQ^T = K^T dS^T ; self terms of query tokens
//   phase 2 (lane = key):       S = Q K^T, dP = dO V^T recomputed in the transposed orientation so
//                               that dS / P~ land in registers as B operands of
//                               dK^T = Q^T dS and dV^T = dO^T P~ (contraction over the 32 rows of a block).
// ---------------------------------------------------------------------------
template <typename HT, int DH, int NJB>
__global__ __launch_bounds__(256) void attn_bwd_mfma(const HT* __restrict__ qkv, const HT* __restrict__ o,
                                                     const float* __restrict__ lse, const HT* __restrict__ d_o,
                                                     HT* __restrict__ dqkv, AttnArgsM a) {
  constexpr int FP = NJB * 32, NKK = DH / 16, NDB = DH / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = sK + FP * DH * 2;
  char* sQ = sV + FP * DH * 2;        // 32-row block of Q
  char* sD = sQ + 32 * DH * 2;        // 32-row block of dO
  float* sLse = reinterpret_cast<float*>(sD + 32 * DH * 2);  // [S] lse
  float* sDel = sLse + a.S;                                  // [S] delta = dO . O
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int S = a.S, F = a.F, E = a.E;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t ld = (size_t)3 * E;
  const HT* base = qkv + (size_t)b * S * ld + (size_t)h * DH;
  HT* dbase = dqkv + (size_t)b * S * ld + (size_t)h * DH;
  const HT* dobase = d_o + (size_t)b * S * E + (size_t)h * DH;
  const HT* obase = o + (size_t)b * S * E + (size_t)h * DH;
  const float* lsebase = lse + ((size_t)b * a.H + h) * S;
  stage_tile<DH>(sK, base + E, ld, FP, F, tid, blockDim.x);
  stage_tile<DH>(sV, base + 2 * E, ld, FP, F, tid, blockDim.x);
  __syncthreads();

  const int li = lane & 31, g = lane >> 5;
  const int nrb = (S + 31) >> 5;

  // ---------------- phase 1 ----------------
  const int nwaves = blockDim.x >> 6;
  for (int rb = wave; rb < nrb; rb += nwaves) {
    const int row = rb * 32 + li;
    const bool valid = row < S;
    const int rowc = valid ? row : S - 1;
    const bool isq = rowc >= F;
    const HT* qp = base + (size_t)rowc * ld;
    const HT* dop = dobase + (size_t)rowc * E;
    const HT* op = obase + (size_t)rowc * E;
    vec8<HT> qf[NKK], df[NKK];
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      qf[kk] = *reinterpret_cast<const vec8<HT>*>(qp + kk * 16 + g * 8);
      df[kk] = *reinterpret_cast<const vec8<HT>*>(dop + kk * 16 + g * 8);
      delta += dot8(df[kk], *reinterpret_cast<const vec8<HT>*>(op + kk * 16 + g * 8));
    }
    delta += __shfl_xor(delta, 32, 64);
    const float l = lsebase[rowc];
    if (valid && g == 0) { sLse[row] = l; sDel[row] = delta; }
    const uint64_t rowbase = (((uint64_t)b * a.H + h) * S + rowc) * (uint64_t)a.LP;

    // self terms (scalar per row)
    float ds_self = 0.f, pt_self = 0.f;
    if (isq) {
      float t = 0.f, u = 0.f;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        t += dot8(qf[kk], *reinterpret_cast<const vec8<HT>*>(qp + E + kk * 16 + g * 8));
        u += dot8(df[kk], *reinterpret_cast<const vec8<HT>*>(qp + 2 * E + kk * 16 + g * 8));
      }
      ds_self = t; pt_self = u;
    }
    {
      const float t2 = __shfl_xor(ds_self, 32, 64), u2 = __shfl_xor(pt_self, 32, 64);
      if (isq) {
        const float t = (ds_self + t2) * a.scale, u = pt_self + u2;
        const float p = __expf(t - l);
        const float keep = a.thr != 0u ? keep1(a, rowbase, F) : 1.f;
        ds_self = p * (u * keep - delta) * a.scale;
        pt_self = p * keep;
      }
    }
    // per key block: S^T, dP^T -> dS^T (registers) -> dQ^T += K^T dS^T
    f32x16_t qa[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) qa[db][r] = 0.f;
#pragma unroll 1
    for (int jb = 0; jb < NJB; ++jb) {
      f32x16_t sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const vec8<HT> kf = *reinterpret_cast<const vec8<HT>*>(sK + tile_off<DH>(jb * 32 + li, kk * 2 + g));
        const vec8<HT> vf = *reinterpret_cast<const vec8<HT>*>(sV + tile_off<DH>(jb * 32 + li, kk * 2 + g));
        sc = mfma16<HT>(kf, qf[kk], sc);
        dp = mfma16<HT>(vf, df[kk], dp);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float k[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.thr != 0u) keep4(a, rowbase, jb * 32 + 8 * q + 4 * g, k[0], k[1], k[2], k[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = 4 * q + t;
          const int key = jb * 32 + 8 * q + 4 * g + t;
          const float p = key < F ? __expf(sc[r] * a.scale - l) : 0.f;
          sc[r] = p * (dp[r] * k[t] - delta) * a.scale;
        }
      }
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
        const vec8<HT> sf = pack8<HT>(sc, aa);
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          const vec8<HT> kf = tr_frag<DH, HT>(sK, jb * 32 + 16 * aa, db, lane);
          qa[db] = mfma16<HT>(kf, sf, qa[db]);
        }
      }
    }
    if (valid) {
      HT* dq = dbase + (size_t)row * ld;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dh = 32 * db + 8 * q + 4 * g;
          float v0 = qa[db][4 * q], v1 = qa[db][4 * q + 1], v2 = qa[db][4 * q + 2], v3 = qa[db][4 * q + 3];
          if (isq) {
            float k0, k1, k2, k3, q0, q1, q2, q3, d0, d1, d2, d3;
            load4<HT>(qp + E + dh, k0, k1, k2, k3);
            load4<HT>(qp + dh, q0, q1, q2, q3);
            load4<HT>(dop + dh, d0, d1, d2, d3);
            v0 = fmaf(ds_self, k0, v0); v1 = fmaf(ds_self, k1, v1); v2 = fmaf(ds_self, k2, v2); v3 = fmaf(ds_self, k3, v3);
            // a query token's own key / value receive the self term only
            store4<HT>(dq + E + dh, ds_self * q0, ds_self * q1, ds_self * q2, ds_self * q3);
            store4<HT>(dq + 2 * E + dh, pt_self * d0, pt_self * d1, pt_self * d2, pt_self * d3);
          }
          store4<HT>(dq + dh, v0, v1, v2, v3);
        }
    }
  }

  // ---------------- phase 2 ----------------
  // wave w owns key blocks w, w+4; accumulators dK^T, dV^T [dh][key] over all row blocks
  for (int jb0 = 0; jb0 < NJB; jb0 += nwaves) {
    const int jb = jb0 + wave;
    const bool active = jb < NJB;
    f32x16_t ka[NDB], va[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) { ka[db][r] = 0.f; va[db][r] = 0.f; }
    const int key = jb * 32 + li;
    const int jbc = active ? jb : 0;
    for (int rb = 0; rb < nrb; ++rb) {
      __syncthreads();  // previous block's sQ/sD fully consumed
      const int r0 = rb * 32;
      const int nvalid = min(32, S - r0);
      stage_tile<DH>(sQ, base + (size_t)r0 * ld, ld, 32, nvalid, tid, blockDim.x);
      stage_tile<DH>(sD, dobase + (size_t)r0 * E, E, 32, nvalid, tid, blockDim.x);
      __syncthreads();
      if (!active) continue;
      // S = Q K^T, dP = dO V^T : lane owns key `key`, registers hold rows (r&3) + 8(r>>2) + 4g
      f32x16_t sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const vec8<HT> qf = *reinterpret_cast<const vec8<HT>*>(sQ + tile_off<DH>(li, kk * 2 + g));
        const vec8<HT> df = *reinterpret_cast<const vec8<HT>*>(sD + tile_off<DH>(li, kk * 2 + g));
        const vec8<HT> kf = *reinterpret_cast<const vec8<HT>*>(sK + tile_off<DH>(jbc * 32 + li, kk * 2 + g));
        const vec8<HT> vf = *reinterpret_cast<const vec8<HT>*>(sV + tile_off<DH>(jbc * 32 + li, kk * 2 + g));
        sc = mfma16<HT>(qf, kf, sc);
        dp = mfma16<HT>(df, vf, dp);
      }
      // dropout keep factors: the 4 lanes of a quad hold 4 consecutive keys, so one Philox call
      // (4 outputs) serves a whole quad; lane t of the quad draws for register-row t and the
      // results are exchanged with DPP quad broadcasts (4 calls per lane instead of 16).
      float keepv[16];
      if (a.thr != 0u) {
        const int tl = lane & 3;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int rr = min(r0 + 8 * rq + 4 * g + tl, S - 1);
          const uint64_t rowbase = (((uint64_t)b * a.H + h) * S + rr) * (uint64_t)a.LP;
          float k0, k1, k2, k3;
          keep4(a, rowbase, min(key & ~3, a.LP - 4), k0, k1, k2, k3);
          keepv[4 * rq + 0] = quad_pick<0>(k0, k1, k2, k3, tl);
          keepv[4 * rq + 1] = quad_pick<1>(k0, k1, k2, k3, tl);
          keepv[4 * rq + 2] = quad_pick<2>(k0, k1, k2, k3, tl);
          keepv[4 * rq + 3] = quad_pick<3>(k0, k1, k2, k3, tl);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) keepv[r] = 1.f;
      }
      f32x16_t dsr, ptr_;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * g;
        const int rr = r0 + rl;
        const bool ok = rr < S && key < F;
        const int rc = min(rr, S - 1);
        const float p = ok ? __expf(sc[r] * a.scale - sLse[rc]) : 0.f;
        dsr[r] = p * (dp[r] * keepv[r] - sDel[rc]) * a.scale;
        ptr_[r] = p * keepv[r];
      }
      // dK^T += Q^T dS ; dV^T += dO^T P~   (contraction over the 32 rows: two K=16 MFMAs)
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
        const vec8<HT> sf = pack8<HT>(dsr, aa), pf = pack8<HT>(ptr_, aa);
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          const vec8<HT> qt = tr_frag<DH, HT>(sQ, 16 * aa, db, lane);
          const vec8<HT> dt = tr_frag<DH, HT>(sD, 16 * aa, db, lane);
          ka[db] = mfma16<HT>(qt, sf, ka[db]);
          va[db] = mfma16<HT>(dt, pf, va[db]);
        }
      }
    }
    if (active && key < F) {
      HT* dk = dbase + (size_t)key * ld + E;
      HT* dv = dbase + (size_t)key * ld + 2 * E;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dh = 32 * db + 8 * q + 4 * g;
          store4<HT>(dk + dh, ka[db][4 * q], ka[db][4 * q + 1], ka[db][4 * q + 2], ka[db][4 * q + 3]);
          store4<HT>(dv + dh, va[db][4 * q], va[db][4 * q + 1], va[db][4 * q + 2], va[db][4 * q + 3]);
        }
    }
  }
}

AttnArgsM make_args(const TimDesc& d) {
  AttnArgsM a;
  a.S = d.S; a.F = d.F; a.E = d.E; a.H = d.H; a.LP = round_up(d.F + 1, 8);
  a.scale = 1.f / sqrtf((float)(d.E / d.H));
  a.thr = d.p_drop > 0.f ? drop_threshold(d.p_drop) : 0u;
  a.dscale = d.p_drop > 0.f ? 1.f / (1.f - d.p_drop) : 1.f;
  a.seed = d.seed; a.site = layer_site(d.layer, SITE_L_ATTN);
  a.rsplit = 1; a.rper = (d.S + 31) / 32;
  a.kbits = nullptr;
  return a;
}

// Row split of a (window, head): enough workgroups for two per CU (512) when B * H alone does not provide them, each with at
// least `waves_min` row blocks (one per wave).  C4 training (B = 16, H = 8, S = 499: 16 row blocks): 4 parts of 4 row blocks.
static inline void attn_row_split(const TimDesc& d, int waves_min, int& rsplit, int& rper) {
  const int nrb = (d.S + 31) / 32, bh = d.B * d.H;
  int want = bh >= 512 ? 1 : (512 + bh - 1) / bh;
  const int most = nrb / (waves_min < 1 ? 1 : waves_min);
  if (want > most) want = most;
  if (want < 1) want = 1;
  rper = (nrb + want - 1) / want;
  rsplit = (nrb + rper - 1) / rper;
}

// one wave per 32-row block of queries, at most 8 waves (2 per SIMD keeps the 256-VGPR budget)
// One wave per 32-row block of queries, but at most FOUR per block: the kernel needs 212 VGPRs (two waves per SIMD = eight wave
// slots per CU) and 64-80 KiB of LDS, so two 4-wave blocks are co-resident on a CU where a 5-wave block (S = 155) runs alone -
// one block's K / V staging and operand loads then overlap the other's arithmetic (C2a forward 33.3 -> 26.9 us; 3 waves 28.8,
// 8 waves 32.8); a wave walks rows rb, rb + 4, ...
static inline int attn_waves(int S) {
  int n = (S + 31) / 32;
  n = n < 1 ? 1 : (n > 4 ? 4 : n);
  const int w = tim_knobs().attn_waves;   // (A/B knob: fewer waves than row blocks - a wave then walks several)
  if (w >= 1 && w <= 8) n = w;
  return n;
}

template <typename HT, int DH, int NJB>
int launch_fwd(const TimDesc& d, const void* qkv, void* o, float* lse, hipStream_t s, const unsigned long long* kbits) {
  const size_t lds = (size_t)2 * NJB * 32 * DH * 2;
  AttnArgsM a = make_args(d);
  // (few windows - B * H < 128, e.g. C2a at 8 windows per GPU - leave most CUs without a block: split down to TIMHIP_ATTN_SPLIT_MIN
  //  row blocks per workgroup there; default 4 = one per wave of a full block)
  attn_row_split(d, d.B * d.H < 128 ? tim_knobs().attn_split_min : 4, a.rsplit, a.rper);
  if constexpr (DH == 128 && NJB == 4) {   // (the keep-bit form exists for the geometry tim_attn_keep_bits serves: C2a / C3 / C4)
    if (kbits && a.thr != 0u) {
      a.kbits = kbits;
      (void)hipFuncSetAttribute((const void*)attn_fwd_mfma<HT, DH, NJB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((attn_fwd_mfma<HT, DH, NJB, true>), dim3(d.B * d.H * a.rsplit), dim3(64 * attn_waves(32 * a.rper)), lds, s,
                         (const HT*)qkv, (HT*)o, lse, a);
      return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
    }
  }
  (void)hipFuncSetAttribute((const void*)attn_fwd_mfma<HT, DH, NJB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_fwd_mfma<HT, DH, NJB>), dim3(d.B * d.H * a.rsplit), dim3(64 * attn_waves(32 * a.rper)), lds, s,
                     (const HT*)qkv, (HT*)o, lse, a);
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}

// keep-bits of the attention dropout, one thread per (window, head, token row): the 8 elements of Philox counter c are the keys
// 8 c .. 8 c + 7; the MFMA kernels' lane (row, g) owns keys 8 c + 4 g + t - so the low nibble of a counter's keep-bits goes
// to word g = 0 and the high nibble to word g = 1, both at bit 4 c
struct KeepBitsArgs { unsigned long long* out[8]; TimSeed seed; uint32_t site[8]; uint32_t thr; int rows, nc, lp8; };
__global__ __launch_bounds__(256) void attn_keep_bits_kernel(KeepBitsArgs k) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= k.rows) return;
  const uint64_t c0 = (uint64_t)row * (uint64_t)k.lp8;
  const uint32_t site = k.site[blockIdx.y];
  // (unrolled over the 16 possible counters: a thread's draws are independent, and a runtime-bounded loop ran their 7-round
  //  dependency chains one after the other - 16 us for 6.2 M calls where the vector pipes need 6; the two 32-bit halves of a word
  //  are built separately: no 64-bit shifts)
  const uint64_t seed = k.seed;
  uint32_t lo0 = 0u, hi0 = 0u, lo1 = 0u, hi1 = 0u;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < k.nc) {
      const uint32_t bits = drop_bits8(seed, site, c0 + (uint64_t)c, k.thr);
      if (c < 8) { lo0 |= (bits & 15u) << (4 * c); lo1 |= (bits >> 4) << (4 * c); }
      else { hi0 |= (bits & 15u) << (4 * (c - 8)); hi1 |= (bits >> 4) << (4 * (c - 8)); }
    }
  }
  const unsigned long long w0 = ((unsigned long long)hi0 << 32) | lo0, w1 = ((unsigned long long)hi1 << 32) | lo1;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  u64x2 v; v[0] = w0; v[1] = w1;
  *reinterpret_cast<u64x2*>(k.out[blockIdx.y] + 2 * (size_t)row) = v;
}
template <typename HT, int DH, int NJB>
int launch_bwd(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv,
               hipStream_t s) {
  const size_t lds = (size_t)2 * NJB * 32 * DH * 2 + (size_t)2 * 32 * DH * 2 + (size_t)2 * d.S * sizeof(float);
  (void)hipFuncSetAttribute((const void*)attn_bwd_mfma<HT, DH, NJB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_bwd_mfma<HT, DH, NJB>), dim3(d.B * d.H), dim3(256), lds, s, (const HT*)qkv,
                     (const HT*)o, lse, (const HT*)d_o, (HT*)dqkv, make_args(d));
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}

}  // namespace

// returns TIMHIP_EUNSUPPORTED when the (head_dim, F) combination has no MFMA instantiation;
// the caller then uses the fp32-arithmetic kernels of attention.hip
#define ATTN_DISPATCH(FN, ...)                                                   \
  do {                                                                           \
    const int DHv = d.E / d.H, NJBv = (d.F + 31) / 32;                           \
    if (DHv == 128) {                                                            \
      switch (NJBv) {                                                            \
        case 1: return FN<HT, 128, 1>(__VA_ARGS__);                                  \
        case 2: return FN<HT, 128, 2>(__VA_ARGS__);                                  \
        case 3: return FN<HT, 128, 3>(__VA_ARGS__);                                  \
        case 4: return FN<HT, 128, 4>(__VA_ARGS__);                                  \
        case 5: return FN<HT, 128, 5>(__VA_ARGS__);                                  \
        default: return TIMHIP_EUNSUPPORTED;                                     \
      }                                                                          \
    } else if (DHv == 64) {                                                      \
      switch (NJBv) {                                                            \
        case 1: return FN<HT, 64, 1>(__VA_ARGS__);                                   \
        case 2: return FN<HT, 64, 2>(__VA_ARGS__);                                   \
        case 4: return FN<HT, 64, 4>(__VA_ARGS__);                                   \
        default: return TIMHIP_EUNSUPPORTED;                                     \
      }                                                                          \
    } else if (DHv == 32) {                                                      \
      switch (NJBv) {                                                            \
        case 1: return FN<HT, 32, 1>(__VA_ARGS__);                                   \
        case 2: return FN<HT, 32, 2>(__VA_ARGS__);                                   \
        default: return TIMHIP_EUNSUPPORTED;                                     \
      }                                                                          \
    }                                                                            \
    return TIMHIP_EUNSUPPORTED;                                                  \
  } while (0)

int tim_attention_fwd_mfma(const TimDesc& d, const void* qkv, void* o, float* lse, hipStream_t s, const unsigned long long* kbits) {
  if (!h16_storage(d.precision) || (d.E % 8) != 0) return TIMHIP_EUNSUPPORTED;
  DISPATCH_H16(d.precision, ATTN_DISPATCH(launch_fwd, d, qkv, o, lse, s, kbits));
  return TIMHIP_EUNSUPPORTED;
}

int tim_attn_keep_bits(const TimDesc& d, int first_layer, int n, unsigned long long* const* out, hipStream_t s) {
  if (n < 1 || n > 8 || !out) return TIMHIP_EINVAL;
  KeepBitsArgs k;
  for (int i = 0; i < 8; ++i) { k.out[i] = i < n ? out[i] : nullptr; k.site[i] = layer_site(first_layer + (i < n ? i : 0), SITE_L_ATTN); }
  k.seed = d.seed; k.thr = drop_threshold(d.p_drop);
  k.rows = d.B * d.H * d.S;
  k.lp8 = round_up(d.F + 1, 8) / 8;
  k.nc = k.lp8 < 16 ? k.lp8 : 16;   // keys 0 .. 127 (the self key of a 128-key window is drawn by the kernels themselves)
  hipLaunchKernelGGL(attn_keep_bits_kernel, dim3((k.rows + 255) / 256, n), dim3(256), 0, s, k);
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}

int tim_attention_bwd_mfma(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o,
                           void* dqkv, hipStream_t s) {
  if (!h16_storage(d.precision) || (d.E % 8) != 0) return TIMHIP_EUNSUPPORTED;
  DISPATCH_H16(d.precision, ATTN_DISPATCH(launch_bwd, d, qkv, o, lse, d_o, dqkv, s));
  return TIMHIP_EUNSUPPORTED;
}
