// MFMA (bf16) structured-attention backward, two-kernel form (see attention_mfma.hip for the math and the
// LDS tile layout):
//   attn_bwd_rows  : lane = query row.  Recomputes S^T and dP^T = V dO^T per 32-row block, forms dS, writes
//                    dQ (+ the self-term dK/dV of query tokens) and hands dS and the dropped probabilities P~
//                    to the second kernel through a bf16 scratch [B*H][S][FP].
//   attn_bwd_keys  : dK = dS^T Q and dV = P~^T dO for the F feature keys: a batched "TN" product whose
//                    contraction runs over the token rows, with both operands read in their natural layout and
//                    transposed by ds_read_b64_tr_b16 (same scheme as wgrad.hip).
// Compared with the single-kernel version (attn_bwd_mfma) nothing is recomputed in the transposed orientation:
// no second exp / Philox pass, and the row kernel runs one wave per row block.
#include <stdlib.h>

#include "common.h"
#include "mfma_tiles.h"

namespace {

struct AttnArgsM {
  int S, F, E, H, LP;
  float scale;
  uint32_t thr; float dscale; TimSeed seed; uint32_t site;
  int abl;  // tuning builds only (TimDesc.reserved >> 8): 1 no scratch stores, 2 no dqkv stores, 4 operand rows alias row 0
  int rsplit, rper;   // two-kernel form: the row blocks of a (window, head) over rsplit workgroups of rper row blocks (attention_mfma.hip)
  const unsigned long long* kbits;   // keep-bits drawn ahead of the layer (attention_mfma.hip: tim_attn_keep_bits), or nullptr
};
#ifdef TIMHIP_TUNING
#define ATT_ABL(a, bit) (((a).abl & (bit)) != 0)
// (tuning, abl bit 16, fused form: wave 0's shader-clock stamps per block into the workspace - tools/attn_one.py prints the phases)
#define ATT_STAMP(i)                                                                                              \
  do {                                                                                                            \
    if (FUSED && ATT_ABL(a, 16) && tid == 0)                                                                      \
      reinterpret_cast<unsigned long long*>(dS_scr)[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define ATT_ABL(a, bit) false
#define ATT_STAMP(i) do { } while (0)
#endif

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

// FUSED (DH = 128, FP = 128, S <= 192): dS and P~ of the (window, head) stay in LDS ([S rounded up to 32][128 keys] each,
// 80 KB at S = 155) and the block computes dK / dV of its feature keys itself after its row blocks are done (phase 2: the
// key-side TN product of attn_bwd_keys on the block's own tiles, Q and dO staged over the K / V space) - no scratch round
// trip through HBM (41 MB written and read back per layer at C2a), no second launch.  (Measured alternative: the three waves
// without a row block prefetching Q / dO into registers during phase 1 - slower, 81 vs 74 us: their loads compete with the
// K / V staging and the row blocks' own operand loads at the start of the block.)
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

// KS (round 5, fused form only): phase 1 as a pipeline over the 32-row blocks on all EIGHT waves (the one-wave-per-row-block
// form left three waves idle at S = 155 and read every operand row with one lane per row: 32 cache lines per load instruction).
//   top      every request in arrival order: K / V tiles, the Q / dO rows of all sweeps by coalesced loads (16 lanes per row; kept
//            in registers for phase 2), sweep 0's O rows; O / own-key / own-value rows of sweep s + 1 one step ahead
//   sweep s  Q / dO rows of row block s into the LDS space that will hold its dS / P~; delta, lse, self terms per row
//   step s   waves 0 - 3: one (row block s, key block) unit each -> dS / P~ (fragments read, LDS-only barrier, same space
//            written); waves 4 - 7: one (row block s - 1, head-dim block) dQ unit each, dS read back from LDS in tr_frag's k-order
//            (same values, same accumulation order as the one-wave form); every thread: sweep s + 1
// DESIGN.md section 5f has the per-phase clocks and what was tried on top.
// KB (round 6, key-split form): a unit's keep factors from the layer's keep-bits - the 16 bits of (row, g, key block), requested
// one pipeline step ahead - instead of two Philox calls per unit.
template <typename HT, int DH, int NJB, bool FUSED = false, bool KS = false, bool KB = false>
__global__ __launch_bounds__(512) void attn_bwd_rows(const HT* __restrict__ qkv, const HT* __restrict__ o,
                                                     const float* __restrict__ lse, const HT* __restrict__ d_o,
                                                     HT* __restrict__ dqkv, HT* __restrict__ dS_scr,
                                                     HT* __restrict__ Pt_scr, AttnArgsM a) {
  constexpr int FP = NJB * 32, NKK = DH / 16, NDB = DH / 32;
  static_assert(!FUSED || (DH == 128 && FP == 128), "the fused form is written for 128 x 128 tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = sK + FP * DH * 2;
  char* sS = sV + FP * DH * 2;                          // FUSED: dS  [SP][FP]
  char* sP = sS + ((a.S + 31) & ~31) * FP * 2;          // FUSED: P~  [SP][FP]
  const int rsplit = FUSED ? 1 : a.rsplit;
  const int bh = blockIdx.x / rsplit, part = blockIdx.x - bh * rsplit;
  const int b = bh / a.H, h = bh % a.H;
  const int S = a.S, F = a.F, E = a.E;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t ld = (size_t)3 * E;
  const HT* base = qkv + (size_t)b * S * ld + (size_t)h * DH;
  HT* dbase = dqkv + (size_t)b * S * ld + (size_t)h * DH;
  const HT* dobase = d_o + (size_t)b * S * E + (size_t)h * DH;
  const HT* obase = o + (size_t)b * S * E + (size_t)h * DH;
  const float* lsebase = lse + ((size_t)b * a.H + h) * S;
  HT* dSs = dS_scr + (size_t)bh * S * FP;
  HT* Pts = Pt_scr + (size_t)bh * S * FP;
  // (K and V staged as a pair: every load of both tiles ahead of the first LDS write - one memory latency, not two)
  ATT_STAMP(0);
  // key-split form: this thread's 16-byte chunks of the Q / dO rows (row (tid >> 4) + 32 sw, chunk tid & 15) - requested before
  // the K / V staging, written to LDS by phase 0, and kept in registers for phase 2 (which wants exactly these chunks again)
  constexpr int NSW = (FUSED && KS) ? 6 : 1;   // row sweeps of 32 (the key-split form fits S <= 160 = 5 sweeps: fused_fits; 6 keeps XU below in step)
  constexpr int NKV = (FUSED && KS) ? 4 : 1;   // 128 key rows / 32
  vec8<HT> qv[NSW], dv[NSW], kst[NKV], vst[NKV];
  vec8<HT> ov, kv, vv;   // one sweep's O / own-key / own-value chunks: requested one pipeline step ahead of their use
  auto load_okv = [&](int sw) {
    const int c = tid & 15, row = (tid >> 4) + 32 * sw;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ov[e] = (HT)0.f; kv[e] = (HT)0.f; vv[e] = (HT)0.f; }
    if (row < S) {
      const size_t ro = ATT_ABL(a, 4) ? (size_t)(row & 1) : (size_t)row;   // (tuning: operand rows from two cached rows)
      ov = *reinterpret_cast<const vec8<HT>*>(obase + ro * E + c * 8);
      if (row >= F) {
        kv = *reinterpret_cast<const vec8<HT>*>(base + ro * ld + E + c * 8);
        vv = *reinterpret_cast<const vec8<HT>*>(base + ro * ld + 2 * E + c * 8);
      }
    }
  };
  if constexpr (FUSED && KS) {
    // request order = arrival order: the K / V tiles, then the Q / dO row sweeps 0, 1, ... - the products on row block 0 start
    // while the later sweeps are still in flight (phase 1 below)
    const int c = tid & 15, r0 = tid >> 4;
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int row = r0 + 32 * u;
#pragma unroll
      for (int e = 0; e < 8; ++e) { kst[u][e] = (HT)0.f; vst[u][e] = (HT)0.f; }
      if (row < F) {
        const size_t ro = ATT_ABL(a, 4) ? (size_t)(row & 1) : (size_t)row;
        kst[u] = *reinterpret_cast<const vec8<HT>*>(base + ro * ld + E + c * 8);
        vst[u] = *reinterpret_cast<const vec8<HT>*>(base + ro * ld + 2 * E + c * 8);
      }
    }
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw) {
      const int row = r0 + 32 * sw;
      const bool valid = row < S;
      const size_t ro = (size_t)(valid ? (ATT_ABL(a, 4) ? (row & 1) : row) : 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) { qv[sw][e] = (HT)0.f; dv[sw][e] = (HT)0.f; }
      if (valid) {
        qv[sw] = *reinterpret_cast<const vec8<HT>*>(base + ro * ld + c * 8);
        dv[sw] = *reinterpret_cast<const vec8<HT>*>(dobase + ro * E + c * 8);
      }
      if (sw == 0) load_okv(0);
    }
    {   // the rows' log-sum-exp, one thread per row
      float* sL0 = reinterpret_cast<float*>(sP + ((S + 31) & ~31) * FP * 2) + ((S + 31) & ~31);
      if (tid < ((S + 31) & ~31)) sL0[tid] = tid < S ? lsebase[tid] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int to = tile_off<128>(r0 + 32 * u, c);
      *reinterpret_cast<vec8<HT>*>(sK + to) = kst[u];
      *reinterpret_cast<vec8<HT>*>(sV + to) = vst[u];
    }
  } else {
    stage_tile_pair<DH>(sK, base + E, sV, base + 2 * E, ld, FP, F, tid, blockDim.x);
    __syncthreads();
  }
  ATT_STAMP(1);

  const int li = lane & 31, g = lane >> 5;
  const int nrb = FUSED ? ((S + 31) >> 5) : min((S + 31) >> 5, (part + 1) * a.rper);

  // ---------------- phase 1 ----------------
  const int nwaves = blockDim.x >> 6;
  if constexpr (FUSED && KS) {
    static_assert(!KS || (FUSED && NJB == 4 && DH == 128), "key-split phase 1: the fused 128 x 128 form");
    const int SPr = (S + 31) & ~31;
    float* sDelta = reinterpret_cast<float*>(sP + SPr * FP * 2);   // [SP] rowsum(dO * O)
    float* sL = sDelta + SPr;                                      // [SP] the row's log-sum-exp
    float* sDs = sL + SPr;                                         // [SP] a query row's self term of dS (0 for feature rows)
    // ---- phase 0 (per sweep): the Q / dO rows into LDS by coalesced loads (16 lanes per 256-byte row), into the space that
    //      will hold the row block's dS / P~; per-row scalars (delta, lse, self terms) and a query token's own key / value
    //      gradient rows on the way.  One lane per row of a 32 x 32 MFMA operand read global memory 32 lines per instruction
    //      before.
    auto sweep_store = [&](const vec8<HT>& q8, const vec8<HT>& d8, const vec8<HT>& o8, const vec8<HT>& k8, const vec8<HT>& v8,
                           int sw) {
      const int c = tid & 15, row = (tid >> 4) + 32 * sw;
      const bool valid = row < S, isq = valid && row >= F;
      const int to = tile_off<128>(row, c);
      *reinterpret_cast<vec8<HT>*>(sS + to) = q8;
      *reinterpret_cast<vec8<HT>*>(sP + to) = d8;
      const float dl = row16_sum(dot8(d8, o8));
      float t = 0.f, u = 0.f;
      if (32 * sw + 31 >= F) {   // (wave-uniform: the sweep holds query rows)
        t = row16_sum(dot8(q8, k8));
        u = row16_sum(dot8(d8, v8));
      }
      float ds_self = 0.f, pt_self = 0.f;
      if (isq) {   // (rows >= F >= 97: never in sweeps 0 .. 2, i.e. always behind the first barrier - sL is complete)
        const uint64_t rowbase = (((uint64_t)b * a.H + h) * S + row) * (uint64_t)a.LP;
        const float p = __expf(t * a.scale - sL[row]);
        const float keep = a.thr != 0u ? keep1(a, rowbase, F) : 1.f;
        ds_self = p * (u * keep - dl) * a.scale;
        pt_self = p * keep;
      }
      if (c == 0) { sDelta[row] = dl; sDs[row] = ds_self; }
      if (isq && !ATT_ABL(a, 2)) {
        HT* dq = dbase + (size_t)row * ld + c * 8;
        float kself[8], vself[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { kself[e] = ds_self * (float)q8[e]; vself[e] = pt_self * (float)d8[e]; }
        store8_h<HT>(dq + E, kself);
        store8_h<HT>(dq + 2 * E, vself);
      }
    };
    // ---- 1a unit: (row block, key block) -> dS / P~ in LDS over the row block's Q / dO (read into registers first, one block
    //      barrier between the reads and the writes)
    vec8<HT> qf[NKK], df[NKK];
    auto read_frags = [&](int rb) {
      const int row = rb * 32 + li;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        qf[kk] = *reinterpret_cast<const vec8<HT>*>(sS + tile_off<128>(row, kk * 2 + g));
        df[kk] = *reinterpret_cast<const vec8<HT>*>(sP + tile_off<128>(row, kk * 2 + g));
      }
    };
    // KB: the 16 keep-bits of (row block rb's row li, key half g, key block jb): bit 8 qp + 4 t' + t = key 32 jb + 16 qp + 8 t' + 4 g + t
    auto load_kw = [&](int rb, int jb) -> uint32_t {
      const int rowc = min(rb * 32 + li, S - 1);
      return (uint32_t)*reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(a.kbits) +
                                                                ((((size_t)bh * S + rowc) << 1) + g) * 8 + 2 * jb);
    };
    auto item_a = [&](int rb, int jb, uint32_t kw) {
      const int row = rb * 32 + li;
      const bool valid = row < S;
      const int rowc = valid ? row : S - 1;
      const float delta = sDelta[row], l2 = sL[row] * 1.44269504088896341f, c2 = a.scale * 1.44269504088896341f;
      const uint64_t rowbase = (((uint64_t)b * a.H + h) * S + rowc) * (uint64_t)a.LP;
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
      for (int qp2 = 0; qp2 < 2; ++qp2) {
        float kk2[2][4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
        if constexpr (KB) {
          const int ds = __float_as_int(a.dscale);
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int t = 0; t < 4; ++t) kk2[h2][t] = __int_as_float(__builtin_amdgcn_sbfe((int)kw, 8 * qp2 + 4 * h2 + t, 1) & ds);
        } else if (a.thr != 0u) keep_pair(a, rowbase, jb * 32 + 16 * qp2, g, kk2[0], kk2[1]);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int q = 2 * qp2 + h2, r = 4 * q + t;
            const int key = jb * 32 + 8 * q + 4 * g + t;
            // p = exp(s scale - lse) as one fma + v_exp_f32 (base 2); dS = p scale (dP keep - delta)
            const float p = key < F ? __builtin_amdgcn_exp2f(fmaf(sc[r], c2, -l2)) : 0.f;
            sc[r] = (p * a.scale) * fmaf(dp[r], kk2[h2][t], -delta);
            dp[r] = p * kk2[h2][t];
          }
      }
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
        float vs[8], vp[8];
        pair_exchange(vs, sc[8 * p2], sc[8 * p2 + 1], sc[8 * p2 + 2], sc[8 * p2 + 3], sc[8 * p2 + 4], sc[8 * p2 + 5],
                      sc[8 * p2 + 6], sc[8 * p2 + 7], g);
        pair_exchange(vp, dp[8 * p2], dp[8 * p2 + 1], dp[8 * p2 + 2], dp[8 * p2 + 3], dp[8 * p2 + 4], dp[8 * p2 + 5],
                      dp[8 * p2 + 6], dp[8 * p2 + 7], g);
        if (!valid) {   // padded rows of the last row block contribute nothing to dK / dV (nor to a dQ that is never stored)
#pragma unroll
          for (int u = 0; u < 8; ++u) { vs[u] = 0.f; vp[u] = 0.f; }
        }
        const int to = tile_off<128>(row, jb * 4 + 2 * p2 + g);
        store8_h<HT>(reinterpret_cast<HT*>(sS + to), vs);
        store8_h<HT>(reinterpret_cast<HT*>(sP + to), vp);
      }
    };
    // ---- 1b unit: (row block, head-dim block) -> dQ over all keys, dS from LDS in tr_frag's k-order
    auto item_b = [&](int rb, int db) {
      const int row = rb * 32 + li;
      const bool valid = row < S;
      const int rowc = valid ? row : S - 1;
      const bool isq = rowc >= F;
      const HT* qp = base + (size_t)rowc * ld;
      f32x16_t qa;
#pragma unroll
      for (int r = 0; r < 16; ++r) qa[r] = 0.f;
      const float ds_self = sDs[row];
      // a query row's own key (the self term of dQ): requested before the products
      vec8<HT> ksf[2];
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ksf[p2][e] = (HT)0.f;
        if (isq) ksf[p2] = *reinterpret_cast<const vec8<HT>*>(qp + E + 32 * db + 16 * p2 + 8 * g);
      }
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) {
          // k-slot u of lane (row, g) <-> key 32 jb + 16 aa + 8 (u >> 2) + 4 g + (u & 3)   (mfma_tiles.h: tr_frag)
          typedef HT h4_t __attribute__((ext_vector_type(4)));
          const h4_t lo = *reinterpret_cast<const h4_t*>(sS + tile_off<128>(row, jb * 4 + 2 * aa) + 8 * g);
          const h4_t hi = *reinterpret_cast<const h4_t*>(sS + tile_off<128>(row, jb * 4 + 2 * aa + 1) + 8 * g);
          vec8<HT> sf;
          sf[0] = lo[0]; sf[1] = lo[1]; sf[2] = lo[2]; sf[3] = lo[3]; sf[4] = hi[0]; sf[5] = hi[1]; sf[6] = hi[2]; sf[7] = hi[3];
          const vec8<HT> kf = tr_frag<DH, HT>(sK, jb * 32 + 16 * aa, db, lane);
          qa = mfma16<HT>(kf, sf, qa);
        }
      HT* dq = dbase + (size_t)row * ld;
      const bool st = valid && !ATT_ABL(a, 2);
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
        float v[8];
        pair_exchange(v, qa[8 * p2], qa[8 * p2 + 1], qa[8 * p2 + 2], qa[8 * p2 + 3], qa[8 * p2 + 4], qa[8 * p2 + 5],
                      qa[8 * p2 + 6], qa[8 * p2 + 7], g);
        if (st) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = fmaf(ds_self, (float)ksf[p2][u], v[u]);   // (ds_self = 0 for feature rows)
          store8_h<HT>(dq + 32 * db + 16 * p2 + 8 * g, v);
        }
      }
    };
    // ---- the pipeline: step s = waves 0 - 3: dS / P~ of row block s (one key block each); waves 4 - 7: dQ of row block s - 1
    //      (one head-dim block each); every thread: sweep s + 1 into LDS.  Two barriers per step: Q / dO fragments read ->
    //      barrier -> the same space written.  Each SIMD hosts one wave of either kind.
    sweep_store(qv[0], dv[0], ov, kv, vv, 0);
    uint32_t kw_cur = 0u, kw_next = 0u;
    if constexpr (KB) { if (wave < 4) kw_cur = load_kw(0, wave); }
    lds_barrier();
    ATT_STAMP(2);
#pragma unroll
    for (int s_ = 0; s_ <= NSW; ++s_) {
      if (s_ <= nrb) {
        if (s_ + 1 < nrb) load_okv(s_ + 1);
        if constexpr (KB) { if (wave < 4 && s_ + 1 < nrb) kw_next = load_kw(s_ + 1, wave); }
        if (wave < 4 && s_ < nrb) read_frags(s_);
        lds_barrier();
        if (wave < 4) {
          if (s_ < nrb && !ATT_ABL(a, 64)) item_a(s_, wave, kw_cur);
          kw_cur = kw_next;
        } else {
          if (s_ >= 1 && !ATT_ABL(a, 32)) item_b(s_ - 1, wave - 4);
        }
        if (s_ + 1 < NSW) {
          if (s_ + 1 < nrb && !ATT_ABL(a, 128)) sweep_store(qv[(s_ + 1) % NSW], dv[(s_ + 1) % NSW], ov, kv, vv, s_ + 1);
        }
        lds_barrier();
      }
    }
    ATT_STAMP(3);
  } else {
    for (int rb = (FUSED ? 0 : part * a.rper) + wave; rb < nrb; rb += nwaves) {
      const int row = rb * 32 + li;
      const bool valid = row < S;
      const int rowc = valid ? row : S - 1;
      const bool isq = rowc >= F;
      const int rowl = ATT_ABL(a, 4) ? (rowc & 1) : rowc;
      const HT* qp = base + (size_t)rowl * ld;
      const HT* dop = dobase + (size_t)rowl * E;
      const HT* op = obase + (size_t)rowl * E;
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
        for (int qp = 0; qp < 2; ++qp) {
          float kk[2][4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
          if (a.thr != 0u) keep_pair(a, rowbase, jb * 32 + 16 * qp, g, kk[0], kk[1]);
  #pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
  #pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int q = 2 * qp + h2, r = 4 * q + t;
              const int key = jb * 32 + 8 * q + 4 * g + t;
              const float p = key < F ? __expf(sc[r] * a.scale - l) : 0.f;
              sc[r] = p * (dp[r] * kk[h2][t] - delta) * a.scale;
              dp[r] = p * kk[h2][t];
            }
        }
        // hand dS and the dropped probabilities P~ to the key-side kernel (16-byte stores: pair_exchange)
  #pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
          float vs[8], vp[8];
          pair_exchange(vs, sc[8 * p2], sc[8 * p2 + 1], sc[8 * p2 + 2], sc[8 * p2 + 3], sc[8 * p2 + 4], sc[8 * p2 + 5],
                        sc[8 * p2 + 6], sc[8 * p2 + 7], g);
          pair_exchange(vp, dp[8 * p2], dp[8 * p2 + 1], dp[8 * p2 + 2], dp[8 * p2 + 3], dp[8 * p2 + 4], dp[8 * p2 + 5],
                        dp[8 * p2 + 6], dp[8 * p2 + 7], g);
          if constexpr (FUSED) {
            if (!valid) {   // padded rows of the last row block contribute nothing to dK / dV
  #pragma unroll
              for (int u = 0; u < 8; ++u) { vs[u] = 0.f; vp[u] = 0.f; }
            }
            const int to = tile_off<128>(row, jb * 4 + 2 * p2 + g);
            store8_h<HT>(reinterpret_cast<HT*>(sS + to), vs);
            store8_h<HT>(reinterpret_cast<HT*>(sP + to), vp);
          } else if (valid && !ATT_ABL(a, 1)) {
            const size_t so = (size_t)row * FP + jb * 32 + 16 * p2 + 8 * g;
            store8_h<HT>(dSs + so, vs);
            store8_h<HT>(Pts + so, vp);
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
      {
        // lanes l and l ^ 32 trade quads (mfma_tiles.h: pair_exchange) so that every store is 16 bytes per lane
        HT* dq = dbase + (size_t)row * ld;
        const bool st = valid && !ATT_ABL(a, 2);
  #pragma unroll
        for (int db = 0; db < NDB; ++db)
  #pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            float v[8];
            pair_exchange(v, qa[db][8 * p2], qa[db][8 * p2 + 1], qa[db][8 * p2 + 2], qa[db][8 * p2 + 3], qa[db][8 * p2 + 4],
                          qa[db][8 * p2 + 5], qa[db][8 * p2 + 6], qa[db][8 * p2 + 7], g);
            const int dh = 32 * db + 16 * p2 + 8 * g;
            if (st) {
              if (isq) {
                const vec8<HT> kf = *reinterpret_cast<const vec8<HT>*>(qp + E + dh);
                // fp16: the q / dO chunks are the fragments the products above used (columns 16 kk + 8 g = dh) - no second read;
                // bf16: read again (kept live as operands AND as values to convert, the fragments cost this kernel 200 spilled
                // registers with this compiler)
                constexpr bool REUSE = sizeof(HT) == 2 && __is_same(HT, f16_t);
                const vec8<HT> qf8 = REUSE ? qf[2 * db + p2] : *reinterpret_cast<const vec8<HT>*>(qp + dh);
                const vec8<HT> d8 = REUSE ? df[2 * db + p2] : *reinterpret_cast<const vec8<HT>*>(dop + dh);
                float kself[8], vself[8];
  #pragma unroll
                for (int u = 0; u < 8; ++u) {
                  v[u] = fmaf(ds_self, (float)kf[u], v[u]);
                  kself[u] = ds_self * (float)qf8[u];      // a query token's own key / value receive the self term only
                  vself[u] = pt_self * (float)d8[u];
                }
                store8_h<HT>(dq + E + dh, kself);
                store8_h<HT>(dq + 2 * E + dh, vself);
              }
              store8_h<HT>(dq + dh, v);
            }
          }
      }
    }

  }

  if constexpr (FUSED) {
    // ---------------- phase 2: dK = dS^T Q, dV = P~^T dO for the F feature keys ----------------
    // out[key][dh] = sum_row Y[row][key] X[row][dh]; wave w: dh block w & 3 (32 columns), key half w >> 2 (two 32-key blocks)
    const int SP = (S + 31) & ~31;
    const int gid = lane >> 4, p = lane & 15, g2 = gid >> 1;
    const int row0 = 4 * g2 + (p >> 2);
    const int wi = wave & 3, wj = wave >> 2;
    int xtr[2], ytr[2][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      xtr[hh] = tile_off<128>(row0 + 8 * hh, 4 * wi + 2 * (gid & 1) + ((p & 3) >> 1)) + (p & 1) * 8;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        ytr[j][hh] = tile_off<128>(row0 + 8 * hh, 4 * (2 * wj + j) + 2 * (gid & 1) + ((p & 3) >> 1)) + (p & 1) * 8;
    }
    char* sX = smem;   // [SP][128] over the K / V tiles (SP * 256 B <= 64 KB for SP <= 256)
    // X rows travel global -> registers -> LDS in two steps so that a tile's load latency sits under something else: Q's under
    // the wait for the block's slowest row-block wave, dO's under the dK product
    constexpr int XU = 6;   // 16-byte chunks per thread: SP * 16 / 512 <= 6 for SP <= 192
    vec8<HT> xr[XU];
    auto x_load = [&](const HT* src, size_t lds_) {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int idx = tid + u * 512, row = idx >> 4, c = idx & 15;
        if (idx < SP * 16 && row < S) xr[u] = *reinterpret_cast<const vec8<HT>*>(src + (size_t)row * lds_ + c * 8);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) xr[u][e] = (HT)0.f;
        }
      }
    };
    auto x_store = [&]() {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int idx = tid + u * 512;
        if (idx < SP * 16) *reinterpret_cast<vec8<HT>*>(sX + tile_off<128>(idx >> 4, idx & 15)) = xr[u];
      }
    };
    ATT_STAMP(4);
    if constexpr (KS) {
#pragma unroll
      for (int u = 0; u < XU; ++u) xr[u] = qv[u];
    } else {
      x_load(base, ld);
    }
#pragma unroll
    for (int prod = 0; prod < 2; ++prod) {
      lds_barrier();   // phase 1 (prod 0) / the previous product's fragment reads (prod 1) are done with this space
      if (prod == 0) ATT_STAMP(5);
      x_store();
      if (prod == 0) {
        if constexpr (KS) {
#pragma unroll
          for (int u = 0; u < XU; ++u) xr[u] = dv[u];
        } else {
          x_load(dobase, (size_t)E);
        }
      }
      lds_barrier();
      if (prod == 1) ATT_STAMP(6);
      const char* sY = prod ? sP : sS;
      // two accumulator sets (even / odd 16-row steps: SP / 16 is even) - four independent MFMA chains per wave, and the
      // fragments of both steps of a pair are requested before the first MFMA
      f32x16_t acc[2], acc2[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; acc2[j][r] = 0.f; }
      {
        for (int ms = 0; ms < SP / 16; ms += 2) {
          const vec8<HT> xf = cat8<HT>(tr_read<HT>(sX + xtr[0] + ms * 4096), tr_read<HT>(sX + xtr[1] + ms * 4096));
          const vec8<HT> xg = cat8<HT>(tr_read<HT>(sX + xtr[0] + ms * 4096 + 4096), tr_read<HT>(sX + xtr[1] + ms * 4096 + 4096));
          vec8<HT> yf[2], yg[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            yf[j] = cat8<HT>(tr_read<HT>(sY + ytr[j][0] + ms * 4096), tr_read<HT>(sY + ytr[j][1] + ms * 4096));
            yg[j] = cat8<HT>(tr_read<HT>(sY + ytr[j][0] + ms * 4096 + 4096), tr_read<HT>(sY + ytr[j][1] + ms * 4096 + 4096));
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[j] = mfma16<HT>(xf, yf[j], acc[j]);
            acc2[j] = mfma16<HT>(xg, yg[j], acc2[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] += acc2[j][r];
        HT* out = dbase + (prod ? 2 * E : E);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = (2 * wj + j) * 32 + li;
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            float v[8];
            pair_exchange(v, acc[j][8 * p2], acc[j][8 * p2 + 1], acc[j][8 * p2 + 2], acc[j][8 * p2 + 3], acc[j][8 * p2 + 4],
                          acc[j][8 * p2 + 5], acc[j][8 * p2 + 6], acc[j][8 * p2 + 7], g);
            const int k = wi * 32 + 16 * p2 + 8 * g;
            if (n < F && !ATT_ABL(a, 2)) store8_h<HT>(out + (size_t)n * ld + k, v);
          }
        }
      }
    }
    ATT_STAMP(7);
  }
}

// dK / dV of the feature keys: out[key][dh] = sum_row Y[row][key] X[row][dh]
//   blockIdx.y = 0: Y = dS, X = Q  -> dK ;  blockIdx.y = 1: Y = P~, X = dO -> dV ;  blockIdx.z = (window, head)
template <typename HT>
__global__ __launch_bounds__(256) void attn_bwd_keys(const HT* __restrict__ dS_scr, const HT* __restrict__ Pt_scr,
                                                     const HT* __restrict__ qkv, const HT* __restrict__ d_o,
                                                     HT* __restrict__ dqkv, int S, int F, int FP, int E, int H,
                                                     int DH) {
  constexpr int WT = 128, WM = 64, TILE_BYTES = WM * WT * 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wn = wave & 1;
  const int bh = blockIdx.z, b = bh / H, h = bh % H, prod = blockIdx.y;
  const HT* Y = (prod ? Pt_scr : dS_scr) + (size_t)bh * S * FP;
  const int ldy = FP;
  const HT* X = prod ? d_o + (size_t)b * S * E + (size_t)h * DH : qkv + (size_t)b * S * 3 * E + (size_t)h * DH;
  const int ldx = prod ? E : 3 * E;
  const int xcols = prod ? E - h * DH : 3 * E - h * DH;   // columns of the row that lie at or after X's first column
  HT* out = dqkv + (size_t)b * S * 3 * E + (prod ? 2 * E : E) + (size_t)h * DH;
  const int ldo = 3 * E;
  const int n0 = blockIdx.x * WT, k0 = 0, M = S, N = F, K = DH;
  const int nsteps = (M + WM - 1) / WM;

  const int lrow = lane >> 4, lc = lane & 15;
  uint32_t yoff[4], xoff[4];
  int srow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 4 + lrow;
    const int c = lc ^ swz<128>(row);
    srow[i] = row;
    yoff[i] = (uint32_t)(((size_t)row * ldy + min(n0 + c * 8, ldy - 8)) * 2);
    // the tile is 128 columns wide whatever DH is: chunks past this head's columns are clamped INSIDE the row (X starts
    // h*DH columns into it), so that the last row of the last head never reads past the end of the buffer
    xoff[i] = (uint32_t)(((size_t)row * ldx + min(k0 + c * 8, xcols - 8)) * 2);
  }
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  auto stage = [&](int step, int buf) {
    const uint32_t bs = lds0 + buf * 2 * TILE_BYTES + wave * 4096;
    const int m0 = step * WM;
    if (m0 + WM <= M) {
      glds16_xn<4>(reinterpret_cast<const char*>(Y) + (size_t)m0 * ldy * 2, yoff, bs);
      glds16_xn<4>(reinterpret_cast<const char*>(X) + (size_t)m0 * ldx * 2, xoff, bs + TILE_BYTES);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int over = max(m0 + srow[i] - (M - 1), 0);
        glds16(reinterpret_cast<const char*>(Y) + ((size_t)m0 - over) * ldy * 2 + yoff[i], bs + i * 1024);
        glds16(reinterpret_cast<const char*>(X) + ((size_t)m0 - over) * ldx * 2 + xoff[i], bs + TILE_BYTES + i * 1024);
      }
    }
  };
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int xtr[2][2], ytr[2][2];
  {
    const int gid = lane >> 4, p = lane & 15, g = gid >> 1;
    const int row0 = 4 * g + (p >> 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cx = 4 * (wk * 2 + i) + 2 * (gid & 1) + ((p & 3) >> 1);
      const int cy = 4 * (wn * 2 + i) + 2 * (gid & 1) + ((p & 3) >> 1);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        xtr[i][hh] = tile_off<128>(row0 + 8 * hh, cx) + (p & 1) * 8;
        ytr[i][hh] = tile_off<128>(row0 + 8 * hh, cy) + (p & 1) * 8;
      }
    }
  }
  stage(0, 0);
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    glds_wait<0>();
    __syncthreads();
    if (st + 1 < nsteps) stage(st + 1, buf ^ 1);
    char* sY = lds + buf * 2 * TILE_BYTES;
    char* sX = sY + TILE_BYTES;
    if (st * WM + WM > M) {
      const int first = M - st * WM;
      vec8<HT> z;
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = (HT)0.f;
      for (int idx = tid; idx < (WM - first) * 16; idx += 256) {
        const int off = (first + idx / 16) * 256 + (idx % 16) * 16;
        *reinterpret_cast<vec8<HT>*>(sY + off) = z;
        *reinterpret_cast<vec8<HT>*>(sX + off) = z;
      }
      __syncthreads();
    }
#pragma unroll
    for (int ms = 0; ms < WM / 16; ++ms) {
      vec8<HT> xf[2], yf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) xf[i] = cat8<HT>(tr_read<HT>(sX + xtr[i][0] + ms * 4096), tr_read<HT>(sX + xtr[i][1] + ms * 4096));
#pragma unroll
      for (int j = 0; j < 2; ++j) yf[j] = cat8<HT>(tr_read<HT>(sY + ytr[j][0] + ms * 4096), tr_read<HT>(sY + ytr[j][1] + ms * 4096));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<HT>(xf[i], yf[j], acc[i][j]);
    }
  }
  // D[i = dh][j = key]: lane owns one key row of dK / dV, 4 consecutive dh per quad
  const int li = lane & 31, g = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {      // 16-byte stores: lanes l and l ^ 32 trade quads (mfma_tiles.h)
        float v[8];
        pair_exchange(v, acc[i][j][8 * p2], acc[i][j][8 * p2 + 1], acc[i][j][8 * p2 + 2], acc[i][j][8 * p2 + 3],
                      acc[i][j][8 * p2 + 4], acc[i][j][8 * p2 + 5], acc[i][j][8 * p2 + 6], acc[i][j][8 * p2 + 7], g);
        const int k = wk * 64 + i * 32 + 16 * p2 + 8 * g;
        if (n < N && k + 7 < K) store8_h<HT>(out + (size_t)n * ldo + k, v);
      }
  }
}

AttnArgsM make_args2(const TimDesc& d) {
  AttnArgsM a;
  a.S = d.S; a.F = d.F; a.E = d.E; a.H = d.H; a.LP = round_up(d.F + 1, 8);
  a.scale = 1.f / sqrtf((float)(d.E / d.H));
  a.thr = d.p_drop > 0.f ? drop_threshold(d.p_drop) : 0u;
  a.dscale = d.p_drop > 0.f ? 1.f / (1.f - d.p_drop) : 1.f;
  a.seed = d.seed; a.site = layer_site(d.layer, SITE_L_ATTN);
  a.abl = (d.reserved >> 8) & 0xff;
  a.rsplit = 1; a.rper = (d.S + 31) / 32;
  a.kbits = nullptr;
  return a;
}

static inline int rows_waves(int S) {
  int n = (S + 31) / 32;
  n = n < 1 ? 1 : (n > 8 ? 8 : n);
  const int w = tim_knobs().attn_waves;   // (A/B knob, as in attention_mfma.hip)
  if (w >= 1 && w <= 8) n = w;
  return n;
}

// fused form: DH = 128, 97..128 feature keys, K / V / dS / P~ within the 160 KB of LDS: S <= 192 for the one-wave-per-row-block
// form; the key-split pipeline (KS) keeps 12 bytes of per-row scalars more and fits S <= 160
static inline bool fused_fits(const TimDesc& d, bool ks) {
  if (tim_knobs().attn_fused == 0) return false;
  const int SP = (d.S + 31) & ~31;
  return d.E / d.H == 128 && (d.F + 31) / 32 == 4 &&
         (size_t)2 * 128 * 128 * 2 + (size_t)2 * SP * 128 * 2 + (ks ? (size_t)SP * 12 : 0) <= 160 * 1024;
}

template <typename HT, bool KS>
int launch_bwd_fused(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, void* stamps,
                     hipStream_t s, const unsigned long long* kbits) {
  const int SP = (d.S + 31) & ~31;
  const size_t lds = (size_t)2 * 128 * 128 * 2 + (size_t)2 * SP * 128 * 2 + (KS ? (size_t)SP * 12 : 0);
  AttnArgsM a = make_args2(d);
  if constexpr (KS) {
    if (kbits && a.thr != 0u) {
      a.kbits = kbits;
      (void)hipFuncSetAttribute((const void*)attn_bwd_rows<HT, 128, 4, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((attn_bwd_rows<HT, 128, 4, true, true, true>), dim3(d.B * d.H), dim3(512), lds, s, (const HT*)qkv, (const HT*)o,
                         lse, (const HT*)d_o, (HT*)dqkv, (HT*)stamps, (HT*)nullptr, a);
      return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
    }
  }
  (void)hipFuncSetAttribute((const void*)attn_bwd_rows<HT, 128, 4, true, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_bwd_rows<HT, 128, 4, true, KS>), dim3(d.B * d.H), dim3(512), lds, s, (const HT*)qkv, (const HT*)o, lse,
                     (const HT*)d_o, (HT*)dqkv, (HT*)stamps, (HT*)nullptr, a);
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}

template <typename HT, int DH, int NJB>
int launch_bwd2(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv,
                void* ws, hipStream_t s) {
  const int FP = NJB * 32;
  HT* dS = (HT*)ws;
  HT* Pt = dS + (size_t)d.B * d.H * d.S * FP;
  const size_t lds1 = (size_t)2 * FP * DH * 2;
  (void)hipFuncSetAttribute((const void*)attn_bwd_rows<HT, DH, NJB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
  // long sequences with few (window, head) pairs (detection: B * H = 128, 16 row blocks): the row blocks of a pair over several
  // workgroups of four waves, two workgroups per CU (the kernel takes 256 VGPRs: eight wave slots per CU)
  AttnArgsM a = make_args2(d);
  {
    const int nrb = (d.S + 31) / 32, bh = d.B * d.H;
    int want = bh >= 512 ? 1 : (512 + bh - 1) / bh;
    if (want > nrb / 4) want = nrb / 4;
    if (want < 1) want = 1;
    a.rper = (nrb + want - 1) / want;
    a.rsplit = (nrb + a.rper - 1) / a.rper;
  }
  const int waves = a.rsplit > 1 ? rows_waves(32 * (a.rper < 4 ? a.rper : 4)) : rows_waves(d.S);
  hipLaunchKernelGGL((attn_bwd_rows<HT, DH, NJB>), dim3(d.B * d.H * a.rsplit), dim3(64 * waves), lds1, s, (const HT*)qkv,
                     (const HT*)o, lse, (const HT*)d_o, (HT*)dqkv, dS, Pt, a);
  if (hipGetLastError() != hipSuccess) return TIMHIP_ELAUNCH;
  const size_t lds2 = 2 * 2 * 64 * 128 * 2;
  hipLaunchKernelGGL(attn_bwd_keys<HT>, dim3((d.F + 127) / 128, 2, d.B * d.H), dim3(256), lds2, s, dS, Pt, (const HT*)qkv,
                     (const HT*)d_o, (HT*)dqkv, d.S, d.F, FP, d.E, d.H, DH);
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}

}  // namespace

size_t tim_attention_bwd2_ws(const TimDesc& d) {
  return (size_t)2 * d.B * d.H * d.S * round_up(d.F, 32) * 2;
}

int tim_attention_bwd2_mfma(const TimDesc& d, const void* qkv, const void* o, const float* lse, const void* d_o,
                            void* dqkv, void* ws, size_t ws_bytes, hipStream_t s, const unsigned long long* kbits) {
  if (!h16_storage(d.precision) || (d.E % 8) != 0 || d.B * d.H > 65535) return TIMHIP_EUNSUPPORTED;
  const bool ks_ok = tim_knobs().attn_ks != 0 && fused_fits(d, true);
  if (ks_ok || fused_fits(d, false)) {   // (160 < S <= 192: the key-split form does not fit, the plain fused form still does)
    void* stamps = nullptr;   // (tuning builds, abl bit 16: per-block phase stamps into the caller's workspace, 64 B per block)
#ifdef TIMHIP_TUNING
    if (((d.reserved >> 8) & 16) && ws && ws_bytes >= (size_t)d.B * d.H * 64) stamps = ws;
#endif
    if (ks_ok) DISPATCH_H16(d.precision, return (launch_bwd_fused<HT, true>(d, qkv, o, lse, d_o, dqkv, stamps, s, kbits)));
    DISPATCH_H16(d.precision, return (launch_bwd_fused<HT, false>(d, qkv, o, lse, d_o, dqkv, stamps, s, nullptr)));
  }
  if (!ws || ws_bytes < tim_attention_bwd2_ws(d)) return TIMHIP_EUNSUPPORTED;
  const int DHv = d.E / d.H, NJBv = (d.F + 31) / 32;
#define B2(DHc, NJBc) DISPATCH_H16(d.precision, return (launch_bwd2<HT, DHc, NJBc>(d, qkv, o, lse, d_o, dqkv, ws, s)))
  if (DHv == 128) {
    switch (NJBv) { case 1: B2(128, 1); case 2: B2(128, 2); case 3: B2(128, 3); case 4: B2(128, 4); case 5: B2(128, 5); default: break; }
  } else if (DHv == 64) {
    switch (NJBv) { case 1: B2(64, 1); case 2: B2(64, 2); case 4: B2(64, 4); default: break; }
  } else if (DHv == 32) {
    switch (NJBv) { case 1: B2(32, 1); case 2: B2(32, 2); default: break; }
  }
#undef B2
  return TIMHIP_EUNSUPPORTED;
}
