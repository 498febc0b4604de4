// NT GEMM  C[M,N] = A[M,K] B[N,K]^T  for the encoder-layer shapes: the one-workgroup-per-CU "ping-pong" kernel.
//
// Why a second kernel.  gemm.hip's 160 x 128 tile runs two 4-wave blocks per CU; its main loop is bound by the operand path
// (71 FLOP per staged byte, 1.2 LDS fragment reads per MFMA - DESIGN.md section 5: 30-36 % MFMA busy).  This kernel is shaped
// for the matrix pipes instead:
//   * 160 x 256 output tile, ONE 8-wave block per CU (2 x 4 waves of 80 x 64): 98 FLOP per staged byte; M = 9920 = 62 x 160
//     rows give 248 / 496 / 744 tiles for N = 1024 / 2048 / 3072 - one, two, three full rounds of the 256 CUs;
//   * v_mfma_f32_16x16x32: an 80 x 64 wave tile is 5 x 4 MFMA tiles -> 9 fragment reads (ds_read_b128) feed 20 MFMAs per
//     32-deep contraction step (0.45 reads per MFMA), accumulators 80 VGPRs;
//   * the two waves that share a SIMD alternate roles ("ping-pong"): while one issues its 20 MFMAs the other reads its next
//     fragments; the LDS-DMA pieces of the stage two contraction steps ahead are issued between a wave's own MFMAs, where
//     issue slots are free.  The roles are kept apart by raw
//     s_barriers - waves 4-7 run one barrier (half a phase) behind waves 0-3 - so the matrix pipe of every SIMD always has
//     a wave in its MFMA phase and no wave has to interleave loads between its own MFMAs;
//   * 3-stage LDS ring of 52 KiB stages (156 KiB), filled by global_load_lds (inline asm, counted vmcnt: a stage stays in
//     flight across four barriers), XOR swizzle on the source address as in gemm.hip (conflict-free ds_read_b128).
// Epilogues: gemm_epi.h (shared with gemm.hip), on 16-row blocks transposed through a wave-private LDS region.
// Measured alternative (round 2, same box): the same tile without phase barriers - one barrier per contraction step, the two
// waves of a SIMD interleaving through the hardware scheduler - is 5-12 % slower per launch on the layer shapes (in-proj
// input gradient 67.6 vs 60.2 us), although an 8-wave s_barrier costs ~150 cycles (tools/wgpp_abl.py): for ds_read_b128
// operands the explicit alternation pays; for the transposing-read weight-gradient kernel (wgrad_pp.hip) it does not.
//
// Hazards (one LOAD + one MFMA phase per contraction step t; G0 = waves 0-3, G1 = waves 4-7, G1 one barrier behind):
//   RAW  stage t+1 is first read by G0 after the barrier that ends its MFMA phase of step t; every wave has waited (vmcnt)
//        for its own DMA pieces of stage t+1 in its LOAD phase of step t, i.e. at least one barrier earlier.
//   WAR  stage (t+2) % 3 = (t-1) % 3 is refilled from G0's MFMA phase of step t on; the last fragment reads of step t-1
//        (G1's LOAD phase) were waited for (lgkmcnt) BEFORE a barrier G0 has to pass to get there.
#include <stdlib.h>
#include <mutex>

#include "gemm_epi.h"

namespace {

constexpr int PP_BN = 256, PP_ROWB = 128, PP_NST = 3, PP_TNW = 4;

__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pp_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- epilogue: one wave's TMW x 4 accumulator tiles (16 x 16 each, D[n][m] orientation) ------------------------------------
// PFD: how many row blocks ahead the fp32 residual (+ LayerNorm statistics) is fetched.  TMW = everything before the first
// store (8-wave kernel, 256 VGPRs per wave); 2 = a two-deep ring refilled after each block's stores (12-wave kernel, 168 VGPRs).
struct PpNoSync { __device__ __forceinline__ void operator()() const {} };
// AFTER: called after every 16-row block (the dual-group kernel keeps its barrier cadence there; default: nothing)
// PFA (round 6): the same ring depth for the 16-bit aux rows (default: all row blocks before the first store, as before)
template <typename HT, int EPI, int TMW, int PFD = TMW, typename AFTER = PpNoSync, int PFA = TMW>
__device__ __forceinline__ void pp_epilogue(const EpiDev& e, const f32x4_t (&acc)[PP_TNW][TMW], int mw0, int nw0, int M, int N,
                                            float* ep, int lane, AFTER after = AFTER{}) {
  constexpr int EP_COLS = 64, EP_LD = EP_COLS + 4, CPR = EP_COLS / 4, OPR = EP_COLS / 8, RBS = 16;
  constexpr int NITQ = RBS * CPR / 64, NITO = RBS * OPR / 64;     // 4 quads / 2 octs per lane per row block
  const float asc = e.acc_scale ? *e.acc_scale : 1.f;
  const int frow = lane & 15, fq = lane >> 4;

  // dropout keep-bits drawn ahead of time (e.mask): this lane's bytes of every row block
  uint32_t mbyte[TMW][NITO];
  const bool use_mask = epi_uses_dropout(EPI) && epi_has_oct(EPI) && e.vec8 && e.mask && e.thr != 0u;
  static_for<TMW>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
#pragma unroll
    for (int it = 0; it < NITO; ++it) {
      const int idx = it * 64 + lane;
      const int m = mw0 + j * RBS + idx / OPR, n = nw0 + (idx % OPR) * 8;
      mbyte[j][it] = (use_mask && m < M && n + 7 < N) ? (uint32_t)e.mask[(size_t)m * e.ldmask + (n >> 3)] : 0u;
    }
  });
  // operands the epilogue reads per chunk (fp32 residual, 16-bit aux) are fetched for ALL row blocks before the first store:
  // a load placed after a store cannot be hoisted above it (possible aliasing) and would cost a memory latency per chunk
  constexpr bool PRE_RES = (EPI == TIMHIP_EPI_DROP_RES_F32 || EPI == TIMHIP_EPI_ADD_F32);
  constexpr bool PRE_AUX = (EPI == TIMHIP_EPI_DGELU_T || EPI == TIMHIP_EPI_DRELU_T || EPI == TIMHIP_EPI_MULAUX_T);
  float4 rbuf[PFD][PRE_RES ? NITQ : 1];
  float2 sbuf[PFD][PRE_RES ? NITQ : 1];
  vec8<HT> abuf[PFA][PRE_AUX ? NITO : 1];
  const bool pre_res = PRE_RES && e.vec && e.res != nullptr;
  const bool pre_ln = pre_res && EPI == TIMHIP_EPI_DROP_RES_F32 && e.ln_stats != nullptr;
  const bool pre_aux = PRE_AUX && e.vec8;
  auto fetch_res = [&](auto jc) {   // row block j -> ring entry j % PFD
    constexpr int j = decltype(jc)::value;
    if constexpr (PRE_RES) {
      if (pre_res) {
#pragma unroll
        for (int it = 0; it < NITQ; ++it) {
          const int idx = it * 64 + lane;
          const int m = mw0 + j * RBS + idx / CPR, n = nw0 + (idx % CPR) * 4;
          if (m < M && n + 3 < N) {   // (read once: nontemporal, as the 16-bit aux rows below - 0.6 % of the step against plain loads)
            typedef float f4_t __attribute__((ext_vector_type(4)));
            const f4_t q_ = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(e.res + (size_t)m * e.ldres + n));
            rbuf[j % PFD][it] = make_float4(q_[0], q_[1], q_[2], q_[3]);
          } else rbuf[j % PFD][it] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (pre_ln) sbuf[j % PFD][it] = m < M ? *reinterpret_cast<const float2*>(e.ln_stats + 2 * (size_t)m) : make_float2(0.f, 1.f);
        }
      }
    }
  };
  auto fetch_aux = [&](auto jc) {   // row block j -> ring entry j % PFA
    constexpr int j = decltype(jc)::value;
    if constexpr (PRE_AUX) {
      if (pre_aux) {
#pragma unroll
        for (int it = 0; it < NITO; ++it) {
          const int idx = it * 64 + lane;
          const int m = mw0 + j * RBS + idx / OPR, n = nw0 + (idx % OPR) * 8;
          if (m < M && n + 7 < N) abuf[j % PFA][it] = __builtin_nontemporal_load(reinterpret_cast<const vec8<HT>*>((const HT*)e.aux + (size_t)m * e.ldaux + n));
        }
      }
    }
  };
  static_for<TMW>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < PFD) fetch_res(jc);
    if constexpr (j < PFA) fetch_aux(jc);
  });
  // per-lane constants: the lane's output columns are the same in every chunk it handles
  float4 bias8[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, bias4 = bias8[0];
  bool pre_b8 = false, pre_b4 = false;
  if (e.bias && e.vec) {
    const int n8 = nw0 + (lane % OPR) * 8, n4 = nw0 + (lane % CPR) * 4;
    if (epi_has_oct(EPI) && e.vec8 && n8 + 7 < N) {
      bias8[0] = *reinterpret_cast<const float4*>(e.bias + n8);
      bias8[1] = *reinterpret_cast<const float4*>(e.bias + n8 + 4);
      pre_b8 = true;
    }
    if (n4 + 3 < N) { bias4 = *reinterpret_cast<const float4*>(e.bias + n4); pre_b4 = true; }
  }
  float4 lng = make_float4(1.f, 1.f, 1.f, 1.f), lnb = make_float4(0.f, 0.f, 0.f, 0.f);
  bool pre_gb = false;
  if (pre_ln) {
    const int n4 = nw0 + (lane % CPR) * 4;
    if (n4 + 3 < N) {
      lng = *reinterpret_cast<const float4*>(e.ln_w + n4);
      lnb = *reinterpret_cast<const float4*>(e.ln_b + n4);
      pre_gb = true;
    }
  }

  static_for<TMW>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    // lane (frow = m, fq): the 4 consecutive columns 16 i + 4 fq .. + 3 of row m, for the 4 column tiles i
#pragma unroll
    for (int i = 0; i < PP_TNW; ++i)
      *reinterpret_cast<float4*>(ep + frow * EP_LD + i * 16 + 4 * fq) =
          make_float4(acc[i][j][0] * asc, acc[i][j][1] * asc, acc[i][j][2] * asc, acc[i][j][3] * asc);
    pp_wait_lds();   // wave-private region: no block barrier needed
    if (epi_has_oct(EPI) && e.vec8) {
#pragma unroll
      for (int it = 0; it < NITO; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / OPR, ch = idx % OPR;
        const float4 lo = *reinterpret_cast<const float4*>(ep + row * EP_LD + ch * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ep + row * EP_LD + ch * 8 + 4);
        const int m = mw0 + j * RBS + row, n = nw0 + ch * 8;
        if (m < M && n + 7 < N) {
          epi_oct<EPI, HT>(e, m, n, N, lo, hi, mbyte[j][it], pre_aux, abuf[j % PFA][PRE_AUX ? it : 0], pre_b8, bias8[0], bias8[1]);
        } else if (m < M) {
          if (n < N) epi_quad<EPI, HT>(e, m, n, N, lo.x, lo.y, lo.z, lo.w);
          if (n + 4 < N) epi_quad<EPI, HT>(e, m, n + 4, N, hi.x, hi.y, hi.z, hi.w);
        }
      }
    } else {
      // dropout + residual: a lane's four quads of a row block sit at the same columns of rows r, r + 4, r + 8, r + 12, and lanes
      // 2 k / 2 k + 1 hold the two halves of one Philox counter's 8 elements in EVERY one of them.  Round 6: the even lane draws
      // the counter of quad `it`, the odd lane that of quad `it + 1`, and they trade halves (two DPP moves) - 2 calls per lane and
      // row block instead of 4, same counters, same bits (as in ln_bwd_kernel).  N % 8 == 0 keeps a pair inside one counter.
      bool has_k = false;
      if constexpr (EPI == TIMHIP_EPI_DROP_RES_F32 && (NITQ % 2) == 0) has_k = e.thr != 0u && !e.mask && e.vec && (N & 7) == 0 && e.pair;
      auto quad = [&](int it, bool hk, float4 kf) {
        const int idx = it * 64 + lane;
        const int row = idx / CPR, ch = idx % CPR;
        const float4 v = *reinterpret_cast<const float4*>(ep + row * EP_LD + ch * 4);
        const int m = mw0 + j * RBS + row, n = nw0 + ch * 4;
        if (m < M && n < N)
          epi_quad<EPI, HT>(e, m, n, N, v.x, v.y, v.z, v.w, pre_res && n + 3 < N, rbuf[j % PFD][PRE_RES ? it : 0], pre_b4 && n + 3 < N,
                            bias4, pre_ln && pre_gb && n + 3 < N, sbuf[j % PFD][PRE_RES ? it : 0], lng, lnb, hk, kf);
      };
      if (has_k) {
        if constexpr ((NITQ % 2) == 0) {
#pragma unroll
          for (int it = 0; it < NITQ; it += 2) {   // (one pair's eight factors live at a time: the 12-wave kernel sits at its 168 registers)
            const int idx = (it + (lane & 1)) * 64 + (lane & ~1);
            const long long m_ = mw0 + j * RBS + idx / CPR, n_ = nw0 + (idx % CPR) * 4;
            const Philox4 r = philox4x32_7(e.seed, e.site, (uint64_t)(m_ * (long long)N + n_) >> 3);
            const bool odd = (lane & 1) != 0;
            const uint32_t sa = odd ? r.x : r.z, sb = odd ? r.y : r.w;
            const uint32_t pa = (uint32_t)__builtin_amdgcn_mov_dpp((int)sa, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
            const uint32_t pb = (uint32_t)__builtin_amdgcn_mov_dpp((int)sb, 0xB1, 0xF, 0xF, true);
            float4 ka, kb;
            drop_mask4_words(odd ? pa : r.x, odd ? pb : r.y, e.thr, e.scale, ka.x, ka.y, ka.z, ka.w);
            drop_mask4_words(odd ? r.z : pa, odd ? r.w : pb, e.thr, e.scale, kb.x, kb.y, kb.z, kb.w);
            quad(it, true, ka);
            quad(it + 1, true, kb);
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < NITQ; ++it) quad(it, false, make_float4(1.f, 1.f, 1.f, 1.f));
      }
    }
    if constexpr (j + PFD < TMW) fetch_res(std::integral_constant<int, j + PFD>{});   // refill this block's ring entry
    if constexpr (j + PFA < TMW) fetch_aux(std::integral_constant<int, j + PFA>{});
    pp_wait_lds();   // reads done before the next row block overwrites the region
    after();
  });
}

// ---- main loop of one wave group (G = 0: waves 0-3; G = 1: waves 4-7, one barrier behind) ---------------------------------
// A stage is 20 + 32 = 52 LDS-DMA pieces of 1 KiB (8 rows x 128 B): the A tile's pieces first, then the B tile's, in LDS as
// in this numbering.  G0 waves issue 7 of them per contraction step (4 during their first MFMA phase, 3 during the second),
// G1 waves 6 (3 + 3): the issue slots between a wave's own MFMAs are free (the matrix pipe is busy for 16 cycles per MFMA),
// while a DMA piece issued in a LOAD phase lengthens the phase the partner's MFMAs have to cover.
template <int TMW, int G> struct PpShare {
  static constexpr int CNT = G == 0 ? 7 : 6, P0 = G == 0 ? 4 : 3;   // pieces per step (P0: unused since the phases were merged)
  static_assert(4 * 7 + 4 * 6 == 2 * TMW * 2 + 32, "the share table is written for the 160 x 256 tile");
};

// ABL (tuning builds): 0 the kernel; 1 no DMA in the loop (the ring keeps its prologue contents); 2 no MFMAs; 3 DMA only (no
// fragment reads, no MFMAs); LATE: this wave's pieces of stage t + 1 are waited for at the END of MFMA phase t (counted wait,
// the pieces of stage t + 2 stay in flight) instead of at the end of LOAD phase t: half a step more flight time
// (G0 only: G1's pieces of stage t + 1 are read by G0 half a step after G1's LOAD phase t ends, so G1 has to keep its wait there)
template <typename HT, int TMW, int G, bool PROF = false, int ABL = 0, bool LATE_ = false>
__device__ __forceinline__ void pp_mainloop(const HT* __restrict__ A, const HT* __restrict__ B, int nk, const char* lds, uint32_t lds0,
                                            const uint32_t (&off)[7], int first_piece, int a_frag, int b_frag, int c0, int c1,
                                            f32x4_t (&acc)[PP_TNW][TMW], long long* prof = nullptr, bool primed = false,
                                            int a_wrap = 0x7fffffff) {
  constexpr int BM = 32 * TMW, A_PIECES = BM / 8;
  constexpr int A_BYTES = BM * PP_ROWB, B_BYTES = PP_BN * PP_ROWB, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int CNT = PpShare<TMW, G>::CNT;
  constexpr bool LATE = LATE_ && G == 0;

  // this wave's i-th piece of stage `kt`, into ring slot `slot`
  auto piece = [&](int kt, int slot, int i) {   // (a_wrap: contraction steps after which the A operand repeats, EpiDev.a_wrap)
    const int p = first_piece + i;
    if constexpr (ABL == 4) {   // tuning: what would tile-major operands buy?  off[] = packed offsets (tile, piece, lane), see the kernel
      const char* g = p < A_PIECES ? reinterpret_cast<const char*>(A) + (size_t)kt * A_BYTES : reinterpret_cast<const char*>(B) + (size_t)kt * B_BYTES;
      glds16_s(uniform_ptr(g), off[i], lds0 + slot * ST_BYTES + p * 1024);
      return;
    }
    const char* g = p < A_PIECES ? reinterpret_cast<const char*>(A) + (size_t)(kt >= a_wrap ? kt - a_wrap : kt) * PP_ROWB
                                 : reinterpret_cast<const char*>(B) + (size_t)kt * PP_ROWB;
    glds16_s(uniform_ptr(g), off[i], lds0 + slot * ST_BYTES + p * 1024);
  };

  // prologue: stages 0 and 1 in flight, stage 0 landed.  primed (persistent tile loop): both were issued before the previous
  // tile's epilogue; everything this wave has outstanding - the two stages and that epilogue's stores - is waited for
  if (primed) {
    glds_wait<0>();
  } else {
#pragma unroll
    for (int i = 0; i < CNT; ++i) piece(0, 0, i);
    if (nk > 1) {
#pragma unroll
      for (int i = 0; i < CNT; ++i) piece(1, 1, i);
      glds_wait<CNT>();
    } else {
      glds_wait<0>();
    }
  }
  pp_barrier();
  if constexpr (G == 1) pp_barrier();   // half a phase behind G0 from here on

  vec8<HT> xa[TMW], wb[PP_TNW];
  int slot = 0;
  long long tp = 0, t_load = 0, t_bar_a = 0, t_mma = 0, t_bar_b = 0;   // PROF (tuning builds): shader cycles per phase part
  if constexpr (PROF) tp = __builtin_readcyclecounter();
  auto lap = [&](long long& acc_t) {
    if constexpr (PROF) { const long long now = __builtin_readcyclecounter(); acc_t += now - tp; tp = now; }
  };
  // one contraction step; MORE: stage t + 2 exists and is issued during this step's MFMA phases (a compile-time flag: the
  // last two steps run a copy of the loop body without the DMA pieces instead of branching around each of them)
  // One LOAD and one MFMA phase per contraction step (the fragments of both 32-deep halves are read in one go: 18
  // ds_read_b128, 72 VGPRs; 40 MFMAs per phase): two barriers per step.  An 8-wave s_barrier costs ~150 cycles
  // (tools/wgpp_abl.py), so with a phase per half step the four barriers were a third of the loop's time.
  vec8<HT> xb[TMW], wc[PP_TNW];   // second half's fragments
  auto step = [&](int t, auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;
    const char* base = lds + slot * ST_BYTES;
    const int nslot = slot >= 1 ? slot - 1 : PP_NST - 1;   // (t + 2) % 3
    // ---- LOAD phase: the fragments of step t
    if constexpr (ABL < 3) {   // (3 ... 8: DMA only)
#pragma unroll
      for (int j = 0; j < TMW; ++j) xa[j] = *reinterpret_cast<const vec8<HT>*>(base + a_frag + j * (16 * PP_ROWB) + c0);
#pragma unroll
      for (int i = 0; i < PP_TNW; ++i) wb[i] = *reinterpret_cast<const vec8<HT>*>(base + b_frag + i * (16 * PP_ROWB) + c0);
#pragma unroll
      for (int j = 0; j < TMW; ++j) xb[j] = *reinterpret_cast<const vec8<HT>*>(base + a_frag + j * (16 * PP_ROWB) + c1);
#pragma unroll
      for (int i = 0; i < PP_TNW; ++i) wc[i] = *reinterpret_cast<const vec8<HT>*>(base + b_frag + i * (16 * PP_ROWB) + c1);
    }
    if constexpr (ABL == 5) glds_wait<2 * CNT>();        // tuning, DMA only: two / four more stages in flight than the ring allows
    else if constexpr (ABL == 6) glds_wait<4 * CNT>();   // (nobody reads the data): is the DMA stream latency- or rate-bound?
    else if constexpr (!LATE) glds_wait<0>();   // every piece of stage t + 1 issued by this wave (during the previous MFMA phase) has landed
    pp_wait_lds();
    lap(t_load);
    pp_barrier();
    lap(t_bar_a);
    // ---- MFMA phase, with this wave's DMA pieces for step t + 2 between the MFMAs (after every fifth one)
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < TMW; ++j)
#pragma unroll
        for (int i = 0; i < PP_TNW; ++i) {
          if constexpr (ABL < 2) acc[i][j] = half == 0 ? mfma16x16<HT>(wb[i], xa[j], acc[i][j]) : mfma16x16<HT>(wc[i], xb[j], acc[i][j]);
          else if constexpr (ABL == 2) {   // keep the fragments alive
            if (half == 0) asm volatile("" ::"v"(wb[i]), "v"(xa[j])); else asm volatile("" ::"v"(wc[i]), "v"(xb[j]));
          }
          const int q = half * (TMW * PP_TNW) + j * PP_TNW + i, k = q / 5;
          if (MORE && ABL != 1 && q % 5 == 2 && k < CNT) {
            __builtin_amdgcn_sched_barrier(0);
            piece(t + 2, nslot, k);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    if constexpr (LATE) {   // stage t + 1 (issued during MFMA phase t - 1) has landed; this phase's pieces of stage t + 2 may fly
      __builtin_amdgcn_sched_barrier(0);
      if (MORE && ABL != 1) glds_wait<CNT>(); else glds_wait<0>();
    }
    lap(t_mma);
    pp_barrier();
    lap(t_bar_b);
    slot = slot + 1 == PP_NST ? 0 : slot + 1;
  };
  int t = 0;
  for (; t + 2 < nk; ++t) step(t, std::true_type{});
  for (; t < nk; ++t) step(t, std::false_type{});
  if constexpr (G == 0) pp_barrier();   // as many barriers as G1
  if constexpr (PROF) { prof[0] = t_load; prof[1] = t_bar_a; prof[2] = t_mma; prof[3] = t_bar_b; }
}

template <typename HT, int EPI, int TMW, bool PROF = false, int ABL = 0, bool LATE = false>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb,
                                                         int M, int N, int K, EpiDev e) {
  constexpr int BM = 32 * TMW;
  constexpr int A_BYTES = BM * PP_ROWB;
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [3][A tile BM x 128 B | B tile 256 x 128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_n = (N + PP_BN - 1) / PP_BN, tiles_m = (M + BM - 1) / BM;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * PP_BN;

  // LDS-DMA source offsets of this wave's pieces (PpShare): a 1-KiB piece = 8 rows x 128 B, lane -> (row, 16-B chunk); the
  // chunk index is swizzled on the SOURCE side so that the LDS image stays lane-linear
  const int lrow = lane >> 3, lchunk = lane & 7;
  const int first_piece = wm == 0 ? 7 * wn : 28 + 6 * wn;
  uint32_t off[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int p = first_piece + i;
    const bool is_a = p < BM / 8;
    const int row = (is_a ? p : p - BM / 8) * 8 + lrow;    // row of the A / B tile
    const int c = (lchunk ^ kswz<64>(row)) * 8;
    off[i] = is_a ? (uint32_t)(((size_t)min(m0 + row, M - 1) * lda + c) * 2) : (uint32_t)(((size_t)min(n0 + row, N - 1) * ldb + c) * 2);
    if constexpr (ABL == 7)   // tuning, DMA only: every block stages tile (0, 0) - all L2 hits after the first touch
      off[i] = is_a ? (uint32_t)(((size_t)row * lda + c) * 2) : (uint32_t)(((size_t)row * ldb + c) * 2);
    if constexpr (ABL == 8)   // tuning, DMA only: every block of an XCD stages the XCD's first tile - one set of misses per XCD and step
      off[i] = is_a ? (uint32_t)(((size_t)min((blockIdx.x & 7) * BM + row, M - 1) * lda + c) * 2) : (uint32_t)(((size_t)row * ldb + c) * 2);
    if constexpr (ABL == 4)   // operand tiles [tile][K / 64][rows][64]: a stage's tile is one contiguous block, a piece 1 KiB of it
      off[i] = is_a ? (uint32_t)((size_t)(m0 / BM) * (K / 64) * A_BYTES + p * 1024 + lane * 16)
                    : (uint32_t)((size_t)(n0 / PP_BN) * (K / 64) * (PP_BN * PP_ROWB) + (p - BM / 8) * 1024 + lane * 16);
  }
  // fragment reads: lane -> (row = lane & 15 of a 16-row tile, 16-B chunk 4 half + (lane >> 4)), swizzled like the stage
  const int frow = lane & 15, fk = lane >> 4, sw = (frow >> 1) & 7;
  const int c0 = (fk ^ sw) << 4, c1 = ((fk + 4) ^ sw) << 4;
  const int a_frag = (wm * 16 * TMW + frow) * PP_ROWB;
  const int b_frag = A_BYTES + (wn * 64 + frow) * PP_ROWB;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));

  f32x4_t acc[PP_TNW][TMW];
#pragma unroll
  for (int i = 0; i < PP_TNW; ++i)
#pragma unroll
    for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = K / 64;
  long long prof[4] = {0, 0, 0, 0}, t_begin = 0, w_begin = 0;
  if constexpr (PROF) { t_begin = __builtin_readcyclecounter(); w_begin = wall_clock64(); }
  const int a_wrap = e.a_wrap > 0 ? e.a_wrap : 0x7fffffff;
  if (wm == 0) pp_mainloop<HT, TMW, 0, PROF, ABL, LATE>(A, B, nk, lds, lds0, off, first_piece, a_frag, b_frag, c0, c1, acc, prof, false, a_wrap);
  else pp_mainloop<HT, TMW, 1, PROF, ABL, LATE>(A, B, nk, lds, lds0, off, first_piece, a_frag, b_frag, c0, c1, acc, prof, false, a_wrap);
  long long t_loop = 0;
  if constexpr (PROF) t_loop = __builtin_readcyclecounter();

  __syncthreads();   // every wave is done with the stage ring: it becomes the epilogue's transposition space
  float* ep = reinterpret_cast<float*>(lds) + wave * (16 * 68);
  pp_epilogue<HT, EPI, TMW>(e, acc, m0 + wm * 16 * TMW, n0 + wn * 64, M, N, ep, lane);
  if constexpr (PROF) {   // tools/pp_phase.py: e.aux -> 16 counters, [0..7] wave 0 (G0), [8..15] wave 4 (G1), summed over blocks
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t_end = __builtin_readcyclecounter();
    if (lane == 0 && (wave == 0 || wave == 4) && e.aux) {
      unsigned long long* c = (unsigned long long*)e.aux + (wave == 4 ? 8 : 0);
      for (int i = 0; i < 4; ++i) atomicAdd(c + i, (unsigned long long)prof[i]);
      atomicAdd(c + 4, (unsigned long long)(t_loop - t_begin));
      atomicAdd(c + 5, 1ull);
      atomicAdd(c + 6, (unsigned long long)(wall_clock64() - w_begin));
      atomicAdd(c + 7, (unsigned long long)(t_end - t_loop));
    }
  }
}

// ---- persistent-tile form (gemm_nt_pt_kernel, round 3) -------------------------------------------------------------------------
// For the shapes that run two or three rounds of 160 x 256 tiles (N = 2048 / 3072 at M = 9920: 496 / 744 tiles) a block walks its
// tiles itself - the 2 or 3 consecutive column tiles of one row panel (the A panel stays in its XCD's L2) - and issues the
// NEXT tile's first two stages before it starts the current tile's epilogue: the epilogue transposes through ring slot 2,
// slots 0 and 1 fill meanwhile, and the next main loop starts without a block launch, a cold prologue or a ragged last round.
// (The epilogue itself still runs with the matrix pipes idle - see gemm_nt_dg_kernel below for the measured attempt at that.)
template <typename HT, int EPI, int TMW>
__global__ __launch_bounds__(512) void gemm_nt_pt_kernel(const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb,
                                                         int M, int N, int K, EpiDev e, int tpb) {
  constexpr int BM = 32 * TMW;
  constexpr int A_BYTES = BM * PP_ROWB, ST_BYTES = (BM + PP_BN) * PP_ROWB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_n = (N + PP_BN - 1) / PP_BN;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int lrow = lane >> 3, lchunk = lane & 7;
  const int first_piece = wm == 0 ? 7 * wn : 28 + 6 * wn;
  const int cnt = wm == 0 ? 7 : 6;
  const int frow = lane & 15, fk = lane >> 4, sw = (frow >> 1) & 7;
  const int c0 = (fk ^ sw) << 4, c1 = ((fk + 4) ^ sw) << 4;
  const int a_frag = (wm * 16 * TMW + frow) * PP_ROWB;
  const int b_frag = A_BYTES + (wn * 64 + frow) * PP_ROWB;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  const int nk = K / 64;

  uint32_t off[7];
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int t) {
    m0 = (t / tiles_n) * BM; n0 = (t % tiles_n) * PP_BN;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int p = first_piece + i;
      const bool is_a = p < BM / 8;
      const int row = (is_a ? p : p - BM / 8) * 8 + lrow;
      const int c = (lchunk ^ kswz<64>(row)) * 8;
      off[i] = is_a ? (uint32_t)(((size_t)min(m0 + row, M - 1) * lda + c) * 2) : (uint32_t)(((size_t)min(n0 + row, N - 1) * ldb + c) * 2);
    }
  };
  auto prime = [&]() {   // stages 0 and 1 of the tile `off` describes -> slots 0 and 1
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int i = 0; i < 7; ++i)
        if (i < cnt) {
          const int p = first_piece + i;
          const char* g = reinterpret_cast<const char*>(p < BM / 8 ? (const void*)A : (const void*)B) + (size_t)st * PP_ROWB;
          glds16_s(uniform_ptr(g), off[i], lds0 + st * ST_BYTES + p * 1024);
        }
  };

  f32x4_t acc[PP_TNW][TMW];
  set_tile(lb * tpb);
  for (int k = 0; k < tpb; ++k) {
#pragma unroll
    for (int i = 0; i < PP_TNW; ++i)
#pragma unroll
      for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (wm == 0) pp_mainloop<HT, TMW, 0>(A, B, nk, lds, lds0, off, first_piece, a_frag, b_frag, c0, c1, acc, nullptr, k > 0);
    else pp_mainloop<HT, TMW, 1>(A, B, nk, lds, lds0, off, first_piece, a_frag, b_frag, c0, c1, acc, nullptr, k > 0);
    __syncthreads();   // every wave is done with the stage ring
    const int em0 = m0 + wm * 16 * TMW, en0 = n0 + wn * 64;
    if (k + 1 < tpb) {   // the next tile's first two stages fly during this tile's epilogue
      set_tile(lb * tpb + k + 1);
      prime();
    }
    float* ep = reinterpret_cast<float*>(lds + 2 * ST_BYTES) + wave * (16 * 68);
    pp_epilogue<HT, EPI, TMW>(e, acc, em0, en0, M, N, ep, lane);
  }
}

// ---- loader-wave form (gemm_nt_ld_kernel, 12 waves): see wgrad_pp.hip (wl_consume / wl_load) for the barrier windows --------
// Waves 0-7 keep the merged LOAD / MFMA phases without a single DMA instruction; waves 8-11 issue the stage's 52 pieces as 13
// groups of four (A: groups 0-4, B: groups 5-12; loaders take 4 / 3 / 3 / 3 groups), two groups per wave in window A_t, the rest
// in window B_t.
// L2 prefetch (round 3): the DMA stream of the loop is what bounds it (the loop with nothing but its DMA pieces takes as long as
// the whole loop), and a fifth of it is the misses: 18-30 % of a stage's lines are first touches of this XCD's L2, they return
// from the fabric in 1-2 us, and loads return in order - nearly every 8-line piece waits for one.  With every line an L2 hit the
// DMA-only loop is 20-30 % shorter.  So each tile touches ITS SHARE of the lines its XCD will need pf_dist stages later with one
// plain dword load per line: the A panel's 160 rows are split between the tiles_n column tiles that share it, the B tile's 256
// rows between the 8 consecutive row panels that run on one XCD.  One load instruction per step from wave 0 (A) and wave 4 (B)
// of the consumer group, which have no other vector-memory operation in the loop (loads return in order: a wave that also
// issued DMA pieces would wait for its prefetches).  The loaded dword is never used.
struct PlPrefetch { const void* base; uint32_t off; int on, dist; };

// ONEBAR (gemm_nt_ld kernel with one barrier per contraction step): with no DMA instruction and no vmcnt wait in the consumer
// waves, the antiphase of the two groups can come from program order instead of a second barrier -
//     interval t (between barriers B_t and B_t+1):   G1:  LOAD(t);  MFMA(t)          G0:  MFMA(t - 1);  LOAD(t)
//     loaders:  issue stage t + 2 -> slot (t + 2) % 3;  wait until stage t + 1 has landed;  barrier
// G1's MFMAs follow G0's on the pipe without a barrier in between, G0 reads its fragments under G1's MFMAs, and a stage has one
// to two whole steps of flight time (issued in interval t, waited for at the end of interval t + 1, read in interval t + 2).
//   RAW  stage t is read in interval t by both groups; the loaders waited for it before B_t.
//   WAR  slot (t + 2) % 3 = (t - 1) % 3 is refilled in interval t; both groups read stage t - 1 in interval t - 1 (lgkmcnt
//        before B_t).
template <typename HT, int TMW, int G, bool ONEBAR = false>
__device__ __forceinline__ void pl_consume(int nk, const char* lds, int a_frag, int b_frag, int c0, int c1, f32x4_t (&acc)[PP_TNW][TMW],
                                           const PlPrefetch& pf) {
  constexpr int BM = 32 * TMW;
  constexpr int ST_BYTES = (BM + PP_BN) * PP_ROWB;
  if constexpr (ONEBAR) {
    vec8<HT> xa[TMW], wb[PP_TNW], xb[TMW], wc[PP_TNW];
    uint32_t pf_sink = 0;
    auto load = [&](int slot) {
      const char* base = lds + slot * ST_BYTES;
#pragma unroll
      for (int j = 0; j < TMW; ++j) xa[j] = *reinterpret_cast<const vec8<HT>*>(base + a_frag + j * (16 * PP_ROWB) + c0);
#pragma unroll
      for (int i = 0; i < PP_TNW; ++i) wb[i] = *reinterpret_cast<const vec8<HT>*>(base + b_frag + i * (16 * PP_ROWB) + c0);
#pragma unroll
      for (int j = 0; j < TMW; ++j) xb[j] = *reinterpret_cast<const vec8<HT>*>(base + a_frag + j * (16 * PP_ROWB) + c1);
#pragma unroll
      for (int i = 0; i < PP_TNW; ++i) wc[i] = *reinterpret_cast<const vec8<HT>*>(base + b_frag + i * (16 * PP_ROWB) + c1);
    };
    auto mma = [&]() {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int j = 0; j < TMW; ++j)
#pragma unroll
          for (int i = 0; i < PP_TNW; ++i)
            acc[i][j] = half == 0 ? mfma16x16<HT>(wb[i], xa[j], acc[i][j]) : mfma16x16<HT>(wc[i], xb[j], acc[i][j]);
    };
    int slot = 0;
    pp_barrier();   // B_0
    for (int t = 0; t < nk; ++t) {
      if (pf.on && t + pf.dist < nk) {
        const char* g = reinterpret_cast<const char*>(pf.base) + pf.off + (size_t)(t + pf.dist) * PP_ROWB;
        asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(g) : "memory");
      }
      if constexpr (G == 1) {
        load(slot);
        pp_wait_lds();
        __builtin_amdgcn_sched_barrier(0);
        mma();
      } else {
        if (t > 0) mma();
        __builtin_amdgcn_sched_barrier(0);
        load(slot);
        pp_wait_lds();
      }
      pp_barrier();   // B_t+1
      slot = slot + 1 == PP_NST ? 0 : slot + 1;
    }
    if constexpr (G == 0) mma();   // MFMA(nk - 1)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink) :: "memory");
    return;
  }
  pp_barrier();                        // b0
  if constexpr (G == 1) pp_barrier();
  vec8<HT> xa[TMW], wb[PP_TNW], xb[TMW], wc[PP_TNW];
  int slot = 0;
  uint32_t pf_sink = 0;
  for (int t = 0; t < nk; ++t) {
    if (pf.on && t + pf.dist < nk) {
      const char* g = reinterpret_cast<const char*>(pf.base) + pf.off + (size_t)(t + pf.dist) * PP_ROWB;
      // "+v": the sink stays one dedicated register from here to the wait after the loop - the data returns asynchronously, a
      // register the compiler considered dead after the asm would be reused (e.g. for the next address) and overwritten late
      asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(g) : "memory");
    }
    const char* base = lds + slot * ST_BYTES;
#pragma unroll
    for (int j = 0; j < TMW; ++j) xa[j] = *reinterpret_cast<const vec8<HT>*>(base + a_frag + j * (16 * PP_ROWB) + c0);
#pragma unroll
    for (int i = 0; i < PP_TNW; ++i) wb[i] = *reinterpret_cast<const vec8<HT>*>(base + b_frag + i * (16 * PP_ROWB) + c0);
#pragma unroll
    for (int j = 0; j < TMW; ++j) xb[j] = *reinterpret_cast<const vec8<HT>*>(base + a_frag + j * (16 * PP_ROWB) + c1);
#pragma unroll
    for (int i = 0; i < PP_TNW; ++i) wc[i] = *reinterpret_cast<const vec8<HT>*>(base + b_frag + i * (16 * PP_ROWB) + c1);
    pp_wait_lds();
    pp_barrier();
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < TMW; ++j)
#pragma unroll
        for (int i = 0; i < PP_TNW; ++i)
          acc[i][j] = half == 0 ? mfma16x16<HT>(wb[i], xa[j], acc[i][j]) : mfma16x16<HT>(wc[i], xb[j], acc[i][j]);
    pp_barrier();
    slot = slot + 1 == PP_NST ? 0 : slot + 1;
  }
  if constexpr (G == 0) pp_barrier();
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink) :: "memory");   // the sink register is free for reuse only now
}

template <typename HT, int TMW, bool ONEBAR = false>
__device__ __forceinline__ void pl_load(const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb, int M, int N, int m0, int n0,
                                        int nk, uint32_t lds0, int lw, int lane) {
  constexpr int BM = 32 * TMW, A_GROUPS = BM / 32;
  constexpr int ST_BYTES = (BM + PP_BN) * PP_ROWB;
  static_assert(A_GROUPS == 5 || A_GROUPS == 4, "group tables: the 160 x 256 tile (13 groups of 32 rows) and the 128 x 256 tile (12)");
  // 160 x 256: groups 0-3 | 4-6 | 7-9 | 10-12;  128 x 256: three groups per loader wave
  const int g0 = A_GROUPS == 5 ? (lw == 0 ? 0 : 1 + 3 * lw) : 3 * lw, ng = (A_GROUPS == 5 && lw == 0) ? 4 : 3;
  const int lrow = lane >> 3, lchunk = lane & 7;
  uint32_t off[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int p = g0 * 4 + i;
    const bool is_a = p < BM / 8;
    const int row = (is_a ? p : p - BM / 8) * 8 + lrow;
    const int c = (lchunk ^ kswz<64>(row)) * 8;
    off[i] = is_a ? (uint32_t)(((size_t)min(m0 + row, M - 1) * lda + c) * 2) : (uint32_t)(((size_t)min(n0 + min(row, PP_BN - 1), N - 1) * ldb + c) * 2);
  }
  auto group = [&](int kt, int slot, int q) {   // this wave's q-th group of stage kt
    if (q >= ng) return;
    const int gq = g0 + q;
    const char* g = reinterpret_cast<const char*>(gq < A_GROUPS ? (const void*)A : (const void*)B) + (size_t)kt * PP_ROWB;
    glds16_x4(uniform_ptr(g), off[4 * q], off[4 * q + 1], off[4 * q + 2], off[4 * q + 3], lds0 + slot * ST_BYTES + gq * 4096);
  };
  group(0, 0, 0); group(0, 0, 1); group(0, 0, 2); group(0, 0, 3);
  if (nk > 1) {
    group(1, 1, 0); group(1, 1, 1); group(1, 1, 2); group(1, 1, 3);
    if (ng == 4) glds_wait<16>(); else glds_wait<12>();
  } else {
    glds_wait<0>();
  }
  pp_barrier();   // b0
  int slot = 0;
  if constexpr (ONEBAR) {
    for (int t = 0; t < nk; ++t) {
      const int nslot = slot >= 1 ? slot - 1 : PP_NST - 1;
      if (t + 2 < nk) {
        group(t + 2, nslot, 0); group(t + 2, nslot, 1); group(t + 2, nslot, 2); group(t + 2, nslot, 3);
        if (ng == 4) glds_wait<16>(); else glds_wait<12>();   // stage t + 1 (issued one interval earlier) has landed
      } else {
        glds_wait<0>();
      }
      pp_barrier();        // B_t+1
      slot = slot + 1 == PP_NST ? 0 : slot + 1;
    }
    return;
  }
  for (int t = 0; t < nk; ++t) {
    const int nslot = slot >= 1 ? slot - 1 : PP_NST - 1;
    const bool more = t + 2 < nk;
    if (more) {            // window A_t
      group(t + 2, nslot, 0); group(t + 2, nslot, 1);
      glds_wait<8>();      // stage t + 1 (issued one step earlier) has landed
    } else {
      glds_wait<0>();
    }
    pp_barrier();          // b_2t+1
    if (more) { group(t + 2, nslot, 2); group(t + 2, nslot, 3); }   // window B_t
    pp_barrier();          // b_2t+2
    slot = slot + 1 == PP_NST ? 0 : slot + 1;
  }
  pp_barrier();            // b_2nk+1
}

template <typename HT, int EPI, int TMW, bool ONEBAR = false>
__global__ __launch_bounds__(768) void gemm_nt_ld_kernel(const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb,
                                                         int M, int N, int K, EpiDev e, int pf_dist, int pf_mode) {
  constexpr int BM = 32 * TMW;
  constexpr int A_BYTES = BM * PP_ROWB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (N + PP_BN - 1) / PP_BN, tiles_m = (M + BM - 1) / BM;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * PP_BN;
  const int nk = K / 64;
  if (wave >= 8) {
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    pl_load<HT, TMW, ONEBAR>(A, lda, B, ldb, M, N, m0, n0, nk, lds0, wave - 8, lane);
    __syncthreads();
    return;
  }
  const int wm = wave >> 2, wn = wave & 3;
  const int frow = lane & 15, fk = lane >> 4, sw = (frow >> 1) & 7;
  const int c0 = (fk ^ sw) << 4, c1 = ((fk + 4) ^ sw) << 4;
  const int a_frag = (wm * 16 * TMW + frow) * PP_ROWB;
  const int b_frag = A_BYTES + (wn * 64 + frow) * PP_ROWB;
  f32x4_t acc[PP_TNW][TMW];
#pragma unroll
  for (int i = 0; i < PP_TNW; ++i)
#pragma unroll
    for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // this tile's share of the XCD's L2 prefetch, one line per lane: the list [A rows a0 .. a0 + a_cnt) [B rows b0 .. b0 + b_cnt), 64
  // entries per consumer wave.  pf_mode 1: the A panel's rows split between the tiles_n column tiles that share it, the B tile's
  // rows between the row panels that run on one XCD at a time (32 CUs / tiles_n, at most 8); 2: every line of the tile
  PlPrefetch pf{nullptr, 0u, 0, pf_dist};
  if (pf_dist > 0) {
    const int conc = max(1, min(8, 32 / tiles_n));
    const int a_cnt = pf_mode == 2 ? BM : (BM + tiles_n - 1) / tiles_n, a0 = pf_mode == 2 ? 0 : (t % tiles_n) * a_cnt;
    const int b_cnt = pf_mode == 2 ? PP_BN : PP_BN / conc, b0 = pf_mode == 2 ? 0 : ((t / tiles_n) % conc) * b_cnt;
    const int li = wave * 64 + lane;
    if (li < a_cnt) {
      const int row = a0 + li;
      pf.base = A; pf.on = row < BM && m0 + row < M;
      pf.off = (uint32_t)((size_t)min(m0 + row, M - 1) * lda * 2);
    } else if (li - a_cnt < b_cnt) {
      const int row = b0 + li - a_cnt;
      pf.base = B; pf.on = n0 + row < N;
      pf.off = (uint32_t)((size_t)min(n0 + row, N - 1) * ldb * 2);
    }
  }
  if (wm == 0) pl_consume<HT, TMW, 0, ONEBAR>(nk, lds, a_frag, b_frag, c0, c1, acc, pf);
  else pl_consume<HT, TMW, 1, ONEBAR>(nk, lds, a_frag, b_frag, c0, c1, acc, pf);
  __syncthreads();   // every wave is done with the stage ring: it becomes the epilogue's transposition space
  float* ep = reinterpret_cast<float*>(lds) + wave * (16 * 68);
  pp_epilogue<HT, EPI, TMW, 2>(e, acc, m0 + wm * 16 * TMW, n0 + wn * 64, M, N, ep, lane);
}

// tpb > 1 (round 3, the shapes of two or three rounds of tiles): a block walks tpb consecutive column tiles of its row panel
// itself.  The grid is then ONE round of co-resident blocks that stay in step - 8 row panels x (tiles_n / tpb) blocks per XCD,
// every block on column tile k of its group at the same time - so the L2 prefetch shares below are as meaningful as for the
// single-round shapes (with one tile per block the tiles of later rounds drift apart, and a share covered a twelfth of a
// panel).  The loader waves issue the next tile's first two stages right after the barrier that ends a tile's loop, i.e. under
// the consumers' epilogue, which transposes through ring slot 2.
// (a kernel of its own: the epilogues of the one-tile kernel above sit at the 168-VGPR limit of three waves per SIMD, and the
//  same source with a tile loop around it spilled up to 132 registers there; this one spills 1-8 on the epilogues that use it)
template <typename HT, int EPI, int TMW>
__global__ __launch_bounds__(768) void gemm_nt_ldp_kernel(const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb,
                                                          int M, int N, int K, EpiDev e, int pf_dist, int pf_mode, int tpb) {
  constexpr int BM = 32 * TMW;
  constexpr int A_BYTES = BM * PP_ROWB, ST_BYTES = (BM + PP_BN) * PP_ROWB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (N + PP_BN - 1) / PP_BN, tiles_m = (M + BM - 1) / BM;
  const int lb = xcd_remap(blockIdx.x, (tiles_m * tiles_n) / tpb);
  const int nk = K / 64;
  const int groups_n = tiles_n / tpb;   // blocks per row panel (tpb divides tiles_n)
  if (wave >= 8) {
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    for (int k = 0; k < tpb; ++k) {
      const int t = lb * tpb + k;
      pl_load<HT, TMW>(A, lda, B, ldb, M, N, (t / tiles_n) * BM, (t % tiles_n) * PP_BN, nk, lds0, wave - 8, lane);
      __syncthreads();
    }
    return;
  }
  const int wm = wave >> 2, wn = wave & 3;
#pragma unroll 1
  for (int k = 0; k < tpb; ++k) {
    // the lane id is laundered once per tile: everything the epilogue derives from it (its LDS offsets, column offsets, ...)
    // is loop-invariant, the compiler hoists it in front of the tile loop, and 20 registers held across the main loop spill
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int t = __builtin_amdgcn_readfirstlane(lb * tpb + k);
    const int m0 = __builtin_amdgcn_readfirstlane((t / tiles_n) * BM), n0 = __builtin_amdgcn_readfirstlane((t % tiles_n) * PP_BN);
    // (fragment offsets recomputed per tile: not live across the epilogue)
    const int frow = ln & 15, fk = ln >> 4, sw = (frow >> 1) & 7;
    const int c0 = (fk ^ sw) << 4, c1 = ((fk + 4) ^ sw) << 4;
    const int a_frag = (wm * 16 * TMW + frow) * PP_ROWB;
    const int b_frag = A_BYTES + (wn * 64 + frow) * PP_ROWB;
    f32x4_t acc[PP_TNW][TMW];
#pragma unroll
    for (int i = 0; i < PP_TNW; ++i)
#pragma unroll
      for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // this tile's share of the XCD's L2 prefetch, one line per lane: the list [A rows a0 .. a0 + a_cnt) [B rows b0 .. b0 + b_cnt), 64
    // entries per consumer wave.  pf_mode 1: the A panel's rows split between the blocks that share the panel at a time, the B
    // tile's rows between the row panels that run on one XCD at a time (32 CUs / blocks per panel, at most 8); 2: every line of the tile
    PlPrefetch pf{nullptr, 0u, 0, pf_dist};
    if (pf_dist > 0) {
      const int conc = max(1, min(8, 32 / groups_n));
      const int a_cnt = pf_mode == 2 ? BM : (BM + groups_n - 1) / groups_n, a0 = pf_mode == 2 ? 0 : ((t % tiles_n) / tpb) * a_cnt;
      const int b_cnt = pf_mode == 2 ? PP_BN : PP_BN / conc, b0 = pf_mode == 2 ? 0 : ((t / tiles_n) % conc) * b_cnt;
      const int li = wave * 64 + ln;
      if (li < a_cnt) {
        const int row = a0 + li;
        pf.base = A; pf.on = row < BM && m0 + row < M;
        pf.off = (uint32_t)((size_t)min(m0 + row, M - 1) * lda * 2);
      } else if (li - a_cnt < b_cnt) {
        const int row = b0 + li - a_cnt;
        pf.base = B; pf.on = n0 + row < N;
        pf.off = (uint32_t)((size_t)min(n0 + row, N - 1) * ldb * 2);
      }
    }
    if (wm == 0) pl_consume<HT, TMW, 0>(nk, lds, a_frag, b_frag, c0, c1, acc, pf);
    else pl_consume<HT, TMW, 1>(nk, lds, a_frag, b_frag, c0, c1, acc, pf);
    __syncthreads();   // every wave is done with the stage ring: slot 2 (slot 0 for a single tile) becomes the epilogue's transposition space
    float* ep = reinterpret_cast<float*>(lds + 2 * ST_BYTES) + wave * (16 * 68);
    pp_epilogue<HT, EPI, TMW, 2>(e, acc, m0 + wm * 16 * TMW, n0 + wn * 64, M, N, ep, ln);
  }
}

// ---- eight-phase form (gemm_nt_p8_kernel, round 6): 256 x 256 / 320 x 256 tiles for the multi-round shapes ---------------------
// The three products of a layer that run two or three rounds of 160 x 256 tiles (in-projection forward N = 3072; linear1 forward
// and linear2's input gradient N = 2048) were the slowest of the eight (920 - 940 TFLOP/s with plain stores against 950 - 1200 for
// the one-round shapes; hipBLASLt's 256 x 256 macro tile reached 1090 on the in-projection): every extra tile of a block's walk
// costs an epilogue the matrix pipes sit out, and a 160-row tile stages 98 FLOP per byte.  This kernel gives them ONE tile per CU
// and round again: 320 x 256 (N = 2048: 31 x 8 = 248 tiles, 142 FLOP per staged byte) or 256 x 256 (N = 3072: 39 x 12 = 468
// tiles = 1.83 rounds, 128 FLOP per byte), on the schedule of cdna_hip_programming.md section 5 ("the 256^2 8-phase template"):
//   * 8 waves = 2 (M) x 4 (N), wave tile 16 TM x 64 (TM = 8 / 10 MFMA row tiles; 128 / 160 accumulator registers), NO loader
//     waves (their 168-register budget cannot hold the accumulators);
//   * a contraction step (64 deep) is FOUR phases, one quadrant of the wave tile each - (rows 0, cols 0) (0, 1) (1, 1) (1, 0) - so
//     that consecutive phases share one operand's fragments: 12 / 4 / TM / 4 ds_read_b128 feed 2 TM MFMAs per phase;
//   * each phase: [fragment reads; ONE half-tile of LDS-DMA pieces (2 - 3 per wave); lgkmcnt(0)] barrier [MFMAs] barrier; the two
//     waves of a SIMD run one barrier apart (waves 4 - 7 behind waves 0 - 3), so a SIMD's matrix pipe always has a wave in its MFMA
//     segment while the partner reads / issues;
//   * LDS: two contraction-step buffers of (BM + 256) x 128 B (128 / 144 KiB).  A step's tile is staged as four half-tiles
//     (A rows of row-quadrant 0 / 1, B rows of column-quadrant 0 / 1) into the space whose last fragment read is ONE phase old:
//       phase 1 of step t: B0 of step t + 1 -> other buffer;  phases 2 / 3 / 4: A0 / B1 / A1 of step t + 2 -> this buffer;
//     ONE counted wait per step (phase 4: vmcnt = the three newest half-tiles stay in flight) - the DMA stream is never drained.
// Hazards (segment = the code between two barriers; group 0 reads in segment 2p and multiplies in 2p + 1, group 1 one later):
//   WAR  every fragment read of phase p is complete (lgkmcnt(0) BEFORE the barrier) by the end of segment 2p + 1; the earliest
//        DMA piece into that space is issued in phase p + 1's read segment: 2p + 2 (group 0) / 2p + 3 (group 1).
//   RAW  the counted wait of phase 4 sits before that phase's first barrier in both groups (segments 2p, 2p + 1); the data is
//        first read in the next phase (segments 2p + 2, 2p + 3), i.e. behind a barrier every waiting wave has passed.
template <int TM, int G> struct P8Share {
  static_assert(TM == 8 || TM == 10, "half-tile tables: the 256- and the 320-row tile");
  static constexpr int NA = TM == 8 ? 2 : (G == 0 ? 3 : 2);   // A half-tile = 2 TM pieces of 1 KiB: 16 = 8 x 2, 20 = 4 x 3 + 4 x 2
  static constexpr int NB = 2;                                 // B half-tile = 16 pieces
};
struct P8Tab { uint32_t offA[2][3], offB[2][2]; int dstA[2][3], dstB[2][2]; };

// VAR (experiment arms, fp16 plain-store instances only - TIMHIP_GEMM_P8_VAR): bit 0 no s_setprio around the MFMA segments,
// bit 1 the fragment reads are waited for AFTER the phase's first barrier (the guide's order) instead of before it
template <typename HT, int TM, int G, int VAR = 0>
__device__ __forceinline__ void p8_mainloop(const HT* __restrict__ A, const HT* __restrict__ B, int nk, const char* lds, uint32_t lds0,
                                            const P8Tab& tb, int a_frag, int b_frag, int c0, int c1, f32x4_t (&acc)[PP_TNW][TM]) {
  constexpr int BM = 32 * TM, KT = (BM + PP_BN) * PP_ROWB, HM = TM / 2;
  constexpr int NA = P8Share<TM, G>::NA, NB = P8Share<TM, G>::NB;
  auto stage_a = [&](int kt, uint32_t buf, int h) {
    const void* g = uniform_ptr(reinterpret_cast<const char*>(A) + (size_t)kt * PP_ROWB);
#pragma unroll
    for (int i = 0; i < NA; ++i) glds16_s(g, tb.offA[h][i], buf + tb.dstA[h][i]);
  };
  auto stage_b = [&](int kt, uint32_t buf, int h) {
    const void* g = uniform_ptr(reinterpret_cast<const char*>(B) + (size_t)kt * PP_ROWB);
#pragma unroll
    for (int i = 0; i < NB; ++i) glds16_s(g, tb.offB[h][i], buf + tb.dstB[h][i]);
  };
  // prologue: steps 0 and 1 whole
  stage_a(0, lds0, 0); stage_b(0, lds0, 0); stage_b(0, lds0, 1); stage_a(0, lds0, 1);
  if (nk > 1) { stage_a(1, lds0 + KT, 0); stage_b(1, lds0 + KT, 0); stage_b(1, lds0 + KT, 1); stage_a(1, lds0 + KT, 1); }
  glds_wait<0>();
  pp_barrier();
  if constexpr (G == 1) pp_barrier();   // one segment behind group 0 from here on

  vec8<HT> xa[HM][2], wb[2][2];
  auto read_a = [&](const char* bb, int qm) {
#pragma unroll
    for (int i = 0; i < HM; ++i) {
      xa[i][0] = *reinterpret_cast<const vec8<HT>*>(bb + a_frag + (qm * 8 * TM + i * 16) * PP_ROWB + c0);
      xa[i][1] = *reinterpret_cast<const vec8<HT>*>(bb + a_frag + (qm * 8 * TM + i * 16) * PP_ROWB + c1);
    }
  };
  auto read_b = [&](const char* bb, int qn) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      wb[j][0] = *reinterpret_cast<const vec8<HT>*>(bb + b_frag + (qn * 32 + j * 16) * PP_ROWB + c0);
      wb[j][1] = *reinterpret_cast<const vec8<HT>*>(bb + b_frag + (qn * 32 + j * 16) * PP_ROWB + c1);
    }
  };
  auto mma = [&](auto qm_c, auto qn_c) {
    constexpr int qm = decltype(qm_c)::value, qn = decltype(qn_c)::value;
    if constexpr (VAR & 2) { pp_wait_lds(); __builtin_amdgcn_sched_barrier(0); }
    if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < HM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[2 * qn + j][qm * HM + i] = mfma16x16<HT>(wb[j][kh], xa[i][kh], acc[2 * qn + j][qm * HM + i]);
    if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);
  };
  auto wait_reads = [&]() { if constexpr (!(VAR & 2)) pp_wait_lds(); };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  for (int t = 0; t < nk; ++t) {
    const int b = t & 1;
    const char* bb = lds + b * KT;
    const uint32_t cur = lds0 + b * KT, oth = lds0 + (b ^ 1) * KT;
    const bool s1 = t >= 1 && t + 1 < nk, s2 = t + 2 < nk;
    // phase 1: quadrant (0, 0)
    read_b(bb, 0); read_a(bb, 0);
    if (s1) stage_b(t + 1, oth, 0);
    wait_reads(); pp_barrier();
    mma(I0{}, I0{});
    pp_barrier();
    // phase 2: quadrant (0, 1) - A fragments kept
    read_b(bb, 1);
    if (s2) stage_a(t + 2, cur, 0);
    wait_reads(); pp_barrier();
    mma(I0{}, I1{});
    pp_barrier();
    // phase 3: quadrant (1, 1) - B fragments kept
    read_a(bb, 1);
    if (s2) stage_b(t + 2, cur, 1);
    wait_reads(); pp_barrier();
    mma(I1{}, I1{});
    pp_barrier();
    // phase 4: quadrant (1, 0) - A fragments kept; the step's one counted wait: B0 of step t + 1 (and everything older) has landed
    read_b(bb, 0);
    if (s2) { stage_a(t + 2, cur, 1); glds_wait<2 * NA + NB>(); } else { glds_wait<0>(); }
    wait_reads(); pp_barrier();
    mma(I1{}, I0{});
    pp_barrier();
  }
  if constexpr (G == 0) pp_barrier();   // as many barriers as group 1
}

// Two phases per step (round 6, after the weight-gradient kernel's: wgrad_pp.hip w8_mainloop2): phase A = row-half 0 against BOTH
// column halves (4 HM MFMAs: A0, B0, B1 read, B kept), phase B = row-half 1 (A1 read) - half the barriers per MFMA, B0 read once,
// 16 fragment registers more.  Restaging one phase after a half-tile's last read: phase A of step t: A1 of step t + 1 -> other
// buffer; phase B: A0, B0, B1 of step t + 2 -> this buffer, then the step's counted wait (those NA + 2 NB pieces stay in flight).
template <typename HT, int TM, int G>
__device__ __forceinline__ void p8_mainloop2(const HT* __restrict__ A, const HT* __restrict__ B, int nk, const char* lds, uint32_t lds0,
                                             const P8Tab& tb, int a_frag, int b_frag, int c0, int c1, f32x4_t (&acc)[PP_TNW][TM]) {
  constexpr int BM = 32 * TM, KT = (BM + PP_BN) * PP_ROWB, HM = TM / 2;
  constexpr int NA = P8Share<TM, G>::NA, NB = P8Share<TM, G>::NB;
  auto stage_a = [&](int kt, uint32_t buf, int h) {
    const void* g = uniform_ptr(reinterpret_cast<const char*>(A) + (size_t)kt * PP_ROWB);
#pragma unroll
    for (int i = 0; i < NA; ++i) glds16_s(g, tb.offA[h][i], buf + tb.dstA[h][i]);
  };
  auto stage_b = [&](int kt, uint32_t buf, int h) {
    const void* g = uniform_ptr(reinterpret_cast<const char*>(B) + (size_t)kt * PP_ROWB);
#pragma unroll
    for (int i = 0; i < NB; ++i) glds16_s(g, tb.offB[h][i], buf + tb.dstB[h][i]);
  };
  stage_a(0, lds0, 0); stage_b(0, lds0, 0); stage_b(0, lds0, 1); stage_a(0, lds0, 1);
  if (nk > 1) { stage_a(1, lds0 + KT, 0); stage_b(1, lds0 + KT, 0); stage_b(1, lds0 + KT, 1); stage_a(1, lds0 + KT, 1); }
  glds_wait<0>();
  pp_barrier();
  if constexpr (G == 1) pp_barrier();   // one segment behind group 0 from here on

  vec8<HT> xa[HM][2], wb[4][2];
  auto read_a = [&](const char* bb, int qm) {
#pragma unroll
    for (int i = 0; i < HM; ++i) {
      xa[i][0] = *reinterpret_cast<const vec8<HT>*>(bb + a_frag + (qm * 8 * TM + i * 16) * PP_ROWB + c0);
      xa[i][1] = *reinterpret_cast<const vec8<HT>*>(bb + a_frag + (qm * 8 * TM + i * 16) * PP_ROWB + c1);
    }
  };
  auto read_b = [&](const char* bb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // (column quadrant qn = j / 2: rows qn * 32 + (j % 2) * 16 = j * 16 of the wave's 64)
      wb[j][0] = *reinterpret_cast<const vec8<HT>*>(bb + b_frag + (j * 16) * PP_ROWB + c0);
      wb[j][1] = *reinterpret_cast<const vec8<HT>*>(bb + b_frag + (j * 16) * PP_ROWB + c1);
    }
  };
  auto mma = [&](auto qm_c) {
    constexpr int qm = decltype(qm_c)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < HM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j][qm * HM + i] = mfma16x16<HT>(wb[j][kh], xa[i][kh], acc[j][qm * HM + i]);
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  for (int t = 0; t < nk; ++t) {
    const int b = t & 1;
    const char* bb = lds + b * KT;
    const uint32_t cur = lds0 + b * KT, oth = lds0 + (b ^ 1) * KT;
    const bool s1 = t >= 1 && t + 1 < nk, s2 = t + 2 < nk;
    // phase A: row-half 0, both column halves
    read_b(bb); read_a(bb, 0);
    if (s1) stage_a(t + 1, oth, 1);
    pp_wait_lds(); pp_barrier();
    mma(I0{});
    pp_barrier();
    // phase B: row-half 1 - B fragments kept; the step's one counted wait: A1 of step t + 1 (and everything older) has landed
    read_a(bb, 1);
    if (s2) { stage_a(t + 2, cur, 0); stage_b(t + 2, cur, 0); stage_b(t + 2, cur, 1); glds_wait<NA + 2 * NB>(); } else { glds_wait<0>(); }
    pp_wait_lds(); pp_barrier();
    mma(I1{});
    pp_barrier();
  }
  if constexpr (G == 0) pp_barrier();   // as many barriers as group 1
}

template <typename HT, int EPI, int TM, int VAR = 0>
__global__ __launch_bounds__(512) void gemm_nt_p8_kernel(const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb,
                                                         int M, int N, int K, EpiDev e) {
  constexpr int BM = 32 * TM, A_BYTES = BM * PP_ROWB;
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [2][A tile BM x 128 B | B tile 256 x 128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (N + PP_BN - 1) / PP_BN, tiles_m = (M + BM - 1) / BM;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * PP_BN;
  // this wave's DMA pieces of the four half-tiles: piece q of an A half = 8 rows at (q / TM) * 16 TM + h * 8 TM + (q % TM) * 8,
  // of a B half at (q / 4) * 64 + h * 32 + (q % 4) * 8; lane -> (row, 16-byte chunk), the chunk swizzled on the SOURCE side
  const int lrow = lane >> 3, lchunk = lane & 7;
  const int na = TM == 8 ? 2 : (wr == 0 ? 3 : 2);
  const int qa0 = TM == 8 ? 2 * wave : (wr == 0 ? 3 * wave : 12 + 2 * (wave - 4));
  P8Tab tb;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int q = qa0 + (i < na ? i : 0);
      const int r0 = (q / TM) * 16 * TM + h * 8 * TM + (q % TM) * 8, row = r0 + lrow;
      const int c = (lchunk ^ kswz<64>(row)) * 8;
      tb.offA[h][i] = (uint32_t)(((size_t)min(m0 + row, M - 1) * lda + c) * 2);
      tb.dstA[h][i] = r0 * PP_ROWB;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = 2 * wave + i;
      const int r0 = (q / 4) * 64 + h * 32 + (q % 4) * 8, row = r0 + lrow;
      const int c = (lchunk ^ kswz<64>(row)) * 8;
      tb.offB[h][i] = (uint32_t)(((size_t)min(n0 + row, N - 1) * ldb + c) * 2);
      tb.dstB[h][i] = A_BYTES + r0 * PP_ROWB;
    }
  }
  // fragment reads: lane -> (row = lane & 15 of a 16-row tile, 16-byte chunk 4 half + (lane >> 4)), swizzled like the stage
  const int frow = lane & 15, fk = lane >> 4, sw = (frow >> 1) & 7;
  const int c0 = (fk ^ sw) << 4, c1 = ((fk + 4) ^ sw) << 4;
  const int a_frag = (wr * 16 * TM + frow) * PP_ROWB;
  const int b_frag = A_BYTES + (wc * 64 + frow) * PP_ROWB;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  f32x4_t acc[PP_TNW][TM];
#pragma unroll
  for (int i = 0; i < PP_TNW; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nk = K / 64;
  if constexpr ((VAR & 4) != 0) {   // two phases per step (the default; TIMHIP_GEMM_P8_PH=4: four)
    if (wr == 0) p8_mainloop2<HT, TM, 0>(A, B, nk, lds, lds0, tb, a_frag, b_frag, c0, c1, acc);
    else p8_mainloop2<HT, TM, 1>(A, B, nk, lds, lds0, tb, a_frag, b_frag, c0, c1, acc);
  } else {
    if (wr == 0) p8_mainloop<HT, TM, 0, VAR>(A, B, nk, lds, lds0, tb, a_frag, b_frag, c0, c1, acc);
    else p8_mainloop<HT, TM, 1, VAR>(A, B, nk, lds, lds0, tb, a_frag, b_frag, c0, c1, acc);
  }
  __syncthreads();   // every wave is done with the buffers: they become the epilogue's transposition space
  float* ep = reinterpret_cast<float*>(lds) + wave * (16 * 68);
  pp_epilogue<HT, EPI, TM, 2, PpNoSync, 2>(e, acc, m0 + wr * 16 * TM, n0 + wc * 64, M, N, ep, lane);
}

template <typename HT, int EPI, int TM>
void launch_p8(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const EpiDev& e, hipStream_t s) {
  constexpr int BM = 32 * TM;
  const size_t shmem = (size_t)2 * (BM + PP_BN) * PP_ROWB;
  static PerDeviceOnce attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute((const void*)gemm_nt_p8_kernel<HT, EPI, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  const dim3 grid(((M + BM - 1) / BM) * ((N + PP_BN - 1) / PP_BN));
#ifdef TIMHIP_P8_VARIANTS   // experiment arms (tools/p8_ab.py with TIMHIP_GEMM_P8_VAR): plain-store fp16 instances only
  if constexpr (EPI == TIMHIP_EPI_STORE_T && sizeof(HT) == 2 && __is_same(HT, f16_t)) {
    static const int var = getenv("TIMHIP_GEMM_P8_VAR") ? atoi(getenv("TIMHIP_GEMM_P8_VAR")) : 0;
    const int v = getenv("TIMHIP_GEMM_P8_VAR") ? atoi(getenv("TIMHIP_GEMM_P8_VAR")) : var;
#define P8V(V) if (v == V) { (void)hipFuncSetAttribute((const void*)gemm_nt_p8_kernel<HT, EPI, TM, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
      hipLaunchKernelGGL((gemm_nt_p8_kernel<HT, EPI, TM, V>), grid, dim3(512), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N, K, e); return; }
    P8V(1) P8V(2) P8V(3)
#undef P8V
  }
#endif
  if (tim_knobs().gemm_p8_ph != 4) {   // two phases per step: the default (in-projection forward 60.0 -> 56.8 us, linear1 forward 54.5 -> 51.9: profiles/r06_af_*)
    static PerDeviceOnce attr2;
    if (attr2.first())
      (void)hipFuncSetAttribute((const void*)gemm_nt_p8_kernel<HT, EPI, TM, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL((gemm_nt_p8_kernel<HT, EPI, TM, 4>), grid, dim3(512), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N, K, e);
    return;
  }
  hipLaunchKernelGGL((gemm_nt_p8_kernel<HT, EPI, TM>), grid, dim3(512), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N, K, e);
}

// Which shapes: the ones that run MORE than one round of 160 x 256 tiles, when one of the two big tiles fills its rounds to at
// least 85 % (useful tile area / (rounds x 256 CUs)).  C2a, M = 9920: N = 3072 -> 256 rows (0.91), N = 2048 -> 320 rows (0.97),
// N = 1024 stays on the one-round 160-row kernel.  TIMHIP_GEMM_P8 = 0: off; 1 (default): by shape, the 16-bit store and the GELU
// epilogue; 2: by shape, also the multiply-by-saved-factor epilogue (linear2's input gradient: measured 40.2 against 46.8 us
// with a plain store but 53.3 against 52.6 with its epilogue - 80 MB of aux rows and results leave through 8 waves instead of
// 12 - profiles/r06_b_p8_ab.txt); 8 / 10: that tile for every legal shape and all three epilogues (tests)
static int p8_choice(int epi, int M, int N, int K, const EpiDev& e) {
  const int kn = tim_knobs().gemm_p8;
  if (kn == 0 || e.a_wrap != 0 || K % 64 || K < 128 || N % PP_BN || M < 1) return 0;
  if (epi != TIMHIP_EPI_STORE_T && epi != TIMHIP_EPI_GELU_DROP_G2 && epi != TIMHIP_EPI_MULAUX_T) return 0;
  if (kn == 8 || kn == 10) return kn;
  if (epi == TIMHIP_EPI_MULAUX_T && kn != 2) return 0;
  const long long t160 = (long long)((M + 159) / 160) * (N / PP_BN);
  if (t160 <= 256) return 0;
  auto fill = [&](int bm) {
    const long long tiles = (long long)((M + bm - 1) / bm) * (N / PP_BN), rounds = (tiles + 255) / 256;
    return (double)M * N / ((double)bm * PP_BN) / (double)(rounds * 256);
  };
  const double f8 = fill(256), f10 = fill(320);
  if (f10 >= f8 && f10 >= 0.85) return 10;
  if (f8 >= 0.85) return 8;
  return 0;
}

// ---- residual + LayerNorm fused into the epilogue (gemm_nt_ldln_kernel, round 3) ---------------------------
// out-projection / linear2 of an encoder layer: y = res + dropout(A B^T + bias) AND LayerNorm(y) in ONE launch, for N = the
// LayerNorm width <= 1024 at M a multiple of 160 (one round of tiles: every block co-resident).  A 160 x 256 tile holds a
// quarter of a row, so the tiles_n column tiles of a row panel exchange per-row (sum, sum of squares) through memory:
//   pass 1   the loader-wave kernel's main loop and its dropout + residual epilogue; y goes to memory (fp32: the backward and
//            the next residual read it) AND stays in registers (80 per lane); per-row partial sums over the wave's 64 columns
//            (16-lane shuffles), over the block's 256 columns through LDS
//   publish  part[tile][160] as 8-byte agent-scope atomic stores (write-through), drained (vmcnt) by the storing waves ->
//            __syncthreads -> lane 0: relaxed agent store of flag[tile] = epoch + 1 (MI355X_MICROARCH.md: "8-B agent atomics both
//            sides" - no L2 write-back fence: a release fence here would flush the XCD's 5 MB of freshly written y); y is stored
//            AFTER the publish, so the drain does not wait for it and the partners catch up meanwhile
//   wait     lane 0 polls the partners' flags (relaxed agent loads, s_sleep, BOUNDED) -> __syncthreads -> the partners'
//            partials by agent-scope atomic loads (past the L1)
//   pass 2   mean / rstd per row, the normalised rows from the registers: 16-bit operand copy (+ fp32 rows when asked), the
//            statistics from column tile 0; the FFN keep-bits LayerNorm-1 used to draw are drawn here (before the wait)
// epoch: ctl[0], read by every block at its start and advanced by the LAST block to finish (ctl[1] counts them) - no host
// state, graph-replay safe, flags never need a reset.  A tile whose wait times out (co-residency is not guaranteed: another
// process on the GPU, a CU mask) sets this launch's time-out word ctl[2 + (epoch & 1)] and skips pass 2; the caller launches
// the stand-alone LayerNorm with run_if = ctl behind this kernel: it reads ctl[2 + ((ctl[0] - 1) & 1)] - the word of the launch
// that has just advanced the epoch - and exits at once unless a tile timed out; the last block of a launch clears the
// OTHER word for the launch after it.  Results are right either way.
struct LnFuseDev {
  void* xt; int ldt; float* xf; int ldx; float* stats; const float* g; const float* b;
  uint32_t* mbits; int mwords; uint32_t mthr; TimSeed mseed; uint32_t msite;
  float2* part; uint32_t* flags; uint32_t* ctl; uint32_t spin_limit;
};

template <typename HT, int TMW>
__global__ __launch_bounds__(768) void gemm_nt_ldln_kernel(const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb,
                                                           int M, int N, int K, EpiDev e, LnFuseDev f, int pf_dist) {
  constexpr int BM = 32 * TMW;
  constexpr int A_BYTES = BM * PP_ROWB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / PP_BN, tiles_m = M / BM;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = t / tiles_n, tn = t % tiles_n;
  const int m0 = tm * BM, n0 = tn * PP_BN;
  const int nk = K / 64;
  if (wave >= 8) {
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    pl_load<HT, TMW>(A, lda, B, ldb, M, N, m0, n0, nk, lds0, wave - 8, lane);
    __syncthreads();
    return;
  }
  const uint32_t epoch = __hip_atomic_load(&f.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int wm = wave >> 2, wn = wave & 3;
  const int frow = lane & 15, fk = lane >> 4, sw = (frow >> 1) & 7;
  const int c0 = (fk ^ sw) << 4, c1 = ((fk + 4) ^ sw) << 4;
  const int a_frag = (wm * 16 * TMW + frow) * PP_ROWB;
  const int b_frag = A_BYTES + (wn * 64 + frow) * PP_ROWB;
  f32x4_t acc[PP_TNW][TMW];
#pragma unroll
  for (int i = 0; i < PP_TNW; ++i)
#pragma unroll
    for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  PlPrefetch pf{nullptr, 0u, 0, pf_dist};
  if (pf_dist > 0) {   // the tile's share of its XCD's lines, as in gemm_nt_ld_kernel
    const int conc = max(1, min(8, 32 / tiles_n));
    const int a_cnt = (BM + tiles_n - 1) / tiles_n, a0 = tn * a_cnt;
    const int b_cnt = PP_BN / conc, b0 = (tm % conc) * b_cnt;
    const int li = wave * 64 + lane;
    if (li < a_cnt) {
      const int row = a0 + li;
      pf.base = A; pf.on = row < BM;
      pf.off = (uint32_t)((size_t)min(m0 + row, M - 1) * lda * 2);
    } else if (li - a_cnt < b_cnt) {
      const int row = b0 + li - a_cnt;
      pf.base = B; pf.on = 1;
      pf.off = (uint32_t)((size_t)(n0 + row) * ldb * 2);
    }
  }
  if (wm == 0) pl_consume<HT, TMW, 0>(nk, lds, a_frag, b_frag, c0, c1, acc, pf);
  else pl_consume<HT, TMW, 1>(nk, lds, a_frag, b_frag, c0, c1, acc, pf);
  __syncthreads();   // every wave is done with the stage ring (the loader waves leave here)

  // ---- pass 1: y = res + dropout(acc + bias), to memory and kept; row sums ------------------------------------------------
  constexpr int EP_LD = 68, RBS = 16, NIT = 4;   // 16 x 64 fp32 row block through a wave-private region; 4 quads per lane
  float* ep = reinterpret_cast<float*>(lds) + wave * (RBS * EP_LD);
  float2* rp = reinterpret_cast<float2*>(lds + 40 * 1024);             // [160][4] per-wave-column partial (sum, sumsq)
  float2* rs = reinterpret_cast<float2*>(lds + 48 * 1024);             // [160] (mean, rstd)
  int* sh_ok = reinterpret_cast<int*>(lds + 52 * 1024);
  const float asc = e.acc_scale ? *e.acc_scale : 1.f;
  const int ch = lane & 15, rl = lane >> 4;                            // this lane's quad column / row within a group of 4 rows
  const int wr0 = wm * 16 * TMW, ncol = n0 + wn * 64 + ch * 4;
  const float4 bias4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + ncol) : make_float4(0.f, 0.f, 0.f, 0.f);
  const bool res_ln = e.ln_stats != nullptr;
  float4 lng = make_float4(1.f, 1.f, 1.f, 1.f), lnb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (res_ln) { lng = *reinterpret_cast<const float4*>(e.ln_w + ncol); lnb = *reinterpret_cast<const float4*>(e.ln_b + ncol); }
  float4 yreg[TMW][NIT];
  float4 rbuf[2][NIT];
  float2 sbuf[2][NIT];
  typedef float f4_t __attribute__((ext_vector_type(4)));
  auto fetch_res = [&](int j) {   // the residual rows of row block j -> ring entry j & 1
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + wr0 + j * RBS + it * 4 + rl;
      const f4_t q_ = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(e.res + (size_t)m * e.ldres + ncol));
      rbuf[j & 1][it] = make_float4(q_[0], q_[1], q_[2], q_[3]);
      if (res_ln) sbuf[j & 1][it] = *reinterpret_cast<const float2*>(e.ln_stats + 2 * (size_t)m);
    }
  };
  fetch_res(0);
  if (TMW > 1) fetch_res(1);
  static_for<TMW>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
#pragma unroll
    for (int i = 0; i < PP_TNW; ++i)
      *reinterpret_cast<float4*>(ep + frow * EP_LD + i * 16 + 4 * fk) =
          make_float4(acc[i][j][0] * asc, acc[i][j][1] * asc, acc[i][j][2] * asc, acc[i][j][3] * asc);
    pp_wait_lds();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * 4 + rl, m = m0 + wr0 + j * RBS + row;
      float4 v = *reinterpret_cast<const float4*>(ep + row * EP_LD + ch * 4);
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      float k0 = 1.f, k1 = 1.f, k2 = 1.f, k3 = 1.f;
      if (e.thr != 0u) drop_mask4(e.seed, e.site, ((uint64_t)m * (uint64_t)N + (uint64_t)ncol) >> 2, e.thr, e.scale, k0, k1, k2, k3);
      float4 r = rbuf[j & 1][it];
      if (res_ln) {
        const float2 st = sbuf[j & 1][it];
        r.x = (r.x - st.x) * st.y * lng.x + lnb.x; r.y = (r.y - st.x) * st.y * lng.y + lnb.y;
        r.z = (r.z - st.x) * st.y * lng.z + lnb.z; r.w = (r.w - st.x) * st.y * lng.w + lnb.w;
      }
      const float4 y = make_float4(r.x + v.x * k0, r.y + v.y * k1, r.z + v.z * k2, r.w + v.w * k3);
      yreg[j][it] = y;   // (stored after the partial sums are published: the publishing waves wait for their own stores)
      float s1 = (y.x + y.y) + (y.z + y.w), s2 = (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      if (ch == 0) rp[(wr0 + j * RBS + row) * 4 + wn] = make_float2(s1, s2);
    }
    if constexpr (j + 2 < TMW) fetch_res(j + 2);
    pp_wait_lds();
  });
  __syncthreads();
  const int ctid = tid;   // 0 .. 511: the consumer threads
  // publish: 8-byte agent-scope atomic stores (write-through, no L2 write-back fence needed), drained by the storing waves
  // before the barrier in front of the flag; the partners read them with agent-scope atomic loads (past their L1)
  if (ctid < BM) {
    const float2 a0 = rp[ctid * 4], a1 = rp[ctid * 4 + 1], a2 = rp[ctid * 4 + 2], a3 = rp[ctid * 4 + 3];
    const float2 q = make_float2((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y));
    unsigned long long bits;
    __builtin_memcpy(&bits, &q, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(f.part + (size_t)t * BM + ctid), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ctid == 0) __hip_atomic_store(&f.flags[t], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // y to memory (fp32: the backward and the next residual read it) while the partners catch up
  static_for<TMW>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + wr0 + j * RBS + it * 4 + rl;
      const float4 y = yreg[j][it];
      store4<float>((float*)e.out0 + (size_t)m * e.ld0 + ncol, y.x, y.y, y.z, y.w);
    }
  });
  if (f.mbits) {   // keep-bits of the dropout site the next GEMM's epilogue applies: this tile's share of its panel's words
    const int wpt = f.mwords / tiles_n;
    for (int q = ctid; q < BM * wpt; q += 512) {
      const int row = m0 + q / wpt, wd = tn * wpt + q % wpt;
      f.mbits[(size_t)row * f.mwords + wd] = drop_bits32(f.mseed, f.msite, ((uint64_t)row * f.mwords + wd) * 8, f.mthr);
    }
  }
  if (ctid == 0) {
    int ok = f.spin_limit != 0u;   // (0: every tile gives up at once - the test of the stand-by path)
    for (int p = 0; p < tiles_n && ok; ++p) {
      if (p == tn) continue;
      uint32_t spins = 0;
      while (__hip_atomic_load(&f.flags[tm * tiles_n + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch + 1u) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > f.spin_limit) { ok = 0; break; }
      }
    }
    if (!ok) atomicOr(&f.ctl[2 + (epoch & 1u)], 1u);   // this launch's word (the stand-by launch reads it: run_if convention)
    *sh_ok = ok;
  }
  __syncthreads();
  if (*sh_ok) {
    if (ctid < BM) {
      float s1 = 0.f, s2 = 0.f;
      for (int p = 0; p < tiles_n; ++p) {
        const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(f.part + (size_t)(tm * tiles_n + p) * BM + ctid),
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float2 q;
        __builtin_memcpy(&q, &bits, 8);
        s1 += q.x; s2 += q.y;
      }
      const float mean = s1 / (float)N;
      const float var = fmaxf(s2 / (float)N - mean * mean, 0.f);
      const float rstd = rsqrtf(var + 1e-5f);
      rs[ctid] = make_float2(mean, rstd);
      if (tn == 0 && f.stats) *reinterpret_cast<float2*>(f.stats + 2 * (size_t)(m0 + ctid)) = make_float2(mean, rstd);
    }
    __syncthreads();
    // ---- pass 2: the normalised rows from the registers
    const float4 g4 = *reinterpret_cast<const float4*>(f.g + ncol), b4 = *reinterpret_cast<const float4*>(f.b + ncol);
    static_for<TMW>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int r = wr0 + j * RBS + it * 4 + rl, m = m0 + r;
        const float2 st = rs[r];
        const float4 y = yreg[j][it];
        const float o0 = (y.x - st.x) * st.y * g4.x + b4.x, o1 = (y.y - st.x) * st.y * g4.y + b4.y;
        const float o2 = (y.z - st.x) * st.y * g4.z + b4.z, o3 = (y.w - st.x) * st.y * g4.w + b4.w;
        if (f.xt) store4<HT>((HT*)f.xt + (size_t)m * f.ldt + ncol, o0, o1, o2, o3);
        if (f.xf) store4<float>(f.xf + (size_t)m * f.ldx + ncol, o0, o1, o2, o3);
      }
    });
  }
  // the last block to finish advances the epoch (every block of this launch read the old value at its start)
  if (ctid == 0) {
    const uint32_t done = atomicAdd(&f.ctl[1], 1u);
    if (done == (uint32_t)(tiles_m * tiles_n) - 1u) {
      __hip_atomic_store(&f.ctl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&f.ctl[2 + ((epoch + 1u) & 1u)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch's time-out word
      __hip_atomic_store(&f.ctl[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- dual-group persistent form (gemm_nt_dg_kernel, round 3) -----------------------------------------------------------------
// What the 160 x 256 kernel above cannot do: keep the matrix pipes busy while a tile's epilogue moves its bytes.  Every block of
// a round finishes its main loop at the same moment, then all of them store (and, for the "+ residual" epilogues, read) at
// the HBM rate with the matrix pipes idle - 20-50 % of a launch.  Here the two wave groups of a block
// work on DIFFERENT tiles, a few contraction steps apart:
//   * group tile 160 x 128 (4 waves, 2 x 2 of the same 80 x 64 wave tile as above: same fragment reads, same MFMA phase,
//     same epilogue code), own 2-slot ring of 36-KiB stages (2 x 72 KiB per block); a block owns PAIRS of adjacent column
//     tiles of one row panel - group 0 the even, group 1 the odd one - and walks its pairs in a persistent loop;
//   * both groups run the LOAD / MFMA phase alternation of the kernel above behind the same workgroup barriers, group 1
//     an ODD number of barriers behind: one group's MFMA phase coincides with the other's LOAD phase as before, but
//     group 1 reaches the end of its tile DG_OFFSET / 2 steps after group 0 - while a group runs its epilogue (five 16-row
//     blocks, one barrier each, to keep the cadence) the other one is still in its main loop, and its next tile's first
//     stage is already in flight;
//   * a 2-slot ring suffices because a LOAD phase reads a whole stage into registers: slot t % 2 is free from the
//     barrier that ends LOAD phase t, stage t + 2 is issued into it during MFMA phase t and waited for at the end of MFMA
//     phase t + 1 (one full step of flight time).  The epilogue transposes through the group's own slot 1 while the next
//     tile's stage 0 lands in slot 0.
// Hazards: as above per group; the barriers are workgroup-wide, i.e. stronger than a group needs.  A group that has
// finished its tiles simply ends; the hardware barrier only counts live waves.
// vmcnt: the DMA pieces are invisible to the compiler's own s_waitcnt bookkeeping; hidden older operations can only make
// its waits for the epilogue's loads longer, never shorter, and the kernel's own counted waits (vmcnt(9): everything but the
// newest stage) cover whatever the epilogue still has outstanding.
constexpr int DG_BM = 160, DG_BN = 128, DG_CNT = 9, DG_TMW = 5;
constexpr int DG_A_PIECES = DG_BM / 8, DG_A_BYTES = DG_BM * PP_ROWB, DG_ST_BYTES = (DG_BM + DG_BN) * PP_ROWB, DG_RING = 2 * DG_ST_BYTES;

template <typename HT, int EPI>
__global__ __launch_bounds__(512) void gemm_nt_dg_kernel(const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb,
                                                         int M, int N, int K, EpiDev e, int ppb, int offset_phases) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [group][2 slots][A tile 160 x 128 B | B tile 128 x 128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w = wave & 3, wm = w >> 1, wn = w & 1;
  const char* ring = lds + grp * DG_RING;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring));
  const int pairs_n = N / (2 * DG_BN);
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int nk = K / 64;

  // this wave's 9 pieces of a stage: tile-independent offsets (full tiles only: no clamping)
  const int lrow = lane >> 3, lchunk = lane & 7;
  const int first_piece = DG_CNT * w;
  uint32_t off[DG_CNT];
#pragma unroll
  for (int i = 0; i < DG_CNT; ++i) {
    const int p = first_piece + i;
    const bool is_a = p < DG_A_PIECES;
    const int row = (is_a ? p : p - DG_A_PIECES) * 8 + lrow;
    const int c = (lchunk ^ kswz<64>(row)) * 8;
    off[i] = (uint32_t)(((size_t)row * (is_a ? lda : ldb) + c) * 2);
  }
  const int frow = lane & 15, fk = lane >> 4, sw = (frow >> 1) & 7;
  const int c0 = (fk ^ sw) << 4, c1 = ((fk + 4) ^ sw) << 4;
  const int a_frag = (wm * 16 * DG_TMW + frow) * PP_ROWB;
  const int b_frag = DG_A_BYTES + (wn * 64 + frow) * PP_ROWB;

  const char* Ab = nullptr; const char* Bb = nullptr;
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int pair) {
    m0 = (pair / pairs_n) * DG_BM;
    n0 = ((pair % pairs_n) * 2 + grp) * DG_BN;
    Ab = reinterpret_cast<const char*>(A + (size_t)m0 * lda);
    Bb = reinterpret_cast<const char*>(B + (size_t)n0 * ldb);
  };
  auto piece = [&](int kt, int slot, int i) {
    const int p = first_piece + i;
    const char* g = (p < DG_A_PIECES ? Ab : Bb) + (size_t)kt * PP_ROWB;
    glds16_s(uniform_ptr(g), off[i], lds0 + slot * DG_ST_BYTES + p * 1024);
  };
  auto issue_stage = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < DG_CNT; ++i) piece(kt, slot, i);
  };

  if (grp == 1)
    for (int i = 0; i < offset_phases; ++i) pp_barrier();

  int pair = lb * ppb;
  set_tile(pair);
  issue_stage(0, 0);
  issue_stage(1, 1);

  f32x4_t acc[PP_TNW][DG_TMW];
  vec8<HT> xa[DG_TMW], wb[PP_TNW], xb[DG_TMW], wc[PP_TNW];
  for (int pi = 0; pi < ppb; ++pi) {
#pragma unroll
    for (int i = 0; i < PP_TNW; ++i)
#pragma unroll
      for (int j = 0; j < DG_TMW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    glds_wait<DG_CNT>();   // stage 0 (and whatever the previous epilogue still had in flight) has landed; stage 1 may fly
    pp_barrier();
    auto step = [&](int t, auto more_c) {
      constexpr bool MORE = decltype(more_c)::value;
      const int slot = t & 1;
      const char* base = ring + slot * DG_ST_BYTES;
#pragma unroll
      for (int j = 0; j < DG_TMW; ++j) xa[j] = *reinterpret_cast<const vec8<HT>*>(base + a_frag + j * (16 * PP_ROWB) + c0);
#pragma unroll
      for (int i = 0; i < PP_TNW; ++i) wb[i] = *reinterpret_cast<const vec8<HT>*>(base + b_frag + i * (16 * PP_ROWB) + c0);
#pragma unroll
      for (int j = 0; j < DG_TMW; ++j) xb[j] = *reinterpret_cast<const vec8<HT>*>(base + a_frag + j * (16 * PP_ROWB) + c1);
#pragma unroll
      for (int i = 0; i < PP_TNW; ++i) wc[i] = *reinterpret_cast<const vec8<HT>*>(base + b_frag + i * (16 * PP_ROWB) + c1);
      pp_wait_lds();
      pp_barrier();   // every wave of the group has read slot t % 2: it may be refilled
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int j = 0; j < DG_TMW; ++j)
#pragma unroll
          for (int i = 0; i < PP_TNW; ++i) {
            acc[i][j] = half == 0 ? mfma16x16<HT>(wb[i], xa[j], acc[i][j]) : mfma16x16<HT>(wc[i], xb[j], acc[i][j]);
            const int q = half * (DG_TMW * PP_TNW) + j * PP_TNW + i, k = q / 4;
            if (MORE && q % 4 == 1 && k < DG_CNT) {
              __builtin_amdgcn_sched_barrier(0);
              piece(t + 2, slot, k);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      if (MORE) glds_wait<DG_CNT>(); else glds_wait<0>();   // stage t + 1 (issued one step ago) has landed
      pp_barrier();
    };
    int t = 0;
    for (; t + 2 < nk; ++t) step(t, std::true_type{});
    for (; t < nk; ++t) step(t, std::false_type{});

    // epilogue of this tile; the next tile's stage 0 flies meanwhile (slot 0), slot 1 is the transposition space
    const int em0 = m0 + wm * 16 * DG_TMW, en0 = n0 + wn * 64;
    const bool next = pi + 1 < ppb;
    if (next) {
      set_tile(pair + pi + 1);
      issue_stage(0, 0);
    }
    float* ep = reinterpret_cast<float*>(const_cast<char*>(ring) + DG_ST_BYTES) + w * (16 * 68);
    pp_epilogue<HT, EPI, DG_TMW>(e, acc, em0, en0, M, N, ep, lane, [] { pp_barrier(); });
    if (next) issue_stage(1, 1);
  }
}

#ifdef TIMHIP_TUNING   // (measured 13 % slower than one tile per block, DESIGN.md section 5c: tuning builds only)
template <typename HT, int EPI>
void launch_dg(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const EpiDev& e, int offset_phases, hipStream_t s) {
  const size_t shmem = 2 * (size_t)DG_RING;
  static PerDeviceOnce attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute((const void*)gemm_nt_dg_kernel<HT, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  const int npairs = (M / DG_BM) * (N / (2 * DG_BN));
  const int ppb = (npairs + 255) / 256;         // pairs per block; tim_gemm_dg_ok() guarantees ppb divides npairs
  hipLaunchKernelGGL((gemm_nt_dg_kernel<HT, EPI>), dim3(npairs / ppb), dim3(512), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N,
                     K, e, ppb, offset_phases);
}

#endif

template <typename HT, int EPI, int TMW = 5>
void launch_pp(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const EpiDev& e, hipStream_t s) {
  constexpr int BM = 32 * TMW;
  const size_t shmem = (size_t)PP_NST * (BM + PP_BN) * PP_ROWB;
  if constexpr (TMW != 5) {   // the 128-row tile exists for the loader-wave kernels only
    if (tim_knobs().gemm_ld == 0 || e.a_wrap != 0) return launch_pp<HT, EPI, 5>(A, lda, B, ldb, M, N, K, e, s);
  }
#ifdef TIMHIP_TUNING   // per-phase cycle counters (tools/pp_phase.py)
  if constexpr (EPI == TIMHIP_EPI_STORE_T && TMW == 5) {
    if (getenv("TIMHIP_PP_PROF")) {
      const char* ab = getenv("TIMHIP_PP_ABL");   // 1 no DMA, 2 no MFMAs, 3 DMA only, L the late wait (two-barrier loop)
      const dim3 g_(((M + BM - 1) / BM) * ((N + PP_BN - 1) / PP_BN));
#define PP_ABL_LAUNCH(ABLV, LATEV) do { \
        (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<HT, EPI, TMW, true, ABLV, LATEV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        hipLaunchKernelGGL((gemm_nt_pp_kernel<HT, EPI, TMW, true, ABLV, LATEV>), g_, dim3(512), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N, K, e); } while (0)
      if (ab && ab[0] == '1') { PP_ABL_LAUNCH(1, false); return; }
      if (ab && ab[0] == '2') { PP_ABL_LAUNCH(2, false); return; }
      if (ab && ab[0] == '3') { PP_ABL_LAUNCH(3, false); return; }
      if (ab && ab[0] == '4') { PP_ABL_LAUNCH(4, false); return; }
      if (ab && ab[0] == '5') { PP_ABL_LAUNCH(5, false); return; }
      if (ab && ab[0] == '7') { PP_ABL_LAUNCH(7, false); return; }
      if (ab && ab[0] == '8') { PP_ABL_LAUNCH(8, false); return; }
      if (ab && ab[0] == '6') { PP_ABL_LAUNCH(6, false); return; }
      if (ab && ab[0] == 'L') { PP_ABL_LAUNCH(0, true); return; }
#undef PP_ABL_LAUNCH
      (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<HT, EPI, TMW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      hipLaunchKernelGGL((gemm_nt_pp_kernel<HT, EPI, TMW, true>), g_, dim3(512), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N, K, e);
      return;
    }
  }
#endif
  static PerDeviceOnce attr_set;   // idempotent; a benign race sets it twice at worst
  if (attr_set.first()) {
    if constexpr (TMW == 5)
      (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<HT, EPI, TMW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void*)gemm_nt_ld_kernel<HT, EPI, TMW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  }
  const TimKnobs& kn = tim_knobs();
  const dim3 grid(((M + BM - 1) / BM) * ((N + PP_BN - 1) / PP_BN));
#ifdef TIMHIP_TUNING
  if constexpr (EPI != TIMHIP_EPI_DROP_RES_F32 && TMW == 5) {   // (that epilogue's residual prefetch leaves no registers for the tile loop's state)
    // two or three full rounds of tiles: the 8-wave persistent-tile kernel - TIMHIP_GEMM_PT=1, tuning builds only (measured equal
    // to one tile per block within the run-to-run spread, in_proj forward 71.8 vs 71.5 us, linear1 61.8 vs 60.3, linear2 input
    // gradient 50.7 vs 50.5 - profiles/r03_nt_kernel_variants_ab.txt)
    const int tiles = (int)grid.x, tpb = (tiles + 255) / 256;
    if (tpb >= 2 && tpb <= 4 && tiles % tpb == 0 && ((N + PP_BN - 1) / PP_BN) % tpb == 0 && K >= 128 && kn.gemm_pt == 1 && e.a_wrap == 0) {
      static PerDeviceOnce pt_attr;
      if (pt_attr.first())
        (void)hipFuncSetAttribute((const void*)gemm_nt_pt_kernel<HT, EPI, TMW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      hipLaunchKernelGGL((gemm_nt_pt_kernel<HT, EPI, TMW>), dim3(tiles / tpb), dim3(512), shmem, s, (const HT*)A, lda, (const HT*)B, ldb,
                         M, N, K, e, tpb);
      return;
    }
  }
#endif
  // Default: the loader-wave kernel with the L2 prefetch (round 3: in the step 5.66 ms with the 8-wave kernel, 5.60-5.65 with
  // loader waves, 5.41-5.44 with loader waves + prefetch 4 stages ahead, NT GEMMs 763 -> 816 TFLOP/s; prefetch distance 2 / 3 / 5
  // / 6 / 8: +1.0 / +0.5 / +0.2 / +0.7 / +0.8 % of the step; every tile prefetching ALL its lines: 5.83-5.89 ms, the
  // prefetch loads then take the vector-memory path's time themselves).  TIMHIP_GEMM_LD=0: the 8-wave kernel (fallback).
  if (kn.gemm_ld != 0 && e.a_wrap == 0) {
    // multi-round shapes (more than 256 tiles; TIMHIP_GEMM_PF_MR): no prefetch - their tiles drift apart after the first round and
    // a share covers a twelfth of a panel; in the step distance 0 / 2 / 4 / 6 / 8 / 12 for them: 5.29 / 5.32 / 5.32 / 5.34 / 5.34 / 5.34 ms
    // Two to four full rounds of tiles run as ONE round of blocks that walk tpb column tiles each (gemm_nt_ldp_kernel;
    // TIMHIP_GEMM_LDP=0: one tile per block) - the tiles of an XCD stay in step, so the prefetch shares apply to them.  In the
    // step 5.343 -> 5.314 ms (four interleaved runs per arm, every run of the walk below every run of the one-tile kernel).
    // The history of that number is in DESIGN.md section 5d: a first A/B had said -2.6 % against a one-tile instance that
    // spilled; against the restored spill-free one-tile kernel the walk first LOST 1.3 % - it spilled 9-21 registers itself,
    // the epilogue's loop-invariant lane arithmetic having been hoisted in front of the tile loop and held across the main
    // loop; with the lane id laundered once per tile (8 / 5 / 1 spills on the three epilogues that use it) it wins 0.5 %.
    const int tiles_ = (int)grid.x, tiles_n_ = (N + PP_BN - 1) / PP_BN;
    int tpb = 1;
    if (kn.gemm_ldp != 0 && tiles_ > 256) {
      const int want = (tiles_ + 255) / 256;
      if (want <= 4 && tiles_n_ % want == 0 && tiles_ % want == 0 && K >= 128) tpb = want;
    }
    const int pf_d_ = (grid.x > 256 && tpb == 1) ? kn.gemm_pf_mr : kn.gemm_pf;
    const dim3 grid_(tiles_ / tpb);
#ifdef TIMHIP_TUNING
    if constexpr (TMW == 5) if (kn.gemm_ld1 == 1) {   // one barrier per contraction step (measured: +4.6 % isolated, +0.5 % in the step)
      static PerDeviceOnce attr_l1;
      if (attr_l1.first())
        (void)hipFuncSetAttribute((const void*)gemm_nt_ld_kernel<HT, EPI, TMW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      hipLaunchKernelGGL((gemm_nt_ld_kernel<HT, EPI, TMW, true>), grid, dim3(768), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N, K, e,
                         grid.x > 256 ? kn.gemm_pf_mr : kn.gemm_pf, kn.gemm_pf_mode);
      return;
    }
#endif
    if (tpb > 1) {
      static PerDeviceOnce attr_p;
      if (attr_p.first())
        (void)hipFuncSetAttribute((const void*)gemm_nt_ldp_kernel<HT, EPI, TMW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      hipLaunchKernelGGL((gemm_nt_ldp_kernel<HT, EPI, TMW>), grid_, dim3(768), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N,
                         K, e, pf_d_, kn.gemm_pf_mode, tpb);
      return;
    }
    hipLaunchKernelGGL((gemm_nt_ld_kernel<HT, EPI, TMW>), grid, dim3(768), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N, K, e,
                       pf_d_, kn.gemm_pf_mode);
    return;
  }
  if constexpr (TMW == 5)
    hipLaunchKernelGGL((gemm_nt_pp_kernel<HT, EPI, TMW>), grid, dim3(512), shmem, s, (const HT*)A, lda, (const HT*)B, ldb, M, N, K, e);
}

// Which row-panel height: 160 (the default: 98 FLOP per staged byte) or 128 (85)?  What a launch costs is (rounds of co-resident
// blocks) x (tiles a block walks) x (bytes a tile stages per step); M = 7984 (detection, B = 16 x 499 tokens) and M = 8000
// (Perception Test) give 50 panels of 160 rows - 200 tiles per 1024 columns on 256 CUs - but 63 panels of 128: 252.
static int pp_cost(int M, int N, int K, int bm) {
  const int tiles_n = (N + PP_BN - 1) / PP_BN, tiles = ((M + bm - 1) / bm) * tiles_n;
  int tpb = 1;
  if (tiles > 256) {
    const int want = (tiles + 255) / 256;
    if (want <= 4 && tiles_n % want == 0 && tiles % want == 0 && K >= 128) tpb = want;
  }
  const int blocks = tiles / tpb, rounds = (blocks + 255) / 256;
  return rounds * tpb * (bm + PP_BN);
}
static int pp_tmw(int M, int N, int K) {
  const int force = tim_knobs().gemm_tmw;   // (A/B knob)
  if (force == 4 || force == 5) return force;
  return pp_cost(M, N, K, 128) * 100 < pp_cost(M, N, K, 160) * 97 ? 4 : 5;
}

}  // namespace

// residual + LayerNorm fused into the GEMM (gemm_nt_ldln_kernel).  TIMHIP_EUNSUPPORTED: the caller runs the two kernels.
// `fail` (out): the device word the stand-alone LayerNorm behind this launch takes as run_if (see the kernel's header).
int tim_gemm_nt_pp_ln(int precision, const void* A, int lda, const void* B, int ldb, int M, int N, int K, const void* epi_dev,
                      const TimLnFuse& lf, const uint32_t** fail, hipStream_t s) {
#ifndef TIMHIP_TUNING
  // Product builds do not carry the fused kernel: measured 64.2 us (fused GEMM 59.5 + the stand-by LayerNorm launch 4.7) against
  // 62.5 us for the two kernels, it stays a tuning-build experiment; the layer runs its two-kernel path.
  (void)precision; (void)A; (void)lda; (void)B; (void)ldb; (void)M; (void)N; (void)K; (void)epi_dev; (void)lf; (void)fail; (void)s;
  return TIMHIP_EUNSUPPORTED;
#else
  const EpiDev& e = *reinterpret_cast<const EpiDev*>(epi_dev);
  constexpr int TMW = 5, BM = 32 * TMW;
  // (N: the widths the stand-by LayerNorm behind this launch supports - the same predicate, so the pair is all-or-nothing)
  if (N != 512 && N != 1024 && N != 2048) return TIMHIP_EUNSUPPORTED;
  if (!h16_storage(precision) || M % BM || N % PP_BN || K % 64 || K < 128 || !e.vec || e.a_wrap != 0 || !e.res || !lf.xt || !lf.stats ||
      !lf.g || !lf.b)
    return TIMHIP_EUNSUPPORTED;
  const int tiles = (M / BM) * (N / PP_BN);
  if (tiles > 256 || tiles < 128 || N / PP_BN > 8 || (lf.ldt % 4) || (lf.xf && lf.ldx % 4)) return TIMHIP_EUNSUPPORTED;   // one co-resident round
  if (lf.mask_out && (lf.mask_cols % (32 * (N / PP_BN)) || ((uintptr_t)lf.mask_out & 3))) return TIMHIP_EUNSUPPORTED;
  // per-device scratch: partial sums, flags, {epoch, done, fail}
  // (one scratch block per device, allocated under a lock at first use - which must not be inside a graph capture; concurrent
  //  fused launches on one device from several streams would share it: the tuning experiment runs one stream)
  struct Scratch { float2* part; uint32_t* flags; uint32_t* ctl; };
  static Scratch scr[32] = {};
  static std::mutex scr_lock;
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = dev < 0 ? 0 : (dev > 31 ? 31 : dev);
  std::lock_guard<std::mutex> hold(scr_lock);
  if (!scr[dev].part) {
    char* p = nullptr;
    const size_t bytes = 256 * (size_t)BM * sizeof(float2) + 256 * 4 + 64;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) return TIMHIP_EUNSUPPORTED;
    scr[dev].part = (float2*)p; scr[dev].flags = (uint32_t*)(p + 256 * (size_t)BM * sizeof(float2)); scr[dev].ctl = scr[dev].flags + 256;
  }
  LnFuseDev f;
  f.xt = lf.xt; f.ldt = lf.ldt; f.xf = lf.xf; f.ldx = lf.ldx; f.stats = lf.stats; f.g = lf.g; f.b = lf.b;
  f.mbits = (lf.mask_out && lf.mask_p > 0.f) ? (uint32_t*)lf.mask_out : nullptr;
  f.mwords = lf.mask_cols / 32; f.mthr = lf.mask_p > 0.f ? drop_threshold(lf.mask_p) : 0u; f.mseed = TimSeed(lf.mask_seed); f.msite = lf.mask_site;
  f.part = scr[dev].part; f.flags = scr[dev].flags; f.ctl = scr[dev].ctl;
  f.spin_limit = (uint32_t)tim_knobs().fuse_ln_spin;   // polls (~0.3 us each) before a tile gives up: default ~30 ms
  const size_t shmem = (size_t)PP_NST * (BM + PP_BN) * PP_ROWB;
  static PerDeviceOnce attr_set[2];
  const int hi = precision == TIMHIP_PREC_F16 ? 1 : 0;
  if (attr_set[hi].first())
    DISPATCH_H16(precision, (void)hipFuncSetAttribute((const void*)gemm_nt_ldln_kernel<HT, TMW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  DISPATCH_H16(precision, hipLaunchKernelGGL((gemm_nt_ldln_kernel<HT, TMW>), dim3(tiles), dim3(768), shmem, s, (const HT*)A, lda, (const HT*)B,
                                            ldb, M, N, K, e, f, tim_knobs().gemm_pf));
  if (fail) *fail = scr[dev].ctl;   // (run_if convention of tim_layernorm_fwd: see the kernel's header)
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
#endif
}

// Is this problem one for the ping-pong kernel?  It needs enough 160 x 256 tiles to fill the 256 CUs about evenly: the
// encoder-layer GEMMs of a production batch (M = B * S in the thousands, N a multiple of 256 from 1024 up).
bool tim_gemm_pp_wins(int M, int N, int K, int splitk) {
  if (splitk != 1 || N < 512 || M < 1280) return false;
  const long long tiles = (long long)((M + 159) / 160) * ((N + PP_BN - 1) / PP_BN);
  const long long rounds = (tiles + 255) / 256;
  if (tiles < 256) return tiles >= tim_knobs().gemm_pp_min;   // (TIMHIP_GEMM_PP_MIN_TILES: half-batch chains run 124-tile launches side by side)
  return tiles * 100 >= rounds * 256 * 75;   // the last round at least three quarters full on average
}

// Shapes for the dual-group persistent kernel: whole 160 x 256 tile pairs, at least two contraction steps, an even spread of the
// pairs over the blocks, vectorised epilogue operands.
#ifdef TIMHIP_TUNING
static bool tim_gemm_dg_ok(int M, int N, int K, const EpiDev& e) {
  if (M % DG_BM || N % (2 * DG_BN) || K % 64 || K < 128 || !e.vec || !e.vec8) return false;
  const int npairs = (M / DG_BM) * (N / (2 * DG_BN));
  const int ppb = (npairs + 255) / 256;
  return npairs >= 192 && npairs % ppb == 0;
}

#endif

// (tests / tools: which row tile of the eight-phase kernel a plain launch of this shape and epilogue would take - 8, 10 or 0)
int tim_gemm_p8_choice(int epi, int M, int N, int K) {
  EpiDev e{};
  return p8_choice(epi, M, N, K, e);
}

int tim_gemm_nt_pp(int precision, int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K, const void* epi_dev,
                   hipStream_t s) {
  const EpiDev& e = *reinterpret_cast<const EpiDev*>(epi_dev);
  if (!h16_storage(precision)) return TIMHIP_EUNSUPPORTED;
  const bool tall = pp_tmw(M, N, K) == 5;
#ifdef TIMHIP_TUNING
  // TIMHIP_GEMM_DG=1 (tuning builds): the dual-group persistent kernel where the shape allows - measured 13 % slower over the
  // layer's eight shapes than the one-tile-per-block kernel: its 160 x 128 group tiles stage 72 KiB per step pair where the
  // 160 x 256 tile stages 52, and the global -> LDS path (~30 B/clk/CU), not the matrix pipe, is what bounds these loops; the hidden
  // epilogues do not pay for that - DESIGN.md section 5c; TIMHIP_GEMM_DG_OFFSET: barriers group 1 runs behind group 0 (odd)
  if (tim_knobs().gemm_dg == 1 && tim_gemm_dg_ok(M, N, K, e) && e.a_wrap == 0) {
    int off = tim_knobs().gemm_dg_offset;
    if (off < 1) off = 1;
    off |= 1;
    switch (epi) {
#define CASE(X) case X: DISPATCH_H16(precision, (launch_dg<HT, X>(A, lda, B, ldb, M, N, K, e, off, s))); break;
      CASE(TIMHIP_EPI_STORE_T)
      CASE(TIMHIP_EPI_DROP_RES_F32)
      CASE(TIMHIP_EPI_ADD_F32)
      CASE(TIMHIP_EPI_GELU_DROP_G2)
      CASE(TIMHIP_EPI_MULAUX_T)
#undef CASE
      default: goto plain;
    }
    return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
  }
plain:
#endif
  if (const int tm8 = p8_choice(epi, M, N, K, e)) {
    switch (epi) {
#define CASE(X) case X: DISPATCH_H16(precision, (tm8 == 8 ? launch_p8<HT, X, 8>(A, lda, B, ldb, M, N, K, e, s) : launch_p8<HT, X, 10>(A, lda, B, ldb, M, N, K, e, s))); break;
      CASE(TIMHIP_EPI_STORE_T)
      CASE(TIMHIP_EPI_GELU_DROP_G2)
      CASE(TIMHIP_EPI_MULAUX_T)
#undef CASE
      default: return TIMHIP_EUNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
  }
  switch (epi) {
#define CASE(X) case X: DISPATCH_H16(precision, (tall ? launch_pp<HT, X, 5>(A, lda, B, ldb, M, N, K, e, s) : launch_pp<HT, X, 4>(A, lda, B, ldb, M, N, K, e, s))); break;
    CASE(TIMHIP_EPI_STORE_T)
    CASE(TIMHIP_EPI_RELU_T)
    CASE(TIMHIP_EPI_STORE_F32)
    CASE(TIMHIP_EPI_DROP_RES_F32)
    CASE(TIMHIP_EPI_ADD_F32)
    CASE(TIMHIP_EPI_GELU_DROP_G2)
    CASE(TIMHIP_EPI_MULAUX_T)
#undef CASE
    default: return TIMHIP_EUNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}
