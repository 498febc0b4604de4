// GEMM  C[M,N] = A[M,K] * B[N,K]^T  with fused epilogues (gfx950 MFMA).
//
// Both operands are K-contiguous ("NT"), which is the natural form of every
// product on the TIM path once the weights are kept in two working copies
// (W and W^T, timhip_cast_weight): forward  y = x W^T, dgrad  dx = dy (W^T)^T,
// wgrad  dW = (dy^T)(x^T)^T.  Replaces nn.Linear / F._in_projection_packed at
// transformers.py:102,107; encodings.py:21-26,140-153; tim.py:66-74; head.py:8-15.
//
// Orientation: the MFMA is issued as D = W_frag x X_frag so that D[i = n][j = m]:
// a lane owns ONE output row m (= lane & 31) and, per accumulator quad, FOUR
// consecutive output columns n.  Row-wise epilogues (bias, residual, GELU,
// Philox dropout keyed on the element index) then work on 4-wide vectors and
// store 8 B (bf16) / 16 B (fp32) per lane.
//
// bf16 kernel: 128x128x64 tile, 4 waves (2x2), each 64x64 = 2x2 MFMA 32x32x16,
// operands staged HBM -> LDS with global_load_lds (16 B/lane), two LDS stages,
// XOR swizzle applied on the SOURCE address (LDS image stays lane-linear) and on
// the ds_read_b128 side so the fragment reads are bank-conflict free.
#include <stdlib.h>

#include "gemm_epi.h"

namespace {

// ---------------------------------------------------------------------------
// 16-bit MFMA kernel (HT = bf16_t or f16_t)
//   BM x BN output tile, WM x WN waves, BKT (32|64) contraction elements per stage, NST-deep LDS ring.
//   Stages are filled by LDS-DMA issued from inline asm and retired with COUNTED vmcnt waits, so up to
//   NST-1 stages stay in flight across the per-step barrier (latency of an L2 miss >> one step).
//   GM > 1: "grouped" tile order - consecutive blocks walk GM row-panels of one column-panel before
//   moving to the next column-panel, so the set of co-resident blocks shares few A and B panels (L2).
// ---------------------------------------------------------------------------
// the body of one block: tile `t` (already mapped to a logical tile index) of split `zsplit`
template <typename HT, int EPI, int BM, int BN, int WM, int WN, int BKT, int NST, int GM, int ABL = 0>
__device__ __forceinline__ void gemm_nt_h16_body(
    const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb, int M, int N,
    int K, int ksteps_per_split, EpiDev e, int t, int zsplit) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int ROWB = BKT * 2;                 // bytes per tile row
  constexpr int NCH = ROWB / 16;                // 16-B chunks per row
  constexpr int RPI = 1024 / ROWB;              // rows per 1-KiB wave instruction
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / RPI / NW, B_INSTR = BN / RPI / NW, LPS = A_INSTR + B_INSTR;
  static_assert(A_INSTR >= 1 && B_INSTR >= 1, "tile too small for the wave count");
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [NST][A_BYTES + B_BYTES]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int pm, pn;
  if (GM > 1) {
    const int per_group = GM * tiles_n;
    const int grp = t / per_group, first = grp * GM;
    const int gsz = min(GM, tiles_m - first);
    const int r = t - grp * per_group;
    pm = first + r % gsz;
    pn = r / gsz;
  } else {
    pm = t / tiles_n;
    pn = t % tiles_n;
  }
  const int m0 = pm * BM, n0 = pn * BN;

  const int nk_total = K / BKT;
  const int spl = ksteps_per_split * (BK / BKT);
  const int kt0 = zsplit * spl;
  const int kt1 = min(nk_total, kt0 + spl);
  if (EPI == TIMHIP_EPI_STORE_F32) e.out0 = (float*)e.out0 + (long long)zsplit * e.slab_stride;

  // ---- staging: each wave-instruction moves RPI rows x ROWB bytes ----
  const int lrow = lane / NCH, lchunk = lane % NCH;
  uint32_t a_off32[A_INSTR], b_off32[B_INSTR];  // per-lane byte offsets; the K advance is a scalar add on the base
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (wave * A_INSTR + i) * RPI + lrow;
    const int c = lchunk ^ kswz<BKT>(row);
    a_off32[i] = (uint32_t)(((size_t)min(m0 + row, M - 1) * lda + c * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int row = (wave * B_INSTR + i) * RPI + lrow;
    const int c = lchunk ^ kswz<BKT>(row);
    b_off32[i] = (uint32_t)(((size_t)min(n0 + row, N - 1) * ldb + c * 8) * 2);
  }
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  const int a_wrap = e.a_wrap > 0 ? e.a_wrap * (64 / BKT) : 0x7fffffff;   // K-steps after which the A operand repeats
  auto stage = [&](int kt, int buf) {
    const uint32_t base = lds0 + buf * ST_BYTES;
    const int kta = kt >= a_wrap ? kt - a_wrap : kt;   // (the operand repeats once: K = 2 * a_wrap_k)
    glds16_xn<A_INSTR>(reinterpret_cast<const char*>(A) + (size_t)kta * BKT * 2, a_off32, base + wave * A_INSTR * 1024);
    glds16_xn<B_INSTR>(reinterpret_cast<const char*>(B) + (size_t)kt * BKT * 2, b_off32,
                       base + A_BYTES + wave * B_INSTR * 1024);
  };

  f32x16_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // precomputed dropout keep-bits (e.mask): all of this lane's bytes are fetched here, ahead of the main loop - a load issued
  // between the epilogue's stores could not be moved ahead of them (possible aliasing) and would cost a memory latency per chunk
  constexpr int MB_COLS = TN * 32, NIT8 = 32 * (MB_COLS / 8) / 64;
  uint32_t mbyte[TM][NIT8 > 0 ? NIT8 : 1];
  if (epi_uses_dropout(EPI) && epi_has_oct(EPI) && e.vec8 && e.mask && e.thr != 0u) {
    static_for<TM>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int it = 0; it < NIT8; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / (MB_COLS / 8), ch = idx % (MB_COLS / 8);
        const int m = m0 + wm * (BM / WM) + j * 32 + row;
        const int n = n0 + wn * (BN / WN) + ch * 8;
        mbyte[j][it] = (m < M && n + 7 < N) ? (uint32_t)e.mask[(size_t)m * e.ldmask + (n >> 3)] : 0u;
      }
    });
  }

  // fragment addresses: row = lane & 31, 16-B chunk = kk*2 + (lane >> 5), swizzled
  const int frow = lane & 31, fhalf = lane >> 5;
  int a_off[TM], b_off[TN], a_swz[TM], b_swz[TN];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int row = wm * (BM / WM) + j * 32 + frow;
    a_off[j] = row * ROWB;
    a_swz[j] = kswz<BKT>(row);
  }
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int row = wn * (BN / WN) + i * 32 + frow;
    b_off[i] = A_BYTES + row * ROWB;
    b_swz[i] = kswz<BKT>(row);
  }

  // prologue: NST-1 stages in flight
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (kt0 + p < kt1) stage(kt0 + p, p);
  int buf = 0;
  vec8<HT> xa[2][TM], wb[2][TN];
  // ABL & 16 (tuning builds): per-phase shader-cycle counters of one wave per block, summed into e.aux[0..7]
  constexpr bool PROF = (ABL & 16) != 0;
  long long t_wait = 0, t_issue = 0, t_mma = 0, t_begin = 0, t0 = 0, t1 = 0;
  long long w_begin = 0;
  if constexpr (PROF) { t_begin = __builtin_readcyclecounter(); w_begin = wall_clock64(); }
  for (int kt = kt0; kt < kt1; ++kt) {
    if constexpr (PROF) t0 = __builtin_readcyclecounter();
    // stage kt must have landed; the (up to NST-2) younger stages may stay in flight
    const int younger = min(NST - 2, kt1 - 1 - kt);
    if (NST >= 5 && younger >= 3) glds_wait<3 * LPS>();
    else if (NST >= 4 && younger >= 2) glds_wait<2 * LPS>();
    else if (NST >= 3 && younger >= 1) glds_wait<LPS>();
    else glds_wait<0>();
    if ((ABL & 7) < 4) __syncthreads();  // everybody's part of stage kt is in LDS; everybody finished reading stage kt-1
    if constexpr (PROF) { t1 = __builtin_readcyclecounter(); t_wait += t1 - t0; }
    if ((ABL & 1) == 0 && kt + NST - 1 < kt1) stage(kt + NST - 1, buf == 0 ? NST - 1 : buf - 1);
    if constexpr (PROF) { t0 = __builtin_readcyclecounter(); t_issue += t0 - t1; }
    const char* base = lds + ((ABL & 1) ? 0 : buf) * ST_BYTES;
    // fragments are double-buffered in registers: the ds_reads of step kk+1 are in flight while
    // the MFMAs of step kk run
    auto load_frags = [&](int kk, int set) {
      const int c = kk * 2 + fhalf;
#pragma unroll
      for (int j = 0; j < TM; ++j)
        xa[set][j] = *reinterpret_cast<const vec8<HT>*>(base + a_off[j] + ((c ^ a_swz[j]) << 4));
#pragma unroll
      for (int i = 0; i < TN; ++i)
        wb[set][i] = *reinterpret_cast<const vec8<HT>*>(base + b_off[i] + ((c ^ b_swz[i]) << 4));
    };
    if ((ABL & 2) == 0 || kt == kt0) load_frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < BKT / 16; ++kk) {
      if (((ABL & 2) == 0 || kt == kt0) && kk + 1 < BKT / 16) load_frags(kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);  // the next step's ds_reads are issued BEFORE this step's MFMAs
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = mfma16<HT>(wb[kk & 1][i], xa[kk & 1][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
    if constexpr (PROF) t_mma += __builtin_readcyclecounter() - t0;
  }
  long long t_loop_end = 0;
  if constexpr (PROF) t_loop_end = __builtin_readcyclecounter();

  // ---- epilogue: D[i = n][j = m]; lane owns row m = lane & 31 ----
  const float asc = e.acc_scale ? *e.acc_scale : 1.f;
  if constexpr ((ABL & 8) != 0) {  // ablation: keep the accumulators alive, store (practically) nothing
    if (e.ld0 != -12345) return;
  }
  // The accumulators (lane = row, 4 columns per quad) are transposed through a wave-private LDS region so
  // that the stores (and the residual / aux loads of the fused epilogues) are ROW-CONTIGUOUS across lanes:
  // 16 lanes cover one 64-column row segment.  Row-scattered 8-B stores cost 15-25 us per GEMM here.
  constexpr int EP_COLS = TN * 32, EP_LD = EP_COLS + 4, CPR = EP_COLS / 4;  // 16-B chunks per row
  if constexpr (NW * 32 * EP_LD * 4 > NST * ST_BYTES) {  // staging does not fit (tuning shapes only): direct stores
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * (BM / WM) + j * 32 + frow;
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * (BN / WN) + i * 32 + 4 * fhalf + 8 * q;
          if (n < N)
            epi_quad<EPI, HT>(e, m, n, N, acc[i][j][4 * q] * asc, acc[i][j][4 * q + 1] * asc, acc[i][j][4 * q + 2] * asc,
                                  acc[i][j][4 * q + 3] * asc);
        }
    }
    return;
  }
  __syncthreads();  // every wave is done with the main-loop tiles
  float* ep = reinterpret_cast<float*>(lds) + wave * (32 * EP_LD);
  // Operands the epilogue READS per output chunk (fp32 residual; bf16 pre-activations for gelu' / relu') are fetched one
  // row block ahead, into registers, BEFORE the stores of the current row block: a load placed after a store cannot be
  // moved ahead of it by the compiler (possible aliasing), so without this every chunk paid a full memory latency.
  constexpr bool PRE_RES = (EPI == TIMHIP_EPI_DROP_RES_F32 || EPI == TIMHIP_EPI_ADD_F32) && !(epi_has_oct(EPI));
  constexpr bool PRE_AUX = (EPI == TIMHIP_EPI_DGELU_T || EPI == TIMHIP_EPI_DRELU_T || EPI == TIMHIP_EPI_MULAUX_T);
  constexpr int NITQ = 32 * CPR / 64, NITO = 32 * (EP_COLS / 8) / 64;
  constexpr int PD = TM <= 5 ? TM : 2;   // row blocks in flight: all of the tile's (<= 5: 80 VGPRs of fp32 residual), else 2
  float4 rbuf[PD][PRE_RES ? NITQ : 1];
  float2 sbuf[PD][PRE_RES ? NITQ : 1];   // (mean, rstd) of the rows in rbuf when the residual is LayerNorm(res)
  const bool pre_ln = PRE_RES && EPI == TIMHIP_EPI_DROP_RES_F32 && e.vec && e.res != nullptr && e.ln_stats != nullptr;
  vec8<HT> abuf[PD][PRE_AUX ? (NITO > 0 ? NITO : 1) : 1];
  const bool pre_res = PRE_RES && e.vec && e.res != nullptr;
  const bool pre_aux = PRE_AUX && e.vec8;
  auto fetch = [&](auto jc, auto slotc) {
    constexpr int j = decltype(jc)::value, slot = decltype(slotc)::value;
    if constexpr (PRE_RES) {
      if (pre_res) {
#pragma unroll
        for (int it = 0; it < NITQ; ++it) {
          const int idx = it * 64 + lane;
          const int row = idx / CPR, ch = idx % CPR;
          const int m = m0 + wm * (BM / WM) + j * 32 + row;
          const int n = n0 + wn * (BN / WN) + ch * 4;
          rbuf[slot][it] = (m < M && n + 3 < N) ? *reinterpret_cast<const float4*>(e.res + (size_t)m * e.ldres + n)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
          if (pre_ln) sbuf[slot][it] = m < M ? *reinterpret_cast<const float2*>(e.ln_stats + 2 * (size_t)m) : make_float2(0.f, 1.f);
        }
      }
    }
    if constexpr (PRE_AUX) {
      if (pre_aux) {
#pragma unroll
        for (int it = 0; it < NITO; ++it) {
          const int idx = it * 64 + lane;
          const int row = idx / (EP_COLS / 8), ch = idx % (EP_COLS / 8);
          const int m = m0 + wm * (BM / WM) + j * 32 + row;
          const int n = n0 + wn * (BN / WN) + ch * 8;
          if (m < M && n + 7 < N)
            abuf[slot][it] = *reinterpret_cast<const vec8<HT>*>((const HT*)e.aux + (size_t)m * e.ldaux + n);
        }
      }
    }
  };
  // bias: a lane's output columns are the same in every row chunk it handles (64 lanes cover whole rows of the 32- or
  // 64-column wave tile), so its bias values are fetched once
  float4 bias8[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, bias4 = bias8[0];
  bool pre_b8 = false, pre_b4 = false;
  if (e.bias && e.vec) {
    if constexpr (64 % (EP_COLS / 8) == 0) {
      const int n = n0 + wn * (BN / WN) + (lane % (EP_COLS / 8)) * 8;
      if (epi_has_oct(EPI) && e.vec8 && n + 7 < N) {
        bias8[0] = *reinterpret_cast<const float4*>(e.bias + n);
        bias8[1] = *reinterpret_cast<const float4*>(e.bias + n + 4);
        pre_b8 = true;
      }
    }
    if constexpr (64 % CPR == 0) {
      const int n = n0 + wn * (BN / WN) + (lane % CPR) * 4;
      if (n + 3 < N) { bias4 = *reinterpret_cast<const float4*>(e.bias + n); pre_b4 = true; }
    }
  }
  float4 lng = make_float4(1.f, 1.f, 1.f, 1.f), lnb = make_float4(0.f, 0.f, 0.f, 0.f);
  bool pre_gb = false;
  if constexpr (64 % CPR == 0) {
    if (pre_ln) {
      const int n = n0 + wn * (BN / WN) + (lane % CPR) * 4;
      if (n + 3 < N) {
        lng = *reinterpret_cast<const float4*>(e.ln_w + n);
        lnb = *reinterpret_cast<const float4*>(e.ln_b + n);
        pre_gb = true;
      }
    }
  }
  static_for<PD - 1>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    fetch(std::integral_constant<int, j>{}, std::integral_constant<int, j % PD>{});
  });
  static_for<TM>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j + PD - 1 < TM)
      fetch(std::integral_constant<int, j + PD - 1>{}, std::integral_constant<int, (j + PD - 1) % PD>{});
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(ep + frow * EP_LD + i * 32 + 8 * q + 4 * fhalf) =
            make_float4(acc[i][j][4 * q] * asc, acc[i][j][4 * q + 1] * asc, acc[i][j][4 * q + 2] * asc, acc[i][j][4 * q + 3] * asc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no block barrier needed
    if (epi_has_oct(EPI) && e.vec8) {
      constexpr int OPR = EP_COLS / 8;  // 16-byte output chunks per row
#pragma unroll
      for (int it = 0; it < 32 * OPR / 64; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / OPR, ch = idx % OPR;
        const float4 lo = *reinterpret_cast<const float4*>(ep + row * EP_LD + ch * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ep + row * EP_LD + ch * 8 + 4);
        const int m = m0 + wm * (BM / WM) + j * 32 + row;
        const int n = n0 + wn * (BN / WN) + ch * 8;
        if (m < M && n + 7 < N) {
          epi_oct<EPI, HT>(e, m, n, N, lo, hi, mbyte[j][it], pre_aux, abuf[j % PD][PRE_AUX ? it : 0], pre_b8, bias8[0], bias8[1]);
        } else if (m < M) {
          if (n < N) epi_quad<EPI, HT>(e, m, n, N, lo.x, lo.y, lo.z, lo.w);
          if (n + 4 < N) epi_quad<EPI, HT>(e, m, n + 4, N, hi.x, hi.y, hi.z, hi.w);
        }
      }
    } else {
      // (round 6, as in gemm_pp.hip's pp_epilogue: the dropout + residual epilogue's lane pairs share the Philox calls of two quads)
      constexpr int NITQ_ = 32 * CPR / 64;
      bool has_k = false;
      if constexpr (EPI == TIMHIP_EPI_DROP_RES_F32 && (NITQ_ % 2) == 0) has_k = e.thr != 0u && !e.mask && e.vec && (N & 7) == 0 && e.pair;
      auto quad = [&](int it, bool hk, float4 kf) {
        const int idx = it * 64 + lane;
        const int row = idx / CPR, ch = idx % CPR;
        const float4 v = *reinterpret_cast<const float4*>(ep + row * EP_LD + ch * 4);
        const int m = m0 + wm * (BM / WM) + j * 32 + row;
        const int n = n0 + wn * (BN / WN) + ch * 4;
        if (m < M && n < N)
          epi_quad<EPI, HT>(e, m, n, N, v.x, v.y, v.z, v.w, pre_res && n + 3 < N, rbuf[j % PD][PRE_RES ? it : 0],
                                pre_b4 && n + 3 < N, bias4, pre_ln && pre_gb && n + 3 < N, sbuf[j % PD][PRE_RES ? it : 0],
                                lng, lnb, hk, kf);
      };
      if (has_k) {
        if constexpr ((NITQ_ % 2) == 0) {
#pragma unroll
          for (int it = 0; it < NITQ_; it += 2) {
            const int idx = (it + (lane & 1)) * 64 + (lane & ~1);
            const long long m_ = m0 + wm * (BM / WM) + j * 32 + idx / CPR, n_ = n0 + wn * (BN / WN) + (idx % CPR) * 4;
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
        for (int it = 0; it < NITQ_; ++it) quad(it, false, make_float4(1.f, 1.f, 1.f, 1.f));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next block of rows overwrites
  });
  if constexpr (PROF) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t_end = __builtin_readcyclecounter();
    if (tid == 0 && e.aux) {
      unsigned long long* c = (unsigned long long*)e.aux;
      atomicAdd(c + 0, (unsigned long long)t_wait);
      atomicAdd(c + 1, (unsigned long long)t_issue);
      atomicAdd(c + 2, (unsigned long long)t_mma);
      atomicAdd(c + 3, (unsigned long long)(t_end - t_loop_end));
      atomicAdd(c + 4, (unsigned long long)(t_end - t_begin));
      atomicAdd(c + 5, 1ull);
      atomicAdd(c + 6, (unsigned long long)(wall_clock64() - w_begin));
    }
  }
}

template <typename HT, int EPI, int BM, int BN, int WM, int WN, int BKT, int NST, int GM, int ABL = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_h16_kernel(
    const HT* __restrict__ A, int lda, const HT* __restrict__ B, int ldb, int M, int N,
    int K, int ksteps_per_split, EpiDev e) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  gemm_nt_h16_body<HT, EPI, BM, BN, WM, WN, BKT, NST, GM, ABL>(A, lda, B, ldb, M, N, K, ksteps_per_split, e,
                                                           xcd_remap(blockIdx.x, tiles), (int)blockIdx.z);
}

// Grouped launch: several independent small GEMMs (the classification heads: four under-filled launches each way) as ONE
// grid - their tile lists concatenated, every block looks its problem up.  Side streams were the alternative and measured
// slower (every cross-stream edge costs more than such a launch lasts).
constexpr int GG_MAX = 6;
struct GemmGroupDev {
  const void* A[GG_MAX]; const void* B[GG_MAX];
  int lda[GG_MAX], ldb[GG_MAX], M[GG_MAX], N[GG_MAX], K[GG_MAX], tile0[GG_MAX + 1];
  EpiDev e[GG_MAX];
  int n;
};
template <typename HT, int EPI, int BM, int BN, int WM, int WN, int BKT, int NST>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_group_kernel(const GemmGroupDev g) {
  const int t = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < GG_MAX; ++k)
    if (k < g.n && t >= g.tile0[k]) i = k;
  gemm_nt_h16_body<HT, EPI, BM, BN, WM, WN, BKT, NST, 1, 0>((const HT*)g.A[i], g.lda[i], (const HT*)g.B[i], g.ldb[i], g.M[i], g.N[i], g.K[i],
                                                        g.K[i] / BK, g.e[i], t - g.tile0[i], 0);
}

// ---------------------------------------------------------------------------
// fp32 kernel: exact f32 MFMA (v_mfma_f32_32x32x2_f32), parity mode.
// 128x128x16 tile, 4 waves (2x2), register-staged, LDS rows padded to 17 floats.
// ---------------------------------------------------------------------------
constexpr int FBK = 16, FLD = 17;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb, int M,
                                                          int N, int K, int ksteps_per_split, EpiDev e) {
  constexpr int BM = 128, BN = 128;
  __shared__ float sA[BM * FLD];
  __shared__ float sB[BN * FLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const int nk_total = K / FBK;
  const int kt0 = blockIdx.z * ksteps_per_split;
  const int kt1 = min(nk_total, kt0 + ksteps_per_split);
  if (EPI == TIMHIP_EPI_STORE_F32) e.out0 = (float*)e.out0 + (long long)blockIdx.z * e.slab_stride;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  float4 ra[2], rb[2];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256, row = idx >> 2, c4 = idx & 3;
      ra[i] = *reinterpret_cast<const float4*>(A + (size_t)min(m0 + row, M - 1) * lda + kt * FBK + c4 * 4);
      rb[i] = *reinterpret_cast<const float4*>(B + (size_t)min(n0 + row, N - 1) * ldb + kt * FBK + c4 * 4);
    }
  };
  if (kt0 < kt1) fetch(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256, row = idx >> 2, c4 = idx & 3;
      float* pa = sA + row * FLD + c4 * 4;
      pa[0] = ra[i].x; pa[1] = ra[i].y; pa[2] = ra[i].z; pa[3] = ra[i].w;
      float* pb = sB + row * FLD + c4 * 4;
      pb[0] = rb[i].x; pb[1] = rb[i].y; pb[2] = rb[i].z; pb[3] = rb[i].w;
    }
    __syncthreads();
    if (kt + 1 < kt1) fetch(kt + 1);   // in flight while this step's MFMAs run
#pragma unroll
    for (int kk = 0; kk < FBK / 2; ++kk) {
      float xa[2], wb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) xa[j] = sA[(wm * 64 + j * 32 + frow) * FLD + kk * 2 + fhalf];
#pragma unroll
      for (int i = 0; i < 2; ++i) wb[i] = sB[(wn * 64 + i * 32 + frow) * FLD + kk * 2 + fhalf];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[i], xa[j], acc[i][j], 0, 0, 0);
    }
  }
  const float asc = e.acc_scale ? *e.acc_scale : 1.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + frow;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nb = n0 + wn * 64 + i * 32 + 4 * fhalf;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nb + 8 * q;
        if (n < N)
          epi_quad<EPI, float>(e, m, n, N, acc[i][j][4 * q] * asc, acc[i][j][4 * q + 1] * asc, acc[i][j][4 * q + 2] * asc,
                               acc[i][j][4 * q + 3] * asc);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// bf16x3 kernel (TIMHIP_PREC_BF16X3): fp32 operands, each split on the fly into hi = bf16(x) and
// lo = bf16(x - hi); three bf16 MFMAs per product (hi*hi + hi*lo + lo*hi, the lo*lo term is < 2^-16
// relative) with fp32 accumulation: ~16 mantissa bits per operand at 3x the MFMA work of plain bf16
// (still ~5x the rate of the exact f32 MFMA).  128x128x32 tile, 4 waves, register-staged.
// ---------------------------------------------------------------------------
constexpr int XBK = 32;

__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = (bf16_t)x;
  lo = (bf16_t)(x - (float)hi);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_x3_kernel(const float* __restrict__ A, int lda,
                                                         const float* __restrict__ B, int ldb, int M, int N, int K,
                                                         int ksteps_per_split, EpiDev e) {
  constexpr int BM = 128, BN = 128, ROWB = XBK * 2, TILE = BM * ROWB;  // bf16 tile bytes
  __shared__ __attribute__((aligned(16))) char sm[4 * TILE];            // Ahi | Alo | Bhi | Blo
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const int nk_total = K / XBK;
  const int spl = ksteps_per_split * (BK / XBK);
  const int kt0 = blockIdx.z * spl;
  const int kt1 = min(nk_total, kt0 + spl);
  if (EPI == TIMHIP_EPI_STORE_F32) e.out0 = (float*)e.out0 + (long long)blockIdx.z * e.slab_stride;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  float4 ra[4], rb[4];
  auto fetch = [&](int kt) {   // the fp32 operand tiles of K-step kt -> registers
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;  // 8 float4 per 32-float row
      ra[i] = *reinterpret_cast<const float4*>(A + (size_t)min(m0 + row, M - 1) * lda + kt * XBK + c4 * 4);
      rb[i] = *reinterpret_cast<const float4*>(B + (size_t)min(n0 + row, N - 1) * ldb + kt * XBK + c4 * 4);
    }
  };
  if (kt0 < kt1) fetch(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();  // previous step's fragments consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
      const int off = row * ROWB + (((c4 >> 1) ^ kswz<32>(row)) << 4) + (c4 & 1) * 8;
      bf16x4_t h, l;
      const float va[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
      for (int u = 0; u < 4; ++u) { bf16_t hh, ll; split_bf16(va[u], hh, ll); h[u] = hh; l[u] = ll; }
      *reinterpret_cast<bf16x4_t*>(sm + off) = h;
      *reinterpret_cast<bf16x4_t*>(sm + TILE + off) = l;
      const float vb[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
#pragma unroll
      for (int u = 0; u < 4; ++u) { bf16_t hh, ll; split_bf16(vb[u], hh, ll); h[u] = hh; l[u] = ll; }
      *reinterpret_cast<bf16x4_t*>(sm + 2 * TILE + off) = h;
      *reinterpret_cast<bf16x4_t*>(sm + 3 * TILE + off) = l;
    }
    __syncthreads();
    if (kt + 1 < kt1) fetch(kt + 1);   // in flight while this step's MFMAs run
#pragma unroll
    for (int kk = 0; kk < XBK / 16; ++kk) {
      const int c = kk * 2 + fhalf;
      bf16x8_t xh[2], xl[2], wh[2], wl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wm * 64 + j * 32 + frow;
        const int off = row * ROWB + ((c ^ kswz<32>(row)) << 4);
        xh[j] = *reinterpret_cast<const bf16x8_t*>(sm + off);
        xl[j] = *reinterpret_cast<const bf16x8_t*>(sm + TILE + off);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wn * 64 + i * 32 + frow;
        const int off = row * ROWB + ((c ^ kswz<32>(row)) << 4);
        wh[i] = *reinterpret_cast<const bf16x8_t*>(sm + 2 * TILE + off);
        wl[i] = *reinterpret_cast<const bf16x8_t*>(sm + 3 * TILE + off);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[i], xh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[i], xl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[i], xh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  const float asc = e.acc_scale ? *e.acc_scale : 1.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + frow;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nb = n0 + wn * 64 + i * 32 + 4 * fhalf;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nb + 8 * q;
        if (n < N)
          epi_quad<EPI, float>(e, m, n, N, acc[i][j][4 * q] * asc, acc[i][j][4 * q + 1] * asc, acc[i][j][4 * q + 2] * asc,
                               acc[i][j][4 * q + 3] * asc);
      }
    }
  }
}

template <typename HT, int EPI, int BM, int BN, int WM, int WN, int BKT, int NST, int GM, int ABL = 0>
void launch_h16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const EpiDev& e, int splitk,
                 hipStream_t s) {
  const int nk = K / BK;
  const int per = (nk + splitk - 1) / splitk;
  dim3 grid(((M + BM - 1) / BM) * ((N + BN - 1) / BN), 1, splitk);
  const size_t shmem = (size_t)NST * (BM + BN) * BKT * 2;
  static PerDeviceOnce attr_set;  // idempotent; a benign race sets it twice at worst
  if (shmem > 64 * 1024 && attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_h16_kernel<HT, EPI, BM, BN, WM, WN, BKT, NST, GM, ABL>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  }
  hipLaunchKernelGGL((gemm_nt_h16_kernel<HT, EPI, BM, BN, WM, WN, BKT, NST, GM, ABL>), grid, dim3(WM * WN * 64), shmem, s,
                     (const HT*)A, lda, (const HT*)B, ldb, M, N, K, per, e);
}

// Tile height.  Two tiles are built: 128 x 128 (2 x 2 waves of 64 x 64) and 160 x 128 (1 x 4 waves of 160 x 32).
// The taller tile stages 71 instead of 64 FLOP per LDS-DMA byte (the kernel is staging-bound) and quantises the
// encoder's M = 64 windows x 155 tokens = 9920 = 62 x 160 rows exactly: N = 1024 -> 496 tiles = ONE round of the
// 512 block slots (2 per CU) instead of 624 = 1.2 rounds.  Measured on M = 9920 (tools/gemm_tune.py, variant 14):
// +8...+35 % on every shape of the layer.  Cost model: rounds of 512 slots x tile height; the tall tile also wins ties.
static inline bool tall_tile_wins(int M, int N, int splitk) {
  if (M <= 128) return false;
  const long long tn = (N + 127) / 128;
  const long long t128 = (long long)((M + 127) / 128) * tn * splitk, t160 = (long long)((M + 159) / 160) * tn * splitk;
  const long long c128 = ((t128 + 511) / 512) * 128, c160 = ((t160 + 511) / 512) * 160;
  if (c160 != c128) return c160 < c128;
  return (long long)((M + 159) / 160) * 160 <= (long long)((M + 127) / 128) * 128 + 32;
}

// LDS stages of the small-problem (64 x 128) instances: TIMHIP_GEMM_SMALL_NST = 2 / 3 / 4 forces; 0 (default) = by the block
// count - as many stages as leave every block of the launch resident at once (24 KiB per stage, 160 KiB per CU)
static inline int small_nst(long long blocks) {
  const int k = tim_knobs().gemm_small_nst;
  if (k >= 2 && k <= 4) return k;
  if (k == 1) return 2;
  return blocks <= 256 ? 4 : (blocks <= 512 ? 3 : 2);
}

// tile variants (bf16): 0 = default, others for tuning on selected epilogues
template <int EPI>
constexpr bool tunable() {
  return EPI == TIMHIP_EPI_STORE_T || EPI == TIMHIP_EPI_ADD_F32 || EPI == TIMHIP_EPI_DROP_RES_F32;
}

template <int EPI>
int launch(int precision, int variant, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
           const EpiDev& e, int splitk, hipStream_t s) {
  if (precision == TIMHIP_PREC_BF16X3) {
    const int nk = K / BK;
    const int per = (nk + splitk - 1) / splitk;
    dim3 grid(((M + 127) / 128) * ((N + 127) / 128), 1, splitk);
    hipLaunchKernelGGL(gemm_nt_x3_kernel<EPI>, grid, dim3(256), 0, s, (const float*)A, lda, (const float*)B, ldb, M,
                       N, K, per, e);
  } else if (precision == TIMHIP_PREC_FP32) {
    const int nk = K / FBK;
    const int per = (nk + splitk - 1) / splitk;
    dim3 grid(((M + 127) / 128) * ((N + 127) / 128), 1, splitk);
    hipLaunchKernelGGL(gemm_nt_f32_kernel<EPI>, grid, dim3(256), 0, s, (const float*)A, lda,
                       (const float*)B, ldb, M, N, K, per, e);
  } else {
#ifdef TIMHIP_TUNING
    if constexpr (EPI == TIMHIP_EPI_STORE_T) {  // ablation builds (tools/gemm_abl.py): variant = 100*ABL + tile
#define ABLV(T, ...) case T: launch_h16<bf16_t, EPI, __VA_ARGS__>(A, lda, B, ldb, M, N, K, e, splitk, s); return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
      switch (variant) {
        ABLV(100, 128, 128, 2, 2, 64, 2, 1, 1) ABLV(200, 128, 128, 2, 2, 64, 2, 1, 2) ABLV(300, 128, 128, 2, 2, 64, 2, 1, 3)
        ABLV(700, 128, 128, 2, 2, 64, 2, 1, 7) ABLV(800, 128, 128, 2, 2, 64, 2, 1, 8)
        ABLV(1600, 128, 128, 2, 2, 64, 2, 1, 16) ABLV(1614, 160, 128, 1, 4, 64, 2, 1, 16)
        ABLV(814, 160, 128, 1, 4, 64, 2, 1, 8) ABLV(114, 160, 128, 1, 4, 64, 2, 1, 1) ABLV(214, 160, 128, 1, 4, 64, 2, 1, 2)
        ABLV(314, 160, 128, 1, 4, 64, 2, 1, 3)
        ABLV(207, 128, 128, 2, 2, 64, 4, 8, 2) ABLV(208, 256, 256, 2, 4, 64, 2, 4, 2) ABLV(205, 256, 256, 4, 2, 32, 4, 4, 2)
        ABLV(203, 256, 128, 4, 2, 64, 3, 8, 2) ABLV(202, 128, 128, 2, 2, 32, 4, 8, 2)

        default: break;
      }
#undef ABLV
    }
    if (tunable<EPI>() && variant != 0) {   // (bf16 tile variants of tools/gemm_tune.py; variant 0 = the production dispatch below)
      switch (variant) {
        case 1: launch_h16<bf16_t, EPI, 128, 128, 2, 2, 64, 2, 8>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 2: launch_h16<bf16_t, EPI, 128, 128, 2, 2, 32, 4, 8>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 3: launch_h16<bf16_t, EPI, 256, 128, 4, 2, 64, 3, 8>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 4: launch_h16<bf16_t, EPI, 256, 256, 2, 4, 32, 4, 4>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 5: launch_h16<bf16_t, EPI, 256, 256, 4, 2, 32, 4, 4>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 6: launch_h16<bf16_t, EPI, 256, 128, 4, 2, 32, 4, 8>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 7: launch_h16<bf16_t, EPI, 128, 128, 2, 2, 64, 4, 8>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 8: launch_h16<bf16_t, EPI, 256, 256, 2, 4, 64, 2, 4>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 9: launch_h16<bf16_t, EPI, 128, 128, 2, 2, 32, 3, 1>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 11: launch_h16<bf16_t, EPI, 128, 128, 2, 2, 64, 2, 4>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 12: launch_h16<bf16_t, EPI, 128, 128, 2, 2, 64, 2, 16>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 13: launch_h16<bf16_t, EPI, 128, 128, 2, 2, 64, 2, 39>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 10: launch_h16<bf16_t, EPI, 128, 128, 2, 2, 32, 3, 8>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 14: launch_h16<bf16_t, EPI, 160, 128, 1, 4, 64, 2, 1>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 19: launch_h16<bf16_t, EPI, 192, 128, 2, 2, 64, 2, 1>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 22: launch_h16<bf16_t, EPI, 320, 128, 2, 4, 64, 2, 1>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 24: launch_h16<bf16_t, EPI, 160, 128, 1, 4, 64, 3, 1>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        case 25: launch_h16<bf16_t, EPI, 192, 128, 1, 4, 64, 2, 1>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
        default: launch_h16<bf16_t, EPI, 128, 128, 2, 2, 64, 2, 1>(A, lda, B, ldb, M, N, K, e, splitk, s); break;
      }
    } else
#endif
    {
      (void)variant;
      // small problems (front end, heads: a few hundred to a few thousand rows): when 128-row tiles would fill at most
      // half of the 512 block slots, 64 x 128 tiles (1 x 4 waves of 64 x 32, 48 KB of LDS: three blocks per CU) double
      // the number of blocks - these launches are occupancy-bound, not staging-bound
      const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * splitk;
      DISPATCH_H16(precision,
        if (t128 <= 256 && M > 64) {
          // (a block of such a launch is a CHAIN of K / 64 stage latencies: with two LDS stages one load is in flight while the
          //  previous stage is multiplied - 0.1 us of MFMAs - so a step lasts a memory latency; deeper rings divide that.
          //  TIMHIP_GEMM_SMALL_NST = 2 / 3 / 4 stages of 24 KiB: 3 / 2 / 1 blocks per CU)
          const int nst = small_nst((long long)((M + 63) / 64) * ((N + 127) / 128) * splitk);
          // With the latency chain shortened, a step costs a wave its six LDS-DMA issue stalls and eight MFMAs: the launches of at
          // most 256 blocks (one per CU: three SIMD slots of four idle) run the tile on EIGHT waves, 2 x 4 of 32 x 32 - half of both
          // per wave: C2a at 8 windows per GPU 1.843 -> 1.790 ms, C1 0.832 -> 0.826 (profiles/r06_aa_small_gemm_w8_ab.txt; the
          // two-blocks-per-CU launches gain nothing from it: r06_ab).  TIMHIP_GEMM_SMALL_W8=0: four waves.
          if (nst >= 4 && tim_knobs().gemm_small_w8 != 0) launch_h16<HT, EPI, 64, 128, 2, 4, 64, 4, 1>(A, lda, B, ldb, M, N, K, e, splitk, s);
          else if (nst >= 4) launch_h16<HT, EPI, 64, 128, 1, 4, 64, 4, 1>(A, lda, B, ldb, M, N, K, e, splitk, s);
          else if (nst == 3) launch_h16<HT, EPI, 64, 128, 1, 4, 64, 3, 1>(A, lda, B, ldb, M, N, K, e, splitk, s);
          else launch_h16<HT, EPI, 64, 128, 1, 4, 64, 2, 1>(A, lda, B, ldb, M, N, K, e, splitk, s);
        } else if (tall_tile_wins(M, N, splitk))
          launch_h16<HT, EPI, 160, 128, 1, 4, 64, 2, 1>(A, lda, B, ldb, M, N, K, e, splitk, s);
        else
          launch_h16<HT, EPI, 128, 128, 2, 2, 64, 2, 1>(A, lda, B, ldb, M, N, K, e, splitk, s));
    }
  }
  if (hipGetLastError() != hipSuccess) return TIMHIP_ELAUNCH;
  return TIMHIP_OK;
}

}  // namespace

// argument checks + device-side epilogue descriptor shared by the single and the grouped launch
static int prepare_gemm(int precision, int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                        const TimEpi& te, int splitk, EpiDev& e) {
  if (!A || !B || !te.out0) return TIMHIP_EINVAL;
  if (M <= 0 || N <= 0 || K <= 0) return TIMHIP_EINVAL;
  if (!valid_precision(precision)) return TIMHIP_EUNSUPPORTED;
  const int Kp = round_up(K, 64);
  if (te.a_wrap_k < 0 || te.a_wrap_k % 64 || (te.a_wrap_k > 0 && (Kp != 2 * te.a_wrap_k || !h16_storage(precision) || splitk > 1)))
    return TIMHIP_EUNSUPPORTED;   // A repeats once after a_wrap_k columns (K = 2 a_wrap_k): whole 64-deep steps, 16-bit kernels, no split of the contraction
  if (lda % 64 || ldb % 64 || lda < (te.a_wrap_k > 0 ? te.a_wrap_k : Kp) || ldb < Kp) return TIMHIP_EALIGN;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return TIMHIP_EALIGN;
  if ((size_t)M * lda * 2 >= (1ull << 32) || (size_t)N * ldb * 2 >= (1ull << 32)) return TIMHIP_EUNSUPPORTED;  // 32-bit lane offsets
  if (splitk > 1 && epi != TIMHIP_EPI_ATOMIC_F32 && epi != TIMHIP_EPI_STORE_F32) return TIMHIP_EINVAL;
  e.out0 = te.out0; e.out1 = te.out1; e.bias = te.bias; e.res = te.res; e.aux = te.aux;
  e.ld0 = te.ld0; e.ld1 = te.ld1; e.ldres = te.ldres; e.ldaux = te.ldaux;
  e.thr = te.p_drop > 0.f ? drop_threshold(te.p_drop) : 0u;
  e.scale = te.p_drop > 0.f ? 1.f / (1.f - te.p_drop) : 1.f;
  e.site = te.site; e.seed = te.seed;
  e.mask = (const uint8_t*)te.mask; e.ldmask = te.ldmask;
  e.ln_stats = te.ln_stats; e.ln_w = te.ln_w; e.ln_b = te.ln_b;
  e.acc_scale = te.acc_scale;
  e.a_wrap = te.a_wrap_k / 64;
  e.pair = tim_knobs().epi_pair;
  if (e.ln_stats && (epi != TIMHIP_EPI_DROP_RES_F32 || !e.res || !e.ln_w || !e.ln_b)) return TIMHIP_EINVAL;
  if (epi == TIMHIP_EPI_RELU_SPLIT3_T && (!h16_storage(precision) || e.out1 || e.ld1 % 64 || e.ld1 < N || e.ld0 < 3 * e.ld1 || splitk > 1))
    return TIMHIP_EINVAL;   // three column blocks of width ld1 in a row of stride ld0
  e.slab_stride = splitk > 1 ? (long long)M * te.ld0 : 0;

  bool vec = (e.ld0 % 4 == 0) && (((uintptr_t)e.out0 & 15) == 0);
  if (e.out1) vec = vec && (e.ld1 % 4 == 0) && (((uintptr_t)e.out1 & 15) == 0);
  if (e.res) vec = vec && (e.ldres % 4 == 0) && (((uintptr_t)e.res & 15) == 0);
  if (e.aux) vec = vec && (e.ldaux % 4 == 0) && (((uintptr_t)e.aux & 15) == 0);
  if (e.bias) vec = vec && (((uintptr_t)e.bias & 15) == 0);
  e.vec = vec ? 1 : 0;
  // 8-element accesses of the bf16 outputs / aux (N % 4 == 0 is implied where it matters: dropout needs it, and a
  // ragged last chunk falls back to quads)
  bool vec8 = vec && h16_storage(precision) && (e.ld0 % 8 == 0);
  if (e.out1) vec8 = vec8 && (e.ld1 % 8 == 0);
  if (e.aux) vec8 = vec8 && (e.ldaux % 8 == 0);
  e.vec8 = vec8 ? 1 : 0;
  if (e.thr != 0u && (N % 4) != 0) return TIMHIP_EUNSUPPORTED;
  return TIMHIP_OK;
}

// n <= 6 independent bf16 problems with the same epilogue as one grid of 64 x 128 tiles (small problems: the heads)
int tim_gemm_nt_group(int precision, int epi, const TimGemmItem* items, int n, hipStream_t s) {
  if (!h16_storage(precision)) return TIMHIP_EUNSUPPORTED;
  if (!items || n < 1 || n > GG_MAX) return TIMHIP_EINVAL;
  if (epi != TIMHIP_EPI_STORE_F32 && epi != TIMHIP_EPI_ADD_F32 && epi != TIMHIP_EPI_STORE_T && epi != TIMHIP_EPI_RELU_T)
    return TIMHIP_EUNSUPPORTED;
  GemmGroupDev g;
  g.n = n;
  g.tile0[0] = 0;
  double flops = 0.0;
  for (int i = 0; i < GG_MAX; ++i) {
    if (i >= n) {
      g.A[i] = g.B[i] = nullptr; g.lda[i] = g.ldb[i] = g.M[i] = g.N[i] = g.K[i] = 0; g.tile0[i + 1] = g.tile0[i];
      g.e[i] = g.e[0];
      continue;
    }
    const TimGemmItem& t = items[i];
    const int rc = prepare_gemm(precision, epi, t.A, t.lda, t.B, t.ldb, t.M, t.N, t.K, t.e, 1, g.e[i]);
    if (rc) return rc;
    g.A[i] = t.A; g.B[i] = t.B; g.lda[i] = t.lda; g.ldb[i] = t.ldb;
    g.M[i] = t.M; g.N[i] = t.N; g.K[i] = round_up(t.K, 64);
    g.tile0[i + 1] = g.tile0[i] + ((t.M + 63) / 64) * ((t.N + 127) / 128);
    flops += 2.0 * t.M * t.N * t.K / (t.reserved > 1 ? t.reserved : 1);   // reserved: operand replication of a split GEMM
  }
  TimGemmScope timing(flops, s);
  if (g.tile0[n] <= 512 && epi == TIMHIP_EPI_ADD_F32) {
    // still under-filled with 64 x 128 tiles (the heads' input gradients: one long-K item dominates): 64 x 64 tiles double
    // the blocks again
    for (int i = 0; i < n; ++i) g.tile0[i + 1] = g.tile0[i] + ((items[i].M + 63) / 64) * ((items[i].N + 63) / 64);
    for (int i = n; i < GG_MAX; ++i) g.tile0[i + 1] = g.tile0[n];
    DISPATCH_H16(precision, hipLaunchKernelGGL((gemm_nt_group_kernel<HT, TIMHIP_EPI_ADD_F32, 64, 64, 2, 2, 64, 2>),
                                               dim3((unsigned)g.tile0[n]), dim3(256), (size_t)2 * (64 + 64) * 64 * 2, s, g));
    return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
  }
  const dim3 grid((unsigned)g.tile0[n]);
  const int nst = small_nst(g.tile0[n]);
  const size_t shmem = (size_t)nst * (64 + 128) * 64 * 2;
#define GROUP_N(X, NS) do { DISPATCH_H16(precision, (void)hipFuncSetAttribute((const void*)gemm_nt_group_kernel<HT, X, 64, 128, 1, 4, 64, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    DISPATCH_H16(precision, hipLaunchKernelGGL((gemm_nt_group_kernel<HT, X, 64, 128, 1, 4, 64, NS>), grid, dim3(256), shmem, s, g)); } while (0)
#define GROUP(X) case X: if (nst >= 4) GROUP_N(X, 4); else if (nst == 3) GROUP_N(X, 3); else GROUP_N(X, 2); break;
  switch (epi) {
    GROUP(TIMHIP_EPI_STORE_F32)
    GROUP(TIMHIP_EPI_ADD_F32)
    GROUP(TIMHIP_EPI_STORE_T)
    GROUP(TIMHIP_EPI_RELU_T)
    default: return TIMHIP_EUNSUPPORTED;
  }
#undef GROUP
#undef GROUP_N
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}

int tim_gemm_nt_fuse_ln(int precision, const void* A, int lda, const void* B, int ldb, int M, int N, int K, const TimEpi& te,
                        const TimLnFuse& lf, const uint32_t** fail, hipStream_t s) {
  EpiDev e;
  const int rc0 = prepare_gemm(precision, TIMHIP_EPI_DROP_RES_F32, A, lda, B, ldb, M, N, K, te, 1, e);
  if (rc0) return rc0;
  if (K % 64) return TIMHIP_EUNSUPPORTED;
  TimGemmScope timing(2.0 * M * N * K, s);
  return tim_gemm_nt_pp_ln(precision, A, lda, B, ldb, M, N, K, &e, lf, fail, s);
}

int tim_gemm_nt(int precision, int epi, const void* A, int lda, const void* B, int ldb, int M, int N,
                int K, const TimEpi& te, int splitk, hipStream_t s) {
  if (splitk < 1) splitk = 1;
  EpiDev e;
  const int rc0 = prepare_gemm(precision, epi, A, lda, B, ldb, M, N, K, te, splitk, e);
  if (rc0) return rc0;
  const int Kp = round_up(K, 64);
  TimGemmScope timing(2.0 * M * N * K / (te.reserved > 1 ? te.reserved : 1), s);   // te.reserved: operand replication (split GEMM)
  // production-batch layer shapes: the one-block-per-CU ping-pong kernel (TIMHIP_GEMM_PP=0 turns it off: A/B switch)
  if (tim_knobs().gemm_pp != 0 && h16_storage(precision) && tim_gemm_pp_wins(M, N, Kp, splitk)) {
    const int rc = tim_gemm_nt_pp(precision, epi, A, lda, B, ldb, M, N, Kp, &e, s);
    if (rc != TIMHIP_EUNSUPPORTED) return rc;
  }
  int variant = 0;
#ifdef TIMHIP_TUNING  // tools/gemm_tune.py, tools/gemm_abl.py: make -C tim_amd/csrc TUNING=1
  if (const char* v = getenv("TIMHIP_GEMM_VARIANT")) variant = atoi(v);
  if (getenv("TIMHIP_GEMM_ALIAS")) { lda = 0; ldb = 0; }  // every operand row aliases row 0 (cache-resident)
#endif
  switch (epi) {
#define CASE(X) case X: return launch<X>(precision, variant, A, lda, B, ldb, M, N, Kp, e, splitk, s);
    CASE(TIMHIP_EPI_STORE_T)
    CASE(TIMHIP_EPI_RELU_T)
    CASE(TIMHIP_EPI_STORE_F32)
    CASE(TIMHIP_EPI_GELU_DROP_T2)
    CASE(TIMHIP_EPI_DROP_RES_F32)
    CASE(TIMHIP_EPI_ADD_F32)
    CASE(TIMHIP_EPI_DGELU_T)
    CASE(TIMHIP_EPI_DRELU_T)
    CASE(TIMHIP_EPI_ATOMIC_F32)
    CASE(TIMHIP_EPI_SIGMOID_F32)
    CASE(TIMHIP_EPI_DRELU_F32IN_T)
    CASE(TIMHIP_EPI_GELU_DROP_G2)
    CASE(TIMHIP_EPI_MULAUX_T)
    CASE(TIMHIP_EPI_RELU_SPLIT3_T)
#undef CASE
    default: return TIMHIP_EINVAL;
  }
}
