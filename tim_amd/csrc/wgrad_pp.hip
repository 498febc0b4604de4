// Weight-gradient GEMMs of one encoder layer  dW_i[N_i,K_i] = dY_i[M,N_i]^T X_i[M,K_i]  (+ db_i = colsum dY_i)  as ONE grid of
// one-workgroup-per-CU "ping-pong" blocks - the transposing-read (TN) counterpart of gemm_pp.hip.
// Replaces the weight/bias gradients autograd computes for transformers.py:73,102,107 (nn.MultiheadAttention in/out
// projections, linear1, linear2).
//
// wgrad.hip's 128 x 128 tiles (two 4-wave blocks per CU, 64 x 64 wave tiles of 32x32x16 MFMAs) are bound by LDS issue: two
// half-width transposing reads per MFMA.  Here:
//   * 128 (n) x 256 (k) output tile, ONE 8-wave block per CU, 2 x 4 waves of 64 x 64 = 4 x 4 v_mfma_f32_16x16x32 tiles: 16
//     ds_read_b64_tr_b16 feed 16 MFMAs per 32-row contraction step (one read per MFMA), accumulators 64 VGPRs.  The four
//     gradients of a C2a layer are 64 + 64 + 32 + 96 = 256 such tiles: exactly one per CU, every block runs the whole M;
//   * both operands are staged in their natural [m][column] layout (the contraction runs over the STRIDED dimension of both),
//     64 rows per stage as three 16-KiB sub-tiles [64][128 columns] (dY, X left half, X right half), by global_load_lds into a
//     3-stage ring (144 KiB); the MFMA fragments (8 consecutive m per lane) come out of two transposing reads each;
//   * the two waves of a SIMD alternate LOAD (fragment reads) and MFMA phases as in gemm_pp.hip (waves 4-7 one barrier
//     behind waves 0-3).
// Four schedules of that tile exist (tools/wgpp_abl.py interleaves them on one box; C2a layer launch, us):
//     MODE 0  a LOAD and an MFMA phase per 32-row half step, 3 DMA pieces between the MFMAs of each phase   196-213
//     MODE 1  no phase barriers (one barrier per 64-row step), hardware interleaving of a SIMD's two waves  205-229
//     MODE 2  ONE LOAD and ONE MFMA phase per 64-row step (32 reads / 32 MFMAs, two barriers instead of four) 187-207
//     wgrad_ld_kernel  MODE 2's phases on 8 consumer waves + 4 loader waves that issue every DMA piece      156-159
//   The last one is the default (1.05 PFLOP/s on the layer).  What it removes: a global_load_lds stalls the issuing wave
//   for 60-180 cycles, and between a consumer's MFMAs that is matrix-pipe time - MODE 2 runs 134 us with its DMA pieces
//   ablated, 187+ with them.  A further variant - FOUR consumer waves with 64 x 128 tiles (0.75 reads per MFMA, each wave
//   overlapping its own reads with its own MFMAs) + four loaders, one barrier per step - measured 192-207 us: one wave per
//   SIMD cannot keep the LDS busy (its reads + DMA alone, MFMAs ablated: 181 us against 129); not kept.  (Round-2 note, superseded: MODE 1 was measured ahead of MODE 0 on one box, 217 vs 232; with
//   the modes interleaved in one process MODE 0 is ahead on every box tried.)
// Hazard bookkeeping of MODES 0 / 2: see gemm_pp.hip (identical phase structure); of the loader form: at wl_consume below.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "mfma_tiles.h"

namespace {

constexpr int WP_TN = 128, WP_TK = 256, WP_M = 64, WP_NST = 3;
constexpr int WP_SUB = WP_M * 256;            // bytes of one [64][128] 16-bit sub-tile (256-byte rows)
constexpr int WP_STAGE = 3 * WP_SUB;          // dY | X[:, 0:128] | X[:, 128:256]
constexpr int WP_MAX = 8;

struct WpGroup {
  const void* dY[WP_MAX]; const void* X[WP_MAX]; float* dW[WP_MAX]; float* db[WP_MAX];
  int ldy[WP_MAX], ldx[WP_MAX], N[WP_MAX], K[WP_MAX];
  int tile0[WP_MAX + 1];
  int n, M, accumulate;
  const float* out_scale;
  int pf_dist;   // wgrad_ld_kernel: L2 prefetch distance in 64-row stages (0: off)
};

__device__ __forceinline__ void wp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wp_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int xcd_remap_wp(int b, int nb) {  // bijective for any grid size
  const int q = nb >> 3, r = nb & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct WpTile {
  const char* dY; const char* X;   // operand bases (bytes)
  int ldy, ldx, N, K, n0, k0;
};

// G = 0: waves 0-3 (n rows 0-63 of the tile), G = 1: waves 4-7 (n rows 64-127), one barrier behind
// ABL (tuning builds, tools/wgpp_abl.py): timing-only ablations - 1: fragment reads only in the first step, 2: no MFMAs,
// 4: no DMA pieces after the prologue
template <typename HT, int G, int ABL = 0>
__device__ __forceinline__ void wp_mainloop(const WpTile& a, int M, const char* lds, uint32_t lds0, int wave, int lane,
                                            const int (&xoff)[4][2], const int (&yoff)[4][2], f32x4_t (&acc)[4][4],
                                            f32x4_t (&accb)[4], bool do_bias) {
  const int nk = (M + WP_M - 1) / WP_M;
  const bool partial = (M % WP_M) != 0;
  // this wave's 6 DMA pieces (1 KiB = 4 rows x 256 B): pieces 6 wave .. 6 wave + 5 of the stage's 48 (16 per sub-tile)
  const int lrow = lane >> 4, lc = lane & 15;
  uint32_t off[6];
  int prow[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int p = wave * 6 + i, sub = p >> 4, row = (p & 15) * 4 + lrow;
    const int c = lc ^ swz<128>(row);
    prow[i] = row;
    // column chunks beyond the operand's own columns (rounded up to 8) are clamped: they only feed outputs that are never stored
    if (sub == 0) off[i] = (uint32_t)(((size_t)row * a.ldy + min(a.n0 + c * 8, ((a.N + 7) & ~7) - 8)) * 2);
    else off[i] = (uint32_t)(((size_t)row * a.ldx + min(a.k0 + (sub - 1) * 128 + c * 8, ((a.K + 7) & ~7) - 8)) * 2);
  }
  auto piece = [&](int kt, int slot, int i) {
    const int p = wave * 6 + i;
    const bool is_y = p < 16;
    const char* g = (is_y ? a.dY : a.X) + (size_t)kt * WP_M * (is_y ? a.ldy : a.ldx) * 2;
    const uint32_t dst = lds0 + slot * WP_STAGE + p * 1024;
    if (partial && kt == nk - 1) {   // rows past the end re-read row M - 1 (finite values; masked to zero in the fragments)
      const int over = max(kt * WP_M + prow[i] - (M - 1), 0);
      glds16(g + off[i] - (size_t)over * (is_y ? a.ldy : a.ldx) * 2, dst);
    } else {
      glds16_s(uniform_ptr(g), off[i], dst);
    }
  };

#pragma unroll
  for (int i = 0; i < 6; ++i) piece(0, 0, i);
  if (nk > 1) {
#pragma unroll
    for (int i = 0; i < 6; ++i) piece(1, 1, i);
    glds_wait<6>();
  } else {
    glds_wait<0>();
  }
  wp_barrier();
  if constexpr (G == 1) wp_barrier();

  const int wk = wave & 3;
  const int gid = lane >> 4;
  vec8<HT> xf[4], yf[4];
  int slot = 0;
  auto step = [&](int t, auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;
    const char* sY = lds + slot * WP_STAGE;
    const char* sX = sY + WP_SUB + (wk >> 1) * WP_SUB;
    const int nslot = slot >= 1 ? slot - 1 : WP_NST - 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // ---- LOAD phase: fragments of the 32 rows (step t, half): lane (gid, p) holds rows 8 gid .. + 7 of column p of a tile
      const int hb = half * (32 * 256);
      if (!(ABL & 1) || t == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = cat8<HT>(tr_read<HT>(sX + xoff[i][0] + hb), tr_read<HT>(sX + xoff[i][1] + hb));
#pragma unroll
        for (int j = 0; j < 4; ++j) yf[j] = cat8<HT>(tr_read<HT>(sY + yoff[j][0] + hb), tr_read<HT>(sY + yoff[j][1] + hb));
      }
      if (half == 1) {
        if constexpr (MORE) glds_wait<3>(); else glds_wait<0>();
      }
      wp_wait_lds();
      if (!MORE && partial && t == nk - 1) {   // last, partial step: rows >= M contribute zeros (one operand suffices: the other is finite)
        const int nvalid = M - (t * WP_M + half * 32 + 8 * gid);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (u >= nvalid) yf[j][u] = (HT)0.f;
      }
      wp_barrier();
      // ---- MFMA phase: 16 MFMAs, this wave's three DMA pieces of stage t + 2 between them
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!(ABL & 2)) acc[i][j] = mfma16x16<HT>(xf[i], yf[j], acc[i][j]);
          else if (i == 0 && j == 0) acc[0][0][0] += (float)xf[0][0] * (float)yf[0][0];
          const int q = i * 4 + j;
          if (MORE && !(ABL & 4) && q % 5 == 2) {   // after MFMAs 2, 7, 12
            __builtin_amdgcn_sched_barrier(0);
            piece(t + 2, nslot, half * 3 + q / 5);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if (do_bias) {   // bias gradient (waves of the first k-tile column): column sums of dY on the matrix pipe too - an
        // all-ones A fragment makes every row of D the column sums (4 MFMAs; summing the fragments on the VALU costs 96
        // instructions per phase and made these waves, hence their whole block, three times slower)
        vec8<HT> ones;
#pragma unroll
        for (int u = 0; u < 8; ++u) ones[u] = (HT)1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) accb[j] = mfma16x16<HT>(ones, yf[j], accb[j]);
      }
      wp_barrier();
    }
    slot = slot + 1 == WP_NST ? 0 : slot + 1;
  };
  int t = 0;
  for (; t + 2 < nk; ++t) step(t, std::true_type{});
  for (; t < nk; ++t) step(t, std::false_type{});
  if constexpr (G == 0) wp_barrier();
}

// Merged-phase ping-pong (MODE 2): ONE LOAD phase (the 32 transposing reads of both 32-row halves, 64 fragment VGPRs) and ONE
// MFMA phase (32 MFMAs, this wave's six DMA pieces of stage t + 2 between them) per 64-row step - two barriers per step
// instead of four, phases long enough (512 matrix-pipe cycles) to carry an 8-wave barrier's ~150 cycles.  Same hazard
// bookkeeping as gemm_pp.hip's loop (its header): G1 runs one barrier behind G0.
// (Issuing the six pieces in the LOAD phase instead, right behind the fragment reads, measured 3-5 % slower: piece issue and
// the transposing reads serialise.)
template <typename HT, int G, int ABL = 0>
__device__ __forceinline__ void wp_mainloop_merged(const WpTile& a, int M, const char* lds, uint32_t lds0, int wave, int lane,
                                                   const int (&xoff)[4][2], const int (&yoff)[4][2], f32x4_t (&acc)[4][4],
                                                   f32x4_t (&accb)[4], bool do_bias) {
  const int nk = (M + WP_M - 1) / WP_M;
  const bool partial = (M % WP_M) != 0;
  const int lrow = lane >> 4, lc = lane & 15;
  uint32_t off[6];
  int prow[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int p = wave * 6 + i, sub = p >> 4, row = (p & 15) * 4 + lrow;
    const int c = lc ^ swz<128>(row);
    prow[i] = row;
    if (sub == 0) off[i] = (uint32_t)(((size_t)row * a.ldy + min(a.n0 + c * 8, ((a.N + 7) & ~7) - 8)) * 2);
    else off[i] = (uint32_t)(((size_t)row * a.ldx + min(a.k0 + (sub - 1) * 128 + c * 8, ((a.K + 7) & ~7) - 8)) * 2);
  }
  auto piece = [&](int kt, int slot, int i) {
    const int p = wave * 6 + i;
    const bool is_y = p < 16;
    const char* g = (is_y ? a.dY : a.X) + (size_t)kt * WP_M * (is_y ? a.ldy : a.ldx) * 2;
    const uint32_t dst = lds0 + slot * WP_STAGE + p * 1024;
    if (partial && kt == nk - 1) {
      const int over = max(kt * WP_M + prow[i] - (M - 1), 0);
      glds16(g + off[i] - (size_t)over * (is_y ? a.ldy : a.ldx) * 2, dst);
    } else {
      glds16_s(uniform_ptr(g), off[i], dst);
    }
  };
#pragma unroll
  for (int i = 0; i < 6; ++i) piece(0, 0, i);
  if (nk > 1) {
#pragma unroll
    for (int i = 0; i < 6; ++i) piece(1, 1, i);
    glds_wait<6>();
  } else {
    glds_wait<0>();
  }
  wp_barrier();
  if constexpr (G == 1) wp_barrier();

  const int wk = wave & 3, gid = lane >> 4;
  vec8<HT> xf[2][4], yf[2][4];
  int slot = 0;
  auto step = [&](int t, auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;
    const char* sY = lds + slot * WP_STAGE;
    const char* sX = sY + WP_SUB + (wk >> 1) * WP_SUB;
    const int nslot = slot >= 1 ? slot - 1 : WP_NST - 1;
    // ---- LOAD phase
    if (!(ABL & 1) || t == 0) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int hb = half * (32 * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[half][i] = cat8<HT>(tr_read<HT>(sX + xoff[i][0] + hb), tr_read<HT>(sX + xoff[i][1] + hb));
#pragma unroll
        for (int j = 0; j < 4; ++j) yf[half][j] = cat8<HT>(tr_read<HT>(sY + yoff[j][0] + hb), tr_read<HT>(sY + yoff[j][1] + hb));
      }
    }
    glds_wait<0>();   // this wave's pieces of stage t + 1 (issued in its previous MFMA phase) have landed
    wp_wait_lds();
    if (!MORE && partial && t == nk - 1) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int nvalid = M - (t * WP_M + half * 32 + 8 * gid);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (u >= nvalid) yf[half][j][u] = (HT)0.f;
      }
    }
    wp_barrier();
    // ---- MFMA phase
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!(ABL & 2)) acc[i][j] = mfma16x16<HT>(xf[half][i], yf[half][j], acc[i][j]);
          else if (i == 0 && j == 0) acc[0][0][0] += (float)xf[half][0][0] * (float)yf[half][0][0];
          const int q = half * 16 + i * 4 + j;
          if (MORE && !(ABL & 4) && q % 5 == 2) {   // after MFMAs 2, 7, 12, 17, 22, 27
            __builtin_amdgcn_sched_barrier(0);
            piece(t + 2, nslot, q / 5);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    if (do_bias) {
      vec8<HT> ones;
#pragma unroll
      for (int u = 0; u < 8; ++u) ones[u] = (HT)1.f;
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int j = 0; j < 4; ++j) accb[j] = mfma16x16<HT>(ones, yf[half][j], accb[j]);
    }
    wp_barrier();
    slot = slot + 1 == WP_NST ? 0 : slot + 1;
  };
  int t = 0;
  for (; t + 2 < nk; ++t) step(t, std::true_type{});
  for (; t < nk; ++t) step(t, std::false_type{});
  if constexpr (G == 0) wp_barrier();
}

// Free-running form (MODE 1): no phase barriers.  The ring alone needs ONE barrier per 64-row step (3 slots: the barrier of
// step t says "everybody's pieces of stage t have landed" and, by program order, "everybody is done reading stage t-1", whose
// slot is refilled right after it); the two waves of a SIMD interleave through the hardware scheduler - a wave stalled on
// its fragment reads or on the issue of a DMA piece leaves the matrix pipe to its partner.
template <typename HT, int ABL = 0>
__device__ __forceinline__ void wp_mainloop_free(const WpTile& a, int M, const char* lds, uint32_t lds0, int wave, int lane,
                                                 const int (&xoff)[4][2], const int (&yoff)[4][2], f32x4_t (&acc)[4][4],
                                                 f32x4_t (&accb)[4], bool do_bias) {
  const int nk = (M + WP_M - 1) / WP_M;
  const bool partial = (M % WP_M) != 0;
  const int lrow = lane >> 4, lc = lane & 15;
  uint32_t off[6];
  int prow[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int p = wave * 6 + i, sub = p >> 4, row = (p & 15) * 4 + lrow;
    const int c = lc ^ swz<128>(row);
    prow[i] = row;
    if (sub == 0) off[i] = (uint32_t)(((size_t)row * a.ldy + min(a.n0 + c * 8, ((a.N + 7) & ~7) - 8)) * 2);
    else off[i] = (uint32_t)(((size_t)row * a.ldx + min(a.k0 + (sub - 1) * 128 + c * 8, ((a.K + 7) & ~7) - 8)) * 2);
  }
  auto piece = [&](int kt, int slot, int i) {
    const int p = wave * 6 + i;
    const bool is_y = p < 16;
    const char* g = (is_y ? a.dY : a.X) + (size_t)kt * WP_M * (is_y ? a.ldy : a.ldx) * 2;
    const uint32_t dst = lds0 + slot * WP_STAGE + p * 1024;
    if (partial && kt == nk - 1) {
      const int over = max(kt * WP_M + prow[i] - (M - 1), 0);
      glds16(g + off[i] - (size_t)over * (is_y ? a.ldy : a.ldx) * 2, dst);
    } else {
      glds16_s(uniform_ptr(g), off[i], dst);
    }
  };
#pragma unroll
  for (int i = 0; i < 6; ++i) piece(0, 0, i);
  if (nk > 1) {
#pragma unroll
    for (int i = 0; i < 6; ++i) piece(1, 1, i);
  }
  const int wk = wave & 3, gid = lane >> 4;
  vec8<HT> xf[2][4], yf[2][4];
  int slot = 0;
  for (int t = 0; t < nk; ++t) {
    // my pieces of stage t have landed (those of stage t + 1 may stay in flight); then everybody's have
    if (t + 1 < nk) glds_wait<6>(); else glds_wait<0>();
    wp_barrier();
    const char* sY = lds + slot * WP_STAGE;
    const char* sX = sY + WP_SUB + (wk >> 1) * WP_SUB;
    const int nslot = slot >= 1 ? slot - 1 : WP_NST - 1;
    auto load = [&](int half, int set) {
      const int hb = half * (32 * 256);
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[set][i] = cat8<HT>(tr_read<HT>(sX + xoff[i][0] + hb), tr_read<HT>(sX + xoff[i][1] + hb));
#pragma unroll
      for (int j = 0; j < 4; ++j) yf[set][j] = cat8<HT>(tr_read<HT>(sY + yoff[j][0] + hb), tr_read<HT>(sY + yoff[j][1] + hb));
      if (partial && t == nk - 1) {
        const int nvalid = M - (t * WP_M + half * 32 + 8 * gid);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (u >= nvalid) yf[set][j][u] = (HT)0.f;
      }
    };
    load(0, 0);
    if (!(ABL & 4) && t + 2 < nk) {   // slot of stage t - 1: free since the barrier above
#pragma unroll
      for (int i = 0; i < 6; ++i) piece(t + 2, nslot, i);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 0) load(1, 1);   // the second half's fragments are in flight under the first half's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(ABL & 2)) acc[i][j] = mfma16x16<HT>(xf[half][i], yf[half][j], acc[i][j]);
      if (do_bias) {
        vec8<HT> ones;
#pragma unroll
        for (int u = 0; u < 8; ++u) ones[u] = (HT)1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) accb[j] = mfma16x16<HT>(ones, yf[half][j], accb[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    wp_wait_lds();   // (all fragment reads of this step retired before the next barrier lets the slot be refilled)
    slot = slot + 1 == WP_NST ? 0 : slot + 1;
  }
}

// ---- loader-wave form (wgrad_ld_kernel, 12 waves) -----------------------------------------------------------------------
// The same tile, ring and merged LOAD / MFMA phases, but the LDS-DMA pieces are issued by four extra waves (one per SIMD)
// that do nothing else: a global_load_lds stalls its wave for 60-180 cycles at issue, and between a consumer's MFMAs that
// stall is matrix-pipe time (tools/wgpp_abl.py: the merged loop runs 187 us with its pieces, 133 without).  The loaders also
// take the bias gradient (column sums of dY by the ones-MFMA) off the consumers - 16 accumulator VGPRs less, which together
// with XOR-derived fragment offsets brings the consumers under the 168 VGPRs three waves per SIMD may use.
// Barriers b0, b1, ...: G0's LOAD(t) runs in window A_t = (b_2t, b_2t+1), its MFMA(t) in B_t = (b_2t+1, b_2t+2); G1 one
// barrier later.  Loaders: pieces 0-7 of stage t+2 in A_t, 8-11 in B_t, into slot (t-1) % 3 - free since G1's reads of
// stage t-1 were waited for before b_2t; "stage t+1 has landed" (vmcnt) before b_2t+1, two barriers before its first reader.
// L2 prefetch (round 3, as in gemm_pp.hip's loader kernel): a consumer wave has no vector-memory operation of its own, so
// it can touch - one plain dword load per 128-byte line, never used - this tile's SHARE of the lines its XCD's tiles will
// stage `dist` steps later: the dY lines of a stage (64 rows x 2) are split between the k tiles that share the dY panel, the X
// lines (64 rows x 4) between the n tiles that run on the XCD at the same time.  pf_on: this lane has a line; pf_base / pf_ld:
// its address at stage 0 and the byte distance between stages.
struct WlPrefetch { const char* base; long long step; int on, dist; };

template <typename HT, int G, int ABL>
__device__ __forceinline__ void wl_consume(const char* lds, int M, int wave, int lane, f32x4_t (&acc)[4][4], const WlPrefetch& pf) {
  const int nk = (M + WP_M - 1) / WP_M;
  uint32_t pf_sink = 0;
  const bool partial = (M % WP_M) != 0;
  const int wk = wave & 3, gid = lane >> 4, p = lane & 15;
  // fragment offsets of MFMA tile 0 (rows r = 0, 1 of a fragment); tile i is at (offset ^ (i << 5)): the tile index sits in
  // bits 1-2 of the 16-byte chunk number, which the swizzle only XORs
  int xo[2], yo[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = 8 * gid + 4 * r + (p >> 2);
    xo[r] = tile_off<128>(row, (wk & 1) * 8 + ((p & 3) >> 1)) + (p & 1) * 8;
    yo[r] = tile_off<128>(row, G * 8 + ((p & 3) >> 1)) + (p & 1) * 8;
  }
  wp_barrier();                          // b0: stage 0 has landed
  if constexpr (G == 1) wp_barrier();
  vec8<HT> xf[2][4], yf[2][4];
  int slot = 0;
  for (int t = 0; t < nk; ++t) {
    const char* sY = lds + slot * WP_STAGE;
    const char* sX = sY + WP_SUB + (wk >> 1) * WP_SUB;
    if (pf.on && t + pf.dist < nk - 1) {   // (not the last stage: its rows may lie past M)
      const char* g = pf.base + (long long)(t + pf.dist) * pf.step;
      // "+v": the sink is one dedicated register until the wait after the loop (the data returns asynchronously)
      asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(g) : "memory");
    }
    if (!(ABL & 1) || t == 0) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int hb = half * (32 * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          xf[half][i] = cat8<HT>(tr_read<HT>(sX + (xo[0] ^ (i << 5)) + hb), tr_read<HT>(sX + (xo[1] ^ (i << 5)) + hb));
#pragma unroll
        for (int j = 0; j < 4; ++j)
          yf[half][j] = cat8<HT>(tr_read<HT>(sY + (yo[0] ^ (j << 5)) + hb), tr_read<HT>(sY + (yo[1] ^ (j << 5)) + hb));
      }
    }
    wp_wait_lds();
    if (partial && t == nk - 1) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int nvalid = M - (t * WP_M + half * 32 + 8 * gid);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (u >= nvalid) yf[half][j][u] = (HT)0.f;
      }
    }
    wp_barrier();
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!(ABL & 2)) acc[i][j] = mfma16x16<HT>(xf[half][i], yf[half][j], acc[i][j]);
          else if (i == 0 && j == 0) acc[0][0][0] += (float)xf[half][0][0] * (float)yf[half][0][0];
        }
    wp_barrier();
    slot = slot + 1 == WP_NST ? 0 : slot + 1;
  }
  if constexpr (G == 0) wp_barrier();
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink) :: "memory");   // the sink register is free for reuse only now
}

template <typename HT, int ABL, int NA = 2>   // NA: how many of a wave's three piece groups go out in window A_t (0 / 1 / 3: within noise of 2 or slower)
__device__ __forceinline__ void wl_load(const WpTile& a, int M, const char* lds, uint32_t lds0, int lw, int lane, bool do_bias,
                                        f32x4_t (&accb)[2]) {
  const int nk = (M + WP_M - 1) / WP_M;
  const bool partial = (M % WP_M) != 0;
  const int lrow = lane >> 4, lc = lane & 15;
  // this wave's 12 DMA pieces (1 KiB = 4 rows x 256 B) of a stage: pieces 12 lw .. 12 lw + 11 of its 48 (16 per sub-tile);
  // groups of four never straddle a sub-tile
  uint32_t off[12];
  int prow[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int pc = lw * 12 + i, sub = pc >> 4, row = (pc & 15) * 4 + lrow;
    const int c = lc ^ swz<128>(row);
    prow[i] = row;
    if (sub == 0) off[i] = (uint32_t)(((size_t)row * a.ldy + min(a.n0 + c * 8, ((a.N + 7) & ~7) - 8)) * 2);
    else off[i] = (uint32_t)(((size_t)row * a.ldx + min(a.k0 + (sub - 1) * 128 + c * 8, ((a.K + 7) & ~7) - 8)) * 2);
  }
  auto group = [&](int kt, int slot, int q) {   // pieces 4q .. 4q+3 of stage kt
    const int p0 = lw * 12 + 4 * q;
    const bool is_y = p0 < 16;
    const char* g = (is_y ? a.dY : a.X) + (size_t)kt * WP_M * (is_y ? a.ldy : a.ldx) * 2;
    const uint32_t dst = lds0 + slot * WP_STAGE + p0 * 1024;
    if (partial && kt == nk - 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int over = max(kt * WP_M + prow[4 * q + u] - (M - 1), 0);
        glds16(g + off[4 * q + u] - (size_t)over * (is_y ? a.ldy : a.ldx) * 2, dst + u * 1024);
      }
    } else {
      glds16_x4(uniform_ptr(g), off[4 * q], off[4 * q + 1], off[4 * q + 2], off[4 * q + 3], dst);
    }
  };
  group(0, 0, 0); group(0, 0, 1); group(0, 0, 2);
  if (nk > 1) {
    group(1, 1, 0); group(1, 1, 1); group(1, 1, 2);
    glds_wait<12>();
  } else {
    glds_wait<0>();
  }
  wp_barrier();   // b0

  const int gid = lane >> 4, p = lane & 15;
  int yo[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) yo[r] = tile_off<128>(8 * gid + 4 * r + (p >> 2), lw * 4 + ((p & 3) >> 1)) + (p & 1) * 8;
  vec8<HT> ones;
#pragma unroll
  for (int u = 0; u < 8; ++u) ones[u] = (HT)1.f;
  int slot = 0;
  for (int t = 0; t < nk; ++t) {
    const int nslot = slot >= 1 ? slot - 1 : WP_NST - 1;
    const bool more = t + 2 < nk && !(ABL & 4);
    // ---- window A_t
    if (more) {
#pragma unroll
      for (int q = 0; q < NA; ++q) group(t + 2, nslot, q);
      glds_wait<4 * NA>();
    } else {
      glds_wait<0>();
    }
    wp_barrier();   // b_2t+1
    // ---- window B_t
    if (more) {
#pragma unroll
      for (int q = NA; q < 3; ++q) group(t + 2, nslot, q);
    }
    if (do_bias) {   // column sums of this wave's two 16-column tiles of dY, stage t: all-ones A fragment (every row of D)
      const char* sY = lds + slot * WP_STAGE;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int hb = half * (32 * 256);
        vec8<HT> yf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          yf[j] = cat8<HT>(tr_read<HT>(sY + (yo[0] ^ (j << 5)) + hb), tr_read<HT>(sY + (yo[1] ^ (j << 5)) + hb));
        if (partial && t == nk - 1) {
          const int nvalid = M - (t * WP_M + half * 32 + 8 * gid);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (u >= nvalid) yf[j][u] = (HT)0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) accb[j] = mfma16x16<HT>(ones, yf[j], accb[j]);
      }
      wp_wait_lds();
    }
    wp_barrier();   // b_2t+2
    slot = slot + 1 == WP_NST ? 0 : slot + 1;
  }
  wp_barrier();     // b_2nk+1
}

template <typename HT, int ABL = 0, int NA = 2>
__global__ __launch_bounds__(768) void wgrad_ld_kernel(const WpGroup g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = xcd_remap_wp(blockIdx.x, gridDim.x);
  WpTile a;
  int tl = t, tiles_k = 1;
  float* dW = nullptr;
  float* db = nullptr;
#pragma unroll
  for (int i = 0; i < WP_MAX; ++i) {
    if (i < g.n && t >= g.tile0[i]) {
      a.dY = (const char*)g.dY[i]; a.X = (const char*)g.X[i]; a.ldy = g.ldy[i]; a.ldx = g.ldx[i]; a.N = g.N[i]; a.K = g.K[i];
      tl = t - g.tile0[i]; tiles_k = (g.K[i] + WP_TK - 1) / WP_TK;
      dW = g.dW[i]; db = g.db[i];
    }
  }
  a.n0 = (tl / tiles_k) * WP_TN; a.k0 = (tl % tiles_k) * WP_TK;
  const float alpha = g.out_scale ? *g.out_scale : 1.f;
  const int gid = lane >> 4, p = lane & 15;

  if (wave >= 8) {   // ---- loader waves
    const int lw = wave - 8;
    const bool do_bias = db != nullptr && a.k0 == 0;
    f32x4_t accb[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    wl_load<HT, ABL, NA>(a, g.M, lds, lds0, lw, lane, do_bias, accb);
    if (do_bias && gid == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = a.n0 + (lw * 2 + j) * 16 + p;
        if (n < a.N) { const float s = accb[j][0] * alpha; db[n] = g.accumulate ? db[n] + s : s; }
      }
    }
    return;
  }

  const int wn = wave >> 2, wk = wave & 3;
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  WlPrefetch pf{nullptr, 0, 0, g.pf_dist};
  if (g.pf_dist > 0) {
    const int sy = min(tiles_k, 8), sx = max(1, min(8, 32 / tiles_k));          // tiles sharing a dY panel / an X column block
    const int ny = (2 * WP_M + sy - 1) / sy, nx = (4 * WP_M + sx - 1) / sx;      // lines of a stage this tile touches
    const int y0 = ((tl % tiles_k) % sy) * ny, x0 = ((tl / tiles_k) % sx) * nx;
    const int li = wave * 64 + lane;
    if (li < ny && y0 + li < 2 * WP_M) {
      const int l = y0 + li, row = l >> 1, col = a.n0 + (l & 1) * 64;
      if (col < a.N) { pf.base = a.dY + ((size_t)row * a.ldy + col) * 2; pf.step = (long long)WP_M * a.ldy * 2; pf.on = 1; }
    } else if (li >= ny && li - ny < nx && x0 + li - ny < 4 * WP_M) {
      const int l = x0 + li - ny, row = l >> 2, col = a.k0 + (l & 3) * 64;
      if (col < a.K) { pf.base = a.X + ((size_t)row * a.ldx + col) * 2; pf.step = (long long)WP_M * a.ldx * 2; pf.on = 1; }
    }
  }
  if (wn == 0) wl_consume<HT, 0, ABL>(lds, g.M, wave, lane, acc, pf);
  else wl_consume<HT, 1, ABL>(lds, g.M, wave, lane, acc, pf);

  const int N = a.N, K = a.K;
  float chk = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = a.n0 + wn * 64 + j * 16 + p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = a.k0 + wk * 64 + i * 16 + 4 * gid;
      if (n < N && k < K) {
        float v[4] = {acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha};
        nf_note(chk, v[0]); nf_note(chk, v[1]); nf_note(chk, v[2]); nf_note(chk, v[3]);
        float* dst = dW + (size_t)n * K + k;
        if (k + 3 < K && ((((size_t)n * K + k) & 3) == 0)) {
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (g.accumulate) { const float4 c = *reinterpret_cast<const float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
          // streaming store: the weight gradients (33 MB per layer) are read next by the optimizer / the gradient exchange, not
          // by this step - as plain stores they displace 200 MB of what the step still reads from the Infinity Cache (in the
          // step 5.452 -> 5.415 ms, three interleaved runs)
          { typedef float f4_t __attribute__((ext_vector_type(4))); f4_t o_ = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(o_, reinterpret_cast<f4_t*>(dst)); }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k + u < K) dst[u] = g.accumulate ? dst[u] + v[u] : v[u];
        }
      }
    }
  }
  nf_commit(g.out_scale, chk);   // (the loader waves' bias sums come from the same dY rows: what overflows there overflows here)
}

template <typename HT, int ABL = 0, int MODE = 2>
__global__ __launch_bounds__(512) void wgrad_pp_kernel(const WpGroup g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = xcd_remap_wp(blockIdx.x, gridDim.x);
  WpTile a;
  int tl = t, tiles_k = 1;
  float* dW = nullptr;
  float* db = nullptr;
#pragma unroll
  for (int i = 0; i < WP_MAX; ++i) {   // static indexing of the kernel-argument arrays (uniform select)
    if (i < g.n && t >= g.tile0[i]) {
      a.dY = (const char*)g.dY[i]; a.X = (const char*)g.X[i]; a.ldy = g.ldy[i]; a.ldx = g.ldx[i]; a.N = g.N[i]; a.K = g.K[i];
      tl = t - g.tile0[i]; tiles_k = (g.K[i] + WP_TK - 1) / WP_TK;
      dW = g.dW[i]; db = g.db[i];
    }
  }
  a.n0 = (tl / tiles_k) * WP_TN; a.k0 = (tl % tiles_k) * WP_TK;   // k-tile fastest: consecutive blocks share the dY panel
  const int wn = wave >> 2, wk = wave & 3;
  const bool do_bias = db != nullptr && a.k0 == 0 && wk == 0;

  // transposing-read offsets inside a [64][128] sub-tile: MFMA tile = 16 columns cb; lane (gid, p) addresses row
  // 8 gid + 4 r + (p >> 2) (r = 0, 1: the two reads of a fragment), the 4 columns 4 (p & 3) .. + 3 of the tile
  const int gid = lane >> 4, p = lane & 15;
  int xoff[4][2], yoff[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = 8 * gid + 4 * r + (p >> 2);
      const int cx = ((wk & 1) * 4 + i) * 2 + ((p & 3) >> 1), cy = (wn * 4 + i) * 2 + ((p & 3) >> 1);
      xoff[i][r] = tile_off<128>(row, cx) + (p & 1) * 8;
      yoff[i][r] = tile_off<128>(row, cy) + (p & 1) * 8;
    }
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t accb[4] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};

  if constexpr (MODE == 1) {
    wp_mainloop_free<HT, ABL>(a, g.M, lds, lds0, wave, lane, xoff, yoff, acc, accb, do_bias);
  } else if constexpr (MODE == 2) {
    if (wn == 0) wp_mainloop_merged<HT, 0, ABL>(a, g.M, lds, lds0, wave, lane, xoff, yoff, acc, accb, do_bias);
    else wp_mainloop_merged<HT, 1, ABL>(a, g.M, lds, lds0, wave, lane, xoff, yoff, acc, accb, do_bias);
  } else {
    if (wn == 0) wp_mainloop<HT, 0, ABL>(a, g.M, lds, lds0, wave, lane, xoff, yoff, acc, accb, do_bias);
    else wp_mainloop<HT, 1, ABL>(a, g.M, lds, lds0, wave, lane, xoff, yoff, acc, accb, do_bias);
  }

  // D[row = k: 4 gid + r][col = n: p] per MFMA tile (i: k tile, j: n tile): a lane owns 4 consecutive k of one row n of dW
  const float alpha = g.out_scale ? *g.out_scale : 1.f;
  const int N = a.N, K = a.K;
  float chk = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = a.n0 + wn * 64 + j * 16 + p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = a.k0 + wk * 64 + i * 16 + 4 * gid;
      if (n < N && k < K) {
        float v[4] = {acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha};
        nf_note(chk, v[0]); nf_note(chk, v[1]); nf_note(chk, v[2]); nf_note(chk, v[3]);
        float* dst = dW + (size_t)n * K + k;
        if (k + 3 < K && ((((size_t)n * K + k) & 3) == 0)) {
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (g.accumulate) { const float4 c = *reinterpret_cast<const float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
          // streaming store: the weight gradients (33 MB per layer) are read next by the optimizer / the gradient exchange, not
          // by this step - as plain stores they displace 200 MB of what the step still reads from the Infinity Cache (in the
          // step 5.452 -> 5.415 ms, three interleaved runs)
          { typedef float f4_t __attribute__((ext_vector_type(4))); f4_t o_ = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(o_, reinterpret_cast<f4_t*>(dst)); }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k + u < K) dst[u] = g.accumulate ? dst[u] + v[u] : v[u];
        }
      }
    }
    if (do_bias && gid == 0 && n < N) {   // every row of the ones-product holds the column sums: take row 0
      const float s = accb[j][0] * alpha;
      db[n] = g.accumulate ? db[n] + s : s;
    }
  }
  nf_commit(g.out_scale, chk);
}


// ---- eight-phase form (wgrad_p8_kernel, round 6): 256 (n) x 256 (k) tiles ---------------------------------------------------------
// The 128 x 256 tile above multiplies with 64 x 64 wave tiles: one transposing fragment read per MFMA, and with the LDS-DMA
// writes the LDS is busier than the matrix pipes (1.04 - 1.09 PFLOP/s; the NT kernels with 80 / 128 / 160 x 64 wave tiles reach
// 1.31 / 1.36 / 1.44 at this contraction length - profiles/r06_n_p8_longk.txt).  Here the schedule of gemm_pp.hip's eight-phase
// kernel on the weight-gradient operands:
//   * 8 waves = 2 (n) x 4 (k), wave tile n 128 x k 64 (8 x 4 MFMA tiles, 128 accumulator registers, 0.75 reads per MFMA), no
//     loader waves; every block runs the whole M.  A layer is 128 such tiles, so the caller hands over TWO layers per launch
//     (timhip_layer_bwd_weights_pair: 256 tiles = one per CU);
//   * a 64-row contraction step is four quadrant phases (n half, k half) = (0,0) (0,1) (1,1) (1,0), 16 MFMAs each; the stage is
//     FOUR sub-tiles [64 m][128 columns] in wgrad_pp's format, one per quadrant operand, so that each can be restaged on its
//     own: Y_q = the n columns {wr 128 + q 64 + c} of both wave rows, X_q = the k columns {wc 64 + q 32 + c} of the four wave
//     columns (the LDS-DMA source side does the gather: 128- / 64-byte runs);
//   * two stage buffers (128 KiB); restaging one phase after a sub-tile's last fragment read, as in p8_mainloop: phase 1: X_0 of
//     step t + 1 -> other buffer; phases 2 / 3 / 4: Y_0 / X_1 / Y_1 of step t + 2 -> this buffer; one counted wait per step.
// Hazards: gemm_pp.hip (identical segment structure).  Bias gradients: one ones-MFMA per wave and step, see the kernel.
constexpr int W8_T = 256, W8_SUB = WP_M * 256, W8_STAGE = 4 * W8_SUB;
struct W8Tab { uint32_t offY[2][2], offX[2][2]; };

// ABL (tuning builds, tools/wg_pair_ab.py): timing-only ablations - 1: no bias MFMAs, 2: the X pieces gathered in 128-byte runs
// like the Y pieces (wrong columns), 4: no stores of the result, 8: fragment reads only in the first step, 16: no DMA after the prologue
// SCH (tuning builds, schedule arms): 1: the phase-1 pieces (X_0 of step t + 1) go out in phase 2 with Y_0's - phase 1 has 26 of the
// step's 58 fragment reads; 2: no s_setprio around the MFMA segments; 4: a phase's pieces before its fragment reads; 8: fragment reads
// waited for behind the phase's first barrier
template <typename HT, int G, int ABL = 0, int SCH = 0, bool RAG = false>
__device__ __forceinline__ void w8_mainloop(const char* dY, const char* X, size_t sy, size_t sx, int nk, const char* lds, uint32_t lds0,
                                            const W8Tab& tb, int wave, const int (&yo)[2], const int (&xo)[2], bool bias_on,
                                            const int (&bo)[2], f32x4_t (&acc)[4][8], f32x4_t& accb, int mlast) {
  const uint32_t pw = (uint32_t)(2 * wave) * 1024u;
  // (mlast: rows of the last stage, 64 = whole; a partial last stage re-reads row M - 1 for its rows past the end - finite data -
  //  and the dY fragments of those rows are zeroed before they are multiplied: zero_tail)
  const int lrow_ = (int)(threadIdx.x & 63) >> 4;
  auto over_of = [&](int i) { return max((2 * wave + i) * 4 + lrow_ - (mlast - 1), 0); };
  auto stage_y = [&](int kt, uint32_t buf, int q) {
    const void* g = uniform_ptr(dY + (size_t)kt * sy);
    if (RAG && mlast < WP_M && kt == nk - 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16_s(g, tb.offY[q][i] - (uint32_t)over_of(i) * (uint32_t)(sy / WP_M), buf + q * W8_SUB + pw + i * 1024);
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_s(g, tb.offY[q][i], buf + q * W8_SUB + pw + i * 1024);
  };
  auto stage_x = [&](int kt, uint32_t buf, int q) {
    const void* g = uniform_ptr(X + (size_t)kt * sx);
    if (RAG && mlast < WP_M && kt == nk - 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16_s(g, tb.offX[q][i] - (uint32_t)over_of(i) * (uint32_t)(sx / WP_M), buf + (2 + q) * W8_SUB + pw + i * 1024);
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_s(g, tb.offX[q][i], buf + (2 + q) * W8_SUB + pw + i * 1024);
  };
  stage_y(0, lds0, 0); stage_x(0, lds0, 0); stage_x(0, lds0, 1); stage_y(0, lds0, 1);
  if (nk > 1) { stage_y(1, lds0 + W8_STAGE, 0); stage_x(1, lds0 + W8_STAGE, 0); stage_x(1, lds0 + W8_STAGE, 1); stage_y(1, lds0 + W8_STAGE, 1); }
  glds_wait<0>();
  wp_barrier();
  if constexpr (G == 1) wp_barrier();   // one segment behind group 0 from here on

  vec8<HT> yf[4][2], xf[2][2], ones, yb;
#pragma unroll
  for (int u = 0; u < 8; ++u) { ones[u] = (HT)1.f; yb[u] = (HT)0.f; }
  auto read_y = [&](const char* bb, int q) {
    const char* sY = bb + q * W8_SUB;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        yf[j][kh] = cat8<HT>(tr_read<HT>(sY + (yo[0] ^ (j << 5)) + kh * (32 * 256)), tr_read<HT>(sY + (yo[1] ^ (j << 5)) + kh * (32 * 256)));
  };
  auto read_x = [&](const char* bb, int q) {
    const char* sX = bb + (2 + q) * W8_SUB;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        xf[i][kh] = cat8<HT>(tr_read<HT>(sX + (xo[0] ^ (i << 5)) + kh * (32 * 256)), tr_read<HT>(sX + (xo[1] ^ (i << 5)) + kh * (32 * 256)));
  };
  auto mma = [&](auto qn_c, auto qk_c) {
    constexpr int qn = decltype(qn_c)::value, qk = decltype(qk_c)::value;
    if constexpr (SCH & 8) { wp_wait_lds(); __builtin_amdgcn_sched_barrier(0); }
    if constexpr (!(SCH & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[2 * qk + i][4 * qn + j] = mfma16x16<HT>(xf[i][kh], yf[j][kh], acc[2 * qk + i][4 * qn + j]);
    if constexpr (qn == 0 && qk == 0) {   // phase 1: this wave's share of the bias gradient (one 32-row half of one dY tile)
      if (!(ABL & 1) && bias_on) accb = mfma16x16<HT>(ones, yb, accb);
    }
    if constexpr (!(SCH & 2)) __builtin_amdgcn_s_setprio(0);
  };
  auto wait_reads = [&]() { if constexpr (!(SCH & 8)) wp_wait_lds(); };
  const int gid_ = (int)(threadIdx.x & 63) >> 4;
  auto zero_tail = [&](int t, bool with_bias, int bkh) {   // last, partial stage: the dY rows >= M contribute nothing
    if (!(RAG && mlast < WP_M && t == nk - 1)) return;
    wp_wait_lds();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int nvalid = mlast - (kh * 32 + 8 * gid_);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (u >= nvalid) yf[j][kh][u] = (HT)0.f;
    }
    if (with_bias) {
      const int nvalid = mlast - (bkh * 32 + 8 * gid_);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u >= nvalid) yb[u] = (HT)0.f;
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  for (int t = 0; t < nk; ++t) {
    const int b = t & 1;
    const char* bb = lds + b * W8_STAGE;
    const uint32_t cur = lds0 + b * W8_STAGE, oth = lds0 + (b ^ 1) * W8_STAGE;
    const bool s1 = t >= 1 && t + 1 < nk && !(ABL & 16), s2 = t + 2 < nk && !(ABL & 16);
    const bool rd = !(ABL & 8) || t == 0;
    // phase 1: quadrant (n 0, k 0)
    if constexpr (SCH & 4) { if (s1 && !(SCH & 1)) stage_x(t + 1, oth, 0); }
    if (rd) { read_x(bb, 0); read_y(bb, 0); }
    if (bias_on) yb = cat8<HT>(tr_read<HT>(bb + bo[0]), tr_read<HT>(bb + bo[1]));
    if constexpr (!(SCH & 4)) { if (s1 && !(SCH & 1)) stage_x(t + 1, oth, 0); }
    wait_reads(); zero_tail(t, bias_on, wave & 1); wp_barrier();
    mma(I0{}, I0{});
    wp_barrier();
    // phase 2: (n 0, k 1) - Y fragments kept
    if constexpr (SCH & 4) { if (s1 && (SCH & 1)) stage_x(t + 1, oth, 0); if (s2) stage_y(t + 2, cur, 0); }
    if (rd) read_x(bb, 1);
    if constexpr (!(SCH & 4)) { if (s1 && (SCH & 1)) stage_x(t + 1, oth, 0); if (s2) stage_y(t + 2, cur, 0); }
    wait_reads(); wp_barrier();
    mma(I0{}, I1{});
    wp_barrier();
    // phase 3: (n 1, k 1) - X fragments kept
    if constexpr (SCH & 4) { if (s2) stage_x(t + 2, cur, 1); }
    if (rd) read_y(bb, 1);
    if constexpr (!(SCH & 4)) { if (s2) stage_x(t + 2, cur, 1); }
    wait_reads(); zero_tail(t, false, 0); wp_barrier();
    mma(I1{}, I1{});
    wp_barrier();
    // phase 4: (n 1, k 0) - Y fragments kept; the step's one counted wait: X_0 of step t + 1 (and everything older) has landed
    if constexpr (SCH & 4) { if (s2) stage_y(t + 2, cur, 1); }
    if (rd) read_x(bb, 0);
    if constexpr (!(SCH & 4)) { if (s2) stage_y(t + 2, cur, 1); }
    if (s2) glds_wait<6>(); else glds_wait<0>();
    wait_reads(); wp_barrier();
    mma(I1{}, I0{});
    wp_barrier();
  }
  if constexpr (G == 0) wp_barrier();   // as many barriers as group 1
}

// Two phases per step (the default since it was measured: two layers 302.8 -> 286.3 us, the C2a step -1.0 %,
// profiles/r06_ae_*; TIMHIP_WGRAD_P8_PH=4: the four-phase loop above): phase A = n-half 0 against BOTH k-halves (32 MFMAs: Y_0, X_0, X_1 read - 32
// transposing reads - X kept), phase B = n-half 1 (32 MFMAs: Y_1 read).  Half the barriers per MFMA and no second read of X_0 (48
// instead of 56 reads per step) for 16 fragment registers more.  Restaging, one phase after a sub-tile's last read: phase A of step
// t: Y_1 of step t + 1 -> other buffer; phase B: Y_0, X_0, X_1 of step t + 2 -> this buffer, then the step's counted wait (the six
// newest pieces stay in flight).  Hazards as in the four-phase loop (segments 2p / 2p + 1, group 1 one segment behind).
template <typename HT, int G, bool RAG = false>
__device__ __forceinline__ void w8_mainloop2(const char* dY, const char* X, size_t sy, size_t sx, int nk, const char* lds, uint32_t lds0,
                                             const W8Tab& tb, int wave, const int (&yo)[2], const int (&xo)[2], bool bias_on,
                                             const int (&bo)[2], f32x4_t (&acc)[4][8], f32x4_t& accb, int mlast) {
  const uint32_t pw = (uint32_t)(2 * wave) * 1024u;
  // (mlast: rows of the last stage, 64 = whole; a partial last stage re-reads row M - 1 for its rows past the end - finite data -
  //  and the dY fragments of those rows are zeroed before they are multiplied: zero_tail)
  const int lrow_ = (int)(threadIdx.x & 63) >> 4;
  auto over_of = [&](int i) { return max((2 * wave + i) * 4 + lrow_ - (mlast - 1), 0); };
  auto stage_y = [&](int kt, uint32_t buf, int q) {
    const void* g = uniform_ptr(dY + (size_t)kt * sy);
    if (RAG && mlast < WP_M && kt == nk - 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16_s(g, tb.offY[q][i] - (uint32_t)over_of(i) * (uint32_t)(sy / WP_M), buf + q * W8_SUB + pw + i * 1024);
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_s(g, tb.offY[q][i], buf + q * W8_SUB + pw + i * 1024);
  };
  auto stage_x = [&](int kt, uint32_t buf, int q) {
    const void* g = uniform_ptr(X + (size_t)kt * sx);
    if (RAG && mlast < WP_M && kt == nk - 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16_s(g, tb.offX[q][i] - (uint32_t)over_of(i) * (uint32_t)(sx / WP_M), buf + (2 + q) * W8_SUB + pw + i * 1024);
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_s(g, tb.offX[q][i], buf + (2 + q) * W8_SUB + pw + i * 1024);
  };
  stage_y(0, lds0, 0); stage_x(0, lds0, 0); stage_x(0, lds0, 1); stage_y(0, lds0, 1);
  if (nk > 1) { stage_y(1, lds0 + W8_STAGE, 0); stage_x(1, lds0 + W8_STAGE, 0); stage_x(1, lds0 + W8_STAGE, 1); stage_y(1, lds0 + W8_STAGE, 1); }
  glds_wait<0>();
  wp_barrier();
  if constexpr (G == 1) wp_barrier();   // one segment behind group 0 from here on

  vec8<HT> yf[4][2], xf[4][2], ones, yb;
#pragma unroll
  for (int u = 0; u < 8; ++u) { ones[u] = (HT)1.f; yb[u] = (HT)0.f; }
  auto read_y = [&](const char* bb, int q) {
    const char* sY = bb + q * W8_SUB;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        yf[j][kh] = cat8<HT>(tr_read<HT>(sY + (yo[0] ^ (j << 5)) + kh * (32 * 256)), tr_read<HT>(sY + (yo[1] ^ (j << 5)) + kh * (32 * 256)));
  };
  auto read_x = [&](const char* bb) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const char* sX = bb + (2 + q) * W8_SUB;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          xf[2 * q + i][kh] = cat8<HT>(tr_read<HT>(sX + (xo[0] ^ (i << 5)) + kh * (32 * 256)), tr_read<HT>(sX + (xo[1] ^ (i << 5)) + kh * (32 * 256)));
    }
  };
  auto mma = [&](auto qn_c) {
    constexpr int qn = decltype(qn_c)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][4 * qn + j] = mfma16x16<HT>(xf[i][kh], yf[j][kh], acc[i][4 * qn + j]);
    if constexpr (qn == 0) {   // this wave's share of the bias gradient (one 32-row half of one dY tile)
      if (bias_on) accb = mfma16x16<HT>(ones, yb, accb);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  const int gid_ = (int)(threadIdx.x & 63) >> 4;
  auto zero_tail = [&](int t, bool with_bias, int bkh) {   // last, partial stage: the dY rows >= M contribute nothing
    if (!(RAG && mlast < WP_M && t == nk - 1)) return;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int nvalid = mlast - (kh * 32 + 8 * gid_);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (u >= nvalid) yf[j][kh][u] = (HT)0.f;
    }
    if (with_bias) {
      const int nvalid = mlast - (bkh * 32 + 8 * gid_);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u >= nvalid) yb[u] = (HT)0.f;
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  for (int t = 0; t < nk; ++t) {
    const int b = t & 1;
    const char* bb = lds + b * W8_STAGE;
    const uint32_t cur = lds0 + b * W8_STAGE, oth = lds0 + (b ^ 1) * W8_STAGE;
    const bool s1 = t >= 1 && t + 1 < nk, s2 = t + 2 < nk;
    // phase A: n-half 0, both k-halves
    read_x(bb); read_y(bb, 0);
    if (bias_on) yb = cat8<HT>(tr_read<HT>(bb + bo[0]), tr_read<HT>(bb + bo[1]));
    if (s1) stage_y(t + 1, oth, 1);
    wp_wait_lds(); zero_tail(t, bias_on, wave & 1); wp_barrier();
    mma(I0{});
    wp_barrier();
    // phase B: n-half 1 - X fragments kept; the step's one counted wait: Y_1 of step t + 1 (and everything older) has landed
    read_y(bb, 1);
    if (s2) { stage_y(t + 2, cur, 0); stage_x(t + 2, cur, 0); stage_x(t + 2, cur, 1); glds_wait<6>(); } else { glds_wait<0>(); }
    wp_wait_lds(); zero_tail(t, false, 0); wp_barrier();
    mma(I1{});
    wp_barrier();
  }
  if constexpr (G == 0) wp_barrier();   // as many barriers as group 1
}

// RAG: the contraction length is not a multiple of 64 (an instance of its own: the ragged-stage branches cost the whole-stage
// launches 2 - 5 % when they were run-time conditions - profiles/r06_al_wgrad_pair_ab.txt)
template <typename HT, int ABL = 0, int SCH = 0, bool PH2 = false, bool RAG = false>
__global__ __launch_bounds__(512) void wgrad_p8_kernel(const WpGroup g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [2][Y_0 | Y_1 | X_0 | X_1], 16 KiB each
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = xcd_remap_wp(blockIdx.x, gridDim.x);
  const char* pY = nullptr; const char* pX = nullptr;
  int ldy = 0, ldx = 0, N = 0, K = 0, tl = t, tiles_k = 1;
  float* dW = nullptr;
  float* db = nullptr;
#pragma unroll
  for (int i = 0; i < WP_MAX; ++i) {   // static indexing of the kernel-argument arrays (uniform select)
    if (i < g.n && t >= g.tile0[i]) {
      pY = (const char*)g.dY[i]; pX = (const char*)g.X[i]; ldy = g.ldy[i]; ldx = g.ldx[i]; N = g.N[i]; K = g.K[i];
      tl = t - g.tile0[i]; tiles_k = g.K[i] / W8_T;
      dW = g.dW[i]; db = g.db[i];
    }
  }
  const int n0 = (tl / tiles_k) * W8_T, k0 = (tl % tiles_k) * W8_T;   // k-tile fastest: consecutive blocks share the dY panel
  const int wr = wave >> 2, wc = wave & 3;
  const int gid = lane >> 4, p = lane & 15;
  // this wave's two LDS-DMA pieces (1 KiB = 4 rows x 256 B) of every sub-tile: rows 8 wave .. 8 wave + 7; lane -> (row, 16-byte
  // chunk), the chunk swizzled and the columns gathered on the SOURCE side
  W8Tab tb;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (2 * wave + i) * 4 + gid, c = p ^ swz<128>(row);
      tb.offY[q][i] = (uint32_t)(((size_t)row * ldy + n0 + (c >> 3) * 128 + q * 64 + (c & 7) * 8) * 2);
      tb.offX[q][i] = (ABL & 2) ? (uint32_t)(((size_t)row * ldx + k0 + (c >> 3) * 128 + q * 64 + (c & 7) * 8) * 2)
                                : (uint32_t)(((size_t)row * ldx + k0 + (c >> 2) * 64 + q * 32 + (c & 3) * 8) * 2);
    }
  // transposing reads: lane (gid, p) addresses row 8 gid + 4 r + (p >> 2) (r = 0, 1: the two reads of a fragment), the 4 columns
  // 4 (p & 3) .. + 3 of MFMA tile 0 of the wave's columns in the sub-tile; tile i is at (offset ^ (i << 5))
  int yo[2], xo[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = 8 * gid + 4 * r + (p >> 2);
    yo[r] = tile_off<128>(row, wr * 8 + ((p & 3) >> 1)) + (p & 1) * 8;
    xo[r] = tile_off<128>(row, wc * 4 + ((p & 3) >> 1)) + (p & 1) * 8;
  }
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // bias gradient: the 16 dY tiles of 16 columns this n-row of tiles has in LDS are split between its tiles_k blocks (block kt:
  // tiles [16 kt / tiles_k, 16 (kt + 1) / tiles_k)), and a block's tiles x two 32-row halves between its waves - one ones-MFMA per
  // wave and step in phase 1 (both Y sub-tiles of the step are resident there), the two halves added through LDS at the end in a
  // fixed order.  (As two tiles per wave of the k0 = 0 blocks only, those blocks - and with them the launch - ran 6 % longer.)
  f32x4_t accb = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int kt = tl % tiles_k, bc0 = 16 * kt / tiles_k, bc1 = 16 * (kt + 1) / tiles_k;
  const int bcol = bc0 + (wave >> 1), bkh = wave & 1;
  const bool bias_on = db != nullptr && bcol < bc1;
  int bo[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = 8 * gid + 4 * r + (p >> 2);
    bo[r] = ((bcol >> 2) & 1) * W8_SUB + bkh * (32 * 256) + tile_off<128>(row, (bcol >> 3) * 8 + (bcol & 3) * 2 + ((p & 3) >> 1)) + (p & 1) * 8;
  }
  const int nk = (g.M + WP_M - 1) / WP_M, mlast = g.M - (nk - 1) * WP_M;   // (rows of the last stage: 1 .. 64)
  const size_t sy = (size_t)WP_M * ldy * 2, sx = (size_t)WP_M * ldx * 2;
  if constexpr (PH2) {
    if (wr == 0) w8_mainloop2<HT, 0, RAG>(pY, pX, sy, sx, nk, lds, lds0, tb, wave, yo, xo, bias_on, bo, acc, accb, mlast);
    else w8_mainloop2<HT, 1, RAG>(pY, pX, sy, sx, nk, lds, lds0, tb, wave, yo, xo, bias_on, bo, acc, accb, mlast);
  } else {
    if (wr == 0) w8_mainloop<HT, 0, ABL, SCH, RAG>(pY, pX, sy, sx, nk, lds, lds0, tb, wave, yo, xo, bias_on, bo, acc, accb, mlast);
    else w8_mainloop<HT, 1, ABL, SCH, RAG>(pY, pX, sy, sx, nk, lds, lds0, tb, wave, yo, xo, bias_on, bo, acc, accb, mlast);
  }

  // D[row = k: 4 gid + r][col = n: p] per MFMA tile (i: k tile, j: n tile): a lane owns 4 consecutive k of one row n of dW
  const float alpha = g.out_scale ? *g.out_scale : 1.f;
  float chk = 0.f;
  typedef float f4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = n0 + wr * 128 + j * 16 + p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + wc * 64 + i * 16 + 4 * gid;
      f4_t o = {acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha};
      nf_note(chk, o[0]); nf_note(chk, o[1]); nf_note(chk, o[2]); nf_note(chk, o[3]);
      f4_t* dst = reinterpret_cast<f4_t*>(dW + (size_t)n * K + k);
      if (g.accumulate) o += *dst;
      if (!(ABL & 4) || o[0] == 12345.678f) __builtin_nontemporal_store(o, dst);   // (streaming: see wgrad_ld_kernel)
    }
  }
  if (db != nullptr) {   // (block-uniform) every row of the ones-product holds the column sums: take row 0; halves added in LDS
    __syncthreads();     // every wave is done with the stage buffers
    float* red = reinterpret_cast<float*>(lds);
    if (bias_on && gid == 0) red[wave * 16 + p] = accb[0];
    __syncthreads();
    if (bias_on && gid == 0 && bkh == 0) {
      const int n = n0 + bcol * 16 + p;
      const float s = (red[wave * 16 + p] + red[(wave + 1) * 16 + p]) * alpha;
      db[n] = g.accumulate ? db[n] + s : s;
    }
  }
  nf_commit(g.out_scale, chk);
  (void)N;
}

}  // namespace

// Does the ping-pong grid suit this group?  Its blocks run the whole contraction (no split), one per CU: it needs about a
// multiple of 256 tiles of 128 x 256 and a long M (the encoder layers of a production batch).
bool tim_wgrad_pp_wins(const TimWgradItem* it, int n, int M) {
  if (!it || n < 1 || n > WP_MAX || M < 2048) return false;
  long long tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (it[i].Nout < 64 || it[i].Kout < 128) return false;
    tiles += (long long)((it[i].Nout + WP_TN - 1) / WP_TN) * ((it[i].Kout + WP_TK - 1) / WP_TK);
  }
  const long long rounds = (tiles + 255) / 256;
  return tiles >= 192 && tiles * 100 >= rounds * 256 * 75;
}

int tim_wgrad_group_pp(int precision, const TimWgradItem* it, int n, int M, int accumulate, const float* out_scale, hipStream_t s) {
  if (!h16_storage(precision)) return TIMHIP_EUNSUPPORTED;
  WpGroup g;
  g.n = n; g.M = M; g.accumulate = accumulate ? 1 : 0; g.out_scale = out_scale;
  g.pf_dist = tim_knobs().wgrad_pf;
  g.tile0[0] = 0;
  for (int i = 0; i < WP_MAX; ++i) {
    if (i >= n) {
      g.dY[i] = g.X[i] = nullptr; g.dW[i] = g.db[i] = nullptr; g.ldy[i] = g.ldx[i] = g.N[i] = g.K[i] = 0; g.tile0[i + 1] = g.tile0[i];
      continue;
    }
    const TimWgradItem& t = it[i];
    if (!t.dY || !t.X || !t.dW || t.Nout <= 0 || t.Kout <= 0) return TIMHIP_EINVAL;
    if ((t.ldy % 8) || (t.ldx % 8) || (((uintptr_t)t.dY | (uintptr_t)t.X | (uintptr_t)t.dW) & 15)) return TIMHIP_EALIGN;
    if ((size_t)M * t.ldy * 2 >= (1ull << 32) || (size_t)M * t.ldx * 2 >= (1ull << 32)) return TIMHIP_EUNSUPPORTED;
    g.dY[i] = t.dY; g.X[i] = t.X; g.dW[i] = t.dW; g.db[i] = t.db; g.ldy[i] = t.ldy; g.ldx[i] = t.ldx; g.N[i] = t.Nout; g.K[i] = t.Kout;
    g.tile0[i + 1] = g.tile0[i] + ((t.Nout + WP_TN - 1) / WP_TN) * ((t.Kout + WP_TK - 1) / WP_TK);
  }
  const size_t shmem = (size_t)WP_NST * WP_STAGE;
  static PerDeviceOnce attr_set[2];
  const int hi = precision == TIMHIP_PREC_F16 ? 1 : 0;
  if (attr_set[hi].first()) {
    DISPATCH_H16(precision, (void)hipFuncSetAttribute((const void*)wgrad_pp_kernel<HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    DISPATCH_H16(precision, (void)hipFuncSetAttribute((const void*)wgrad_ld_kernel<HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  }
  // the 12-wave loader form is the default; TIMHIP_WGRAD_LD=0 selects the 8-wave merged-phase kernel (A/B switch)
  const bool ld_on = tim_knobs().wgrad_ld != 0;
#ifdef TIMHIP_TUNING
  if (const char* v = getenv("TIMHIP_WGPP_ABL")) {
    const int abl = atoi(v);
    const int mode = getenv("TIMHIP_WGPP_MODE") ? atoi(getenv("TIMHIP_WGPP_MODE")) : 4;
#define WABL(X, MD) case X + 8 * MD: (void)hipFuncSetAttribute((const void*)wgrad_pp_kernel<f16_t, X, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL((wgrad_pp_kernel<f16_t, X, MD>), dim3((unsigned)g.tile0[n]), dim3(512), shmem, s, g); return TIMHIP_OK;
#define WLD(X) case X + 8 * 4: (void)hipFuncSetAttribute((const void*)wgrad_ld_kernel<f16_t, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL((wgrad_ld_kernel<f16_t, X>), dim3((unsigned)g.tile0[n]), dim3(768), shmem, s, g); return TIMHIP_OK;
    switch (abl + 8 * mode) { WLD(0) WLD(1) WLD(2) WLD(4) WLD(6) WABL(0, 0) WABL(1, 0) WABL(2, 0) WABL(4, 0) WABL(6, 0) WABL(7, 0) WABL(0, 1) WABL(2, 1) WABL(4, 1) WABL(6, 1) WABL(0, 2) WABL(1, 2) WABL(2, 2) WABL(4, 2) WABL(6, 2) default: break; }
#undef WABL
#undef WLD
  }
#endif
  if (ld_on) {
    DISPATCH_H16(precision, hipLaunchKernelGGL(wgrad_ld_kernel<HT>, dim3((unsigned)g.tile0[n]), dim3(768), shmem, s, g));
  } else {
    DISPATCH_H16(precision, hipLaunchKernelGGL(wgrad_pp_kernel<HT>, dim3((unsigned)g.tile0[n]), dim3(512), shmem, s, g));
  }
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}

// Does the eight-phase grid suit this group?  Whole 256 x 256 tiles (any contraction length: a ragged last stage is handled) and rounds of 256
// tiles filled to 90 %: at C2a that is the eight gradients of TWO encoder layers (TIMHIP_WGRAD_P8=0: off).
bool tim_wgrad_p8_wins(const TimWgradItem* it, int n, int M) {
  if (tim_knobs().wgrad_p8 == 0 || !it || n < 1 || n > WP_MAX || M < 2048) return false;
  long long tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (it[i].Nout % W8_T || it[i].Kout % W8_T || it[i].Nout <= 0 || it[i].Kout < 4 * W8_T) return false;   // (>= 4 k tiles: one bias MFMA per wave)
    tiles += (long long)(it[i].Nout / W8_T) * (it[i].Kout / W8_T);
  }
  const long long rounds = (tiles + 255) / 256;
  return tiles * 100 >= rounds * 256 * 90;
}

int tim_wgrad_group_p8(int precision, const TimWgradItem* it, int n, int M, int accumulate, const float* out_scale, hipStream_t s) {
  if (!h16_storage(precision)) return TIMHIP_EUNSUPPORTED;
  if (!it || n < 1 || n > WP_MAX || M < 2 * WP_M) return TIMHIP_EINVAL;
  WpGroup g;
  g.n = n; g.M = M; g.accumulate = accumulate ? 1 : 0; g.out_scale = out_scale; g.pf_dist = 0;
  g.tile0[0] = 0;
  for (int i = 0; i < WP_MAX; ++i) {
    if (i >= n) {
      g.dY[i] = g.X[i] = nullptr; g.dW[i] = g.db[i] = nullptr; g.ldy[i] = g.ldx[i] = g.N[i] = g.K[i] = 0; g.tile0[i + 1] = g.tile0[i];
      continue;
    }
    const TimWgradItem& t = it[i];
    if (!t.dY || !t.X || !t.dW || t.Nout <= 0 || t.Kout < 4 * W8_T || t.Nout % W8_T || t.Kout % W8_T) return TIMHIP_EINVAL;
    if ((t.ldy % 8) || (t.ldx % 8) || (((uintptr_t)t.dY | (uintptr_t)t.X | (uintptr_t)t.dW) & 15)) return TIMHIP_EALIGN;
    if ((size_t)M * t.ldy * 2 >= (1ull << 32) || (size_t)M * t.ldx * 2 >= (1ull << 32)) return TIMHIP_EUNSUPPORTED;
    g.dY[i] = t.dY; g.X[i] = t.X; g.dW[i] = t.dW; g.db[i] = t.db; g.ldy[i] = t.ldy; g.ldx[i] = t.ldx; g.N[i] = t.Nout; g.K[i] = t.Kout;
    g.tile0[i + 1] = g.tile0[i] + (t.Nout / W8_T) * (t.Kout / W8_T);
  }
  const size_t shmem = (size_t)2 * W8_STAGE;
  static PerDeviceOnce attr_set[2];
  const int hi = precision == TIMHIP_PREC_F16 ? 1 : 0;
  if (attr_set[hi].first())
    DISPATCH_H16(precision, (void)hipFuncSetAttribute((const void*)wgrad_p8_kernel<HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
#ifdef TIMHIP_TUNING
  if (const char* v = getenv("TIMHIP_W8_ABL")) {
#define W8A(X) case X: (void)hipFuncSetAttribute((const void*)wgrad_p8_kernel<f16_t, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL((wgrad_p8_kernel<f16_t, X>), dim3((unsigned)g.tile0[n]), dim3(512), shmem, s, g); return TIMHIP_OK;
    switch (atoi(v)) { W8A(1) W8A(2) W8A(4) W8A(8) W8A(16) W8A(24) W8A(7) default: break; }
#undef W8A
  }
  if (const char* v = getenv("TIMHIP_W8_SCH")) {
#define W8S(X) case X: (void)hipFuncSetAttribute((const void*)wgrad_p8_kernel<f16_t, 0, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL((wgrad_p8_kernel<f16_t, 0, X>), dim3((unsigned)g.tile0[n]), dim3(512), shmem, s, g); return TIMHIP_OK;
    switch (atoi(v)) { W8S(1) W8S(2) W8S(4) W8S(5) W8S(8) W8S(3) default: break; }
#undef W8S
  }
#endif
  const bool two = tim_knobs().wgrad_p8_ph != 4;   // two 32-MFMA phases per step (TIMHIP_WGRAD_P8_PH=4: four 16-MFMA phases)
  const bool rag = (M % WP_M) != 0;
#define W8_LAUNCH(PH2_, RAG_) do { \
    DISPATCH_H16(precision, (void)hipFuncSetAttribute((const void*)wgrad_p8_kernel<HT, 0, 0, PH2_, RAG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    DISPATCH_H16(precision, hipLaunchKernelGGL((wgrad_p8_kernel<HT, 0, 0, PH2_, RAG_>), dim3((unsigned)g.tile0[n]), dim3(512), shmem, s, g)); } while (0)
  if (two && rag) W8_LAUNCH(true, true);
  else if (two) W8_LAUNCH(true, false);
  else if (rag) W8_LAUNCH(false, true);
  else W8_LAUNCH(false, false);
#undef W8_LAUNCH
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}
