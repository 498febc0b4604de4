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
//     behind waves 0-3); every wave issues 3 of the stage's 48 DMA pieces between the MFMAs of each of its MFMA phases.
//     That form is kept (MODE 0); the default (MODE 1) drops the phase barriers - one barrier per 64-row step is all the ring
//     needs - and lets the hardware scheduler interleave the two waves of a SIMD: measured 217 vs 232 us for the C2a layer
//     on one box (tools/wgpp_abl.py; the old two-blocks-per-CU kernel: 243).  The kernel stays bound by the LDS: its 256
//     transposing reads per 64-row step cost ~4 LDS cycles each (half the bytes per instruction of ds_read_b128) - as long
//     as the MFMAs of the step - and the stage's DMA writes share the same LDS.
// Hazard bookkeeping of MODE 0: see gemm_pp.hip (identical phase structure).
#include <stdlib.h>

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

template <typename HT, int ABL = 0, int MODE = 1>
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
  } else {
    if (wn == 0) wp_mainloop<HT, 0, ABL>(a, g.M, lds, lds0, wave, lane, xoff, yoff, acc, accb, do_bias);
    else wp_mainloop<HT, 1, ABL>(a, g.M, lds, lds0, wave, lane, xoff, yoff, acc, accb, do_bias);
  }

  // D[row = k: 4 gid + r][col = n: p] per MFMA tile (i: k tile, j: n tile): a lane owns 4 consecutive k of one row n of dW
  const float alpha = g.out_scale ? *g.out_scale : 1.f;
  const int N = a.N, K = a.K;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = a.n0 + wn * 64 + j * 16 + p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = a.k0 + wk * 64 + i * 16 + 4 * gid;
      if (n < N && k < K) {
        float v[4] = {acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha};
        float* dst = dW + (size_t)n * K + k;
        if (k + 3 < K && ((((size_t)n * K + k) & 3) == 0)) {
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (g.accumulate) { const float4 c = *reinterpret_cast<const float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
          *reinterpret_cast<float4*>(dst) = o;
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
  static bool attr_set[2] = {false, false};
  const int hi = precision == TIMHIP_PREC_F16 ? 1 : 0;
  if (!attr_set[hi]) {
    DISPATCH_H16(precision, (void)hipFuncSetAttribute((const void*)wgrad_pp_kernel<HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_set[hi] = true;
  }
#ifdef TIMHIP_TUNING
  if (const char* v = getenv("TIMHIP_WGPP_ABL")) {
    const int abl = atoi(v);
    const int mode = getenv("TIMHIP_WGPP_MODE") ? atoi(getenv("TIMHIP_WGPP_MODE")) : 1;
#define WABL(X, MD) case X + 8 * MD: (void)hipFuncSetAttribute((const void*)wgrad_pp_kernel<f16_t, X, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL((wgrad_pp_kernel<f16_t, X, MD>), dim3((unsigned)g.tile0[n]), dim3(512), shmem, s, g); return TIMHIP_OK;
    switch (abl + 8 * mode) { WABL(0, 0) WABL(1, 0) WABL(2, 0) WABL(4, 0) WABL(6, 0) WABL(7, 0) WABL(0, 1) WABL(2, 1) WABL(4, 1) WABL(6, 1) default: break; }
#undef WABL
  }
#endif
  DISPATCH_H16(precision, hipLaunchKernelGGL(wgrad_pp_kernel<HT>, dim3((unsigned)g.tile0[n]), dim3(512), shmem, s, g));
  return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
}
