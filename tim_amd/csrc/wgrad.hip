// Weight-gradient GEMM  dW[N,K] += dY[M,N]^T X[M,K]  (+ db[N] += colsum dY)  for nn.Linear
// (backward of transformers.py:102,107; encodings.py:22,141,148; tim.py:69-71; head.py:8-15).
//
// The contraction runs over the rows m of two row-major activations, i.e. over the STRIDED
// dimension of both operands.  Instead of materialising transposed copies, both tiles are staged in
// their natural [m][col] layout (global_load_lds, 16 B/lane, swizzle on the source address) and the
// MFMA fragments are produced by ds_read_b64_tr_b16 (gfx950 transposing LDS read).
//
//   D[i = k][j = n] = sum_m X[m][k] dY[m][n]   ->  a lane owns one output row n of dW and 4
//   consecutive k per accumulator quad: 16-byte fp32 stores.
//   bias gradient: the blocks of the first k-tile column also sum their dY fragments (registers).
//
// M is split across blockIdx.z; every split writes its partial tile into an fp32 slab with plain
// coalesced stores and timhip's slab-reduce kernel adds the slabs into dW / db (no atomics).
#include <cstdlib>
#include "common.h"
#include "mfma_tiles.h"

namespace {

__device__ __forceinline__ int xcd_remap_w(int b, int nb) {  // bijective for any grid size
  const int q = nb >> 3, r = nb & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr int WT = 128;   // output tile: 128 (n) x 128 (k)
constexpr int WM = 64;    // contraction rows per step

// One work item: the 128 x 128 tile (n0, k0) of dW over the contraction steps [s0, s1) of 64 rows.  `out` is an [N, K] fp32
// matrix (a slab of partial sums, or dW itself when the contraction is not split), `db_out` an [N] vector (first k-tile
// column only).  accumulate: out += / db_out += instead of =.
template <typename HT>
struct WgTile {
  const HT* dY; const HT* X; float* out; float* db_out;
  int ldy, ldx, M, N, K, n0, k0, s0, s1, do_bias, accumulate;
  float alpha;   // factor on what is written (1 for slabs; 1 / gradient scale for final outputs of the fp16 mode)
  const float* nf;   // out_scale when this tile writes FINAL gradients (non-finite watch, common.h:nf_note), else NULL
};

template <typename HT>
__device__ __forceinline__ void wgrad_tile(const WgTile<HT>& a, char* lds) {
  constexpr int TILE_BYTES = WM * WT * 2;
  const HT* __restrict__ dY = a.dY;
  const HT* __restrict__ X = a.X;
  const int ldy = a.ldy, ldx = a.ldx, M = a.M, N = a.N, K = a.K, n0 = a.n0, k0 = a.k0, s0 = a.s0, s1 = a.s1;
  const bool do_bias = a.do_bias != 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wn = wave & 1;

  // staging: one wave-instruction = 1 KiB = 4 rows x 256 B; lane -> (row, chunk').  Column chunks beyond the
  // operand's own columns (rounded up to 8: the leading dimensions are multiples of 8) are clamped (they only feed output columns that are never stored); rows beyond M
  // (last step only) must contribute zeros: that stage reads clamped rows and the padding rows of both LDS tiles
  // are zeroed before the fragments are read.
  const int lrow = lane >> 4, lc = lane & 15;
  uint32_t yoff[4], xoff[4];
  int srow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 4 + lrow;
    const int c = lc ^ swz<128>(row);
    srow[i] = row;
    yoff[i] = (uint32_t)(((size_t)row * ldy + min(n0 + c * 8, ((N + 7) & ~7) - 8)) * 2);
    xoff[i] = (uint32_t)(((size_t)row * ldx + min(k0 + c * 8, ((K + 7) & ~7) - 8)) * 2);
  }
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  auto stage = [&](int step, int buf) {
    const uint32_t base = lds0 + buf * 2 * TILE_BYTES + wave * 4096;
    const int m0 = step * WM;
    if (m0 + WM <= M) {
      glds16_xn<4>(reinterpret_cast<const char*>(dY) + (size_t)m0 * ldy * 2, yoff, base);
      glds16_xn<4>(reinterpret_cast<const char*>(X) + (size_t)m0 * ldx * 2, xoff, base + TILE_BYTES);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int over = max(m0 + srow[i] - (M - 1), 0);  // rows past the end re-read row M-1
        glds16(reinterpret_cast<const char*>(dY) + ((size_t)m0 - over) * ldy * 2 + yoff[i], base + i * 1024);
        glds16(reinterpret_cast<const char*>(X) + ((size_t)m0 - over) * ldx * 2 + xoff[i], base + TILE_BYTES + i * 1024);
      }
    }
  };

  f32x16_t acc[2][2];
  float bsum[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // per-lane byte offsets of the first transposing read of each fragment (rows 0..7 of a 16-row slice)
  int xtr[2][2], ytr[2][2];  // [fragment][rows 0-7 | rows 8-15]
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
  if (s0 < s1) stage(s0, 0);
  for (int st = s0; st < s1; ++st) {
    const int buf = (st - s0) & 1;
    glds_wait<0>();
    __syncthreads();
    if (st + 1 < s1) stage(st + 1, buf ^ 1);
    char* sY = lds + buf * 2 * TILE_BYTES;
    char* sX = sY + TILE_BYTES;
    if (st * WM + WM > M) {  // last, partial step: zero the rows >= M of both tiles (uniform branch)
      const int first = M - st * WM;
      vec8<HT> z;
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = (HT)0.f;
      for (int idx = tid; idx < (WM - first) * 16; idx += 256) {
        const int off = (first + idx / 16) * 256 + (idx % 16) * 16;  // whole rows: the chunk swizzle stays inside a row
        *reinterpret_cast<vec8<HT>*>(sY + off) = z;
        *reinterpret_cast<vec8<HT>*>(sX + off) = z;
      }
      __syncthreads();
    }
    vec8<HT> xf[2][2], yf[2][2];
    auto load_frags = [&](int ms, int set) {
      // rows advance by 16 per ms (16*256 B) and swz<128>(row + 16) == swz<128>(row): immediates
#pragma unroll
      for (int i = 0; i < 2; ++i)
        xf[set][i] = cat8<HT>(tr_read<HT>(sX + xtr[i][0] + ms * 4096), tr_read<HT>(sX + xtr[i][1] + ms * 4096));
#pragma unroll
      for (int j = 0; j < 2; ++j)
        yf[set][j] = cat8<HT>(tr_read<HT>(sY + ytr[j][0] + ms * 4096), tr_read<HT>(sY + ytr[j][1] + ms * 4096));
    };
    load_frags(0, 0);
#pragma unroll
    for (int ms = 0; ms < WM / 16; ++ms) {
      if (ms + 1 < WM / 16) load_frags(ms + 1, (ms + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = mfma16<HT>(xf[ms & 1][i], yf[ms & 1][j], acc[i][j]);
      if (do_bias && wk == 0) {  // bias gradient: column sums of dY straight from the B fragments
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int u = 0; u < 8; ++u) bsum[j] += (float)yf[ms & 1][j][u];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  const int li = lane & 31, g = lane >> 5;
  float* __restrict__ out = a.out;
  // transpose the accumulators through a wave-private LDS region so that 16 lanes store one contiguous
  // 64-column row segment of the slab (same scheme as the NT GEMM epilogue)
  constexpr int EP_LD = 64 + 4;
  __syncthreads();
  float* ep = reinterpret_cast<float*>(lds) + wave * (32 * EP_LD);
  float chk = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(ep + li * EP_LD + i * 32 + 8 * q + 4 * g) =
            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx >> 4, ch = idx & 15;
      float4 v = *reinterpret_cast<const float4*>(ep + row * EP_LD + ch * 4);
      v.x *= a.alpha; v.y *= a.alpha; v.z *= a.alpha; v.w *= a.alpha;
      nf_note(chk, v.x); nf_note(chk, v.y); nf_note(chk, v.z); nf_note(chk, v.w);
      const int n = n0 + wn * 64 + j * 32 + row;
      const int k = k0 + wk * 64 + ch * 4;
      if (n < N) {
        if (k + 3 < K) {
          float4* dst = reinterpret_cast<float4*>(out + (size_t)n * K + k);
          if (a.accumulate) {
            const float4 o = *dst;
            *dst = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
          } else {
            *dst = v;
          }
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
            if (k + tt < K) out[(size_t)n * K + k + tt] = a.accumulate ? out[(size_t)n * K + k + tt] + vv[tt] : vv[tt];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (do_bias && wk == 0) {
      const int n = n0 + wn * 64 + j * 32 + li;
      const float t2 = (bsum[j] + __shfl_xor(bsum[j], 32, 64)) * a.alpha;
      if (g == 0 && n < N) a.db_out[n] = a.accumulate ? a.db_out[n] + t2 : t2;
    }
  }
  nf_commit(a.nf, chk);
}

template <typename HT>
__global__ __launch_bounds__(256) void wgrad_tn_kernel(const HT* __restrict__ dY, int ldy,
                                                            const HT* __restrict__ X, int ldx, int M, int N,
                                                            int K, int steps_per_split, float* __restrict__ slab,
                                                            long long slab_stride, float* __restrict__ db_slab) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [2][sY 16 KB | sX 16 KB]
  const int tiles_k = (K + WT - 1) / WT, tiles_n = (N + WT - 1) / WT;
  // 1-D grid over (split, tile) work items, split-major.  Block b runs on XCD b % 8 (observed), so every XCD
  // is given a CONTIGUOUS range of work items: one split's row range of dY / X and a few n-tile rows, instead
  // of every XCD's L2 streaming all of dY and X (fabric reads 245 MB -> ~1.5x the operand bytes).
  const int ntiles = tiles_k * tiles_n;
  const int w = xcd_remap_w(blockIdx.x, gridDim.x);
  const int zsplit = w / ntiles;
  const int t = w - zsplit * ntiles;
  const int nsteps = (M + WM - 1) / WM;
  WgTile<HT> a;
  a.alpha = 1.f; a.nf = nullptr;
  a.dY = dY; a.X = X; a.ldy = ldy; a.ldx = ldx; a.M = M; a.N = N; a.K = K;
  // consecutive work items share the dY panel (same n-tile): k-tile fastest
  a.n0 = (t / tiles_k) * WT; a.k0 = (t % tiles_k) * WT;
  a.do_bias = db_slab != nullptr && (t % tiles_k) == 0;
  a.s0 = zsplit * steps_per_split;
  a.s1 = min(nsteps, a.s0 + steps_per_split);
  a.out = slab + (long long)zsplit * slab_stride;
  a.db_out = db_slab ? db_slab + (size_t)zsplit * N : nullptr;
  a.accumulate = 0;
  wgrad_tile(a, lds);
}

// ---- grouped form: the weight gradients of several Linear layers that share the contraction length M (one encoder
// layer: linear2, linear1, out-proj, in-proj) as ONE grid.  All work items cost the same (same M), so the tile lists are
// simply concatenated and the number of splits is chosen for the total: at C2a the four gradients are 128 + 128 + 64 + 192 =
// 512 tiles = exactly the 512 block slots, so the contraction is not split at all - every block runs the whole M and writes
// (or accumulates into) dW / db directly: no slabs (was 330 MB of slab writes + reads per layer), no reduce launches, one
// tail instead of four.
constexpr int WG_MAX = 8;
struct WgGroup {
  const void* dY[WG_MAX]; const void* X[WG_MAX]; float* dW[WG_MAX]; float* db[WG_MAX];
  int ldy[WG_MAX], ldx[WG_MAX], N[WG_MAX], K[WG_MAX];
  int tile0[WG_MAX + 1];          // first work tile of every item (prefix sums), tile0[n] = total
  long long off[WG_MAX + 1];      // element offset of every item's [N, K] block inside one slab
  int boff[WG_MAX + 1];           // same for the bias slabs
  int n, M, steps_per_split, splits, accumulate;
  float* slab; float* db_slab;    // [splits][off[n]] and [splits][boff[n]] (splits > 1 only)
  const float* out_scale;         // device scalar or NULL: factor on the final outputs (1 / gradient scale of the fp16 mode)
};

template <typename HT>
__global__ __launch_bounds__(256) void wgrad_group_kernel(const WgGroup g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int ntiles = g.tile0[g.n];
  const int w = xcd_remap_w(blockIdx.x, gridDim.x);
  const int zsplit = w / ntiles;
  const int t = w - zsplit * ntiles;
  WgTile<HT> a;
  int tl = t, tiles_k = 1;
  long long off = 0;
  int boff = 0;
  float* dW = nullptr;
  float* db = nullptr;
#pragma unroll
  for (int i = 0; i < WG_MAX; ++i) {   // static indexing of the kernel-argument arrays (uniform select)
    if (i < g.n && t >= g.tile0[i]) {
      a.dY = (const HT*)g.dY[i]; a.X = (const HT*)g.X[i]; a.ldy = g.ldy[i]; a.ldx = g.ldx[i]; a.N = g.N[i]; a.K = g.K[i];
      tl = t - g.tile0[i]; tiles_k = (g.K[i] + WT - 1) / WT;
      off = g.off[i]; boff = g.boff[i]; dW = g.dW[i]; db = g.db[i];
    }
  }
  a.M = g.M;
  a.n0 = (tl / tiles_k) * WT; a.k0 = (tl % tiles_k) * WT;
  a.do_bias = db != nullptr && (tl % tiles_k) == 0;
  const int nsteps = (g.M + WM - 1) / WM;
  a.s0 = zsplit * g.steps_per_split;
  a.s1 = min(nsteps, a.s0 + g.steps_per_split);
  if (g.splits == 1) {
    a.out = dW; a.db_out = db; a.accumulate = g.accumulate;
    a.alpha = g.out_scale ? *g.out_scale : 1.f; a.nf = g.out_scale;
  } else {
    a.alpha = 1.f; a.nf = nullptr;
    a.out = g.slab + (long long)zsplit * g.off[g.n] + off;
    a.db_out = g.db_slab + (long long)zsplit * g.boff[g.n] + boff;
    a.accumulate = 0;
  }
  wgrad_tile(a, lds);
}

// splits > 1: dW_i (+)= sum_z slab[z][off_i + .], db_i (+)= sum_z db_slab[z][boff_i + .], all items in one launch
__global__ void wgrad_group_reduce_kernel(const WgGroup g) {
  const long long nq = g.off[g.n] >> 2;           // every N*K is a multiple of 4 (checked by the launcher)
  const int nb = g.boff[g.n];
  const float alpha = g.out_scale ? *g.out_scale : 1.f;
  float chk = 0.f;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nq + nb; q += (long long)gridDim.x * blockDim.x) {
    if (q < nq) {
      const long long i = q << 2;
      float* dst = nullptr;
#pragma unroll
      for (int it = 0; it < WG_MAX; ++it)
        if (it < g.n && i >= g.off[it]) dst = g.dW[it] + (i - g.off[it]);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int z = 0; z < g.splits; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(g.slab + (long long)z * g.off[g.n] + i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
      nf_note(chk, acc.x); nf_note(chk, acc.y); nf_note(chk, acc.z); nf_note(chk, acc.w);
      if (g.accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(dst);
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
      }
      *reinterpret_cast<float4*>(dst) = acc;
    } else {
      const int j = (int)(q - nq);
      float* dst = nullptr;
#pragma unroll
      for (int it = 0; it < WG_MAX; ++it)
        if (it < g.n && j >= g.boff[it]) dst = g.db[it] + (j - g.boff[it]);   // items without a bias have an empty range
      if (!dst) continue;
      float acc = 0.f;
      for (int z = 0; z < g.splits; ++z) acc += g.db_slab[(long long)z * nb + j];
      *dst = acc * alpha + (g.accumulate ? *dst : 0.f);
    }
  }
  nf_commit(g.out_scale, chk);
}

// dW[i] += sum_z slab[z*n + i] (float4) and db[j] += sum_z dbs[z*nb + j]: one launch for both
// accumulate == 0: dW / db are WRITTEN (the caller's gradient buffer need not be zeroed nor read)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, long long n, int nslab, float* __restrict__ dW,
                                    const float* __restrict__ dbs, int nb, float* __restrict__ db, int accumulate,
                                    const float* __restrict__ out_scale) {
  const long long nq = n >> 2;
  const float alpha = out_scale ? *out_scale : 1.f;
  float chk = 0.f;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nq + nb; q += (long long)gridDim.x * blockDim.x) {
    if (q < nq) {
      const long long i = q << 2;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int z = 0; z < nslab; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (long long)z * n + i);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      a.x *= alpha; a.y *= alpha; a.z *= alpha; a.w *= alpha;
      nf_note(chk, a.x); nf_note(chk, a.y); nf_note(chk, a.z); nf_note(chk, a.w);
      if (accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(dW + i);
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
      *reinterpret_cast<float4*>(dW + i) = a;
    } else if (db) {
      const int j = (int)(q - nq);
      float a = 0.f;
      for (int z = 0; z < nslab; ++z) a += dbs[(long long)z * nb + j];
      db[j] = a * alpha + (accumulate ? db[j] : 0.f);
    }
  }
  nf_commit(out_scale, chk);
}

// out[i] += sum_z slab[z*stride + i]
__global__ void slab_reduce2_kernel(const float* __restrict__ slab, long long n, long long stride, int nslab,
                                    float* __restrict__ out, int accumulate, const float* __restrict__ out_scale) {
  const float alpha = out_scale ? *out_scale : 1.f;
  float chk = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int z = 0; z < nslab; ++z) a += slab[(long long)z * stride + i];
    nf_note(chk, a);
    out[i] = a * alpha + (accumulate ? out[i] : 0.f);
  }
  nf_commit(out_scale, chk);
}

}  // namespace

// Number of splits of the contraction (token) dimension.  Work items = tiles x splits run two per CU (512 slots).
// Cost model in units of one 64-row K-step of one block, fitted to tools/wgrad_splits.py on the C2a shapes:
//   steps per item x (full rounds + a partial round priced at 0.5 + 0.5 x its fill) + slab traffic per split
// e.g. in-proj (192 tiles): 3 splits = 576 items (1.125 rounds) 108 us, 5 splits = 960 items 100 us.
static int splits_for_tiles(int tiles, int M) {
  const int nsteps = (M + WM - 1) / WM;
#ifdef TIMHIP_TUNING
  if (const char* v = getenv("TIMHIP_WGRAD_SPLITS")) { int sk = atoi(v); if (sk >= 1 && sk <= nsteps) return sk; }
#endif
  int maxsk = nsteps / 4;
  if (maxsk > 16) maxsk = 16;
  if (maxsk < 1) maxsk = 1;
  int best = 1;
  float best_cost = 0.f;
  for (int sk = 1; sk <= maxsk; ++sk) {
    const float per = (float)((nsteps + sk - 1) / sk);
    const float r = (float)tiles * sk / 512.f;
    const float full = floorf(r), frac = r - full;
    const float cost = per * (full + (frac > 0.f ? 0.5f + 0.5f * frac : 0.f)) + 1.5f * (float)tiles / 128.f * sk;
    if (sk == 1 || cost < best_cost) { best = sk; best_cost = cost; }
  }
  return best;
}

int tim_wgrad_splits(int Nout, int Kout, int M) {
  return splits_for_tiles(((Nout + WT - 1) / WT) * ((Kout + WT - 1) / WT), M);
}

// ---- grouped launch (one encoder layer's four weight gradients) ------------------------------------------------------
static int group_tiles(const TimWgradItem* it, int n) {
  int tiles = 0;
  for (int i = 0; i < n; ++i) tiles += ((it[i].Nout + WT - 1) / WT) * ((it[i].Kout + WT - 1) / WT);
  return tiles;
}

int tim_wgrad_group_splits(const TimWgradItem* it, int n, int M) { return splits_for_tiles(group_tiles(it, n), M); }

size_t tim_wgrad_group_ws(const TimWgradItem* it, int n, int M) {
  const int sk = tim_wgrad_group_splits(it, n, M);
  if (sk == 1) return 0;
  size_t elems = 0, bias = 0;
  for (int i = 0; i < n; ++i) { elems += (size_t)it[i].Nout * it[i].Kout; bias += (size_t)it[i].Nout; }
  return align_up(sk * elems * 4, 256) + align_up(sk * bias * 4, 256);
}

int tim_wgrad_group_h16(int precision, const TimWgradItem* it, int n, int M, int accumulate, void* ws, size_t ws_bytes,
                        const float* out_scale, hipStream_t s) {
  if (!h16_storage(precision)) return TIMHIP_EUNSUPPORTED;
  if (!it || n < 1 || n > WG_MAX || M <= 0) return TIMHIP_EINVAL;
  // two encoder layers of a production batch: one round of eight-phase 256 x 256 tiles (wgrad_pp.hip; TIMHIP_WGRAD_P8=0: A/B switch)
  if (tim_wgrad_p8_wins(it, n, M)) {
    double fl = 0.0;
    for (int i = 0; i < n; ++i) fl += 2.0 * M * it[i].Nout * it[i].Kout;
    TimGemmScope timing(fl, s);
    return tim_wgrad_group_p8(precision, it, n, M, accumulate, out_scale, s);
  }
  // an encoder layer of a production batch: the one-block-per-CU ping-pong grid (wgrad_pp.hip; TIMHIP_WGRAD_PP=0: A/B switch)
  if (tim_knobs().wgrad_pp != 0 && tim_wgrad_pp_wins(it, n, M)) {
    double fl = 0.0;
    for (int i = 0; i < n; ++i) fl += 2.0 * M * it[i].Nout * it[i].Kout;
    TimGemmScope timing(fl, s);
    return tim_wgrad_group_pp(precision, it, n, M, accumulate, out_scale, s);
  }
  WgGroup g;
  g.n = n; g.M = M; g.accumulate = accumulate ? 1 : 0; g.out_scale = out_scale;
  g.tile0[0] = 0; g.off[0] = 0; g.boff[0] = 0;
  double flops = 0.0;
  for (int i = 0; i < WG_MAX; ++i) {
    if (i >= n) {
      g.dY[i] = g.X[i] = nullptr; g.dW[i] = g.db[i] = nullptr; g.ldy[i] = g.ldx[i] = g.N[i] = g.K[i] = 0;
      g.tile0[i + 1] = g.tile0[i]; g.off[i + 1] = g.off[i]; g.boff[i + 1] = g.boff[i];
      continue;
    }
    const TimWgradItem& t = it[i];
    if (!t.dY || !t.X || !t.dW || t.Nout <= 0 || t.Kout <= 0) return TIMHIP_EINVAL;
    if ((t.ldy % 8) || (t.ldx % 8) || (((uintptr_t)t.dY | (uintptr_t)t.X | (uintptr_t)t.dW) & 15)) return TIMHIP_EALIGN;
    if (((long long)t.Nout * t.Kout) & 3) return TIMHIP_EUNSUPPORTED;
    g.dY[i] = t.dY; g.X[i] = t.X; g.dW[i] = t.dW; g.db[i] = t.db;
    g.ldy[i] = t.ldy; g.ldx[i] = t.ldx; g.N[i] = t.Nout; g.K[i] = t.Kout;
    g.tile0[i + 1] = g.tile0[i] + ((t.Nout + WT - 1) / WT) * ((t.Kout + WT - 1) / WT);
    g.off[i + 1] = g.off[i] + (long long)t.Nout * t.Kout;
    g.boff[i + 1] = g.boff[i] + (t.db ? t.Nout : 0);
    flops += 2.0 * M * t.Nout * t.Kout;
  }
  const int sk = splits_for_tiles(g.tile0[n], M);
  const int nsteps = (M + WM - 1) / WM;
  const int per = (nsteps + sk - 1) / sk;
  const int sk_eff = (nsteps + per - 1) / per;  // no empty splits
  g.steps_per_split = per; g.splits = sk_eff;
  g.slab = nullptr; g.db_slab = nullptr;
  if (sk_eff > 1) {
    if (ws_bytes < tim_wgrad_group_ws(it, n, M) || !ws || ((uintptr_t)ws & 15)) return TIMHIP_EWORKSPACE;
    g.slab = (float*)ws;
    g.db_slab = (float*)((char*)ws + align_up((size_t)sk * g.off[n] * 4, 256));
  }
  TimGemmScope timing(flops, s);   // kernel (+ reduce)
  const size_t shmem = 2 * 2 * WM * WT * 2;
  DISPATCH_H16(precision, hipLaunchKernelGGL(wgrad_group_kernel<HT>, dim3((unsigned)(g.tile0[n] * sk_eff)), dim3(256), shmem, s, g));
  if (hipGetLastError() != hipSuccess) return TIMHIP_ELAUNCH;
  if (sk_eff > 1) {
    long long blocks = ((g.off[n] >> 2) + g.boff[n] + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g);
    if (hipGetLastError() != hipSuccess) return TIMHIP_ELAUNCH;
  }
  return TIMHIP_OK;
}

// workspace: [slab sk*N*K fp32][db slab sk*N fp32]
size_t tim_wgrad_tn_ws(int Nout, int Kout, int M) {
  const int sk = tim_wgrad_splits(Nout, Kout, M);
  return align_up((size_t)sk * Nout * Kout * 4, 256) + align_up((size_t)sk * Nout * 4, 256);
}

int tim_wgrad_tn_h16(int precision, const void* dY, int ldy, int Nout, const void* X, int ldx, int Kout, int M, float* dW,
                     float* db, void* ws, size_t ws_bytes, hipStream_t s, int accumulate, const float* out_scale) {
  if (!h16_storage(precision)) return TIMHIP_EUNSUPPORTED;
  if (ws_bytes < tim_wgrad_tn_ws(Nout, Kout, M)) return TIMHIP_EWORKSPACE;
  if ((ldy % 8) || (ldx % 8) || (((uintptr_t)dY | (uintptr_t)X | (uintptr_t)ws) & 15)) return TIMHIP_EALIGN;
  const int sk = tim_wgrad_splits(Nout, Kout, M);
  TimGemmScope timing(2.0 * M * Nout * Kout, s);   // kernel + slab reduce
  char* w = (char*)ws;
  float* slab = (float*)w;
  float* dbs = (float*)(w + align_up((size_t)sk * Nout * Kout * 4, 256));
  const int nsteps = (M + WM - 1) / WM;
  const int per = (nsteps + sk - 1) / sk;
  const int sk_eff = (nsteps + per - 1) / per;  // no empty splits
  dim3 grid(((Nout + WT - 1) / WT) * ((Kout + WT - 1) / WT) * sk_eff, 1, 1);
  const size_t shmem = 2 * 2 * WM * WT * 2;
  DISPATCH_H16(precision, hipLaunchKernelGGL(wgrad_tn_kernel<HT>, grid, dim3(256), shmem, s, (const HT*)dY, ldy, (const HT*)X, ldx,
                                             M, Nout, Kout, per, slab, (long long)Nout * Kout, db ? dbs : nullptr));
  if (hipGetLastError() != hipSuccess) return TIMHIP_ELAUNCH;
  const long long n = (long long)Nout * Kout;
  if ((n & 3) == 0) {
    long long blocks = (n / 4 + Nout + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slab, n, sk_eff, dW, dbs, Nout, db, accumulate, out_scale);
    return hipGetLastError() == hipSuccess ? TIMHIP_OK : TIMHIP_ELAUNCH;
  }
  hipLaunchKernelGGL(slab_reduce2_kernel, dim3(256), dim3(256), 0, s, slab, n, n, sk_eff, dW, accumulate, out_scale);
  if (hipGetLastError() != hipSuccess) return TIMHIP_ELAUNCH;
  if (db) {
    hipLaunchKernelGGL(slab_reduce2_kernel, dim3((Nout + 255) / 256), dim3(256), 0, s, dbs, (long long)Nout,
                       (long long)Nout, sk_eff, db, accumulate, out_scale);
    if (hipGetLastError() != hipSuccess) return TIMHIP_ELAUNCH;
  }
  return TIMHIP_OK;
}
