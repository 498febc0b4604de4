// Batched 1-D segment NMS / soft-NMS: the GPU counterpart of the reference's only native component,
// detection/eval_detection/csrc/nms_cpu.cpp (nms_1d_cpu :19-60, softnms_1d_cpu :69-170), which the evaluation drives
// once per (video, class) from joblib workers (format_predictions_epic.py:146-156, nms.py:97-180).
//
// Parallelism: the groups (one per video x class) are independent -> one wavefront per group (8 for groups of more
// than 1024 segments, 16 beyond 4096), thousands in flight.
// Inside a group soft-NMS is a sequential selection (pick the current maximum, decay the rest, prune), so the wave runs
// the outer loop and its 64 lanes share each inner pass (arg-max, decay, compaction).  The result is BIT-IDENTICAL to
// the CPU routine, including its order-dependent details:
//   * arg-max keeps the FIRST maximum of the current array order (strict `<` in nms_cpu.cpp:100-106);
//   * pruning swaps the dead element with the current last one and re-examines the slot (nms_cpu.cpp:150-160), which
//     permutes the array; the same permutation is produced in parallel: a dead slot among the first (i+1+alive) positions
//     receives the k-th alive element counted from the end, k = the slot's rank among such dead slots.
//   * float arithmetic in the same order.  The gaussian weight is exp() evaluated in double and rounded to float: glibc's
//     expf (max error 0.502 ulp) and this agree except for arguments within ~0.002 ulp of a rounding boundary, whereas the
//     1-ulp device expf differs in ~15 % of calls and, compounded over hundreds of decay steps, flips near-tied selections.
// Working arrays live in LDS when the group fits (four size classes, one launch each) and in a global scratch otherwise.
#include <algorithm>
#include "common.h"

namespace {

struct NmsArgs {
  const float* segs; const float* scores; const int* goff; const int* glist; int ngroups;
  float iou_thr, sigma, min_score; int method;
  float* scratch_f;   // [4][N] x1, x2, sc, area   (global fallback)
  int* scratch_i;     // [2][N] ind, tail
  long long N;
  float* dets; int* inds; int* count;
  int lds_cap;        // elements per group held in LDS (0: global scratch)
};

__device__ __forceinline__ float seg_weight(float ix1, float ix2, float iarea, float x1, float x2, float area, int method,
                                            float iou_thr, float sigma) {
  const float xx1 = fmaxf(ix1, x1), xx2 = fminf(ix2, x2);
  const float inter = fmaxf(0.f, xx2 - xx1);
  const float ovr = inter / (iarea + area - inter);
  float w = 1.f;
  if (method == 0) { if (ovr >= iou_thr) w = 0.f; }
  else if (method == 1) { if (ovr >= iou_thr) w = 1.f - ovr; }
  else { w = (float)exp((double)(-(ovr * ovr) / sigma)); }   // see the header comment: correctly rounded like glibc's expf
  return w;
}

// rank of this thread among the threads with flag set (thread order) and their number, over a block of NT threads
template <int NT>
__device__ __forceinline__ void flag_rank(bool flag, int* wtot, int& rank, int& total) {
  const int lane = threadIdx.x & 63;
  const unsigned long long m = __ballot(flag);
  rank = __popcll(m & ((1ull << lane) - 1ull));
  total = __popcll(m);
  if (NT > 64) {
    const int w = threadIdx.x >> 6;
    __syncthreads();                       // wtot may still be read from the previous call
    if (lane == 0) wtot[w] = total;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) { const int t = wtot[k]; all += t; if (k < w) before += t; }
    rank += before;
    total = all;
  }
}

// NT = 64: one wave per group (small groups: many groups per CU); NT = 512 / 1024: 8 / 16 waves share the passes of a
// large group (its selection loop is sequential, so the largest groups set the makespan of the whole call)
template <int NT>
__global__ __launch_bounds__(NT) void softnms_kernel(NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int wtot[16];
  __shared__ float wbest[16];
  __shared__ int wbpos[16];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.ngroups) return;
  const int g = a.glist ? a.glist[blockIdx.x] : (int)blockIdx.x;
  const int base = a.goff[g], n = a.goff[g + 1] - base;
  float *x1, *x2, *sc, *ar; int *ind, *tail;
  if (a.lds_cap > 0) {
    float* f = reinterpret_cast<float*>(smem);
    x1 = f; x2 = f + a.lds_cap; sc = f + 2 * a.lds_cap; ar = f + 3 * a.lds_cap;
    ind = reinterpret_cast<int*>(f + 4 * a.lds_cap); tail = ind + a.lds_cap;
  } else {
    x1 = a.scratch_f + base; x2 = a.scratch_f + a.N + base; sc = a.scratch_f + 2 * a.N + base;
    ar = a.scratch_f + 3 * a.N + base; ind = a.scratch_i + base; tail = a.scratch_i + a.N + base;
  }
  auto sync = [&]() { if (NT > 64) __syncthreads(); else __threadfence_block(); };
  for (int p = tid; p < n; p += NT) {
    const float s0 = a.segs[2 * (size_t)(base + p)], s1 = a.segs[2 * (size_t)(base + p) + 1];
    x1[p] = s0; x2[p] = s1; sc[p] = a.scores[base + p];
    ar[p] = (s1 - s0) + 1e-6f;
    ind[p] = p;
  }
  sync();
  int nsegs = n;
  for (int i = 0; i < nsegs; ++i) {
    // ---- first maximum of sc[i .. nsegs)
    float best = -INFINITY; int bpos = 0x7fffffff;
    for (int p = i + tid; p < nsegs; p += NT) {
      const float v = sc[p];
      if (v > best || bpos == 0x7fffffff) { best = v; bpos = p; }
    }
    auto better = [](float ob, int op, float b, int bp) {
      return op != 0x7fffffff && (bp == 0x7fffffff || ob > b || (ob == b && op < bp));
    };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64); const int op = __shfl_xor(bpos, o, 64);
      if (better(ob, op, best, bpos)) { best = ob; bpos = op; }
    }
    if (NT > 64) {
      if ((tid & 63) == 0) { wbest[tid >> 6] = best; wbpos[tid >> 6] = bpos; }
      __syncthreads();
      best = wbest[0]; bpos = wbpos[0];
#pragma unroll
      for (int k = 1; k < NT / 64; ++k)
        if (better(wbest[k], wbpos[k], best, bpos)) { best = wbest[k]; bpos = wbpos[k]; }
    }
    // ---- select it: dets[i] <- it, swap positions i and bpos
    const float ix1 = x1[bpos], ix2 = x2[bpos], isc = sc[bpos], iar = ar[bpos];
    const int iind = ind[bpos];
    sync();
    if (tid == 0) {
      x1[bpos] = x1[i]; x2[bpos] = x2[i]; sc[bpos] = sc[i]; ar[bpos] = ar[i]; ind[bpos] = ind[i];
      x1[i] = ix1; x2[i] = ix2; sc[i] = isc; ar[i] = iar; ind[i] = iind;
      float* d = a.dets + 3 * (size_t)(base + i);
      d[0] = ix1; d[1] = ix2; d[2] = isc;
      a.inds[base + i] = iind;
    }
    sync();
    // ---- decay every remaining score once; count the survivors
    int mine = 0;
    for (int p = i + 1 + tid; p < nsegs; p += NT) {
      const float s = sc[p] * seg_weight(ix1, ix2, iar, x1[p], x2[p], ar[p], a.method, a.iou_thr, a.sigma);
      sc[p] = s;
      mine += !(s < a.min_score);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    int alive_cnt = mine;
    if (NT > 64) {
      __syncthreads();
      if ((tid & 63) == 0) wtot[tid >> 6] = mine;
      __syncthreads();
      alive_cnt = 0;
#pragma unroll
      for (int k = 0; k < NT / 64; ++k) alive_cnt += wtot[k];
    }
    const int new_n = i + 1 + alive_cnt;
    sync();
    if (new_n < nsegs) {
      // ---- the alive elements of the tail [new_n, nsegs), last first
      int k = 0;
      for (int q0 = nsegs - 1; q0 >= new_n; q0 -= NT) {
        const int p = q0 - tid;
        const bool alive = p >= new_n && !(sc[p] < a.min_score);
        int rank, total;
        flag_rank<NT>(alive, wtot, rank, total);
        if (alive) tail[k + rank] = p;
        k += total;
      }
      sync();
      // ---- dead slots of the front (i, new_n), first first, receive them
      k = 0;
      for (int p0 = i + 1; p0 < new_n; p0 += NT) {
        const int p = p0 + tid;
        const bool dead = p < new_n && (sc[p] < a.min_score);
        int rank, total;
        flag_rank<NT>(dead, wtot, rank, total);
        if (dead) {
          const int src = tail[k + rank];
          x1[p] = x1[src]; x2[p] = x2[src]; sc[p] = sc[src]; ar[p] = ar[src]; ind[p] = ind[src];
        }
        k += total;
      }
      sync();
    }
    nsegs = new_n;
  }
  if (tid == 0) a.count[g] = nsegs;
}

// vanilla NMS (nms_cpu.cpp:19-60) on segments already ordered by descending score inside each group: lane-parallel
// suppression by each kept segment in turn; keep[] receives the kept positions (in that order), count the number.
__global__ __launch_bounds__(64) void nms_kernel(const float* __restrict__ segs, const int* __restrict__ order,
                                                 const int* __restrict__ goff, int ngroups, float iou_thr,
                                                 unsigned char* __restrict__ removed, int* __restrict__ keep,
                                                 int* __restrict__ count) {
  const int lane = threadIdx.x, g = blockIdx.x;
  if (g >= ngroups) return;
  const int base = goff[g], n = goff[g + 1] - base;
  for (int p = lane; p < n; p += 64) removed[base + p] = 0;
  __threadfence_block();
  int m = 0;
  for (int _i = 0; _i < n; ++_i) {
    if (removed[base + _i]) continue;               // wave-uniform (same address for every lane)
    const int i = order[base + _i];
    const float ix1 = segs[2 * (size_t)(base + i)], ix2 = segs[2 * (size_t)(base + i) + 1];
    const float iarea = (ix2 - ix1) + 1e-6f;
    if (lane == 0) keep[base + m] = i;
    ++m;
    for (int _j = _i + 1 + lane; _j < n; _j += 64) {
      if (removed[base + _j]) continue;
      const int j = order[base + _j];
      const float jx1 = segs[2 * (size_t)(base + j)], jx2 = segs[2 * (size_t)(base + j) + 1];
      const float inter = fmaxf(0.f, fminf(ix2, jx2) - fmaxf(ix1, jx1));
      const float ovr = inter / (iarea + ((jx2 - jx1) + 1e-6f) - inter);
      if (ovr >= iou_thr) removed[base + _j] = 1;
    }
    __threadfence_block();
  }
  if (lane == 0) count[g] = m;
}

}  // namespace

extern "C" {

size_t timhip_softnms_1d_workspace_bytes(int64_t n_total, int n_groups) {
  return (size_t)align_up((size_t)n_total * 4 * sizeof(float), 256) + align_up((size_t)n_total * 2 * sizeof(int), 256) +
         align_up((size_t)(n_groups > 0 ? n_groups : 1) * 5 * sizeof(int), 256);
}

int timhip_softnms_1d(const float* segs, const float* scores, const int32_t* group_offsets,
                      const int32_t* group_offsets_host, int n_groups, float iou_threshold, float sigma, float min_score,
                      int method, float* dets, int32_t* inds, int32_t* count, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (!segs || !scores || !group_offsets || !group_offsets_host || !dets || !inds || !count || n_groups < 0)
    return TIMHIP_EINVAL;
  if (method < 0 || method > 2) return TIMHIP_EINVAL;
  if (n_groups == 0) return TIMHIP_OK;
  const long long N = group_offsets_host[n_groups];
  if (!workspace || workspace_bytes < timhip_softnms_1d_workspace_bytes(N, n_groups)) return TIMHIP_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)workspace;
  float* sf = (float*)w;
  int* si = (int*)(w + align_up((size_t)N * 4 * sizeof(float), 256));
  int* lists = si + (align_up((size_t)N * 2 * sizeof(int), 256) / sizeof(int));
  // size classes: groups of up to 256 / 1024 / 4096 elements keep their arrays in LDS (24 B per element), larger ones
  // use the global scratch
  const int caps[4] = {256, 1024, 4096, 0};
  int* host_list = (int*)malloc(sizeof(int) * (size_t)n_groups * 4);
  int cnt[4] = {0, 0, 0, 0};
  for (int g = 0; g < n_groups; ++g) {
    const int n = group_offsets_host[g + 1] - group_offsets_host[g];
    if (n < 0) { free(host_list); return TIMHIP_EINVAL; }
    const int c = n <= 256 ? 0 : (n <= 1024 ? 1 : (n <= 4096 ? 2 : 3));
    host_list[(size_t)c * n_groups + cnt[c]++] = g;
  }
  // longest first: a group's time grows with (size x survivors) and the biggest ones set the makespan
  for (int c = 0; c < 4; ++c) {
    int* l = host_list + (size_t)c * n_groups;
    std::sort(l, l + cnt[c], [&](int x, int y) {
      return group_offsets_host[x + 1] - group_offsets_host[x] > group_offsets_host[y + 1] - group_offsets_host[y];
    });
  }
  int rc = TIMHIP_OK;
  int off = 0;
  for (int c = 3; c >= 0 && rc == TIMHIP_OK; --c) {
    if (cnt[c] == 0) continue;
    if (hipMemcpyAsync(lists + off, host_list + (size_t)c * n_groups, sizeof(int) * (size_t)cnt[c], hipMemcpyHostToDevice,
                       s) != hipSuccess) { rc = TIMHIP_ELAUNCH; break; }
    NmsArgs a;
    a.segs = segs; a.scores = scores; a.goff = group_offsets; a.glist = lists + off; a.ngroups = cnt[c];
    a.iou_thr = iou_threshold; a.sigma = sigma; a.min_score = min_score; a.method = method;
    a.scratch_f = sf; a.scratch_i = si; a.N = N; a.dets = dets; a.inds = inds; a.count = count; a.lds_cap = caps[c];
    const size_t shmem = (size_t)caps[c] * 24;
    if (c == 3) {          // more than 4096 segments (global scratch): sixteen waves per group
      hipLaunchKernelGGL(softnms_kernel<1024>, dim3(cnt[c]), dim3(1024), shmem, s, a);
    } else if (c == 2) {   // 1025 .. 4096 segments: eight waves per group (96 KB of LDS: one group per CU anyway)
      if (shmem > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)softnms_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      hipLaunchKernelGGL(softnms_kernel<512>, dim3(cnt[c]), dim3(512), shmem, s, a);
    } else {
      hipLaunchKernelGGL(softnms_kernel<64>, dim3(cnt[c]), dim3(64), shmem, s, a);
    }
    if (hipGetLastError() != hipSuccess) rc = TIMHIP_ELAUNCH;
    off += cnt[c];
  }
  // the pageable host list is consumed by the asynchronous copies: wait for them before freeing it
  if (hipStreamSynchronize(s) != hipSuccess && rc == TIMHIP_OK) rc = TIMHIP_ELAUNCH;
  free(host_list);
  return rc;
}

int timhip_nms_1d(const float* segs, const int32_t* order, const int32_t* group_offsets, int n_groups, float iou_threshold,
                  uint8_t* removed_scratch, int32_t* keep, int32_t* count, void* stream) {
  if (!segs || !order || !group_offsets || !removed_scratch || !keep || !count || n_groups < 0) return TIMHIP_EINVAL;
  if (n_groups == 0) return TIMHIP_OK;
  hipLaunchKernelGGL(nms_kernel, dim3(n_groups), dim3(64), 0, (hipStream_t)stream, segs, order, group_offsets, n_groups,
                     iou_threshold, removed_scratch, keep, count);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

}  // extern "C"
