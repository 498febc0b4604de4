// Detection query labelling (detection/time_interval_machine/models/tim.py:157-270): every interval query of the multi-scale
// pyramid is matched to the ground-truth segment of its window with the largest 1-D IoU (first maximum wins, as torch.argmax),
// queries under the IoU threshold become negatives, and the classification targets are label-smoothed one-hot rows.
//
// HBM-bound index/byte work, no MFMA: kernel 1 is one thread per query over the handful of segments of its window (the
// arithmetic order of get_query_ious :186-212 is kept operation by operation, so the IoUs are the reference's fp32 values bit
// for bit); kernel 2 writes the [B*Nq, C] target matrix at store bandwidth (C = 3806 action classes: 97 MB per 16-window batch).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void label_queries_kernel(const float* __restrict__ queries, const float* __restrict__ segs,
                                                            const long long* __restrict__ labels, int B, int Nq, int Ng, int NL,
                                                            float thr, float* __restrict__ targets, float* __restrict__ ious,
                                                            long long* __restrict__ qlabels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // b * Nq + q
  if (i >= B * Nq) return;
  const int b = i / Nq;
  const float* sg = segs + (size_t)b * Ng * 2;
  // tim.py:196-203: shift the window so that its earliest ground-truth start is not negative
  float mn = sg[0];
  for (int g = 1; g < Ng; ++g) mn = fminf(mn, sg[2 * g]);
  const float off = fabsf(fminf(mn, 0.0f));
  const float qs = __fadd_rn(queries[2 * (size_t)i], off), qe = __fadd_rn(queries[2 * (size_t)i + 1], off);
  const float qlen = __fsub_rn(qe, qs);
  float best = 0.f, bs = 0.f, be = 0.f;
  int bi = -1;
  for (int g = 0; g < Ng; ++g) {
    const float gs = __fadd_rn(sg[2 * g], off), ge = __fadd_rn(sg[2 * g + 1], off);
    const float inter = fmaxf(__fsub_rn(fminf(qe, ge), fmaxf(qs, gs)), 0.0f);
    const float uni = __fsub_rn(__fadd_rn(__fsub_rn(ge, gs), qlen), inter);
    const float iou = __fdiv_rn(inter, uni);
    // torch.argmax: first maximum; a NaN (0/0: a zero-length segment on a zero-length query) counts as the maximum
    const bool take = bi < 0 || (iou > best) || (iou != iou && best == best);
    if (take) { best = iou; bi = g; bs = gs; be = ge; }
  }
  const bool negative = best < thr;
  ious[i] = best;
  targets[2 * (size_t)i] = negative ? INFINITY : bs;        // the SHIFTED segment, as the reference returns it
  targets[2 * (size_t)i + 1] = negative ? INFINITY : be;
  for (int c = 0; c < NL; ++c)
    qlabels[(size_t)i * NL + c] = negative ? -1ll : labels[((size_t)b * Ng + bi) * NL + c];
}

// out[r, c] = (c == label[r]) ? on : base for c < n; a label of -1 (negative query) or >= n leaves the whole row at `base`
// (tim.py:170-183: one-hot over n + 1 classes, the extra "no object" column dropped).  Flat 16-byte stores.
__global__ __launch_bounds__(256) void smooth_one_hot_kernel(const long long* __restrict__ qlabels, int ld, int col, long long rows,
                                                             int n, float on, float base, float* __restrict__ out) {
  const long long total = rows * (long long)n;
  for (long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i4 < total; i4 += (long long)gridDim.x * blockDim.x * 4) {
    long long r = i4 / n;
    int c = (int)(i4 - r * n);
    long long lab = qlabels[r * ld + col];
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = (lab == (long long)c) ? on : base;
      if (++c == n) { c = 0; ++r; if (r < rows) lab = qlabels[r * ld + col]; }
    }
    if (i4 + 3 < total) {
      *reinterpret_cast<float4*>(out + i4) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int k = 0; k < 4 && i4 + k < total; ++k) out[i4 + k] = v[k];
    }
  }
}

}  // namespace

extern "C" {

int timhip_label_queries(const float* queries, const float* segs, const int64_t* labels, int B, int Nq, int Ng, int NL,
                         float iou_threshold, float* targets, float* ious, int64_t* qlabels, void* stream) {
  if (!queries || !segs || !labels || !targets || !ious || !qlabels) return TIMHIP_EINVAL;
  if (B <= 0 || Nq <= 0 || Ng <= 0 || NL <= 0) return TIMHIP_EINVAL;
  hipLaunchKernelGGL(label_queries_kernel, dim3((B * Nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, queries, segs,
                     (const long long*)labels, B, Nq, Ng, NL, iou_threshold, targets, ious, (long long*)qlabels);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_smooth_one_hot(const int64_t* qlabels, int ld, int col, int64_t rows, int n, float on, float base, float* out,
                          void* stream) {
  if (!qlabels || !out || ld <= 0 || col < 0 || col >= ld || rows <= 0 || n <= 0) return TIMHIP_EINVAL;
  if (((uintptr_t)out) & 15) return TIMHIP_EALIGN;
  long long blocks = (rows * (long long)n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
  hipLaunchKernelGGL(smooth_one_hot_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)qlabels, ld,
                     col, (long long)rows, n, on, base, out);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

}  // extern "C"
