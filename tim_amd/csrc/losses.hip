// Loss tail of the training step, HBM/latency-bound row kernels:
//   * label-smoothed cross entropy under mixup: train.py:46-49 (CrossEntropyLoss(label_smoothing=0.2,
//     ignore_index=-1)) applied twice through mixup.py:24-39 (lam * CE(pred[valid_a], y_a).mean() +
//     (1-lam) * CE(pred[valid_b], y_b).mean()).  One pass over the logits gives both terms and, in the backward,
//     one pass writes the gradient of their combination.
//   * DRLoc (drloc.py:4-41): gather of the sampled feature-token pairs straight into the operand buffer of the
//     drloc MLP's first GEMM, and the scatter-add of its input gradient back into the feature gradient.
#include "common.h"

namespace {

// ---- cross entropy --------------------------------------------------------------------------------------------------
// stats[r] = (logsumexp_r, mean_c x_rc); accum = (sum_a, n_a, sum_b, n_b) accumulated with atomics (rows <= a few
// thousand: the order-dependence of the fp32 sums is below 1e-7 relative)
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ x, int rows, int C, int ld,
                                                      const long long* __restrict__ ya,
                                                      const long long* __restrict__ yb, float eps,
                                                      float* __restrict__ stats, float* __restrict__ accum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const float* xr = x + (size_t)r * ld;
  float mx = -INFINITY, sm = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = xr[c]; mx = fmaxf(mx, v); sm += v; }
  mx = wave_max(mx);
  sm = wave_sum(sm);
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += __expf(xr[c] - mx);
  se = wave_sum(se);
  const float lse = mx + __logf(se), mean = sm / (float)C;
  if (lane == 0) {
    stats[2 * r] = lse;
    stats[2 * r + 1] = mean;
    const long long a = ya[r], b = yb ? yb[r] : -1;
    // CE_smooth(r, y) = (1-eps) (lse - x_y) + eps (lse - mean)
    if (a >= 0 && a < C) { atomicAdd(accum + 0, (1.f - eps) * (lse - xr[a]) + eps * (lse - mean)); atomicAdd(accum + 1, 1.f); }
    if (b >= 0 && b < C) { atomicAdd(accum + 2, (1.f - eps) * (lse - xr[b]) + eps * (lse - mean)); atomicAdd(accum + 3, 1.f); }
  }
}

// loss = lam * sum_a / n_a + (1 - lam) * sum_b / n_b   (a term with no valid row is 0, as the training loop skips it)
__global__ void ce_finish_kernel(const float* __restrict__ accum, float lam, float* __restrict__ loss) {
  const float la = accum[1] > 0.f ? accum[0] / accum[1] : 0.f;
  const float lb = accum[3] > 0.f ? accum[2] / accum[3] : 0.f;
  loss[0] = lam * la + (1.f - lam) * lb;
}

// d loss / d x_rc = g * [ (wa + wb) softmax_rc - wa ((1-eps) 1[c = ya] + eps/C) - wb ((1-eps) 1[c = yb] + eps/C) ]
//   wa = lam / n_a if ya valid, wb = (1-lam) / n_b if yb valid
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ x, int rows, int C, int ld,
                                                     const long long* __restrict__ ya,
                                                     const long long* __restrict__ yb, float lam, float eps,
                                                     const float* __restrict__ stats,
                                                     const float* __restrict__ accum, const float* __restrict__ gout,
                                                     float* __restrict__ dx, int ldd) {
  const int r = blockIdx.y;
  const float g = gout ? gout[0] : 1.f;
  const long long a = ya[r], b = yb ? yb[r] : -1;
  const float wa = (a >= 0 && a < C && accum[1] > 0.f) ? g * lam / accum[1] : 0.f;
  const float wb = (b >= 0 && b < C && accum[3] > 0.f) ? g * (1.f - lam) / accum[3] : 0.f;
  const float lse = stats[2 * r], w = wa + wb, u = w * eps / (float)C;
  const float* xr = x + (size_t)r * ld;
  float* dr = dx + (size_t)r * ldd;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
    float v = w * __expf(xr[c] - lse) - u;
    if (c == a) v -= wa * (1.f - eps);
    if (c == b) v -= wb * (1.f - eps);
    dr[c] = v;
  }
}

// ---- DRLoc ----------------------------------------------------------------------------------------------------------
// out[(b*m + i), 0:D] = x1[b, pos1[b,i], :],  out[.., D:2D] = x2[b, pos2[b,i], :]   (collect_samples + cat, drloc.py:11-15,
// 24-26); x1 / x2 are [n, l, D] views with element strides (sb, sl, 1)
template <typename T>
__global__ void drloc_gather_kernel(const float* __restrict__ x1, const float* __restrict__ x2, long long sb,
                                    long long sl, int l, int D, const long long* __restrict__ pos1,
                                    const long long* __restrict__ pos2, int m, T* __restrict__ out, int ld) {
  const int r = blockIdx.x, b = r / m;
  const int half = blockIdx.y;
  const long long p = half ? pos2[r] : pos1[r];
  const float* src = (half ? x2 : x1) + (size_t)b * sb + (size_t)p * sl;
  T* dst = out + (size_t)r * ld + (size_t)half * D;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    const float4 v = *reinterpret_cast<const float4*>(src + c);
    store4<T>(dst + c, v.x, v.y, v.z, v.w);
  }
}

// dx1[b, pos1[b,i], :] += g[(b*m+i), 0:D], dx2[b, pos2[b,i], :] += g[.., D:2D]   (positions repeat: atomics)
__global__ void drloc_scatter_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dx1,
                                     float* __restrict__ dx2, long long sb, long long sl, int D,
                                     const long long* __restrict__ pos1, const long long* __restrict__ pos2, int m) {
  const int r = blockIdx.x, b = r / m;
  const int half = blockIdx.y;
  const long long p = half ? pos2[r] : pos1[r];
  float* dst = (half ? dx2 : dx1) + (size_t)b * sb + (size_t)p * sl;
  const float* src = g + (size_t)r * ldg + (size_t)half * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) atomicAdd(dst + c, src[c]);
}

}  // namespace

extern "C" {

int timhip_ce_mixup_fwd(const float* logits, int rows, int C, int ld, const int64_t* target_a, const int64_t* target_b,
                        float lam, float smoothing, float* stats, float* accum, float* loss, void* stream) {
  if (!logits || !target_a || !stats || !accum || !loss || rows <= 0 || C <= 0 || ld < C) return TIMHIP_EINVAL;
  if (smoothing < 0.f || smoothing >= 1.f) return TIMHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(accum, 0, 4 * sizeof(float), s) != hipSuccess) return TIMHIP_ELAUNCH;
  hipLaunchKernelGGL(ce_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, rows, C, ld,
                     (const long long*)target_a, (const long long*)target_b, smoothing, stats, accum);
  TIM_CHECK_LAUNCH();
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(1), 0, s, accum, lam, loss);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_ce_mixup_bwd(const float* logits, int rows, int C, int ld, const int64_t* target_a, const int64_t* target_b,
                        float lam, float smoothing, const float* stats, const float* accum, const float* grad_out,
                        float* dlogits, int ldd, void* stream) {
  if (!logits || !target_a || !stats || !accum || !dlogits || rows <= 0 || C <= 0 || ld < C || ldd < C)
    return TIMHIP_EINVAL;
  dim3 grid((C + 1023) / 1024 > 8 ? 8 : (C + 1023) / 1024, rows);
  hipLaunchKernelGGL(ce_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, rows, C, ld,
                     (const long long*)target_a, (const long long*)target_b, lam, smoothing, stats, accum, grad_out,
                     dlogits, ldd);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_drloc_gather(int precision, const float* x1, const float* x2, int64_t batch_stride, int64_t row_stride,
                        int n, int l, int D, const int64_t* pos1, const int64_t* pos2, int m, void* out, int ld,
                        void* stream) {
  if (!x1 || !x2 || !pos1 || !pos2 || !out || n <= 0 || l <= 0 || m <= 0 || D <= 0 || D % 4 || ld < 2 * D || ld % 4 ||
      row_stride % 4 || batch_stride % 4)
    return TIMHIP_EINVAL;
  if ((((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)out) & 15) != 0) return TIMHIP_EALIGN;
  if (f32_storage(precision)) {
    hipLaunchKernelGGL(drloc_gather_kernel<float>, dim3(n * m, 2), dim3(128), 0, (hipStream_t)stream, x1, x2,
                       (long long)batch_stride, (long long)row_stride, l, D, (const long long*)pos1,
                       (const long long*)pos2, m, (float*)out, ld);
  } else {
    hipLaunchKernelGGL(drloc_gather_kernel<bf16_t>, dim3(n * m, 2), dim3(128), 0, (hipStream_t)stream, x1, x2,
                       (long long)batch_stride, (long long)row_stride, l, D, (const long long*)pos1,
                       (const long long*)pos2, m, (bf16_t*)out, ld);
  }
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_drloc_scatter_add(const float* d_pts, int ldg, float* dx1, float* dx2, int64_t batch_stride,
                             int64_t row_stride, int n, int l, int D, const int64_t* pos1, const int64_t* pos2, int m,
                             void* stream) {
  if (!d_pts || !dx1 || !dx2 || !pos1 || !pos2 || n <= 0 || l <= 0 || m <= 0 || D <= 0 || ldg < 2 * D)
    return TIMHIP_EINVAL;
  hipLaunchKernelGGL(drloc_scatter_kernel, dim3(n * m, 2), dim3(256), 0, (hipStream_t)stream, d_pts, ldg, dx1, dx2,
                     (long long)batch_stride, (long long)row_stride, D, (const long long*)pos1, (const long long*)pos2, m);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

}  // extern "C"
