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
// stats[r] = (logsumexp_r, mean_c x_rc, CE_a(r) or -1 if target_a[r] is ignored, CE_b(r) or -1); ce_finish_kernel sums them
// into accum = (sum_a, n_a, sum_b, n_b) in a fixed order (no atomics: 4 hot addresses serialise a thousand rows)
// RW = waves per row: 1 (four rows per block) for small heads, 4 (one row per 256-thread block) for the wide action head
template <int RW>
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ x, int rows, int C, int ld,
                                                      const long long* __restrict__ ya,
                                                      const long long* __restrict__ yb, float eps,
                                                      float* __restrict__ stats) {
  __shared__ float red[3][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = RW == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
  const bool active = r < rows;                       // whole waves (RW = 1) or whole blocks (RW = 4) are inactive together
  const float* xr = x + (size_t)(active ? r : 0) * ld;
  const int t0 = RW == 1 ? lane : threadIdx.x, nt = RW * 64;
  float mx = -INFINITY, sm = 0.f;
  if (active)
    for (int c = t0; c < C; c += nt) { const float v = xr[c]; mx = fmaxf(mx, v); sm += v; }
  mx = wave_max(mx);
  sm = wave_sum(sm);
  if (RW > 1) {
    if (lane == 0) { red[0][wave] = mx; red[1][wave] = sm; }
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    sm = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
  float se = 0.f;
  if (active)
    for (int c = t0; c < C; c += nt) se += __expf(xr[c] - mx);
  se = wave_sum(se);
  if (RW > 1) {
    if (lane == 0) red[2][wave] = se;
    __syncthreads();
    se = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  }
  if (!active) return;
  const float lse = mx + __logf(se), mean = sm / (float)C;
  if (t0 == 0) {
    const long long a = ya[r], b = yb ? yb[r] : -1;
    // CE_smooth(r, y) = (1-eps) (lse - x_y) + eps (lse - mean) >= 0; -1 marks an ignored target
    stats[4 * r] = lse;
    stats[4 * r + 1] = mean;
    stats[4 * r + 2] = (a >= 0 && a < C) ? (1.f - eps) * (lse - xr[a]) + eps * (lse - mean) : -1.f;
    stats[4 * r + 3] = (b >= 0 && b < C) ? (1.f - eps) * (lse - xr[b]) + eps * (lse - mean) : -1.f;
  }
}

// loss = lam * sum_a / n_a + (1 - lam) * sum_b / n_b   (a term with no valid row is 0, as the training loop skips it)
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ stats, int rows, float lam,
                                                        float* __restrict__ accum, float* __restrict__ loss) {
  __shared__ float red[4][4];
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float a = stats[4 * r + 2], b = stats[4 * r + 3];
    if (a >= 0.f) { v[0] += a; v[1] += 1.f; }
    if (b >= 0.f) { v[2] += b; v[3] += 1.f; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x >> 6] = v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4];
    for (int k = 0; k < 4; ++k) { t[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]); accum[k] = t[k]; }
    const float la = t[1] > 0.f ? t[0] / t[1] : 0.f;
    const float lb = t[3] > 0.f ? t[2] / t[3] : 0.f;
    loss[0] = lam * la + (1.f - lam) * lb;
  }
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
  const float lse = stats[4 * r], w = wa + wb, u = w * eps / (float)C;
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

// ---- detection losses ---------------------------------------------------------------------------------
// sigmoid focal loss (detection models/helpers/losses/sigmoid.py:5-52) with the per-row weights and the "sum"
// reduction of get_loss (losses/loss.py:5-14), rows with valid[r] == 0 skipped (train.py:224-226 filters them):
//   loss = sum_r w_r sum_c alpha_t ce (1 - p_t)^gamma
struct FocalTerm { float loss, dx; };
__device__ __forceinline__ FocalTerm focal_term(float x, float t, float alpha, float gamma) {
  const float p = 1.f / (1.f + __expf(-x));
  const float ce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));     // BCE with logits
  const float pt = p * t + (1.f - p) * (1.f - t);
  const float q = 1.f - pt;
  const float mod = gamma == 2.f ? q * q : powf(q, gamma);
  const float dmod = gamma == 2.f ? 2.f * q : (q > 0.f ? gamma * powf(q, gamma - 1.f) : 0.f);
  const float at = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
  FocalTerm r;
  r.loss = at * ce * mod;
  // d ce / dx = p - t ; d q / dx = -(2t - 1) p (1 - p)
  r.dx = at * ((p - t) * mod - ce * dmod * (2.f * t - 1.f) * p * (1.f - p));
  return r;
}

__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                        long long n, int C, const float* __restrict__ w,
                                                        const unsigned char* __restrict__ valid, float alpha,
                                                        float gamma, float* __restrict__ out,
                                                        float* __restrict__ elem) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C;
    float v = 0.f;
    if (!valid || valid[r]) v = (w ? w[r] : 1.f) * focal_term(x[i], t[i], alpha, gamma).loss;
    if (elem) elem[i] = v;   // reduction = "none" (the training loop's positive / negative split meters)
    acc += v;
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                        long long n, int C, const float* __restrict__ w,
                                                        const unsigned char* __restrict__ valid, float alpha,
                                                        float gamma, const float* __restrict__ gout,
                                                        float* __restrict__ dx) {
  const float g = gout ? gout[0] : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C;
    float v = 0.f;
    if (!valid || valid[r]) v = g * (w ? w[r] : 1.f) * focal_term(x[i], t[i], alpha, gamma).dx;
    dx[i] = v;
  }
}

// 1-D centre-offset DIoU loss (losses/iou.py:4-65), "sum" reduction over the rows with valid[r] != 0; off = (left, right)
__global__ void diou_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, int n,
                            const unsigned char* __restrict__ valid, float eps, const float* __restrict__ gout,
                            float* __restrict__ loss, float* __restrict__ dpred) {
  float acc = 0.f;
  const float g = gout ? gout[0] : 1.f;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    float dl = 0.f, dr = 0.f;
    if (!valid || valid[r]) {
      const float lp = pred[2 * r], rp = pred[2 * r + 1], lg = tgt[2 * r], rg = tgt[2 * r + 1];
      const float lk = fminf(lp, lg), rk = fminf(rp, rg);
      const float I = rk + lk, U = (lp + rp) + (lg + rg) - I, Uc = fmaxf(U, eps);
      const float lc = fmaxf(lp, lg), rc = fmaxf(rp, rg), Lc = lc + rc, Lcc = fmaxf(Lc, eps);
      const float rho = 0.5f * (rp - lp - rg + lg), z = rho / Lcc;
      acc += 1.f - I / Uc + z * z;
      if (dpred) {
        // The reference function is TorchScript (@torch.jit.script, iou.py:3): once compiled, its autodiff of min / max
        // passes the gradient under STRICT comparison (nothing at an exact tie) and clamp(min=eps) where the input >= eps;
        // its first (profiling) calls split ties like eager torch.  Only exact ties differ; this follows the compiled form.
        const float dI_l = lp < lg ? 1.f : 0.f, dI_r = rp < rg ? 1.f : 0.f;
        const float dLc_l = lp > lg ? 1.f : 0.f, dLc_r = rp > rg ? 1.f : 0.f;
        const float uok = U >= eps ? 1.f : 0.f, lok = Lc >= eps ? 1.f : 0.f;
        // d(I/Uc): dI/Uc - I/Uc^2 dUc ; dU/dlp = 1 - dI_l
        const float a = 1.f / Uc, b = I / (Uc * Uc);
        const float diou_l = dI_l * a - b * uok * (1.f - dI_l), diou_r = dI_r * a - b * uok * (1.f - dI_r);
        // d(z^2) = 2 z (d rho / Lcc - rho / Lcc^2 dLcc)
        const float dz_l = (-0.5f) / Lcc - rho / (Lcc * Lcc) * lok * dLc_l;
        const float dz_r = (0.5f) / Lcc - rho / (Lcc * Lcc) * lok * dLc_r;
        dl = g * (-diou_l + 2.f * z * dz_l);
        dr = g * (-diou_r + 2.f * z * dz_r);
      }
    }
    if (dpred) { dpred[2 * r] = dl; dpred[2 * r + 1] = dr; }
  }
  if (loss) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc);
  }
}

// ---- one modality side of the detection training loss in a handful of launches (det scripts/train.py:222-349) -----------------
// The loop derives, per query row, validity flags and weights from the IoU / offset targets the labelling kernel wrote
// (valid_cls = iou >= 0, weight = iou < threshold ? 1 : iou, positive = offsets[r, 0] != inf), counts the positives, advances
// the EMA normaliser and divides: ~25 elementwise / reduction launches of a few microseconds each per side in eager torch.
// Here the flags are derived where they are used.  block = {loss, focal sum, DIoU sum, positives, normaliser used, 0, 0, 0}.
__global__ void det_zero_kernel(float* block) { if (threadIdx.x < 8) block[threadIdx.x] = 0.f; }

__device__ __forceinline__ bool det_positive(const float* __restrict__ off, int r) { return off[2 * r] != INFINITY; }

__global__ __launch_bounds__(256) void det_focal_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t, long long n, int C,
                                                            const float* __restrict__ iou, float thr, float alpha, float gamma,
                                                            float* __restrict__ block) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float u = iou[i / C];
    if (u >= 0.f) acc += (u < thr ? 1.f : u) * focal_term(x[i], t[i], alpha, gamma).loss;
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(block + 1, (part[0] + part[1]) + (part[2] + part[3]));
}

// positives counted, DIoU loss of the positive rows summed (forward: dpred == NULL), or its gradient written (backward)
__global__ __launch_bounds__(256) void det_rows_kernel(const float* __restrict__ pred, const float* __restrict__ off, int n, float eps,
                                                       float* __restrict__ block, const float* __restrict__ gout, float lambda_reg,
                                                       float* __restrict__ dpred) {
  float acc = 0.f, cnt = 0.f;
  const float g = dpred ? (gout ? gout[0] : 1.f) * lambda_reg / block[4] : 0.f;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    float dl = 0.f, dr = 0.f;
    if (det_positive(off, r)) {
      cnt += 1.f;
      const float lp = pred[2 * r], rp = pred[2 * r + 1], lg = off[2 * r], rg = off[2 * r + 1];
      const float lk = fminf(lp, lg), rk = fminf(rp, rg);
      const float I = rk + lk, U = (lp + rp) + (lg + rg) - I, Uc = fmaxf(U, eps);
      const float lc = fmaxf(lp, lg), rc = fmaxf(rp, rg), Lc = lc + rc, Lcc = fmaxf(Lc, eps);
      const float rho = 0.5f * (rp - lp - rg + lg), z = rho / Lcc;
      acc += 1.f - I / Uc + z * z;
      if (dpred) {   // (the compiled TorchScript form of the reference's autodiff: see diou_kernel)
        const float dI_l = lp < lg ? 1.f : 0.f, dI_r = rp < rg ? 1.f : 0.f;
        const float dLc_l = lp > lg ? 1.f : 0.f, dLc_r = rp > rg ? 1.f : 0.f;
        const float uok = U >= eps ? 1.f : 0.f, lok = Lc >= eps ? 1.f : 0.f;
        const float a = 1.f / Uc, b = I / (Uc * Uc);
        const float diou_l = dI_l * a - b * uok * (1.f - dI_l), diou_r = dI_r * a - b * uok * (1.f - dI_r);
        const float dz_l = (-0.5f) / Lcc - rho / (Lcc * Lcc) * lok * dLc_l;
        const float dz_r = (0.5f) / Lcc - rho / (Lcc * Lcc) * lok * dLc_r;
        dl = g * (-diou_l + 2.f * z * dz_l);
        dr = g * (-diou_r + 2.f * z * dz_r);
      }
    }
    if (dpred) { dpred[2 * r] = dl; dpred[2 * r + 1] = dr; }
  }
  if (!dpred) {
    acc = wave_sum(acc); cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { atomicAdd(block + 2, acc); atomicAdd(block + 3, cnt); }
  }
}

// normaliser <- momentum * normaliser + (1 - momentum) * max(positives, 1)   (train.py:230: a running value across steps and sides);
// loss = focal / (heads * normaliser) + lambda_reg * DIoU / normaliser   (no regression term without positives, train.py:277)
__global__ void det_finish_kernel(float* __restrict__ block, float* __restrict__ normaliser, float momentum, float lambda_reg, int nheads) {
  const float npos = block[3];
  const float nm = momentum * normaliser[0] + (1.f - momentum) * fmaxf(npos, 1.f);
  normaliser[0] = nm;
  block[4] = nm;
  block[0] = block[1] / ((float)nheads * nm) + (npos > 0.f ? lambda_reg * block[2] / nm : 0.f);
}

__global__ __launch_bounds__(256) void det_focal_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, long long n, int C,
                                                            const float* __restrict__ iou, float thr, float alpha, float gamma,
                                                            const float* __restrict__ block, const float* __restrict__ gout, int nheads,
                                                            float* __restrict__ dx) {
  const float g = (gout ? gout[0] : 1.f) / ((float)nheads * block[4]);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float u = iou[i / C];
    dx[i] = u >= 0.f ? g * (u < thr ? 1.f : u) * focal_term(x[i], t[i], alpha, gamma).dx : 0.f;
  }
}

}  // namespace

namespace {
// (a kernel, not hipMemsetAsync: a 4-byte memset node inside a captured HIP graph left the word unset on replay - the
//  detection training step's replayed loss read 1e32 while its logits and labels were right; tests/test_gpu_graph.py)
__global__ void zero_word_kernel(float* p) { *p = 0.f; }
}  // namespace

extern "C" {

int timhip_focal_loss_fwd(const float* logits, const float* targets, int rows, int C, const float* row_weights,
                          const uint8_t* row_valid, float alpha, float gamma, float* loss_sum, float* loss_elem,
                          void* stream) {
  if (!logits || !targets || !loss_sum || rows < 0 || C <= 0) return TIMHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(zero_word_kernel, dim3(1), dim3(1), 0, s, loss_sum);
  if (rows == 0) return TIMHIP_OK;
  const long long n = (long long)rows * C;
  const int blocks = (int)((n + 255) / 256 > 512 ? 512 : (n + 255) / 256);   // one atomic per block on a single address
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(blocks), dim3(256), 0, s, logits, targets, n, C, row_weights, row_valid, alpha,
                     gamma, loss_sum, loss_elem);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_focal_loss_bwd(const float* logits, const float* targets, int rows, int C, const float* row_weights,
                          const uint8_t* row_valid, float alpha, float gamma, const float* grad_out, float* dlogits,
                          void* stream) {
  if (!logits || !targets || !dlogits || rows < 0 || C <= 0) return TIMHIP_EINVAL;
  if (rows == 0) return TIMHIP_OK;
  const long long n = (long long)rows * C;
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, targets, n, C, row_weights,
                     row_valid, alpha, gamma, grad_out, dlogits);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_diou_1d(const float* pred_offsets, const float* target_offsets, int n, const uint8_t* row_valid, float eps,
                   const float* grad_out, float* loss_sum, float* dpred, void* stream) {
  if (!pred_offsets || !target_offsets || n < 0 || (!loss_sum && !dpred)) return TIMHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (loss_sum) hipLaunchKernelGGL(zero_word_kernel, dim3(1), dim3(1), 0, s, loss_sum);
  if (n == 0) return TIMHIP_OK;
  hipLaunchKernelGGL(diou_kernel, dim3((n + 255) / 256 > 256 ? 256 : (n + 255) / 256), dim3(256), 0, s, pred_offsets,
                     target_offsets, n, row_valid, eps, grad_out, loss_sum, dpred);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_det_side_loss_fwd(const float* const* logits, const float* const* targets, const int* C, int nheads, int rows,
                             const float* iou, const float* offsets, const float* reg_pred, float iou_threshold, float alpha,
                             float gamma, float eps, float lambda_reg, float momentum, float* normaliser, float* block,
                             void* stream) {
  if (!logits || !targets || !C || nheads < 1 || nheads > 4 || rows < 0 || !iou || !offsets || !reg_pred || !normaliser || !block)
    return TIMHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(det_zero_kernel, dim3(1), dim3(64), 0, s, block);
  if (rows > 0) {
    for (int k = 0; k < nheads; ++k) {
      if (!logits[k] || !targets[k] || C[k] <= 0) return TIMHIP_EINVAL;
      const long long n = (long long)rows * C[k];
      const int blocks = (int)((n + 255) / 256 > 512 ? 512 : (n + 255) / 256);   // one atomic per block on a single address
      hipLaunchKernelGGL(det_focal_fwd_kernel, dim3(blocks), dim3(256), 0, s, logits[k], targets[k], n, C[k], iou, iou_threshold,
                         alpha, gamma, block);
    }
    hipLaunchKernelGGL(det_rows_kernel, dim3((rows + 255) / 256 > 256 ? 256 : (rows + 255) / 256), dim3(256), 0, s, reg_pred, offsets,
                       rows, eps, block, (const float*)nullptr, lambda_reg, (float*)nullptr);
  }
  hipLaunchKernelGGL(det_finish_kernel, dim3(1), dim3(1), 0, s, block, normaliser, momentum, lambda_reg, nheads);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_det_side_loss_bwd(const float* const* logits, const float* const* targets, const int* C, int nheads, int rows,
                             const float* iou, const float* offsets, const float* reg_pred, float iou_threshold, float alpha,
                             float gamma, float eps, float lambda_reg, const float* block, const float* grad_out,
                             float* const* dlogits, float* dreg, void* stream) {
  if (!logits || !targets || !C || nheads < 1 || nheads > 4 || rows < 0 || !iou || !offsets || !reg_pred || !block || !dlogits)
    return TIMHIP_EINVAL;
  if (rows == 0) return TIMHIP_OK;
  hipStream_t s = (hipStream_t)stream;
  for (int k = 0; k < nheads; ++k) {
    if (!dlogits[k]) continue;
    const long long n = (long long)rows * C[k];
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(det_focal_bwd_kernel, dim3(blocks), dim3(256), 0, s, logits[k], targets[k], n, C[k], iou, iou_threshold,
                       alpha, gamma, block, grad_out, nheads, dlogits[k]);
  }
  if (dreg)
    hipLaunchKernelGGL(det_rows_kernel, dim3((rows + 255) / 256 > 256 ? 256 : (rows + 255) / 256), dim3(256), 0, s, reg_pred, offsets,
                       rows, eps, const_cast<float*>(block), grad_out, lambda_reg, dreg);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_ce_mixup_fwd(const float* logits, int rows, int C, int ld, const int64_t* target_a, const int64_t* target_b,
                        float lam, float smoothing, float* stats, float* accum, float* loss, void* stream) {
  if (!logits || !target_a || !stats || !accum || !loss || rows <= 0 || C <= 0 || ld < C) return TIMHIP_EINVAL;
  if (smoothing < 0.f || smoothing >= 1.f) return TIMHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (C >= 1024)
    hipLaunchKernelGGL(ce_rows_kernel<4>, dim3(rows), dim3(256), 0, s, logits, rows, C, ld, (const long long*)target_a,
                       (const long long*)target_b, smoothing, stats);
  else
    hipLaunchKernelGGL(ce_rows_kernel<1>, dim3((rows + 3) / 4), dim3(256), 0, s, logits, rows, C, ld,
                       (const long long*)target_a, (const long long*)target_b, smoothing, stats);
  TIM_CHECK_LAUNCH();
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, s, stats, rows, lam, accum, loss);
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
  DISPATCH_T(precision, hipLaunchKernelGGL(drloc_gather_kernel<T>, dim3(n * m, 2), dim3(128), 0, (hipStream_t)stream, x1, x2,
                                           (long long)batch_stride, (long long)row_stride, l, D, (const long long*)pos1,
                                           (const long long*)pos2, m, (T*)out, ld));
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
