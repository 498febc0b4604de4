// Sliding-window batch assembly on the device: the GPU counterpart of
// recognition/time_interval_machine/datasets/sliding_window.py:341-421 (__getitem__) + the default collate, for feature
// stores that are resident in HBM (a whole EPIC-100 feature set is tens of GB; the MI355X has 288 GB).  Pure HBM streams:
//   * timhip_window_gather : out[b, j, :] = feats[(video_row0[w_b] + feat_idx[w_b, j]) * num_aug + aug[b, j], :]
//                            (self.v_feats[video_id][feat_indices, v_aug_indices], :358,370)
//   * timhip_window_times  : times[b] = clamp((cat(v_feat_times[idx,:2], a_feat_times[idx,:2], v_queries_pad, a_queries_pad)
//                            - start_sec) / window_size, min=0)   (:359-360,371-372,402-404)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void window_gather_kernel(const float* __restrict__ feats, int C, int num_aug,
                                                            const long long* __restrict__ video_row0,
                                                            const int* __restrict__ feat_idx, int nf,
                                                            const int* __restrict__ win, const int* __restrict__ aug,
                                                            float* __restrict__ out) {
  const int r = blockIdx.x;            // b * nf + j
  const int b = r / nf, j = r % nf;
  const int w = win[b];
  const long long row = (video_row0[w] + feat_idx[(size_t)w * nf + j]) * num_aug + (aug ? aug[r] : 0);
  const float* src = feats + (size_t)row * C;
  float* dst = out + (size_t)r * C;
  if ((C & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    for (int c = threadIdx.x * 4; c < C; c += 256 * 4) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
  } else {
    for (int c = threadIdx.x; c < C; c += 256) dst[c] = src[c];
  }
}

// one block per batch row; T = nfv + nfa + maxv + maxa rows of (start, end)
__global__ void window_times_kernel(const float* __restrict__ v_ft, int v_ld, const long long* __restrict__ v_row0,
                                    const float* __restrict__ a_ft, int a_ld, const long long* __restrict__ a_row0,
                                    const int* __restrict__ feat_idx, int nf, const int* __restrict__ win,
                                    const float* __restrict__ v_q, int maxv, const float* __restrict__ a_q, int maxa,
                                    const float* __restrict__ start_sec, float window_size, float* __restrict__ times) {
  const int b = blockIdx.x, w = win[b];
  const int nfv = v_ft ? nf : 0, nfa = a_ft ? nf : 0, T = nfv + nfa + maxv + maxa;
  const float st = start_sec[w];
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float s, e;
    if (t < nfv) {
      const float* p = v_ft + (size_t)(v_row0[w] + feat_idx[(size_t)w * nf + t]) * v_ld;
      s = p[0]; e = p[1];
    } else if (t < nfv + nfa) {
      const float* p = a_ft + (size_t)(a_row0[w] + feat_idx[(size_t)w * nf + (t - nfv)]) * a_ld;
      s = p[0]; e = p[1];
    } else if (t < nfv + nfa + maxv) {
      const float* p = v_q + ((size_t)w * maxv + (t - nfv - nfa)) * 2;
      s = p[0]; e = p[1];
    } else {
      const float* p = a_q + ((size_t)w * maxa + (t - nfv - nfa - maxv)) * 2;
      s = p[0]; e = p[1];
    }
    float* o = times + ((size_t)b * T + t) * 2;
    o[0] = fmaxf((s - st) / window_size, 0.f);
    o[1] = fmaxf((e - st) / window_size, 0.f);
  }
}

}  // namespace

extern "C" {

int timhip_window_gather(const float* feats, int C, int num_aug, const int64_t* video_row0, const int32_t* feat_indices,
                         int num_feats, const int32_t* windows, int B, const int32_t* aug_indices, float* out,
                         void* stream) {
  if (!feats || !video_row0 || !feat_indices || !windows || !out || C <= 0 || num_aug <= 0 || num_feats <= 0 || B < 0)
    return TIMHIP_EINVAL;
  if (B == 0) return TIMHIP_OK;
  hipLaunchKernelGGL(window_gather_kernel, dim3(B * num_feats), dim3(256), 0, (hipStream_t)stream, feats, C, num_aug,
                     (const long long*)video_row0, feat_indices, num_feats, windows, aug_indices, out);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

int timhip_window_times(const float* v_feat_times, int v_ld, const int64_t* v_row0, const float* a_feat_times, int a_ld,
                        const int64_t* a_row0, const int32_t* feat_indices, int num_feats, const int32_t* windows, int B,
                        const float* v_queries, int max_v, const float* a_queries, int max_a, const float* start_sec,
                        float window_size, float* times, void* stream) {
  if (!feat_indices || !windows || !start_sec || !times || num_feats <= 0 || B < 0 || max_v < 0 || max_a < 0 ||
      !(window_size > 0.f))
    return TIMHIP_EINVAL;
  if ((v_feat_times && (!v_row0 || v_ld < 2)) || (a_feat_times && (!a_row0 || a_ld < 2)) || (max_v > 0 && !v_queries) ||
      (max_a > 0 && !a_queries))
    return TIMHIP_EINVAL;
  if (B == 0) return TIMHIP_OK;
  hipLaunchKernelGGL(window_times_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, v_feat_times, v_ld,
                     (const long long*)v_row0, a_feat_times, a_ld, (const long long*)a_row0, feat_indices, num_feats, windows,
                     v_queries, max_v, a_queries, max_a, start_sec, window_size, times);
  TIM_CHECK_LAUNCH();
  return TIMHIP_OK;
}

}  // extern "C"
