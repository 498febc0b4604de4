/* TEST INFRASTRUCTURE ONLY - CPU restatement (plain C) of the reference's 1-D segment NMS, the only native component of
 * JacobChalk/TIM:  detection/eval_detection/csrc/nms_cpu.cpp:19-60 (nms_1d_cpu) and :69-170 (softnms_1d_cpu).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product path (tim_amd/nms.py ->
 * libtimhip.so) never does.  Pinned by tests/golden/nms_*.npz, which tests/golden/make_golden_nms.py generates by
 * compiling the reference's own nms_cpu.cpp in the build container and running it on the same seeded inputs.
 *
 *   gcc -O2 -shared -fPIC -o oracle/libnms_oracle.so oracle/nms_oracle.c -lm      (oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* nms_cpu.cpp:19-60.  `order` = indices sorted by descending score, supplied by the caller (the reference takes them from
 * torch's sort, whose order among equal scores is unspecified); keep[] receives the kept ORIGINAL indices in that order.
 * returns the number kept. */
int64_t nms_1d_oracle(const float* segs, const int64_t* order, int64_t n, float iou_threshold, int64_t* keep) {
  if (n == 0) return 0;
  unsigned char* select = (unsigned char*)malloc((size_t)n);
  for (int64_t k = 0; k < n; ++k) select[k] = 1;
  for (int64_t _i = 0; _i < n; ++_i) {
    if (!select[_i]) continue;
    const int64_t i = order[_i];
    const float ix1 = segs[2 * i], ix2 = segs[2 * i + 1];
    const float iarea = (ix2 - ix1) + 1e-6f;   /* nms_cpu.cpp:26: float tensor + scalar = float add */
    for (int64_t _j = _i + 1; _j < n; ++_j) {
      if (!select[_j]) continue;
      const int64_t j = order[_j];
      const float xx1 = fmaxf(ix1, segs[2 * j]), xx2 = fminf(ix2, segs[2 * j + 1]);
      const float inter = fmaxf(0.f, xx2 - xx1);
      const float jarea = (segs[2 * j + 1] - segs[2 * j]) + 1e-6f;
      const float ovr = inter / (iarea + jarea - inter);
      if (ovr >= iou_threshold) select[_j] = 0;
    }
  }
  int64_t m = 0;
  for (int64_t k = 0; k < n; ++k)
    if (select[k]) keep[m++] = order[k];
  free(select);
  return m;
}

/* nms_cpu.cpp:69-170.  dets [n,3] receives (x1, x2, score-at-selection) of the selected segments in selection order,
 * inds [n] their original indices; returns the number selected.  method: 0 vanilla, 1 linear, 2 gaussian. */
int64_t softnms_1d_oracle(const float* segs, const float* scores, int64_t n, float iou_threshold, float sigma,
                          float min_score, int method, float* dets, int64_t* inds) {
  if (n == 0) return 0;
  float* x1 = (float*)malloc(sizeof(float) * (size_t)n);
  float* x2 = (float*)malloc(sizeof(float) * (size_t)n);
  float* sc = (float*)malloc(sizeof(float) * (size_t)n);
  float* areas = (float*)malloc(sizeof(float) * (size_t)n);
  for (int64_t k = 0; k < n; ++k) {
    x1[k] = segs[2 * k]; x2[k] = segs[2 * k + 1]; sc[k] = scores[k];
    areas[k] = (x2[k] - x1[k]) + 1e-6f;
    inds[k] = k;
  }
  int64_t nsegs = n;
  for (int64_t i = 0; i < nsegs; ++i) {
    float max_score = sc[i];
    int64_t max_pos = i;
    for (int64_t pos = i + 1; pos < nsegs; ++pos)
      if (max_score < sc[pos]) { max_score = sc[pos]; max_pos = pos; }
    const float ix1 = dets[i * 3 + 0] = x1[max_pos];
    const float ix2 = dets[i * 3 + 1] = x2[max_pos];
    const float iscore = dets[i * 3 + 2] = sc[max_pos];
    const float iarea = areas[max_pos];
    const int64_t iind = inds[max_pos];
    x1[max_pos] = x1[i]; x2[max_pos] = x2[i]; sc[max_pos] = sc[i]; areas[max_pos] = areas[i]; inds[max_pos] = inds[i];
    x1[i] = ix1; x2[i] = ix2; sc[i] = iscore; areas[i] = iarea; inds[i] = iind;
    int64_t pos = i + 1;
    while (pos < nsegs) {
      const float xx1 = fmaxf(ix1, x1[pos]), xx2 = fminf(ix2, x2[pos]);
      const float inter = fmaxf(0.f, xx2 - xx1);
      const float ovr = inter / (iarea + areas[pos] - inter);
      float weight = 1.f;
      if (method == 0) { if (ovr >= iou_threshold) weight = 0.f; }
      else if (method == 1) { if (ovr >= iou_threshold) weight = 1.f - ovr; }
      else if (method == 2) { weight = expf(-(ovr * ovr) / sigma); }
      sc[pos] *= weight;
      if (sc[pos] < min_score) {
        x1[pos] = x1[nsegs - 1]; x2[pos] = x2[nsegs - 1]; sc[pos] = sc[nsegs - 1]; areas[pos] = areas[nsegs - 1];
        inds[pos] = inds[nsegs - 1];
        nsegs = nsegs - 1;
        pos = pos - 1;
      }
      pos = pos + 1;
    }
  }
  free(x1); free(x2); free(sc); free(areas);
  return nsegs;
}
