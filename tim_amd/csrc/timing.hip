// Live timing of the GEMM family for bench.py's roofline object: while armed, every NT / TN GEMM launch of at least
// `min_flops` algorithmic FLOPs is bracketed by two HIP events recorded on the stream it is launched on; stop() returns the
// summed durations and FLOPs.  Not part of the compute path: one branch on an atomic flag per launch when not armed.
// Round 6: two more families ride along - the attention launches (family 1) and the LayerNorm launches (family 2), whose "work"
// figure is their algorithmic BYTES (they are fabric-bound) - so that the bench line carries the non-GEMM kernel time of the
// step it brackets (timhip_timing_stop_families).
#include <atomic>
#include <mutex>
#include <vector>
#include "common.h"

namespace {
std::mutex g_mu;
std::atomic<int> g_armed{0};
std::vector<hipEvent_t> g_ev;     // 2 per slot
std::vector<double> g_flops;
std::vector<int> g_fam;
int g_cap = 0, g_n = 0;
double g_min_flops = 0.0;
}  // namespace

TimGemmScope::TimGemmScope(double flops, hipStream_t s, int family) : slot(-1), stream(s) {
  if (!g_armed.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_armed.load() || (family == 0 && flops < g_min_flops) || g_n >= g_cap) return;
  slot = g_n++;
  g_flops[slot] = flops;
  g_fam[slot] = family;
  (void)hipEventRecord(g_ev[2 * slot], stream);
}
TimGemmScope::~TimGemmScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot < (int)g_flops.size()) (void)hipEventRecord(g_ev[2 * slot + 1], stream);
}

extern "C" {

int timhip_gemm_timing_start(int capacity, double min_flops) {
  if (capacity <= 0) return TIMHIP_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_armed.load()) return TIMHIP_EINVAL;
  g_ev.resize(2 * (size_t)capacity);
  g_flops.assign((size_t)capacity, 0.0);
  g_fam.assign((size_t)capacity, 0);
  for (auto& e : g_ev)
    if (hipEventCreate(&e) != hipSuccess) return TIMHIP_ELAUNCH;
  g_cap = capacity; g_n = 0; g_min_flops = min_flops;
  g_armed.store(1);
  return TIMHIP_OK;
}

// ms[f], work[f], launches[f] for f = 0 (GEMM: FLOPs), 1 (attention: algorithmic bytes), 2 (LayerNorm: algorithmic bytes)
int timhip_timing_stop_families(double* ms3, double* work3, int* launches3) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_armed.load()) return TIMHIP_EINVAL;
  g_armed.store(0);
  double ms[3] = {0.0, 0.0, 0.0}, fl[3] = {0.0, 0.0, 0.0};
  int nl[3] = {0, 0, 0};
  int rc = TIMHIP_OK;
  for (int i = 0; i < g_n; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(g_ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&t, g_ev[2 * i], g_ev[2 * i + 1]) != hipSuccess)
      rc = TIMHIP_ELAUNCH;
    const int f = g_fam[i] >= 0 && g_fam[i] < 3 ? g_fam[i] : 0;
    ms[f] += t; fl[f] += g_flops[i]; nl[f] += 1;
  }
  for (auto& e : g_ev) (void)hipEventDestroy(e);
  g_ev.clear();
  for (int f = 0; f < 3; ++f) {
    if (ms3) ms3[f] = ms[f];
    if (work3) work3[f] = fl[f];
    if (launches3) launches3[f] = nl[f];
  }
  g_n = 0; g_cap = 0;
  return rc;
}

int timhip_gemm_timing_stop(double* total_ms, double* total_flops, int* launches) {
  double ms[3], fl[3];
  int nl[3];
  const int rc = timhip_timing_stop_families(ms, fl, nl);
  if (rc == TIMHIP_EINVAL) return rc;
  if (total_ms) *total_ms = ms[0];
  if (total_flops) *total_flops = fl[0];
  if (launches) *launches = nl[0];
  return rc;
}

}  // extern "C"
