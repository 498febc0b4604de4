// Live timing of the GEMM family for bench.py's roofline object: while armed, every NT / TN GEMM launch of at least
// `min_flops` algorithmic FLOPs is bracketed by two HIP events recorded on the stream it is launched on; stop() returns the
// summed durations and FLOPs.  Not part of the compute path: one branch on an atomic flag per launch when not armed.
#include <atomic>
#include <mutex>
#include <vector>
#include "common.h"

namespace {
std::mutex g_mu;
std::atomic<int> g_armed{0};
std::vector<hipEvent_t> g_ev;     // 2 per slot
std::vector<double> g_flops;
int g_cap = 0, g_n = 0;
double g_min_flops = 0.0;
}  // namespace

TimGemmScope::TimGemmScope(double flops, hipStream_t s) : slot(-1), stream(s) {
  if (!g_armed.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_armed.load() || flops < g_min_flops || g_n >= g_cap) return;
  slot = g_n++;
  g_flops[slot] = flops;
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
  for (auto& e : g_ev)
    if (hipEventCreate(&e) != hipSuccess) return TIMHIP_ELAUNCH;
  g_cap = capacity; g_n = 0; g_min_flops = min_flops;
  g_armed.store(1);
  return TIMHIP_OK;
}

int timhip_gemm_timing_stop(double* total_ms, double* total_flops, int* launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_armed.load()) return TIMHIP_EINVAL;
  g_armed.store(0);
  double ms = 0.0, fl = 0.0;
  int rc = TIMHIP_OK;
  for (int i = 0; i < g_n; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(g_ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&t, g_ev[2 * i], g_ev[2 * i + 1]) != hipSuccess)
      rc = TIMHIP_ELAUNCH;
    ms += t; fl += g_flops[i];
  }
  for (auto& e : g_ev) (void)hipEventDestroy(e);
  g_ev.clear();
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = g_n;
  g_n = 0; g_cap = 0;
  return rc;
}

}  // extern "C"
