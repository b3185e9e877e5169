// capi_internal.hpp -- what the three translation units behind include/tsd_hip.h share (capi.hip: context, grid and the unfused
// entry points; capi_io.hip: tile / text / map I/O and profiling; capi_scan.hip: the fused, split and batched scan paths).
#pragma once
#include "tsd_ctx.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

namespace tsd {

extern const char* const kKernelNames[7];           // the kernels tsd_profile_select knows, in profile_mask's bit order
hipEvent_t pool_get(tsd_ctx* ctx);                  // a timing event from the context's pool
char* stage_acquire(tsd_ctx* ctx, int* slot_out);   // next pinned staging slot; waits for the copy that last used it
double distance_filter_multiplier(double maxdist, double mindist, int icp_iterations);
void fill_icp_args(IcpArgs& a, const double pose33[9], const tsd_icp_params* p);
void fill_raycast_args(const tsd_ctx* ctx, RaycastArgs& a, const double pose33[9], int beams, double min_range, double max_range);
void copy_icp_result(const IcpResultDev* h, tsd_icp_result* r);
void fill_stats(tsd_ctx* ctx, const unsigned long long t[7], tsd_push_stats* out);
int read_last_push_stats(tsd_ctx* ctx, tsd_push_stats* out);
int read_total_stats(tsd_ctx* ctx, tsd_push_stats* out, int64_t* pushes, bool reset);
bool host_saw_event(hipEvent_t ev, int us);         // true once `ev` has completed; polls for at most ~`us` microseconds
int wait_for_readers(tsd_ctx* ctx);                 // grid writes on the context's stream go behind the ray casts of the split path

inline unsigned long long now_ns()
{
  return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// TSD_CONC_TIMING=1: host time spent inside the split-scan calls (wall clock, summed over all threads), printed by
// tsd_destroy.  Diagnostic only.
struct ConcTiming {
  std::atomic<unsigned long long> ns[8];
  std::atomic<unsigned long long> n;
  bool on;
  ConcTiming() : on(getenv("TSD_CONC_TIMING") != nullptr) { for (auto& v : ns) v = 0; n = 0; }
};
extern ConcTiming g_conc_timing;
extern ConcTiming g_scan_timing;      // the same for tsd_scan (one robot): where the host time of a scan goes
extern ConcTiming g_stage_timing;     // ... and of the staging of a scan (acquire, host copy, hipMemcpyAsync, records, tables)
struct ConcLap {
  unsigned long long t;
  ConcLap() : t(g_conc_timing.on ? now_ns() : 0) {}
  void lap(int i) { if (g_conc_timing.on) { const unsigned long long u = now_ns(); g_conc_timing.ns[i] += u - t; t = u; } }
};
extern unsigned long long g_scan_lap_max[8];
extern unsigned long long g_scan_lap_max_at[8];
struct ScanLap {
  unsigned long long t;
  ScanLap() : t(g_scan_timing.on ? now_ns() : 0) {}
  void lap(int i)
  {
    if (!g_scan_timing.on) return;
    const unsigned long long u = now_ns();
    g_scan_timing.ns[i] += u - t;
    if (u - t > g_scan_lap_max[i]) { g_scan_lap_max[i] = u - t; g_scan_lap_max_at[i] = (unsigned long long)g_scan_timing.n; }
    t = u;
  }
};
extern unsigned long long g_scan_last_return;


}  // namespace tsd
