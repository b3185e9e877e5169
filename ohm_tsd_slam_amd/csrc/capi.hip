// capi.hip -- the extern "C" entry points of include/tsd_hip.h: context life cycle, staging of the
// scan through a ring of pinned buffers, and the launch order of the kernels on the ctx stream.
// No CPU fall-back exists: every compute entry point needs a gfx950 device and fails loudly without.
#include "tsd_ctx.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <new>
#include <numeric>
#include <vector>

namespace tsd {

thread_local const LaunchTarget* g_launch_target = nullptr;

int set_error(tsd_ctx* ctx, int code, const char* what, hipError_t e)
{
  if (ctx) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof(buf), "%s", what);
    ctx->err = buf;
  }
  return code;
}

static const char* const kKernelNames[] = {"push_classify", "push_update", "push_halo", "raycast", "icp", "occupancy", "tsdpdf"};

bool kernel_is_timed(const tsd_ctx* ctx, const char* name)
{
  if (!ctx->profile) return false;
  for (unsigned i = 0; i < sizeof(kKernelNames) / sizeof(kKernelNames[0]); i++)
    if (std::strcmp(kKernelNames[i], name) == 0) return (ctx->profile_mask >> i) & 1u;
  return false;
}

static hipEvent_t pool_get(tsd_ctx* ctx)
{
  if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  // (timing only: no system-scope fence when the event completes -- the default makes every sampled kernel write its dirty lines back
  // to memory before the next one starts)
  if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr;
  return e;
}

ScopedKernelTimer::ScopedKernelTimer(tsd_ctx* c, const char* n, bool around_) : ctx(c), name(n), around(around_)
{
  if (!kernel_is_timed(ctx, n)) return;
  std::lock_guard<std::mutex> lk(ctx->misc_mutex);
  unsigned every = ctx->profile_every;
  for (unsigned i = 0; i < sizeof(kKernelNames) / sizeof(kKernelNames[0]); i++)
    if (ctx->profile_every_k[i] && std::strcmp(kKernelNames[i], n) == 0) every = ctx->profile_every_k[i];
  if (every > 1 && (ctx->timers[n].tick++ % every) != 0) return;   // every n-th launch of THIS kernel
  a = pool_get(ctx); b = pool_get(ctx);
  if (!a || !b) { a = b = nullptr; return; }
  if (around) hipEventRecord(a, ctx->stream);
}
ScopedKernelTimer::~ScopedKernelTimer()
{
  if (!a) return;
  if (around) hipEventRecord(b, ctx->stream);
  std::lock_guard<std::mutex> lk(ctx->misc_mutex);
  ctx->timers[name].pending.emplace_back(a, b);
}
void drain_timers(tsd_ctx* ctx)
{
  std::lock_guard<std::mutex> lk(ctx->misc_mutex);
  for (auto& kv : ctx->timers) {
    for (auto& pr : kv.second.pending) {
      float ms = 0.f;
      if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        kv.second.total_ms += (double)ms;
        kv.second.sum_sq += (double)ms * (double)ms;
        if ((double)ms < kv.second.min_ms) kv.second.min_ms = (double)ms;
        if ((double)ms > kv.second.max_ms) kv.second.max_ms = (double)ms;
        kv.second.launches++;
        if (kv.second.samples.size() < KERNEL_TIMER_SAMPLES) kv.second.samples.push_back(ms);
      }
      ctx->event_pool.push_back(pr.first);     // recycled: no event creation in steady state
      ctx->event_pool.push_back(pr.second);
    }
    kv.second.pending.clear();
  }
}

// obvious::Matrix::invert (gsl/Matrix.cpp:168-179): LU with partial pivoting, inverse by column solves
void mat3_inv(const double A[9], double Ainv[9])
{
  double lu[9];
  int perm[3] = {0, 1, 2};
  std::memcpy(lu, A, sizeof(lu));
  for (int j = 0; j < 3; j++) {
    int piv = j;
    double best = std::fabs(lu[3 * j + j]);
    for (int i = j + 1; i < 3; i++)
      if (std::fabs(lu[3 * i + j]) > best) { best = std::fabs(lu[3 * i + j]); piv = i; }
    if (piv != j) {
      for (int k = 0; k < 3; k++) std::swap(lu[3 * j + k], lu[3 * piv + k]);
      std::swap(perm[j], perm[piv]);
    }
    for (int i = j + 1; i < 3; i++) {
      lu[3 * i + j] = lu[3 * i + j] / lu[3 * j + j];
      for (int k = j + 1; k < 3; k++) lu[3 * i + k] -= lu[3 * i + j] * lu[3 * j + k];
    }
  }
  for (int c = 0; c < 3; c++) {
    double x[3];
    for (int i = 0; i < 3; i++) x[i] = (perm[i] == c) ? 1.0 : 0.0;
    for (int i = 1; i < 3; i++)
      for (int k = 0; k < i; k++) x[i] -= lu[3 * i + k] * x[k];
    for (int i = 2; i >= 0; i--) {
      for (int k = i + 1; k < 3; k++) x[i] -= lu[3 * i + k] * x[k];
      x[i] = x[i] / lu[3 * i + i];
    }
    for (int i = 0; i < 3; i++) Ainv[3 * i + c] = x[i];
  }
}

// next pinned staging slot; waits for the copy that last used it
static char* stage_acquire(tsd_ctx* ctx, int* slot_out)
{
  const int s = ctx->slot;
  ctx->slot = (ctx->slot + 1) % tsd_ctx::kSlots;
  hipEventSynchronize(ctx->stage_ev[s]);
  *slot_out = s;
  return ctx->h_stage[s];
}

// DistanceFilter ctor as called from ThreadLocalize.cpp:212 (int -> unsigned conversion of
// icp_iterations - 10; DistanceFilter.cpp:11-20)
static double distance_filter_multiplier(double maxdist, double mindist, int icp_iterations)
{
  unsigned int iterations = (unsigned int)(icp_iterations - 10);
  double it = (double)(iterations - 1);
  if (iterations < 1) it = 1.0;
  return std::pow((mindist / maxdist), 1.0 / it);
}

static void fill_icp_args(IcpArgs& a, const double pose33[9], const tsd_icp_params* p)
{
  for (int i = 0; i < 6; i++) a.P[i] = pose33[i];
  a.min_x = p->min_x; a.max_x = p->max_x; a.min_y = p->min_y; a.max_y = p->max_y;
  a.thr0 = p->dist_filter_max * p->dist_filter_max;
  a.min_sqr = p->dist_filter_min * p->dist_filter_min;
  a.multiplier = distance_filter_multiplier(p->dist_filter_max, p->dist_filter_min, p->iterations);
  a.iterations = p->iterations;
  a.n_model = a.n_scene = a.beams = 0;
  a.ccw = 1; a.estimator = p->estimator;
  // Tinit of Icp::iterate (rows 0, 1 of the 3x3): the identity unless the caller hands a pre-registration result over
  const double I6[6] = {1, 0, 0, 0, 1, 0};
  for (int i = 0; i < 6; i++) a.Tinit[i] = p->use_t_init ? p->t_init[i] : I6[i];
  a.Tinit_dev = nullptr;
}

static void fill_raycast_args(const tsd_ctx* ctx, RaycastArgs& a, const double pose33[9], int beams,
                              double min_range, double max_range)
{
  double Pi[9];
  mat3_inv(pose33, Pi);
  for (int i = 0; i < 6; i++) a.Pi[i] = Pi[i];
  a.trx = pose33[2]; a.try_ = pose33[5];
  const GridDev& g = ctx->grid;
  // TsdGrid::isInsideGrid (TsdGrid.h:342-347) -> RayCastPolar2D.cpp:128-146
  if (a.trx > g.min_x && a.trx < g.max_x && a.try_ > g.min_y && a.try_ < g.max_y) {
    a.gxmin = -10e9; a.gymin = -10e9; a.gxmax = 10e9; a.gymax = 10e9;
  } else {
    a.gxmin = 10e9; a.gymin = 10e9; a.gxmax = -10e9; a.gymax = -10e9;
  }
  a.idx_min = min_range / g.cs;
  a.idx_max = max_range / g.cs;
  a.beams = beams;
}

static void copy_icp_result(const IcpResultDev* h, tsd_icp_result* r)
{
  for (int i = 0; i < 9; i++) r->T[i] = h->T[i];
  r->rms = h->rms; r->pairs = h->pairs; r->iterations = h->iterations; r->state = h->state;
  r->n_model = h->n_model; r->n_scene = h->n_scene; r->reserved = h->reserved;
}

// Statistics of the pushes: every tile keeps a record of the last push (k_push_tiles) and running totals
// (k_push_halo); they are summed on the host when somebody asks -- tests and the end of a benchmark.
static void fill_stats(tsd_ctx* ctx, const unsigned long long t[7], tsd_push_stats* out)
{
  out->cells_updated = (int64_t)t[0];
  out->cells_visited = 1024ll * (int64_t)t[2];     // every UPDATE tile back-projects its 32x32 cells
  out->tiles_total = ctx->grid.tiles;
  out->tiles_range_pass = (int32_t)t[1]; out->tiles_update = (int32_t)t[2]; out->tiles_new = (int32_t)t[3];
  out->tiles_new_from_empty = (int32_t)t[4]; out->tiles_emptied_init = (int32_t)t[5];
  out->tiles_emptied_uninit = (int32_t)t[6];
}

static int read_last_push_stats(tsd_ctx* ctx, tsd_push_stats* out)
{
  const size_t T = (size_t)ctx->grid.tiles;
  std::vector<uint32_t> rec(T);
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(rec.data(), ctx->d_tile_rec, T * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long t[7] = {0, 0, 0, 0, 0, 0, 0};
  for (size_t p = 0; p < T; p++) {
    const uint32_t r = rec[p];
    if (!r) continue;
    t[0] += r >> 8;
    for (int k = 1; k < 7; k++) t[k] += (r >> (k - 1)) & 1u;
  }
  fill_stats(ctx, t, out);
  return TSD_OK;
}

static int read_total_stats(tsd_ctx* ctx, tsd_push_stats* out, int64_t* pushes, bool reset)
{
  const size_t T = (size_t)ctx->grid.tiles;
  std::vector<uint32_t> tot(T * 8);
  unsigned long long np[2] = {0, 0};
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(tot.data(), ctx->d_tile_totals, T * 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(np, ctx->d_pushes, sizeof(np), hipMemcpyDeviceToHost, ctx->stream));
  if (reset) {
    TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_tile_totals, 0, T * 8 * sizeof(uint32_t), ctx->stream));
    TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_pushes, 0, sizeof(np), ctx->stream));
  }
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (np[1] != 0) return set_error(ctx, TSD_E_HIP, "push launch window missed the sensor position", hipSuccess);
  unsigned long long t[7] = {0, 0, 0, 0, 0, 0, 0};
  for (size_t p = 0; p < T; p++)
    for (int k = 0; k < 7; k++) t[k] += tot[p * 8 + k];
  if (out) fill_stats(ctx, t, out);
  if (pushes) *pushes = (int64_t)np[0];
  return TSD_OK;
}

// true once `ev` has completed; polls for at most ~`us` microseconds
static bool host_saw_event(hipEvent_t ev, int us)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (hipEventQuery(ev) == hipSuccess) return true;
    (void)hipGetLastError();                 // hipErrorNotReady is sticky as "last error": do not leave it for other libraries
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(us)) return false;
  }
}

// TSD_CONC_TIMING=1: host time spent inside the split-scan calls (wall clock, summed over all threads), printed by
// tsd_destroy.  Diagnostic only.
struct ConcTiming {
  std::atomic<unsigned long long> ns[8];
  std::atomic<unsigned long long> n;
  bool on;
  ConcTiming() : on(getenv("TSD_CONC_TIMING") != nullptr) { for (auto& v : ns) v = 0; n = 0; }
};
static ConcTiming g_conc_timing;
static ConcTiming g_scan_timing;      // the same for tsd_scan (one robot): where the host time of a scan goes
static ConcTiming g_stage_timing;     // ... and of the staging of a scan (acquire, host copy, hipMemcpyAsync, records, tables)
static inline unsigned long long now_ns()
{
  return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct ConcLap {
  unsigned long long t;
  ConcLap() : t(g_conc_timing.on ? now_ns() : 0) {}
  void lap(int i) { if (g_conc_timing.on) { const unsigned long long u = now_ns(); g_conc_timing.ns[i] += u - t; t = u; } }
};
static unsigned long long g_scan_lap_max[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static unsigned long long g_scan_lap_max_at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
static unsigned long long g_scan_last_return = 0;

// every grid WRITE enqueued on the context's stream goes behind the ray casts the concurrent multi-robot path has in
// flight on the sensors' own streams (no-op without such sensors)
static int wait_for_readers(tsd_ctx* ctx)
{
  // (caller holds ctx->order_mutex.)  Ray casts ticketed after the last grid write that waited are the ones to wait
  // for: earlier ones are ordered through that write already.  A sensor whose thread has taken its ticket but not yet
  // issued the event record (a few microseconds) is waited for on the host.
  for (tsd_sensor* t : ctx->sensors) {
    if (!t->rc_event_valid || t->rc_ticket <= ctx->last_push_ticket) continue;
    while (!__atomic_load_n(&t->rc_recorded, __ATOMIC_ACQUIRE)) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, t->ev_rc_done, 0));
  }
  // (the ray casts of the batched path run on the context's stream itself: ordered by it)
  ctx->last_push_ticket = ctx->ticket;
  return TSD_OK;
}

// Asynchronous mapping (tsd_sensor_set_async_mapping): the fused scan's push may still be running on the push stream.  Everything
// else that goes to the context's stream -- every entry point but the fused scan's own -- is ordered behind it first.
int drain_async_push(tsd_ctx* ctx)
{
  if (ctx->async_pending) {
    TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_async_push, 0));
    ctx->async_pending = false;
  }
  return TSD_OK;
}

}  // namespace tsd

using namespace tsd;

extern "C" {

int tsd_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

tsd_ctx* tsd_create(int device, int map_size_log2, double cell_size, double max_trunc)
{
  if (map_size_log2 < 5 || map_size_log2 > 15 || !(cell_size > 0.0)) return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    fprintf(stderr, "tsd_create: no usable HIP device (requested %d of %d); this library has no CPU path\n", device, ndev);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  tsd_ctx* ctx = new (std::nothrow) tsd_ctx();
  if (!ctx) return nullptr;
  ctx->device = device;
  ctx->map_log2 = map_size_log2;
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus; }
  // (tests: a device that shows fewer compute units -- a compute partition, a CU-masked queue -- sizes the resident grids smaller)
  if (const char* e = std::getenv("TSD_DEBUG_N_CUS")) { const int v = std::atoi(e); if (v > 0 && v < ctx->n_cus) ctx->n_cus = v; }
  GridDev& g = ctx->grid;
  // TsdGrid::init (TsdGrid.cpp:112-169)
  g.N = 1 << map_size_log2;
  g.PX = g.N / TILE_DIM;
  g.tiles = g.PX * g.PX;
  g.cs = cell_size;
  g.inv_cs = 1.0 / cell_size;
  g.min_x = 0.0; g.max_x = ((double)g.N + 0.5) * cell_size;
  g.min_y = 0.0; g.max_y = ((double)g.N + 0.5) * cell_size;
  // setMaxTruncation (TsdGrid.cpp:206-215)
  double val = max_trunc;
  if (val < 2 * cell_size) val = 2 * cell_size;
  g.max_trunc = val;

  bool ok = true;
  auto A = [&](hipError_t e) { if (e != hipSuccess) { if (ok) fprintf(stderr, "tsd_create: %s\n", hipGetErrorString(e)); ok = false; } };
  A(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  const size_t T = (size_t)g.tiles;
  A(hipMalloc(&g.flags, T));
  A(hipMalloc(&g.init_weight, T * sizeof(double)));
  A(hipMalloc(&g.tsd, T * TILE_STRIDE * sizeof(tsd_cell_t)));
  A(hipMalloc(&g.weight, T * TILE_STRIDE * sizeof(w_cell_t)));
  A(hipMalloc(&g.negmask, T * sizeof(unsigned long long)));
  A(hipMalloc(&ctx->d_rmq2[0], push_rmq_bytes(TSD_MAX_BEAMS)));
  A(hipMalloc(&ctx->d_rmq2[1], push_rmq_bytes(TSD_MAX_BEAMS)));
  ctx->d_rmq = ctx->d_rmq2[0];
  A(hipEventCreateWithFlags(&ctx->ev_h2d, hipEventDisableTiming));
  A(hipMalloc(&ctx->d_tile_rec, T * sizeof(uint32_t)));
  A(hipMalloc(&ctx->d_dirty, T));
  A(hipMalloc(&ctx->d_tile_totals, T * 8 * sizeof(uint32_t)));
  A(hipMalloc(&ctx->d_pushes, 2 * sizeof(unsigned long long)));
  A(hipMalloc(&ctx->d_list, T * sizeof(uint32_t)));
  A(hipMalloc(&ctx->d_list_h, T * sizeof(uint32_t)));
  A(hipMalloc(&ctx->d_list_aux, T * push_list_aux_bytes()));
  A(hipMalloc(&ctx->d_push_args, sizeof(PushArgs)));
  A(hipMalloc(&ctx->d_list_cnt, push_list_cnt_bytes()));
  A(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  A(hipEventCreateWithFlags(&ctx->ev_tables, hipEventDisableTiming));
  ctx->stage_bytes = (size_t)TSD_MAX_BEAMS * (8 * 5 + 1) + 256;   // ranges + 2x rays(2) + mask; >= icp staging (80 KB)
  for (int s = 0; s < tsd_ctx::kSlots; s++) {
    A(hipHostMalloc(&ctx->h_stage[s], ctx->stage_bytes, hipHostMallocDefault));
    A(hipEventCreateWithFlags(&ctx->stage_ev[s], hipEventDisableTiming));
  }
  A(hipMalloc(&ctx->d_ranges, TSD_MAX_BEAMS * sizeof(double)));
  A(hipMalloc(&ctx->d_mask, TSD_MAX_BEAMS));
  A(hipMalloc(&ctx->d_rays, 2 * TSD_MAX_BEAMS * sizeof(double)));
  A(hipMalloc(&ctx->d_rays_local, 2 * TSD_MAX_BEAMS * sizeof(double)));
  // the ray cast's three outputs in ONE block, laid out like tsd_raycast's host buffer: one device-to-host copy brings them back
  A(hipMalloc(&ctx->d_coords, (size_t)TSD_MAX_BEAMS * 33));
  if (ctx->d_coords) {
    ctx->d_normals = ctx->d_coords + 2 * TSD_MAX_BEAMS;
    ctx->d_mask_m = reinterpret_cast<uint8_t*>(ctx->d_coords) + (size_t)TSD_MAX_BEAMS * 32;
  }
  A(hipMalloc(&ctx->d_mnormals, 2 * TSD_MAX_ICP_POINTS * sizeof(double)));
  A(hipMalloc(&ctx->d_model, 2 * TSD_MAX_ICP_POINTS * sizeof(double)));
  A(hipMalloc(&ctx->d_scene, 2 * TSD_MAX_ICP_POINTS * sizeof(double)));
  A(hipMalloc(&ctx->d_morig, TSD_MAX_ICP_POINTS * sizeof(int)));
  A(hipMalloc(&ctx->d_start, TSD_MAX_ICP_POINTS * sizeof(int)));
  if (const char* e = std::getenv("TSD_ICP_SHAPE")) ctx->icp_shape = std::atoi(e);
  if (const char* e = std::getenv("TSD_ICP_HELPERS")) ctx->icp_helpers = std::atoi(e) != 0;
  A(hipMalloc(&ctx->d_icp_res, sizeof(IcpResultDev)));
  A(hipMalloc(&ctx->d_icp_seed, icp_seed_bytes(TSD_MAX_ICP_POINTS)));
  if (ok) A(hipMemsetAsync(ctx->d_icp_seed, 0, icp_seed_bytes(TSD_MAX_ICP_POINTS), ctx->stream));
  A(hipMalloc(&ctx->d_icp_trace, sizeof(double) * TSD_ICP_TRACE_STRIDE * TSD_ICP_TRACE_MAX * 2));     // (second half: scratch of the diagnostic stamp builds)
  // (never the NULL stream: HIP maps streams onto a few hardware queues in the order they are first used, and a null stream that
  // comes alive here takes one of them -- the two batch slots of the multi-robot path then share a queue and their registrations
  // run one after the other: 24.6 k -> 19.9 k scans/s with eight robots, measured in round 3)
  if (ok) A(hipMemsetAsync(ctx->d_icp_trace, 0, sizeof(double) * TSD_ICP_TRACE_STRIDE * TSD_ICP_TRACE_MAX * 2, ctx->stream));
  A(hipHostMalloc(&ctx->h_icp_res, sizeof(IcpResultDev), hipHostMallocDefault));
  ctx->h_out_bytes = (size_t)TSD_MAX_BEAMS * (8 * 4 + 1) + 256;
  A(hipHostMalloc(&ctx->h_out, ctx->h_out_bytes, hipHostMallocDefault));
  A(hipMalloc(&ctx->d_occ, (size_t)g.N * g.N));
  A(hipMalloc(&ctx->d_occ_count, sizeof(int)));
  A(hipMalloc(&ctx->d_occ_heads, occ_heads_bytes()));
  A(hipMemsetAsync(ctx->d_occ_heads, 0, occ_heads_bytes(), ctx->stream));
  A(hipMalloc(&ctx->d_occ_list, (T + 32) * sizeof(uint32_t)));
  A(hipEventCreateWithFlags(&ctx->ev_grid, hipEventDisableTiming));
  if (!ok) { tsd_destroy(ctx); return nullptr; }
  if (tsd_reset(ctx) != TSD_OK) { fprintf(stderr, "tsd_create: %s\n", ctx->err.c_str()); tsd_destroy(ctx); return nullptr; }
  return ctx;
}

void tsd_destroy(tsd_ctx* ctx)
{
  if (!ctx) return;
  if (g_stage_timing.on && g_stage_timing.n) {
    const double n = (double)g_stage_timing.n;
    static const char* names[8] = {"acquire", "host copy", "hipMemcpyAsync", "records", "tables launch", "-", "-", "-"};
    fprintf(stderr, "TSD_CONC_TIMING stagings %.0f; host us each:", n);
    for (int i = 0; i < 5; i++) fprintf(stderr, " %s %.1f", names[i], 1e-3 * (double)g_stage_timing.ns[i] / n);
    fprintf(stderr, "\n");
  }
  if (g_scan_timing.on && g_scan_timing.n) {
    const double n = (double)g_scan_timing.n;
    static const char* names[8] = {"caller (between calls)", "stage+copy+tables", "ray cast", "wait copy + icp launch", "push launches", "next ray cast", "wait result", "result"};
    fprintf(stderr, "TSD_CONC_TIMING tsd_scan calls %.0f; host us per scan:", n);
    for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.1f", names[i], 1e-3 * (double)g_scan_timing.ns[i] / n);
    fprintf(stderr, "\nTSD_CONC_TIMING longest single lap (us @ call):");
    for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.0f @%llu", names[i], 1e-3 * (double)g_scan_lap_max[i], g_scan_lap_max_at[i]);
    fprintf(stderr, "\n");
  }
  if (g_conc_timing.on && g_conc_timing.n) {
    const double n = (double)g_conc_timing.n;
    static const char* names[8] = {"begin:copy+tables", "begin:lock", "begin:ordered", "begin:raycast+icp", "wait", "finish:lock", "finish:push", "finish:result"};
    fprintf(stderr, "TSD_CONC_TIMING scans %.0f; host us per scan:", n);
    for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.1f", names[i], 1e-3 * (double)g_conc_timing.ns[i] / n);
    fprintf(stderr, "\n");
  }
  hipSetDevice(ctx->device);
  // sensors and batch slots that outlive their grid are detached: their own destroy calls then only free what is theirs
  for (tsd_batch* bt : ctx->batches) { if (bt->stream) hipStreamSynchronize(bt->stream); bt->ctx = nullptr; }
  for (tsd_sensor* sn : ctx->sensors) { if (sn->stream) hipStreamSynchronize(sn->stream); sn->ctx = nullptr; }
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  GridDev& g = ctx->grid;
  hipFree(g.flags); hipFree(g.init_weight); hipFree(g.tsd); hipFree(g.weight); hipFree(g.negmask);
  if (ctx->stream2) hipStreamSynchronize(ctx->stream2);
  hipFree(ctx->d_rmq2[0]); hipFree(ctx->d_rmq2[1]); hipFree(ctx->d_tile_rec); hipFree(ctx->d_dirty); hipFree(ctx->d_tile_totals); hipFree(ctx->d_pushes); hipFree(ctx->d_list); hipFree(ctx->d_list_h); hipFree(ctx->d_list_aux); hipFree(ctx->d_push_args); hipFree(ctx->d_list_cnt);
  if (ctx->ev_tables) hipEventDestroy(ctx->ev_tables);
  if (ctx->ev_h2d) hipEventDestroy(ctx->ev_h2d);
  if (ctx->ev_grid) hipEventDestroy(ctx->ev_grid);
  if (ctx->stream2) hipStreamDestroy(ctx->stream2);
  if (ctx->stream_push) { hipStreamSynchronize(ctx->stream_push); hipStreamDestroy(ctx->stream_push); }
  if (ctx->ev_async_rc) hipEventDestroy(ctx->ev_async_rc);
  // (ev_async_push is one of a sensor's ev_slot_push[]: the sensor owns it)
  for (hipEvent_t e : ctx->event_pool) hipEventDestroy(e);
  for (int s = 0; s < tsd_ctx::kSlots; s++) {
    if (ctx->h_stage[s]) hipHostFree(ctx->h_stage[s]);
    if (ctx->stage_ev[s]) hipEventDestroy(ctx->stage_ev[s]);
  }
  hipFree(ctx->d_ranges); hipFree(ctx->d_mask); hipFree(ctx->d_rays); hipFree(ctx->d_rays_local);
  hipFree(ctx->d_coords); /* (+ d_normals, d_mask_m: one block) */ hipFree(ctx->d_mnormals); hipFree(ctx->d_model);
  hipFree(ctx->d_scene); hipFree(ctx->d_morig); hipFree(ctx->d_start); hipFree(ctx->d_icp_res); hipFree(ctx->d_icp_seed); hipFree(ctx->d_icp_trace); hipHostFree(ctx->h_icp_res); hipHostFree(ctx->h_out);
  hipFree(ctx->d_occ); hipFree(ctx->d_occ_count); hipFree(ctx->d_occ_heads); hipFree(ctx->d_occ_list);
  if (ctx->d_occ_out) hipFree(ctx->d_occ_out);
  if (ctx->stream_io) { hipStreamSynchronize(ctx->stream_io); hipStreamDestroy(ctx->stream_io); }
  if (ctx->ev_io) hipEventDestroy(ctx->ev_io);
  if (ctx->d_img) hipFree(ctx->d_img);
  if (ctx->d_pdf) hipFree(ctx->d_pdf);
  if (ctx->h_pdf) hipHostFree(ctx->h_pdf);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

int tsd_reset(tsd_ctx* ctx)
{
  if (ctx) ctx->epoch++;                  // (invalidates ray casts enqueued ahead of their scan)
  if (!ctx) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  std::lock_guard<std::mutex> lk_order(ctx->order_mutex);
  if (int rcw = wait_for_readers(ctx)) return rcw;
  GridDev& g = ctx->grid;
  const size_t T = (size_t)g.tiles;
  // cells are materialised lazily (flags == 0 means "no cell storage yet", TsdGridPartition.cpp:88)
  TSD_HIP_CHECK(ctx, hipMemsetAsync(g.flags, 0, T, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemsetAsync(g.init_weight, 0, T * sizeof(double), ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_dirty, 0, T, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemsetAsync(g.negmask, 0, T * sizeof(unsigned long long), ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_tile_rec, 0, T * sizeof(uint32_t), ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_tile_totals, 0, T * 8 * sizeof(uint32_t), ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_pushes, 0, 2 * sizeof(unsigned long long), ctx->stream));
  ctx->box_prev = TileBox{}; ctx->box_dirty = TileBox{};
  TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_list_cnt, 0, push_list_cnt_bytes(), ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_occ, 0xFF, (size_t)g.N * g.N, ctx->stream));   // -1 (ThreadGrid.cpp:27-28)
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

int tsd_set_max_truncation(tsd_ctx* ctx, double val)
{
  if (ctx) ctx->epoch++;                  // (invalidates ray casts enqueued ahead of their scan)
  if (!ctx) return TSD_E_ARG;
  // TsdGrid::setMaxTruncation (TsdGrid.cpp:206-215): at least 2 x cell size
  if (val < 2 * ctx->grid.cs) val = 2 * ctx->grid.cs;
  ctx->grid.max_trunc = val;
  return TSD_OK;
}

int tsd_sync(tsd_ctx* ctx)
{
  if (!ctx) return TSD_E_ARG;
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

const char* tsd_last_error(const tsd_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
int   tsd_device(const tsd_ctx* ctx) { return ctx ? ctx->device : -1; }
void* tsd_stream(tsd_ctx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

int    tsd_cells(const tsd_ctx* ctx) { return ctx ? ctx->grid.N : 0; }
int    tsd_tiles(const tsd_ctx* ctx) { return ctx ? ctx->grid.tiles : 0; }
double tsd_cell_size(const tsd_ctx* ctx) { return ctx ? ctx->grid.cs : 0.0; }
double tsd_max_truncation(const tsd_ctx* ctx) { return ctx ? ctx->grid.max_trunc : 0.0; }
double tsd_min_x(const tsd_ctx* ctx) { return ctx ? ctx->grid.min_x : 0.0; }
double tsd_max_x(const tsd_ctx* ctx) { return ctx ? ctx->grid.max_x : 0.0; }
double tsd_min_y(const tsd_ctx* ctx) { return ctx ? ctx->grid.min_y : 0.0; }
double tsd_max_y(const tsd_ctx* ctx) { return ctx ? ctx->grid.max_y : 0.0; }

int tsd_free_footprint(tsd_ctx* ctx, const double center[2], double width, double height)
{
  if (ctx) ctx->epoch++;                  // (invalidates ray casts enqueued ahead of their scan)
  if (!ctx || !center) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const GridDev& g = ctx->grid;
  // TsdGrid.cpp:611-622
  const unsigned minX = static_cast<unsigned>((center[0] - width * 0.5) / g.cs + 0.5);
  const unsigned maxX = static_cast<unsigned>((center[0] + width * 0.5) / g.cs + 0.5);
  const unsigned minY = static_cast<unsigned>((center[1] - height * 0.5) / g.cs + 0.5);
  const unsigned maxY = static_cast<unsigned>((center[1] + height * 0.5) / g.cs + 0.5);
  const unsigned N = (unsigned)g.N;
  if (minX > N || maxX > N || minY > N || maxY > N)
    return set_error(ctx, TSD_E_BOUNDS, "freeFootprint: indices out of bounds", hipSuccess);
  std::lock_guard<std::mutex> lk_order(ctx->order_mutex);
  if (int rcw = wait_for_readers(ctx)) return rcw;
  // cells == N would index past the last tile in the reference (undefined there); clamp
  const unsigned cx1 = maxX > N ? N : maxX, cy1 = maxY > N ? N : maxY;
  return launch_free_footprint(ctx, minX, cx1, minY, cy1);
}

int tsd_push(tsd_ctx* ctx, const double pose33[9], const double* ranges, const uint8_t* mask,
             int beams, double ang_res, double phi_min, double max_range, double min_range,
             double low_refl_range, tsd_push_stats* stats)
{
  if (ctx) ctx->epoch++;                  // (invalidates ray casts enqueued ahead of their scan)
  if (!ctx || !pose33 || !ranges || !mask) return TSD_E_ARG;
  if (beams < 1 || beams > TSD_MAX_BEAMS) return set_error(ctx, TSD_E_CAPACITY, "beams out of range", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  PushArgs a;
  double Pi[9];
  mat3_inv(pose33, Pi);
  for (int i = 0; i < 6; i++) a.Pi[i] = Pi[i];
  a.trx = pose33[2]; a.try_ = pose33[5];                     // Sensor::getPosition (Sensor.cpp:114-118)
  a.phi_min = phi_min; a.ang_res_inv = 1.0 / ang_res;
  a.phi_lower = -0.5 * ang_res + phi_min;                    // SensorPolar2D.cpp:26-30
  a.phi_upper = phi_min + (((double)beams) - 0.5) * ang_res;
  a.max_range = max_range; a.min_range = min_range; a.low_refl = low_refl_range;
  a.beams = beams; a.enabled = 1;

  int s;
  char* h = stage_acquire(ctx, &s);
  std::memcpy(h, ranges, (size_t)beams * sizeof(double));
  std::memcpy(h + (size_t)TSD_MAX_BEAMS * 8, mask, (size_t)beams);
  std::memcpy(h + (size_t)TSD_MAX_BEAMS * 9, &a, sizeof(a));        // the kernels read their arguments from device memory
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_ranges, h, (size_t)beams * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_mask, h + (size_t)TSD_MAX_BEAMS * 8, (size_t)beams, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_push_args, h + (size_t)TSD_MAX_BEAMS * 9, sizeof(a), hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[s], ctx->stream));

  std::lock_guard<std::mutex> lk_order(ctx->order_mutex);
  int rc = wait_for_readers(ctx);
  if (rc != TSD_OK) return rc;
  rc = launch_push_tables(ctx, ctx->stream, beams, nullptr, nullptr, phi_min, ang_res);
  if (rc != TSD_OK) return rc;
  rc = launch_push(ctx, a, a.trx, a.try_, 0.0, ctx->d_push_args);
  if (rc != TSD_OK) return rc;
  if (stats) {
    rc = read_last_push_stats(ctx, stats);
    if (rc != TSD_OK) return rc;
  }
  return TSD_OK;
}

int tsd_raycast(tsd_ctx* ctx, const double pose33[9], const double* rays_world_2xB, int beams,
                double min_range, double max_range, double* coords_2B, double* normals_2B,
                uint8_t* mask_B, int* n_valid)
{
  if (ctx) ctx->epoch++;                  // (invalidates ray casts enqueued ahead of their scan)
  if (!ctx || !pose33 || !rays_world_2xB || !coords_2B || !normals_2B || !mask_B) return TSD_E_ARG;
  if (beams < 1 || beams > TSD_MAX_BEAMS) return set_error(ctx, TSD_E_CAPACITY, "beams out of range", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  RaycastArgs a;
  fill_raycast_args(ctx, a, pose33, beams, min_range, max_range);
  int s;
  char* h = stage_acquire(ctx, &s);
  std::memcpy(h, rays_world_2xB, (size_t)beams * 2 * sizeof(double));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_rays, h, (size_t)beams * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[s], ctx->stream));
  int rc = launch_raycast(ctx, a);
  if (rc != TSD_OK) return rc;
  const size_t nb = (size_t)beams;
  char* o = ctx->h_out;
  // coords | normals | mask: one block on both sides (tsd_create), one copy (three copies cost ~3 x 15 us of a 12 us kernel's call)
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(o, ctx->d_coords, (size_t)TSD_MAX_BEAMS * 32 + nb, hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const double* hc = reinterpret_cast<const double*>(o);
  const double* hn = reinterpret_cast<const double*>(o + (size_t)TSD_MAX_BEAMS * 16);
  const uint8_t* hm = reinterpret_cast<const uint8_t*>(o + (size_t)TSD_MAX_BEAMS * 32);
  int cnt = 0;
  for (int b = 0; b < beams; b++) {
    mask_B[b] = hm[b];
    if (hm[b]) {   // only hit slots are written (RayCastPolar2D.cpp:171-178)
      coords_2B[2 * b] = hc[2 * b]; coords_2B[2 * b + 1] = hc[2 * b + 1];
      normals_2B[2 * b] = hn[2 * b]; normals_2B[2 * b + 1] = hn[2 * b + 1];
      cnt++;
    }
  }
  if (n_valid) *n_valid = cnt;
  return TSD_OK;
}

int tsd_icp(tsd_ctx* ctx, const double* model_xy, int n_model, const double* scene_xy, int n_scene,
            const double pose33[9], const tsd_icp_params* params, tsd_icp_result* result)
{
  return tsd_icp_normals(ctx, model_xy, nullptr, n_model, scene_xy, n_scene, pose33, params, result);
}

// Direct-mode inputs of a registration on their way to the device.  The kernel's exact nearest-neighbour walk wants the model in
// angular order about the origin of the sensor frame (what the ray cast emits by construction); arbitrary callers get it sorted
// here; the original indices travel along for the lowest-index tie rule (`order`: slot -> original model index), and every scene
// point gets the slot where its own direction falls as first search position.  Ordering only: no arithmetic on the data.
static int stage_icp_inputs(tsd_ctx* ctx, const IcpArgs& a, const double* model_xy, const double* model_normals_xy, int n_model,
                            const double* scene_xy, int n_scene, bool normals, std::vector<int>& order)
{
  order.resize((size_t)n_model);
  std::vector<int> start((size_t)n_scene);
  std::vector<double> ang((size_t)n_model);
  for (int j = 0; j < n_model; j++) ang[(size_t)j] = std::atan2(model_xy[2 * j + 1], model_xy[2 * j]);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return ang[(size_t)x] < ang[(size_t)y]; });
  std::vector<double> sorted_ang((size_t)n_model);
  for (int k = 0; k < n_model; k++) sorted_ang[(size_t)k] = ang[(size_t)order[(size_t)k]];
  for (int i = 0; i < n_scene; i++) {
    // (search hint only: where the point's direction falls after Tinit)
    const double hx = a.Tinit[0] * scene_xy[2 * i] + a.Tinit[1] * scene_xy[2 * i + 1] + a.Tinit[2];
    const double hy = a.Tinit[3] * scene_xy[2 * i] + a.Tinit[4] * scene_xy[2 * i + 1] + a.Tinit[5];
    const double t = std::atan2(hy, hx);
    int k = (int)(std::lower_bound(sorted_ang.begin(), sorted_ang.end(), t) - sorted_ang.begin());
    start[(size_t)i] = (n_model > 0 && k >= n_model) ? 0 : k;
  }
  // model, scene, permutation and start slots share one staging slot (2 * 2048 * (16 + 4) B = 80 KB)
  int s;
  char* h = stage_acquire(ctx, &s);
  const size_t mb = (size_t)n_model * 16, sb = (size_t)n_scene * 16, ob = (size_t)n_model * 4, tb = (size_t)n_scene * 4;
  const size_t nb = normals ? mb : 0;
  if (mb + sb + ob + tb + nb > ctx->stage_bytes) return set_error(ctx, TSD_E_CAPACITY, "icp staging", hipSuccess);
  double* hm = reinterpret_cast<double*>(h);
  for (int k = 0; k < n_model; k++) { const int j = order[(size_t)k]; hm[2 * k] = model_xy[2 * j]; hm[2 * k + 1] = model_xy[2 * j + 1]; }
  if (sb) std::memcpy(h + mb, scene_xy, sb);
  if (ob) std::memcpy(h + mb + sb, order.data(), ob);
  if (tb) std::memcpy(h + mb + sb + ob, start.data(), tb);
  if (mb) TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_model, h, mb, hipMemcpyHostToDevice, ctx->stream));
  if (sb) TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_scene, h + mb, sb, hipMemcpyHostToDevice, ctx->stream));
  if (ob) TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_morig, h + mb + sb, ob, hipMemcpyHostToDevice, ctx->stream));
  if (tb) TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_start, h + mb + sb + ob, tb, hipMemcpyHostToDevice, ctx->stream));
  if (nb) {
    double* hn = reinterpret_cast<double*>(h + mb + sb + ob + ((tb + 7) & ~(size_t)7));
    if (mb + sb + ob + ((tb + 7) & ~(size_t)7) + nb > ctx->stage_bytes) return set_error(ctx, TSD_E_CAPACITY, "icp staging", hipSuccess);
    for (int k = 0; k < n_model; k++) { const int j = order[(size_t)k]; hn[2 * k] = model_normals_xy[2 * j]; hn[2 * k + 1] = model_normals_xy[2 * j + 1]; }
    TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_mnormals, hn, nb, hipMemcpyHostToDevice, ctx->stream));
  }
  TSD_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[s], ctx->stream));
  return TSD_OK;
}

int tsd_icp_normals(tsd_ctx* ctx, const double* model_xy, const double* model_normals_xy, int n_model,
                    const double* scene_xy, int n_scene, const double pose33[9], const tsd_icp_params* params,
                    tsd_icp_result* result)
{
  if (ctx) ctx->epoch++;                  // (invalidates ray casts enqueued ahead of their scan)
  if (!ctx || !pose33 || !params || !result || n_model < 0 || n_scene < 0) return TSD_E_ARG;
  if (params->estimator != TSD_ESTIMATOR_CLOSED_FORM && params->estimator != TSD_ESTIMATOR_POINT_TO_LINE)
    return set_error(ctx, TSD_E_ARG, "tsd_icp_params.estimator", hipSuccess);
  if (params->estimator == TSD_ESTIMATOR_POINT_TO_LINE && n_model > 0 && !model_normals_xy)
    return set_error(ctx, TSD_E_ARG, "the point-to-line estimator needs the model normals (tsd_icp_normals)", hipSuccess);
  if ((n_model > 0 && !model_xy) || (n_scene > 0 && !scene_xy)) return TSD_E_ARG;
  if (n_model > TSD_MAX_ICP_POINTS || n_scene > TSD_MAX_ICP_POINTS)
    return set_error(ctx, TSD_E_CAPACITY, "icp points > TSD_MAX_ICP_POINTS", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  IcpArgs a;
  fill_icp_args(a, pose33, params);
  a.n_model = n_model; a.n_scene = n_scene; a.beams = 0;
  std::vector<int> order;
  int rc = stage_icp_inputs(ctx, a, model_xy, model_normals_xy, n_model, scene_xy, n_scene, params->estimator == TSD_ESTIMATOR_POINT_TO_LINE, order);
  if (rc != TSD_OK) return rc;
  rc = launch_icp(ctx, a);
  if (rc != TSD_OK) return rc;
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_icp_res, ctx->d_icp_res, sizeof(IcpResultDev), hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  copy_icp_result(ctx->h_icp_res, result);
  return TSD_OK;
}

int tsd_icp_pairs(tsd_ctx* ctx, const double* model_xy, int n_model, const double* scene_xy, int n_scene, const double pose33[9],
                  const tsd_icp_params* params, int calls, int* n_pairs, int* model_idx, int* scene_idx)
{
  if (ctx) ctx->epoch++;
  if (!ctx || !pose33 || !params || !n_pairs || !model_idx || !scene_idx || n_model < 1 || n_scene < 1 || !model_xy || !scene_xy) return TSD_E_ARG;
  if (calls < 1 || calls > TSD_ICP_TRACE_MAX) return set_error(ctx, TSD_E_ARG, "tsd_icp_pairs: calls out of range", hipSuccess);
  if (params->estimator != TSD_ESTIMATOR_CLOSED_FORM) return set_error(ctx, TSD_E_ARG, "tsd_icp_pairs: closed-form instantiation only", hipSuccess);
  if (n_model > TSD_MAX_ICP_POINTS || n_scene > TSD_MAX_ICP_POINTS)
    return set_error(ctx, TSD_E_CAPACITY, "icp points > TSD_MAX_ICP_POINTS", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  IcpArgs a;
  fill_icp_args(a, pose33, params);        // (the threshold schedule comes from params->iterations, like the node's DistanceFilter)
  a.n_model = n_model; a.n_scene = n_scene; a.beams = 0;
  a.iterations = calls;                    // ... the number of determinePairs calls from `calls`
  std::vector<int> order;
  int rc = stage_icp_inputs(ctx, a, model_xy, nullptr, n_model, scene_xy, n_scene, false, order);
  if (rc != TSD_OK) return rc;
  const int cap = icp_pairs_cap(n_model, n_scene);
  const size_t words = (size_t)calls * (size_t)cap;
  int* d_pairs = nullptr;
  TSD_HIP_CHECK(ctx, hipMalloc(&d_pairs, words * sizeof(int)));
  std::vector<int> h_pairs(words);
  hipError_t e = hipMemsetAsync(d_pairs, 0xFF, words * sizeof(int), ctx->stream);
  if (e == hipSuccess) rc = launch_icp_pairs(ctx, a, d_pairs);
  if (e == hipSuccess && rc == TSD_OK) e = hipMemcpyAsync(h_pairs.data(), d_pairs, words * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && rc == TSD_OK) e = hipStreamSynchronize(ctx->stream);
  hipFree(d_pairs);
  if (e != hipSuccess) return set_error(ctx, TSD_E_HIP, "tsd_icp_pairs", e);
  if (rc != TSD_OK) return rc;
  // ReciprocalFilter leaves at most one pair per model point and emits them in ascending MODEL index (ReciprocalFilter.cpp:16-21,
  // :32-78): slot -> original model index, then that order
  std::vector<std::pair<int, int>> pr;
  for (int k = 0; k < calls; k++) {
    pr.clear();
    for (int slot = 0; slot < n_model; slot++) {
      const int si = h_pairs[(size_t)k * (size_t)cap + (size_t)slot];
      if (si >= 0) pr.emplace_back(order[(size_t)slot], si);
    }
    std::sort(pr.begin(), pr.end());
    n_pairs[k] = (int)pr.size();
    for (size_t i = 0; i < pr.size() && i < (size_t)n_scene; i++) {
      model_idx[(size_t)k * (size_t)n_scene + i] = pr[i].first;
      scene_idx[(size_t)k * (size_t)n_scene + i] = pr[i].second;
    }
  }
  return TSD_OK;
}

int tsd_localize(tsd_ctx* ctx, const double pose33[9], const double* rays_world_2xB,
                 const double* rays_local_2xB, const double* ranges, const uint8_t* mask, int beams,
                 double min_range, double max_range, const tsd_icp_params* params,
                 tsd_icp_result* result)
{
  if (ctx) ctx->epoch++;                  // (invalidates ray casts enqueued ahead of their scan)
  if (!ctx || !pose33 || !rays_world_2xB || !rays_local_2xB || !ranges || !mask || !params || !result) return TSD_E_ARG;
  if (beams < 1 || beams > TSD_MAX_BEAMS || beams > TSD_MAX_ICP_POINTS)
    return set_error(ctx, TSD_E_CAPACITY, "beams out of range for fused localize", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const size_t nb = (size_t)beams;
  int s;
  char* h = stage_acquire(ctx, &s);
  char* h_rw = h;                         // 2B doubles
  char* h_rl = h + nb * 16;               // 2B doubles
  char* h_r = h + nb * 32;                // B doubles
  char* h_m = h + nb * 40;                // B bytes
  std::memcpy(h_rw, rays_world_2xB, nb * 16);
  std::memcpy(h_rl, rays_local_2xB, nb * 16);
  std::memcpy(h_r, ranges, nb * 8);
  std::memcpy(h_m, mask, nb);
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_rays, h_rw, nb * 16, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_rays_local, h_rl, nb * 16, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_ranges, h_r, nb * 8, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_mask, h_m, nb, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[s], ctx->stream));

  RaycastArgs ra;
  fill_raycast_args(ctx, ra, pose33, beams, min_range, max_range);
  int rc = launch_raycast(ctx, ra);
  if (rc != TSD_OK) return rc;
  IcpArgs ia;
  fill_icp_args(ia, pose33, params);
  ia.beams = beams;
  // beam order is counter-clockwise when consecutive local rays turn left (positive angle increment)
  ia.ccw = (beams < 2) || (rays_local_2xB[0] * rays_local_2xB[nb + 1] - rays_local_2xB[nb] * rays_local_2xB[1] >= 0.0);
  rc = launch_icp(ctx, ia);
  if (rc != TSD_OK) return rc;
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_icp_res, ctx->d_icp_res, sizeof(IcpResultDev), hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  copy_icp_result(ctx->h_icp_res, result);
  return TSD_OK;
}

int tsd_icp_trace(tsd_ctx* ctx, double* out, int max_iters)
{
  if (!ctx || !out || max_iters < 0) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const int n = max_iters < 2 * TSD_ICP_TRACE_MAX ? max_iters : 2 * TSD_ICP_TRACE_MAX;     // (rows beyond TSD_ICP_TRACE_MAX: diagnostic builds' scratch)
  // (every copy of this library goes through the context's own stream: a plain hipMemcpy / hipMemset brings the NULL stream
  // alive, and that stream takes one of the few hardware queues the scan / batch streams are mapped onto -- DESIGN 5)
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(out, ctx->d_icp_trace, sizeof(double) * TSD_ICP_TRACE_STRIDE * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

int tsd_download_tile_state(tsd_ctx* ctx, uint8_t* initialized, double* init_weight)
{
  if (!ctx || !initialized || !init_weight) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const size_t T = (size_t)ctx->grid.tiles;
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(initialized, ctx->grid.flags, T, hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(init_weight, ctx->grid.init_weight, T * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

// tiles per chunk of the canonical tile I/O: 2 x 4096 x 1089 doubles = 71 MB of device staging
static constexpr int kIoChunk = 4096;

int tsd_download_tiles(tsd_ctx* ctx, uint8_t* initialized, double* init_weight, double* tsd_out,
                       double* weight_out)
{
  if (!ctx || !initialized || !init_weight || !tsd_out || !weight_out) return TSD_E_ARG;
  int rc = tsd_download_tile_state(ctx, initialized, init_weight);
  if (rc != TSD_OK) return rc;
  const GridDev& g = ctx->grid;
  const double qnan = std::nan("");
  const int chunk = g.tiles < kIoChunk ? g.tiles : kIoChunk;
  double* d_t = nullptr; double* d_w = nullptr;
  const size_t cb = (size_t)chunk * TSD_TILE_CELLS * sizeof(double);
  TSD_HIP_CHECK(ctx, hipMalloc(&d_t, cb));
  hipError_t e = hipMalloc(&d_w, cb);
  if (e != hipSuccess) { hipFree(d_t); return set_error(ctx, TSD_E_HIP, "tsd_download_tiles staging", e); }
  for (int t0 = 0; t0 < g.tiles && rc == TSD_OK; t0 += chunk) {
    const int n = g.tiles - t0 < chunk ? g.tiles - t0 : chunk;
    bool any = false;
    for (int p = t0; p < t0 + n; p++) any |= initialized[p] != 0;
    double* t = tsd_out + (size_t)t0 * TSD_TILE_CELLS;
    double* w = weight_out + (size_t)t0 * TSD_TILE_CELLS;
    if (!any) {                                    // most of a big grid: nothing to fetch
      for (size_t i = 0; i < (size_t)n * TSD_TILE_CELLS; i++) { t[i] = qnan; w[i] = 0.0; }
      continue;
    }
    rc = launch_export_tiles(ctx, t0, n, d_t, d_w);
    if (rc != TSD_OK) break;
    e = hipMemcpyAsync(t, d_t, (size_t)n * TSD_TILE_CELLS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(w, d_w, (size_t)n * TSD_TILE_CELLS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_download_tiles copy", e);
  }
  hipFree(d_t); hipFree(d_w);
  return rc;
}

int tsd_upload_tiles(tsd_ctx* ctx, const uint8_t* initialized, const double* init_weight,
                     const double* tsd_in, const double* weight_in)
{
  if (ctx) ctx->epoch++;                  // (invalidates ray casts enqueued ahead of their scan)
  if (!ctx || !initialized || !init_weight || !tsd_in || !weight_in) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  std::lock_guard<std::mutex> lk_order(ctx->order_mutex);
  if (int rcw = wait_for_readers(ctx)) return rcw;
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const GridDev& g = ctx->grid;
  const size_t T = (size_t)g.tiles;
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(g.flags, initialized, T, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(g.init_weight, init_weight, T * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const int chunk = g.tiles < kIoChunk ? g.tiles : kIoChunk;
  double* d_t = nullptr; double* d_w = nullptr;
  const size_t cb = (size_t)chunk * TSD_TILE_CELLS * sizeof(double);
  TSD_HIP_CHECK(ctx, hipMalloc(&d_t, cb));
  hipError_t e = hipMalloc(&d_w, cb);
  if (e != hipSuccess) { hipFree(d_t); return set_error(ctx, TSD_E_HIP, "tsd_upload_tiles staging", e); }
  int rc = TSD_OK;
  for (int t0 = 0; t0 < g.tiles && rc == TSD_OK; t0 += chunk) {
    const int n = g.tiles - t0 < chunk ? g.tiles - t0 : chunk;
    bool any = false;
    for (int p = t0; p < t0 + n; p++) any |= initialized[p] != 0;
    if (!any) continue;
    e = hipMemcpyAsync(d_t, tsd_in + (size_t)t0 * TSD_TILE_CELLS, (size_t)n * TSD_TILE_CELLS * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_w, weight_in + (size_t)t0 * TSD_TILE_CELLS, (size_t)n * TSD_TILE_CELLS * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { rc = set_error(ctx, TSD_E_HIP, "tsd_upload_tiles copy", e); break; }
    rc = launch_import_tiles(ctx, t0, n, d_t, d_w);
    if (rc == TSD_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_upload_tiles sync", hipGetLastError());
  }
  hipFree(d_t); hipFree(d_w);
  if (rc != TSD_OK) return rc;
  rc = launch_neg_scan(ctx);              // which tiles can show a sign change to the ray cast
  if (rc != TSD_OK) return rc;
  // The halos came as they were given (the text format does not store them: NaN): nothing says they agree with the neighbours' edge
  // cells, which the incremental propagateBorders of the push relies on for the tiles it does not touch.  Every tile that holds data is
  // marked like a freeFootprint write: the next push refreshes the halos around all of them -- the reference's full sweep after its
  // first push (TsdGrid.cpp:372-427).
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_dirty, initialized, T, hipMemcpyHostToDevice, ctx->stream));
  {
    TileBox all; all.x0 = 0; all.y0 = 0; all.x1 = g.PX - 1; all.y1 = g.PX - 1;
    ctx->box_dirty.add(all);
  }
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

int tsd_grid_digest(tsd_ctx* ctx, tsd_grid_digest_t* out)
{
  if (!ctx || !out) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const size_t T = (size_t)ctx->grid.tiles;
  unsigned long long* d_o = nullptr; double* d_s = nullptr;
  TSD_HIP_CHECK(ctx, hipMalloc(&d_o, T * 2 * sizeof(unsigned long long)));
  hipError_t e = hipMalloc(&d_s, T * 2 * sizeof(double));
  if (e != hipSuccess) { hipFree(d_o); return set_error(ctx, TSD_E_HIP, "tsd_grid_digest", e); }
  int rc = launch_grid_digest(ctx, d_o, d_s);
  std::vector<unsigned long long> ho(T * 2); std::vector<double> hs(T * 2); std::vector<uint8_t> fl(T);
  if (rc == TSD_OK) {
    e = hipMemcpyAsync(ho.data(), d_o, T * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hs.data(), d_s, T * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(fl.data(), ctx->grid.flags, T, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_grid_digest copy", e);
  }
  hipFree(d_o); hipFree(d_s);
  if (rc != TSD_OK) return rc;
  out->hash = 0; out->cells_valid = 0; out->tiles_initialized = 0; out->sum_tsd = 0.0; out->sum_weight = 0.0;
  for (size_t p = 0; p < T; p++) {          // tile order: the sums are reproducible
    out->hash += ho[2 * p]; out->cells_valid += (int64_t)ho[2 * p + 1];
    out->sum_tsd += hs[2 * p]; out->sum_weight += hs[2 * p + 1];
    out->tiles_initialized += fl[p] ? 1 : 0;
  }
  return TSD_OK;
}

int tsd_storage_bits(void) { return (int)(8 * sizeof(tsd_cell_t)); }

int tsd_abi_sizeof(const char* n)
{
  if (!n) return 0;
#define TSD_SZ(T) if (std::strcmp(n, #T) == 0) return (int)sizeof(T)
  TSD_SZ(tsd_push_stats); TSD_SZ(tsd_icp_params); TSD_SZ(tsd_icp_result); TSD_SZ(tsd_gate_params); TSD_SZ(tsd_scan_result);
  TSD_SZ(tsd_grid_digest_t); TSD_SZ(tsd_tsdpdf_params); TSD_SZ(tsd_tsdpdf_result);
#undef TSD_SZ
  return 0;
}

int tsd_store_grid_text(tsd_ctx* ctx, const char* path)
{
  if (!ctx || !path || !path[0]) return TSD_E_ARG;
  const GridDev& g = ctx->grid;
  const size_t T = (size_t)g.tiles;
  std::vector<uint8_t> init(T);
  std::vector<double> iw(T), tsd(T * TSD_TILE_CELLS), w(T * TSD_TILE_CELLS);
  int rc = tsd_download_tiles(ctx, init.data(), iw.data(), tsd.data(), w.data());
  if (rc != TSD_OK) return rc;
  std::FILE* f = std::fopen(path, "w");
  if (!f) return set_error(ctx, TSD_E_ARG, "tsd_store_grid_text: cannot open the file", hipSuccess);
  int map_log2 = 0;
  while ((1 << map_log2) < g.N) map_log2++;
  // "%g" is the default ostream format of the reference's `outFile << value`
  std::fprintf(f, "%g\n%d\n%d\n%g\n", g.cs, 5 /* LAYOUT_32x32 */, map_log2, g.max_trunc);
  for (size_t p = 0; p < T; p++) {
    if (init[p]) {
      std::fprintf(f, "2\n");
      for (int py = 0; py < TILE_DIM; py++)
        for (int px = 0; px < TILE_DIM; px++) {
          const size_t i = p * TSD_TILE_CELLS + (size_t)(py * TILE_PITCH + px);
          std::fprintf(f, "%g\n%g\n", tsd[i], w[i]);
        }
    } else if (iw[p] > 0.0) {            // isEmpty()
      std::fprintf(f, "1\n%g\n", iw[p]);
    } else {
      std::fprintf(f, "0\n");
    }
  }
  std::fclose(f);
  return TSD_OK;
}

// getDoubleLine / getIntLine (obcore/base/tools.cpp:190-215)
static double text_double_line(std::FILE* f)
{
  char line[1024];
  if (!std::fgets(line, sizeof(line), f) || line[0] == '\n' || line[0] == 0) return std::nan("");
  return std::strtod(line, nullptr);
}
static int text_int_line(std::FILE* f)
{
  char line[1024];
  if (!std::fgets(line, sizeof(line), f) || line[0] == '\n' || line[0] == 0) return 0;
  return std::atoi(line);
}

int tsd_load_grid_text(tsd_ctx* ctx, const char* path)
{
  if (!ctx || !path || !path[0]) return TSD_E_ARG;
  std::FILE* f = std::fopen(path, "r");
  if (!f) return set_error(ctx, TSD_E_ARG, "tsd_load_grid_text: cannot open the file", hipSuccess);
  const GridDev& g = ctx->grid;
  const double cell_size = text_double_line(f);
  const int layout_part = text_int_line(f), layout_grid = text_int_line(f);
  const double max_trunc = text_double_line(f);
  int map_log2 = 0;
  while ((1 << map_log2) < g.N) map_log2++;
  if (layout_part != 5 || layout_grid != map_log2 || !(std::fabs(cell_size - g.cs) <= 1e-5 * g.cs)) {
    std::fclose(f);
    return set_error(ctx, TSD_E_ARG, "tsd_load_grid_text: the file's layout / cell size is not this grid's", hipSuccess);
  }
  const size_t T = (size_t)g.tiles;
  std::vector<uint8_t> init(T, 0);
  std::vector<double> iw(T, 0.0), tsd(T * TSD_TILE_CELLS), w(T * TSD_TILE_CELLS, 0.0);
  for (size_t p = 0; p < T; p++) {
    const int id = text_int_line(f);
    if (id == 0) continue;
    if (id == 1) { iw[p] = std::fmin(text_double_line(f), 32.0 /* TSDGRIDMAXWEIGHT */); continue; }
    if (id != 2) { std::fclose(f); return set_error(ctx, TSD_E_ARG, "tsd_load_grid_text: unknown tile identifier", hipSuccess); }
    // curPart->init(maxTruncation) on a fresh partition (_initWeight 0): every cell NaN / 0, halo included; then the
    // interior cells from the file
    init[p] = 1;
    for (int i = 0; i < TSD_TILE_CELLS; i++) tsd[p * TSD_TILE_CELLS + (size_t)i] = std::nan("");
    for (int py = 0; py < TILE_DIM; py++)
      for (int px = 0; px < TILE_DIM; px++) {
        const size_t i = p * TSD_TILE_CELLS + (size_t)(py * TILE_PITCH + px);
        tsd[i] = text_double_line(f);
        w[i] = text_double_line(f);
      }
  }
  std::fclose(f);
  int rc = tsd_reset(ctx);
  if (rc != TSD_OK) return rc;
  rc = tsd_set_max_truncation(ctx, max_trunc);
  if (rc != TSD_OK) return rc;
  return tsd_upload_tiles(ctx, init.data(), iw.data(), tsd.data(), w.data());
}

int tsd_occupancy_dev_async(tsd_ctx* ctx, void* occ_dev, int inflate, int inflate_factor)
{
  if (!ctx || !occ_dev) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  return launch_occupancy(ctx, static_cast<int8_t*>(occ_dev), inflate, inflate_factor);
}

int tsd_occupancy_dev(tsd_ctx* ctx, void* occ_dev, int inflate, int inflate_factor)
{
  int rc = tsd_occupancy_dev_async(ctx, occ_dev, inflate, inflate_factor);
  if (rc != TSD_OK) return rc;
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

int tsd_color_image(tsd_ctx* ctx, uint8_t* rgb_host, unsigned int width, unsigned int height)
{
  if (!ctx || !rgb_host || width == 0 || height == 0) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  // px / py exactly as the reference accumulates them (TsdGrid.cpp:433-486): start at 0, += step per pixel
  std::vector<double> pq((size_t)width + height);
  const double stepW = ctx->grid.max_x / (double)width, stepH = ctx->grid.max_y / (double)height;
  { double v = 0.0; for (unsigned w = 0; w < width; w++) { pq[w] = v; v += stepW; } }
  { double v = 0.0; for (unsigned h = 0; h < height; h++) { pq[(size_t)width + h] = v; v += stepH; } }
  // (device staging kept by the context and grown on demand: ThreadGrid publishes map and image every occ_grid_time_interval
  // beside the localisers, ThreadGrid.cpp:72-133 -- a hipMalloc / hipFree pair per call would stall every stream of the device)
  const size_t img_bytes = (size_t)3 * width * height, pq_bytes = pq.size() * sizeof(double);
  if (ctx->img_bytes < img_bytes + pq_bytes + 64) {
    TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_img) hipFree(ctx->d_img);
    ctx->d_img = nullptr; ctx->img_bytes = 0;
    TSD_HIP_CHECK(ctx, hipMalloc(&ctx->d_img, img_bytes + pq_bytes + 64));
    ctx->img_bytes = img_bytes + pq_bytes + 64;
  }
  double* d_pq = reinterpret_cast<double*>(ctx->d_img);
  uint8_t* d_img = ctx->d_img + ((pq_bytes + 63) & ~(size_t)63);
  int rc = TSD_OK;
  hipError_t e = hipMemcpyAsync(d_pq, pq.data(), pq_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) rc = launch_color_image(ctx, d_pq, d_pq + width, width, height, d_img);
  if (e == hipSuccess && rc == TSD_OK) e = hipMemcpyAsync(rgb_host, d_img, img_bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);       // (also when the launch failed: `pq` must outlive its copy)
  if (e != hipSuccess) return set_error(ctx, TSD_E_HIP, "tsd_color_image", e);
  return rc;
}

int tsd_occupancy(tsd_ctx* ctx, int8_t* occ_host, int inflate, int inflate_factor, int* n_surface)
{
  if (!ctx || !occ_host) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const size_t cells = (size_t)ctx->grid.N * ctx->grid.N;
  if (!ctx->d_occ_out) TSD_HIP_CHECK(ctx, hipMalloc(&ctx->d_occ_out, cells));      // once per context (see tsd_color_image)
  // The map leaves on a stream of its own behind the extraction kernels' event: 16 MiB at 4096^2 are ~0.7 ms of PCIe, and on the grid's
  // stream every push and ray cast enqueued meanwhile (the localisers keep running: ThreadGrid.cpp:72-133 extracts beside them) would
  // sit behind the copy; the caller waits for the copy stream only.
  if (!ctx->stream_io) {
    TSD_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->stream_io, hipStreamNonBlocking));
    TSD_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_io, hipEventDisableTiming));
  }
  int rc = launch_occupancy(ctx, ctx->d_occ_out, inflate, inflate_factor);
  if (rc == TSD_OK) {
    int n = 0;
    hipError_t e = hipEventRecord(ctx->ev_io, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream_io, ctx->ev_io, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(occ_host, ctx->d_occ_out, cells, hipMemcpyDeviceToHost, ctx->stream_io);
    if (e == hipSuccess) e = hipMemcpyAsync(&n, ctx->d_occ_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream_io);
    const hipError_t es = hipStreamSynchronize(ctx->stream_io);
    if (e == hipSuccess) e = es;
    if (n_surface) *n_surface = n;
    if (e != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_occupancy copy", e);
  }
  return rc;
}

int tsd_calibrate_rmw(tsd_ctx* ctx, int64_t n_doubles, int reps)
{
  if (!ctx || n_doubles <= 0 || reps <= 0) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  double *t = nullptr, *w = nullptr;
  TSD_HIP_CHECK(ctx, hipMalloc(&t, (size_t)n_doubles * sizeof(double)));
  hipError_t e = hipMalloc(&w, (size_t)n_doubles * sizeof(double));
  if (e != hipSuccess) { hipFree(t); return set_error(ctx, TSD_E_HIP, "tsd_calibrate_rmw", e); }
  hipMemsetAsync(t, 0, (size_t)n_doubles * sizeof(double), ctx->stream);
  hipMemsetAsync(w, 0, (size_t)n_doubles * sizeof(double), ctx->stream);
  int rc = TSD_OK;
  for (int r = 0; r < reps && rc == TSD_OK; r++) rc = launch_calibrate(ctx, t, w, (size_t)n_doubles);
  hipStreamSynchronize(ctx->stream);
  hipFree(t); hipFree(w);
  return rc;
}

int tsd_measure_stream(tsd_ctx* ctx, int64_t n_doubles, int reps, double* gbs_best, double* gbs_mean)
{
  if (!ctx || n_doubles <= 0 || reps <= 0 || reps > 64) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  double *t = nullptr, *w = nullptr;
  const size_t bytes = (size_t)n_doubles * sizeof(double);
  TSD_HIP_CHECK(ctx, hipMalloc(&t, bytes));
  hipError_t e = hipMalloc(&w, bytes);
  if (e != hipSuccess) { hipFree(t); return set_error(ctx, TSD_E_HIP, "tsd_measure_stream", e); }
  hipMemsetAsync(t, 0, bytes, ctx->stream);
  hipMemsetAsync(w, 0, bytes, ctx->stream);
  int rc = launch_calibrate(ctx, t, w, (size_t)n_doubles);      // untimed: first touch
  std::vector<hipEvent_t> ev((size_t)reps + 1, nullptr);
  for (auto& x : ev) if (hipEventCreate(&x) != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_measure_stream: events", hipGetLastError());
  if (rc == TSD_OK) hipEventRecord(ev[0], ctx->stream);
  for (int r = 0; r < reps && rc == TSD_OK; r++) { rc = launch_calibrate(ctx, t, w, (size_t)n_doubles); hipEventRecord(ev[(size_t)r + 1], ctx->stream); }
  hipStreamSynchronize(ctx->stream);
  double best = 0.0, sum = 0.0; int n = 0;
  for (int r = 0; r < reps && rc == TSD_OK; r++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev[(size_t)r], ev[(size_t)r + 1]) != hipSuccess || !(ms > 0.f)) continue;
    const double gbs = 4.0 * (double)bytes / ((double)ms * 1e-3) / 1e9;     // two arrays, each read and written once
    if (gbs > best) best = gbs;
    sum += gbs; n++;
  }
  for (auto x : ev) if (x) hipEventDestroy(x);
  hipFree(t); hipFree(w);
  if (rc != TSD_OK) return rc;
  if (!n) return set_error(ctx, TSD_E_HIP, "tsd_measure_stream: no timed launch", hipSuccess);
  if (gbs_best) *gbs_best = best;
  if (gbs_mean) *gbs_mean = sum / n;
  return TSD_OK;
}

int tsd_profile_enable(tsd_ctx* ctx, int on)
{
  if (!ctx) return TSD_E_ARG;
  ctx->profile = on != 0;
  if (ctx->profile && ctx->profile_mask == 0) ctx->profile_mask = ~0u;
  return TSD_OK;
}

int tsd_profile_select(tsd_ctx* ctx, const char* kernels_csv)
{
  if (!ctx || !kernels_csv) return TSD_E_ARG;
  // "a,b,c/n": the listed kernels (or "all"), every n-th launch of each; a name may carry its own period, "a:m", which
  // wins over the list's (bench.py times EVERY dispatch of the roofline kernel in a short run and every n-th of the others)
  unsigned mask = 0;
  std::string csv(kernels_csv);
  ctx->profile_every = 1;
  const size_t slash = csv.rfind('/');
  if (slash != std::string::npos) {
    const int n = std::atoi(csv.c_str() + slash + 1);
    ctx->profile_every = n > 1 ? (unsigned)n : 1u;
    csv = csv.substr(0, slash);
  }
  constexpr unsigned NK = sizeof(kKernelNames) / sizeof(kKernelNames[0]);
  for (unsigned i = 0; i < NK; i++) ctx->profile_every_k[i] = 0;
  size_t pos = 0;
  while (pos <= csv.size()) {
    size_t end = csv.find(',', pos);
    if (end == std::string::npos) end = csv.size();
    std::string tok = csv.substr(pos, end - pos);
    pos = end + 1;
    if (tok.empty()) continue;
    unsigned own = 0;
    const size_t colon = tok.find(':');
    if (colon != std::string::npos) { const int m = std::atoi(tok.c_str() + colon + 1); own = m >= 1 ? (unsigned)m : 1u; tok = tok.substr(0, colon); }
    for (unsigned i = 0; i < NK; i++)
      if (tok == "all" || tok == kKernelNames[i]) { mask |= 1u << i; if (own && tok != "all") ctx->profile_every_k[i] = own; }
  }
  ctx->profile_mask = mask;
  return TSD_OK;
}

int tsd_profile_reset(tsd_ctx* ctx)
{
  if (!ctx) return TSD_E_ARG;
  hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  ctx->timers.clear();
  return TSD_OK;
}

int tsd_profile_get(tsd_ctx* ctx, const char* kernel, double* total_ms, int* launches)
{
  if (!ctx || !kernel) return TSD_E_ARG;
  hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  auto it = ctx->timers.find(kernel);
  if (total_ms) *total_ms = (it == ctx->timers.end()) ? 0.0 : it->second.total_ms;
  if (launches) *launches = (it == ctx->timers.end()) ? 0 : it->second.launches;
  return TSD_OK;
}

int tsd_profile_get_spread(tsd_ctx* ctx, const char* kernel, double* min_ms, double* max_ms, double* std_ms)
{
  if (!ctx || !kernel) return TSD_E_ARG;
  hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  std::lock_guard<std::mutex> lk(ctx->misc_mutex);
  auto it = ctx->timers.find(kernel);
  const bool have = it != ctx->timers.end() && it->second.launches > 0;
  const double n = have ? (double)it->second.launches : 1.0;
  const double mean = have ? it->second.total_ms / n : 0.0;
  double var = have ? it->second.sum_sq / n - mean * mean : 0.0;
  if (var < 0.0) var = 0.0;
  if (min_ms) *min_ms = have ? it->second.min_ms : 0.0;
  if (max_ms) *max_ms = have ? it->second.max_ms : 0.0;
  if (std_ms) *std_ms = std::sqrt(var);
  return TSD_OK;
}

int tsd_profile_get_samples(tsd_ctx* ctx, const char* kernel, float* ms_out, int cap)
{
  if (!ctx || !kernel || cap < 0 || (cap > 0 && !ms_out)) return TSD_E_ARG;
  hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  std::lock_guard<std::mutex> lk(ctx->misc_mutex);
  auto it = ctx->timers.find(kernel);
  if (it == ctx->timers.end()) return 0;
  const int n = (int)std::min(it->second.samples.size(), (size_t)cap);
  for (int i = 0; i < n; i++) ms_out[i] = it->second.samples[(size_t)i];
  return (int)it->second.samples.size();
}

int tsd_push_stats_total(tsd_ctx* ctx, tsd_push_stats* total, int64_t* pushes, int reset)
{
  if (!ctx) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  return read_total_stats(ctx, total, pushes, reset != 0);
}

// ---------------------------------------------------------------------------------- fused scan path
tsd_sensor* tsd_sensor_create(tsd_ctx* ctx, int beams, double ang_res, double phi_min, double max_range,
                              double min_range, double low_refl_range)
{
  if (!ctx || beams < 1 || beams > TSD_MAX_BEAMS || beams > TSD_MAX_ICP_POINTS) return nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  tsd_sensor* s = new (std::nothrow) tsd_sensor();
  if (!s) return nullptr;
  s->ctx = ctx; s->beams = beams; s->ang_res = ang_res; s->phi_min = phi_min;
  s->max_range = max_range; s->min_range = min_range; s->low_refl = low_refl_range;
  bool ok = true;
  auto A = [&](hipError_t e) { if (e != hipSuccess) ok = false; };
  const size_t nb = (size_t)beams;
  A(hipMalloc(&s->d_state, sizeof(SensorDev)));
  A(hipMalloc(&s->d_rays, nb * 16));
  A(hipMalloc(&s->d_rays_local, nb * 16));
  for (int i = 0; i < 3; i++) A(hipMalloc(&s->d_scan2[i], nb * 10 + 64));
  // the scan result is written by the kernel straight into coherent pinned host memory
  A(hipHostMalloc(&s->h_result, sizeof(ScanResultDev), hipHostMallocMapped | hipHostMallocCoherent));
  if (ok) { std::memset(s->h_result, 0, sizeof(ScanResultDev)); A(hipHostGetDevicePointer((void**)&s->d_result, s->h_result, 0)); }
  if (!ok) { tsd_sensor_destroy(s); return nullptr; }
  ctx->sensors.push_back(s);
  return s;
}

void tsd_sensor_destroy(tsd_sensor* s)
{
  if (!s) return;
  if (s->ctx) {
    hipSetDevice(s->ctx->device); hipStreamSynchronize(s->ctx->stream2); hipStreamSynchronize(s->ctx->stream);
    auto& v = s->ctx->sensors;
    v.erase(std::remove(v.begin(), v.end(), s), v.end());
  }
  if (s->stream) hipStreamSynchronize(s->stream);
  for (hipEvent_t e : {s->ev_rc_done, s->ev_icp_done}) if (e) hipEventDestroy(e);
  if (s->stream) hipStreamDestroy(s->stream);
  hipFree(s->d_coords); hipFree(s->d_normals); hipFree(s->d_mask_m); hipFree(s->d_icp_res); hipFree(s->d_icp_seed); hipFree(s->d_icp_trace);
  if (s->ev_pre) hipEventDestroy(s->ev_pre);
  if (s->ev_pre_done) hipEventDestroy(s->ev_pre_done);
  if (s->d_push_slot) { if (s->ctx && s->ctx->stream_push) hipStreamSynchronize(s->ctx->stream_push); hipFree(s->d_push_slot); }
  for (int i = 0; i < 3; i++)
    if (s->ev_slot_push[i]) {
      if (s->ctx && s->ctx->ev_async_push == s->ev_slot_push[i]) {      // (the push stream was drained above)
        if (s->ctx->async_pending) hipStreamWaitEvent(s->ctx->stream, s->ev_slot_push[i], 0);
        s->ctx->async_pending = false; s->ctx->ev_async_push = nullptr;
      }
      hipEventDestroy(s->ev_slot_push[i]);
    }
  if (s->d_pre) hipFree(s->d_pre);
  if (s->h_pre) hipHostFree(s->h_pre);
  hipFree(s->d_rmq2[0]); hipFree(s->d_rmq2[1]); hipFree(s->d_rmq2[2]);
  if (s->h_stage2[0]) hipHostFree(s->h_stage2[0]);
  if (s->h_stage2[1]) hipHostFree(s->h_stage2[1]);
  hipFree(s->d_state); hipFree(s->d_rays); hipFree(s->d_rays_local); hipFree(s->d_scan2[0]); hipFree(s->d_scan2[1]); hipFree(s->d_scan2[2]);
  hipHostFree(s->h_result);
  delete s;
}

int tsd_sensor_set_pose(tsd_sensor* s, const double pose33[9], const double* rays_world_2xB,
                        const double* rays_local_2xB)
{
  if (s && s->ctx) s->ctx->epoch++;
  if (!s || !s->ctx || !pose33 || !rays_world_2xB || !rays_local_2xB) return TSD_E_ARG;
  tsd_ctx* ctx = s->ctx;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const size_t nb = (size_t)s->beams;
  SensorDev st;
  std::memset(&st, 0, sizeof(st));
  for (int i = 0; i < 9; i++) st.pose[i] = pose33[i];
  s->pos[0] = pose33[2]; s->pos[1] = pose33[5];
  st.have_last_pose = 0;
  // constant parts of the kernel arguments (the pose dependent parts are derived on the device)
  st.rc.idx_min = s->min_range / ctx->grid.cs;
  st.rc.idx_max = s->max_range / ctx->grid.cs;
  st.rc.beams = s->beams;
  st.push.phi_min = s->phi_min; st.push.ang_res_inv = 1.0 / s->ang_res;
  st.push.phi_lower = -0.5 * s->ang_res + s->phi_min;                    // SensorPolar2D.cpp:26-30
  st.push.phi_upper = s->phi_min + (((double)s->beams) - 0.5) * s->ang_res;
  st.push.max_range = s->max_range; st.push.min_range = s->min_range; st.push.low_refl = s->low_refl;
  st.push.beams = s->beams; st.push.enabled = 0;
  s->ccw = (s->beams < 2) || (rays_local_2xB[0] * rays_local_2xB[nb + 1] - rays_local_2xB[nb] * rays_local_2xB[1] >= 0.0);
  int slot;
  char* h = stage_acquire(ctx, &slot);
  if (nb * 32 + sizeof(SensorDev) > ctx->stage_bytes) return set_error(ctx, TSD_E_CAPACITY, "sensor staging", hipSuccess);
  std::memcpy(h, rays_world_2xB, nb * 16);
  std::memcpy(h + nb * 16, rays_local_2xB, nb * 16);
  std::memcpy(h + nb * 32, &st, sizeof(SensorDev));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(s->d_rays, h, nb * 16, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(s->d_rays_local, h + nb * 16, nb * 16, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(s->d_state, h + nb * 32, sizeof(SensorDev), hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[slot], ctx->stream));
  int rc = launch_scan_prepare(ctx, s->d_state);
  if (rc != TSD_OK) return rc;
  s->posed = true;
  return TSD_OK;
}

static int sensor_conc_init(tsd_sensor* s, bool own_stream);

// copy + range-query tables of one scan on the side stream, into the sensor's buffers the scan in flight does not use
static int scan_stage_impl(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push)
{
  tsd_ctx* ctx = s->ctx;
  const size_t nb = (size_t)s->beams;
  if (int rc = sensor_conc_init(s, false)) return rc;       // (the sensor's own table buffers)
  // One H2D: ranges | mask | mask_push, on the side stream into the buffer the previous scan does not use: the
  // copy and the range-query tables of this scan's push (which only depend on the scan) run while the previous
  // push and this scan's ray cast are still busy on the main stream.
  int slot;
  unsigned long long tl = g_stage_timing.on ? now_ns() : 0;
  auto LAP = [&](int i) { if (g_stage_timing.on) { const unsigned long long u = now_ns(); g_stage_timing.ns[i] += u - tl; tl = u; } };
  char* h = stage_acquire(ctx, &slot);
  LAP(0);
  std::memcpy(h, ranges, nb * 8);
  std::memcpy(h + nb * 8, mask, nb);
  std::memcpy(h + nb * 9, mask_push ? mask_push : mask, nb);
  LAP(1);
  // three buffers in turn (see tsd_sensor::stage_slot): the push that read this one three scans ago is done
  const int sslot = s->stage_slot;
  char* d_scan = s->d_scan2[sslot];
  s->stage_slot = (s->stage_slot + 1) % 3;
  // Asynchronous mapping: the push that last read this buffer (three scans back) ran on the push stream beside a registration, and
  // nothing the host has seen since is ordered behind it -- the copy and the tables below wait for that push's own event (done long
  // ago in practice: one query; the stream-side wait is the fall-back).  Strict order: see tsd_sensor::stage_slot.
  if (s->slot_push_valid[sslot]) {
    if (!host_saw_event(s->ev_slot_push[sslot], 0)) TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream2, s->ev_slot_push[sslot], 0));
    else s->slot_push_valid[sslot] = false;
  }
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(d_scan, h, nb * 10, hipMemcpyHostToDevice, ctx->stream2));
  LAP(2);
  TSD_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[slot], ctx->stream2));
  TSD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_h2d, ctx->stream2));
  LAP(3);
  s->st_ranges = reinterpret_cast<const double*>(d_scan);
  s->st_mask = reinterpret_cast<const uint8_t*>(d_scan + nb * 8);
  s->st_mask_push = reinterpret_cast<const uint8_t*>(d_scan + nb * 9);
  s->st_rmq = s->d_rmq2[sslot]; s->st_slot = sslot;
  LaunchTarget tg;
  tg.rmq = s->st_rmq;
  TargetScope scope(ctx, &tg);
  int rc = launch_push_tables(ctx, ctx->stream2, s->beams, s->st_ranges, s->st_mask_push, s->phi_min, s->ang_res);
  if (rc != TSD_OK) return rc;
  TSD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_tables, ctx->stream2));
  // make sure the side stream's commands are on their way now: with more streams in the process than hardware queues (a
  // communicator's, a framework's) the runtime was seen to hold them back until the next synchronisation, and the
  // registration that waits for this copy with them (a 40 ms stall once in ~200 scans)
  (void)hipStreamQuery(ctx->stream2);
  LAP(4);
  if (g_stage_timing.on) g_stage_timing.n++;
  s->staged = true;
  return TSD_OK;
}

int tsd_sensor_set_async_mapping(tsd_sensor* s, int on)
{
  if (!s || !s->ctx) return TSD_E_ARG;
  tsd_ctx* ctx = s->ctx;
  if (s->submitted) return set_error(ctx, TSD_E_ARG, "tsd_sensor_set_async_mapping: a scan is in flight", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  if (on) {
    if (!ctx->stream_push) TSD_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->stream_push, hipStreamNonBlocking));
    // (both events order kernels of ONE device against each other: no system-scope fence when they complete)
    if (!ctx->ev_async_rc) TSD_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_async_rc, hipEventDisableTiming | hipEventDisableSystemFence));
    // "the push of the scan that used scan / table buffer i is done": one event per buffer, so that the staging of a later scan
    // into that buffer can be ordered behind the push that last read it (scan_stage_impl); ctx->ev_async_push is the newest of them
    for (int i = 0; i < 3; i++)
      if (!s->ev_slot_push[i]) TSD_HIP_CHECK(ctx, hipEventCreateWithFlags(&s->ev_slot_push[i], hipEventDisableTiming | hipEventDisableSystemFence));
    if (!s->d_push_slot) TSD_HIP_CHECK(ctx, hipMalloc(&s->d_push_slot, 2 * sizeof(tsd::PushArgs)));
  }
  s->async_mapping = on != 0;
  // a ray cast enqueued ahead by the previous scan saw (strict) or did not see (asynchronous) that scan's push: the next scan of the
  // other kind casts again
  s->rc_pending = false;
  return TSD_OK;
}

int tsd_debug_stall_push_stream(tsd_ctx* ctx, unsigned int microseconds)
{
  if (!ctx) return TSD_E_ARG;
  ctx->debug_push_stall_us = microseconds;
  return TSD_OK;
}

int tsd_debug_set_icp_helpers(tsd_ctx* ctx, int on)
{
  if (!ctx) return TSD_E_ARG;
  ctx->icp_helpers = on ? 1 : 0;
  return TSD_OK;
}

int tsd_scan_stage(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push)
{
  if (!s || !s->ctx || !ranges || !mask) return TSD_E_ARG;
  tsd_ctx* ctx = s->ctx;
  if (!s->posed) return set_error(ctx, TSD_E_ARG, "tsd_scan_stage before tsd_sensor_set_pose", hipSuccess);
  if (s->staged) return set_error(ctx, TSD_E_ARG, "tsd_scan_stage: a staged scan is waiting for tsd_scan_submit already", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return scan_stage_impl(s, ranges, mask, mask_push);
}

int tsd_scan_submit(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push,
                    const tsd_icp_params* params, const tsd_gate_params* gates)
{
  if (!s || !s->ctx || !params || !gates || (ranges && !mask)) return TSD_E_ARG;
  tsd_ctx* ctx = s->ctx;
  if (!s->posed) return set_error(ctx, TSD_E_ARG, "tsd_scan before tsd_sensor_set_pose", hipSuccess);
  if (s->submitted) return set_error(ctx, TSD_E_ARG, "tsd_scan_submit: the previous scan was not collected", hipSuccess);
  if (!ranges && !s->staged) return set_error(ctx, TSD_E_ARG, "tsd_scan_submit without a scan (none given, none staged)", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ScanLap lap;
  if (g_scan_timing.on && g_scan_last_return) {
    const unsigned long long gap = lap.t - g_scan_last_return;
    if (gap < 1000000ull) g_scan_timing.ns[0] += gap;   // (one-off pauses of the caller excluded from the average)
    if (gap > g_scan_lap_max[0] && g_scan_timing.n > 8) { g_scan_lap_max[0] = gap; g_scan_lap_max_at[0] = (unsigned long long)g_scan_timing.n; }
  }
  const bool staged_ahead = ranges == nullptr;
  if (ranges) {
    // A scan staged ahead that is not the one that came is dropped -- and its buffers are REUSED for the scan that did come: the
    // three-buffer rotation is only safe when it advances once per scan (the buffer two rotations back may still be read by the
    // push of the previous scan, which is ordered behind nothing the host has seen).  The new copy and tables follow the dropped
    // ones on the side stream, and nothing else ever read the dropped data.
    if (s->staged) s->stage_slot = s->st_slot;
    s->staged = false;
    int rcs = scan_stage_impl(s, ranges, mask, mask_push);
    if (rcs != TSD_OK) return rcs;
  }
  s->staged = false;
  const double* d_ranges = s->st_ranges;
  const uint8_t* d_mask = s->st_mask;
  const uint8_t* d_mask_push = s->st_mask_push;
  int rc = TSD_OK;
  lap.lap(1);

  // The ray cast needs nothing from the scan (its pose arguments were left on the device by the previous
  // registration), so the previous tsd_scan enqueued it right behind its push; it is launched here only if
  // something touched the grid, the sensor or the context's ray-cast outputs since.
  RaycastArgs ra;
  std::memset(&ra, 0, sizeof(ra));
  ra.beams = s->beams;                                   // grid size of the launch; the rest is read on the device
  if (!(s->rc_pending && s->rc_epoch == ctx->epoch)) {
    if (int rcd_ = drain_async_push(ctx)) return rcd_;     // (asynchronous mapping: a push still on the push stream comes first)
    rc = launch_raycast(ctx, ra, &s->d_state->rc, s->d_rays);
    if (rc != TSD_OK) return rc;
  }
  s->rc_pending = false;
  lap.lap(2);
  IcpArgs ia;
  const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  fill_icp_args(ia, ident, params);
  ia.beams = s->beams; ia.ccw = s->ccw ? 1 : 0;
  // registration_mode 3: the pre-registration armed by tsd_scan_preregister runs here, between the ray cast and the registration,
  // whose Tinit it leaves on the device
  s->pre_ran = false;
  if (s->pre_armed) {
    s->pre_armed = false;
    // (asynchronous mapping: the SCORING reads the grid -- the previous scan's push, still on the push stream, has to land first; the
    // normals and the list building ahead of it do not, and run beside that push.  The order is then ray cast, previous push,
    // pre-registration, registration: still one of the reference's interleavings)
    hipEvent_t before_score = nullptr;
    if (ctx->async_pending) { before_score = ctx->ev_async_push; ctx->async_pending = false; }
    const LaunchTarget* tgp = launch_target();
    rc = launch_preregistration(ctx, s, launch_stream(ctx), tgp && tgp->coords ? tgp->coords : ctx->d_coords,
                                tgp && tgp->mask_m ? tgp->mask_m : ctx->d_mask_m, s->d_state->icpP, &ia.Tinit_dev, before_score);
    if (rc != TSD_OK) return rc;
    s->pre_ran = true;
  }
  // the gates, Sensor::transform and the push decision run as the epilogue of the registration kernel
  const unsigned long long seq = ++s->seq;
  ScanPostArgs sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.st = s->d_state; sp.rays = s->d_rays; sp.out = s->d_result; sp.seq = seq; sp.beams = s->beams;
  sp.gmin_x = ctx->grid.min_x; sp.gmax_x = ctx->grid.max_x; sp.gmin_y = ctx->grid.min_y; sp.gmax_y = ctx->grid.max_y;
  sp.gates = GateArgs{gates->reg_trs_max, gates->reg_sin_rot_max, gates->trs_min, gates->rot_min};
  const bool async_map = s->async_mapping && s->d_push_slot != nullptr;
  tsd::PushArgs* const push_slot = async_map ? s->d_push_slot + (seq & 1ull) : nullptr;
  sp.push_copy = push_slot;
  // The ray cast did not need the scan, the registration does.  The copy is short and the ray cast long, so the
  // host waits for the copy itself (a few microseconds, the device is busy meanwhile) instead of putting a
  // cross-stream barrier between the two kernels; the barrier is the fall-back.  (A scan staged ahead was copied
  // during the previous registration: nothing to wait for.)
  if (!host_saw_event(ctx->ev_h2d, staged_ahead ? 2 : 40)) TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_h2d, 0));
  rc = launch_icp(ctx, ia, s->d_state->icpP, s->d_rays_local, d_ranges, d_mask, &sp);
  if (rc != TSD_OK) return rc;
  lap.lap(3);
  PushArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.beams = s->beams;                                   // LDS size of the launch
  pa.max_range = s->max_range;                           // tile window of the launch (the rest is read on the device)
  if (!async_map) {
    if (int rcd_ = drain_async_push(ctx)) return rcd_;   // (a push left on the push stream by an earlier, asynchronous scan)
    if (!host_saw_event(ctx->ev_tables, staged_ahead ? 2 : 60)) TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_tables, 0));
    {
      LaunchTarget tg;
      tg.rmq = s->st_rmq;                                  // this scan's tables (the sensor's own buffers)
      TargetScope scope(ctx, &tg);
      // the registration moves the sensor by at most the gate (a larger step is rejected: pose unchanged)
      rc = launch_push(ctx, pa, s->pos[0], s->pos[1], gates->reg_trs_max, &s->d_state->push, d_ranges, d_mask_push);
    }
    if (rc != TSD_OK) return rc;
    ctx->epoch++;                                          // the grid changes
    lap.lap(4);
    // the next scan's ray cast, right behind the push (see above): the host's work on the next scan no longer sits
    // between this push and that ray cast
    rc = launch_raycast(ctx, ra, &s->d_state->rc, s->d_rays);
    if (rc != TSD_OK) return rc;
    s->rc_pending = true; s->rc_epoch = ctx->epoch;
    lap.lap(5);
  } else {
    // Asynchronous mapping (the reference's ThreadMapping: queuePush returns at once and the push lands when the mapping thread gets
    // to it, ThreadMapping.cpp:51-76): the NEXT scan's ray cast goes right behind this registration, on a grid that does not hold
    // this scan's push yet -- exactly one push behind, every scan -- and this scan's push runs beside the next registration on the
    // push stream.  Grid accesses stay ordered: ray cast (k+1) behind push (k-1) [first wait], push (k) behind ray cast (k+1)
    // [second wait]; the push reads its own copy of its arguments (the next registration's epilogue rewrites the sensor's).
    if (ctx->async_pending) { TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_async_push, 0)); ctx->async_pending = false; }
    {
      // (the event the push waits for is the ray cast's own completion -- a marker behind it would sit between the ray cast and the
      // next registration)
      LaunchTarget tgr;
      tgr.rc_done = ctx->ev_async_rc;
      TargetScope scope_r(ctx, &tgr);
      rc = launch_raycast(ctx, ra, &s->d_state->rc, s->d_rays);
      if (rc != TSD_OK) return rc;
      if (!tgr.rc_done_used) TSD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_async_rc, ctx->stream));
    }
    lap.lap(4);
    TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream_push, ctx->ev_async_rc, 0));
    TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream_push, ctx->ev_tables, 0));
    {
      LaunchTarget tg;
      tg.rmq = s->st_rmq;
      TargetScope scope(ctx, &tg);
      if (ctx->debug_push_stall_us) launch_stall(ctx, ctx->stream_push, ctx->debug_push_stall_us);     // (tests: a push stream that lags)
      rc = launch_push(ctx, pa, s->pos[0], s->pos[1], gates->reg_trs_max, push_slot, d_ranges, d_mask_push, ctx->stream_push);
    }
    if (rc != TSD_OK) return rc;
    ctx->ev_async_push = s->ev_slot_push[s->st_slot];      // (st_slot: the buffers of the scan being submitted)
    s->slot_push_valid[s->st_slot] = true;
    TSD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_async_push, ctx->stream_push));
    (void)hipStreamQuery(ctx->stream_push);
    ctx->async_pending = true;
    ctx->epoch++;                                          // the grid changes ...
    s->rc_pending = true; s->rc_epoch = ctx->epoch;        // ... and the ray cast enqueued above is, by design, the one that does not see it
    lap.lap(5);
  }
  s->submitted = true;
  return TSD_OK;
}

int tsd_scan_collect(tsd_sensor* s, tsd_scan_result* result)
{
  if (!s || !s->ctx || !result) return TSD_E_ARG;
  tsd_ctx* ctx = s->ctx;
  if (!s->submitted) return set_error(ctx, TSD_E_ARG, "tsd_scan_collect without tsd_scan_submit", hipSuccess);
  s->submitted = false;
  ScanLap lap;
  const unsigned long long seq = s->seq;
  // The result is known once k_scan_post has run; the push kernels behind it only touch the grid, and
  // whatever the caller enqueues next is ordered behind them on the stream.  So the host does not wait for
  // the stream: it polls the sequence number and prepares the next scan while the push is still running.
  {
    volatile unsigned long long* vseq = &s->h_result->seq;
    unsigned long long spins = 0;
    while (__atomic_load_n(vseq, __ATOMIC_ACQUIRE) != seq) {
      ++spins;
      // A registration takes 0.15-0.3 ms.  Past that, nudge the runtime: with other streams in the process (a communicator's,
      // a framework's) it was seen to sit on an enqueued launch until the next query / synchronisation of the stream -- a
      // 40 ms stall at the same scan of every run (profiles/r2_dist_stall.txt); a stream query is a few microseconds.
      if ((spins & 0x3FFFull) == 0) { (void)hipStreamQuery(ctx->stream); (void)hipStreamQuery(ctx->stream2); }
      if (spins > 4000000ull) {            // ~ a tenth of a second: something is wrong, fall back to a real wait
        TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (__atomic_load_n(vseq, __ATOMIC_ACQUIRE) != seq)
          return set_error(ctx, TSD_E_HIP, "tsd_scan: result record never arrived", hipSuccess);
        break;
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
  lap.lap(6);
  copy_icp_result(&s->h_result->icp, &result->icp);
  for (int i = 0; i < 9; i++) result->pose[i] = s->h_result->pose[i];
  s->pos[0] = result->pose[2]; s->pos[1] = result->pose[5];
  result->reg_error = s->h_result->reg_error; result->pushed = s->h_result->pushed;
  result->no_model = s->h_result->no_model; result->reserved = 0;
  lap.lap(7);
  if (g_scan_timing.on) { g_scan_timing.n++; g_scan_last_return = now_ns(); }
  return TSD_OK;
}

int tsd_scan(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push,
             const tsd_icp_params* params, const tsd_gate_params* gates, tsd_scan_result* result)
{
  if (!s || !s->ctx || !ranges || !mask || !params || !gates || !result) return TSD_E_ARG;
  const int rc = tsd_scan_submit(s, ranges, mask, mask_push, params, gates);
  if (rc != TSD_OK) return rc;
  return tsd_scan_collect(s, result);
}


// ------------------------------------------------------------------ concurrent multi-robot scans (one shared grid)
// The reference's multi-robot mode is N ThreadLocalize workers on ONE TsdGrid (SlamNode.cpp:101-122).  With tsd_scan
// every robot's whole scan sits on the grid's one stream, so N robots run their 0.17 ms registrations -- which do not
// touch the grid at all and occupy ONE compute unit each -- back to back while 255 CUs idle.  Here a scan is split:
//   tsd_scan_begin   from the robot's own thread: copy, tables, ray cast and registration (+ gates, Sensor::transform)
//                    on the SENSOR's own stream into the sensor's own buffers
//   tsd_scan_wait    the thread waits for the result record (written right after the registration)
//   tsd_scan_finish  the push on the GRID stream (pushes of all robots are serialised there, like the reference's one
//                    ThreadMapping serialises them)
// Ordering is by events, in the order the calls reach two short sections locked by ctx->order_mutex: a ray cast waits
// for every grid write enqueued before it, a push for every ray cast ticketed since the last grid write; registrations
// overlap freely.  The push is enqueued only once its registration has finished: events order by ENQUEUE time, so a
// push enqueued ahead of time would pull every later ray cast of every robot behind its own registration.
static int sensor_conc_init(tsd_sensor* s, bool own_stream)
{
  tsd_ctx* ctx = s->ctx;
  bool ok = true;
  auto A = [&](hipError_t e) { if (e != hipSuccess) ok = false; };
  if (own_stream && !s->stream) {            // (a sensor that only ever runs in batches uses the batch's stream and events)
    A(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    for (hipEvent_t* e : {&s->ev_rc_done, &s->ev_icp_done}) A(hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (!ok) return set_error(ctx, TSD_E_HIP, "tsd_scan_begin: per-sensor stream", hipGetLastError());
  }
  if (s->conc_ready) return TSD_OK;
  const size_t nb = (size_t)s->beams;
  A(hipMalloc(&s->d_coords, nb * 16)); A(hipMalloc(&s->d_normals, nb * 16)); A(hipMalloc(&s->d_mask_m, nb));
  A(hipMalloc(&s->d_icp_res, sizeof(IcpResultDev))); A(hipMalloc(&s->d_icp_trace, sizeof(double) * TSD_ICP_TRACE_STRIDE * TSD_ICP_TRACE_MAX));
  A(hipMalloc(&s->d_icp_seed, icp_seed_bytes(s->beams)));
  if (ok) A(hipMemset(s->d_icp_seed, 0, icp_seed_bytes(s->beams)));
  for (int i = 0; i < 3; i++) A(hipMalloc(&s->d_rmq2[i], push_rmq_bytes(s->beams)));
  A(hipHostMalloc(&s->h_stage2[0], nb * 10 + 64, hipHostMallocDefault)); A(hipHostMalloc(&s->h_stage2[1], nb * 10 + 64, hipHostMallocDefault));
  if (!ok) return set_error(ctx, TSD_E_HIP, "tsd_scan_begin: per-sensor streams / buffers", hipGetLastError());
  s->conc_ready = true;
  return TSD_OK;
}

int tsd_scan_begin(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push,
                   const tsd_icp_params* params, const tsd_gate_params* gates)
{
  if (!s || !s->ctx || !ranges || !mask || !params || !gates) return TSD_E_ARG;
  tsd_ctx* ctx = s->ctx;
  if (!s->posed) return set_error(ctx, TSD_E_ARG, "tsd_scan_begin before tsd_sensor_set_pose", hipSuccess);
  if (s->inflight) return set_error(ctx, TSD_E_ARG, "tsd_scan_begin: the previous scan of this sensor was not finished", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  int rc = sensor_conc_init(s, true);
  if (rc != TSD_OK) return rc;
  const size_t nb = (size_t)s->beams;
  ConcLap lap;
  // the scan: ranges | mask | mask_push through the sensor's own pinned buffer (the buffer two scans back is free: its
  // copy was waited for by that scan's registration)
  char* h = s->h_stage2[s->scan_slot];
  char* d_scan = s->d_scan2[s->scan_slot];
  s->scan_slot ^= 1;
  std::memcpy(h, ranges, nb * 8);
  std::memcpy(h + nb * 8, mask, nb);
  std::memcpy(h + nb * 9, mask_push ? mask_push : mask, nb);
  // (copy and tables on the sensor's ONE stream, ahead of the ray cast: every further stream is one more candidate for
  // sharing a hardware queue with another robot's 0.17 ms registration -- HIP multiplexes streams onto a few in-order
  // hardware queues, GPU_MAX_HW_QUEUES -- and 17 us ahead of a 190 us chain is the cheaper price)
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(d_scan, h, nb * 10, hipMemcpyHostToDevice, s->stream));
  const double* d_ranges = reinterpret_cast<const double*>(d_scan);
  const uint8_t* d_mask = reinterpret_cast<const uint8_t*>(d_scan + nb * 8);
  const uint8_t* d_mask_push = reinterpret_cast<const uint8_t*>(d_scan + nb * 9);
  s->rmq_slot ^= 1;                         // (the previous push of this sensor may still read its tables)
  LaunchTarget tg;
  tg.stream = s->stream; tg.coords = s->d_coords; tg.normals = s->d_normals; tg.mask_m = s->d_mask_m;
  tg.icp_res = s->d_icp_res; tg.trace = s->d_icp_trace; tg.icp_seed = s->d_icp_seed; tg.icp_seed_points = s->beams; tg.rmq = s->d_rmq2[s->rmq_slot];
  TargetScope scope(ctx, &tg);
  rc = launch_push_tables(ctx, s->stream, s->beams, d_ranges, d_mask_push, s->phi_min, s->ang_res);
  if (rc != TSD_OK) return rc;
  lap.lap(0);
  {
    // ORDERED SECTION (the only part of begin that other robots' threads wait for): the ray cast reads the grid, so it
    // goes behind every grid write enqueued so far, and takes its place in the order for the writes that follow
    std::lock_guard<std::mutex> lk(ctx->order_mutex);
    lap.lap(1);
    TSD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_grid, ctx->stream));
    TSD_HIP_CHECK(ctx, hipStreamWaitEvent(s->stream, ctx->ev_grid, 0));
    __atomic_store_n(&s->rc_recorded, 0, __ATOMIC_RELEASE);
    s->rc_ticket = ++ctx->ticket;
    s->rc_event_valid = true;
    lap.lap(2);
  }
  RaycastArgs ra;
  std::memset(&ra, 0, sizeof(ra));
  ra.beams = s->beams;
  rc = launch_raycast(ctx, ra, &s->d_state->rc, s->d_rays);
  const hipError_t e_rc = hipEventRecord(s->ev_rc_done, s->stream);
  __atomic_store_n(&s->rc_recorded, 1, __ATOMIC_RELEASE);      // (always: a writer may be spinning on it)
  if (rc != TSD_OK) return rc;
  if (e_rc != hipSuccess) return set_error(ctx, TSD_E_HIP, "hipEventRecord(ev_rc_done)", e_rc);
  s->rc_pending = false;
  IcpArgs ia;
  const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  fill_icp_args(ia, ident, params);
  ia.beams = s->beams; ia.ccw = s->ccw ? 1 : 0;
  const unsigned long long seq = ++s->seq;
  ScanPostArgs sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.st = s->d_state; sp.rays = s->d_rays; sp.out = s->d_result; sp.seq = seq; sp.beams = s->beams;
  sp.gmin_x = ctx->grid.min_x; sp.gmax_x = ctx->grid.max_x; sp.gmin_y = ctx->grid.min_y; sp.gmax_y = ctx->grid.max_y;
  sp.gates = GateArgs{gates->reg_trs_max, gates->reg_sin_rot_max, gates->trs_min, gates->rot_min};
  rc = launch_icp(ctx, ia, s->d_state->icpP, s->d_rays_local, d_ranges, d_mask, &sp);
  if (rc != TSD_OK) return rc;
  TSD_HIP_CHECK(ctx, hipEventRecord(s->ev_icp_done, s->stream));
  s->conc_gates = *gates; s->conc_ranges = d_ranges; s->conc_mask_push = d_mask_push;
  s->inflight = true;
  lap.lap(3);
  return TSD_OK;
}

int tsd_scan_wait(tsd_sensor* s)
{
  if (!s || !s->inflight) return TSD_E_ARG;
  volatile unsigned long long* vseq = &s->h_result->seq;
  unsigned long long spins = 0;
  while (__atomic_load_n(vseq, __ATOMIC_ACQUIRE) != s->seq) {
    if (++spins > 4000000ull) {            // something is wrong: a real wait on the sensor's stream
      if (hipStreamSynchronize(s->stream) != hipSuccess || __atomic_load_n(vseq, __ATOMIC_ACQUIRE) != s->seq) return TSD_E_HIP;
      break;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  return TSD_OK;
}

int tsd_scan_finish(tsd_sensor* s, tsd_scan_result* result)
{
  if (!s || !s->ctx || !result || !s->inflight) return TSD_E_ARG;
  tsd_ctx* ctx = s->ctx;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  ConcLap lap;
  int rc = tsd_scan_wait(s);
  if (rc != TSD_OK) return set_error(ctx, TSD_E_HIP, "tsd_scan_finish: result record never arrived", hipSuccess);
  s->inflight = false;
  lap.lap(4);
  {
    // ORDERED SECTION: the push on the grid stream -- enqueued only now, when the registration has finished, so it
    // never sits on the grid stream waiting for it while other robots' pushes and ray casts queue up behind (events
    // order by ENQUEUE time: a push enqueued early would pull every later ray cast of every robot behind its own
    // registration and serialise the robots; measured: 3.7 k scans/s for any N).  Behind the ray casts ticketed since
    // the last grid write.
    std::lock_guard<std::mutex> lk_order(ctx->order_mutex);
    lap.lap(5);
    TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, s->ev_icp_done, 0));
    if (int rcw = wait_for_readers(ctx)) return rcw;
    PushArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    pa.beams = s->beams;
    pa.max_range = s->max_range;
    LaunchTarget tg;
    tg.rmq = s->d_rmq2[s->rmq_slot];
    TargetScope scope(ctx, &tg);
    // the registration moves the sensor by at most the gate (a larger step is rejected: pose unchanged)
    rc = launch_push(ctx, pa, s->pos[0], s->pos[1], s->conc_gates.reg_trs_max, &s->d_state->push, s->conc_ranges, s->conc_mask_push);
    if (rc != TSD_OK) return rc;
    ctx->epoch++;
    lap.lap(6);
  }
  copy_icp_result(&s->h_result->icp, &result->icp);
  for (int i = 0; i < 9; i++) result->pose[i] = s->h_result->pose[i];
  s->pos[0] = result->pose[2]; s->pos[1] = result->pose[5];
  result->reg_error = s->h_result->reg_error; result->pushed = s->h_result->pushed;
  result->no_model = s->h_result->no_model; result->reserved = 0;
  lap.lap(7); g_conc_timing.n++;
  return TSD_OK;
}

// ------------------------------------------------------------------ batched multi-robot scans (one shared grid)
// The split scan above gives every robot its own stream; HIP multiplexes streams onto a few in-order hardware queues, so with
// more than two or three robots a 0.17 ms registration blocks whatever shares its queue (measured: the time from begin to the
// result record grows from 0.19 ms at two robots to 0.8 ms at eight while the host calls stay at 0.1 ms per scan).  A batch
// does the robots that have a scan pending in ONE launch of each kernel on the batch's own stream -- tables (workgroup =
// scan), ray casts (block row = sensor), registrations (workgroup = robot, one compute unit each) -- and their pushes one
// after the other on the grid's stream.  Two or three batch slots used in turn keep the device busy with three or four streams
// in total: while one batch registers, the other one's pushes run.  All ray casts of a batch see the same grid state, the
// pushes follow in the order of the batch: one of the interleavings the reference's N ThreadLocalize + one ThreadMapping
// threads can produce.  tsd_batch_push may be called before the registrations have finished (the pushes are gated on the
// device like tsd_scan's); a ray cast enqueued later waits for it, one enqueued earlier does not.
static inline size_t align64(size_t v) { return (v + 63u) & ~(size_t)63u; }
// The two hand-offs of a batch (ray casts -> registrations, a robot's registration -> its push) are waits ON THE DEVICE (a flag /
// a gate kernel, see below) when -- and only when -- a start-up probe on the very streams involved has shown that a kernel on one can
// wait for a kernel launched after it on the other (probe_cross_stream_wait, both directions).  That is not a given: HIP maps streams
// onto a few in-order hardware queues (GPU_MAX_HW_QUEUES), so two streams may share one; rocprofv3's counter collection, blocking
// launches or a debugger serialise dispatches altogether.  Otherwise: stream events.  Whatever the mode, a device-side wait is
// bounded and a wait that runs out is an ERROR the caller sees (k_icp_batch / k_wait_seq), never a registration on stale data.
//   TSD_BATCH_EVENT_WAIT=1          stream events, no probe (A/B measurements)
//   TSD_BATCH_FORCE_DEVICE_WAIT=1   device waits whatever the probe says (tests of the failure path)
//   TSD_BATCH_POLL_BOUND=<polls>    bound of the device-side waits, ~1 us per poll (default 2^21)
static int batch_choose_wait_mode(tsd_batch* b)
{
  tsd_ctx* ctx = b->ctx;
  b->dev_wait = false;
  if (const char* e = getenv("TSD_BATCH_POLL_BOUND")) { const long v = std::atol(e); if (v >= 16 && v <= (1l << 30)) b->poll_bound = (unsigned int)v; }
  if (getenv("TSD_BATCH_EVENT_WAIT")) return TSD_OK;
  bool ab = false, ba = false;
  int rc = probe_cross_stream_wait(ctx, b->stream, ctx->stream, b->d_rc_flag, &ab);     // a registration waiting for the ray casts' flag
  if (rc == TSD_OK) rc = probe_cross_stream_wait(ctx, ctx->stream, b->stream, b->d_rc_flag, &ba);   // a push gate waiting for the registration
  if (rc != TSD_OK) return rc;
  TSD_HIP_CHECK(ctx, hipMemsetAsync(b->d_rc_flag, 0, 2 * sizeof(unsigned int), ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  b->dev_wait = ab && ba;
  if (!b->dev_wait && getenv("TSD_BATCH_FORCE_DEVICE_WAIT")) b->dev_wait = true;
  if (getenv("TSD_BATCH_VERBOSE"))
    fprintf(stderr, "tsd_batch_create: cross-stream probe %d/%d -> %s hand-offs\n", (int)ab, (int)ba, b->dev_wait ? "device-side" : "event");
  return TSD_OK;
}

// a push gate of this slot gave up (its registration never reported done): an error for whoever calls next, and events from now on
static int batch_gate_error(tsd_batch* b)
{
  if (!b->h_gate_err || __atomic_load_n(b->h_gate_err, __ATOMIC_ACQUIRE) == 0u) return TSD_OK;
  __atomic_store_n(b->h_gate_err, 0u, __ATOMIC_RELEASE);
  b->dev_wait = false;
  return set_error(b->ctx, TSD_E_HIP, "batched path: a push gate timed out waiting for its registration; that push was skipped "
                                       "(the slot uses stream events from now on)", hipSuccess);
}

// leave a batch that cannot be completed: nothing of it stays in flight, the sensors are free again
static void batch_abandon(tsd_batch* b, bool registration_launched)
{
  tsd_ctx* ctx = b->ctx;
  if (registration_launched && b->dev_wait) {
    // the registration kernel is (or will be) polling: tell it that this batch is off, on a stream it does not wait behind
    (void)launch_set_flag(ctx, ctx->stream, b->d_rc_flag + 1, b->rc_batches);
  }
  if (b->stream) hipStreamSynchronize(b->stream);
  hipStreamSynchronize(ctx->stream);
  for (tsd_sensor* s : b->sensors) if (s) s->inflight = false;
  b->n = 0; b->push_enqueued = false;
}

tsd_batch* tsd_batch_create(tsd_ctx* ctx, int max_scans)
{
  if (!ctx || max_scans < 1 || max_scans > TSD_BATCH_MAX_SCANS) return nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  tsd_batch* b = new (std::nothrow) tsd_batch();
  if (!b) return nullptr;
  b->ctx = ctx; b->max_scans = max_scans;
  b->head_bytes = align64((size_t)max_scans * sizeof(IcpBatchEntry)) + align64((size_t)max_scans * sizeof(RaycastBatchEntry)) +
                  align64((size_t)max_scans * sizeof(TablesBatchEntry));
  b->scan_bytes = align64((size_t)TSD_MAX_BEAMS * 10);
  const size_t bytes = b->head_bytes + (size_t)max_scans * b->scan_bytes;
  bool ok = true;
  auto A = [&](hipError_t e) { if (e != hipSuccess) ok = false; };
  A(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  for (hipEvent_t* e : {&b->ev_rc_done, &b->ev_icp_done, &b->ev_copy_done}) A(hipEventCreateWithFlags(e, hipEventDisableTiming));
  A(hipHostMalloc(&b->h_stage, bytes, hipHostMallocDefault));
  A(hipMalloc(&b->d_stage2[0], bytes)); A(hipMalloc(&b->d_stage2[1], bytes));
  A(hipMalloc(&b->d_rc_flag, 2 * sizeof(unsigned int)));
  if (ok) { A(hipMemsetAsync(b->d_rc_flag, 0, 2 * sizeof(unsigned int), ctx->stream)); A(hipStreamSynchronize(ctx->stream)); }
  A(hipHostMalloc(&b->h_gate_err, sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent));
  if (ok) { *b->h_gate_err = 0u; A(hipHostGetDevicePointer((void**)&b->d_gate_err, b->h_gate_err, 0)); }
  if (!ok) { set_error(ctx, TSD_E_HIP, "tsd_batch_create", hipGetLastError()); tsd_batch_destroy(b); return nullptr; }
  std::lock_guard<std::mutex> lk(ctx->order_mutex);
  if (batch_choose_wait_mode(b) != TSD_OK) { b->ctx = nullptr; tsd_batch_destroy(b); return nullptr; }   // (detached: destroy takes no lock)
  ctx->batches.push_back(b);
  return b;
}

void tsd_batch_destroy(tsd_batch* b)
{
  if (!b) return;
  if (b->ctx) {
    hipSetDevice(b->ctx->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    hipStreamSynchronize(b->ctx->stream);
    std::lock_guard<std::mutex> lk(b->ctx->order_mutex);
    auto& v = b->ctx->batches;
    v.erase(std::remove(v.begin(), v.end(), b), v.end());
  }
  for (tsd_sensor* s : b->sensors) if (s) s->inflight = false;
  for (hipEvent_t e : {b->ev_rc_done, b->ev_icp_done, b->ev_copy_done}) if (e) hipEventDestroy(e);
  if (b->stream) hipStreamDestroy(b->stream);
  if (b->h_stage) hipHostFree(b->h_stage);
  hipFree(b->d_stage2[0]); hipFree(b->d_stage2[1]); hipFree(b->d_rc_flag);
  if (b->h_gate_err) hipHostFree(b->h_gate_err);
  delete b;
}

int tsd_batch_capacity(const tsd_batch* b) { return b ? b->max_scans : 0; }
int tsd_batch_inflight(const tsd_batch* b) { return b ? b->n : 0; }

int tsd_batch_begin(tsd_batch* b, int n, tsd_sensor* const* sensors, const double* const* ranges, const uint8_t* const* mask,
                    const uint8_t* const* mask_push, const tsd_icp_params* params, const tsd_gate_params* gates)
{
  if (!b || !b->ctx || n < 1 || !sensors || !ranges || !mask || !params || !gates) return TSD_E_ARG;
  tsd_ctx* ctx = b->ctx;
  if (n > b->max_scans) return set_error(ctx, TSD_E_CAPACITY, "tsd_batch_begin: more scans than the batch was created for", hipSuccess);
  if (b->n) return set_error(ctx, TSD_E_ARG, "tsd_batch_begin: the previous batch of this slot was not collected (tsd_batch_results)", hipSuccess);
  if (int rcg = batch_gate_error(b)) return rcg;
  // everything that can be refused is refused HERE, before any state of the slot or of a sensor changes and before any launch
  for (int i = 0; i < n; i++) {
    if (params[i].estimator != params[0].estimator) return set_error(ctx, TSD_E_ARG, "tsd_batch_begin: one estimator per batch", hipSuccess);
    if (params[i].estimator != TSD_ESTIMATOR_CLOSED_FORM && params[i].estimator != TSD_ESTIMATOR_POINT_TO_LINE)
      return set_error(ctx, TSD_E_ARG, "tsd_icp_params.estimator", hipSuccess);
  }
  for (int i = 0; i < n; i++) {
    tsd_sensor* s = sensors[i];
    if (!s || s->ctx != ctx || !ranges[i] || !mask[i]) return set_error(ctx, TSD_E_ARG, "tsd_batch_begin: sensor / scan", hipSuccess);
    if (!s->posed) return set_error(ctx, TSD_E_ARG, "tsd_batch_begin before tsd_sensor_set_pose", hipSuccess);
    if (s->inflight) return set_error(ctx, TSD_E_ARG, "tsd_batch_begin: a sensor has a scan in flight already", hipSuccess);
    for (int j = 0; j < i; j++) if (sensors[j] == s) return set_error(ctx, TSD_E_ARG, "tsd_batch_begin: a sensor appears twice", hipSuccess);
  }
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  for (int i = 0; i < n; i++) if (int rc = sensor_conc_init(sensors[i], false)) return rc;

  // staging: [registration entries | ray-cast entries | tables entries | scan 0 | scan 1 ...], one copy for all of it, into the
  // device buffer the previous batch of this slot does not use (its pushes may still be reading their scans)
  b->stage_slot ^= 1;
  char* const d_base = b->d_stage2[b->stage_slot];
  char* const h_base = b->h_stage;
  IcpBatchEntry* h_icp = reinterpret_cast<IcpBatchEntry*>(h_base);
  const size_t off_rc = align64((size_t)b->max_scans * sizeof(IcpBatchEntry));
  const size_t off_tb = off_rc + align64((size_t)b->max_scans * sizeof(RaycastBatchEntry));
  RaycastBatchEntry* h_rc = reinterpret_cast<RaycastBatchEntry*>(h_base + off_rc);
  TablesBatchEntry* h_tb = reinterpret_cast<TablesBatchEntry*>(h_base + off_tb);
  b->sensors.assign(sensors, sensors + n);
  b->seqs.resize((size_t)n); b->gates.assign(gates, gates + n); b->scan_off.resize((size_t)n);
  size_t off = b->head_bytes;
  int max_beams = 0;
  const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < n; i++) {
    tsd_sensor* s = sensors[i];
    const size_t nb = (size_t)s->beams;
    if (s->beams > max_beams) max_beams = s->beams;
    char* h = h_base + off;
    std::memcpy(h, ranges[i], nb * 8);
    std::memcpy(h + nb * 8, mask[i], nb);
    std::memcpy(h + nb * 9, (mask_push && mask_push[i]) ? mask_push[i] : mask[i], nb);
    const double* d_ranges = reinterpret_cast<const double*>(d_base + off);
    const uint8_t* d_mask = reinterpret_cast<const uint8_t*>(d_base + off + nb * 8);
    const uint8_t* d_mask_push = reinterpret_cast<const uint8_t*>(d_base + off + nb * 9);
    b->scan_off[(size_t)i] = off;
    off += align64(nb * 10);
    s->rmq_slot ^= 1;                       // (the previous push of this sensor may still read its tables)
    h_tb[i] = TablesBatchEntry{d_ranges, d_mask_push, s->d_rmq2[s->rmq_slot], s->phi_min, s->ang_res, s->beams, 0};
    h_rc[i] = RaycastBatchEntry{&s->d_state->rc, s->d_rays, s->d_coords, s->d_normals, s->d_mask_m};
    IcpBatchEntry& e = h_icp[i];
    std::memset(&e, 0, sizeof(e));
    fill_icp_args(e.a, ident, &params[i]);
    e.a.beams = s->beams; e.a.ccw = s->ccw ? 1 : 0;
    // registration_mode 3 (tsd_scan_preregister armed this sensor): the registration starts from the pre-registration's result, which
    // k_pdf_argmax leaves in the sensor's own buffer (the kernels go out below, behind the batch's ray casts)
    if (s->pre_armed) e.a.Tinit_dev = reinterpret_cast<const double*>(s->d_pre + s->pre.off_res);
    e.P_dev = s->d_state->icpP; e.coords = s->d_coords; e.mask_m = s->d_mask_m; e.rays_local = s->d_rays_local;
    e.ranges = d_ranges; e.mask = d_mask; e.out = s->d_icp_res; e.trace = nullptr /* no reader in the fused path */; e.normals = s->d_normals;
    const unsigned long long seq = ++s->seq;
    b->seqs[(size_t)i] = seq;
    if (!s->pre_armed) s->pre_ran = false;                 // (tsd_scan_preregistration_result: this scan has none)
    e.post.st = s->d_state; e.post.rays = s->d_rays; e.post.out = s->d_result; e.post.seq = seq; e.post.beams = s->beams;
    e.post.gmin_x = ctx->grid.min_x; e.post.gmax_x = ctx->grid.max_x; e.post.gmin_y = ctx->grid.min_y; e.post.gmax_y = ctx->grid.max_y;
    e.post.gates = GateArgs{gates[i].reg_trs_max, gates[i].reg_sin_rot_max, gates[i].trs_min, gates[i].rot_min};
  }
  for (int i = 0; i < n; i++) h_icp[i].seed = icp_batch_seed_args(ctx, sensors[i]->d_icp_seed, sensors[i]->beams, max_beams);
  // the registrations go out AHEAD of the ray casts and wait for the slot's flag on the device (where the probe allowed it).
  // A batch that carries a pre-registration (registration_mode 3) orders its registrations behind the grid stream's work by an
  // event instead: the pre-registration kernels sit between the ray casts and the registrations, on the grid's stream -- the scoring
  // reads the grid, like the ray casts, and takes its place between the pushes the same way.
  bool any_pre = false;
  for (int i = 0; i < n; i++) any_pre |= sensors[i]->pre_armed;
  const bool dev_wait = b->dev_wait && !any_pre;
  if (dev_wait) {
    b->rc_batches++;
    for (int i = 0; i < n; i++) { h_icp[i].rc_flag = b->d_rc_flag; h_icp[i].rc_target = b->rc_batches; h_icp[i].poll_bound = b->poll_bound; }
  }
  bool icp_launched = false;
  // (from here on a failure leaves through batch_abandon: nothing of the batch stays in flight, no kernel keeps polling)
  auto FAIL = [&](int code) { batch_abandon(b, icp_launched); return code; };
  if (hipMemcpyAsync(d_base, h_base, off, hipMemcpyHostToDevice, b->stream) != hipSuccess)
    return FAIL(set_error(ctx, TSD_E_HIP, "tsd_batch_begin: copy", hipGetLastError()));
  b->d_stage_cur = d_base;
  int rc = launch_push_tables_batch(ctx, b->stream, reinterpret_cast<const TablesBatchEntry*>(d_base + off_tb), n, max_beams);
  if (rc != TSD_OK) return FAIL(rc);
  {
    // ORDERED SECTION: the ray casts read the grid, so they go behind every grid write enqueued so far and take their place in
    // the order for the writes that follow
    // The batched ray cast runs on the GRID's stream, between the pushes: the stream's own order keeps it behind every grid
    // write enqueued so far and ahead of the writes that follow, with no cross-queue hand-off (13-23 us each as measured,
    // profiles/r2_multi_robot_timeline.txt) on the chain ray casts -> pushes -> ray casts that bounds a round.  The entries
    // it reads come with the batch's copy; the registration waits for it by event.
    std::lock_guard<std::mutex> lk(ctx->order_mutex);
    if (dev_wait) {
      rc = launch_icp_batch(ctx, b->stream, h_icp, reinterpret_cast<const IcpBatchEntry*>(d_base), n);
      if (rc != TSD_OK) return FAIL(rc);
      icp_launched = true;
    }
    if (n <= RC_BATCH_BYVAL) {
      // (the entries as kernel arguments: nothing of the batch's copy is needed, one wait less on the grid's stream)
      rc = launch_raycast_batch_byval(ctx, ctx->stream, h_rc, n, max_beams);
    } else {
      if (hipEventRecord(b->ev_copy_done, b->stream) != hipSuccess || hipStreamWaitEvent(ctx->stream, b->ev_copy_done, 0) != hipSuccess)
        return FAIL(set_error(ctx, TSD_E_HIP, "tsd_batch_begin: copy event", hipGetLastError()));
      rc = launch_raycast_batch(ctx, ctx->stream, reinterpret_cast<const RaycastBatchEntry*>(d_base + off_rc), n, max_beams);
    }
    if (rc != TSD_OK) return FAIL(rc);
    if (any_pre) {
      // TSD_PDFMatching::match of every armed robot (ThreadLocalize.cpp:557-567, each robot's own thread in the reference) on the
      // model its ray cast just produced; all of them score against the grid as it is before any push of this batch
      for (int i = 0; i < n; i++) {
        tsd_sensor* s = sensors[i];
        if (!s->pre_armed) continue;
        s->pre_armed = false;
        const double* tinit = nullptr;
        rc = launch_preregistration(ctx, s, ctx->stream, s->d_coords, s->d_mask_m, s->d_state->icpP, &tinit, nullptr);
        if (rc != TSD_OK) return FAIL(rc);
        s->pre_ran = true;
      }
    }
    if (dev_wait) {
      rc = launch_set_flag(ctx, ctx->stream, b->d_rc_flag, b->rc_batches);
      if (rc != TSD_OK) return FAIL(rc);
    } else {
      if (hipEventRecord(b->ev_rc_done, ctx->stream) != hipSuccess || hipStreamWaitEvent(b->stream, b->ev_rc_done, 0) != hipSuccess)
        return FAIL(set_error(ctx, TSD_E_HIP, "tsd_batch_begin: ray-cast event", hipGetLastError()));
    }
  }
  if (!dev_wait) {
    rc = launch_icp_batch(ctx, b->stream, h_icp, reinterpret_cast<const IcpBatchEntry*>(d_base), n);
    if (rc != TSD_OK) return FAIL(rc);
    icp_launched = true;
  }
  if (hipEventRecord(b->ev_icp_done, b->stream) != hipSuccess) return FAIL(set_error(ctx, TSD_E_HIP, "tsd_batch_begin: event", hipGetLastError()));
  for (int i = 0; i < n; i++) { sensors[i]->inflight = true; sensors[i]->rc_pending = false; }
  b->n = n; b->push_enqueued = false;
  return TSD_OK;
}

int tsd_batch_push(tsd_batch* b)
{
  if (!b || !b->ctx) return TSD_E_ARG;
  if (!b->n || b->push_enqueued) return TSD_OK;
  tsd_ctx* ctx = b->ctx;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  std::lock_guard<std::mutex> lk_order(ctx->order_mutex);
  const bool gate = b->dev_wait;                          // (else: the stream event for the whole batch's kernel)
  if (!gate) TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, b->ev_icp_done, 0));
  if (int rcw = wait_for_readers(ctx)) return rcw;
  for (int i = 0; i < b->n; i++) {
    tsd_sensor* s = b->sensors[(size_t)i];
    // robot i's push starts when robot i's registration is done (its epilogue has left the push arguments and published the
    // scan's sequence number), not when the slowest registration of the batch is
    if (gate) { if (int rcg = launch_wait_seq(ctx, &s->d_state->done_seq, b->seqs[(size_t)i], &s->d_state->push, b->d_gate_err, b->poll_bound)) return rcg; }
    const size_t nb = (size_t)s->beams;
    PushArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    pa.beams = s->beams;
    pa.max_range = s->max_range;
    LaunchTarget tg;
    tg.rmq = s->d_rmq2[s->rmq_slot];
    TargetScope scope(ctx, &tg);
    const char* d_scan = b->d_stage_cur + b->scan_off[(size_t)i];
    // the registration moves the sensor by at most the gate (a larger step is rejected: pose unchanged); s->pos is the
    // position after the previous scan, which the host has seen
    int rc = launch_push(ctx, pa, s->pos[0], s->pos[1], b->gates[(size_t)i].reg_trs_max, &s->d_state->push,
                         reinterpret_cast<const double*>(d_scan), reinterpret_cast<const uint8_t*>(d_scan + nb * 9));
    if (rc != TSD_OK) return rc;
  }
  ctx->epoch++;
  b->push_enqueued = true;
  return TSD_OK;
}

int tsd_batch_poll(tsd_batch* b)
{
  if (!b) return TSD_E_ARG;
  for (int i = 0; i < b->n; i++)
    if (__atomic_load_n(&b->sensors[(size_t)i]->h_result->seq, __ATOMIC_ACQUIRE) != b->seqs[(size_t)i]) return 0;
  return 1;
}

int tsd_batch_results(tsd_batch* b, tsd_scan_result* results)
{
  if (!b || !b->ctx || !results) return TSD_E_ARG;
  tsd_ctx* ctx = b->ctx;
  if (!b->n) return set_error(ctx, TSD_E_ARG, "tsd_batch_results without tsd_batch_begin", hipSuccess);
  unsigned long long spins = 0;
  while (tsd_batch_poll(b) != 1) {
    if (++spins > 4000000ull) {              // something is wrong: a real wait on the batch's stream
      if (hipStreamSynchronize(b->stream) != hipSuccess || tsd_batch_poll(b) != 1) {
        batch_abandon(b, true);              // (the slot and its sensors are usable again; this batch's scans are lost)
        return set_error(ctx, TSD_E_HIP, "tsd_batch_results: result records never arrived", hipSuccess);
      }
      break;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  int rc = tsd_batch_push(b);               // (no-op when the caller enqueued the pushes ahead of the results)
  if (rc != TSD_OK) { batch_abandon(b, true); return rc; }
  int failed = 0;
  for (int i = 0; i < b->n; i++) {
    tsd_sensor* s = b->sensors[(size_t)i];
    tsd_scan_result* r = &results[i];
    copy_icp_result(&s->h_result->icp, &r->icp);
    for (int k = 0; k < 9; k++) r->pose[k] = s->h_result->pose[k];
    r->reg_error = s->h_result->reg_error; r->pushed = s->h_result->pushed;
    r->no_model = s->h_result->no_model; r->reserved = s->h_result->reserved;
    if (r->reserved != 0) failed = r->reserved;       // this robot's registration never ran (k_icp_batch): flagged, pose untouched
    else { s->pos[0] = r->pose[2]; s->pos[1] = r->pose[5]; }
    s->inflight = false;
  }
  b->n = 0;
  if (failed) {
    // a device-side wait gave up, so kernels of the two streams do not run side by side here (any more): events from now on
    b->dev_wait = false;
    return set_error(ctx, TSD_E_HIP, failed == BATCH_FAIL_TIMEOUT
                       ? "batched path: a registration's device-side wait for its ray casts timed out; its scan was NOT registered "
                         "(tsd_scan_result.reserved = 1 marks the robots concerned; the slot uses stream events from now on)"
                       : "batched path: the batch was abandoned before its registrations ran", hipSuccess);
  }
  return batch_gate_error(b);
}

}  // extern "C"
