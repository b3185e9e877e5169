// capi.hip -- the extern "C" entry points of include/tsd_hip.h: context life cycle, staging of the
// scan through a ring of pinned buffers, and the launch order of the kernels on the ctx stream.
// No CPU fall-back exists: every compute entry point needs a gfx950 device and fails loudly without.
#include "capi_internal.hpp"

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

const char* const kKernelNames[] = {"push_classify", "push_update", "push_halo", "raycast", "icp", "occupancy", "tsdpdf"};

bool kernel_is_timed(const tsd_ctx* ctx, const char* name)
{
  if (!ctx->profile) return false;
  for (unsigned i = 0; i < sizeof(kKernelNames) / sizeof(kKernelNames[0]); i++)
    if (std::strcmp(kKernelNames[i], name) == 0) return (ctx->profile_mask >> i) & 1u;
  return false;
}

hipEvent_t pool_get(tsd_ctx* ctx)
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
char* stage_acquire(tsd_ctx* ctx, int* slot_out)
{
  const int s = ctx->slot;
  ctx->slot = (ctx->slot + 1) % tsd_ctx::kSlots;
  hipEventSynchronize(ctx->stage_ev[s]);
  *slot_out = s;
  return ctx->h_stage[s];
}

// DistanceFilter ctor as called from ThreadLocalize.cpp:212 (int -> unsigned conversion of
// icp_iterations - 10; DistanceFilter.cpp:11-20)
double distance_filter_multiplier(double maxdist, double mindist, int icp_iterations)
{
  unsigned int iterations = (unsigned int)(icp_iterations - 10);
  double it = (double)(iterations - 1);
  if (iterations < 1) it = 1.0;
  return std::pow((mindist / maxdist), 1.0 / it);
}

void fill_icp_args(IcpArgs& a, const double pose33[9], const tsd_icp_params* p)
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

void fill_raycast_args(const tsd_ctx* ctx, RaycastArgs& a, const double pose33[9], int beams,
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

void copy_icp_result(const IcpResultDev* h, tsd_icp_result* r)
{
  for (int i = 0; i < 9; i++) r->T[i] = h->T[i];
  r->rms = h->rms; r->pairs = h->pairs; r->iterations = h->iterations; r->state = h->state;
  r->n_model = h->n_model; r->n_scene = h->n_scene; r->reserved = h->reserved;
}

// Statistics of the pushes: every tile keeps a record of the last push (k_push_tiles) and running totals
// (k_push_halo); they are summed on the host when somebody asks -- tests and the end of a benchmark.
void fill_stats(tsd_ctx* ctx, const unsigned long long t[7], tsd_push_stats* out)
{
  out->cells_updated = (int64_t)t[0];
  out->cells_visited = 1024ll * (int64_t)t[2];     // every UPDATE tile back-projects its 32x32 cells
  out->tiles_total = ctx->grid.tiles;
  out->tiles_range_pass = (int32_t)t[1]; out->tiles_update = (int32_t)t[2]; out->tiles_new = (int32_t)t[3];
  out->tiles_new_from_empty = (int32_t)t[4]; out->tiles_emptied_init = (int32_t)t[5];
  out->tiles_emptied_uninit = (int32_t)t[6];
}

int read_last_push_stats(tsd_ctx* ctx, tsd_push_stats* out)
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

int read_total_stats(tsd_ctx* ctx, tsd_push_stats* out, int64_t* pushes, bool reset)
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
bool host_saw_event(hipEvent_t ev, int us)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (hipEventQuery(ev) == hipSuccess) return true;
    (void)hipGetLastError();                 // hipErrorNotReady is sticky as "last error": do not leave it for other libraries
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(us)) return false;
  }
}

ConcTiming g_conc_timing;
ConcTiming g_scan_timing;      // the same for tsd_scan (one robot): where the host time of a scan goes
ConcTiming g_stage_timing;     // ... and of the staging of a scan (acquire, host copy, hipMemcpyAsync, records, tables)
unsigned long long g_scan_lap_max[8] = {0, 0, 0, 0, 0, 0, 0, 0};
unsigned long long g_scan_lap_max_at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
unsigned long long g_scan_last_return = 0;

// every grid WRITE enqueued on the context's stream goes behind the ray casts the concurrent multi-robot path has in
// flight on the sensors' own streams (no-op without such sensors)
int wait_for_readers(tsd_ctx* ctx)
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

int tsd_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes)
{
  if (!free_bytes || !total_bytes) return TSD_E_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return TSD_E_NODEVICE; }
  int prev = 0;
  (void)hipGetDevice(&prev);
  size_t f = 0, t = 0;
  const bool ok = hipSetDevice(device) == hipSuccess && hipMemGetInfo(&f, &t) == hipSuccess;
  (void)hipSetDevice(prev);
  if (!ok) { (void)hipGetLastError(); return TSD_E_HIP; }
  *free_bytes = (uint64_t)f; *total_bytes = (uint64_t)t;
  return TSD_OK;
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
  if (const char* e = std::getenv("TSD_PUSH_MULTI")) ctx->push_multi = std::atoi(e) != 0;
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
  hipFree(ctx->d_rmq2[0]); hipFree(ctx->d_rmq2[1]); hipFree(ctx->d_tile_rec); hipFree(ctx->d_dirty); hipFree(ctx->d_tile_totals); hipFree(ctx->d_pushes); hipFree(ctx->d_list); hipFree(ctx->d_list_h); hipFree(ctx->d_mp_mask); hipFree(ctx->d_mp_rec); hipFree(ctx->d_mp_list); hipFree(ctx->d_list_aux); hipFree(ctx->d_push_args); hipFree(ctx->d_list_cnt);
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

}  // extern "C"

