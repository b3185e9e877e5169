// capi_scan.hip -- the fused scan path of include/tsd_hip.h (tsd_sensor_*, tsd_scan_*: the whole event-loop body of ThreadLocalize in
// stream order, sensor state on the device), its split form for several robots on one grid (tsd_scan_begin / _wait / _finish) and the
// batched form (tsd_batch_*: one launch of each kernel for the robots of a batch).
#include "capi_internal.hpp"
#include "push_device.hpp"
#include "tsdpdf_device.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

using namespace tsd;

extern "C" {
// ---------------------------------------------------------------------------------- fused scan path
// Can this process store into fine-grained device memory of `device` through the PCIe BAR, and does a kernel launched right behind the
// stores read them?  Asked once per device, without provoking a fault: hipDeviceAttributeIsLargeBar is the runtime's own answer to "may
// the host access this agent's local memory pool" (what HSA's pool-access query says for the CPU agent), and the pointer attributes of
// the allocation must name it device memory of this device.  What remains is whether the ACCESS PATTERN of the scan path works here --
// a kernel reads the buffer, the host rewrites it with plain memcpy + sfence, the next kernel (launched at once, nothing drained in
// between) must read the new contents: a stale L2 line or a write-combined store that lands behind the launch would show as an old
// word.  That pattern is run for several rounds on the context's own stream (the stream the registrations run on) and on the side
// stream (the tables'), over a whole buffer of a scan's size.  TSD_SCAN_PINNED=1 says no without asking.
static __global__ void k_bar_probe(const unsigned long long* __restrict__ buf, unsigned long long* __restrict__ out, int n)
{
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x) out[i] = buf[i];
}
// TSD_SCAN_BAR_VERIFY=1 (debug): every scan is ALSO written to the pinned buffer, and ahead of its registration this kernel compares
// the two copies as the device sees them; a difference is reported by tsd_scan_collect as TSD_E_HIP instead of becoming a wrong pose.
static __global__ void k_bar_verify(const unsigned char* __restrict__ bar, const unsigned char* __restrict__ pinned, int n, unsigned int* __restrict__ mismatches)
{
  unsigned int bad = 0;
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x) bad += bar[i] != pinned[i];
  if (bad) atomicAdd_system(mismatches, bad);
}
// (TSD_SCAN_BAR_VERIFY=2, tests only: the pinned copy of every scan is spoiled in one byte, so the cross-check must fire)
static int bar_verify_mode()
{
  static const int mode = [] { const char* e = std::getenv("TSD_SCAN_BAR_VERIFY"); return e && (*e == '1' || *e == '2') ? *e - '0' : 0; }();
  return mode;
}
static bool bar_verify_requested() { return bar_verify_mode() != 0; }
static bool host_writes_device_memory(tsd_ctx* ctx)
{
  const int device = ctx->device;
  static std::mutex mu;
  static int known[64];                          // 0: not asked, 1: yes, 2: no
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0 || device >= 64) return false;
  if (known[device]) return known[device] == 1;
  known[device] = 2;
#if defined(__x86_64__)
  if (const char* e = std::getenv("TSD_SCAN_PINNED")) if (*e == '1') return false;
  int large_bar = 0;
  if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess || !large_bar) { (void)hipGetLastError(); return false; }
  constexpr int kWords = (TSD_MAX_BEAMS * 10 + 64 + 7) / 8;      // a scan buffer's size
  constexpr int kRounds = 6;
  unsigned long long* p = nullptr;
  unsigned long long* d_out = nullptr;
  std::vector<unsigned long long> src(kWords), back(kWords);
  bool ok = hipExtMallocWithFlags((void**)&p, kWords * 8, hipDeviceMallocFinegrained) == hipSuccess && p;
  hipPointerAttribute_t at;
  std::memset(&at, 0, sizeof(at));
  ok = ok && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice && at.device == device;
  ok = ok && hipMalloc((void**)&d_out, kWords * 8) == hipSuccess;
  for (int r = 0; ok && r < kRounds; r++) {
    hipStream_t st = (r & 1) && ctx->stream2 ? ctx->stream2 : ctx->stream;
    for (int i = 0; i < kWords; i++) src[i] = 0x9E3779B97F4A7C15ull * (unsigned long long)(r * kWords + i + 1);
    std::memcpy(p, src.data(), (size_t)kWords * 8);               // (what scan_stage_host does)
    _mm_sfence();
    hipLaunchKernelGGL(k_bar_probe, dim3(4), dim3(256), 0, st, p, d_out, kWords);
    ok = hipGetLastError() == hipSuccess &&
         hipMemcpyAsync(back.data(), d_out, (size_t)kWords * 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess && std::memcmp(back.data(), src.data(), (size_t)kWords * 8) == 0;
  }
  (void)hipGetLastError();
  if (p) (void)hipFree(p);
  if (d_out) (void)hipFree(d_out);
  if (ok) known[device] = 1;
  return ok;
#else
  return false;
#endif
}

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
  if (host_writes_device_memory(ctx)) {
    s->scan_bar = true;
    for (int i = 0; i < 3; i++)
      if (hipExtMallocWithFlags((void**)&s->d_scan2[i], nb * 10 + 64, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); s->scan_bar = false; }
    if (!s->scan_bar) for (int i = 0; i < 3; i++) { if (s->d_scan2[i]) hipFree(s->d_scan2[i]); s->d_scan2[i] = nullptr; }
    if (s->scan_bar && bar_verify_requested()) {
      A(hipHostMalloc((void**)&s->h_bar_mismatch, 64, hipHostMallocMapped | hipHostMallocCoherent));
      if (ok) { *s->h_bar_mismatch = 0; A(hipHostGetDevicePointer((void**)&s->d_bar_mismatch, s->h_bar_mismatch, 0)); }
    }
  }
  if (!s->scan_bar) for (int i = 0; i < 3; i++) A(hipMalloc(&s->d_scan2[i], nb * 10 + 64));
  for (int i = 0; i < 3; i++) {
    A(hipHostMalloc(&s->h_scan3[i], nb * 10 + 64, hipHostMallocMapped));
    if (ok) A(hipHostGetDevicePointer((void**)&s->hd_scan3[i], s->h_scan3[i], 0));
    A(hipEventCreateWithFlags(&s->ev_scan_copy[i], hipEventDisableTiming));
  }
  // the scan result is written by the kernel straight into coherent pinned host memory
  s->h_result = new (std::nothrow) ScanResultDev();
  if (!s->h_result) ok = false; else std::memset(s->h_result, 0, sizeof(ScanResultDev));
  A(hipHostMalloc(&s->h_rwords, sizeof(unsigned long long) * SCAN_RESULT_WORDS, hipHostMallocMapped | hipHostMallocCoherent));
  if (ok) { std::memset(s->h_rwords, 0, sizeof(unsigned long long) * SCAN_RESULT_WORDS); A(hipHostGetDevicePointer((void**)&s->d_result, s->h_rwords, 0)); }
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
  if (s->d_pre_flag) hipFree(s->d_pre_flag);
  if (s->h_pre) hipHostFree(s->h_pre);
  hipFree(s->d_rmq2[0]); hipFree(s->d_rmq2[1]); hipFree(s->d_rmq2[2]);
  for (int i = 0; i < 3; i++) { if (s->h_scan3[i]) hipHostFree(s->h_scan3[i]); if (s->ev_scan_copy[i]) hipEventDestroy(s->ev_scan_copy[i]); }
  if (s->h_stage2[0]) hipHostFree(s->h_stage2[0]);
  if (s->h_stage2[1]) hipHostFree(s->h_stage2[1]);
  hipFree(s->d_state); hipFree(s->d_rays); hipFree(s->d_rays_local); hipFree(s->d_scan2[0]); hipFree(s->d_scan2[1]); hipFree(s->d_scan2[2]);
  hipHostFree(s->h_rwords); delete s->h_result;
  if (s->h_bar_mismatch) hipHostFree(s->h_bar_mismatch);
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

// Has the result record of scan `seq` arrived?  It travels as tagged 8-byte words {low half of seq, four bytes of the record}
// (scan_device.hpp: scan_post_body); once every word carries the tag the record is decoded into s->h_result.  The last word first: the
// usual answer is "not yet" after one load.
static bool scan_result_arrived(tsd_sensor* s, unsigned long long seq)
{
  if (s->h_result->seq == seq) return true;                      // (decoded by an earlier call)
  const unsigned int tag = (unsigned int)seq;
  unsigned long long w[SCAN_RESULT_WORDS];
  for (int i = SCAN_RESULT_WORDS - 1; i >= 0; i--) {
    w[i] = __atomic_load_n(&s->h_rwords[i], __ATOMIC_RELAXED);
    if ((unsigned int)(w[i] >> 32) != tag) return false;
  }
  unsigned int* out = reinterpret_cast<unsigned int*>(s->h_result);
  for (int i = 0; i < SCAN_RESULT_WORDS; i++) out[i] = (unsigned int)w[i];
  s->h_result->seq = seq;                                          // (its own words carry the low half only)
  return true;
}

// A scan comes in two steps.  scan_stage_host: the caller's arrays into the sensor's scan buffer -- device memory that the host stores
// into through the PCIe BAR (tsd_sensor::scan_bar), or a pinned host buffer where that mapping is missing (the three scan / table
// buffers are used in turn).  scan_stage_device: the range-query tables of this scan's push on the side stream (and, pinned mode, the
// copy into device memory ahead of them) -- they only depend on the scan and run while the main stream is busy.  tsd_scan_stage does
// both at once (a scan announced ahead); tsd_scan_submit with a scan launches the registration between the two.
static int scan_stage_host(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push)
{
  tsd_ctx* ctx = s->ctx;
  const size_t nb = (size_t)s->beams;
  if (int rc = sensor_conc_init(s, false)) return rc;       // (the sensor's own table buffers)
  unsigned long long tl = g_stage_timing.on ? now_ns() : 0;
  auto LAP = [&](int i) { if (g_stage_timing.on) { const unsigned long long u = now_ns(); g_stage_timing.ns[i] += u - tl; tl = u; } };
  // three buffers in turn (see tsd_sensor::stage_slot): the push that read this one three scans ago is done -- and so is the device
  // copy out of the pinned buffer, which that push was ordered behind; its event is looked at all the same (one query)
  const int sslot = s->stage_slot;
  s->stage_slot = (s->stage_slot + 1) % 3;
  if (s->scan_copy_valid[sslot]) { TSD_HIP_CHECK(ctx, hipEventSynchronize(s->ev_scan_copy[sslot])); s->scan_copy_valid[sslot] = false; }
  // (the host writes the device buffer itself: with asynchronous mapping the push that last read it has to be SEEN done first)
  if (s->scan_bar && s->slot_push_valid[sslot]) { TSD_HIP_CHECK(ctx, hipEventSynchronize(s->ev_slot_push[sslot])); s->slot_push_valid[sslot] = false; }
  LAP(0);
  char* h = s->scan_bar ? s->d_scan2[sslot] : s->h_scan3[sslot];
  std::memcpy(h, ranges, nb * 8);
  std::memcpy(h + nb * 8, mask, nb);
  std::memcpy(h + nb * 9, mask_push ? mask_push : mask, nb);
#if defined(__x86_64__)
  if (s->scan_bar) _mm_sfence();              // (write-combined stores: on their way before any launch that reads them)
#endif
  if (s->d_bar_mismatch) {                    // (debug: the pinned copy the device compares the scan buffer with, launch_bar_verify)
    std::memcpy(s->h_scan3[sslot], ranges, nb * 8);
    std::memcpy(s->h_scan3[sslot] + nb * 8, mask, nb);
    std::memcpy(s->h_scan3[sslot] + nb * 9, mask_push ? mask_push : mask, nb);
    if (bar_verify_mode() == 2) s->h_scan3[sslot][nb * 4] ^= 0x40;
  }
  LAP(1);
  char* d_scan = s->d_scan2[sslot];
  s->st_ranges = reinterpret_cast<const double*>(d_scan);
  s->st_mask = reinterpret_cast<const uint8_t*>(d_scan + nb * 8);
  s->st_mask_push = reinterpret_cast<const uint8_t*>(d_scan + nb * 9);
  s->st_h_ranges = s->scan_bar ? s->st_ranges : reinterpret_cast<const double*>(s->hd_scan3[sslot]);
  s->st_h_mask = s->scan_bar ? s->st_mask : reinterpret_cast<const uint8_t*>(s->hd_scan3[sslot] + nb * 8);
  s->st_rmq = s->d_rmq2[sslot]; s->st_slot = sslot;
  s->st_device_done = false;
  s->staged = true;
  return TSD_OK;
}

static int scan_stage_device(tsd_sensor* s)
{
  tsd_ctx* ctx = s->ctx;
  const size_t nb = (size_t)s->beams;
  const int sslot = s->st_slot;
  unsigned long long tl = g_stage_timing.on ? now_ns() : 0;
  auto LAP = [&](int i) { if (g_stage_timing.on) { const unsigned long long u = now_ns(); g_stage_timing.ns[i] += u - tl; tl = u; } };
  // Asynchronous mapping: the push that last read this buffer (three scans back) ran on the push stream beside a registration, and
  // nothing the host has seen since is ordered behind it -- the copy and the tables below wait for that push's own event (done long
  // ago in practice: one query; the stream-side wait is the fall-back).  Strict order: see tsd_sensor::stage_slot.
  if (s->slot_push_valid[sslot]) {
    if (!host_saw_event(s->ev_slot_push[sslot], 0)) TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream2, s->ev_slot_push[sslot], 0));
    else s->slot_push_valid[sslot] = false;
  }
  if (!s->scan_bar) {
    TSD_HIP_CHECK(ctx, hipMemcpyAsync(s->d_scan2[sslot], s->h_scan3[sslot], nb * 10, hipMemcpyHostToDevice, ctx->stream2));
    LAP(2);
    TSD_HIP_CHECK(ctx, hipEventRecord(s->ev_scan_copy[sslot], ctx->stream2));
    s->scan_copy_valid[sslot] = true;
    TSD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_h2d, ctx->stream2));
    LAP(3);
  }
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
  s->st_device_done = true;
  return TSD_OK;
}

static int scan_stage_impl(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push)
{
  if (int rc = scan_stage_host(s, ranges, mask, mask_push)) return rc;
  return scan_stage_device(s);
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

int tsd_debug_sensor_scan_path(const tsd_sensor* s)
{
  if (!s) return TSD_E_ARG;
  return s->scan_bar ? (s->d_bar_mismatch ? 2 : 1) : 0;
}

int tsd_debug_set_push_multi(tsd_ctx* ctx, int on)
{
  if (!ctx) return TSD_E_ARG;
  ctx->push_multi = on ? 1 : 0;
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
    // ones on the side stream (pinned mode), and nothing else ever read the dropped data.
    if (s->staged) {
      s->stage_slot = s->st_slot;
      // (BAR mode: the host is about to store into that buffer itself -- the dropped scan's tables kernel on the side stream, which
      // reads it, has to be SEEN done first; pinned mode orders the new copy behind it on the side stream)
      if (s->scan_bar && s->st_device_done) TSD_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_tables));
    }
    s->staged = false;
    int rcs = scan_stage_host(s, ranges, mask, mask_push);
    if (rcs != TSD_OK) return rcs;
  }
  s->staged = false;
  // A scan that came with this call is in the scan buffer (device memory the host stored into, or the pinned buffer) and its tables are
  // not built yet: its registration reads it from there, and the tables (pinned mode: the device copy first) are enqueued BEHIND the
  // registration's launch -- the host's work on them no longer sits between the result of the
  // previous scan and this launch (the main stream ran dry for ~10 us per scan there; the push needs them 100+ us from now).
  const bool icp_from_host = !s->st_device_done;
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
  IcpPreLaunch prel;
  std::memset(&prel, 0, sizeof(prel));
  bool fold_argmax = false;
  if (s->pre_armed) {
    s->pre_armed = false;
    // (asynchronous mapping: the SCORING reads the grid -- the previous scan's push, still on the push stream, has to land first; the
    // normals and the list building ahead of it do not, and run beside that push.  The order is then ray cast, previous push,
    // pre-registration, registration: still one of the reference's interleavings)
    hipEvent_t before_score = nullptr;
    if (ctx->async_pending) { before_score = ctx->ev_async_push; ctx->async_pending = false; }
    const LaunchTarget* tgp = launch_target();
    // (the node's registration shape: the arg-max rides with the registration's launch -- TSD_PDF_ARGMAX_KERNEL=1: as a kernel, A/B)
    static const bool argmax_kernel = [] { const char* e = std::getenv("TSD_PDF_ARGMAX_KERNEL"); return e && *e == '1'; }();
    fold_argmax = !argmax_kernel && icp_pre_supported(ctx, ia);
    rc = launch_preregistration(ctx, s, launch_stream(ctx), tgp && tgp->coords ? tgp->coords : ctx->d_coords,
                                tgp && tgp->mask_m ? tgp->mask_m : ctx->d_mask_m, s->d_state->icpP, &ia.Tinit_dev, before_score, fold_argmax ? &prel : nullptr);
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
  if (s->d_bar_mismatch) {
    hipLaunchKernelGGL(k_bar_verify, dim3(4), dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned char*>(s->d_scan2[s->st_slot]),
                       reinterpret_cast<const unsigned char*>(s->hd_scan3[s->st_slot]), (int)((size_t)s->beams * 10), s->d_bar_mismatch);
    TSD_HIP_CHECK(ctx, hipGetLastError());
  }
  if (icp_from_host) {
    rc = launch_icp(ctx, ia, s->d_state->icpP, s->d_rays_local, s->st_h_ranges, s->st_h_mask, &sp, fold_argmax ? &prel : nullptr);
    if (rc != TSD_OK) return rc;
    rc = scan_stage_device(s);
    if (rc != TSD_OK) return rc;
  } else {
    if (!s->scan_bar && !host_saw_event(ctx->ev_h2d, staged_ahead ? 2 : 40)) TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_h2d, 0));
    rc = launch_icp(ctx, ia, s->d_state->icpP, s->d_rays_local, d_ranges, d_mask, &sp, fold_argmax ? &prel : nullptr);
    if (rc != TSD_OK) return rc;
  }
  lap.lap(3);
  PushArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.beams = s->beams;                                   // LDS size of the launch
  pa.max_range = s->max_range;                           // tile window of the launch (the rest is read on the device)
  // TSD_HALO_KERNEL=1: the push's halo pass as a kernel of its own even here (the form every other path uses; A/B)
  static const bool halo_in_raycast = [] { const char* e = std::getenv("TSD_HALO_KERNEL"); return !(e && *e == '1'); }();
  HaloArgs halo;
  std::memset(&halo, 0, sizeof(halo));
  if (!async_map) {
    if (int rcd_ = drain_async_push(ctx)) return rcd_;   // (a push left on the push stream by an earlier, asynchronous scan)
    if (!host_saw_event(ctx->ev_tables, staged_ahead ? 2 : 60)) TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_tables, 0));
    {
      LaunchTarget tg;
      tg.rmq = s->st_rmq;                                  // this scan's tables (the sensor's own buffers)
      TargetScope scope(ctx, &tg);
      // the registration moves the sensor by at most the gate (a larger step is rejected: pose unchanged)
      // (the push's halo pass is left to the ray cast that follows it at once: k_raycast's prologue, raycast_kernels.hip)
      rc = launch_push(ctx, pa, s->pos[0], s->pos[1], gates->reg_trs_max, &s->d_state->push, d_ranges, d_mask_push, nullptr, halo_in_raycast ? &halo : nullptr);
    }
    if (rc != TSD_OK) return rc;
    ctx->epoch++;                                          // the grid changes
    lap.lap(4);
    // the next scan's ray cast, right behind the push (see above): the host's work on the next scan no longer sits
    // between this push and that ray cast
    rc = launch_raycast(ctx, ra, &s->d_state->rc, s->d_rays, halo_in_raycast ? &halo : nullptr);
    if (rc != TSD_OK) {
      if (halo_in_raycast) (void)launch_push_halo(ctx, halo);      // (the grid's halos must not stay behind the push whatever happened to the ray cast)
      return rc;
    }
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
    unsigned long long spins = 0;
    while (!scan_result_arrived(s, seq)) {
      ++spins;
      // A registration takes 0.15-0.3 ms.  Past that, nudge the runtime: with other streams in the process (a communicator's,
      // a framework's) it was seen to sit on an enqueued launch until the next query / synchronisation of the stream -- a
      // 40 ms stall at the same scan of every run (profiles/r2_dist_stall.txt); a stream query is a few microseconds.
      if ((spins & 0x3FFFull) == 0) { (void)hipStreamQuery(ctx->stream); (void)hipStreamQuery(ctx->stream2); }
      if (spins > 4000000ull) {            // ~ a tenth of a second: something is wrong, fall back to a real wait
        TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (!scan_result_arrived(s, seq))
          return set_error(ctx, TSD_E_HIP, "tsd_scan: result record never arrived", hipSuccess);
        break;
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
  lap.lap(6);
  if (s->h_bar_mismatch && __atomic_load_n(s->h_bar_mismatch, __ATOMIC_ACQUIRE) != 0)
    return set_error(ctx, TSD_E_HIP, "tsd_scan: the scan the host stored into device memory through the PCIe BAR is not what the device read "
                                     "(TSD_SCAN_BAR_VERIFY); run with TSD_SCAN_PINNED=1 on this platform", hipSuccess);
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
  // (through the context's stream and waited for: a plain hipMemset would bring the NULL stream alive, see capi.hip)
  if (ok) { A(hipMemsetAsync(s->d_icp_seed, 0, icp_seed_bytes(s->beams), ctx->stream)); A(hipStreamSynchronize(ctx->stream)); }
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
  unsigned long long spins = 0;
  while (!scan_result_arrived(s, s->seq)) {
    if (++spins > 4000000ull) {            // something is wrong: a real wait on the sensor's stream
      if (hipStreamSynchronize(s->stream) != hipSuccess || !scan_result_arrived(s, s->seq)) return TSD_E_HIP;
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
    e.post.publish_done = 1;               // (tsd_batch_push gates this robot's push on it)
  }
  // (step 0's searches on helper workgroups: measured worth it for batches of up to four registrations -- +6 % scans/s at two per batch,
  // even at four, -2 to -10 % at eight, where forty more polling workgroups sit beside the ray casts)
  for (int i = 0; i < n; i++) {
    h_icp[i].seed = icp_batch_seed_args(ctx, sensors[i]->d_icp_seed, sensors[i]->beams, max_beams);
    if (n > 4) h_icp[i].seed.helpers = 0;
  }
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
      // (four launches for all of them: launch_preregistration_batch)
      std::vector<tsd_sensor*> armed;
      for (int i = 0; i < n; i++) if (sensors[i]->pre_armed) armed.push_back(sensors[i]);
      rc = launch_preregistration_batch(ctx, ctx->stream, armed.data(), (int)armed.size());
      if (rc != TSD_OK) return FAIL(rc);
      for (tsd_sensor* s : armed) { s->pre_armed = false; s->pre_ran = true; }
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
  if (ctx->push_multi && b->n >= 2 && b->n <= push_multi_max_robots()) {
    // The robots' pushes in ONE pass per tile (push_multi.hip): every tile the batch touches is read and written once, each robot's
    // update applied in the batch's order -- the grid the serial pushes below leave, cell for cell.  One gate for all registrations.
    const int n = b->n;
    const unsigned long long* seqp[16]; unsigned long long seqv[16]; PushArgs* pushp[16];
    const PushArgs* ap[16]; const double* rg[16]; const uint8_t* mk[16]; const char* rq[16];
    double cx[16], cy[16], sl[16], mr[16]; int bm[16];
    for (int i = 0; i < n; i++) {
      tsd_sensor* s = b->sensors[(size_t)i];
      const size_t nb = (size_t)s->beams;
      const char* d_scan = b->d_stage_cur + b->scan_off[(size_t)i];
      seqp[i] = &s->d_state->done_seq; seqv[i] = b->seqs[(size_t)i]; pushp[i] = &s->d_state->push;
      ap[i] = &s->d_state->push; rg[i] = reinterpret_cast<const double*>(d_scan); mk[i] = reinterpret_cast<const uint8_t*>(d_scan + nb * 9);
      rq[i] = s->d_rmq2[s->rmq_slot];
      cx[i] = s->pos[0]; cy[i] = s->pos[1]; sl[i] = b->gates[(size_t)i].reg_trs_max; mr[i] = s->max_range; bm[i] = s->beams;
    }
    if (gate) { if (int rcg = launch_wait_seq_multi(ctx, n, seqp, seqv, pushp, b->d_gate_err, b->poll_bound)) return rcg; }
    if (int rc = launch_push_multi(ctx, ctx->stream, n, ap, rg, mk, rq, cx, cy, sl, bm, mr)) return rc;
    ctx->epoch++;
    b->push_enqueued = true;
    return TSD_OK;
  }
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
    if (!scan_result_arrived(b->sensors[(size_t)i], b->seqs[(size_t)i])) return 0;
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

