// tsd_ctx.hpp -- host-side state behind the opaque tsd_ctx of include/tsd_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <mutex>

#include "tsd_device.hpp"
#include "../../include/tsd_hip.h"

namespace tsd {

// scalar inputs of a push, passed by value as a kernel argument
struct PushArgs {
  double Pi[6];              // first two rows of pose^-1
  double trx, try_;          // Sensor::getPosition
  double phi_min, ang_res_inv, phi_lower, phi_upper;
  double max_range, min_range, low_refl;
  int beams;
  int enabled;               // 0: this push was gated off on the device (fused scan): every kernel is a no-op
};

struct RaycastArgs {
  double Pi[6];
  double trx, try_;
  double gxmin, gymin, gxmax, gymax;   // RayCastPolar2D::_xmin.. (RayCastPolar2D.cpp:128-146)
  double idx_min, idx_max;             // _idxMin / _idxMax (:148-149)
  int beams;
  int pad;
};

struct IcpArgs {
  double P[6];               // first two rows of the pre-registration sensor pose
  double min_x, max_x, min_y, max_y;
  double thr0, min_sqr, multiplier;    // DistanceFilter state (DistanceFilter.cpp:11-20)
  int iterations;
  int n_model, n_scene;      // direct mode
  int beams;                 // fused mode (compaction from per-beam arrays) when > 0
  int ccw;                   // model slots ascend counter-clockwise about the sensor (1) or clockwise (0)
  int estimator;             // TSD_ESTIMATOR_*
  double Tinit[6];           // rows 0, 1 of Icp::iterate's Tinit (identity in registration_mode 0)
  const double* Tinit_dev;   // the same on the device (fused registration_mode 3: the pre-registration's result, first six doubles), or nullptr
};

struct IcpResultDev {
  double T[9];
  double rms;
  int pairs, iterations, state, n_model, n_scene, reserved;
};

// what one fused scan reports (layout of tsd_scan_result in include/tsd_hip.h)
struct ScanResultDev {
  IcpResultDev icp;
  double pose[9];
  int reg_error, pushed, no_model, reserved;
  unsigned long long seq;    // the scan's sequence number
};
// On its way to the host the record travels as SCAN_RESULT_WORDS tagged 8-byte words {low half of seq, four bytes of the record}
// (scan_post_body): the pinned buffer holds those, the host decodes them into a ScanResultDev of its own once all carry the tag.
constexpr int SCAN_RESULT_WORDS = (int)(sizeof(ScanResultDev) / 4);
static_assert(sizeof(ScanResultDev) % 8 == 0 && SCAN_RESULT_WORDS <= 64, "ScanResultDev: one store instruction of one wave");

// gates of ThreadLocalize (ThreadLocalize.cpp:593-600, :728-736; ThreadLocalize.h:63-64)
struct GateArgs {
  double reg_trs_max, reg_sin_rot_max, trs_min, rot_min;
};

// device-resident SensorPolar2D + ThreadLocalize pose bookkeeping of one robot (fused scan path)
struct SensorDev {
  double pose[9];            // Sensor::_T
  double last_pose[9];       // ThreadLocalize::_lastPose
  int have_last_pose;
  int pad;
  double last_angle;         // ThreadLocalize::calcAngle(_lastPose), kept with it (the push gate needs it every scan)
  RaycastArgs rc;            // arguments of the NEXT ray cast / registration, derived from `pose`
  double icpP[6];
  PushArgs push;             // arguments of this scan's push (enabled = gate result)
  unsigned long long done_seq;   // sequence number of the latest scan whose epilogue has written all of the above (device scope):
                                 // what the gate kernel ahead of a batched robot's push waits for (launch_wait_seq)
};

// fused scan path: what k_icp's epilogue needs (st == nullptr: plain registration)
struct ScanPostArgs {
  SensorDev* st;
  double* rays;              // world ray map of the sensor, turned in place
  ScanResultDev* out;        // coherent host memory
  unsigned long long seq;
  GateArgs gates;
  double gmin_x, gmax_x, gmin_y, gmax_y;   // TsdGrid::getMin/Max* (RayCastPolar2D's isInsideGrid)
  int beams;
  int publish_done;          // also publish st->done_seq (agent-scope release): a gate kernel on another stream waits for it (batched scans)
  PushArgs* push_copy;       // asynchronous mapping: where this scan's push arguments are left for a push that runs beside the NEXT
                             // registration (whose epilogue rewrites st->push); nullptr = strict order, the push reads st->push
};

// inclusive tile rectangle (empty when x1 < x0)
struct TileBox {
  int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
  bool empty() const { return x1 < x0 || y1 < y0; }
  void add(const TileBox& o)
  {
    if (o.empty()) return;
    if (empty()) { *this = o; return; }
    if (o.x0 < x0) x0 = o.x0; if (o.y0 < y0) y0 = o.y0; if (o.x1 > x1) x1 = o.x1; if (o.y1 > y1) y1 = o.y1;
  }
};

// Where a ray cast / registration of the concurrent multi-robot path runs and writes (tsd_scan_begin / _finish): the
// sensor's own stream and output buffers instead of the context's.  Set for the duration of the launches by the entry
// point (calls on one context are serialised by the caller), nullptr otherwise.
struct LaunchTarget {
  hipStream_t stream = nullptr;                                       // ray cast + registration
  double* coords = nullptr; double* normals = nullptr; uint8_t* mask_m = nullptr;   // ray-cast outputs
  IcpResultDev* icp_res = nullptr; double* trace = nullptr;
  void* icp_seed = nullptr; int icp_seed_points = 0;                  // the registration's helper hand-off (icp_seed_bytes(points))
  char* rmq = nullptr;                                                // range-query tables of the scan's push
  hipEvent_t rc_done = nullptr;                                       // launch_raycast: completes with the ray cast itself (the kernel's own stop
  bool rc_done_used = false;                                          // event: no marker behind it); used = false when the dispatch is being timed
};

// ---- batched scans (tsd_batch_*): one launch of each kernel for the robots of a batch; block (.., y) of the batched ray cast /
// block x of the batched registration and tables kernels read their arguments from entry y / x of these device arrays
struct RaycastBatchEntry {
  const RaycastArgs* a_dev; const double* rays;
  double* coords; double* normals; uint8_t* mask;
};
constexpr int RC_BATCH_BYVAL = 16;     // ray-cast entries that travel as kernel arguments (640 B): no copy to wait for
struct RaycastBatchArgs { RaycastBatchEntry e[RC_BATCH_BYVAL]; };
struct TablesBatchEntry {
  const double* ranges; const uint8_t* mask; char* rmq;
  double phi_min, ang_res;
  int beams, pad;
};
// step 0's searches of a registration, done by helper workgroups (icp_kernels.hip, ICP_HELPER_POINTS): the hand-off granules
// ([2][stride] 8-byte {launch number, value}), this launch's number, the helper workgroups the launch brings
struct IcpSeedArgs { unsigned long long* g; unsigned int seq; int helpers; int stride; };
struct IcpBatchEntry {
  IcpArgs a;
  const double* P_dev; const double* coords; const uint8_t* mask_m; const double* rays_local; const double* ranges;
  const uint8_t* mask; IcpResultDev* out; double* trace; const double* normals;
  ScanPostArgs post;
  const unsigned int* rc_flag;       // the slot's hand-off words (nullptr: the stream waited already): [0] number of the latest batch
                                     // whose ray casts are done, [1] number of the latest batch the host gave up on (tsd_batch_begin failed)
  unsigned int rc_target;
  unsigned int poll_bound;           // polls (about a microsecond each) before the wait gives up and REPORTS it (BATCH_FAIL_*)
  IcpSeedArgs seed;
};
// why a batched registration did not run (ScanResultDev::reserved / tsd_scan_result.reserved; 0 = it ran)
constexpr int BATCH_FAIL_TIMEOUT = 1;    // its ray casts never reported done within the poll bound
constexpr int BATCH_FAIL_ABORTED = 2;    // the host abandoned the batch
constexpr unsigned int BATCH_POLL_BOUND = 1u << 21;   // ~2 s

constexpr size_t KERNEL_TIMER_SAMPLES = 65536;
struct KernelTimer {
  double total_ms = 0.0;
  double sum_sq = 0.0, min_ms = 1e300, max_ms = 0.0;   // of the timed dispatches (tsd_profile_get_spread)
  int launches = 0;
  unsigned tick = 0;         // launches seen (sampling: every profile_every-th one is timed)
  std::vector<float> samples;    // the timed dispatches themselves, in order (tsd_profile_get_samples; at most KERNEL_TIMER_SAMPLES are kept)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace tsd

struct tsd_sensor;
struct tsd_ctx {
  int device = 0;
  int n_cus = 256;                           // compute units of the device (MI355X: 256): persistent grids are sized by it
  hipStream_t stream = nullptr;
  std::mutex order_mutex;                    // the ordered sections of the concurrent multi-robot path and every grid-writing entry point
  std::mutex misc_mutex;                     // timers / per-kernel attribute cache: touched by launches that run outside the caller's lock
  unsigned long long ticket = 0;             // order of the ray casts / pushes of the concurrent multi-robot path
  unsigned long long last_push_ticket = 0;
  std::vector<tsd_sensor*> sensors;          // device sensors attached to this grid (multi-robot mode, SlamNode.cpp:101-122)
  std::vector<struct tsd_batch*> batches;    // batch slots of the multi-robot path
  hipEvent_t ev_grid = nullptr;              // "every grid write enqueued so far is done" (recorded on `stream` by tsd_scan_begin)
  tsd::GridDev grid{};
  int map_log2 = 0;
  std::string err;

  // push state
  char* d_rmq = nullptr;                     // range-query tables of the current scan (k_push_tables): one of d_rmq2
  char* d_rmq2[2] = {nullptr, nullptr};      // (the next scan's tables are built while the current push still reads its own)
  int rmq_slot = 0;
  size_t tables_lds_configured = 0;          // dynamic LDS k_push_tables was configured for on this context's device
  std::map<const void*, size_t> lds_configured;   // the same for the k_icp instantiations
  hipEvent_t ev_h2d = nullptr;               // fused scan: the scan's copy (side stream) is complete
  uint32_t* d_tile_rec = nullptr;            // [tiles] what the last push did to every tile
  uint8_t* d_dirty = nullptr;                // [tiles] written by freeFootprint since the last push
  uint32_t* d_tile_totals = nullptr;         // [tiles][8] records summed over the pushes since the last reset
  unsigned long long* d_pushes = nullptr;    // [2] pushes since the last reset, pushes whose launch window missed the sensor
  // tile window of the push launches: what the last push covered and what freeFootprint dirtied since
  tsd::TileBox box_prev{}, box_dirty{};
  uint32_t* d_list = nullptr;               // [tiles] work list of the current push (tile | kind << 28)
  uint32_t* d_list_h = nullptr;             // [tiles] the UPDATE tiles of that list k_push_halo has work for (materialised by the push, or dirty)
  char* d_list_aux = nullptr;               // [tiles] PushListAux of every list entry (push_kernels.hip): beam window, partition weight,
                                            // the linear forms of k_push_update's beam estimate
  tsd::PushArgs* d_push_args = nullptr;     // arguments of an unfused tsd_push (the fused scan keeps them in the sensor state)
  unsigned int* d_list_cnt = nullptr;       // by push parity: UPDATE tiles listed, other tiles listed, the ticket heads of k_push_update's tile queue
  unsigned int push_parity = 0;
  unsigned long long epoch = 0;             // bumped by everything that changes the grid, a sensor pose or the ctx's ray-cast outputs
  hipStream_t stream2 = nullptr;             // side stream: the tables are built while ray cast / ICP run
  // asynchronous mapping (tsd_sensor_set_async_mapping): the fused scan's push on a stream of its own, beside the next registration
  hipStream_t stream_push = nullptr;
  hipEvent_t ev_async_rc = nullptr;          // "the next scan's ray cast has read the grid" (recorded on `stream`)
  hipEvent_t ev_async_push = nullptr;        // "the push enqueued last on stream_push is done" (one of a sensor's ev_slot_push[], not owned)
  unsigned int debug_push_stall_us = 0;      // tsd_debug_stall_push_stream (tests)
  bool async_pending = false;                // a push on stream_push that `stream` has not been ordered behind yet
  hipEvent_t ev_tables = nullptr;

  // scan staging: ring of pinned slots + device buffers
  static constexpr int kSlots = 8;
  int slot = 0;
  char* h_stage[kSlots] = {};
  hipEvent_t stage_ev[kSlots] = {};
  size_t stage_bytes = 0;
  double* d_ranges = nullptr;    // [TSD_MAX_BEAMS]
  uint8_t* d_mask = nullptr;     // [TSD_MAX_BEAMS]
  double* d_rays = nullptr;      // [2*TSD_MAX_BEAMS] world rays
  double* d_rays_local = nullptr;// [2*TSD_MAX_BEAMS]
  double* d_coords = nullptr;    // [2*TSD_MAX_BEAMS]
  double* d_normals = nullptr;   // [2*TSD_MAX_BEAMS]
  double* d_mnormals = nullptr;  // [2*TSD_MAX_ICP_POINTS] model normals of tsd_icp_normals, in the model's slot order
  uint8_t* d_mask_m = nullptr;   // [TSD_MAX_BEAMS]
  double* d_model = nullptr;     // [2*TSD_MAX_ICP_POINTS]
  double* d_scene = nullptr;     // [2*TSD_MAX_ICP_POINTS]
  int* d_morig = nullptr;        // [TSD_MAX_ICP_POINTS] original index of every angle-sorted model point
  int* d_start = nullptr;        // [TSD_MAX_ICP_POINTS] first search slot of every scene point
  int icp_shape = 0;             // 0 = default workgroup shape (env TSD_ICP_SHAPE for experiments)
  int push_multi = 1;            // the pushes of a batch of robots in one pass per tile (push_multi.hip; tsd_debug_set_push_multi: 0 = one push per robot)
  void* d_mp_mask = nullptr; void* d_mp_rec = nullptr; void* d_mp_list = nullptr;      // its per-window-tile masks, (tile, robot) records, tile list
  size_t mp_tiles = 0; unsigned mp_parity = 0;
  int icp_helpers = 1;           // step 0's searches by helper workgroups (tsd_debug_set_icp_helpers: 0 = the registration searches itself)
  void* d_icp_seed = nullptr;    // their hand-off buffer (icp_seed_bytes(TSD_MAX_ICP_POINTS))
  tsd::IcpResultDev* d_icp_res = nullptr;
  double* d_icp_trace = nullptr;            // [TSD_ICP_TRACE_MAX][TSD_ICP_TRACE_STRIDE]
  tsd::IcpResultDev* h_icp_res = nullptr;    // pinned
  char* h_out = nullptr;                     // pinned D2H staging (ray-cast outputs)
  size_t h_out_bytes = 0;

  // TSD_PDF pre-registration (tsdpdf.hip): one device + one pinned staging buffer, grown on demand
  char* d_pdf = nullptr; char* h_pdf = nullptr; size_t pdf_bytes = 0;

  // occupancy
  int8_t* d_occ = nullptr;       // persistent map (ThreadGrid::_occGridContent)
  int* d_occ_count = nullptr;
  unsigned int* d_occ_heads = nullptr;   // sharded counters (two sets, used in turn) of k_occ_mark's work list (occupancy_kernels.hip)
  int occ_parity = 0;
  uint32_t* d_occ_list = nullptr;
  int8_t* d_occ_out = nullptr;   // tsd_occupancy's device staging (allocated on first use, kept)
  hipStream_t stream_io = nullptr; hipEvent_t ev_io = nullptr;   // ... and the stream its copy to the host leaves on (created on first use)
  uint8_t* d_img = nullptr; size_t img_bytes = 0;   // tsd_color_image's (coordinate tables + image), grown on demand

  // profiling: bit i of profile_mask times kernel i (names in capi.hip: kKernelNames)
  unsigned profile_mask = 0;
  unsigned profile_every = 1;   // time every n-th launch of a selected kernel ("name/n" in tsd_profile_select)
  unsigned profile_every_k[16] = {};   // a kernel's own period ("name:m"), 0 = the list's
  bool profile = false;
  std::map<std::string, tsd::KernelTimer> timers;
  std::vector<hipEvent_t> event_pool;
};

// device-resident sensor of one robot (tsd_sensor_* / tsd_scan in include/tsd_hip.h)
struct tsd_sensor {
  tsd_ctx* ctx = nullptr;
  int beams = 0;
  double ang_res = 0, phi_min = 0, max_range = 0, min_range = 0, low_refl = 0;
  bool ccw = true;
  bool posed = false;
  tsd::SensorDev* d_state = nullptr;
  double* d_rays = nullptr;        // [2*beams] world rays, normalised to the cell size
  double* d_rays_local = nullptr;  // [2*beams]
  char* d_scan2[3] = {nullptr, nullptr, nullptr};   // ranges[beams] | mask[beams] | mask_push[beams], used in turn (split scan: 0 / 1)
  // The same three scans in pinned host memory, where the caller's arrays are copied first.  A scan that was NOT staged ahead is read
  // by its registration from here, over the host link (10 KB, requested at once at the top of k_icp): the registration is launched as
  // soon as the scan is in this buffer, and the device copy + the push's range tables follow on the side stream beside it.
  char* h_scan3[3] = {nullptr, nullptr, nullptr};
  char* hd_scan3[3] = {nullptr, nullptr, nullptr};  // device addresses of h_scan3
  hipEvent_t ev_scan_copy[3] = {nullptr, nullptr, nullptr};   // "the device copy out of h_scan3[i] is done" (before the host rewrites it)
  bool scan_copy_valid[3] = {false, false, false};
  const double* st_h_ranges = nullptr; const uint8_t* st_h_mask = nullptr;   // the staged scan's ranges / mask in h_scan3 (device addresses)
  // Where the device's memory is mapped into the host's address space (large PCIe BAR: every MI300-class server), d_scan2[] is
  // fine-grained device memory and the HOST writes a scan straight into it: no pinned copy, no device copy, and the registration reads
  // its 10 KB from local memory (1 us) instead of over the host link (3.5-4 us at the top of every registration: tools/exp/bar.hip).
  bool scan_bar = false;
  unsigned int* h_bar_mismatch = nullptr;   // TSD_SCAN_BAR_VERIFY=1: bytes in which a scan's two copies differed on the device (pinned, coherent)
  unsigned int* d_bar_mismatch = nullptr;   // ... its device address
  bool st_device_done = false;     // the staged scan's device copy and tables are enqueued
  int scan_slot = 0;
  tsd::ScanResultDev* h_result = nullptr;   // the last record that arrived, decoded (ordinary host memory)
  unsigned long long* h_rwords = nullptr;   // pinned, coherent: SCAN_RESULT_WORDS tagged words, written by the registration's epilogue
  tsd::ScanResultDev* d_result = nullptr;   // device address of h_rwords (ScanPostArgs::out)
  unsigned long long seq = 0;
  double pos[2] = {0, 0};          // host mirror of the sensor position (window of the push launches)
  // tsd_scan_stage / _submit / _collect: the scan that is staged (copied, tables built) and not yet submitted
  bool staged = false, submitted = false;
  const double* st_ranges = nullptr; const uint8_t* st_mask = nullptr; const uint8_t* st_mask_push = nullptr;
  char* st_rmq = nullptr; int st_slot = 0;
  int stage_slot = 0;              // the scan / table buffers are used in turn, THREE of them: the scan staged ahead of scan k+2 goes
                                   // where scan k was, and by then the host has seen the result of scan k+1, whose ray cast ran behind
                                   // the push of scan k on the stream -- so that push is done without any event on the stream.
                                   // That argument needs the STRICT order.  With asynchronous mapping ray cast k+1 runs behind push
                                   // k-1 and push k beside registration k+1 on the push stream: the host has no proof that it is
                                   // done, so the staging waits for the buffer's own push event (below).
  hipEvent_t ev_slot_push[3] = {nullptr, nullptr, nullptr};   // asynchronous mapping: "the push that read buffer i is done"
  bool slot_push_valid[3] = {false, false, false};            // ... recorded and not yet seen complete
  // fused registration_mode 3 (tsd_scan_preregister, tsdpdf.hip): inputs of the pre-registration that the next tsd_scan_submit runs
  // on the device between its ray cast and its registration; one device + one pinned buffer, grown on demand
  char* d_pre = nullptr; char* h_pre = nullptr; size_t pre_bytes = 0;
  char* h_pre_dev = nullptr;                // h_pre as the device sees it (looked up once per allocation)
  hipEvent_t ev_pre_done = nullptr;         // the in-flight scan's pre-registration kernels have read their inputs (the arg-max's own stop event)
  bool pre_done_valid = false;
  size_t pre_res_off_hdr = 0, pre_res_off_res = 0;      // where the collected scan's header / result are in h_pre (the layout may be re-armed meanwhile)
  bool async_mapping = false;               // tsd_sensor_set_async_mapping
  tsd::PushArgs* d_push_slot = nullptr;          // [2] push arguments by scan parity (asynchronous mapping)
  hipEvent_t ev_pre = nullptr;              // the pre-registration's inputs are on the device (copied on the side stream by tsd_scan_preregister)
  bool pre_copied = false;
  bool pre_bar = false;       // d_pre is fine-grained device memory (tsd_sensor::scan_bar) ...
  bool pre_direct = false;    // ... and the armed inputs were stored into it by the host: nothing to copy, nothing to wait for
  bool pre_armed = false, pre_ran = false;
  unsigned int* d_pre_flag = nullptr;        // k_icp_pre: the launch number of the last arg-max whose TBest stands (its own small allocation)
  unsigned int pre_seq = 0;
  struct PreLayout {
    size_t off_S, off_ms, off_msp, off_dc, off_dt, in_bytes;             // inputs (host -> device each scan)
    size_t off_mo_m, off_mo_s, off_phi_m, off_phi_s, off_C, off_K, off_prob, off_hdr, off_res;
    int n, span, trials, size_control_set, max_cand;
    double phi_max, zrand;
  } pre{};
  bool rc_pending = false;         // the next scan's ray cast was enqueued behind this scan's push ...
  unsigned long long rc_epoch = 0; // ... when the context was in this state

  // concurrent multi-robot path (tsd_scan_begin / tsd_scan_wait / tsd_scan_finish): ray cast + registration on the
  // sensor's own stream into its own buffers, created on first use
  bool conc_ready = false;
  hipStream_t stream = nullptr;      // ONE stream per sensor: streams are multiplexed onto a few in-order hardware queues
  hipEvent_t ev_rc_done = nullptr, ev_icp_done = nullptr;
  bool rc_event_valid = false;     // ev_rc_done has been recorded at least once
  unsigned long long rc_ticket = 0;            // ticket of the sensor's most recent ray cast ...
  volatile int rc_recorded = 1;                // ... whose ev_rc_done record has been issued (by the sensor's own thread)
  double* d_coords = nullptr; double* d_normals = nullptr; uint8_t* d_mask_m = nullptr;
  tsd::IcpResultDev* d_icp_res = nullptr; double* d_icp_trace = nullptr; void* d_icp_seed = nullptr;
  char* d_rmq2[3] = {nullptr, nullptr, nullptr}; int rmq_slot = 0;
  char* h_stage2[2] = {nullptr, nullptr};   // pinned staging of the scan, alternating
  bool inflight = false;           // begin() without finish()
  tsd_gate_params conc_gates{};
  const double* conc_ranges = nullptr; const uint8_t* conc_mask_push = nullptr;
};

// one batch slot of the multi-robot path (tsd_batch_* in include/tsd_hip.h): its own stream, events and staging
struct tsd_batch {
  tsd_ctx* ctx = nullptr;
  int max_scans = 0;
  size_t scan_bytes = 0;             // bytes of one scan (ranges | mask | mask_push) in the staging buffers, 64-byte aligned
  size_t head_bytes = 0;             // bytes of the three entry arrays in front of the scans
  hipStream_t stream = nullptr;
  hipEvent_t ev_rc_done = nullptr, ev_icp_done = nullptr, ev_copy_done = nullptr;
  unsigned int* d_rc_flag = nullptr;    // [2]: number of the slot's latest batch whose ray casts have finished (set by a one-wave kernel
  unsigned int rc_batches = 0;          // behind them on the grid's stream) / that the host abandoned; batches begun on this slot
  bool dev_wait = false;                // the two hand-offs of a batch are waits ON THE DEVICE (proven possible by the probe in
                                        // tsd_batch_create) instead of stream events
  unsigned int poll_bound = tsd::BATCH_POLL_BOUND;
  unsigned int* h_gate_err = nullptr;   // pinned, coherent: set by a push gate (k_wait_seq) whose registration never reported done
  unsigned int* d_gate_err = nullptr;   // its device address
  char* h_stage = nullptr;           // pinned: entries + scans of the batch being enqueued
  char* d_stage2[2] = {nullptr, nullptr};   // device copies, alternating (the pushes of the previous batch still read theirs)
  int stage_slot = 0;
  int n = 0;                         // scans of the batch in flight (0: free)
  bool push_enqueued = false;
  std::vector<tsd_sensor*> sensors;
  std::vector<unsigned long long> seqs;
  std::vector<tsd_gate_params> gates;
  std::vector<size_t> scan_off;      // offset of scan i in d_stage2[slot of the batch]
  char* d_stage_cur = nullptr;
};

namespace tsd {

int set_error(tsd_ctx* ctx, int code, const char* what, hipError_t e);
// launch target of the entry point that is enqueueing on THIS thread (the sensors' private streams are driven by their
// own threads, outside the caller's grid lock)
extern thread_local const LaunchTarget* g_launch_target;
inline const LaunchTarget* launch_target() { return g_launch_target; }
inline hipStream_t launch_stream(const tsd_ctx* c) { return (g_launch_target && g_launch_target->stream) ? g_launch_target->stream : c->stream; }
struct TargetScope {
  TargetScope(tsd_ctx*, const LaunchTarget* t) { g_launch_target = t; }
  ~TargetScope() { g_launch_target = nullptr; }
};
#define TSD_HIP_CHECK(ctx, call)                                                     \
  do {                                                                               \
    hipError_t _e = (call);                                                          \
    if (_e != hipSuccess) return tsd::set_error((ctx), TSD_E_HIP, #call, _e);        \
  } while (0)

// Event-timed launch.  Default: the two events are handed to hipExtLaunchKernelGGL, which stamps them with the
// dispatch's own begin / end (the duration rocprofv3 reports, free of the gaps between stream operations).
// around = true brackets whatever the scope enqueues with hipEventRecord (several operations).
struct ScopedKernelTimer {
  tsd_ctx* ctx; const char* name; hipEvent_t a = nullptr, b = nullptr; bool around;
  ScopedKernelTimer(tsd_ctx* c, const char* n, bool around_ = false);
  ~ScopedKernelTimer();
};
void drain_timers(tsd_ctx* ctx);
int drain_async_push(tsd_ctx* ctx);          // asynchronous mapping: order the context's stream behind the push stream's last push (capi.hip)
bool kernel_is_timed(const tsd_ctx* ctx, const char* name);

// per-file launchers.  The *_dev pointers are the fused scan path: the kernels then read their pose
// dependent arguments from the device-resident sensor state instead of the by-value copy.
// (cx, cy) is where the host knows the sensor to be and `slack` how far the device-side pose may be from it
// defer_halo != nullptr: k_push_halo is NOT launched; *defer_halo receives its arguments for the ray cast that follows the push and
// carries that pass in its prologue (launch_raycast(.., halo)) -- or for launch_push_halo() where no such ray cast follows
struct HaloArgs;
int launch_push(tsd_ctx* ctx, const PushArgs& a, double cx, double cy, double slack, const PushArgs* a_dev = nullptr,
                const double* d_ranges = nullptr, const uint8_t* d_mask = nullptr, hipStream_t stream = nullptr /* nullptr: ctx->stream */,
                HaloArgs* defer_halo = nullptr);
int launch_push_halo(tsd_ctx* ctx, const HaloArgs& h, int n_window = 2048, hipStream_t stream = nullptr);
int launch_push_tables(tsd_ctx* ctx, hipStream_t stream, int beams, const double* d_ranges, const uint8_t* d_mask,
                       double phi_min, double ang_res);
size_t push_rmq_bytes(int beams);
size_t push_list_aux_bytes();
// fused registration_mode 3: normals -> lists -> scoring -> arg-max on `stream`, model = the ray cast's outputs on the device;
// *tinit_dev = where the registration kernel finds Tinit (tsdpdf.hip)
// fold != nullptr: the arg-max is NOT launched; *fold receives what launch_icp(.., pre) needs to carry it
struct IcpPreLaunch;
int launch_preregistration(tsd_ctx* ctx, tsd_sensor* s, hipStream_t stream, const double* d_coords, const uint8_t* d_mask_m,
                           const double* d_pose6, const double** tinit_dev, hipEvent_t before_score = nullptr, IcpPreLaunch* fold = nullptr);
int launch_preregistration_batch(tsd_ctx* ctx, hipStream_t stream, tsd_sensor* const* sensors, int n);
size_t push_list_cnt_bytes();
int launch_free_footprint(tsd_ctx* ctx, unsigned minX, unsigned maxX, unsigned minY, unsigned maxY);
int launch_neg_scan(tsd_ctx* ctx);
int launch_export_tiles(tsd_ctx* ctx, int t0, int n, double* d_t, double* d_w);
int launch_import_tiles(tsd_ctx* ctx, int t0, int n, const double* d_t, const double* d_w);
int launch_grid_digest(tsd_ctx* ctx, unsigned long long* d_out, double* d_sums);
int launch_raycast(tsd_ctx* ctx, const RaycastArgs& a, const RaycastArgs* a_dev = nullptr, const double* d_rays = nullptr,
                   const HaloArgs* halo = nullptr /* the push right ahead of this launch left its halo pass to it */);
// pre: fused registration_mode 3 -- the pre-registration's arg-max rides as the first workgroup of the registration's launch (k_icp_pre;
// only where icp_pre_supported() says so); pre->done is recorded when the launch has completed
bool icp_pre_supported(const tsd_ctx* ctx, const IcpArgs& a);
int launch_icp(tsd_ctx* ctx, const IcpArgs& a, const double* P_dev = nullptr, const double* d_rays_local = nullptr,
               const double* d_ranges = nullptr, const uint8_t* d_mask = nullptr, const ScanPostArgs* post = nullptr, const IcpPreLaunch* pre = nullptr);
int launch_icp_pairs(tsd_ctx* ctx, const IcpArgs& a, int* d_pairs);
int icp_pairs_cap(int n_model, int n_scene);
int launch_scan_prepare(tsd_ctx* ctx, SensorDev* st);
// one wave on the context's stream that waits (on the device, bounded) until *seq == value; when the bound runs out it switches
// the push behind it off (push->enabled = 0) and raises *err_host (coherent host memory) instead of letting stale arguments through
int launch_wait_seq(tsd_ctx* ctx, const unsigned long long* seq, unsigned long long value, PushArgs* push, unsigned int* err_host,
                    unsigned int poll_bound);
// one wave on `stream` that publishes *flag = value (device scope) once everything ahead of it on the stream is done
int launch_set_flag(tsd_ctx* ctx, hipStream_t stream, unsigned int* flag, unsigned int value);
int launch_stall(tsd_ctx* ctx, hipStream_t stream, unsigned int us);
// start-up probe of tsd_batch_create: can a kernel on stream `a` wait for a flag that a kernel launched AFTER it on stream `b`
// sets?  (No when the two streams share an in-order hardware queue, or when something serialises dispatches.)
int probe_cross_stream_wait(tsd_ctx* ctx, hipStream_t a, hipStream_t b, unsigned int* d_flag2, bool* ok);
// batched launches on `stream`; the entry arrays live in device memory, `host` is the host copy they were staged from
int launch_raycast_batch(tsd_ctx* ctx, hipStream_t stream, const RaycastBatchEntry* d_entries, int n, int max_beams);
int launch_raycast_batch_byval(tsd_ctx* ctx, hipStream_t stream, const RaycastBatchEntry* h_entries, int n, int max_beams);   // n <= RC_BATCH_BYVAL
int launch_push_tables_batch(tsd_ctx* ctx, hipStream_t stream, const TablesBatchEntry* d_entries, int n, int max_beams);
int launch_icp_batch(tsd_ctx* ctx, hipStream_t stream, const IcpBatchEntry* host, const IcpBatchEntry* d_entries, int n);
int push_multi_max_robots();
int launch_push_multi(tsd_ctx* ctx, hipStream_t stream, int n, const PushArgs* const* a_dev, const double* const* d_ranges, const uint8_t* const* d_mask,
                      const char* const* d_rmq, const double* cx, const double* cy, const double* slack, const int* beams, const double* max_range);
int launch_wait_seq_multi(tsd_ctx* ctx, int n, const unsigned long long* const* seq, const unsigned long long* value, PushArgs* const* push,
                          unsigned int* err_host, unsigned int poll_bound);
size_t icp_seed_bytes(int points);
IcpSeedArgs icp_batch_seed_args(const tsd_ctx* ctx, void* buf, int beams, int batch_beams);

int launch_calibrate(tsd_ctx* ctx, double* t, double* w, size_t n);
int launch_occupancy(tsd_ctx* ctx, int8_t* d_out, int inflate, int inflate_factor);
size_t occ_heads_bytes();
int launch_color_image(tsd_ctx* ctx, const double* d_px, const double* d_py, unsigned width, unsigned height, uint8_t* d_image);
size_t icp_lds_bytes();

// pose helpers (host): textbook LU inverse with partial pivoting in the order of
// gsl_linalg_LU_decomp / LU_invert as used by obvious::Matrix::invert (gsl/Matrix.cpp:168-179)
void mat3_inv(const double A[9], double Ainv[9]);

}  // namespace tsd
