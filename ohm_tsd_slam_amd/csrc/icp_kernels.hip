// icp_kernels.hip -- the registration step of ThreadLocalize::doRegistration, registration_mode 0
// (ThreadLocalize.cpp:571-581): Icp::iterate (Icp.cpp:464-512) over Icp::step (:410-462) with
//   OutOfBoundsFilter2D (OutOfBoundsFilter2D.cpp:27-37)  -> pre-filter
//   exact 1-NN (FlannPairAssignment.cpp:64-92)            -> bounded angular search in LDS (below)
//   DistanceFilter (DistanceFilter.cpp:32-64)             -> d2 <= thr, thr = max(thr*m, min^2)
//   ReciprocalFilter (ReciprocalFilter.cpp:32-78)         -> best scene point per model point
//   ClosedFormEstimator2D (ClosedFormEstimator2D.cpp:36-109)
// as ONE persistent single-workgroup kernel: all icp_iterations steps run inside one launch; the
// model lives in LDS, every thread keeps its scene points in registers, the pair sums are reduced with
// DPP row shifts inside a wave and through LDS across waves, and there is no host round trip between
// steps.  In the fused mode the kernel first does dataToCartesianVectorMask (Sensor.cpp:168-190) and
// the maskMatrix compaction (ThreadLocalize.cpp:738-755) from the ray-cast outputs.
//
// The kernel is instruction-issue bound on ONE compute unit (30 dependent steps leave no room for a
// grid-wide barrier), so the design minimises instructions per step:
//
// Exact nearest neighbour without a kd-tree.  The model points are ordered by polar angle about the
// sensor (the ray-cast emits them in beam order; tsd_icp sorts on the host).  For a scene point s and
// a model point m at angular separation sigma <= 90 deg, |s - m| >= |s| sin(sigma), and >= |s| beyond
// 90 deg.  Hence, if a candidate on the counter-clockwise side of s and one on the clockwise side both
// have that bound above min(best d2, thr), no slot outside the arc between them can hold a nearer point
// or one within the DistanceFilter threshold: the filtered pair list equals exact-NN + DistanceFilter.
// Ties (equal d2) go to the lower original model index, like a first-minimum linear scan.
//   tier 0  every point keeps its last neighbour k and a lower bound lb on its distance to every OTHER
//           model point; after the scene moved by at most `disp` the bound is lb - disp.  If
//           |s - m_k| < lb the neighbour is unchanged; if |s - m_k|^2 > thr and lb^2 > thr the pair is
//           dropped by the DistanceFilter whoever the neighbour is.  One LDS read per point.
//   tier 1  points failing tier 0 are appended to a dense LDS work list (so that the SIMT lanes stay
//           full however few points need it) and get a 13-slot window around k evaluated with all LDS
//           reads in flight; the two window ends supply the bound for everything outside.
//   tier 2  what the window cannot prove (outliers far from the model, NN far from k) is searched by a
//           whole wave: 64 consecutive slots per step with a DPP minimum, widened until proven.
//
// No dense contraction anywhere => no MFMA; fp64 VALU + LDS.  Latency-bound: reported as ms/iterate.
#include "tsd_ctx.hpp"
#include <climits>

namespace tsd {

constexpr int ICP_MAXW = 16;                    // waves per workgroup at most
constexpr double SLACK = 1.0 - 1e-9;            // conservative factor on every pruning bound
constexpr int HW = 6;                           // tier-1 window: k-6 .. k+6
constexpr int IR_CNT = 32, IR_RMAX = 33;   // words of IcpLds::ired
#ifdef TSD_ICP_STAMPS
constexpr int IR_DBG = 40;
#endif

struct IcpLds {
  double2* mxy;                    // model (angular order)
  double2* uxy;                    // unit direction of every model point (0,0 for a point at the origin)
  unsigned long long* slotD;       // [cap] reciprocal filter: min d2 (bit pattern) per model slot
  int* slotI;                      // [cap] winning scene index per model slot
  int* morig;                      // original model index of a slot (tie-breaking)
  double2* list_xy;                // [lcap] work list: point
  int* list_k;                     // [lcap]            its last neighbour slot
  double* res_d;                   // [lcap] results: squared distance to the nearest neighbour
  double* res_lb;                  // [lcap]          lower bound (distance) to every other model point
  int* res_k;                      // [lcap]          neighbour slot (-1: unresolved)
  double* red;                     // [ICP_MAXW][8] wave partials (first pass)
  double* red2;                    // [ICP_MAXW][2] wave partials (second pass)
  int* ired;                       // [64] counters
  // setup only (alias the work list)
  double2* stage_s;                // compacted scene
  int* start;                      // first search position of every compacted scene point
};

__host__ __device__ inline int icp_list_cap(int cap) { return cap < 1024 ? cap : 1024; }
__host__ __device__ inline size_t icp_lds_bytes_for(int cap)
{
  const size_t lc = (size_t)icp_list_cap(cap);
  // the staging (cap double2 + cap int) aliases the list + result arrays: 40 * lc >= 20 * cap
  return sizeof(double2) * 2 * (size_t)cap + sizeof(unsigned long long) * (size_t)cap + sizeof(int) * 2 * (size_t)cap +
         (sizeof(double2) + sizeof(int) + 2 * sizeof(double) + sizeof(int)) * lc +
         sizeof(double) * (ICP_MAXW * 8 + ICP_MAXW * 2) + sizeof(int) * 64 + 64;
}

// ---- wave reductions: DPP row shifts inside the 16-lane rows, then the four row results through
// SGPRs.  Fixed order => deterministic.  (v_add_f64 has no DPP form: two 32-bit DPP moves + add.)
template <int CTRL>
__device__ __forceinline__ double dpp_shr0(double v)      // lanes without a source read 0.0
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int src)   // src wave-uniform
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_total(double v)
{
  v += dpp_shr0<0x111>(v);    // row_shr:1
  v += dpp_shr0<0x112>(v);    // row_shr:2
  v += dpp_shr0<0x114>(v);    // row_shr:4
  v += dpp_shr0<0x118>(v);    // row_shr:8   -> lane 15 of every row holds the row total
  return ((read_lane(v, 15) + read_lane(v, 31)) + read_lane(v, 47)) + read_lane(v, 63);
}
template <int CTRL>
__device__ __forceinline__ double dpp_shr_inf(double v)   // lanes without a source read +inf
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0x7FF00000, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min(double v)
{
  v = fmin(v, dpp_shr_inf<0x111>(v));
  v = fmin(v, dpp_shr_inf<0x112>(v));
  v = fmin(v, dpp_shr_inf<0x114>(v));
  v = fmin(v, dpp_shr_inf<0x118>(v));
  return fmin(fmin(read_lane(v, 15), read_lane(v, 31)), fmin(read_lane(v, 47), read_lane(v, 63)));
}

struct NnResult { double best, lbsq; int bk; bool resolved; };

// separation bound of model slot with unit direction u for the point (x, y): squared lower bound on the
// distance from (x, y) to ANY point at that angular separation or more; `cr` returns the side.
__device__ __forceinline__ double sep_bound(double x, double y, double rs2, double2 u, double& cr)
{
  cr = x * u.y - y * u.x;
  const double dt = x * u.x + y * u.y;
  const bool valid_u = (u.x != 0.0) || (u.y != 0.0);
  return valid_u ? (dt > 0.0 ? cr * cr : rs2) * SLACK : 0.0;
}

// tier 1: the 2*HW+1 slots around `c`, one lane per point, all LDS reads issued together
__device__ __forceinline__ NnResult window_search(const IcpLds& L, int nM, double x, double y, int c,
                                                  double thr, double sgn)
{
  NnResult r;
  r.best = __builtin_inf(); r.lbsq = 0.0; r.bk = -1; r.resolved = false;
  if (nM <= 2 * HW + 1) return r;
  int kk[2 * HW + 1];
  double2 m[2 * HW + 1];
#pragma unroll
  for (int j = 0; j < 2 * HW + 1; j++) {
    int k = c + j - HW;
    if (k < 0) k += nM;
    if (k >= nM) k -= nM;
    kk[j] = k;
    m[j] = L.mxy[k];
  }
  const double2 ulo = L.uxy[kk[0]], uhi = L.uxy[kk[2 * HW]];
  double best = __builtin_inf(), second = __builtin_inf();
  int bj = 0;
  bool tie = false;
#pragma unroll
  for (int j = 0; j < 2 * HW + 1; j++) {
    const double dx = x - m[j].x, dy = y - m[j].y;
    const double d = dx * dx + dy * dy;
    tie = tie || (d == best);
    if (d < best) { second = best; best = d; bj = j; tie = false; }
    else if (d < second) second = d;
  }
  int bk = kk[0];
#pragma unroll
  for (int j = 1; j < 2 * HW + 1; j++) if (bj == j) bk = kk[j];
  if (tie) {                                         // exact tie: lowest original index (rare)
    second = best;
#pragma unroll
    for (int j = 0; j < 2 * HW + 1; j++) {
      const double dx = x - m[j].x, dy = y - m[j].y;
      if (dx * dx + dy * dy == best && L.morig[kk[j]] < L.morig[bk]) bk = kk[j];
    }
  }
  const double rs2 = x * x + y * y;
  double crl, crh;
  const double l2lo = sep_bound(x, y, rs2, ulo, crl), l2hi = sep_bound(x, y, rs2, uhi, crh);
  // the low end must lie clockwise of s (in slot order) and the high end counter-clockwise
  const double lbo = fmin(crl * sgn <= 0.0 ? l2lo : 0.0, crh * sgn >= 0.0 ? l2hi : 0.0);
  r.best = best; r.bk = bk;
  r.lbsq = fmin(second, lbo);
  r.resolved = lbo > fmin(best, thr);
  return r;
}

// tier 2: the same proof by the whole wave for one point (x, y, start wave-uniform): 64 consecutive
// slots per step, widened towards the side that is not yet bounded.
__device__ __forceinline__ NnResult wave_search(const IcpLds& L, int nM, double x, double y, int start,
                                                double thr, double sgn, int lane)
{
  const double rs2 = x * x + y * y;
  double best = __builtin_inf(), second = __builtin_inf();
  double l2u = __builtin_inf(), l2d = __builtin_inf();
  bool up_done = false, dn_done = false;
  int bk = -1;
  int lo = 0, hi = -1;                               // visited offsets relative to `start` (empty)
  int cnt = nM < 64 ? nM : 64;
  int w0 = -(cnt / 2);
  for (;;) {
    const int o = w0 + lane;
    const bool act = lane < cnt;
    int k = start + o;
    if (k >= nM) k -= nM;
    if (k >= nM) k -= nM;
    if (k < 0) k += nM;
    if (k < 0) k += nM;
    double d = __builtin_inf(), l2 = 0.0, cr = 0.0;
    if (act) {
      const double2 m = L.mxy[k], u = L.uxy[k];
      const double dx = x - m.x, dy = y - m.y;
      d = dx * dx + dy * dy;
      l2 = sep_bound(x, y, rs2, u, cr);
    }
    const double wmin = wave_min(d);
    const unsigned long long eq = __ballot(act && d == wmin);
    if (eq) {
      int wl = __ffsll((long long)eq) - 1;
      if (__popcll(eq) > 1) {                        // exact tie inside the window: lowest original index
        int bo = INT_MAX;
        unsigned long long e = eq;
        while (e) {
          const int l = __ffsll((long long)e) - 1; e &= e - 1;
          const int mo = L.morig[__builtin_amdgcn_readlane(k, l)];
          if (mo < bo) { bo = mo; wl = l; }
        }
      }
      const int wk = __builtin_amdgcn_readlane(k, wl);
      const double wsec = wave_min(lane == wl ? __builtin_inf() : d);
      if (wmin < best) { second = fmin(best, wsec); best = wmin; bk = wk; }
      else {
        second = fmin(second, wmin);
        if (wmin == best && bk >= 0 && L.morig[wk] < L.morig[bk]) bk = wk;
      }
    }
    const double limit = fmin(best, thr);
    const bool sc = act && l2 > limit;
    const unsigned long long bu = __ballot(sc && o >= 0 && cr * sgn >= 0.0);
    const unsigned long long bdn = __ballot(sc && o < 0 && cr * sgn <= 0.0);
    if (bu) { up_done = true; l2u = read_lane(l2, 63 - __clzll((long long)bu)); }         // outermost stopper
    if (bdn) { dn_done = true; l2d = read_lane(l2, __ffsll((long long)bdn) - 1); }
    if (hi < lo) { lo = w0; hi = w0 + cnt - 1; }
    else { if (w0 < lo) lo = w0; if (w0 + cnt - 1 > hi) hi = w0 + cnt - 1; }
    const int total = hi - lo + 1;
    if ((up_done && dn_done) || total >= nM) {
      NnResult r;
      r.best = best; r.bk = bk; r.resolved = true;
      r.lbsq = (total >= nM) ? second : fmin(second, fmin(l2u, l2d));
      if (nM <= 1) r.lbsq = __builtin_inf();
      return r;
    }
    const int remaining = nM - total;
    cnt = remaining < 64 ? remaining : 64;
    w0 = !up_done ? hi + 1 : lo - cnt;
  }
}

template <int R, int MAXT>
__global__ void __launch_bounds__(MAXT)
k_icp(IcpArgs a, const double* __restrict__ P_dev, int cap, const double* __restrict__ g_model, const double* __restrict__ g_scene,
      const int* __restrict__ g_morig, const int* __restrict__ g_start,
      const double* __restrict__ g_coords, const uint8_t* __restrict__ g_mask_m,
      const double* __restrict__ g_rays_local, const double* __restrict__ g_ranges,
      const uint8_t* __restrict__ g_mask, IcpResultDev* __restrict__ out,
      double* __restrict__ trace /* [TSD_ICP_TRACE_MAX][4] = pairs, rms, thr_before, state */)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  IcpLds L;
  const int lcap = icp_list_cap(cap);
  {
    char* p = smem;
    L.mxy = reinterpret_cast<double2*>(p); p += sizeof(double2) * (size_t)cap;
    L.uxy = reinterpret_cast<double2*>(p); p += sizeof(double2) * (size_t)cap;
    L.list_xy = reinterpret_cast<double2*>(p); L.stage_s = reinterpret_cast<double2*>(p);
    p += sizeof(double2) * (size_t)lcap;
    L.res_d = reinterpret_cast<double*>(p); p += sizeof(double) * (size_t)lcap;
    L.res_lb = reinterpret_cast<double*>(p); p += sizeof(double) * (size_t)lcap;
    L.list_k = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)lcap;
    L.res_k = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)lcap;
    // staging view of the same 40*lcap bytes: cap double2 then cap int (40*lcap >= 20*cap)
    L.start = reinterpret_cast<int*>(reinterpret_cast<char*>(L.stage_s) + sizeof(double2) * (size_t)cap);
    L.slotD = reinterpret_cast<unsigned long long*>(p); p += sizeof(unsigned long long) * (size_t)cap;
    L.red = reinterpret_cast<double*>(p); p += sizeof(double) * ICP_MAXW * 8;
    L.red2 = reinterpret_cast<double*>(p); p += sizeof(double) * ICP_MAXW * 2;
    L.slotI = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)cap;
    L.morig = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)cap;
    L.ired = reinterpret_cast<int*>(p);
  }

  if (P_dev) {   // fused scan: the pre-registration sensor pose lives on the device
#pragma unroll
    for (int i = 0; i < 6; i++) a.P[i] = P_dev[i];
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = blockDim.x, W = T >> 6;
  int nM = 0, nS = 0;
#ifdef TSD_ICP_STAMPS   // diagnostic build: cycles per phase (thread 0), written behind the trace records
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long st_t = clock64();
#define STAMP(i) do { const long long now_ = clock64(); st_acc[i] += now_ - st_t; st_t = now_; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

  // ---------------------------------------------------------------- inputs
  if (a.beams > 0) {
    // fused mode: maskMatrix compaction of the ray-cast model and of the scan's cartesian points.
    // Model points stay in beam order = angular order about the sensor.
    int baseM = 0, baseS = 0;
    for (int b0 = 0; b0 < a.beams; b0 += T) {
      const int b = b0 + tid;
      bool fm = false, fs = false;
      double r = 0.0;
      if (b < a.beams) {
        fm = g_mask_m[b] != 0;
        r = g_ranges[b];
        fs = !isinf(r) && (g_mask[b] != 0);
      }
      const unsigned long long bm = __ballot(fm), bs = __ballot(fs);
      const unsigned long long lt = (1ull << lane) - 1ull;
      if (lane == 0) { L.ired[wave * 2] = __popcll(bm); L.ired[wave * 2 + 1] = __popcll(bs); }
      __syncthreads();
      int offM = baseM, offS = baseS, totM = 0, totS = 0;
      for (int w = 0; w < W; w++) {
        const int cm = L.ired[w * 2], cs_ = L.ired[w * 2 + 1];
        if (w < wave) { offM += cm; offS += cs_; }
        totM += cm; totS += cs_;
      }
      offM += __popcll(bm & lt); offS += __popcll(bs & lt);
      if (fm && offM < cap) { L.mxy[offM] = make_double2(g_coords[2 * b], g_coords[2 * b + 1]); L.morig[offM] = offM; }
      if (fs && offS < cap) {
        // coords = raysLocal(j,i) * data[i] (Sensor.cpp:176-179)
        L.stage_s[offS] = make_double2(g_rays_local[b] * r, g_rays_local[a.beams + b] * r);
        L.start[offS] = offM;      // model slot of the same beam (or of the next hit beam)
      }
      baseM += totM; baseS += totS;
      __syncthreads();
    }
    nM = baseM; nS = baseS;
  } else {
    nM = a.n_model; nS = a.n_scene;
    if (nM <= cap && nS <= cap) {
      for (int j = tid; j < nM; j += T) { L.mxy[j] = make_double2(g_model[2 * j], g_model[2 * j + 1]); L.morig[j] = g_morig[j]; }
      for (int i = tid; i < nS; i += T) { L.stage_s[i] = make_double2(g_scene[2 * i], g_scene[2 * i + 1]); L.start[i] = g_start[i]; }
    }
    __syncthreads();
  }
  if (tid == 0) { L.ired[IR_CNT] = 0; L.ired[IR_RMAX] = 0; }

  double Tf[6] = {1, 0, 0, 0, 1, 0};   // rows 0,1 of _Tfinal4x4: [r00 r01 tx ; r10 r11 ty]
  double rms = 0.0;
  int pairs = 0, state = TSD_ICP_PROCESSING;
  unsigned int iter = 0;

  if (nM == 0 || nS == 0 || nM > cap || nS > cap) {
    // Icp::iterate early-out (Icp.cpp:467-471); ThreadLocalize never gets here with nM == 0
    if (tid == 0) {
      for (int i = 0; i < 9; i++) out->T[i] = (i % 4 == 0) ? 1.0 : 0.0;
      out->rms = 0.0; out->pairs = 0; out->iterations = 0; out->state = TSD_ICP_NOTMATCHABLE;
      out->n_model = nM; out->n_scene = nS; out->reserved = (nM > cap || nS > cap) ? TSD_E_CAPACITY : 0;
    }
    return;
  }

  // every thread takes its scene points into registers; unit directions of the model
  double sx[R], sy[R], lb[R], r0[R];
  int hint[R];
  bool have[R];
  float rmaxf = 0.f;
#pragma unroll
  for (int q = 0; q < R; q++) {
    const int i = tid + q * T;
    have[q] = i < nS;
    sx[q] = 0.0; sy[q] = 0.0; hint[q] = 0; lb[q] = -1.0; r0[q] = 0.0;
    if (have[q]) {
      const double2 s = L.stage_s[i];
      sx[q] = s.x; sy[q] = s.y;
      // |s| rounded up: fp32 is plenty for a bound
      const float rf = sqrtf((float)(s.x * s.x + s.y * s.y) * 1.000001f) * 1.000001f;
      r0[q] = (double)rf;
      rmaxf = fmaxf(rmaxf, rf);
      int h = L.start[i];
      hint[q] = h < 0 ? 0 : (h >= nM ? nM - 1 : h);
    }
  }
  for (int k = tid; k < nM; k += T) {
    const double2 m = L.mxy[k];
    const double r2 = m.x * m.x + m.y * m.y;
    double2 u = make_double2(0.0, 0.0);
    if (r2 > 0.0) { const double inv = 1.0 / sqrt(r2); u.x = m.x * inv; u.y = m.y * inv; }
    L.uxy[k] = u;
  }
  for (int k = tid; k < cap; k += T) { L.slotD[k] = ~0ull; L.slotI[k] = INT_MAX; }
#ifdef TSD_ICP_STAMPS
  if (tid == 0) for (int i = 0; i < 8; i++) L.ired[IR_DBG + i] = 0;
#endif
  __syncthreads();                     // staging consumed (it aliases the work list); counters zeroed
  if (rmaxf > 0.f) atomicMax(&L.ired[IR_RMAX], __float_as_int(rmaxf));   // positive floats order like ints
  __syncthreads();
  const double scene_rmax = (double)__int_as_float(L.ired[IR_RMAX]);
  STAMP(0);

  // ---------------------------------------------------------------- iterate
  const double sgn = a.ccw ? 1.0 : -1.0;     // +1: slots ascend counter-clockwise
  double thr = a.thr0;                       // DistanceFilter::_distSqr after reset()
  double rms_prev = 10e12;
  unsigned int conv_cnt = 0;
  const unsigned int max_it = (unsigned)a.iterations, conv_need = (unsigned)a.iterations;
  // rows of the pose's rotation block are unit vectors up to rounding: |R_p s| <= pnorm |s| per axis
  const double pnorm = fmax(sqrt(a.P[0] * a.P[0] + a.P[1] * a.P[1]), sqrt(a.P[3] * a.P[3] + a.P[4] * a.P[4])) * (1.0 + 1e-9);

  while (state == TSD_ICP_PROCESSING) {
    const double thr_before = thr;

    // -- phase A: pre-filter + exact NN + distance filter (per scene point)
    // OutOfBoundsFilter2D: when even a disc of the largest possible scene radius around the sensor
    // lies inside the bounds nothing can be filtered and the per-point transform is skipped.
    const double tcum = sqrt(Tf[2] * Tf[2] + Tf[5] * Tf[5]) * (1.0 + 1e-9);
    const double reach = (scene_rmax + tcum) * pnorm * (1.0 + 1e-6) + 1e-6;
    const bool all_in = (a.P[2] - reach > a.min_x) && (a.P[2] + reach < a.max_x) &&
                        (a.P[5] - reach > a.min_y) && (a.P[5] + reach < a.max_y);
    double bd[R]; bool keep[R], need[R];
    int ent[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
      keep[q] = false; need[q] = false; bd[q] = __builtin_inf(); ent[q] = -1;
      if (have[q]) {
        const double x = sx[q], y = sy[q];
        bool pre = true;
        if (!all_in) {
          // S.transform(P): (0 + x*R00) + y*R01, then + t (gsl/Matrix.cpp:403-432)
          double wx = 0.0, wy = 0.0;
          wx += x * a.P[0]; wx += y * a.P[1];
          wy += x * a.P[3]; wy += y * a.P[4];
          wx += a.P[2]; wy += a.P[5];
          pre = !(wx < a.min_x || wx > a.max_x || wy < a.min_y || wy > a.max_y);
        }
        if (pre) {
          const double2 m = L.mxy[hint[q]];
          const double dx = x - m.x, dy = y - m.y;
          const double d = dx * dx + dy * dy;
          const double lbq = lb[q];
          const double lb2 = lbq * lbq;
          bd[q] = d;
          if (lbq > 0.0 && d < lb2) keep[q] = d <= thr;                 // neighbour unchanged
          else if (lbq > 0.0 && d > thr && lb2 > thr) { }               // no pair whoever it is
          else need[q] = true;
        }
      }
    }
    // work list of the points that need a search
#pragma unroll
    for (int q = 0; q < R; q++)
      if (need[q]) ent[q] = atomicAdd(&L.ired[IR_CNT], 1);
    __syncthreads();
    const int n_need = L.ired[IR_CNT];
    for (int base = 0; base < n_need; base += lcap) {       // one pass unless more than lcap points search
      const int n = (n_need - base) < lcap ? (n_need - base) : lcap;
#pragma unroll
      for (int q = 0; q < R; q++)
        if (need[q] && ent[q] >= base && ent[q] < base + lcap) {
          L.list_xy[ent[q] - base] = make_double2(sx[q], sy[q]);
          L.list_k[ent[q] - base] = hint[q];
        }
      __syncthreads();
      for (int e0 = wave * 64; e0 < n; e0 += T) {
        const int e = e0 + lane;
        bool unresolved = false;
        if (e < n) {
          const double2 s = L.list_xy[e];
          const NnResult r = window_search(L, nM, s.x, s.y, L.list_k[e], thr, sgn);
          if (r.resolved) { L.res_d[e] = r.best; L.res_k[e] = r.bk; L.res_lb[e] = sqrt(r.lbsq) * SLACK; }
          else unresolved = true;
        }
        unsigned long long todo = __ballot(unresolved);
#ifdef TSD_ICP_STAMPS
        if (lane == 0) { atomicAdd(&L.ired[IR_DBG + 1], __popcll(todo)); }
#endif
        while (todo) {
          const int src = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const int es = e0 + src;
          const double2 s = L.list_xy[es];
          const NnResult r = wave_search(L, nM, s.x, s.y, L.list_k[es], thr, sgn, lane);
          if (lane == 0) { L.res_d[es] = r.best; L.res_k[es] = r.bk; L.res_lb[es] = sqrt(r.lbsq) * SLACK; }
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < R; q++)
        if (need[q] && ent[q] >= base && ent[q] < base + lcap) {
          const int e = ent[q] - base;
          const int k = L.res_k[e];
          if (k >= 0) { bd[q] = L.res_d[e]; hint[q] = k; lb[q] = L.res_lb[e]; keep[q] = bd[q] <= thr; }   // DistanceFilter::filter
          else { bd[q] = __builtin_inf(); lb[q] = -1.0; }                                                // non-finite input point
        }
      if (base + lcap < n_need) __syncthreads();
    }
#ifdef TSD_ICP_STAMPS
    if (tid == 0) L.ired[IR_DBG] += n_need;
#endif
    // threshold schedule (DistanceFilter.cpp:62-63)
    thr *= a.multiplier;
    if (thr < a.min_sqr) thr = a.min_sqr;
    STAMP(1);

    // -- phase B/C: ReciprocalFilter = per model point keep the pair with the smallest d2
#pragma unroll
    for (int q = 0; q < R; q++)
      if (keep[q]) atomicMin(&L.slotD[hint[q]], (unsigned long long)__double_as_longlong(bd[q]));
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++)
      if (keep[q] && L.slotD[hint[q]] == (unsigned long long)__double_as_longlong(bd[q]))
        atomicMin(&L.slotI[hint[q]], tid + q * T);
    __syncthreads();
    STAMP(2);

    // -- phase D: ClosedFormEstimator2D::setPairs: centroids, "rms" (mean squared distance), count
    double v[5] = {0, 0, 0, 0, 0};
    double wmx[R], wmy[R];
    bool win[R];
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < R; q++) {
      win[q] = keep[q] && L.slotI[hint[q]] == tid + q * T;
      wmx[q] = 0.0; wmy[q] = 0.0;
      if (win[q]) {
        const double2 m = L.mxy[hint[q]];
        wmx[q] = m.x; wmy[q] = m.y;
        v[0] += m.x; v[1] += m.y; v[2] += sx[q]; v[3] += sy[q];
        const double dx = sx[q] - m.x, dy = sy[q] - m.y;
        v[4] += dx * dx + dy * dy;
      }
      cnt += __popcll(__ballot(win[q]));
    }
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = wave_total(v[k]);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) L.red[wave * 8 + k] = v[k];
      L.ired[wave] = cnt;
    }
    __syncthreads();
    // everybody is past the winner test: give the touched slots back, clear the work list counter
#pragma unroll
    for (int q = 0; q < R; q++)
      if (keep[q]) { L.slotD[hint[q]] = ~0ull; L.slotI[hint[q]] = INT_MAX; }
    if (tid == 0) L.ired[IR_CNT] = 0;
    {
      double t[5] = {0, 0, 0, 0, 0};
      int c = 0;
      for (int w = 0; w < W; w++) {
#pragma unroll
        for (int k = 0; k < 5; k++) t[k] += L.red[w * 8 + k];
        c += L.ired[w];
      }
#pragma unroll
      for (int k = 0; k < 5; k++) v[k] = t[k];
      pairs = c;
    }
    STAMP(3);

    if (pairs > 2) {
      const double size_inv = 1.0 / (double)pairs;
      rms = v[4] * size_inv;
      const double cmx = v[0] * size_inv, cmy = v[1] * size_inv, csx = v[2] * size_inv, csy = v[3] * size_inv;
      // -- phase F: estimateTransformation: nominator / denominator over centred pairs
      double nom = 0.0, den = 0.0;
#pragma unroll
      for (int q = 0; q < R; q++) {
        if (win[q]) {
          const double xF = wmx[q] - cmx, yF = wmy[q] - cmy;
          const double xS = sx[q] - csx, yS = sy[q] - csy;
          nom += yF * xS - xF * yS;
          den += xF * xS + yF * yS;
        }
      }
      nom = wave_total(nom); den = wave_total(den);
      if (lane == 0) { L.red2[wave * 2] = nom; L.red2[wave * 2 + 1] = den; }
      __syncthreads();
      nom = 0.0; den = 0.0;
      for (int w = 0; w < W; w++) { nom += L.red2[w * 2]; den += L.red2[w * 2 + 1]; }
      // every wave evaluates the closed form itself (wave-uniform inputs): no broadcast barrier
      double co, si;
#ifdef TSD_ICP_EXACT_TRIG
      { const double th_ = atan2(nom, den); co = cos(th_); si = sin(th_); }
#else
      // cos(atan2(n, d)) = d / hypot, sin = n / hypot: same angle without three libm calls; differs
      // from the reference's atan2 -> cos/sin by rounding only (DESIGN.md "ICP", tolerance 1e-4)
      {
        const double h2 = nom * nom + den * den;
        if (h2 > 0.0) { const double inv = 1.0 / sqrt(h2); co = den * inv; si = nom * inv; }
        else { co = signbit(den) ? -1.0 : 1.0; si = 0.0; }
      }
#endif
      const double dX = (cmx - (co * csx - si * csy));
      const double dY = (cmy - (co * csy + si * csx));
      STAMP(4);
      // How far can this step move a point?  |R s - s| = chord * |s| and |s| <= r0 + |t_cum|, so
      // disp <= chord * (r0 + tcum) + |t_step|: it eats into the neighbour bounds of tier 0.
      const float chordf = sqrtf((float)((1.0 - co) * (1.0 - co) + si * si) * 1.000001f) * 1.000001f;
      const float tstepf = sqrtf((float)(dX * dX + dY * dY) * 1.000001f) * 1.000001f;
      const double chord = (double)chordf;
      const double dfix = chord * tcum + (double)tstepf;
      // applyTransformation(sceneTmp): data * R^T (dgemm NoTrans,Trans), then + t (Icp.cpp:371-408)
#pragma unroll
      for (int q = 0; q < R; q++) {
        if (have[q]) {
          const double x = sx[q], y = sy[q];
          double nx = 0.0, ny = 0.0;
          nx += x * co; nx += y * (-si);
          ny += x * si; ny += y * co;
          sx[q] = nx + dX; sy[q] = ny + dY;
          lb[q] = lb[q] - (chord * r0[q] + dfix);
        }
      }
      {
        // Tfinal = Tlast * Tfinal (Icp.cpp:452): the 4x4 product restricted to its non-trivial entries
        // (the dropped terms are exact zeros / ones, so the rounding is the dgemm's)
        double n00 = 0.0, n01 = 0.0, n02 = 0.0, n10 = 0.0, n11 = 0.0, n12 = 0.0;
        n00 += co * Tf[0]; n00 += (-si) * Tf[3];
        n01 += co * Tf[1]; n01 += (-si) * Tf[4];
        n02 += co * Tf[2]; n02 += (-si) * Tf[5]; n02 += 0.0; n02 += dX * 1.0;
        n10 += si * Tf[0]; n10 += co * Tf[3];
        n11 += si * Tf[1]; n11 += co * Tf[4];
        n12 += si * Tf[2]; n12 += co * Tf[5]; n12 += 0.0; n12 += dY * 1.0;
        Tf[0] = n00; Tf[1] = n01; Tf[2] = n02; Tf[3] = n10; Tf[4] = n11; Tf[5] = n12;
      }
      state = TSD_ICP_PROCESSING;
    } else {
      state = TSD_ICP_NOTMATCHABLE;
    }
    // -- loop control (Icp.cpp:489-511)
    iter++;
    if (fabs(rms - rms_prev) < 10e-10) conv_cnt++; else conv_cnt = 0;
    if (rms <= 0.0 || conv_cnt >= conv_need) state = TSD_ICP_SUCCESS;
    else if (iter >= max_it) state = TSD_ICP_MAXITERATIONS;
    rms_prev = rms;
    STAMP(5);
    if (tid == 0 && iter <= TSD_ICP_TRACE_MAX) {
      double* tr = trace + 4 * (iter - 1);
      tr[0] = (double)pairs; tr[1] = rms; tr[2] = thr_before; tr[3] = (double)state;
#ifdef TSD_ICP_STAMPS
      tr[2] = (double)st_acc[1];        // cumulative phase-A cycles      (diagnostic build only)
      tr[1] = (double)L.ired[IR_DBG];   // cumulative searched points
#endif
    }
  }

#ifdef TSD_ICP_STAMPS
  __syncthreads();
  if (tid == 0) {
    st_acc[6] = L.ired[IR_DBG]; st_acc[7] = L.ired[IR_DBG + 1];     // searched points / wave searches
    for (int i = 0; i < 8; i++) trace[4 * TSD_ICP_TRACE_MAX - 8 + i] = (double)st_acc[i];
  }
#endif
  if (tid == 0) {
    // Icp::getFinalTransformation (Icp.cpp:528-546)
    out->T[0] = Tf[0]; out->T[1] = Tf[1]; out->T[2] = Tf[2];
    out->T[3] = Tf[3]; out->T[4] = Tf[4]; out->T[5] = Tf[5];
    out->T[6] = 0.0; out->T[7] = 0.0; out->T[8] = 1.0;
    out->rms = rms; out->pairs = pairs; out->iterations = (int)iter; out->state = state;
    out->n_model = nM; out->n_scene = nS; out->reserved = 0;
  }
}

static int icp_cap_for(int n)
{
  int cap = (n + 63) & ~63;
  if (cap < 64) cap = 64;
  return cap;
}

// workgroup shape: R scene points per thread, T threads.  One CU runs the whole registration and is
// issue bound, so few waves (per-wave reduction / control cost paid once per SIMD) win.
template <int R, int MAXT>
static int launch_icp_shape(tsd_ctx* ctx, const IcpArgs& a, int n, int cap, size_t lds, const double* P_dev,
                            const double* d_rays_local, const double* d_ranges, const uint8_t* d_mask)
{
  int T = ((n + R - 1) / R + 63) & ~63;
  if (T < 64) T = 64;
  if (T > MAXT) return set_error(ctx, TSD_E_CAPACITY, "icp workgroup shape", hipSuccess);
  static size_t configured = 0;
  if (lds > configured) {
    TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_icp<R, MAXT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = lds;
  }
  ScopedKernelTimer t(ctx, "icp");
  hipLaunchKernelGGL((k_icp<R, MAXT>), dim3(1), dim3(T), lds, ctx->stream, a, P_dev, cap, ctx->d_model, ctx->d_scene,
                     ctx->d_morig, ctx->d_start, ctx->d_coords, ctx->d_mask_m,
                     d_rays_local ? d_rays_local : ctx->d_rays_local, d_ranges ? d_ranges : ctx->d_ranges,
                     d_mask ? d_mask : ctx->d_mask, ctx->d_icp_res, ctx->d_icp_trace);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_icp(tsd_ctx* ctx, const IcpArgs& a, const double* P_dev, const double* d_rays_local,
               const double* d_ranges, const uint8_t* d_mask)
{
  const int n = a.beams > 0 ? a.beams : (a.n_model > a.n_scene ? a.n_model : a.n_scene);
  if (n > TSD_MAX_ICP_POINTS) return set_error(ctx, TSD_E_CAPACITY, "icp points > TSD_MAX_ICP_POINTS", hipSuccess);
  const int cap = icp_cap_for(n);
  const size_t lds = icp_lds_bytes_for(cap);
  const int nthr = a.beams > 0 ? a.beams : a.n_scene;     // scene points decide the thread count
  switch (ctx->icp_shape) {      // TSD_ICP_SHAPE: experiments only
    case 2: return launch_icp_shape<2, 576>(ctx, a, nthr, cap, lds, P_dev, d_rays_local, d_ranges, d_mask);
    case 5: return launch_icp_shape<5, 256>(ctx, a, nthr, cap, lds, P_dev, d_rays_local, d_ranges, d_mask);
    case 8: return launch_icp_shape<8, 256>(ctx, a, nthr, cap, lds, P_dev, d_rays_local, d_ranges, d_mask);
    default: break;
  }
  if (nthr <= 3 * 512) return launch_icp_shape<3, 512>(ctx, a, nthr, cap, lds, P_dev, d_rays_local, d_ranges, d_mask);
  return launch_icp_shape<8, 256>(ctx, a, nthr, cap, lds, P_dev, d_rays_local, d_ranges, d_mask);
}

size_t icp_lds_bytes() { return icp_lds_bytes_for(TSD_MAX_ICP_POINTS); }

}  // namespace tsd
