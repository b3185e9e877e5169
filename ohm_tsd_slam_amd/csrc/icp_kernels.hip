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
#include "scan_device.hpp"
#include "tsdpdf_device.hpp"
#include <atomic>
#include <climits>
#include <cstring>

// Likelihood hints decide the LAYOUT of the step loop: a region the compiler thinks likely loses its skip branch and sits in line, an
// unlikely one goes out of line -- the steady step then takes one taken branch (the loop's) instead of ten (tools/exp/ctl.hip: 24-38
// cycles each; profiles/r6_icp_branch_cost_ab.txt).
#define LIKELY(x) __builtin_expect(!!(x), 1)
#define UNLIKELY(x) __builtin_expect(!!(x), 0)
namespace tsd {

constexpr int ICP_MAXW = 16;                    // waves per workgroup at most
constexpr double SLACK = 1.0 - 1e-9;            // conservative factor on every pruning bound
constexpr int HW = 6;                           // tier-1 window: k-HW .. k+HW (3 / 4 / 8 measured slower)
constexpr int ICP_PAD = 96;                     // wrapped copies of the model at both ends of its LDS array
constexpr unsigned REFRESH_A = 6, REFRESH_B = 15;      // steps with a scheduled bound renewal
constexpr double WEAK_MULT = 36.0;                     // a scheduled renewal takes the bounds with less than sqrt(this) x slack in distance
constexpr int IR_CNT = 32, IR_RMAX = 33, IR_CNT2 = 34, IR_TIE = 35, IR_SEEDED = 36;   // words of IcpLds::ired (SEEDED: points whose step-0 neighbour a helper delivered)
// Step 0 of a registration is the one step in which EVERY scene point searches (no neighbour is known yet): ~20 000 cycles on the one
// compute unit that runs the registration.  That search does not depend on anything the loop produces, so it is shared out: the launch
// brings `helpers` more workgroups, each of which runs the same set-up (model, unit directions, padding in ITS OWN LDS) and then the
// tier-1 window search of step 0 for ICP_HELPER_POINTS of the scene points -- the same function on the same inputs as the registration's
// own list pass would run.  A helper lane hands its result over as two 8-byte GRANULES {launch number, value}: (1) the bits of the
// fp32 square root the bound is formed from, (2) neighbour slot | runner-up slot << 16 (0xFFFF: the window could not prove the point),
// each ONE relaxed agent-scope atomic store -- a write-through store that carries its own tag, so there is no flag, no fence and no
// barrier on either side (MI355X hand-off recipe R2).  The registering workgroup finishes its own set-up meanwhile; every lane then
// re-reads ITS points' granules until their tags are this launch's and starts step 0 from neighbour, runner-up and bound -- tier 0
// confirms them like any other step's (the distance is recomputed from the same coordinates by the same expression, the bound is
// rebuilt from the same fp32 root: the state after step 0 is bit for bit what the workgroup's own search would have left).  The wait
// is bounded and per wave: points whose granules did not arrive in time (helpers that got no compute unit: a push on another stream
// filling the device) simply search in step 0 as they always did; the results are the same either way, only the time differs.
constexpr int ICP_HELPER_POINTS = 256;              // scene points per helper workgroup: one per lane of its waves 0-3 (one wave per SIMD)
constexpr int ICP_MAX_HELPERS = 16;
constexpr long long ICP_SEED_WAIT_TICKS = 2500;     // of the 100 MHz wall clock: 25 us

// what the kernel needs again only after the last step (and the trace pointer, once per step by one
// thread): parked in LDS so that it does not sit in scalar registers through the loop
struct IcpTail { IcpResultDev* out; double* trace; ScanPostArgs post; ScanPostPre pre; };

struct IcpLds {
  IcpTail* tail;
  double2* mxy;                    // model (angular order)
  double2* uxy;                    // unit direction of every model point (0,0 for a point at the origin)
  double2* nxy;                    // model normals (point-to-line estimator only; nullptr otherwise)
  unsigned long long* slotD;       // [2][cap] reciprocal filter: min d2 (bit pattern) per model slot, the two halves used by alternate steps
  int* slotI;                      // [cap] winning scene index per model slot
  int* morig;                      // original model index of a slot (tie-breaking)
  double2* list_xy;                // [lcap] work list: point
  int* list_k;                     // [lcap]            its last neighbour slot
  double* res_d;                   // [lcap] results: squared distance to the nearest neighbour
  double* res_lb;                  // [lcap]          lower bound (distance) to every other model point
  int* res_k;                      // [lcap]          neighbour slot (-1: unresolved)
  int* res_k2;                     // [lcap]          runner-up slot
  int* list2;                      // [lcap] entries the window could not prove (tier 2 work list)
  double* red;                     // [2][ICP_MAXW][16] wave partials of the pair sums, per-wave broadcast rows
  double* cst;                     // [16] IcpArgs scalars (kept out of the scalar register file)
  double* tr;                      // [T][NSUMP] transpose buffer of the pair sums (aliases the work list)
  int* ired;                       // [64] counters
  // setup only (alias the work list)
  double2* stage_s;                // compacted scene
  int* start;                      // first search position of every compacted scene point
};

// (up to the node's shape -- 1081 beams, capacity 1088 -- the list holds every point: one pass, always)
__host__ __device__ inline int icp_list_cap(int cap) { return cap <= 1088 ? cap : 1024; }
// bytes of the region shared by the work list (48 B per entry), the setup staging and the transpose buffer
__host__ __device__ inline size_t icp_region_bytes(int cap, int threads)
{
  size_t b = 48u * (size_t)icp_list_cap(cap);
  const size_t tr = (size_t)threads * 11u * sizeof(double);      // nsum_pitch(NSUM_PTL): the wider of the two
  if (tr > b) b = tr;
  return (b + 15u) & ~(size_t)15u;
}
// a half of the reciprocal filter's slot array: a whole number of entries per thread, so that giving a half back is the same stores in every
// thread (no bound to test: a test is a branch)
__host__ __device__ inline int icp_slot_cap(int cap, int threads) { return (cap + threads - 1) / threads * threads; }
__host__ __device__ inline size_t icp_lds_base_bytes(int cap, int threads, bool normals)      // with ONE half of the slot array
{
  // the staging (cap double2 + cap int) aliases the list + result arrays: 40 * lc >= 20 * cap
  return sizeof(double2) * 2 * (size_t)cap + sizeof(double2) * 2 * ICP_PAD + sizeof(unsigned long long) * (size_t)icp_slot_cap(cap, threads) + sizeof(int) * 2 * (size_t)cap +
         icp_region_bytes(cap, threads) + sizeof(double) * (2 * ICP_MAXW * 16 + 16) + ((sizeof(IcpTail) + 15) & ~(size_t)15) +
         sizeof(int) * 64 + 64 + (normals ? sizeof(double2) * (size_t)cap : 0);
}
#ifdef TSD_ICP_TIMELINE
#ifndef TSD_ICP_TL_FIRST
#define TSD_ICP_TL_FIRST 19
#define TSD_ICP_TL_STEPS 4
#endif
constexpr size_t ICP_TL_BYTES = (size_t)TSD_ICP_TL_STEPS * 8 * 16 * sizeof(long long);     // the timeline build's stamp buffer behind the kernel's LDS (<= 8 waves)
#else
constexpr size_t ICP_TL_BYTES = 0;
#endif
// The reciprocal filter's slot array has two halves used by alternate steps wherever the CU's 160 KB hold them (every shape the node
// runs; not the largest point counts of tsd_icp with the point-to-line estimator's normals): see the loop.  Same rule on both sides.
__host__ __device__ inline int icp_slot_halves(int cap, int threads, bool normals)
{
  return icp_lds_base_bytes(cap, threads, normals) + sizeof(unsigned long long) * (size_t)icp_slot_cap(cap, threads) + ICP_TL_BYTES <= 160u * 1024u ? 2 : 1;
}
__host__ __device__ inline size_t icp_lds_bytes_for(int cap, int threads, bool normals = false)
{
  return icp_lds_base_bytes(cap, threads, normals) + sizeof(unsigned long long) * (size_t)icp_slot_cap(cap, threads) * (size_t)(icp_slot_halves(cap, threads, normals) - 1);
}

// a wave-uniform value the compiler must keep in a vector register
__device__ __forceinline__ double vreg(double x) { asm volatile("" : "+v"(x)); return x; }

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

// best = squared distance to the nearest slot bk, bk2 = runner-up slot, lbsq = squared lower bound on the
// distance to every slot other than those two
struct NnResult { double best, lbsq; int bk, bk2; bool resolved; };

// lower bound (distance) from its square: fp32 square root rounded down by more than its error
__device__ __forceinline__ double lb_from_sq(double lbsq)
{
  return (double)__builtin_amdgcn_sqrtf((float)lbsq) * (1.0 - 1e-6);
}

// running three smallest squared distances (the two smallest with their slots), the whole-wave walk's.  EXACT ties are ordered by the
// original model index (the first-minimum rule of a linear scan): a candidate at the SAME distance as a kept one goes in front of it
// when its original index is lower -- three and more model points at exactly the same distance from a scene point (a scene point at the
// centre of a lattice cell: tools/fuzz_icp.py) arrive in different 64-slot windows of the walk, and comparing distances alone kept
// whichever came first.
struct Top3 { double b1, b2, b3; int k1, k2; };
__device__ __forceinline__ bool top3_before(const IcpLds& L, double d, int k, double dk, int kk)      // (d, k) ahead of the kept (dk, kk)?
{
  return d < dk || (d == dk && kk >= 0 && k >= 0 && L.morig[k] < L.morig[kk]);
}
__device__ __forceinline__ void top3_insert(const IcpLds& L, Top3& t, double d, int k)
{
  if (top3_before(L, d, k, t.b1, t.k1)) { t.b3 = t.b2; t.b2 = t.b1; t.k2 = t.k1; t.b1 = d; t.k1 = k; }
  else if (top3_before(L, d, k, t.b2, t.k2)) { t.b3 = t.b2; t.b2 = d; t.k2 = k; }
  else if (d < t.b3) t.b3 = d;
}
// exact tie between the two nearest: the lower original model index is the neighbour
__device__ __forceinline__ void top3_tiebreak(const IcpLds& L, Top3& t)
{
  if (t.k2 >= 0 && t.k1 >= 0 && t.b2 == t.b1 && L.morig[t.k2] < L.morig[t.k1]) { const int k = t.k1; t.k1 = t.k2; t.k2 = k; }
}

// separation bound of model slot with unit direction u for the point (x, y): squared lower bound on the
// distance from (x, y) to ANY point at that angular separation or more; `cr` returns the side.
__device__ __forceinline__ double sep_bound(double x, double y, double rs2, double2 u, double& cr)
{
  cr = x * u.y - y * u.x;
  const double dt = x * u.x + y * u.y;
  const bool valid_u = (u.x != 0.0) || (u.y != 0.0);
  return valid_u ? (dt > 0.0 ? cr * cr : rs2) * SLACK : 0.0;
}

// tier 1: the 2*HW+1 slots around `c`, one lane per point, all LDS reads of a round issued together.  When
// the two ends cannot bound what lies outside, the arc grows by another 2*HW+1 slots on its weaker side (the
// neighbour of a scan that turned against the prediction sits the same number of slots away for every
// point), up to WIN_ROUNDS times; only then the point goes to the whole-wave search.
//
// Instruction diet (this is the hot loop of the first steps): the model array carries ICP_PAD wrapped
// copies at both ends, so a round is 13 reads at constant offsets from one address; the running three
// smallest distances carry their slot offset in the low 8 mantissa bits, so the insertion is five
// v_min/v_max_f64 and no index bookkeeping.  Packed order equals true order unless the upper 56 bits
// agree; the two nearest are therefore re-evaluated exactly at the end (ties: lower original index) and
// a third candidate in the same 256-ulp bucket sends the point to the exact whole-wave search.
constexpr int WIN = 2 * HW + 1;
constexpr int WIN_ROUNDS = 6;
constexpr double WIN_REACH = 0.03;               // sin^2 of ~10 degrees: about the widest arc the window grows to at 0.25 degree per slot
constexpr int LIST_PAST_WINDOW = 1 << 30;         // work-list entry: the tier-1 window was already tried
constexpr double LB_WINDOW_TRIED = -2.0;          // a point's bound: unknown (<= 0), and a helper's window search has already failed for it
static_assert(ICP_PAD >= HW + WIN_ROUNDS * WIN, "padding must cover the widest window");
__device__ __forceinline__ int wrap_slot(int k, int nM)
{
  k += (k < 0) ? nM : 0;
  k -= (k >= nM) ? nM : 0;
  return k;
}
// v_min_f64 / v_max_f64 as they are: fmin() / fmax() put a canonicalising v_max in front of every operand the compiler cannot prove
// quiet, i.e. of every packed distance (13 extra instructions per window round).  A NaN operand loses against a number either way.
__device__ __forceinline__ double min_raw(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double max_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double pack_code(double d, int code)
{
  return __hiloint2double(__double2hiint(d), (__double2loint(d) & ~0xFF) | code);
}
__device__ __forceinline__ NnResult window_search(const IcpLds& L, int nM, double x, double y, int c,
                                                  double thr, double sgn, int* rounds_out = nullptr, long long* tl = nullptr)
{
  NnResult r;
  r.best = __builtin_inf(); r.lbsq = 0.0; r.bk = -1; r.bk2 = -1; r.resolved = false;
  if (nM <= WIN) return r;
  double b1 = __builtin_inf(), b2 = __builtin_inf(), b3 = __builtin_inf();   // packed (offset + 128 in the low byte)
  const double rs2 = x * x + y * y;
  int lo = -HW, hi = HW;                        // visited arc, offsets relative to c
  double blo = 0.0, bhi = 0.0;                  // bounds on everything below lo / above hi
  int b = lo;                                   // first offset of the round
  int side = 0;                                 // 0: first round (both ends), -1: grew downwards, +1: upwards
  bool proven = false;
  for (int round = 0;; round++) {
    const double2* base = L.mxy + (c + b);      // padded: c + b + j stays inside [-ICP_PAD, nM + ICP_PAD)
    double2 m[WIN];
#pragma unroll
    for (int j = 0; j < WIN; j++) m[j] = base[j];
    const double2 ulo = L.uxy[wrap_slot(c + b, nM)], uhi = L.uxy[wrap_slot(c + b + WIN - 1, nM)];
#ifdef TSD_ICP_TIMELINE
    if (tl && round == 0) { double t0 = m[0].x + m[WIN - 1].x + ulo.x + uhi.x; asm volatile("" : "+v"(t0)); tl[0] = clock64(); }
#endif
    const int code0 = b + 128;
#pragma unroll
    for (int j = 0; j < WIN; j++) {
      const double dx = x - m[j].x, dy = y - m[j].y;
      // (a filter value: fused, i.e. within 3 ulp of what the exact re-evaluation below computes -- the bucket test there allows for it)
      const double d = pack_code(__builtin_fma(dy, dy, dx * dx), code0 + j);
      const double h1 = max_raw(b1, d); b1 = min_raw(b1, d);
      const double h2 = max_raw(b2, h1); b2 = min_raw(b2, h1);
      b3 = min_raw(b3, h2);
    }
    // the low end must lie clockwise of s (in slot order) and the high end counter-clockwise
    double cr;
    if (side <= 0) { const double l2 = sep_bound(x, y, rs2, ulo, cr); blo = cr * sgn <= 0.0 ? l2 : 0.0; }
    if (side >= 0) { const double l2 = sep_bound(x, y, rs2, uhi, cr); bhi = cr * sgn >= 0.0 ? l2 : 0.0; }
    // (b1 packed differs from the true distance by < 256 ulp; SLACK in the bounds covers that)
    if (fmin(blo, bhi) > fmin(b1, thr)) { proven = true; break; }
    if (round == WIN_ROUNDS || hi - lo + 1 + WIN > nM) break;
    // the arc can grow to +-(HW + WIN_ROUNDS * WIN) slots; what needs a wider separation than a scan's beams
    // have over that many slots goes to tier 2 right away (something far from the whole model)
    if (fmin(b1, thr) > WIN_REACH * rs2) break;
    if (blo <= bhi) { side = -1; lo -= WIN; b = lo; }
    else { side = 1; b = hi + 1; hi += WIN; }
  }
  if (!(b1 < __builtin_inf())) return r;        // no finite distance at all (non-finite point): tier 2 sorts it out
#ifdef TSD_ICP_TIMELINE
  if (tl) { asm volatile("" : "+v"(b1), "+v"(b2), "+v"(b3), "+v"(blo), "+v"(bhi)); tl[1] = clock64(); }
  if (rounds_out) *rounds_out = lo == -HW && hi == HW ? 1 : 1 + (hi - lo + 1 - WIN) / WIN;
#endif
  // unpack the two nearest and evaluate them exactly
  const int o1 = (__double2loint(b1) & 0xFF) - 128, o2 = (__double2loint(b2) & 0xFF) - 128;
  int k1 = wrap_slot(c + o1, nM), k2 = wrap_slot(c + o2, nM);
  const double2 m1 = L.mxy[k1], m2 = L.mxy[k2];
  double d1, d2;
  { const double dx = x - m1.x, dy = y - m1.y; d1 = dx * dx + dy * dy; }
  { const double dx = x - m2.x, dy = y - m2.y; d2 = dx * dx + dy * dy; }
  if (d2 < d1 || (d2 == d1 && L.morig[k2] < L.morig[k1])) { const int t = k1; k1 = k2; k2 = t; d1 = d2; }
  // third candidate in the same 256-ulp bucket as the nearest or in the next one: order unknown here (the filter values are fused
  // multiply-adds, up to 3 ulp from the unfused distances that decide: two buckets apart, the order of the unfused values is the same)
  const double b3c = pack_code(b3, 0);
  const double b1n = __longlong_as_double(__double_as_longlong(pack_code(b1, 0)) + 0x100ll);      // start of the bucket after the nearest's
  const bool crowded = !(b3c > b1n);
  r.best = d1; r.bk = k1; r.bk2 = k2;
  r.lbsq = fmin(b3c, fmin(blo, bhi));
  r.resolved = proven && !crowded && !isnan(d1);
  return r;
}

// Every slot of the model for one point, by the whole wave: each lane keeps the three smallest distances of its
// slots (slot number in the low 11 mantissa bits, five v_min/max_f64 per slot), three wave minima merge them.
// The two nearest are re-evaluated exactly (ties: lower original index); false if the third shares their
// 2048-ulp bucket (the caller then finishes the exact walk).  lbsq = the third smallest distance: a bound on
// every other slot.
__device__ __forceinline__ bool sweep_all(const IcpLds& L, int nM, double x, double y, int lane, NnResult& r)
{
  const double inf = __builtin_inf();
  double b1 = inf, b2 = inf, b3 = inf;
  for (int k = lane; k < nM; k += 64) {
    const double2 m = L.mxy[k];
    const double dx = x - m.x, dy = y - m.y;
    const double dd = __builtin_fma(dy, dy, dx * dx);          // (a filter value, like window_search's)
    const double d = __hiloint2double(__double2hiint(dd), (__double2loint(dd) & ~0x7FF) | k);
    const double h1 = max_raw(b1, d); b1 = min_raw(b1, d);
    const double h2 = max_raw(b2, h1); b2 = min_raw(b2, h1);
    b3 = min_raw(b3, h2);
  }
  double g[3];
#pragma unroll
  for (int rnk = 0; rnk < 3; rnk++) {
    const double wmin = wave_min(b1);
    g[rnk] = wmin;
    if (b1 == wmin && wmin < inf) { b1 = b2; b2 = b3; b3 = inf; }       // (packed values are unique: one lane pops)
  }
  if (!(g[0] < inf)) return false;
  int k1 = __double2loint(g[0]) & 0x7FF, k2 = g[1] < inf ? (__double2loint(g[1]) & 0x7FF) : k1;
  const double2 m1 = L.mxy[k1], m2 = L.mxy[k2];
  double d1, d2;
  { const double dx = x - m1.x, dy = y - m1.y; d1 = dx * dx + dy * dy; }
  { const double dx = x - m2.x, dy = y - m2.y; d2 = dx * dx + dy * dy; }
  if (d2 < d1 || (d2 == d1 && L.morig[k2] < L.morig[k1])) { const int t = k1; k1 = k2; k2 = t; d1 = d2; }
  const double c0 = __hiloint2double(__double2hiint(g[0]), __double2loint(g[0]) & ~0x7FF);
  const double c2 = g[2] < inf ? __hiloint2double(__double2hiint(g[2]), __double2loint(g[2]) & ~0x7FF) : inf;
  if (!(c2 > __longlong_as_double(__double_as_longlong(c0) + 0x800ll))) return false;      // (same or next bucket: the exact walk decides)
  r.best = d1; r.bk = k1; r.bk2 = k2; r.lbsq = c2; r.resolved = true;
  return !isnan(d1);
}

// tier 2: the same proof by the whole wave for one point (x, y, start wave-uniform): 64 consecutive
// slots per step, widened towards the side that is not yet bounded.
__device__ __forceinline__ NnResult wave_search(const IcpLds& L, int nM, double x, double y, int start,
                                                double thr, double sgn, int lane)
{
  const double rs2 = x * x + y * y;
  Top3 t;
  t.b1 = t.b2 = t.b3 = __builtin_inf(); t.k1 = t.k2 = -1;
  double l2u = __builtin_inf(), l2d = __builtin_inf();
  bool up_done = false, dn_done = false;
  int lo = 0, hi = -1;                               // visited offsets relative to `start` (empty)
  int cnt = nM < 64 ? nM : 64;
  int w0 = -(cnt / 2);
  // What tier 1 could not prove within +-84 slots of the hint is far from the model (something the map does not
  // hold yet).  One step of the walk below costs ~700 instructions (three ranked wave minima) and such a point
  // needs several, while sweeping ALL slots costs ~400: sweep first, walk only if the sweep cannot rank.
  if (nM <= (1 << 11)) {
    NnResult r;
    if (sweep_all(L, nM, x, y, lane, r)) return r;
    // (three candidates in one 2048-ulp bucket, or a non-finite point: the exact walk sorts it out)
  }
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
    // the three smallest of this window (exact ties: lowest original index first), merged into the run.
    // While nothing lies within the filter distance only the bound matters: one rank, entered three times.
    double dd = d;
    const bool far_so_far = t.b1 > thr;
#pragma unroll
    for (int rnk = 0; rnk < 3; rnk++) {
      const double wmin = wave_min(dd);
      const unsigned long long eq = __ballot(act && dd == wmin);
      if (!eq) break;
      int wl = __ffsll((long long)eq) - 1;
      if (__popcll(eq) > 1) {
        int bo = INT_MAX;
        unsigned long long e = eq;
        while (e) {
          const int l = __ffsll((long long)e) - 1; e &= e - 1;
          const int mo = L.morig[__builtin_amdgcn_readlane(k, l)];
          if (mo < bo) { bo = mo; wl = l; }
        }
      }
      const int wk = __builtin_amdgcn_readlane(k, wl);
      top3_insert(L, t, wmin, wk);
      if (rnk == 0 && far_so_far && wmin > thr) { top3_insert(L, t, wmin, wk); top3_insert(L, t, wmin, wk); break; }
      if (lane == wl) dd = __builtin_inf();
    }
    top3_tiebreak(L, t);
    // the walk runs on until the bound exceeds 4x what exactness needs: the extra slots (evaluated 64 at a
    // time anyway) buy a bound that survives the following steps
    // (a point with nothing within the filter distance only has to stay dropped: 1.5x is plenty)
    const double limit = (t.b1 > thr ? 1.5 : 4.0) * fmin(t.b1, thr);
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
      r.best = t.b1; r.bk = t.k1; r.bk2 = t.k2 >= 0 ? t.k2 : t.k1; r.resolved = true;
      r.lbsq = (total >= nM) ? t.b3 : fmin(t.b3, fmin(l2u, l2d));
      return r;
    }
    const int remaining = nM - total;
    cnt = remaining < 64 ? remaining : 64;
    w0 = !up_done ? hi + 1 : lo - cnt;
  }
}


// pair sums of one step.  Closed form: sum mx, my, sx, sy, d2 and the four centred products (9; the estimator's two sums kept as
// four: accumulating nominator and denominator per pair, 7 sums, measured no faster); point to line:
// the six entries of A, the three of b, sum |n.(s - m)| (10).  The row pitch of the transpose buffer is the
// next odd number (conflict-free columns).
constexpr int NSUM_CF = 9, NSUM_PTL = 10;
__host__ __device__ constexpr int nsum_pitch(int ns) { return ns | 1; }

// Sums over the whole workgroup of NSUM doubles per thread (+ one wave-uniform integer per wave).
// Every lane writes its row into an LDS transpose buffer; lane l of a wave then adds 16 rows of column
// l/4 (4 lanes per column), two DPP shifts finish the column, the wave partials meet in LDS.  ~80
// instructions for nine values where nine shuffle trees cost ~200; fixed order => deterministic.
template <int MAXW, int NSUM>
__device__ __forceinline__ void block_totals(const IcpLds& L, const double (&v)[NSUM], int cnt, double (&tot)[NSUM],
                                             int& cnt_total, int tid, int lane, int wave, int W, long long* tl = nullptr)
{
  constexpr int NSUMP = nsum_pitch(NSUM);
  double* row = L.tr + (size_t)tid * NSUMP;
#pragma unroll
  for (int k = 0; k < NSUM; k++) row[k] = v[k];
  // (same wave wrote the rows it reads: LDS executes a wave's accesses in order)
  const int col = lane >> 2, part = lane & 3;
  double acc = 0.0;
  if (col < NSUM) {
    const double* base = L.tr + ((size_t)wave * 64 + part) * NSUMP + col;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += base[(size_t)(4 * i) * NSUMP];
  }
  acc += dpp_shr0<0x111>(acc);     // row_shr:1
  acc += dpp_shr0<0x112>(acc);     // row_shr:2  -> lane 4*col + 3 holds the column total of this wave
  if (part == 3 && col < NSUM) L.red[wave * 16 + col] = acc;
  if (lane == 0) L.red[wave * 16 + NSUM] = (double)cnt;      // the pair count rides along (exact in fp64)
  if (tl && lane == 0) { asm volatile("" : "+v"(acc)); tl[7] = clock64(); }       // (timeline build) the wave's partial sums are on their way
  __syncthreads();
  if (tl && lane == 0) tl[8] = clock64();                                        // past barrier 2
  double t = 0.0;
  if (lane <= NSUM) {
    double x[MAXW];
#pragma unroll
    for (int w = 0; w < MAXW; w++) x[w] = (w < W) ? L.red[w * 16 + lane] : 0.0;   // independent reads in flight
#pragma unroll
    for (int w = 0; w < MAXW; w++) t += x[w];
  }
  // broadcast through LDS (a wave's LDS accesses execute in order): the totals stay in vector registers,
  // which this kernel has plenty of, instead of 20 scalar registers it has not
  double* bc = L.red + (ICP_MAXW + wave) * 16;
  if (lane <= NSUM) bc[lane] = t;
#pragma unroll
  for (int k = 0; k < NSUM; k++) tot[k] = bc[k];
  cnt_total = (int)bc[NSUM];
}

// The same for EIGHT values per thread without the LDS transpose: the transpose costs every wave 9 LDS writes and 16
// reads per lane and step, all waves at the same moment -- 1 150 cycles of the 8 600-cycle step, more with more waves
// (profiles/r4_icp_critical_path.txt).  Here a wave reduces in registers by halving: lane pairs exchange HALF of their values (the even
// lane keeps and completes values 0-3, the odd lane 4-7; DPP quad permutes), quads half of those, so that after two steps lane j of every quad holds the
// quad's sums of two values; two row shifts, then the two cross-row swaps gfx950 has (v_permlane16_swap / v_permlane32_swap) finish
// them: ~75 vector instructions, no LDS.  Lanes 12-15 hand the wave's eight sums to LDS, and behind the barrier eight lanes add the
// waves' rows (rows of absent waves are zero: no branches) and pass the totals on through the wave's broadcast row.
// Fixed order => deterministic; the order differs from the transpose's, i.e. the totals' last bits do.
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v)                   // quad_perm CTRL of v (every lane has a source)
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swap16_sum(double x)                 // x[row r] + x[row r ^ 1] in every lane
{
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swap32_sum(double x)                 // x[lane] + x[lane ^ 32] in every lane
{
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
template <int MAXW>
__device__ __forceinline__ void block_totals8(const IcpLds& L, const double (&v)[8], double (&tot)[8], int lane, int wave, long long* tl = nullptr)
{
  constexpr int X1 = 0xB1, X2 = 0x4E;                // quad_perm [1,0,3,2] (lane ^ 1), [2,3,0,1] (lane ^ 2)
  const bool odd = (lane & 1) != 0, up = (lane & 2) != 0;
  double w[4], u[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    // even lanes: v[i] + the odd neighbour's v[i]; odd lanes: v[i+4] + the even neighbour's v[i+4]
    const double keep = odd ? v[i + 4] : v[i], send = odd ? v[i] : v[i + 4];
    w[i] = keep + dpp_quad<X1>(send);
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    // lanes 0, 1 of a quad: w[i] + (lane ^ 2)'s w[i]; lanes 2, 3: w[i+2] + (lane ^ 2)'s w[i+2]
    const double keep = up ? w[i + 2] : w[i], send = up ? w[i] : w[i + 2];
    u[i] = keep + dpp_quad<X2>(send);
  }
  // lane j of a quad now holds the quad's sums of values base(j), base(j) + 1 with base = {0, 4, 2, 6}[j]
#pragma unroll
  for (int i = 0; i < 2; i++) {
    u[i] += dpp_shr0<0x114>(u[i]);      // row_shr:4
    u[i] += dpp_shr0<0x118>(u[i]);      // row_shr:8  -> lanes 12-15 of a row: the row's sums
    u[i] = swap16_sum(u[i]);
    u[i] = swap32_sum(u[i]);            // -> lanes 12-15 of every row: the wave's sums
  }
  const int j = lane & 3;
  const int base = ((j & 1) << 2) | (j & 2);
  if (lane >= 12 && lane < 16) *reinterpret_cast<double2*>(L.red + wave * 16 + base) = make_double2(u[0], u[1]);
  if (tl && lane == 0) { asm volatile("" : "+v"(u[0])); tl[7] = clock64(); }
  __syncthreads();
  if (tl && lane == 0) tl[8] = clock64();
  double* bc = L.red + (ICP_MAXW + wave) * 16;
  {
    // every lane adds the column lane & 7 (eight lanes of a wave used to, inside an exec-masked region with its skip branch: the same
    // instructions per wave, but one basic block from the barrier to the totals and ~30 cycles of every step less)
    const int c = lane & 7;
    double x[MAXW];
#pragma unroll
    for (int r = 0; r < MAXW; r++) x[r] = L.red[r * 16 + c];
#pragma unroll
    for (int st = 1; st < MAXW; st <<= 1)
#pragma unroll
      for (int r = 0; r + st < MAXW; r += 2 * st) x[r] += x[r + st];
    bc[c] = x[0];                           // (eight lanes per word, the same value)
  }
  // broadcast through LDS (a wave's LDS accesses execute in order)
#pragma unroll
  for (int k = 0; k < 8; k += 2) { const double2 t2 = *reinterpret_cast<const double2*>(bc + k); tot[k] = t2.x; tot[k + 1] = t2.y; }
}

// the whole registration of one workgroup; k_icp (one registration per launch) and k_icp_batch (workgroup x = registration x
// of a batch) are thin wrappers
// PAIRS (parity / debug instantiation, tsd_icp_pairs): the scene is NOT moved between the steps and every step's surviving pair list
// is written out -- the repeated PairAssignment::determinePairs calls on a static scene that the compiled reference's chain
// (PairAssignment.cpp:38-84 -> DistanceFilter -> ReciprocalFilter) is driven with in tests/golden/ref_chain_pairs.npz.
// PRE (fused registration_mode 3, k_icp_pre): Tinit is not read from memory at the top but taken from the launch's own arg-max workgroup
// when the set-up first needs it (tsdpdf_device.hpp).
template <int R, int MAXT, bool PTL, bool PAIRS = false, int FCAP = 0, int FT = 0, bool PRE = false>
__device__ __forceinline__ void
icp_workgroup(IcpArgs a, const double* __restrict__ P_dev, int cap_rt, const double* __restrict__ g_model, const double* __restrict__ g_scene,
      const int* __restrict__ g_morig, const int* __restrict__ g_start,
      const double* __restrict__ g_coords, const uint8_t* __restrict__ g_mask_m,
      const double* __restrict__ g_rays_local, const double* __restrict__ g_ranges,
      const uint8_t* __restrict__ g_mask, IcpResultDev* __restrict__ out,
      double* __restrict__ trace /* [TSD_ICP_TRACE_MAX][TSD_ICP_TRACE_STRIDE] = pairs, rms, thr_before, state, Tlast (co, si, dX, dY) */, const ScanPostArgs& post,
      const double* __restrict__ g_mnormals /* direct mode */, const double* __restrict__ g_normals /* fused: ray cast */,
      const IcpSeedArgs seed = IcpSeedArgs{nullptr, 0u, 0, 0}, const int role = 0 /* 0: the registration; h > 0: helper h of step 0's searches */,
      int* __restrict__ pairs_out = nullptr /* PAIRS: [steps][cap] winning scene index per model slot, preset to -1 */,
      const IcpPreArgs* __restrict__ prep = nullptr /* PRE */)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef TSD_ICP_TIMELINE
  const long long tk_entry = clock64();          // (whole-kernel phases: set-up, loop, epilogue -- row TSD_ICP_TRACE_MAX + 192 of the trace)
#endif
  // fused mode: every global read of the inputs is issued HERE, first thing, unconditionally and at once (one memory latency instead of
  // one per dependent step; the launcher guarantees beams <= R * T) -- ahead of the LDS layout and of everything else the waves do on
  // their way in: the younger wave of a SIMD issues at half rate and used to reach its loads 3 000 cycles after the older one
  // (tools/icp_tail.sh, "inputs arrived"), and the first barrier of the set-up waits for the last wave's inputs.
  bool in_fm[R], in_fs[R];
  double in_rr[R], in_lx[R], in_ly[R];
  double in_cmx[R], in_cmy[R], in_nnx[R], in_nny[R];       // (plain doubles: an array of double2 stays an alloca -- scratch memory -- in this compiler)
  if (a.beams > 0) {
    const int tid0 = threadIdx.x, T0 = FT ? FT : (int)blockDim.x;
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int b = q * T0 + tid0;
      const int bc = b < a.beams ? b : 0;
      const uint8_t mm = g_mask_m[bc], ms = g_mask[bc];
      in_rr[q] = g_ranges[bc];
      { const double2 c2 = *reinterpret_cast<const double2*>(g_coords + 2 * (size_t)bc); in_cmx[q] = c2.x; in_cmy[q] = c2.y; }
      in_lx[q] = g_rays_local[bc]; in_ly[q] = g_rays_local[a.beams + bc];
      { const double2 n2 = PTL ? *reinterpret_cast<const double2*>(g_normals + 2 * (size_t)bc) : make_double2(0.0, 0.0); in_nnx[q] = n2.x; in_nny[q] = n2.y; }
      in_fm[q] = (b < a.beams) && mm != 0;
      in_fs[q] = (b < a.beams) && !isinf(in_rr[q]) && ms != 0;
    }
  }
  IcpLds L;
  // (FCAP / FT: the capacity and thread count as compile-time constants -- the LDS layout's offsets then cost no scalar registers)
  const int cap = FCAP ? FCAP : cap_rt;
  const int lcap = icp_list_cap(cap);
  {
    char* p = smem;
    L.mxy = reinterpret_cast<double2*>(p) + ICP_PAD; p += sizeof(double2) * (size_t)(cap + 2 * ICP_PAD);
    L.uxy = reinterpret_cast<double2*>(p); p += sizeof(double2) * (size_t)cap;
    L.list_xy = reinterpret_cast<double2*>(p); L.stage_s = reinterpret_cast<double2*>(p);
    p += sizeof(double2) * (size_t)lcap;
    L.res_d = reinterpret_cast<double*>(p); p += sizeof(double) * (size_t)lcap;
    L.res_lb = reinterpret_cast<double*>(p); p += sizeof(double) * (size_t)lcap;
    L.list_k = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)lcap;
    L.res_k = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)lcap;
    L.res_k2 = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)lcap;
    L.list2 = reinterpret_cast<int*>(p);
    p = reinterpret_cast<char*>(L.list_xy) + icp_region_bytes(cap, (FT ? FT : (int)blockDim.x));
    // staging view of the same 40*lcap bytes: cap double2 then cap int (40*lcap >= 20*cap)
    L.start = reinterpret_cast<int*>(reinterpret_cast<char*>(L.stage_s) + sizeof(double2) * (size_t)cap);
    L.slotD = reinterpret_cast<unsigned long long*>(p); p += sizeof(unsigned long long) * (size_t)icp_slot_halves(cap, (FT ? FT : (int)blockDim.x), PTL) * (size_t)icp_slot_cap(cap, (FT ? FT : (int)blockDim.x));
    L.red = reinterpret_cast<double*>(p); p += sizeof(double) * 2 * ICP_MAXW * 16;
    L.cst = reinterpret_cast<double*>(p); p += sizeof(double) * 16;
    L.tail = reinterpret_cast<IcpTail*>(p); p += (sizeof(IcpTail) + 15) & ~(size_t)15;
    L.tr = reinterpret_cast<double*>(L.list_xy);     // T * 72 B <= 40 * lcap B (checked by the launcher)
    L.slotI = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)cap;
    L.morig = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)cap;
    L.ired = reinterpret_cast<int*>(p); p += sizeof(int) * 64;
    p = reinterpret_cast<char*>(((uintptr_t)p + 15) & ~(uintptr_t)15);
    L.nxy = PTL ? reinterpret_cast<double2*>(p) : nullptr;
  }

  if (P_dev) {   // fused scan: the pre-registration sensor pose lives on the device
#pragma unroll
    for (int i = 0; i < 6; i++) a.P[i] = P_dev[i];
  }
  if constexpr (!PRE) {
    if (a.Tinit_dev) {   // fused registration_mode 3: so does Tinit (k_pdf_argmax's result, the kernel right before this one)
#pragma unroll
      for (int i = 0; i < 6; i++) a.Tinit[i] = a.Tinit_dev[i];
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = FT ? FT : (int)blockDim.x, W = T >> 6;
  int nM = 0, nS = 0;
#ifdef TSD_ICP_TIMELINE
  // diagnostic build (tools/icp_timeline.sh): lane 0 of EVERY wave stamps the shader clock at 14 points of four steady-state steps
  // (TL_FIRST ..), so that each wave's own chain and the waits at the two barriers can be read off: profiles/r4_icp_critical_path.txt.
  // A stamp is s_memtime + wait + one LDS write (~60 cycles, the same for every interval).
  constexpr int TL_FIRST = TSD_ICP_TL_FIRST, TL_STEPS = TSD_ICP_TL_STEPS, TL_N = 16;
  long long* tlbuf = reinterpret_cast<long long*>(smem + icp_lds_bytes_for(cap, (FT ? FT : (int)blockDim.x), PTL));       // [TL_STEPS][W][TL_N], behind the kernel's own LDS
#define TL(i) do { if (lane == 0 && iter >= (unsigned)TL_FIRST && iter < (unsigned)(TL_FIRST + TL_STEPS)) \
                     tlbuf[((iter - TL_FIRST) * W + wave) * TL_N + (i)] = clock64(); } while (0)
  long long tk_loop0 = 0, tk_loop1 = 0, tk_seed = 0, tk_s[6] = {0, 0, 0, 0, 0, 0};
#else
#define TL(i) do {} while (0)
#endif

  // fused scan: the sensor state the epilogue needs (pose, _lastPose) is requested NOW, ahead of the inputs, and parked in LDS once
  // it is there -- its memory round trip rides along with the inputs' instead of opening the epilogue
  // (one 8-byte word per lane of wave 0, one load instruction: pose, _lastPose, calcAngle(_lastPose) and the flag are the first 20 words
  // of SensorDev; thread 0 alone used to carry all of it in registers and store it word by word behind the staging barrier, 1 400
  // cycles that every other wave waited for at the next one)
  static_assert(offsetof(SensorDev, last_pose) == 72 && offsetof(SensorDev, have_last_pose) == 144 && offsetof(SensorDev, last_angle) == 152, "SensorDev");
  static_assert(offsetof(ScanPostPre, last) == 72 && offsetof(ScanPostPre, last_angle) == 144 && offsetof(ScanPostPre, have_last) == 152, "ScanPostPre");
  unsigned long long pre_word = 0ull;
  if (post.st && role == 0 && tid < 20) pre_word = reinterpret_cast<const unsigned long long*>(post.st)[tid];

  // ---------------------------------------------------------------- inputs
  if (a.beams > 0) {
    // fused mode: maskMatrix compaction of the ray-cast model and of the scan's cartesian points (read at the top of the function).
    // Model points stay in beam order = angular order about the sensor.
    bool (&fm)[R] = in_fm, (&fs)[R] = in_fs;
    double (&rr)[R] = in_rr, (&lx)[R] = in_lx, (&ly)[R] = in_ly;
    double (&cmx)[R] = in_cmx, (&cmy)[R] = in_cmy, (&nnx)[R] = in_nnx, (&nny)[R] = in_nny;
#ifdef TSD_ICP_TIMELINE
    { double t0 = rr[0] + cmx[0] + lx[0] + (double)fm[0] + (double)fs[0]; asm volatile("" : "+v"(t0)); tk_s[0] = clock64(); }     // inputs arrived
#endif
    int* cnts = reinterpret_cast<int*>(L.red);           // [R][W][2] (the reduction rows are idle during setup)
    unsigned long long bm[R], bs[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
      bm[q] = __ballot(fm[q]); bs[q] = __ballot(fs[q]);
      if (lane == 0) { cnts[(q * W + wave) * 2] = __popcll(bm[q]); cnts[(q * W + wave) * 2 + 1] = __popcll(bs[q]); }
    }
    __syncthreads();
    if constexpr (PRE) {
      // Tinit, first needed here (the scene's points below): the launch's arg-max workgroup hands TBest's two rows over as twelve tagged
      // granules (tsdpdf_device.hpp), re-read at agent scope (sc1) until every one carries this launch's number.  Its chain (scores and
      // candidates -> reduction -> the winner's two points -> sine and cosine) is about as long as this workgroup's way here, so the wait
      // is short; bounded all the same (1 ms of the 100 MHz clock: the first workgroup of a grid cannot fail to run).
      const long long t0w = wall_clock64();
      const unsigned long long* gr = reinterpret_cast<const unsigned long long*>(prep->flag);
      unsigned long long gv[12];
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 12; i++) gv[i] = __hip_atomic_load(gr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 12; i++) ok &= (unsigned int)(gv[i] >> 32) == prep->seq;
        if (ok || wall_clock64() - t0w > 100000) break;
        __builtin_amdgcn_s_sleep(2);
      }
#pragma unroll
      for (int i = 0; i < 6; i++) a.Tinit[i] = __longlong_as_double((long long)((gv[2 * i] & 0xFFFFFFFFull) | (gv[2 * i + 1] << 32)));
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    int runM = 0, runS = 0;
#pragma unroll
    for (int q = 0; q < R; q++) {
      int offM = 0, offS = 0;
      for (int w = 0; w < W; w++) {
        const int c_m = cnts[(q * W + w) * 2], c_s = cnts[(q * W + w) * 2 + 1];
        if (w == wave) { offM = runM; offS = runS; }
        runM += c_m; runS += c_s;
      }
      offM += __popcll(bm[q] & lt); offS += __popcll(bs[q] & lt);
      if (fm[q] && offM < cap) { L.mxy[offM] = make_double2(cmx[q], cmy[q]); L.morig[offM] = offM; if (L.nxy) L.nxy[offM] = make_double2(nnx[q], nny[q]); }
      if (fs[q] && offS < cap) {
        // coords = raysLocal(j,i) * data[i] (Sensor.cpp:176-179), then applyTransformation(_sceneTmp, Tinit) (Icp.cpp:481-486,
        // :371-408) like the direct mode below: (0 + x*R00) + y*R01, then + t.  Tinit is the identity unless a pre-registration ran
        // (registration_mode 3), and x*1 + y*0 + 0 == x exactly: mode 0 is unchanged bit for bit
        const double x = lx[q] * rr[q], y = ly[q] * rr[q];
        double nx = 0.0, ny = 0.0;
        nx += x * a.Tinit[0]; nx += y * a.Tinit[1]; ny += x * a.Tinit[3]; ny += y * a.Tinit[4];
        L.stage_s[offS] = make_double2(nx + a.Tinit[2], ny + a.Tinit[5]);
        L.start[offS] = offM;      // model slot of the same beam (or of the next hit beam): a search hint only
      }
    }
    nM = runM; nS = runS;
    if (a.beams > R * T) nM = cap + 1;                   // (not reachable through launch_icp)
    __syncthreads();
#ifdef TSD_ICP_TIMELINE
    tk_s[1] = clock64();                                 // compacted model and scene in LDS
#endif
  } else {
    nM = a.n_model; nS = a.n_scene;
    if (nM <= cap && nS <= cap) {
      for (int j = tid; j < nM; j += T) { L.mxy[j] = make_double2(g_model[2 * j], g_model[2 * j + 1]); L.morig[j] = g_morig[j]; }
      if (L.nxy) for (int j = tid; j < nM; j += T) L.nxy[j] = make_double2(g_mnormals[2 * j], g_mnormals[2 * j + 1]);
      // applyTransformation(_sceneTmp, Tinit) (Icp.cpp:481-486, :371-408): (0 + x*R00) + y*R01, then + t; Tinit is the
      // identity unless a pre-registration ran (x*1 + y*0 + 0 == x exactly, so mode 0 is unchanged)
      for (int i = tid; i < nS; i += T) {
        const double x = g_scene[2 * i], y = g_scene[2 * i + 1];
        double nx = 0.0, ny = 0.0;
        nx += x * a.Tinit[0]; nx += y * a.Tinit[1]; ny += x * a.Tinit[3]; ny += y * a.Tinit[4];
        L.stage_s[i] = make_double2(nx + a.Tinit[2], ny + a.Tinit[5]); L.start[i] = g_start[i];
      }
    }
    __syncthreads();
  }
  if (role == 0) {       // (a helper searches and leaves: it has no tail)
    if (tid == 0) {
      L.tail->out = out; L.tail->trace = trace;
      L.ired[IR_CNT] = 0; L.ired[IR_RMAX] = 0; L.ired[IR_CNT2] = 0; L.ired[IR_TIE] = 0; L.ired[IR_SEEDED] = 0;
    }
    if (tid == 64 % T) L.tail->post = post;                  // (another wave's lane: ~30 stores)
    // fused scan: the sensor state the epilogue starts from, requested at the top (words 18 / 19 change places: see the assertions)
    if (post.st && tid < 20) reinterpret_cast<unsigned long long*>(&L.tail->pre)[tid < 18 ? tid : 37 - tid] = pre_word;
  }

  // rows 0,1 of _Tfinal4x4: [r00 r01 tx ; r10 r11 ty]; (*_Tfinal4x4) = (*Tinit) * I (Icp.cpp:485)
  double Tf[6] = {a.Tinit[0], a.Tinit[1], a.Tinit[2], a.Tinit[3], a.Tinit[4], a.Tinit[5]};
  double rms = 0.0;
  int pairs = 0, state = TSD_ICP_PROCESSING;
  unsigned int iter = 0;

  if (role > 0 && (nM == 0 || nS == 0 || nM > cap || nS > cap)) return;      // (nothing to search; the registration does not wait)
  if (nM == 0 || nS == 0 || nM > cap || nS > cap) {
    // Icp::iterate early-out (Icp.cpp:467-471); ThreadLocalize never gets here with nM == 0
    IcpResultDev r;
    for (int i = 0; i < 9; i++) r.T[i] = (i % 4 == 0) ? 1.0 : 0.0;
    r.rms = 0.0; r.pairs = 0; r.iterations = 0; r.state = TSD_ICP_NOTMATCHABLE;
    r.n_model = nM; r.n_scene = nS; r.reserved = (nM > cap || nS > cap) ? TSD_E_CAPACITY : 0;
    if (tid == 0) *out = r;
    __syncthreads();            // (the tail, incl. the preloaded sensor state, is thread 0's)
    if (post.st) scan_post_body(post, L.tail->pre, r.T, r, post.gmin_x, post.gmax_x, post.gmin_y, post.gmax_y);
    return;
  }

  // every thread takes its scene points into registers; unit directions of the model
  double sx[R], sy[R], lb[R];
  int hint[R], hint2[R];
  bool have[R];
  float rmaxf = 0.f;
  // Scene points -> register slots (blk = 64-point group).  Waves i and i + 4 of a workgroup share a SIMD (tools/exp/hwid.hip).
  int pid[R];
  {
    // waves 0-3 are the older wave of their SIMD and issue at full rate, waves 4+ get the slots the older wave leaves (tools/exp/valu.hip:
    // 4.9 against 8.9 cycles per fp64 instruction): waves 0-3 take R 64-point blocks each, waves 4+ share what is left, one block each
    const int nOld = W < 4 ? W : 4, nYoung = W - nOld;
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int blk = wave < 4 ? q * nOld + wave : nOld * R + q * nYoung + (wave - 4);
      pid[q] = blk * 64 + lane;
    }
  }
#pragma unroll
  for (int q = 0; q < R; q++) {
    const int i = pid[q];
    have[q] = i < nS;
    sx[q] = 0.0; sy[q] = 0.0; hint[q] = 0; hint2[q] = 0; lb[q] = -1.0;
    if (have[q]) {
      const double2 s = L.stage_s[i];
      sx[q] = s.x; sy[q] = s.y;
      // |s| rounded up: fp32 is plenty for a bound
      const float rf = __builtin_amdgcn_sqrtf((float)(s.x * s.x + s.y * s.y) * 1.000001f) * 1.000001f;
      rmaxf = fmaxf(rmaxf, rf);
      int h = L.start[i];
      hint[q] = h < 0 ? 0 : (h >= nM ? nM - 1 : h);
      hint2[q] = hint[q];
    }
  }
  for (int k = tid; k < nM; k += T) {
    const double2 m = L.mxy[k];
    const double r2 = m.x * m.x + m.y * m.y;
    double2 u = make_double2(0.0, 0.0);
    if (r2 > 0.0) { const double inv = rsqrt(r2); u.x = m.x * inv; u.y = m.y * inv; }   // |u| = 1 within 1e-15: SLACK covers it
    L.uxy[k] = u;
  }
  for (int i = tid; i < ICP_PAD; i += T) {        // wrapped copies at both ends (window_search)
    L.mxy[nM + i] = L.mxy[i % nM];
    L.mxy[-1 - i] = L.mxy[nM - 1 - (i % nM)];
  }
  if (role > 0) {
    // helper: step 0's window search for every seed.helpers-th scene point, from the staged scene (the registration's
    // own list pass reads the same coordinates and the same hint from its registers); what the window cannot prove -- a point far from
    // the whole model: the scan turned into space the map does not hold yet -- goes to the whole-wave search right here, shared out
    // over ALL waves of ALL helpers (the registration's own eight waves had them to themselves: 30 000-80 000 cycles of the slowest scans)
    if (tid == 0) L.ired[IR_CNT2] = 0;
    __syncthreads();                   // unit directions and padding in place
    const int per = T < ICP_HELPER_POINTS ? T : ICP_HELPER_POINTS;
    const int i = tid * seed.helpers + (role - 1);       // interleaved: a sector of far points is every helper's in equal parts
    const unsigned long long tag = (unsigned long long)seed.seq << 32;
    int* const far_list = L.slotI;     // (a helper pairs nothing: the slot arrays are free)
    if (tid < per && i < nS) {
      const double2 s = L.stage_s[i];
      int h = L.start[i];
      h = h < 0 ? 0 : (h >= nM ? nM - 1 : h);
      const NnResult r = window_search(L, nM, s.x, s.y, h, a.thr0, a.ccw ? 1.0 : -1.0);
      if (r.resolved) {
        const unsigned int root = __float_as_uint(__builtin_amdgcn_sqrtf((float)r.lbsq));       // lb_from_sq's fp32 root
        __hip_atomic_store(seed.g + i, tag | root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(seed.g + seed.stride + i, tag | ((unsigned)r.bk | ((unsigned)r.bk2 << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        far_list[atomicAdd(&L.ired[IR_CNT2], 1)] = i;
      }
    }
    __syncthreads();
    const int n_far = L.ired[IR_CNT2];
    for (int e = wave; e < n_far; e += W) {
      const int j = far_list[e];
      const double2 s = L.stage_s[j];
      int h = L.start[j];
      h = h < 0 ? 0 : (h >= nM ? nM - 1 : h);
      const NnResult r = wave_search(L, nM, s.x, s.y, h, a.thr0, a.ccw ? 1.0 : -1.0, lane);
      if (lane == 0) {
        const unsigned int root = r.bk >= 0 ? __float_as_uint(__builtin_amdgcn_sqrtf((float)r.lbsq)) : 0u;
        const unsigned int kk = r.bk >= 0 ? ((unsigned)r.bk | ((unsigned)(r.bk2 >= 0 ? r.bk2 : r.bk) << 16)) : 0xFFFFu;     // (a non-finite point stays the registration's)
        __hip_atomic_store(seed.g + j, tag | root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(seed.g + seed.stride + j, tag | kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
#ifdef TSD_ICP_TIMELINE
  tk_s[2] = clock64();                                   // scene in registers, unit directions and padding written
#endif
  const int slot_halves = icp_slot_halves(cap, T, PTL);
  const int scap = icp_slot_cap(cap, T);
  for (int k = tid; k < scap; k += T) { L.slotD[k] = ~0ull; if (slot_halves == 2) L.slotD[scap + k] = ~0ull; }
  for (int k = tid; k < cap; k += T) L.slotI[k] = INT_MAX;
  for (int k = tid; k < ICP_MAXW * 16; k += T) L.red[k] = 0.0;    // (block_totals8 reads the rows of absent waves)
  if (role == 0 && !PTL) {
    // the waves' sums of the scene coordinates (c0 below), parked in the broadcast rows (scratch until the first step)
    double cen[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < R; q++)       // (a NaN, an infinity or an absurd coordinate never finds a partner: it must not poison the centre either)
      if (have[q] && fabs(sx[q]) < 1e6 && fabs(sy[q]) < 1e6) { cen[0] += sx[q]; cen[1] += sy[q]; }
#pragma unroll
    for (int k = 0; k < 2; k++) { const double t = wave_total(cen[k]); if (lane == 0) L.red[(ICP_MAXW + wave) * 16 + k] = t; }
  }
  __syncthreads();                     // staging consumed (it aliases the work list); counters zeroed
#ifdef TSD_ICP_TIMELINE
  tk_s[3] = clock64();
#endif
  // (the wave's maximum first: 512 atomics on one LDS word are 512 turns -- 5 800 cycles of the set-up, tools/icp_tail.sh)
  for (int o = 32; o; o >>= 1) rmaxf = fmaxf(rmaxf, __shfl_xor(rmaxf, o));
  if (lane == 0 && rmaxf > 0.f) atomicMax(&L.ired[IR_RMAX], __float_as_int(rmaxf));   // positive floats order like ints
  __syncthreads();
  const double scene_rmax = (double)__int_as_float(L.ired[IR_RMAX]);
#ifdef TSD_ICP_TIMELINE
  { double t0 = scene_rmax; asm volatile("" : "+v"(t0)); tk_s[4] = clock64(); }
#endif

  // ---------------------------------------------------------------- iterate
  // loop-invariant scalars: vector registers (the scalar file is the scarce one in this kernel)
  const double P00 = vreg(a.P[0]), P01 = vreg(a.P[1]), P02 = vreg(a.P[2]), P10 = vreg(a.P[3]), P11 = vreg(a.P[4]), P12 = vreg(a.P[5]);
  const double bmin_x = vreg(a.min_x), bmax_x = vreg(a.max_x), bmin_y = vreg(a.min_y), bmax_y = vreg(a.max_y);
  const double thr_mult = vreg(a.multiplier), thr_min = vreg(a.min_sqr);
  const double sgn = a.ccw ? 1.0 : -1.0;     // +1: slots ascend counter-clockwise
  double thr = a.thr0;                       // DistanceFilter::_distSqr after reset()
  double rms_prev = 10e12;
  unsigned int conv_cnt = 0;
  const unsigned int max_it = (unsigned)a.iterations, conv_need = PAIRS ? ~0u : (unsigned)a.iterations;
  // rows of the pose's rotation block are unit vectors up to rounding: |R_p s| <= pnorm |s| per axis
  const double pnorm = fmax(sqrt(P00 * P00 + P01 * P01), sqrt(P10 * P10 + P11 * P11)) * (1.0 + 1e-9);

  // OutOfBoundsFilter2D can filter nothing while a disc of radius (largest scene radius + accumulated translation) x pnorm around the
  // sensor lies inside the bounds: the largest accumulated translation (squared, rounded down) for which that holds
  double tcum_lim2 = -1.0;
  {
    const double slack = fmin(fmin(P02 - bmin_x, bmax_x - P02), fmin(P12 - bmin_y, bmax_y - P12));
    const double lim = ((slack - 2e-6) / (pnorm * (1.0 + 2e-6)) - scene_rmax) * (1.0 - 1e-6);
    if (lim > 0.0) tcum_lim2 = lim * lim * (1.0 - 1e-6);
  }
  // centring point of the pair sums: last step's centroids; for the first step the centroid of ALL scene points, for the model side
  // too (model and scene share the frame, and the pairs' own centroids lie within decimetres of it: the correction term below stays as
  // well conditioned as in every other step, and the first step no longer runs its pair sums twice -- 3 000 cycles of every
  // registration).  Not the model's own centroid: the order of the model differs between tsd_icp and the fused calls (a rotation of the
  // beams), the scene's does not, and the two have to agree to the bit (tests/test_gpu_parity.py)
  double c0[4] = {0.0, 0.0, 0.0, 0.0};
  if (!PTL) {
    double t2[2] = {0.0, 0.0};
    for (int w = 0; w < W; w++) { t2[0] += L.red[(ICP_MAXW + w) * 16]; t2[1] += L.red[(ICP_MAXW + w) * 16 + 1]; }
    c0[0] = c0[2] = t2[0] / (double)nS; c0[1] = c0[3] = t2[1] / (double)nS;
  }

#ifdef TSD_ICP_TIMELINE
  const bool has_trace = trace != nullptr && !post.st;
#else
  const bool has_trace = trace != nullptr;   // (a scalar: reading the pointer back from LDS every step cost thread 0's wave a round trip)
#endif
  int Rn = 0;                                // register slots of this wave that hold scene points (wave-uniform)
  for (int q = 0; q < R; q++) Rn += (pid[q] - lane < nS) ? 1 : 0;
  Rn = __builtin_amdgcn_readfirstlane(Rn);   // (a scalar: "this register slot is empty" is then a scalar branch, which the likelihood hints below do not remove)

#ifdef TSD_ICP_TIMELINE
  tk_seed = clock64();
  if (lane == 0 && trace && role == 0) {        // every wave's own set-up stamps (rows TSD_ICP_TRACE_MAX + 194 ..)
    double* pw = trace + TSD_ICP_TRACE_STRIDE * (TSD_ICP_TRACE_MAX + 194 + wave);
    pw[0] = (double)(tk_s[0] - tk_entry); pw[1] = (double)(tk_s[1] - tk_entry); pw[2] = (double)(tk_s[2] - tk_entry); pw[3] = (double)(tk_seed - tk_entry);
    pw[4] = (double)(tk_s[3] - tk_entry); pw[5] = (double)(tk_s[4] - tk_entry);
  }
#endif
  // step 0's searches, done by the helper workgroups while this one set itself up: every lane re-reads its points' granules until
  // they carry this launch's number (a bounded wait, per wave) and takes neighbour, runner-up and bound from them
  if (!PAIRS && seed.helpers > 0) {
    const long long t0 = wall_clock64();
    unsigned long long g0[R], g1[R];
    bool got[R];
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < R; q++) {
        const int i = have[q] ? pid[q] : 0;
        g0[q] = __hip_atomic_load(seed.g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g1[q] = __hip_atomic_load(seed.g + seed.stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int q = 0; q < R; q++) {
        got[q] = (unsigned)(g0[q] >> 32) == seed.seq && (unsigned)(g1[q] >> 32) == seed.seq;
        ok &= got[q] | !have[q];
      }
      if (__all(ok) || wall_clock64() - t0 > ICP_SEED_WAIT_TICKS) break;
      __builtin_amdgcn_s_sleep(2);
    }
    {
      // (observability: how many points started from a helper's granules -- tsd_icp_result.reserved; a broken tag or stride, or
      // helpers that never got a compute unit, would otherwise only cost time, the results being the same either way)
      int seeded = 0;
#pragma unroll
      for (int q = 0; q < R; q++) seeded += __popcll(__ballot(have[q] && got[q]));
      if (lane == 0 && seeded) atomicAdd(&L.ired[IR_SEEDED], seeded);
    }
#pragma unroll
    for (int q = 0; q < R; q++) {
      const unsigned kk = (unsigned)g1[q];
      if (have[q] && got[q]) {
        if ((kk & 0xFFFFu) != 0xFFFFu) {
          hint[q] = (int)(kk & 0xFFFFu); hint2[q] = (int)(kk >> 16);
          lb[q] = (double)__uint_as_float((unsigned)g0[q]) * (1.0 - 1e-6);          // lb_from_sq
        } else {
          lb[q] = LB_WINDOW_TRIED;      // the helper's window search -- the one this workgroup would run -- proved nothing: tier 2 directly
        }
      }
    }
  }

#ifdef TSD_ICP_TIMELINE
  tk_loop0 = clock64();
#endif
  while (state == TSD_ICP_PROCESSING) {
    const double thr_before = thr;
    TL(0);
    // The slots a step's pairs went into are given back behind barrier 2 of that step, and the next step's atomics come before ITS
    // barrier 1: nothing orders the two.  They are a transform and a tier 0 apart (~2 500 cycles) while the waves leave a barrier within
    // tens of cycles of each other, so the hand-back always won -- but only by timing.  Alternate steps use alternate halves of the slot
    // array (where the LDS holds two, icp_slot_halves): a half is given back a whole step (two barriers) before it is used again.
    unsigned long long* const slotD = L.slotD + (size_t)(iter & (unsigned)(slot_halves - 1)) * (size_t)scap;

    // -- phase A: pre-filter + exact NN + distance filter (per scene point)
    // OutOfBoundsFilter2D: when even a disc of the largest possible scene radius around the sensor
    // lies inside the bounds nothing can be filtered and the per-point transform is skipped.
    // (the sufficient condition solved for the accumulated translation once, ahead of the loop: three operations per step)
    const bool all_in = (Tf[2] * Tf[2] + Tf[5] * Tf[5]) * (1.0 + 4e-6) < tcum_lim2;
    double bd[R]; bool keep[R], need[R];
    int ent[R];
    double2 mw[R];                            // the neighbour's coordinates
    const bool refresh = (iter - REFRESH_A) * (iter - REFRESH_B) == 0u;      // (one scalar compare; `== || ==` compiles to two branches on the steady path)
    // OutOfBoundsFilter2D for one point: S.transform(P): (0 + x*R00) + y*R01, then + t (gsl/Matrix.cpp:403-432)
    auto inside_bounds = [&](double x, double y) {
      double wx = 0.0, wy = 0.0;
      wx += x * P00; wx += y * P01;
      wy += x * P10; wy += y * P11;
      wx += P02; wy += P12;
      return !((int)(wx < bmin_x) | (int)(wx > bmax_x) | (int)(wy < bmin_y) | (int)(wy > bmax_y));
    };
    {
      double2 mh[R], mh2[R];
#pragma unroll
      for (int q = 0; q < R; q++) { mh[q] = L.mxy[hint[q]]; mh2[q] = L.mxy[hint2[q]]; }   // all reads in flight
#ifdef TSD_ICP_TIMELINE
#pragma unroll
      for (int q = 0; q < R; q++) { asm volatile("" : "+v"(mh[q].x), "+v"(mh2[q].x)); }
      TL(1);                                   // the neighbours' coordinates have arrived
#endif
      // (the bounds test of all register slots behind ONE wave-uniform branch: a region per slot was three taken branches per step)
      bool inb[R];
#pragma unroll
      for (int q = 0; q < R; q++) inb[q] = true;
      if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)all_in) == 0, 0)) {
#pragma unroll
        for (int q = 0; q < R; q++) inb[q] = inside_bounds(sx[q], sy[q]);
      }
#pragma unroll
      for (int q = 0; q < R; q++) {
        bd[q] = __builtin_inf(); keep[q] = false; need[q] = false; ent[q] = -1; mw[q] = mh[q];
        if (q >= Rn) continue;                  // (wave-uniform) no scene point in this register slot
        const double x = sx[q], y = sy[q];
        const bool pre = have[q] & inb[q];
        // the nearer of the last neighbour and its runner-up is the exact neighbour as long as it beats
        // the bound on everything else (a point hovering between two model points never searches)
        const double dx1 = x - mh[q].x, dy1 = y - mh[q].y, dx2 = x - mh2[q].x, dy2 = y - mh2[q].y;
        const double d1 = dx1 * dx1 + dy1 * dy1, d2 = dx2 * dx2 + dy2 * dy2;
        bool swp = d2 < d1;
        // (an exact tie between two different candidates -- rare -- is settled by the original model index: looked for once per wave)
        if (__builtin_expect(__ballot((d2 == d1) & (hint2[q] != hint[q])) != 0ull, 0))
          if (d2 == d1 && hint2[q] != hint[q]) swp = L.morig[hint2[q]] < L.morig[hint[q]];
        const double d = swp ? d2 : d1;
        const int kn = swp ? hint2[q] : hint[q], ko = swp ? hint[q] : hint2[q];
        mw[q] = swp ? mh2[q] : mh[q];
        hint[q] = kn; hint2[q] = ko;
        {
          // The decisions from three compares.  "known" (lb > 0) is folded into the bound (max(lb, 0)^2 = 0 makes `same` and `drop`
          // false: the point searches), "pre" into the operands (distance and bound +inf: `drop` is true, nothing is kept or searched).
          double lbp;
          asm("v_max_f64 %0, %1, 0" : "=v"(lbp) : "v"(lb[q]));      // (fmax() brings a canonicalising v_max along)
          const double inf = __builtin_inf();
          const double lb2 = pre ? lbp * lbp : inf;
          const double de = pre ? d : inf;
          const bool same = de < lb2;                              // neighbour proven
          const bool le = de <= thr;                               // DistanceFilter
          const bool drop = (lb2 > thr) & !le;                     // no pair whoever the neighbour is
          bd[q] = de;
          keep[q] = same & le;
          need[q] = !(same | drop);
        }
        ent[q] = -1;
      }
      // Bounds only ever decay, and a point whose slack runs out costs a work-list pass however few such points there are in that step.
      // Two scheduled passes (steps 6 and 15) renew every bound with less than 6x slack in distance instead of a trickle of passes later
      // (round 3, per-step search profile: with 2x slack and steps 6 / 12, 13-15 of the last 17 steps still had 1-60 searching points --
      // good pairs at 3 m range whose slack of a few millimetres the scene's remaining motion eats; a third renewal buys nothing).
      // Formed here, in the two steps that renew, from what the loop above left -- bd is +inf for a point outside the bounds or an empty
      // register slot, which makes `drop` true.
      if (UNLIKELY(refresh)) {
#pragma unroll
        for (int q = 0; q < R; q++) {
          if (q >= Rn) continue;
          const double lbp = fmax(lb[q], 0.0), lb2 = bd[q] < __builtin_inf() ? lbp * lbp : __builtin_inf();
          const bool drop = (lb2 > thr) & !(bd[q] <= thr);
          const bool weak = !drop & (lb2 < WEAK_MULT * bd[q]) & (lb2 > 0.0);
          need[q] = need[q] | weak;
          keep[q] = keep[q] & !weak;
        }
      }
    }
    TL(2);                                     // tier 0 decided
    // -- ReciprocalFilter, first half: per model slot the smallest d2 (LDS atomic min on the bit pattern).
    // Pairs settled by tier 0 go in right away; two scene points with the SAME d2 to one slot are the only
    // case that needs the index round below, and the later of the two sees its own value come back.
    bool tie = false;
    {
      // (the three returning atomics leave together and are taken delivery of once: consumed inside its branch, each one was
      // waited for on the spot -- three LDS round trips in a row, 760 cycles of the step: profiles/r4_icp_critical_path.txt)
      unsigned long long mine[R], was[R];
#pragma unroll
      for (int q = 0; q < R; q++) { mine[q] = (unsigned long long)__double_as_longlong(bd[q]); was[q] = ~0ull; }
#pragma unroll
      for (int q = 0; q < R; q++)
        if (keep[q]) was[q] = atomicMin(&slotD[hint[q]], mine[q]);
      // work list of the points that need a search (an entry beyond the list capacity waits for its pass)
      // (ONE test per wave and step instead of one per register slot: a conditional branch costs a wave 15-30 cycles even when it is not
      // taken, an exec-masked region about as much -- tools/exp/ctl.hip -- and here the wait for the atomics above covers it)
      bool need_any = false;
#pragma unroll
      for (int q = 0; q < R; q++) need_any |= need[q];
      if (__builtin_expect(__ballot(need_any) != 0ull, 0))
#pragma unroll
      for (int q = 0; q < R; q++)
        if (need[q]) {
          ent[q] = atomicAdd(&L.ired[IR_CNT], 1);
          if (ent[q] < lcap) { L.list_xy[ent[q]] = make_double2(sx[q], sy[q]); L.list_k[ent[q]] = hint[q] | (lb[q] == LB_WINDOW_TRIED ? LIST_PAST_WINDOW : 0); }
        }
#pragma unroll
      for (int q = 0; q < R; q++) tie |= keep[q] & (was[q] == (unsigned long long)__double_as_longlong(bd[q]));
    }
    if (UNLIKELY(tie)) L.ired[IR_TIE] = 1;
    TL(3);                                     // the reciprocal filter's atomics are back
    __syncthreads();
    TL(4);                                     // past barrier 1
    // One LDS round trip for everything the step needs behind barrier 1: the work-list counter, the tie flag and -- speculatively, they
    // are final only when nobody searches, which is the steady state -- the slot minima.
    unsigned long long sd[R];
    int tie_any = 0;
    int n_need;
    {
      int nn = L.ired[IR_CNT];
      tie_any = L.ired[IR_TIE];
#pragma unroll
      for (int q = 0; q < R; q++) sd[q] = slotD[hint[q]];
      asm volatile("" : "+v"(nn), "+v"(tie_any));                 // (all five reads issued before the first is waited for)
      n_need = nn;
    }
    if (lcap == cap) __builtin_assume(n_need <= lcap);           // (no more points than capacity: a list that holds them all needs one pass)
    if (UNLIKELY(n_need > 0)) {
      tie = false;
      for (int base = 0; base < n_need; base += lcap) {       // one pass unless more than lcap points search
        const int n = (n_need - base) < lcap ? (n_need - base) : lcap;
        if (base > 0) {
#pragma unroll
          for (int q = 0; q < R; q++)
            if (need[q] && ent[q] >= base && ent[q] < base + lcap) {
              L.list_xy[ent[q] - base] = make_double2(sx[q], sy[q]);
              L.list_k[ent[q] - base] = hint[q] | (lb[q] == LB_WINDOW_TRIED ? LIST_PAST_WINDOW : 0);
            }
          __syncthreads();
        }
        for (int e0 = wave * 64; e0 < n; e0 += T) {
          const int e = e0 + lane;
          if (e < n) {
            const double2 s = L.list_xy[e];
            const int lk = L.list_k[e];
            NnResult r; r.resolved = false;
#ifdef TSD_ICP_TIMELINE
            int rounds = 0;
            long long st[2] = {0, 0};
            const long long ws0 = clock64();
            { double t0 = s.x + (double)lk; asm volatile("" : "+v"(t0)); }
            const long long ws1 = clock64();
            if (!(lk & LIST_PAST_WINDOW)) r = window_search(L, nM, s.x, s.y, lk, thr, sgn, &rounds, st);
            asm volatile("" : "+v"(r.best), "+v"(r.lbsq));
            const long long ws2 = clock64();
            if (L.tail->trace && iter < 16u && !(lk & LIST_PAST_WINDOW)) {
              // (diagnostic, cumulative over every registration of the run) what the window pass's entries were: rounds 1..7 x {resolved with a
              // partner within the filter distance, resolved without one, not resolved -> whole-wave search}, by step
              const int rb = (rounds < 1 ? 1 : (rounds > 7 ? 7 : rounds)) - 1;
              const int cat = rb + 7 * (r.resolved ? (r.best <= thr ? 0 : 1) : 2);
              atomicAdd(L.tail->trace + TSD_ICP_TRACE_STRIDE * (TSD_ICP_TRACE_MAX + 202) + 21 * (int)iter + cat, 1.0);
            }
            {
              // (diagnostic) per list round of wave 0: entries, lanes' largest / mean number of window rounds, cycles of the call
              int mx = rounds;
              for (int o = 32; o; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
              int sm = rounds;
              for (int o = 32; o; o >>= 1) sm += __shfl_xor(sm, o);
              if (has_trace && wave == 0 && lane == 0 && iter < 32u) {
                double* dd = L.tail->trace + TSD_ICP_TRACE_STRIDE * (TSD_ICP_TRACE_MAX + 128) + 8 * iter;
                if (e0 == 0) { dd[0] = (double)n; dd[1] = (double)mx; dd[2] = (double)sm; dd[3] = (double)(ws2 - ws0); dd[4] = (double)__popcll(__ballot(true));
                               dd[5] = (double)(ws1 - tlbuf[((iter - TL_FIRST) * W) * TL_N + 4]); dd[6] = (double)(st[0] - ws1); dd[7] = (double)(st[1] - st[0]); }
              }
            }
#else
            if (!(lk & LIST_PAST_WINDOW)) r = window_search(L, nM, s.x, s.y, lk, thr, sgn);
#endif
            if (r.resolved) {
              L.res_d[e] = r.best; L.res_k[e] = r.bk; L.res_k2[e] = r.bk2; L.res_lb[e] = lb_from_sq(r.lbsq);
              // DistanceFilter + the reciprocal filter's first half for this entry, by the lane that searched: the slot minima are final
              // behind the barrier that ends the pass, and the owners only read (they used to issue these atomics themselves behind
              // the results -- one more LDS round trip and one more barrier in every step that searches)
              if (r.best <= thr) {
                const unsigned long long mine = (unsigned long long)__double_as_longlong(r.best);
                if (atomicMin(&slotD[r.bk], mine) == mine) L.ired[IR_TIE] = 1;
              }
            }
            else L.list2[atomicAdd(&L.ired[IR_CNT2], 1)] = e;      // tier 2, shared out over all waves below
          }
        }
        __syncthreads();
        TL(13);                                // (search step) the window pass is done
        const int n2 = L.ired[IR_CNT2];
        if (n2 > 0) {
          for (int i = wave; i < n2; i += W) {
            const int es = L.list2[i];
            const double2 s = L.list_xy[es];
            const NnResult r = wave_search(L, nM, s.x, s.y, L.list_k[es] & ~LIST_PAST_WINDOW, thr, sgn, lane);
            if (lane == 0) {
              L.res_d[es] = r.best; L.res_k[es] = r.bk; L.res_k2[es] = r.bk2; L.res_lb[es] = lb_from_sq(r.lbsq);
              if (r.bk >= 0 && r.best <= thr) {
                const unsigned long long mine = (unsigned long long)__double_as_longlong(r.best);
                if (atomicMin(&slotD[r.bk], mine) == mine) L.ired[IR_TIE] = 1;
              }
            }
          }
          __syncthreads();
          if (tid == 0) L.ired[IR_CNT2] = 0;
        }
        TL(14);                                // (search step) the whole-wave searches are done
#pragma unroll
        for (int q = 0; q < R; q++)
          if (need[q] && ent[q] >= base && ent[q] < base + lcap) {
            const int e = ent[q] - base;
            const int k = L.res_k[e];
            if (k >= 0) {
              bd[q] = L.res_d[e]; hint[q] = k; hint2[q] = L.res_k2[e]; lb[q] = L.res_lb[e];
              mw[q] = L.mxy[k];
              keep[q] = bd[q] <= thr;                                  // DistanceFilter::filter (its slot entry: made by the searching lane)
            } else { bd[q] = __builtin_inf(); lb[q] = -1.0; }          // non-finite input point
          }
        if (base + lcap < n_need) __syncthreads();
      }
      // (point-to-line: the pair sums' transpose buffer ALIASES the work list -- the results must have been read before the first wave
      // gets there; the closed form reduces in registers and needs no barrier here)
      if constexpr (PTL) __syncthreads();
      tie_any = L.ired[IR_TIE];                                    // the searches' pairs went into the slots (behind the passes' barriers): read again
#pragma unroll
      for (int q = 0; q < R; q++) sd[q] = slotD[hint[q]];
    }
    // threshold schedule (DistanceFilter.cpp:62-63)
    thr *= thr_mult;
    if (thr < thr_min) thr = thr_min;

    // -- ReciprocalFilter, second half: the pair whose d2 stands in its slot wins
    bool win[R];
#pragma unroll
    for (int q = 0; q < R; q++) win[q] = keep[q] & (sd[q] == (unsigned long long)__double_as_longlong(bd[q]));
    TL(5);                                     // winners known (work-list counter + slot minima read)
    if (UNLIKELY(tie_any)) {
      // equal d2 somewhere: the lowest scene index of the candidates wins its slot
#pragma unroll
      for (int q = 0; q < R; q++)
        if (win[q]) atomicMin(&L.slotI[hint[q]], pid[q]);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < R; q++) {
        if (win[q]) { win[q] = L.slotI[hint[q]] == pid[q]; }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < R; q++)
        if (keep[q]) L.slotI[hint[q]] = INT_MAX;
      if (tid == 0) L.ired[IR_TIE] = 0;
    }
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < R; q++) cnt += __popcll(__ballot(win[q]));
    if constexpr (PAIRS) {
#pragma unroll
      for (int q = 0; q < R; q++)
        if (win[q]) pairs_out[(size_t)iter * (size_t)cap + (size_t)hint[q]] = pid[q];
    }

    // -- phase D/F: ClosedFormEstimator2D::setPairs + estimateTransformation in ONE pass over the pairs.
    // The reference centres the pairs on their centroids (two passes).  Centring on the previous
    // step's centroids c0 instead and correcting, sum (a-ca)(b-cb) = sum (a-c0a)(b-c0b) - n (ca-c0a)(cb-c0b),
    // is the same quantity with the same conditioning (c0 is within millimetres of c; the very first step
    // centres on the scene's centroid, see c0).
    constexpr bool RED8 = !PTL;                         // closed form: seven sums + the pair count, reduced in registers (block_totals8)
    constexpr int NSUM = PTL ? NSUM_PTL : (RED8 ? 8 : NSUM_CF);
    double tot[NSUM];
    {
      double v[NSUM];
#pragma unroll
      for (int k = 0; k < NSUM; k++) v[k] = 0.0;
      if constexpr (RED8) {
        // the estimator's nominator / denominator terms accumulated per pair (7 sums instead of 9: what the reduction pays per value
        // is what counts now), the lane's pair count as the eighth value (exact in fp64)
#pragma unroll
        for (int q = 0; q < R; q++) {
          if (q >= Rn) continue;                // (an empty register slot: skipped by a scalar branch, not run through with no lane active)
          if (LIKELY(win[q])) {
            const double2 m = mw[q];
            v[0] += m.x; v[1] += m.y; v[2] += sx[q]; v[3] += sy[q];
            v[4] += bd[q];                    // (the pair's squared distance: the same expression on the same operands as tier 0's / the search's)
            const double xF = m.x - c0[0], yF = m.y - c0[1], xS = sx[q] - c0[2], yS = sy[q] - c0[3];
            v[5] += yF * xS - xF * yS; v[6] += xF * xS + yF * yS;
            v[7] += 1.0;
          }
        }
      } else {
        // PointToLine2DEstimator::setPairs + estimateTransformation (PointToLineEstimator2D.cpp:52-132): per pair
        // a_z = p_x n_y - p_y n_x, A += (a_z, n_x, n_y)(a_z, n_x, n_y)^T, b -= (a_z, n_x, n_y) ((p - q).n),
        // "rms" += |(p - q).n|, with p the scene point, q the model point, n the model normal
        double2 nw[R];
#pragma unroll
        for (int q = 0; q < R; q++) nw[q] = L.nxy[hint[q]];
#pragma unroll
        for (int q = 0; q < R; q++) {
          if (win[q]) {
            const double2 m = mw[q], n = nw[q];
            const double az = sx[q] * n.y - sy[q] * n.x;
            const double tmp = (sx[q] - m.x) * n.x + (sy[q] - m.y) * n.y;
            v[0] += az * az; v[1] += az * n.x; v[2] += az * n.y;
            v[3] += n.x * n.x; v[4] += n.x * n.y; v[5] += n.y * n.y;
            v[6] -= az * tmp; v[7] -= n.x * tmp; v[8] -= n.y * tmp;
            v[NSUM - 1] += fabs(tmp);
          }
        }
      }
      TL(6);                                   // this thread's pair sums
#ifdef TSD_ICP_TIMELINE
      long long* const tl_row = (iter >= (unsigned)TL_FIRST && iter < (unsigned)(TL_FIRST + TL_STEPS)) ? tlbuf + ((iter - TL_FIRST) * W + wave) * TL_N : nullptr;
#else
      long long* const tl_row = nullptr;
#endif
      if constexpr (RED8) { block_totals8<MAXT / 64>(L, v, tot, lane, wave, tl_row); pairs = (int)tot[7]; }
      else block_totals<MAXT / 64, NSUM>(L, v, cnt, tot, pairs, tid, lane, wave, W, tl_row);
      TL(10);                                  // totals in registers
    }
    // everybody is past the winner test: give the touched slots back, clear the work list counter
    if (slot_halves == 2) {
      // (the whole half, every thread its share at constant addresses: no per-point test, no address arithmetic; the half is used again
      // two barriers from here)
#pragma unroll
      for (int k = 0; k < (cap + T - 1) / T; k++) slotD[tid + k * T] = ~0ull;
    } else {
#pragma unroll
      for (int q = 0; q < R; q++)
        if (keep[q]) slotD[hint[q]] = ~0ull;
    }
    if (tid == 0) L.ired[IR_CNT] = 0;

    double co = __builtin_nan(""), si = __builtin_nan(""), dX = __builtin_nan(""), dY = __builtin_nan("");   // Tlast of this step (trace)
    pairs = __builtin_amdgcn_readfirstlane(pairs);
    if (LIKELY(pairs > 2)) {
      if constexpr (PTL) {
        // PointToLine2DEstimator: Matrix::solve = gsl_linalg_LU_decomp + LU_solve (gsl/Matrix.cpp:343-355);
        // psi = x[0] (cos, sin by libm like the reference), t = (x[1], x[2]); "rms" = mean |n.(p - q)|
        rms = tot[NSUM - 1] / (double)pairs;
        const double A[9] = {tot[0], tot[1], tot[2], tot[1], tot[3], tot[4], tot[2], tot[4], tot[5]};
        const double bb[3] = {tot[6], tot[7], tot[8]};
        double xs[3];
        d_lu3_solve(A, bb, xs);
        co = cos(xs[0]); si = sin(xs[0]); dX = xs[1]; dY = xs[2];
      } else {
        const double np = (double)pairs;
        const double size_inv = 1.0 / np;
        rms = tot[4] * size_inv;
        const double cmx = tot[0] * size_inv, cmy = tot[1] * size_inv, csx = tot[2] * size_inv, csy = tot[3] * size_inv;
        const double emx = cmx - c0[0], emy = cmy - c0[1], esx = csx - c0[2], esy = csy - c0[3];
        const double nom = tot[5] - np * (emy * esx - emx * esy);
        const double den = tot[6] - np * (emx * esx + emy * esy);
        c0[0] = cmx; c0[1] = cmy; c0[2] = csx; c0[3] = csy;
        // every wave evaluates the closed form itself (wave-uniform inputs): no broadcast barrier
#ifdef TSD_ICP_EXACT_TRIG
        { const double th_ = atan2(nom, den); co = cos(th_); si = sin(th_); }
#else
        // cos(atan2(n, d)) = d / hypot, sin = n / hypot: same angle without three libm calls; differs
        // from the reference's atan2 -> cos/sin by rounding only (DESIGN.md "ICP", tolerance 1e-4)
        {
          const double h2 = nom * nom + den * den;
          if (LIKELY(h2 > 0.0)) { const double inv = rsqrt(h2); co = den * inv; si = nom * inv; }
          else { co = signbit(den) ? -1.0 : 1.0; si = 0.0; }
        }
#endif
        dX = (cmx - (co * csx - si * csy));
        dY = (cmy - (co * csy + si * csx));
      }
      TL(11);                                  // closed form done
      // applyTransformation(sceneTmp): data * R^T (dgemm NoTrans,Trans), then + t (Icp.cpp:371-408).
      // The distance each point moves (rounded up, fp32 is plenty for a bound) eats into its neighbour
      // bound of tier 0.
#pragma unroll
      for (int q = 0; q < R; q++) {
        if (q >= Rn || PAIRS) continue;          // (PAIRS: a static scene, determinePairs called again and again)
        const double x = sx[q], y = sy[q];
        // applyTransformation's ((0 + x*co) + y*(-si)) + dX without the leading zero: the same value unless x*co is -0, and then only the
        // sign of a zero coordinate differs -- which no distance, comparison or sum can see (two instructions per point and step less)
        double nx = x * co + y * (-si), ny = x * si + y * co;
        nx = nx + dX; ny = ny + dY;
        const double ex = nx - x, ey = ny - y;
        const float disp = __builtin_amdgcn_sqrtf((float)(ex * ex + ey * ey) * 1.000001f) * 1.000001f;
        lb[q] = lb[q] - (double)disp;
        sx[q] = nx; sy[q] = ny;
      }
      {
        // Tfinal = Tlast * Tfinal (Icp.cpp:452): the 4x4 product restricted to its non-trivial entries
        // (the dropped terms are exact zeros / ones, so the rounding is the dgemm's up to the sign of a zero: the leading `0 +` of each
        // accumulation is left out like in the points' transform above)
        // (the translation column in every wave -- the bounds test of the next step reads it --, the rotation block in wave 0 alone: it is
        // needed behind the loop only, where wave 0 hands it to the others; 16 instructions per step less in seven waves)
        const double n02 = (co * Tf[2] + (-si) * Tf[5]) + dX, n12 = (si * Tf[2] + co * Tf[5]) + dY;
        if (wave == 0) {
          const double n00 = co * Tf[0] + (-si) * Tf[3], n01 = co * Tf[1] + (-si) * Tf[4];
          const double n10 = si * Tf[0] + co * Tf[3], n11 = si * Tf[1] + co * Tf[4];
          Tf[0] = n00; Tf[1] = n01; Tf[3] = n10; Tf[4] = n11;
        }
        Tf[2] = n02; Tf[5] = n12;
      }
      state = TSD_ICP_PROCESSING;
    } else {
      state = TSD_ICP_NOTMATCHABLE;
    }
    TL(12);                                    // scene moved, bounds updated
    // -- loop control (Icp.cpp:489-511)
    iter++;
    {
      // (every lane holds the same rms: the decisions as wave-uniform scalars, so that the loop is a scalar branch and not an
      // exec-mask loop over a per-lane `state`)
      // (ONE hand-over from the vector to the scalar unit -- ~30 cycles each -- for "converged or exactly matched"; the counters behind it)
      const bool hit = fabs(rms - rms_prev) < 10e-10, done = rms <= 0.0;
      state = __builtin_amdgcn_readfirstlane(state);
      if (UNLIKELY(__ballot(hit | done) != 0ull)) {
        const int conv_hit = __ballot(hit) != 0ull, rms_done = __ballot(done) != 0ull;
        conv_cnt = conv_hit ? conv_cnt + 1 : 0;
        if (rms_done || conv_cnt >= conv_need) state = TSD_ICP_SUCCESS;
        else if (iter >= max_it) state = TSD_ICP_MAXITERATIONS;
      } else {
        conv_cnt = 0;
        if (iter >= max_it) state = TSD_ICP_MAXITERATIONS;
      }
    }
    rms_prev = rms;
    if (UNLIKELY(has_trace && tid == 0 && iter <= TSD_ICP_TRACE_MAX)) {
      double* tr = L.tail->trace + TSD_ICP_TRACE_STRIDE * (iter - 1);
      tr[0] = (double)pairs; tr[1] = rms; tr[2] = thr_before; tr[3] = (double)state;
      tr[4] = co; tr[5] = si; tr[6] = dX; tr[7] = dY;          // Tlast = [[co, -si, dX], [si, co, dY]] (NaN: no estimate this step)
    }
  }

#ifdef TSD_ICP_TIMELINE
  tk_loop1 = clock64();
  __syncthreads();
  if (L.tail->trace) {
    if constexpr (TL_STEPS <= 4) {
      for (int i = tid; i < TL_STEPS * W * TL_N; i += T)
        L.tail->trace[TSD_ICP_TRACE_STRIDE * TSD_ICP_TRACE_MAX + i] = (double)(tlbuf[i] - tlbuf[0]);
    } else {                                    // many steps: wave 0's stamps only
      for (int i = tid; i < TL_STEPS * TL_N; i += T)
        L.tail->trace[TSD_ICP_TRACE_STRIDE * TSD_ICP_TRACE_MAX + i] = (double)(tlbuf[((i / TL_N) * W) * TL_N + (i % TL_N)] - tlbuf[0]);
    }
  }
#endif
  if (W > 1) {
    // the rotation block of Tfinal from wave 0 (see the loop): through the reduction rows, idle by now
    __syncthreads();
    if (tid == 0) { L.red[0] = Tf[0]; L.red[1] = Tf[1]; L.red[2] = Tf[3]; L.red[3] = Tf[4]; }
    __syncthreads();
    Tf[0] = L.red[0]; Tf[1] = L.red[1]; Tf[3] = L.red[2]; Tf[4] = L.red[3];
  }
  {
    // Icp::getFinalTransformation (Icp.cpp:528-546)
    IcpResultDev r;
    r.T[0] = Tf[0]; r.T[1] = Tf[1]; r.T[2] = Tf[2];
    r.T[3] = Tf[3]; r.T[4] = Tf[4]; r.T[5] = Tf[5];
    r.T[6] = 0.0; r.T[7] = 0.0; r.T[8] = 1.0;
    r.rms = rms; r.pairs = pairs; r.iterations = (int)iter; r.state = state;
    r.n_model = nM; r.n_scene = nS; r.reserved = L.ired[IR_SEEDED];
    const IcpTail& tl = *L.tail;
    if (tid == 0) *tl.out = r;
    // fused scan: gates, Sensor::transform, push / next-scan arguments, result record for the host
#ifdef TSD_ICP_TIMELINE
    long long tke[6] = {0, 0, 0, 0, 0, 0};
    if (tl.post.st) scan_post_body(tl.post, L.tail->pre, r.T, r, tl.post.gmin_x, tl.post.gmax_x, tl.post.gmin_y, tl.post.gmax_y, tke);
    if (tid == 0 && tl.trace) {
      double* pe = tl.trace + TSD_ICP_TRACE_STRIDE * (TSD_ICP_TRACE_MAX + 193);
      pe[0] = (double)(tke[0] - tk_loop1); for (int i = 1; i < 6; i++) pe[i] = (double)(tke[i] - tke[i - 1]);
      pe[6] = (double)(clock64() - tke[5]);
    }
#else
    if (tl.post.st) scan_post_body(tl.post, L.tail->pre, r.T, r, tl.post.gmin_x, tl.post.gmax_x, tl.post.gmin_y, tl.post.gmax_y);
#endif
#ifdef TSD_ICP_TIMELINE
    if (tid == 0 && tl.trace) {
      double* ph = tl.trace + TSD_ICP_TRACE_STRIDE * (TSD_ICP_TRACE_MAX + 192);
      const long long tk_end = clock64();
      ph[4] = (double)(tk_s[0] - tk_entry); ph[5] = (double)(tk_s[1] - tk_s[0]); ph[6] = (double)(tk_s[2] - tk_s[1]); ph[7] = (double)(tk_seed - tk_s[2]);
      ph[0] = (double)(tk_seed - tk_entry); ph[1] = (double)(tk_loop0 - tk_seed); ph[2] = (double)(tk_loop1 - tk_loop0); ph[3] = (double)(tk_end - tk_loop1);
    }
#endif
  }
}

template <int R, int MAXT, bool PTL, int FCAP = 0, int FT = 0>
__global__ void __launch_bounds__(MAXT)
k_icp(IcpArgs a, const double* __restrict__ P_dev, int cap, const double* __restrict__ g_model, const double* __restrict__ g_scene,
      const int* __restrict__ g_morig, const int* __restrict__ g_start,
      const double* __restrict__ g_coords, const uint8_t* __restrict__ g_mask_m,
      const double* __restrict__ g_rays_local, const double* __restrict__ g_ranges,
      const uint8_t* __restrict__ g_mask, IcpResultDev* __restrict__ out, double* __restrict__ trace, ScanPostArgs post,
      const double* __restrict__ g_mnormals, const double* __restrict__ g_normals, IcpSeedArgs seed)
{
  // workgroup 0 registers; workgroups 1 .. seed.helpers do step 0's searches for it
  icp_workgroup<R, MAXT, PTL, false, FCAP, FT>(a, P_dev, cap, g_model, g_scene, g_morig, g_start, g_coords, g_mask_m, g_rays_local, g_ranges, g_mask, out, trace,
                              post, g_mnormals, g_normals, seed, (int)blockIdx.x);
}

// registration_mode 3 inside the fused scan: workgroup 0 is the pre-registration's ARG-MAX (its own kernel, k_pdf_argmax, everywhere else:
// 5.2 us + a kernel boundary between the scoring and the registration), workgroup 1 registers, workgroups 2 .. are step 0's helpers.
// The registering workgroup's set-up needs Tinit ~2.5 us in, about when the arg-max has it.
template <int R, int MAXT, int FCAP, int FT>
__global__ void __launch_bounds__(MAXT)
k_icp_pre(IcpArgs a, const double* __restrict__ P_dev, int cap, const double* __restrict__ g_coords, const uint8_t* __restrict__ g_mask_m,
          const double* __restrict__ g_rays_local, const double* __restrict__ g_ranges, const uint8_t* __restrict__ g_mask,
          IcpResultDev* __restrict__ out, double* __restrict__ trace, ScanPostArgs post, IcpSeedArgs seed, IcpPreArgs pre)
{
  if (blockIdx.x == 0) {
    pdf_argmax_body<3>(pre.am.prob, pre.am.cand, pre.am.max_cand, pre.am.M, pre.am.S, pre.am.out, pre.am.hdr, pre.am.host_hdr, pre.am.host_res, pre.flag, pre.seq);
    return;
  }
  icp_workgroup<R, MAXT, false, false, FCAP, FT, true>(a, P_dev, cap, nullptr, nullptr, nullptr, nullptr, g_coords, g_mask_m, g_rays_local, g_ranges, g_mask, out, trace,
                                                     post, nullptr, nullptr, seed, (int)blockIdx.x - 1, nullptr, &pre);
}

// the same kernel with the per-step pair lists written out and the scene held still (direct mode, closed form): tsd_icp_pairs
template <int R, int MAXT>
__global__ void __launch_bounds__(MAXT)
k_icp_pairs(IcpArgs a, int cap, const double* __restrict__ g_model, const double* __restrict__ g_scene,
            const int* __restrict__ g_morig, const int* __restrict__ g_start, IcpResultDev* __restrict__ out, double* __restrict__ trace,
            int* __restrict__ pairs_out)
{
  ScanPostArgs post{};
  icp_workgroup<R, MAXT, false, true>(a, nullptr, cap, g_model, g_scene, g_morig, g_start, nullptr, nullptr, nullptr, nullptr, nullptr, out, trace,
                                      post, nullptr, nullptr, IcpSeedArgs{nullptr, 0u, 0, 0}, 0, pairs_out);
}

// the registrations of a batch of robots in ONE launch (tsd_batch_begin): workgroup x = entry x, fused mode only (model and
// scene come from the ray cast's / the scan's per-beam arrays); each registration still runs on one compute unit
template <int R, int MAXT, bool PTL, int FCAP = 0, int FT = 0>
__global__ void __launch_bounds__(MAXT)
k_icp_batch(const IcpBatchEntry* __restrict__ entries, int cap, int n_entries)
{
  // workgroups 0 .. n - 1 register entry x; workgroup n * h + x (h >= 1) is helper h of entry x
  const IcpBatchEntry& e = entries[blockIdx.x % (unsigned)n_entries];
  const int role = (int)(blockIdx.x / (unsigned)n_entries);
  if (e.rc_flag) {
    // launched ahead of the batch's ray casts: wait until the word behind them says this batch's are done (one thread polls).
    // Bounded -- and a wait that runs out, or a batch the host abandoned, does NOT register on whatever the ray-cast buffers hold:
    // the workgroup reports why (reserved = BATCH_FAIL_*), leaves pose and grid alone and publishes the sequence numbers, so the
    // host gets an error instead of a pose (tsd_batch_results -> TSD_E_HIP).
    __shared__ int s_fail;
    if (threadIdx.x == 0) {
      unsigned int polls = 0u;
      int fail = 0;
      for (;;) {
        if ((int)(__hip_atomic_load(e.rc_flag + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - e.rc_target) >= 0) { fail = BATCH_FAIL_ABORTED; break; }
        if ((int)(__hip_atomic_load(e.rc_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - e.rc_target) >= 0) break;
        if (++polls >= e.poll_bound) { fail = BATCH_FAIL_TIMEOUT; break; }
        __builtin_amdgcn_s_sleep(32);
      }
      s_fail = fail;
    }
    __syncthreads();
    if (s_fail) {
      if (threadIdx.x == 0 && role == 0) scan_post_failed(e.post, s_fail);
      return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (role > e.seed.helpers) return;
  icp_workgroup<R, MAXT, PTL, false, FCAP, FT>(e.a, e.P_dev, cap, nullptr, nullptr, nullptr, nullptr, e.coords, e.mask_m, e.rays_local, e.ranges, e.mask, e.out,
                              e.trace, e.post, nullptr, e.normals, e.seed, role);
}

static int icp_cap_for(int n)
{
  int cap = (n + 63) & ~63;
  if (cap < 64) cap = 64;
  return cap;
}

// helper workgroups of a registration with n scene points in workgroups of T threads (see ICP_HELPER_POINTS)
static std::atomic<unsigned int> g_seed_seq{1u};
int icp_helpers_for(const tsd_ctx* ctx, int n, int T)
{
  if (!ctx->icp_helpers) return 0;
  const int per = T < ICP_HELPER_POINTS ? T : ICP_HELPER_POINTS;
  const int h = (n + per - 1) / per;
  return h > ICP_MAX_HELPERS ? 0 : h;                // (more points than the helpers reach: the registration searches itself)
}
size_t icp_seed_bytes(int points) { return 2 * sizeof(unsigned long long) * (size_t)((points + 63) & ~63); }
IcpSeedArgs icp_seed_args(void* buf, int points, int helpers)
{
  IcpSeedArgs sa;
  sa.g = reinterpret_cast<unsigned long long*>(buf);
  sa.stride = (points + 63) & ~63;                   // the second granules follow the first (as laid out by icp_seed_bytes(points))
  unsigned int q = g_seed_seq.fetch_add(1u);
  if (q == 0u) q = g_seed_seq.fetch_add(1u);         // (0 is what a fresh buffer holds)
  sa.seq = q; sa.helpers = buf ? helpers : 0;
  return sa;
}

// threads of a registration of n points with R register slots per lane (four "old" waves take R blocks of 64 points each,
// younger waves one block each where the blocks left allow it)
static int icp_threads_for(int n, int R, int maxt)
{
  int T = ((n + R - 1) / R + 63) & ~63;
  if (T < 64) T = 64;
  const int B = (n + 63) / 64;
  if (B > 4 * R) {
    int W = 4 + (B - 4 * R);
    if (W > maxt / 64) W = maxt / 64;
    if (64 * W > T) T = 64 * W;
  }
  return T;
}

// workgroup shape: R scene points per thread, T threads.  One CU runs the whole registration and is
// issue bound, so few waves (per-wave reduction / control cost paid once per SIMD) win.
template <int R, int MAXT, bool PTL>
static int launch_icp_shape_est(tsd_ctx* ctx, const IcpArgs& a, int n, int cap, const double* P_dev,
                            const double* d_rays_local, const double* d_ranges, const uint8_t* d_mask,
                            const ScanPostArgs& post, int force_T = 0, const IcpPreLaunch* pre = nullptr)
{
  int T = icp_threads_for(n, R, MAXT);
  if (force_T > T) T = force_T;
  if (T > MAXT) return set_error(ctx, TSD_E_CAPACITY, "icp workgroup shape", hipSuccess);
  const bool ptl = a.estimator == TSD_ESTIMATOR_POINT_TO_LINE;
  const size_t lds = icp_lds_bytes_for(cap, T, ptl) + ICP_TL_BYTES;
  if (lds > 160u * 1024u) return set_error(ctx, TSD_E_CAPACITY, "registration does not fit the LDS of one CU (point-to-line: model normals too)", hipSuccess);
  {
    // the attribute is per device: remembered per context (and kernel instantiation), not per process
    std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
    size_t& configured = ctx->lds_configured[reinterpret_cast<const void*>(k_icp<R, MAXT, PTL>)];
    if (lds > configured) {
      TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_icp<R, MAXT, PTL>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured = lds;
    }
  }
  ScopedKernelTimer t(ctx, "icp");
  const LaunchTarget* tg = launch_target();       // concurrent multi-robot path: the sensor's own stream and buffers
  // the per-iteration record (tsd_icp_trace) is kept by tsd_icp / tsd_localize; the fused scan has no reader for it and skips
  // the 64-byte store per step
  double* trace_buf = tg && tg->trace ? tg->trace : ctx->d_icp_trace;
  if (post.st) trace_buf = nullptr;
#ifdef TSD_ICP_TIMELINE
  if (post.st) trace_buf = ctx->d_icp_trace;     // (diagnostic build: the stamps of a fused scan's registration; no per-step record)
#endif
  const bool own_seed = tg && tg->icp_seed;
  const IcpSeedArgs sa = icp_seed_args(own_seed ? tg->icp_seed : ctx->d_icp_seed, own_seed ? tg->icp_seed_points : TSD_MAX_ICP_POINTS,
                                       icp_helpers_for(ctx, n, T));
  // The default scanner's shape (1081 beams: capacity 1088, 512 threads, closed form) has an instantiation of its own in which the
  // capacity and the thread count are compile-time constants: the twenty offsets of the LDS layout then cost no scalar registers
  // (spilled scalars of the loop 166 -> 56; -0.7 us per registration, profiles/r5_icp_helpers_ab.txt)
  if (pre) {
    // fused registration_mode 3, the node's shape (icp_pre_supported): the arg-max rides as the launch's first workgroup
    if (!(R == 3 && MAXT == 512 && !PTL && cap == 1088 && T == 512) || !post.st) return set_error(ctx, TSD_E_ARG, "launch_icp: the arg-max can only ride with the node's registration shape", hipSuccess);
    {
      std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
      size_t& configured = ctx->lds_configured[reinterpret_cast<const void*>(k_icp_pre<3, 512, 1088, 512>)];
      if (lds > configured) {
        TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_icp_pre<3, 512, 1088, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
      }
    }
    // (the launch's own completion is the event a pre-registration armed AHEAD waits for before it overwrites the inputs; a timed
    // dispatch needs its stop event for the timer: a marker behind it then)
    hipEvent_t stop = t.b ? t.b : pre->done;
    hipExtLaunchKernelGGL((k_icp_pre<3, 512, 1088, 512>), dim3(2 + sa.helpers), dim3(T), lds, launch_stream(ctx), t.a, stop, 0, a, P_dev, cap,
                     tg && tg->coords ? tg->coords : ctx->d_coords, tg && tg->mask_m ? tg->mask_m : ctx->d_mask_m,
                     d_rays_local ? d_rays_local : ctx->d_rays_local, d_ranges ? d_ranges : ctx->d_ranges,
                     d_mask ? d_mask : ctx->d_mask, tg && tg->icp_res ? tg->icp_res : ctx->d_icp_res, trace_buf, post, sa, pre->dev);
    TSD_HIP_CHECK(ctx, hipGetLastError());
    if (t.b && pre->done) TSD_HIP_CHECK(ctx, hipEventRecord(pre->done, launch_stream(ctx)));
    return TSD_OK;
  }
  if (R == 3 && MAXT == 512 && !PTL && cap == 1088 && T == 512) {
    {
      std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
      size_t& configured = ctx->lds_configured[reinterpret_cast<const void*>(k_icp<3, 512, false, 1088, 512>)];
      if (lds > configured) {
        TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_icp<3, 512, false, 1088, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
      }
    }
    hipExtLaunchKernelGGL((k_icp<3, 512, false, 1088, 512>), dim3(1 + sa.helpers), dim3(T), lds, launch_stream(ctx), t.a, t.b, 0, a, P_dev, cap, ctx->d_model, ctx->d_scene,
                     ctx->d_morig, ctx->d_start, tg && tg->coords ? tg->coords : ctx->d_coords, tg && tg->mask_m ? tg->mask_m : ctx->d_mask_m,
                     d_rays_local ? d_rays_local : ctx->d_rays_local, d_ranges ? d_ranges : ctx->d_ranges,
                     d_mask ? d_mask : ctx->d_mask, tg && tg->icp_res ? tg->icp_res : ctx->d_icp_res,
                     trace_buf, post, ctx->d_mnormals, tg && tg->normals ? tg->normals : ctx->d_normals, sa);
    TSD_HIP_CHECK(ctx, hipGetLastError());
    return TSD_OK;
  }
  hipExtLaunchKernelGGL((k_icp<R, MAXT, PTL>), dim3(1 + sa.helpers), dim3(T), lds, launch_stream(ctx), t.a, t.b, 0, a, P_dev, cap, ctx->d_model, ctx->d_scene,
                     ctx->d_morig, ctx->d_start, tg && tg->coords ? tg->coords : ctx->d_coords, tg && tg->mask_m ? tg->mask_m : ctx->d_mask_m,
                     d_rays_local ? d_rays_local : ctx->d_rays_local, d_ranges ? d_ranges : ctx->d_ranges,
                     d_mask ? d_mask : ctx->d_mask, tg && tg->icp_res ? tg->icp_res : ctx->d_icp_res,
                     trace_buf, post, ctx->d_mnormals, tg && tg->normals ? tg->normals : ctx->d_normals, sa);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

template <int R, int MAXT>
static int launch_icp_shape(tsd_ctx* ctx, const IcpArgs& a, int n, int cap, const double* P_dev,
                            const double* d_rays_local, const double* d_ranges, const uint8_t* d_mask,
                            const ScanPostArgs& post, int force_T = 0, const IcpPreLaunch* pre = nullptr)
{
  // the estimator is a compile-time choice: the node's closed form does not pay for the other one's tenth sum
  if (a.estimator == TSD_ESTIMATOR_POINT_TO_LINE)
    return launch_icp_shape_est<R, MAXT, true>(ctx, a, n, cap, P_dev, d_rays_local, d_ranges, d_mask, post, force_T, pre);
  return launch_icp_shape_est<R, MAXT, false>(ctx, a, n, cap, P_dev, d_rays_local, d_ranges, d_mask, post, force_T, pre);
}

// can the pre-registration's arg-max ride with this registration's launch (k_icp_pre: the node's shape, closed form, fused scan)?
bool icp_pre_supported(const tsd_ctx* ctx, const IcpArgs& a)
{
  if (a.beams < 1 || a.estimator != TSD_ESTIMATOR_CLOSED_FORM || ctx->icp_shape != 0) return false;
  return icp_cap_for(a.beams) == 1088 && icp_threads_for(a.beams, 3, 512) == 512;
}

int launch_icp(tsd_ctx* ctx, const IcpArgs& a, const double* P_dev, const double* d_rays_local,
               const double* d_ranges, const uint8_t* d_mask, const ScanPostArgs* post_in, const IcpPreLaunch* pre)
{
  ScanPostArgs post;
  std::memset(&post, 0, sizeof(post));
  if (post_in) post = *post_in;
  if (a.estimator != TSD_ESTIMATOR_CLOSED_FORM && a.estimator != TSD_ESTIMATOR_POINT_TO_LINE)
    return set_error(ctx, TSD_E_ARG, "tsd_icp_params.estimator", hipSuccess);
  const int n = a.beams > 0 ? a.beams : (a.n_model > a.n_scene ? a.n_model : a.n_scene);
  if (n > TSD_MAX_ICP_POINTS) return set_error(ctx, TSD_E_CAPACITY, "icp points > TSD_MAX_ICP_POINTS", hipSuccess);
  const int cap = icp_cap_for(n);
  const int nthr = a.beams > 0 ? a.beams : a.n_scene;     // scene points decide the thread count
  // (round 3: the experimental shapes <2,576>, <5,512> and <5,256> are gone -- measured no faster in round 2, and the first spilled
  // 21-27 registers per lane; TSD_ICP_SHAPE=8 forces the 8-points-per-thread shape, TSD_ICP_SHAPE >= 64 a thread count of <3,512>)
  if (ctx->icp_shape == 8) return launch_icp_shape<8, 256>(ctx, a, nthr, cap, P_dev, d_rays_local, d_ranges, d_mask, post);
  if (pre && !icp_pre_supported(ctx, a)) return set_error(ctx, TSD_E_ARG, "launch_icp: the arg-max can only ride with the node's registration shape", hipSuccess);
  if (nthr <= 3 * 512) return launch_icp_shape<3, 512>(ctx, a, nthr, cap, P_dev, d_rays_local, d_ranges, d_mask, post, ctx->icp_shape >= 64 ? ctx->icp_shape : 0, pre);
  return launch_icp_shape<8, 256>(ctx, a, nthr, cap, P_dev, d_rays_local, d_ranges, d_mask, post);
}

template <int R, int MAXT>
static int launch_icp_pairs_shape(tsd_ctx* ctx, const IcpArgs& a, int n, int cap, int* d_pairs)
{
  int T = icp_threads_for(n, R, MAXT);
  if (T > MAXT) return set_error(ctx, TSD_E_CAPACITY, "icp workgroup shape", hipSuccess);
  const size_t lds = icp_lds_bytes_for(cap, T, false);
  if (lds > 160u * 1024u) return set_error(ctx, TSD_E_CAPACITY, "registration does not fit the LDS of one CU", hipSuccess);
  {
    std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
    size_t& configured = ctx->lds_configured[reinterpret_cast<const void*>(k_icp_pairs<R, MAXT>)];
    if (lds > configured) {
      TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_icp_pairs<R, MAXT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured = lds;
    }
  }
  hipLaunchKernelGGL((k_icp_pairs<R, MAXT>), dim3(1), dim3(T), lds, ctx->stream, a, cap, ctx->d_model, ctx->d_scene, ctx->d_morig, ctx->d_start,
                     ctx->d_icp_res, ctx->d_icp_trace, d_pairs);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

// direct mode, closed form, the default workgroup shapes of launch_icp; d_pairs = [a.iterations][icp_pairs_cap(n)] ints preset to -1
int icp_pairs_cap(int n_model, int n_scene) { return icp_cap_for(n_model > n_scene ? n_model : n_scene); }
int launch_icp_pairs(tsd_ctx* ctx, const IcpArgs& a, int* d_pairs)
{
  const int n = a.n_model > a.n_scene ? a.n_model : a.n_scene;
  if (n > TSD_MAX_ICP_POINTS) return set_error(ctx, TSD_E_CAPACITY, "icp points > TSD_MAX_ICP_POINTS", hipSuccess);
  const int cap = icp_cap_for(n);
  if (a.n_scene <= 3 * 512) return launch_icp_pairs_shape<3, 512>(ctx, a, a.n_scene, cap, d_pairs);
  return launch_icp_pairs_shape<8, 256>(ctx, a, a.n_scene, cap, d_pairs);
}

template <int R, int MAXT, bool PTL>
static int launch_icp_batch_shape(tsd_ctx* ctx, hipStream_t stream, const IcpBatchEntry* d_entries, int n, int nthr, int cap, int helpers)
{
  int T = icp_threads_for(nthr, R, MAXT);
  if (T > MAXT) return set_error(ctx, TSD_E_CAPACITY, "icp workgroup shape", hipSuccess);
  const size_t lds = icp_lds_bytes_for(cap, T, PTL);
  if (lds > 160u * 1024u) return set_error(ctx, TSD_E_CAPACITY, "registration does not fit the LDS of one CU", hipSuccess);
  {
    std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
    size_t& configured = ctx->lds_configured[reinterpret_cast<const void*>(k_icp_batch<R, MAXT, PTL>)];
    if (lds > configured) {
      TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_icp_batch<R, MAXT, PTL>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured = lds;
    }
  }
  ScopedKernelTimer t(ctx, "icp");
  if (R == 3 && MAXT == 512 && !PTL && cap == 1088 && T == 512) {        // (the default scanner's shape: compile-time LDS layout, see launch_icp_shape_est)
    {
      std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
      size_t& configured = ctx->lds_configured[reinterpret_cast<const void*>(k_icp_batch<3, 512, false, 1088, 512>)];
      if (lds > configured) {
        TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_icp_batch<3, 512, false, 1088, 512>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
      }
    }
    hipExtLaunchKernelGGL((k_icp_batch<3, 512, false, 1088, 512>), dim3(n * (1 + helpers)), dim3(T), lds, stream, t.a, t.b, 0, d_entries, cap, n);
    TSD_HIP_CHECK(ctx, hipGetLastError());
    return TSD_OK;
  }
  hipExtLaunchKernelGGL((k_icp_batch<R, MAXT, PTL>), dim3(n * (1 + helpers)), dim3(T), lds, stream, t.a, t.b, 0, d_entries, cap, n);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

// every registration of a batch runs in the same workgroup shape (the widest sensor decides) and with the same estimator
int launch_icp_batch(tsd_ctx* ctx, hipStream_t stream, const IcpBatchEntry* host, const IcpBatchEntry* d_entries, int n)
{
  if (n < 1) return TSD_OK;
  int beams = 0;
  for (int i = 0; i < n; i++) {
    if (host[i].a.estimator != host[0].a.estimator) return set_error(ctx, TSD_E_ARG, "tsd_batch_begin: one estimator per batch", hipSuccess);
    if (host[i].a.beams < 1) return set_error(ctx, TSD_E_ARG, "tsd_batch_begin: fused registrations only", hipSuccess);
    if (host[i].a.beams > beams) beams = host[i].a.beams;
  }
  if (beams > TSD_MAX_ICP_POINTS) return set_error(ctx, TSD_E_CAPACITY, "icp points > TSD_MAX_ICP_POINTS", hipSuccess);
  const int cap = icp_cap_for(beams);
  const bool ptl = host[0].a.estimator == TSD_ESTIMATOR_POINT_TO_LINE;
  if (host[0].a.estimator != TSD_ESTIMATOR_CLOSED_FORM && !ptl) return set_error(ctx, TSD_E_ARG, "tsd_icp_params.estimator", hipSuccess);
  // (the entries' seed arguments were filled by the caller with icp_batch_helpers(): the largest helper count of the batch sizes the grid)
  int helpers = 0;
  for (int i = 0; i < n; i++) if (host[i].seed.helpers > helpers) helpers = host[i].seed.helpers;
  if (beams <= 3 * 512)
    return ptl ? launch_icp_batch_shape<3, 512, true>(ctx, stream, d_entries, n, beams, cap, helpers)
               : launch_icp_batch_shape<3, 512, false>(ctx, stream, d_entries, n, beams, cap, helpers);
  return ptl ? launch_icp_batch_shape<8, 256, true>(ctx, stream, d_entries, n, beams, cap, helpers)
             : launch_icp_batch_shape<8, 256, false>(ctx, stream, d_entries, n, beams, cap, helpers);
}

size_t icp_lds_bytes() { return icp_lds_bytes_for(TSD_MAX_ICP_POINTS, 256); }

// seed arguments of one entry of a batch (tsd_batch_begin): the workgroup shape launch_icp_batch will choose for `batch_beams`;
// `buf` was sized icp_seed_bytes(beams)
IcpSeedArgs icp_batch_seed_args(const tsd_ctx* ctx, void* buf, int beams, int batch_beams)
{
  const int T = batch_beams <= 3 * 512 ? icp_threads_for(batch_beams, 3, 512) : icp_threads_for(batch_beams, 8, 256);
  return icp_seed_args(buf, beams, icp_helpers_for(ctx, beams, T));
}

}  // namespace tsd
