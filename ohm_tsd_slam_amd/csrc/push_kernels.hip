// push_kernels.hip -- TsdGrid::push (TsdGrid.cpp:217-284) on gfx950:
//
//   k_push_tables   (one workgroup, off the critical path in the fused scan) range-query tables of the
//                   scan: the two beam-range tests of TsdGridComponent::isInRange become O(1) look-ups.
//   k_push_classify isInRange for every tile of the launch window, one LANE per tile (range cull, four corner
//                   back-projections, table look-ups): the reference's `#pragma omp for` over all partitions
//                   (TsdGrid.cpp:228-277) without its load imbalance.  Tiles that need a workgroup go to a
//                   work list (one atomic per wave) with the beam interval their cells can project to.
//   k_push_update   one 256-thread workgroup per listed tile: scan window staged in LDS, 4 cells per thread,
//                   row-major => coalesced 8-byte RMW; beam index from a fp32 estimate proven by two fp64
//                   cross products (exact atan2 fallback); lazy TsdGridPartition::init
//                   (TsdGridPartition.cpp:88-134) folded in; TsdGridPartition::increaseEmptiness
//                   (TsdGridPartition.cpp:136-164) for EMPTY tiles.  No same-address atomics: every tile
//                   leaves a 4-byte record (what happened, cells updated) and running totals.
//   k_push_halo     TsdGrid::propagateBorders (TsdGrid.cpp:372-427) restricted to what can have
//                   changed: for every listed tile (touched, or freeFootprint's dirty mark) refresh its
//                   own halo from R/U/UR and the halos of L/D/DL that mirror its first column/row/cell.
//                   Equal to the reference's full sweep by induction (untouched pairs are already
//                   consistent).  One wave per tile.
//
// HBM-bound integer/fp64 work: no MFMA.  Roofline accounting in DESIGN.md.
#include "tsd_ctx.hpp"
#include <climits>

namespace tsd {

#ifndef TSD_UPDATE_BLOCK
#define TSD_UPDATE_BLOCK 256
#endif
constexpr int UPDATE_BLOCK = TSD_UPDATE_BLOCK;     // threads of the per-tile workgroup (64, 128 or 256)

// ---- per-tile record written by k_push_tiles -------------------------------------------------------
constexpr uint32_t REC_RANGE_PASS = 1u, REC_UPDATE = 2u, REC_NEW = 4u, REC_NEW_FROM_EMPTY = 8u,
                   REC_EMPTIED_INIT = 16u, REC_EMPTIED_UNINIT = 32u,
                   REC_LISTED = 64u;      // on this push's work list (k_push_halo: the tile refreshes its own halo itself)
constexpr int REC_CELLS_SHIFT = 8;

// ---- range-query tables of one scan (global memory, built by k_push_tables) ---------------------------
//   visible := any j in [lo,hi]: data[j] > closest && mask[j]          <=> max A > closest
//   empty   := all j in [lo,hi]: isinf(data[j]) ? distance < lowReflectivityRange : (data[j] > farthest && mask[j])
//                                                                      <=> min B > farthest and (no inf or near)
// with A[j] = mask ? data : -inf and B[j] = isinf ? +inf : (mask ? data : -inf).  Sparse tables of
// ARG-max / ARG-min indices (levels x beams x 2 B; the values stay fp64) + a prefix count of infinite beams.
struct RmqView {
  const double* A; const double* Bv;
  const unsigned short* inf;        // [B + 1]
  const unsigned short* tmax;       // [levels][Bp]
  const unsigned short* tmin;
  const double2* bdir;              // [B + 1] unit vectors of the beam boundaries phi_min + (j - 0.5) * res (fast_index)
  int Bp, levels;
};

__host__ __device__ inline int rmq_levels(int beams) { int l = 1; while ((1 << l) <= beams) l++; return l; }
__host__ __device__ inline size_t rmq_bytes(int beams)
{
  const size_t bp = (size_t)((beams + 3) & ~3);
  return 2 * bp * sizeof(double) + ((size_t)(beams + 1 + 7) & ~(size_t)7) * 2 + 2 * (size_t)rmq_levels(beams) * bp * 2 + 64 +
         ((size_t)beams + 2) * sizeof(double2);
}
__host__ __device__ inline RmqView rmq_view(char* buf, int beams)
{
  RmqView v;
  const size_t bp = (size_t)((beams + 3) & ~3);
  v.Bp = (int)bp; v.levels = rmq_levels(beams);
  double* A = reinterpret_cast<double*>(buf);
  v.A = A; v.Bv = A + bp;
  unsigned short* inf = reinterpret_cast<unsigned short*>(A + 2 * bp);
  v.inf = inf;
  unsigned short* tmax = inf + ((size_t)(beams + 1 + 7) & ~(size_t)7);
  v.tmax = tmax; v.tmin = tmax + (size_t)v.levels * bp;
  const size_t used = 2 * bp * sizeof(double) + ((size_t)(beams + 1 + 7) & ~(size_t)7) * 2 + 2 * (size_t)v.levels * bp * 2;
  v.bdir = reinterpret_cast<const double2*>(buf + ((used + 15) & ~(size_t)15));
  return v;
}

__device__ __forceinline__ void
push_tables_body(const double* __restrict__ ranges, const uint8_t* __restrict__ mask, int B, char* __restrict__ buf,
                 double phi_min, double ang_res)
{
  // LDS: values (2 x Bp doubles) + two levels of both index tables (ping-pong); each finished level is
  // streamed to global memory, so 4096 beams need 96 KB whatever the number of levels
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RmqView gv = rmq_view(buf, B);
  const int Bp = gv.Bp;
  double* sA = reinterpret_cast<double*>(smem);
  double* sB = sA + Bp;
  unsigned short* lv = reinterpret_cast<unsigned short*>(sB + Bp);      // [2][2][Bp]: parity, max/min
  double* gA = const_cast<double*>(gv.A); double* gB = const_cast<double*>(gv.Bv);
  unsigned short* ginf = const_cast<unsigned short*>(gv.inf);
  unsigned short* gmax = const_cast<unsigned short*>(gv.tmax); unsigned short* gmin = const_cast<unsigned short*>(gv.tmin);
  const int tid = threadIdx.x, T = blockDim.x;
  {
    double2* gb = const_cast<double2*>(gv.bdir);
    for (int j = tid; j <= B; j += T) {
      const double beta = phi_min + ((double)j - 0.5) * ang_res;
      gb[j] = make_double2(cos(beta), sin(beta));
    }
  }
  for (int i = tid; i < B; i += T) {
    const double d = ranges[i];
    const bool mk = mask[i] != 0;
    const double a = mk ? d : -__builtin_inf();
    const double b = isinf(d) ? __builtin_inf() : (mk ? d : -__builtin_inf());
    sA[i] = a; sB[i] = b; gA[i] = a; gB[i] = b;
    lv[i] = (unsigned short)i; lv[Bp + i] = (unsigned short)i;
    gmax[i] = (unsigned short)i; gmin[i] = (unsigned short)i;
  }
  if (tid < 64) {
    // prefix count of infinite readings by one wave (B <= 4096: 64 lanes x 64 beams)
    const int per = (B + 63) / 64;
    const int j0 = tid * per;
    int c = 0;
    for (int j = j0; j < j0 + per && j < B; j++) c += isinf(ranges[j]) ? 1 : 0;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (tid >= off) incl += t; }
    int run = incl - c;
    if (tid == 0) ginf[0] = 0;
    for (int j = j0; j < j0 + per && j < B; j++) { run += isinf(ranges[j]) ? 1 : 0; ginf[j + 1] = (unsigned short)run; }
  }
  __syncthreads();
  for (int k = 1; k < gv.levels; k++) {
    const int half = 1 << (k - 1), span = 1 << k;
    const unsigned short* pmax = lv + (size_t)((k - 1) & 1) * 2 * Bp; const unsigned short* pmin = pmax + Bp;
    unsigned short* cmax = lv + (size_t)(k & 1) * 2 * Bp; unsigned short* cmin = cmax + Bp;
    for (int j = tid; j + span <= B; j += T) {
      const unsigned short a0 = pmax[j], a1 = pmax[j + half];
      const unsigned short am = sA[a1] > sA[a0] ? a1 : a0;
      const unsigned short b0 = pmin[j], b1 = pmin[j + half];
      const unsigned short bm = sB[b1] < sB[b0] ? b1 : b0;
      cmax[j] = am; cmin[j] = bm;
      gmax[(size_t)k * Bp + j] = am; gmin[(size_t)k * Bp + j] = bm;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(1024)
k_push_tables(const double* __restrict__ ranges, const uint8_t* __restrict__ mask, int B, char* __restrict__ buf,
              double phi_min, double ang_res)
{
  push_tables_body(ranges, mask, B, buf, phi_min, ang_res);
}

// the tables of a batch of scans in ONE launch (tsd_batch_begin): workgroup x builds the tables of scan x
__global__ void __launch_bounds__(1024)
k_push_tables_batch(const TablesBatchEntry* __restrict__ entries)
{
  const TablesBatchEntry e = entries[blockIdx.x];
  push_tables_body(e.ranges, e.mask, e.beams, e.rmq, e.phi_min, e.ang_res);
}

// TsdGridPartition ctor geometry (TsdGridPartition.cpp:48-70)
__device__ __forceinline__ void tile_geometry(const GridDev& g, int p, double e[4][2], double& cx,
                                              double& cy, double& rad)
{
  const unsigned x = (unsigned)(p % g.PX) * TILE_DIM, y = (unsigned)(p / g.PX) * TILE_DIM;
  e[0][0] = ((double)x + 0.5) * g.cs;              e[0][1] = ((double)y + 0.5) * g.cs;
  e[1][0] = ((double)(x + TILE_DIM) + 0.5) * g.cs; e[1][1] = ((double)y + 0.5) * g.cs;
  e[2][0] = ((double)x + 0.5) * g.cs;              e[2][1] = ((double)(y + TILE_DIM) + 0.5) * g.cs;
  e[3][0] = ((double)(x + TILE_DIM) + 0.5) * g.cs; e[3][1] = ((double)(y + TILE_DIM) + 0.5) * g.cs;
  cx = (e[0][0] + e[1][0] + e[2][0] + e[3][0]) / 4.0;
  cy = (e[0][1] + e[1][1] + e[2][1] + e[3][1]) / 4.0;
  const double dx = e[3][0] - e[0][0], dy = e[3][1] - e[0][1];
  rad = sqrt(dx * dx + dy * dy) * 0.5;
}

// ---- correctly rounded fp64 square root and quotient for operands in the NORMAL range -----------------------
// The compiler's expansions of sqrt() / operator/ (v_rsq_f64 / v_rcp_f64 seed + Goldschmidt / Newton steps + one
// correction step, which is what makes them correctly rounded) wrap that core in scaling for tiny / huge operands
// (v_div_scale, v_ldexp, v_cmp_class, v_div_fixup).  Cell distances (1e-3 .. 1e3 m, squared) and the running
// average's operands (|numerator| <= 33, 1e-6 < denominator <= 33) never need the scaling, so the core alone gives
// bit-identical results with two thirds of the instructions.  (tests: every cell of every push bit-identical to the
// oracle's libm sqrt and IEEE division -- the grid digests of tests/golden pin exactly that.)
__device__ __forceinline__ double sqrt_normal(double x)
{
  // AMDGPU's f64 sqrt lowering without the 2^+-256 scaling and the zero / inf pass-through
  const double y = __builtin_amdgcn_rsq(x);
  const double g0 = x * y, h0 = 0.5 * y;
  const double r0 = __builtin_fma(-h0, g0, 0.5);
  const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
  const double d0 = __builtin_fma(-g1, g1, x);
  const double g2 = __builtin_fma(d0, h1, g1);
  const double d1 = __builtin_fma(-g2, g2, x);
  return __builtin_fma(d1, h1, g2);
}
__device__ __forceinline__ double div_normal(double n, double d)
{
  // AMDGPU's f64 division lowering without v_div_scale / v_div_fixup: reciprocal seed, two Newton steps, quotient,
  // one residual correction (the step v_div_fmas performs)
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double q = n * r;
  return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

// TsdGridPartition::addTsd (TsdGridPartition.h:170-212); the `fabs(sd) < _eps` branch is dead because
// _eps = -cellSize/2 (TsdGridPartition.cpp:95) and is kept only as a comparison against eps.
__device__ __forceinline__ bool add_tsd(double& tsd, double& weight, double sd, double part_weight,
                                        double max_trunc, double inv_max_trunc, double eps)
{
  if (sd >= -max_trunc) {
    const double v = fmin(sd * inv_max_trunc, 1.0);
    double w = 0.01;
    if (fabs(sd) < eps) w = 1.0;
    w *= part_weight;
    if (isnan(tsd)) {
      tsd = v;
      weight += w;
    } else {
      tsd = div_normal(tsd * weight + v * w, weight + w);
      weight = fmin(weight + w, MAX_WEIGHT);
    }
    return true;
  }
  return false;
}

// ---- work list of one push: tiles that need a workgroup (UPDATE, increaseEmptiness of a materialised tile)
// or only their halos refreshed (freeFootprint marks).  Entry = tile | kind << 28.
constexpr uint32_t KIND_UPDATE = 1u, KIND_EMPTY = 2u, KIND_HALO = 3u;
constexpr int KIND_SHIFT = 28;
constexpr uint32_t LIST_TILE_MASK = (1u << 20) - 1u;   // tile index (<= 2^20 tiles at map_size 15)
constexpr uint32_t LIST_FAR = 1u << 27;          // entry flag: the sensor is further than 3 circumradii from the tile centre
constexpr uint32_t LIST_INTERIOR = 1u << 26;     // far tile whose four corners all project to valid beams 1 .. beams-2 within the angular
                                                 // diameter of a far tile: every cell's beam is valid, inside [lo, hi], and no cell is near
                                                 // an end of the field of view or the +-pi cut (k_push_update skips those tests)
constexpr int TOT_FIELDS = 8;   // cells updated, range pass, update, new, new from empty, emptied init, emptied uninit, -

// isInRange for every tile of the launch window, one LANE per tile (TsdGridComponent.cpp:43-124: range cull,
// four corner back-projections, the two beam-range tests as O(1) table look-ups).  Tiles that need work go
// to the list (one atomic per wave); increaseEmptiness of a tile that was never materialised is done here
// (TsdGridPartition.cpp:157-162).  Every tile of the window gets its record.
__global__ void __launch_bounds__(64)
k_push_classify(GridDev g, const PushArgs* __restrict__ a_dev, const char* __restrict__ rmq_buf,
                uint32_t* __restrict__ tile_rec, const uint8_t* __restrict__ dirty, uint32_t* __restrict__ tile_totals,
                uint32_t* __restrict__ list, uint32_t* __restrict__ list_win, double* __restrict__ list_pw,
                unsigned int* __restrict__ list_cnt /* [2] */, int parity, int tx0, int ty0, int ntx, int nty)
{
  // FOUR lanes per tile, one corner each: the four back-projections (an fp64 atan2 apiece, by far the longest chain of this
  // kernel, and the kernel a pure latency chain: a few dozen waves on the whole chip) run side by side instead of one after
  // the other.  The quad shares everything else (same values in its four lanes); its first lane owns the tile's side effects.
  const PushArgs a = *a_dev;
  const int lane = threadIdx.x;
  const int corner = lane & 3;
  const bool owner = corner == 0;
  const int t = blockIdx.x * 16 + (lane >> 2);
  if (t == 0 && owner) list_cnt[parity ^ 1] = 0u;           // the next push's counter (nobody uses it now)
  const bool in_window = t < ntx * nty;
  const int p = in_window ? (ty0 + t / ntx) * g.PX + tx0 + t % ntx : 0;
  uint32_t rec = 0u, kind = 0u, far_flag = 0u;
  double pw = 0.0;
  uint32_t win = (uint32_t)(a.beams - 1) << 16;              // beams the cells of the tile can project to: lo | hi << 16
  if (in_window && a.enabled) {
    double e[4][2], cx, cy, rad;
    tile_geometry(g, p, e, cx, cy, rad);
    // euklideanDistance<obfloat>(pos, _centroid, 2) (mathbase.h:369-378)
    double sqr = 0.0;
    { const double t0 = a.trx - cx; sqr += t0 * t0; const double t1 = a.try_ - cy; sqr += t1 * t1; }
    const double distance = sqrt(sqr);
    const double closest = distance - rad - g.max_trunc;
    const double farthest = distance + rad + g.max_trunc;
    if (!(closest > a.max_range || farthest < a.min_range)) {
      rec = REC_RANGE_PASS;
      bool all_vis = true, any_vis = false;
      int lo = 0, hi = 0;
      {
        const int k = corner;
        const double ex = (k & 1) ? e[1][0] : e[0][0], ey = (k & 2) ? e[2][1] : e[0][1];
        int ik = backproject(a.Pi, ex, ey, a.phi_min, a.ang_res_inv, a.phi_lower, a.phi_upper);
        if (ik == -1) { ik = a.beams - 1; all_vis = false; }
        else if (ik == -2) { ik = 0; all_vis = false; }
        else any_vis = true;
        // minmaxArray<int> (mathbase.h:55-64) over the four corners = minimum and maximum over the quad's lanes
        lo = ik; hi = ik;
      }
      // (the whole quad is here or nowhere: the range cull above depends on the tile only)
#pragma unroll
      for (int m = 1; m <= 2; m <<= 1) {
        const int lo2 = __shfl_xor(lo, m, 64), hi2 = __shfl_xor(hi, m, 64);
        const int av2 = __shfl_xor((int)all_vis, m, 64), an2 = __shfl_xor((int)any_vis, m, 64);
        lo = lo2 < lo ? lo2 : lo; hi = hi2 > hi ? hi2 : hi;
        all_vis = all_vis && av2 != 0; any_vis = any_vis || an2 != 0;
      }
      int action = 0;
      if (any_vis) {
        const RmqView rv = rmq_view(const_cast<char*>(rmq_buf), a.beams);
        const int len = hi - lo + 1;
        const int k = 31 - __clz(len);                                         // floor(log2(len))
        const unsigned short* tm = rv.tmax + (size_t)k * rv.Bp;
        const unsigned short* tn = rv.tmin + (size_t)k * rv.Bp;
        const int j2 = hi - (1 << k) + 1;
        const unsigned short i0 = tm[lo], i1 = tm[j2], i2 = tn[lo], i3 = tn[j2];
        const unsigned short n0 = rv.inf[lo], n1 = rv.inf[hi + 1];
        const double amax = fmax(rv.A[i0], rv.A[i1]);
        const double bmin = fmin(rv.Bv[i2], rv.Bv[i3]);
        const bool has_inf = n1 != n0;
        const bool visible = amax > closest;
        const bool empty = (bmin > farthest) && (!has_inf || distance < a.low_refl);
        if (visible) action = (all_vis && empty) ? 1 : 2;
      }
      // The cell centres of a tile lie inside the quadrilateral of the four corner points; seen from a sensor
      // well outside of it the extreme angles are those of corners, so every cell projects into [lo, hi]
      // (corners outside the field of view were mapped to its ends above).
      if (distance > 3.0 * rad) {
        win = (uint32_t)lo | ((uint32_t)hi << 16); far_flag = LIST_FAR;
        // angular diameter of a far tile < 2 asin(1/3) = 0.68 rad; a tile that straddles the cut of a full-circle sensor has its
        // corner indices at both ends of the scan instead
        if (all_vis && lo >= 1 && hi <= a.beams - 2 && (double)(hi - lo) <= 0.7 * a.ang_res_inv + 2.0) far_flag |= LIST_INTERIOR;
      }
      if (action == 2) {
        kind = KIND_UPDATE;
        // partition weight (TsdGrid.cpp:239-243): ((maxRange - min(distance to the centroid, maxRange)) / maxRange)^2.
        // `distance` above is that distance bit for bit ((a - b)^2 == (b - a)^2, 0.0 + x == x), so the per-tile
        // square root and division are done once here instead of by every thread of the update workgroup.
        double dc = distance;
        if (dc > a.max_range) dc = a.max_range;
        pw = (a.max_range - dc) / a.max_range;
        pw *= pw;
      }
      else if (action == 1) {
        // TsdGridPartition::increaseEmptiness (TsdGridPartition.cpp:136-164), isInRange then returns false
        if (g.flags[p]) kind = KIND_EMPTY;
        else {
          if (owner) {
            double v = g.init_weight[p] + 1.0; v = fmin(v, MAX_WEIGHT); g.init_weight[p] = v;
            tile_totals[(size_t)p * TOT_FIELDS + 6] += 1u;
          }
          rec |= REC_EMPTIED_UNINIT;
        }
      }
      if (owner) tile_totals[(size_t)p * TOT_FIELDS + 1] += 1u;         // (this lane owns the tile: no atomics)
    }
    if (kind == 0u && dirty[p] != 0) { kind = KIND_HALO; rec |= REC_LISTED; }       // written by freeFootprint since the last push
  }
  if (in_window && owner && (kind == 0u || kind == KIND_HALO)) tile_rec[p] = rec;   // UPDATE / EMPTY: the workgroup writes the final record
  if (!owner) kind = 0u;
  const unsigned long long listed = __ballot(kind != 0u);
  if (listed) {
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(&list_cnt[parity], (unsigned int)__popcll(listed));
    base = __shfl(base, 0, 64);
    if (kind != 0u) {
      const unsigned int slot = base + __popcll(listed & ((1ull << lane) - 1ull));
      list[slot] = (uint32_t)p | far_flag | (kind << KIND_SHIFT);
      list_win[slot] = win;
      list_pw[slot] = pw;
    }
  }
}

// ---- beam index of a cell without the fp64 atan2 ------------------------------------------------------------
// SensorPolar2D::backProject (SensorPolar2D.cpp:117-135) decides round((atan2(ly, lx) - phi_min) / res) and the two
// bound checks from fp64 values.  The kernel ESTIMATES the beam coordinate u = (angle - phi_min) / res in fp32:
//   far tiles (sensor further than 3 circumradii from the tile centre -- all but a handful): the cell's angle is the
//       tile centre's plus a small delta, |delta| < 0.34 rad, and delta = atan(cross / dot) by a four-term series;
//       ~17 fp32 instructions
//   near tiles: a six-term minimax arctangent over the full circle; ~30
// Every source of error (fp32 coordinates relative to the tile centre, v_rcp_f32, the series truncation t^9 / 9, the
// fp32 product with 1 / res at u <= 4096) stays below 4e-3 beams by the error budget, so an estimate further than 0.02
// beams from a rounding boundary (j +- 0.5), from the ends of the field of view and from the +-pi cut of atan2 names the
// reference's beam with a margin of 5x.  Checked on the device: the -DTSD_PUSH_VERIFY_INDEX build compares every decided
// cell with the exact formulation inside the kernel -- 749 M cells over the BASELINE scenes, none decided wrongly
// (tools/push_verify_index.sh, profiles/r2_push_index_estimate_verified.txt).  The other cells (4-5 %) are not decided by the estimate at all: they go to
// a queue in LDS and get the exact fp64 formulation, densely (one lane per queued cell), instead of dragging their
// whole wave through it.
constexpr int IDX_UNSURE = INT_MIN;
constexpr float IDX_MARGIN = 0.02f;
__device__ __forceinline__ float atan2_estimate(float y, float x)      // |error| < 2e-6 rad
{
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float t = mn * __builtin_amdgcn_rcpf(mx);
  const float t2 = t * t;
  // minimax odd polynomial of atan on [0, 1] (Abramowitz-Stegun 4.4.49 class, 6 terms)
  float r = -0.0117212f;
  r = fmaf(r, t2, 0.05265332f);
  r = fmaf(r, t2, -0.11643287f);
  r = fmaf(r, t2, 0.19354346f);
  r = fmaf(r, t2, -0.33262347f);
  r = fmaf(r, t2, 0.99997726f);
  r *= t;
  r = ay > ax ? 1.57079637f - r : r;
  r = x < 0.f ? 3.14159274f - r : r;
  return y < 0.f ? -r : r;
}
// beam index from the estimated angle `th` (radians, possibly outside (-pi, pi] by the small delta)
// interior (wave-uniform, LIST_INTERIOR): the tile lies inside the field of view by a beam and away from the cut, the angle is
// continuous over it -- only the distance to a rounding boundary is left to test
__device__ __forceinline__ int index_from_angle(float th, float phi_min_f, float inv_res_f, int beams, bool interior = false)
{
  const float PI_F = 3.14159274f;
  if (!interior) {
    if (th > PI_F) th -= 2.0f * PI_F;                        // the reference's atan2 lives in (-pi, pi]
    else if (th <= -PI_F) th += 2.0f * PI_F;
    if (fabsf(th) > PI_F - 1e-3f) return IDX_UNSURE;         // at the cut the two branches are 2 pi apart: exact path
  }
  const float u = (th - phi_min_f) * inv_res_f;            // beam coordinate
  if (!interior) {
    if (u < -0.5f - IDX_MARGIN || u > (float)beams - 0.5f + IDX_MARGIN) return -1;     // outside the field of view for sure
    if (!(u > -0.5f + IDX_MARGIN && u < (float)beams - 0.5f - IDX_MARGIN)) return IDX_UNSURE;
  }
  const float j = rintf(u);
  if (fabsf(u - j) > 0.5f - IDX_MARGIN) return IDX_UNSURE;
  return (int)j;
}


// One workgroup per listed tile (TsdGrid.cpp:237-274): scan window staged in LDS, 4 cells per thread, row-major =>
// coalesced 8-byte RMW; lazy TsdGridPartition::init (TsdGridPartition.cpp:88-134) folded in (a fresh tile's
// old value is known, so it is written once, halo included).  KIND_EMPTY: increaseEmptiness over the 33x33
// cells.  The workgroup leaves the tile's record and adds it to the tile's running totals.
//
// Instruction diet (the kernel is issue bound once the pushes are large: cfg3 / comb visits 10 M cells):
//   pass A  beam index of every cell from the fp32 estimate above; undecided cells queue up in LDS
//   drain   the queued cells, one lane each: the exact fp64 atan2 formulation
//   pass B  mask / range of the beam from LDS; an fp32 test throws out the cells that lie behind the surface by more
//           than the truncation for sure (|l|^2 against (range + maxTruncation)^2 with a 1e-5 margin, fp32 being good
//           to 6e-7 here) -- a wave whose cells are all such skips the exact distance altogether -- then the exact
//           IEEE distance / signed distance for the rest, the reads of the cells addTsd will touch, addTsd, the writes.
#ifndef TSD_UPDATE_WPS
#define TSD_UPDATE_WPS 4
#endif
#ifndef TSD_UPDATE_BATCH
#define TSD_UPDATE_BATCH 2
#endif
__global__ void __launch_bounds__(UPDATE_BLOCK, TSD_UPDATE_WPS)      // 4 waves per SIMD = four workgroups per CU: the listed tiles of a usual push are resident at once
k_push_update(GridDev g, const PushArgs* __restrict__ a_dev, const double* __restrict__ ranges,
              const uint8_t* __restrict__ mask, uint32_t* __restrict__ tile_rec, uint32_t* __restrict__ tile_totals,
              const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_win, const double* __restrict__ list_pw,
              const unsigned int* __restrict__ list_cnt, int parity, double* __restrict__ dbg)
{
#ifdef TSD_PUSH_STAMPS   // diagnostic: 100 MHz wall clock at the phases of every 8th listed tile (thread 0)
#define PSTAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 128) dbg[(blockIdx.x >> 3) * 8 + (i)] = (double)wall_clock64(); } while (0)
#else
#define PSTAMP(i) do {} while (0)
#endif
  PSTAMP(0);
  // the list length, this workgroup's first entry (read speculatively) and the arguments arrive together
  const unsigned int n_list = list_cnt[parity];
  const uint32_t first = list[blockIdx.x];
  const uint32_t first_win = list_win[blockIdx.x];
  const double first_pw = list_pw[blockIdx.x];
  const PushArgs a = *a_dev;
  if (blockIdx.x >= n_list) return;
  PSTAMP(1);
  const int tid = threadIdx.x, lane = tid & 63;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned int* s_upd = reinterpret_cast<unsigned int*>(smem);             // [4] cells updated per wave
  __shared__ unsigned long long s_neg_store;
  unsigned long long* s_neg = &s_neg_store;                               // groups of the tile that received a negative value
  __shared__ unsigned int s_qn;                                           // cells queued for the exact beam index
  __shared__ unsigned short s_queue[TILE_INTERIOR];
  __shared__ short s_idx[TILE_INTERIOR];                                  // their exact index (-1 / -2: outside the field of view)
  double* s_ranges = reinterpret_cast<double*>(smem + 16);
  uint8_t* s_mask = reinterpret_cast<uint8_t*>(smem + 16 + (size_t)((a.beams + 1) & ~1) * sizeof(double));
  const float phi_min_f = (float)a.phi_min, inv_res_f = (float)a.ang_res_inv;
  // The scan is staged once per workgroup, together with the first tile's state: only the beams the tile can project
  // to when the workgroup has a single tile (the usual case), all of them when it will loop over several.  Reads
  // outside the staged window go to global memory.
  const int p_first = (int)(first & LIST_TILE_MASK);
  const uint8_t flag_first = g.flags[p_first];
  const double iw_first = g.init_weight[p_first];
  int wlo = 0, whi = a.beams - 1;
  if (n_list <= gridDim.x) {
    wlo = (int)(first_win & 0xFFFFu) - 1; whi = (int)(first_win >> 16) + 1;
    if (wlo < 0) wlo = 0;
    if (whi > a.beams - 1) whi = a.beams - 1;
  }
  if (whi - wlo + 1 <= UPDATE_BLOCK) {
    // the usual tile, seen from outside: a few dozen beams, at most one element of each array per thread
    const int j = wlo + tid;
    const bool in_r = j <= whi;
    const double rj = in_r ? ranges[j] : 0.0;
    const uint8_t mj = in_r ? mask[j] : (uint8_t)0;
    if (in_r) { s_ranges[j] = rj; s_mask[j] = mj; }
  } else if (a.beams <= 2048) {
    // every read issued before the first LDS write: one memory latency for the whole staging
    constexpr int NR = 2048 / 2 / UPDATE_BLOCK, NM = 2048 / UPDATE_BLOCK;
    const int p0 = wlo >> 1, p1 = whi >> 1;                    // pairs of ranges
    const double2* r2 = reinterpret_cast<const double2*>(ranges);
    double2 rr[NR]; uint8_t mm[NM];
#pragma unroll
    for (int i = 0; i < NR; i++) { const int j = p0 + tid + i * UPDATE_BLOCK; rr[i] = j <= p1 ? r2[j] : make_double2(0.0, 0.0); }
#pragma unroll
    for (int i = 0; i < NM; i++) { const int j = wlo + tid + i * UPDATE_BLOCK; mm[i] = j <= whi ? mask[j] : (uint8_t)0; }
    double2* s_r2 = reinterpret_cast<double2*>(s_ranges);
#pragma unroll
    for (int i = 0; i < NR; i++) { const int j = p0 + tid + i * UPDATE_BLOCK; if (j <= p1) s_r2[j] = rr[i]; }
#pragma unroll
    for (int i = 0; i < NM; i++) { const int j = wlo + tid + i * UPDATE_BLOCK; if (j <= whi) s_mask[j] = mm[i]; }
  } else {
    for (int i = tid; i < a.beams; i += UPDATE_BLOCK) { s_ranges[i] = ranges[i]; s_mask[i] = mask[i]; }
  }

  // Software pipeline over the workgroup's tiles (a large push gives every workgroup a dozen): the list entry of the
  // NEXT tile is requested at the top of an iteration and its tile state after pass A, so an iteration's only
  // dependent trip to memory is the read of the cells it updates.
  uint32_t nx_entry = first; double nx_pw = first_pw; uint8_t nx_flag = flag_first; double nx_iw = iw_first;
  for (unsigned int li = blockIdx.x; li < n_list; li += gridDim.x) {
    const uint32_t entry = nx_entry;
    const double pw = nx_pw;
    const bool initialised = nx_flag != 0;
    const double iw = nx_iw;
    const unsigned int li_n = li + gridDim.x;
    const bool has_next = li_n < n_list;
    if (has_next) { nx_entry = list[li_n]; nx_pw = list_pw[li_n]; }
    const uint32_t kind = entry >> KIND_SHIFT;
    const int p = (int)(entry & LIST_TILE_MASK);
    if (kind == KIND_HALO) {
      if (has_next) { const int pn = (int)(nx_entry & LIST_TILE_MASK); nx_flag = g.flags[pn]; nx_iw = g.init_weight[pn]; }
      continue;
    }

    tsd_cell_t* __restrict__ T = g.tsd + (size_t)p * TILE_STRIDE;
    w_cell_t* __restrict__ W = g.weight + (size_t)p * TILE_STRIDE;
    uint32_t rec = REC_RANGE_PASS;

    if (kind == KIND_EMPTY) {
      // all 33x33 cells, halo included; the average uses the NEW weight
      for (int i = tid; i < TILE_CELLS; i += UPDATE_BLOCK) {      // (interior, halo column, halo row: offsets 0..1088)
        double t = ld_tsd(T + i), w = ld_w(W + i);
        if (isnan(t)) { w += 1.0; t = 1.0; }
        else { w = fmin(w + 1, MAX_WEIGHT); t = (t * (w - 1.0) + 1.0) / w; }
        st_tsd(T + i, t); st_w(W + i, w);
      }
      rec |= REC_EMPTIED_INIT | REC_LISTED;
      if (tid == 0) { tile_rec[p] = rec; atomicAdd(&tile_totals[(size_t)p * TOT_FIELDS + 5], 1u); }
      if (has_next) { const int pn = (int)(nx_entry & LIST_TILE_MASK); nx_flag = g.flags[pn]; nx_iw = g.init_weight[pn]; }
      continue;
    }

    // ---- UPDATE ----
    if (tid == 0) { *s_neg = 0ull; s_qn = 0u; }
    __syncthreads();               // scan staged; s_upd / queue of a previous tile consumed
    PSTAMP(2);

    const bool fresh = !initialised;
    rec |= REC_UPDATE | REC_LISTED;
    if (fresh) rec |= REC_NEW | (iw > 0.0 ? REC_NEW_FROM_EMPTY : 0u);
    // TsdGridPartition::init values (TsdGridPartition.cpp:98-120)
    const double t_init = (iw > 0.0) ? 1.0 : __builtin_nan("");
    const double w_init = iw;
    const double max_trunc = g.max_trunc;
    const double inv_max_trunc = 1.0 / max_trunc;
    const double eps = -g.cs / 2.0;
    // (partition weight `pw`, TsdGrid.cpp:239-243: evaluated once per tile by k_push_classify)

    const unsigned x0 = (unsigned)(p % g.PX) * TILE_DIM, y0 = (unsigned)(p / g.PX) * TILE_DIM;
    unsigned int n_upd = 0;
    constexpr int CPT = (TILE_DIM * TILE_DIM) / UPDATE_BLOCK;
    int bidx[CPT]; float d2f[CPT];
    // Cell k of this thread: (ix, iy0 + 8 k).  A wave covers a 32 x 2 strip of cells per k: two 256-byte rows per
    // access.  (Measured alternative, -DTSD_UPDATE_BLOCKS: 16 x 4 blocks -- four 128-byte segments per access -- so that
    // fewer waves straddle the edge of the region addTsd touches: 25 % SLOWER at cfg3 / comb, 111 vs 89 us; the access
    // shape outweighs the divergence.)
#ifndef TSD_UPDATE_BLOCKS
    const unsigned ix = (unsigned)tid & 31u, iy0 = (unsigned)tid >> 5;
#else
    const unsigned ix = (((unsigned)tid >> 6) & 1u) * 16u + ((unsigned)tid & 15u);
    const unsigned iy0 = ((unsigned)tid >> 7) * 4u + (((unsigned)tid >> 4) & 3u);
#endif
    const int c0 = (int)(iy0 * 32u + ix);                                  // offset of cell k: c0 + 256 k

    // ---- pass A: beam index of every cell from the fp32 estimate (sensor frame relative to the tile centre)
    {
      // tile centre = corner of the four middle cells; l_c = PoseInv * (centre, 1) in fp64 once, the cells relative
      // to it in fp32: l = l_c + (PoseInv's rotation * cellSize) * (cell offset in cells), exact small offsets
      const double ccx_c = (double)(x0 + 16u) * g.cs, ccy_c = (double)(y0 + 16u) * g.cs;
      const double lcx = a.Pi[0] * ccx_c + a.Pi[1] * ccy_c + a.Pi[2], lcy = a.Pi[3] * ccx_c + a.Pi[4] * ccy_c + a.Pi[5];
      const float lcxf = (float)lcx, lcyf = (float)lcy;
      const float axx = (float)(a.Pi[0] * g.cs), axy = (float)(a.Pi[1] * g.cs), ayx = (float)(a.Pi[3] * g.cs), ayy = (float)(a.Pi[4] * g.cs);
      const bool far = (entry & LIST_FAR) != 0u;
      const bool interior = (entry & LIST_INTERIOR) != 0u;
      const float th_c = atan2_estimate(lcyf, lcxf);
      const float dxc = (float)ix - 15.5f;
      const float bx = fmaf(axx, dxc, lcxf), by = fmaf(ayx, dxc, lcyf);
#pragma unroll
      for (int k = 0; k < CPT; k++) {
        const float dyc = (float)(iy0 + 8u * (unsigned)k) - 15.5f;
        const float lxf = fmaf(axy, dyc, bx), lyf = fmaf(ayy, dyc, by);
        d2f[k] = fmaf(lxf, lxf, lyf * lyf);
        float th;
        if (far) {
          // angle relative to the tile centre: tan(delta) = cross / dot, |delta| < 0.34 rad; atan by its series
          const float cr = lcxf * lyf - lcyf * lxf, dt = fmaf(lcxf, lxf, lcyf * lyf);
          const float t = cr * __builtin_amdgcn_rcpf(dt);
          const float t2 = t * t;
          float r = fmaf(t2, -0.142857143f, 0.2f);
          r = fmaf(r, t2, -0.333333333f);
          r = fmaf(r, t2, 1.0f);
          th = fmaf(r, t, th_c);
        } else {
          th = atan2_estimate(lyf, lxf);
        }
        bidx[k] = index_from_angle(th, phi_min_f, inv_res_f, a.beams, interior);
      }
#ifdef TSD_PUSH_VERIFY_INDEX   // diagnostic build: every decided cell against the exact formulation
#pragma unroll
      for (int k = 0; k < CPT; k++) {
        const double ccx = ((double)(x0 + ix) + 0.5) * g.cs, ccy = ((double)(y0 + iy0 + 8u * (unsigned)k) + 0.5) * g.cs;
        const int ex = backproject(a.Pi, ccx, ccy, a.phi_min, a.ang_res_inv, a.phi_lower, a.phi_upper);
        if (bidx[k] != IDX_UNSURE && (bidx[k] < 0 ? ex >= 0 : ex != bidx[k])) atomicAdd(reinterpret_cast<unsigned long long*>(dbg + 1000), 1ull);
        if (bidx[k] == IDX_UNSURE) atomicAdd(reinterpret_cast<unsigned long long*>(dbg + 1001), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(dbg + 1002), 1ull);
      }
#endif
      // undecided cells -> queue (one LDS atomic per wave and k)
#pragma unroll
      for (int k = 0; k < CPT; k++) {
        const bool un = bidx[k] == IDX_UNSURE;
        const unsigned long long m = __ballot(un);
        if (m) {
          unsigned int base = 0;
          if (lane == 0) base = atomicAdd(&s_qn, (unsigned int)__popcll(m));
          base = __shfl(base, 0, 64);
          if (un) s_queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(c0 + UPDATE_BLOCK * k);
        }
      }
    }
    if (has_next) { const int pn = (int)(nx_entry & LIST_TILE_MASK); nx_flag = g.flags[pn]; nx_iw = g.init_weight[pn]; }   // (next tile's state: in flight during the rest)
    __syncthreads();
    // ---- drain: the exact formulation (SensorPolar2D::backProject: fp64 atan2, bound checks, round) for the queued cells
    {
      const unsigned int nq = s_qn;
      for (unsigned int q = (unsigned)tid; q < nq; q += UPDATE_BLOCK) {
        const unsigned c = s_queue[q];
        const double ccx = ((double)(x0 + (c & 31u)) + 0.5) * g.cs;   // TsdGridPartition.cpp:127-128
        const double ccy = ((double)(y0 + (c >> 5)) + 0.5) * g.cs;
        s_idx[c] = (short)backproject(a.Pi, ccx, ccy, a.phi_min, a.ang_res_inv, a.phi_lower, a.phi_upper);
      }
    }
    __syncthreads();
    PSTAMP(3);

    // ---- pass B: signed distance and whether addTsd will touch the cell (sd >= -maxTruncation: cells behind the
    // surface cost no HBM traffic)
    unsigned cnd = 0u;                                           // bit k: cell k is a candidate
    const double ccx = ((double)(x0 + ix) + 0.5) * g.cs;
    const double dxw = ccx - a.trx, dxw2 = dxw * dxw;
    const float low2f = (float)(a.low_refl * a.low_refl) * 1.00001f;
    const float mtf = (float)max_trunc;
    // (B1) candidates: valid beam, and not behind the surface by more than the truncation for sure.  (Registers are
    // tight at four workgroups per CU -- a spilled dword costs scratch traffic per tile -- so the range is looked up
    // again in B3 instead of being kept.)
#pragma unroll
    for (int k = 0; k < CPT; k++) {
      const int c = c0 + UPDATE_BLOCK * k;
      int index = bidx[k];
      if (index == IDX_UNSURE) index = s_idx[c];
      bidx[k] = index;
      // mask and range of the beam from LDS, both at once (a select between an LDS and a global pointer would turn
      // into a flat load with a full wait per cell); a beam outside the staged window -- possible only through
      // rounding at the window's ends -- is fetched from global memory by the lanes concerned
      const bool staged_beam = index >= wlo && index <= whi;
      const int il = index < wlo ? wlo : (index > whi ? whi : index);
      unsigned mk = s_mask[il];
      double r = s_ranges[il];
      asm volatile("" : "+v"(mk), "+v"(r));      // (keeps the two LDS reads LDS reads: no pointer select)
      if (__builtin_expect(index >= 0 && !staged_beam, 0)) { mk = mask[index]; r = ranges[index]; }
      bool cand = index >= 0 && mk != 0u;
      if (cand) {
        const float rf = (float)r + mtf;
        const float lim2 = isinf(r) ? low2f : rf * rf * 1.00001f;
        cand = !(d2f[k] > lim2);
      }
      cnd |= cand ? (1u << k) : 0u;
    }
    // (B2-B4) in two halves of two cells (a rolled loop: half the registers of doing all four at once, which is what keeps
    // the kernel at four workgroups per CU without scratch spills): the candidates' reads go out -- candidates are the
    // updated cells plus a sliver at the truncation boundary -- and are in flight during the exact distances; then
    // addTsd and the writes
    unsigned long long wrote_neg = 0ull;
    constexpr int HB = TSD_UPDATE_BATCH;          // cells per batch: 2 (default) or 4
#pragma unroll 1
    for (int h = 0; h < CPT / HB; h++) {
      int idx2[HB];
      if constexpr (HB == 2) { idx2[0] = h ? bidx[2] : bidx[0]; idx2[1] = h ? bidx[3] : bidx[1]; }
      else { for (int j = 0; j < HB; j++) idx2[j] = bidx[j]; }
      const unsigned cn2 = cnd >> (HB * h);
      double tv[HB], wv[HB], sdv[HB]; bool hit[HB];
#pragma unroll
      for (int j = 0; j < HB; j++) {
        const int c = c0 + UPDATE_BLOCK * (HB * h + j);
        tv[j] = t_init; wv[j] = w_init;
        if (((cn2 >> j) & 1u) && !fresh) { tv[j] = ld_tsd(T + c); wv[j] = ld_w(W + c); }
      }
#pragma unroll
      for (int j = 0; j < HB; j++) {
        hit[j] = false; sdv[j] = 0.0;
        if ((cn2 >> j) & 1u) {
          const int index = idx2[j];
          const bool staged_beam = index >= wlo && index <= whi;
          const int il = index < wlo ? wlo : (index > whi ? whi : index);
          double r = s_ranges[il];
          asm volatile("" : "+v"(r));
          if (__builtin_expect(!staged_beam, 0)) r = ranges[index];
          const double ccy = ((double)(y0 + iy0 + 8u * (unsigned)(HB * h + j)) + 0.5) * g.cs;
          const double dyw = ccy - a.try_;
          const double dist = sqrt_normal(dxw2 + dyw * dyw);        // (ccx - trx)^2 + (ccy - try)^2, then the IEEE root
          double sd = 0.0; bool ok = false;
          if (!isinf(r)) { sd = r - dist; ok = true; }
          else if (dist < a.low_refl) { sd = max_trunc; ok = true; }
          hit[j] = ok && sd >= -max_trunc;
          sdv[j] = sd;
        }
      }
#pragma unroll
      for (int j = 0; j < HB; j++) {
        const int c = c0 + UPDATE_BLOCK * (HB * h + j);
        bool touched = false;
        if (hit[j]) touched = add_tsd(tv[j], wv[j], sdv[j], pw, max_trunc, inv_max_trunc, eps);
        if (touched) n_upd++;
        if (touched && tv[j] < 0.0) wrote_neg |= neg_bit((unsigned)c & 31u, (unsigned)c >> 5);
        if (touched || fresh) { st_tsd(T + c, tv[j]); st_w(W + c, wv[j]); }
      }
    }
    PSTAMP(4);
    if (wrote_neg) atomicOr(s_neg, wrote_neg);                      // (LDS; folded into the tile's mask below)
    if (fresh) {
      // halo cells of a freshly materialised tile keep the init value until k_push_halo
      for (int h = tid; h < 2 * TILE_DIM + 1; h += UPDATE_BLOCK) {      // the halo strip: column 32, then row 32
        st_tsd(T + HALO_COL + h, t_init); st_w(W + HALO_COL + h, w_init);
      }
    }
    PSTAMP(5);
    const unsigned wu = (unsigned)wave_sum_i((int)n_upd);
    if (lane == 0) s_upd[tid >> 6] = wu;
    __syncthreads();               // every thread is done with the cells (and has read `initialised`)
    PSTAMP(6);
    if (tid == 0) {
      unsigned cells = 0;
      for (int w = 0; w < UPDATE_BLOCK / 64; w++) cells += s_upd[w];
      tile_rec[p] = rec | (cells << REC_CELLS_SHIFT);
      // The tile's running totals and mask: NO-RETURN atomics (fire and forget; every tile has its own words, so
      // nothing contends).  As read-add-write they were four dependent trips to memory by thread 0 at the end of every
      // tile, with the rest of the workgroup waiting for it at the next tile's first barrier.
      const unsigned long long nm = *s_neg;
      uint32_t* tot = tile_totals + (size_t)p * TOT_FIELDS;
#ifndef TSD_RMW_RECORDS
      if (nm) atomicOr(&g.negmask[p], nm);
      atomicAdd(&tot[0], cells); atomicAdd(&tot[2], 1u);
      if (fresh) { atomicAdd(&tot[3], 1u); if (iw > 0.0) atomicAdd(&tot[4], 1u); }
#else
      if (nm) g.negmask[p] |= nm;
      tot[0] += cells; tot[2] += 1u;
      if (fresh) { tot[3] += 1u; if (iw > 0.0) tot[4] += 1u; }
#endif
      if (fresh) g.flags[p] = 1;   // publish the tile
    }
  }
}

// TsdGrid::propagateBorders (TsdGrid.cpp:372-427), incremental form.
// One wave per listed tile: the tile's own halo from R/U/UR and the halos of L/D/DL that mirror its first
// column/row/cell.  Equal to the reference's full sweep by induction (untouched pairs are already consistent).
// Every lane has up to three copy jobs (source cell -> destination cell, tsd and weight); all loads of a tile are
// issued before the first store, so a tile costs one memory round trip after its flags (round 1 ran the six copies
// one after the other: 40 us of latency chain at cfg3 / comb).  Lanes 0..31: the two column copies; lanes 32..63:
// the two row copies; lane 0 / lane 32: the corner cells.
__global__ void __launch_bounds__(256)
k_push_halo(GridDev g, uint8_t* __restrict__ dirty, unsigned long long* __restrict__ pushes,
            const PushArgs* __restrict__ a_dev, const uint32_t* __restrict__ list, const uint32_t* __restrict__ tile_rec,
            const unsigned int* __restrict__ list_cnt, int parity, double cx, double cy, double slack)
{
  const int lane = threadIdx.x & 63;
  const unsigned int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv == 0 && lane == 0) {
    const bool enabled = a_dev->enabled != 0;
    if (enabled) {
      pushes[0] += 1ull;
      // the window was laid around (cx, cy) +- slack by the host: a sensor outside of that is a host-side bug
      const double sx = a_dev->trx, sy = a_dev->try_;
      if (!(fabs(sx - cx) <= slack && fabs(sy - cy) <= slack)) pushes[1] += 1ull;
    }
  }
  const unsigned int n_list = list_cnt[parity];
  const uint32_t first = list[wv];                            // speculative: arrives with the list length
  const int PX = g.PX;
  const bool colhalf = lane < TILE_DIM;
  const int i = lane & 31;
  for (unsigned int li = wv; li < n_list; li += gridDim.x * 4) {
    const uint32_t entry = (li == wv) ? first : list[li];
    const int p = (int)(entry & LIST_TILE_MASK);
    if (lane == 0 && dirty[p] != 0) dirty[p] = 0;
    const int px = p % PX, py = p / PX;
    const bool hasR = px < PX - 1, hasU = py < PX - 1, hasL = px > 0, hasD = py > 0;
    // all nine flags in flight at once
    const uint8_t f0 = g.flags[p];
    const uint8_t fR = hasR ? g.flags[p + 1] : 0, fU = hasU ? g.flags[p + PX] : 0, fUR = (hasR && hasU) ? g.flags[p + PX + 1] : 0;
    uint8_t fL = hasL ? g.flags[p - 1] : 0, fD = hasD ? g.flags[p - PX] : 0, fDL = (hasL && hasD) ? g.flags[p - PX - 1] : 0;
    // a left / lower / diagonal neighbour that is on this push's list refreshes its own halo from this tile itself
    // (its job 0 / corner job is the very same copy): skipping the mirror job halves the column gathers where the
    // listed tiles are dense.  Records outside this push's window are never "listed" (see launch_push).
    const uint32_t rL = hasL ? tile_rec[p - 1] : 0u, rD = hasD ? tile_rec[p - PX] : 0u, rDL = (hasL && hasD) ? tile_rec[p - PX - 1] : 0u;
    if (!f0) continue;
    if (rL & REC_LISTED) fL = 0;
    if (rD & REC_LISTED) fD = 0;
    if (rDL & REC_LISTED) fDL = 0;
    const size_t own = (size_t)p * TILE_STRIDE;
    // job 0: own halo from the right / upper neighbour; job 1: the left / lower neighbour's halo from this tile;
    // job 2 (lanes 0 and 32 only): the corner cells
    size_t src[3], dst[3]; bool on[3];
    if (colhalf) {
      on[0] = fR != 0;  src[0] = (size_t)(p + 1) * TILE_STRIDE + (size_t)i * TILE_DIM;  dst[0] = own + HALO_COL + i;
      on[1] = fL != 0;  src[1] = own + (size_t)i * TILE_DIM;                           dst[1] = (size_t)(p - 1) * TILE_STRIDE + HALO_COL + i;
      on[2] = lane == 0 && fUR != 0; src[2] = (size_t)(p + PX + 1) * TILE_STRIDE;      dst[2] = own + HALO_ROW + TILE_DIM;
    } else {
      on[0] = fU != 0;  src[0] = (size_t)(p + PX) * TILE_STRIDE + i;                   dst[0] = own + HALO_ROW + i;
      on[1] = fD != 0;  src[1] = own + i;                                              dst[1] = (size_t)(p - PX) * TILE_STRIDE + HALO_ROW + i;
      on[2] = lane == 32 && fDL != 0; src[2] = own;                                    dst[2] = (size_t)(p - PX - 1) * TILE_STRIDE + HALO_ROW + TILE_DIM;
    }
    tsd_cell_t tv[3]; w_cell_t wv_[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { tv[k] = tsd_cell_t(); wv_[k] = w_cell_t(); if (on[k]) { tv[k] = g.tsd[src[k]]; wv_[k] = g.weight[src[k]]; } }
#pragma unroll
    for (int k = 0; k < 3; k++) if (on[k]) { g.tsd[dst[k]] = tv[k]; g.weight[dst[k]] = wv_[k]; }
  }
}

// TsdGrid::freeFootprint (TsdGrid.cpp:609-638): lazily initialise touched tiles, set tsd = 1.0
// (weight untouched).  One 256-thread block per tile of the rectangle's tile range.
__global__ void __launch_bounds__(256)
k_free_footprint(GridDev g, unsigned minX, unsigned maxX, unsigned minY, unsigned maxY,
                 unsigned tx0, unsigned ty0, unsigned ntx, uint8_t* __restrict__ dirty)
{
  const unsigned tx = tx0 + blockIdx.x % ntx, ty = ty0 + blockIdx.x / ntx;
  const int p = (int)(ty * (unsigned)g.PX + tx);
  const int tid = threadIdx.x;
  tsd_cell_t* T = g.tsd + (size_t)p * TILE_STRIDE;
  w_cell_t* W = g.weight + (size_t)p * TILE_STRIDE;
  const bool fresh = g.flags[p] == 0;
  const double iw = g.init_weight[p];
  const double t_init = (iw > 0.0) ? 1.0 : __builtin_nan("");
  for (int i = tid; i < TILE_CELLS; i += 256) {
    const int can = canonical_of_off(i);                       // i is the device offset
    const unsigned lx = (unsigned)(can % TILE_PITCH), ly = (unsigned)(can / TILE_PITCH);
    const unsigned col = tx * TILE_DIM + lx, row = ty * TILE_DIM + ly;
    const bool inside = lx < TILE_DIM && ly < TILE_DIM && col >= minX && col < maxX && row >= minY && row < maxY;
    if (inside) { st_tsd(T + i, 1.0); if (fresh) st_w(W + i, iw); }
    else if (fresh) { st_tsd(T + i, t_init); st_w(W + i, iw); }
  }
  __syncthreads();            // every thread has read `fresh`
  if (tid == 0) {
    if (fresh) g.flags[p] = 1;
    dirty[p] = 1;             // its halo (and its neighbours') are refreshed by the next push
  }
}

// PMC calibration (MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are only calibrated for 16 B/lane
// streams): a read-modify-write of n doubles with the push kernel's access shape, 8 B per lane, two
// arrays, so that the counters of a known byte count (16 n read, 16 n written) can be compared with
// what rocprofv3 reports for k_push_tiles.
__global__ void __launch_bounds__(256)
k_calib_rmw(double* __restrict__ t, double* __restrict__ w, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double a = t[i], b = w[i];
    t[i] = a * 0.5 + b; w[i] = b + 1.0;
  }
}

int launch_calibrate(tsd_ctx* ctx, double* t, double* w, size_t n)
{
  hipLaunchKernelGGL(k_calib_rmw, dim3(2048), dim3(256), 0, ctx->stream, t, w, n);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

// ------------------------------------------------------------------------------------------------
// exact `negmask` of a grid whose cells were written by the host (tsd_upload_tiles): one workgroup per tile;
// a negative halo cell is charged to the tile that owns the cell (uploaded halos may be stale)
__global__ void __launch_bounds__(256)
k_neg_scan(GridDev g)
{
  const int p = blockIdx.x;
  if (!g.flags[p]) return;
  const int PX = g.PX, px = p % PX, py = p / PX;
  const tsd_cell_t* T = g.tsd + (size_t)p * TILE_STRIDE;
  for (int i = threadIdx.x; i < TILE_CELLS; i += 256) {
    if (!(ld_tsd(T + i) < 0.0)) continue;
    const int can = canonical_of_off(i);
    const unsigned ix = (unsigned)can % TILE_PITCH, iy = (unsigned)can / TILE_PITCH;
    const int qx = px + (ix == TILE_DIM ? 1 : 0), qy = py + (iy == TILE_DIM ? 1 : 0);
    if (qx >= PX || qy >= PX) continue;                        // (no tile owns the outermost halo)
    atomicOr(&g.negmask[qy * PX + qx], neg_bit(ix & 31u, iy & 31u));
  }
}

// ------------------------------------------------------------------------------------------------
// Canonical tile I/O (tsd_download_tiles / tsd_upload_tiles): the C ABI and the oracle speak 33 x 33 row-major fp64
// tiles (TsdGridPartition::_grid); the device layout (interior + halo strip, fp64 or Q32 cells) is converted here.
// One workgroup per tile of the chunk [t0, t0 + n).
__global__ void __launch_bounds__(256)
k_export_tiles(GridDev g, int t0, double* __restrict__ out_t, double* __restrict__ out_w)
{
  const int p = t0 + blockIdx.x;
  const bool init = g.flags[p] != 0;
  const tsd_cell_t* T = g.tsd + (size_t)p * TILE_STRIDE;
  const w_cell_t* W = g.weight + (size_t)p * TILE_STRIDE;
  double* ot = out_t + (size_t)blockIdx.x * TILE_CELLS;
  double* ow = out_w + (size_t)blockIdx.x * TILE_CELLS;
  for (int i = threadIdx.x; i < TILE_CELLS; i += 256) {
    const int can = canonical_of_off(i);
    ot[can] = init ? ld_tsd(T + i) : __builtin_nan("");      // uninitialised tiles read back NaN / 0
    ow[can] = init ? ld_w(W + i) : 0.0;
  }
}
__global__ void __launch_bounds__(256)
k_import_tiles(GridDev g, int t0, const double* __restrict__ in_t, const double* __restrict__ in_w)
{
  const int p = t0 + blockIdx.x;
  if (!g.flags[p]) return;
  tsd_cell_t* T = g.tsd + (size_t)p * TILE_STRIDE;
  w_cell_t* W = g.weight + (size_t)p * TILE_STRIDE;
  const double* it = in_t + (size_t)blockIdx.x * TILE_CELLS;
  const double* iw = in_w + (size_t)blockIdx.x * TILE_CELLS;
  for (int i = threadIdx.x; i < TILE_CELLS; i += 256) {
    const int can = canonical_of_off(i);
    st_tsd(T + i, it[can]); st_w(W + i, iw[can]);
  }
}
int launch_export_tiles(tsd_ctx* ctx, int t0, int n, double* d_t, double* d_w)
{
  hipLaunchKernelGGL(k_export_tiles, dim3(n), dim3(256), 0, ctx->stream, ctx->grid, t0, d_t, d_w);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}
int launch_import_tiles(tsd_ctx* ctx, int t0, int n, const double* d_t, const double* d_w)
{
  hipLaunchKernelGGL(k_import_tiles, dim3(n), dim3(256), 0, ctx->stream, ctx->grid, t0, d_t, d_w);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

// Digest of the canonical dump (tsd_grid_digest): an order-free 64-bit hash of (tile flag, initWeight, every cell's
// tsd and weight bit patterns, NaN and -0.0 canonicalised) plus sums over the valid cells -- what the cfg 1-3 golden
// fixtures pin without shipping a 4.6 GB dump.  One workgroup per tile; out[p] = {hash, n_valid} as
// u64, sums[p] = {sum tsd, sum weight}; the host adds the per-tile records in tile order.
__device__ __forceinline__ unsigned long long digest_mix(unsigned long long k, unsigned long long a, unsigned long long b)
{
  unsigned long long x = k * 0x9E3779B97F4A7C15ull + a;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  x += b;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return x;
}
__device__ __forceinline__ unsigned long long digest_bits(double v)
{
  if (isnan(v)) return 0x7FF8000000000000ull;
  return (unsigned long long)__double_as_longlong(v + 0.0);      // -0.0 -> +0.0
}
__global__ void __launch_bounds__(256)
k_grid_digest(GridDev g, unsigned long long* __restrict__ out, double* __restrict__ sums)
{
  const int p = blockIdx.x;
  const bool init = g.flags[p] != 0;
  const tsd_cell_t* T = g.tsd + (size_t)p * TILE_STRIDE;
  const w_cell_t* W = g.weight + (size_t)p * TILE_STRIDE;
  unsigned long long h = 0ull, nv = 0ull; double st = 0.0, sw = 0.0;
  if (init) {
    for (int i = threadIdx.x; i < TILE_CELLS; i += 256) {
      const double t = ld_tsd(T + i), w = ld_w(W + i);
      h += digest_mix((unsigned long long)p * 2048ull + 1ull + (unsigned long long)canonical_of_off(i), digest_bits(t), digest_bits(w));
      if (!isnan(t)) { nv++; st += t; sw += w; }
    }
  }
  __shared__ unsigned long long s_h[256], s_n[256]; __shared__ double s_t[256], s_w[256];
  s_h[threadIdx.x] = h; s_n[threadIdx.x] = nv; s_t[threadIdx.x] = st; s_w[threadIdx.x] = sw;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      s_h[threadIdx.x] += s_h[threadIdx.x + k]; s_n[threadIdx.x] += s_n[threadIdx.x + k];
      s_t[threadIdx.x] += s_t[threadIdx.x + k]; s_w[threadIdx.x] += s_w[threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[2 * p] = s_h[0] + digest_mix((unsigned long long)p * 2048ull, init ? 1ull : 0ull, digest_bits(g.init_weight[p]));
    out[2 * p + 1] = s_n[0];
    sums[2 * p] = s_t[0]; sums[2 * p + 1] = s_w[0];
  }
}
int launch_grid_digest(tsd_ctx* ctx, unsigned long long* d_out, double* d_sums)
{
  hipLaunchKernelGGL(k_grid_digest, dim3(ctx->grid.tiles), dim3(256), 0, ctx->stream, ctx->grid, d_out, d_sums);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_neg_scan(tsd_ctx* ctx)
{
  const GridDev& g = ctx->grid;
  TSD_HIP_CHECK(ctx, hipMemsetAsync(g.negmask, 0, (size_t)g.tiles * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(k_neg_scan, dim3(g.tiles), dim3(256), 0, ctx->stream, g);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_free_footprint(tsd_ctx* ctx, unsigned minX, unsigned maxX, unsigned minY, unsigned maxY)
{
  if (maxX <= minX || maxY <= minY) return TSD_OK;
  const unsigned tx0 = minX / TILE_DIM, tx1 = (maxX - 1) / TILE_DIM;
  const unsigned ty0 = minY / TILE_DIM, ty1 = (maxY - 1) / TILE_DIM;
  const unsigned ntx = tx1 - tx0 + 1, nty = ty1 - ty0 + 1;
  hipLaunchKernelGGL(k_free_footprint, dim3(ntx * nty), dim3(256), 0, ctx->stream, ctx->grid, minX,
                     maxX, minY, maxY, tx0, ty0, ntx, ctx->d_dirty);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  TileBox b; b.x0 = (int)tx0; b.y0 = (int)ty0; b.x1 = (int)tx1; b.y1 = (int)ty1;
  ctx->box_dirty.add(b);             // the next push refreshes the halos there
  return TSD_OK;
}

size_t push_rmq_bytes(int beams) { return rmq_bytes(beams); }

int launch_push_tables(tsd_ctx* ctx, hipStream_t stream, int beams, const double* d_ranges, const uint8_t* d_mask,
                       double phi_min, double ang_res)
{
  const size_t bp = (size_t)((beams + 3) & ~3);
  const size_t lds = 2 * bp * sizeof(double) + 4 * bp * 2 + 64;
  std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
  if (lds > ctx->tables_lds_configured) {     // the attribute is per device: remembered per context, not per process
    TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_push_tables),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->tables_lds_configured = lds;
  }
  char* rmq;
  if (launch_target() && launch_target()->rmq) rmq = launch_target()->rmq;       // concurrent multi-robot path: the sensor's own (double) buffer
  else {
    ctx->rmq_slot ^= 1;                                  // the push that may still be running keeps its tables
    ctx->d_rmq = ctx->d_rmq2[ctx->rmq_slot];
    rmq = ctx->d_rmq;
  }
  hipLaunchKernelGGL(k_push_tables, dim3(1), dim3(1024), lds, stream, d_ranges ? d_ranges : ctx->d_ranges,
                     d_mask ? d_mask : ctx->d_mask, beams, rmq, phi_min, ang_res);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_push_tables_batch(tsd_ctx* ctx, hipStream_t stream, const TablesBatchEntry* d_entries, int n, int max_beams)
{
  const size_t bp = (size_t)((max_beams + 3) & ~3);
  const size_t lds = 2 * bp * sizeof(double) + 4 * bp * 2 + 64;
  {
    std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
    size_t& configured = ctx->lds_configured[reinterpret_cast<const void*>(k_push_tables_batch)];
    if (lds > configured) {
      TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_push_tables_batch),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured = lds;
    }
  }
  hipLaunchKernelGGL(k_push_tables_batch, dim3(n), dim3(1024), lds, stream, d_entries);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

// the tables of this scan must already be in ctx->d_rmq (launch_push_tables, ordered before this)
int launch_push(tsd_ctx* ctx, const PushArgs& a, double cx, double cy, double slack, const PushArgs* a_dev,
                const double* d_ranges, const uint8_t* d_mask)
{
  const GridDev& g = ctx->grid;
  if (!a_dev) return set_error(ctx, TSD_E_ARG, "launch_push: the arguments must be on the device", hipSuccess);
  char* const rmq = (launch_target() && launch_target()->rmq) ? launch_target()->rmq : ctx->d_rmq;
  if (!d_ranges) d_ranges = ctx->d_ranges;
  if (!d_mask) d_mask = ctx->d_mask;
  // Tile window: a tile passes the range cull of isInRange only if its centre is within
  // max_range + radius + max_trunc of the sensor (TsdGridComponent.cpp:52-60); the sensor is within `slack`
  // of (cx, cy).  The window also covers the previous push (its records are rewritten) and whatever
  // freeFootprint touched since.  sensor max_range comes from the by-value args or the attached sensor.
  TileBox box;
  {
    const double tile = TILE_DIM * g.cs;
    const double reach = a.max_range + 0.75 * tile + g.max_trunc + slack + g.cs;      // radius = sqrt(2)/2 tile < 0.75 tile
    const double last = (double)(g.PX - 1);
    const double fx0 = floor((cx - reach) / tile) - 1.0, fy0 = floor((cy - reach) / tile) - 1.0;
    const double fx1 = floor((cx + reach) / tile) + 1.0, fy1 = floor((cy + reach) / tile) + 1.0;
    if (!(reach < 1e300) || !(fx0 == fx0)) { box.x0 = 0; box.y0 = 0; box.x1 = g.PX - 1; box.y1 = g.PX - 1; }
    else {
      box.x0 = (int)fmax(0.0, fmin(last, fx0)); box.y0 = (int)fmax(0.0, fmin(last, fy0));
      box.x1 = (int)fmax(0.0, fmin(last, fx1)); box.y1 = (int)fmax(0.0, fmin(last, fy1));
    }
  }
  const TileBox cur = box;
  box.add(ctx->box_prev);
  box.add(ctx->box_dirty);
  ctx->box_prev = cur; ctx->box_dirty = TileBox{};
  const int ntx = box.x1 - box.x0 + 1, nty = box.y1 - box.y0 + 1;
  const int parity = (int)(ctx->push_parity & 1u);
  ctx->push_parity++;
  const int n_window = ntx * nty;
  {
    ScopedKernelTimer t(ctx, "push_classify");
    hipExtLaunchKernelGGL(k_push_classify, dim3((n_window + 15) / 16), dim3(64), 0, ctx->stream, t.a, t.b, 0, g, a_dev, rmq,
                       ctx->d_tile_rec, ctx->d_dirty, ctx->d_tile_totals, ctx->d_list, ctx->d_list_win, ctx->d_list_pw, ctx->d_list_cnt, parity,
                       box.x0, box.y0, ntx, nty);
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  const int n_groups = n_window < 2048 ? n_window : 2048;      // resident at once; a longer list is looped over
  {
    ScopedKernelTimer t(ctx, "push_update");
    const size_t lds = 16 + (size_t)((a.beams + 1) & ~1) * sizeof(double) + (size_t)((a.beams + 15) & ~15);
    hipExtLaunchKernelGGL(k_push_update, dim3(n_groups), dim3(UPDATE_BLOCK), lds, ctx->stream, t.a, t.b, 0, g, a_dev, d_ranges, d_mask,
                       ctx->d_tile_rec, ctx->d_tile_totals, ctx->d_list, ctx->d_list_win, ctx->d_list_pw, ctx->d_list_cnt, parity,
                       ctx->d_icp_trace);
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  {
    ScopedKernelTimer t(ctx, "push_halo");
    const int n_waves = n_window < 4096 ? n_window : 4096;        // one wave per listed tile; a longer list is looped over
    hipExtLaunchKernelGGL(k_push_halo, dim3((n_waves + 3) / 4), dim3(256), 0, ctx->stream, t.a, t.b, 0, g, ctx->d_dirty, ctx->d_pushes,
                       a_dev, ctx->d_list, ctx->d_tile_rec, ctx->d_list_cnt, parity, cx, cy, slack + g.cs);
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

}  // namespace tsd
