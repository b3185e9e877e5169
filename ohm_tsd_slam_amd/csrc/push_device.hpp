// push_device.hpp -- what the push kernels share (push_kernels.hip: one scan into the grid; push_multi.hip: the scans of a batch of robots
// into the grid in one pass per tile): the per-tile records, the range-query tables' layout, tile geometry, the correctly rounded
// square root / quotient, addTsd, the work-list constants and records, and the fp32 beam estimate of the update kernels.
#pragma once
#include "tsd_ctx.hpp"
#include <climits>

namespace tsd {

constexpr int UPDATE_BLOCK = 256;                  // threads of the per-tile workgroup

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
  const double2* bdir;              // [B + 1] unit vectors of the beam boundaries beta_j = phi_min + (j - 0.5) * res
  const double2* rot;               // [ROT_N] (cos, sin) of k * res: beta_(j0 + k) = beta_j0 turned by rot[k] (k_push_update's fix-up, in LDS)
  int Bp, levels;
};
constexpr int ROT_N = 128;

__host__ __device__ inline int rmq_levels(int beams) { int l = 1; while ((1 << l) <= beams) l++; return l; }
__host__ __device__ inline size_t rmq_bytes(int beams)
{
  const size_t bp = (size_t)((beams + 3) & ~3);
  return 2 * bp * sizeof(double) + ((size_t)(beams + 1 + 7) & ~(size_t)7) * 2 + 2 * (size_t)rmq_levels(beams) * bp * 2 + 64 +
         ((size_t)beams + 2) * sizeof(double2) + 128 * sizeof(double2);
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
  v.rot = v.bdir + ((size_t)beams + 2);
  return v;
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
  // AMDGPU's f64 sqrt lowering without the 2^+-256 scaling and the inf pass-through.  Zero -- the cell whose centre IS the sensor
  // position, to the last bit: a start pose configured onto a cell centre (tools/fuzz_parity.py, exact poses) -- makes the core 0 * inf
  // = NaN, which silently skipped that cell's update: the closing v_max_f64 with 0 turns exactly that NaN into the root of zero (IEEE
  // maximum: the operand that is a number) and leaves every other result bit for bit what it was.
  const double y = __builtin_amdgcn_rsq(x);
  const double g0 = x * y, h0 = 0.5 * y;
  const double r0 = __builtin_fma(-h0, g0, 0.5);
  const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
  const double d0 = __builtin_fma(-g1, g1, x);
  const double g2 = __builtin_fma(d0, h1, g1);
  const double d1 = __builtin_fma(-g2, g2, x);
  const double r = __builtin_fma(d1, h1, g2);
  double out;
  asm("v_max_f64 %0, %1, 0" : "=v"(out) : "v"(r));
  return out;
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

// TsdGridPartition::addTsd (TsdGridPartition.h:170-212).  The reference's `if(fabs(sd) < _eps) w = 1.0` never fires: _eps is
// -cellSize / 2 (TsdGridPartition.cpp:95), negative for every grid tsd_create accepts, and fabs() is not -- so the weight of a
// measurement is 0.01 * the partition weight for every cell of the tile, `w` here, formed once per tile.
__device__ __forceinline__ bool add_tsd(double& tsd, double& weight, double sd, double w,
                                        double max_trunc, double inv_max_trunc)
{
  if (sd >= -max_trunc) {
    const double v = fmin(sd * inv_max_trunc, 1.0);
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

// The work of one push is two lists in ONE array of `tiles` words: UPDATE tiles from the front (list[0 .. nU)), everything else that
// needs a visit -- increaseEmptiness of a materialised tile, tiles whose halos freeFootprint dirtied -- from the back
// (list[tiles - 1 - k], k < nO).  Counters, by push parity: cnt[CNT_WORDS * parity + {0: nU, 1: nO}], then the TICKET HEADS of
// k_push_update's tile queue: TICKET_HEADS counters, each on a 128-byte line of its own (one word shared by every workgroup
// saturates at ~88 returning atomics per microsecond, MI355X_MICROARCH.md "dequeue": a 10 000-tile push would take 120 us for its
// tickets alone).  Head h hands out the tiles G + h + TICKET_HEADS * k beyond the G that the G workgroups start with.
constexpr int CNT_U = 0, CNT_O = 1, CNT_H = 2 /* UPDATE tiles k_push_halo has work for (list_h) */, CNT_HDONE = 3 /* halo-list entries done (the ray cast's prologue) */, TICKET_HEADS = 32, TICKET_STRIDE = 32 /* words */, CNT_TICKET = 32;
constexpr int CNT_WORDS = CNT_TICKET + TICKET_HEADS * TICKET_STRIDE;
// What k_push_classify leaves for the workgroup of an UPDATE tile: one 128-byte record (a cache line, at the entry's own index),
// fetched as ONE vector register per wave -- lane i holds word i -- and unpacked with v_readlane.
struct PushListAuxBody {
  uint32_t entry, win;           // the list word (tile | flags | kind << 28); beams the tile's cells can project to: lo | hi << 16
  double pw;                     // 0.01 * partition weight (TsdGrid.cpp:239-243; TsdGridPartition.h:193-196: w = 0.01, then w *= the weight):
                                 // the product is formed where the weight is, so that the update kernel needs no vector-register constant
  // phase A of k_push_update (fp32 beam estimate), all relative to the tile's centroid c = ((x0 + 16.5) cs, (y0 + 16.5) cs) -- the
  // centre of cell (ix, iy) is c + (ix - 16, iy - 16) cs -- with l_c = PoseInv (c, 1) and M = PoseInv's rotation * cs:
  float A, B;                    // l_c x (M d) = dx A + dy B   (d = cell offset in cells)
  float C, D;                    // l_c . (M d) = dx C + dy D
  float lc2, th_c;               // |l_c|^2, angle of l_c (atan2_estimate)
  float lcx, lcy;                // l_c itself (near tiles)
  double iw;                     // the tile's _initWeight and ...
  uint32_t flag, jb0;            // ... _initialized when it was classified (nothing changes them before the tile's own workgroup does);
                                 // jb0 = max(lo - 1, 0): the boundary the tile's fix-up starts from, and ...
  double2 bd;                    // ... its direction (cos, sin)(beta_jb0), words 16..19
};
struct alignas(128) PushListAux : PushListAuxBody {};      // (the writer stores the 80 bytes of the body only)
static_assert(sizeof(PushListAuxBody) == 80 && sizeof(PushListAux) == 128, "PushListAux");

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
// Workgroup barrier that orders LDS only.  __syncthreads() also drains the wave's global-memory counter (s_waitcnt vmcnt(0)): every
// wave would sit out the full latency of the stores it has just issued at the end of each tile.  Nothing in k_push_update hands
// GLOBAL data from one wave to another inside the launch (a cell is read and written by one lane), so LDS order is all it needs.
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Classification of one cell by its estimated angle `th` (radians, possibly outside (-pi, pi] by the small delta).  The beam
// coordinate u = (angle - phi_min) / res is shifted by one half, v = u + 0.5, so that the rounding boundaries of
// round() -- and the two ends of the field of view, phi_lower = phi_min - res / 2 and phi_upper = phi_min + (beams - 0.5) res --
// all sit at INTEGER v: boundary jb (0 .. beams) is the direction beta_jb = phi_min + (jb - 0.5) res.
//   decided inside   (!uns && !out)  j = the beam (v further than IDX_MARGIN from every boundary)
//   decided outside  (out)           outside the field of view
//   undecided        (uns)           within IDX_MARGIN of boundary j (0 .. beams), or -- j = IDX_CUT -- at the +-pi cut of atan2
// The three answers are lane masks, never integers in a vector register: the compiler keeps them in scalar register pairs and the
// ballots of the compaction are those pairs.
constexpr int IDX_CUT = 0x1FFF;
struct CellClass { int j; bool uns, out; };
__device__ __forceinline__ CellClass classify_angle(float th, float phi_min_f, float inv_res_f, int beams)
{
  const float PI_F = 3.14159274f;
  if (th > PI_F) th -= 2.0f * PI_F;                          // the reference's atan2 lives in (-pi, pi]
  else if (th <= -PI_F) th += 2.0f * PI_F;
  const bool cut = fabsf(th) > PI_F - 1e-3f;                 // at the cut the two branches are 2 pi apart: exact path
  const float v = fmaf(th - phi_min_f, inv_res_f, 0.5f);    // beam coordinate + 1/2
  const float vb = (float)beams;
  const bool outside = v < -IDX_MARGIN || v > vb + IDX_MARGIN;        // outside the field of view for sure
  // whatever is not STRICTLY inside by the margin belongs to the end's boundary (0 / beams): a decided beam is always 0 .. beams - 1
  const bool end = !(v > IDX_MARGIN && v < vb - IDX_MARGIN);
  const float jf = rintf(v);
  const bool close = !(fabsf(v - jf) >= IDX_MARGIN);         // (also a NaN: never a decided beam)
  CellClass c;
  c.uns = cut || (!outside && (end || close));
  c.out = !cut && outside;
  c.j = cut ? IDX_CUT : end ? (v < 1.0f ? 0 : beams) : close ? (int)jf : (int)floorf(v);
  return c;
}

// SensorPolar2D::backProject itself (fp64 atan2, bound checks, round) for the cells nothing cheaper can decide: OUT OF LINE on
// purpose.  Inlined, the atan2 expansion's ~40 extra live registers would be part of k_push_update's allocation (64 VGPRs = 8
// waves per SIMD) although the path runs for a handful of cells per push; as a call the caller's registers are saved around it
// only when it is taken.  The arguments are re-read from memory for the same reason.
static __device__ __noinline__ int backproject_cold(const PushArgs* __restrict__ a_dev, double x, double y)
{
  const PushArgs a = *a_dev;
  return backproject(a.Pi, x, y, a.phi_min, a.ang_res_inv, a.phi_lower, a.phi_upper);
}

// fp32 limit of the squared sensor distance up to which a cell of beam (r, mask) can be touched by addTsd: the cells that lie
// behind the surface by more than the truncation FOR SURE are beyond it (|l|^2 against (range + maxTruncation)^2 with a 1e-5
// margin, fp32 being good to 6e-7 here); an infinite reading updates up to lowReflectivityRange; a masked beam never (-1).
__device__ __forceinline__ float beam_limit(double r, unsigned mk, float mtf, float low2f)
{
  if (mk == 0u) return -1.0f;
  if (isinf(r)) return low2f;
  const float rf = (float)r + mtf;
  return rf * rf * 1.00001f;
}

// One workgroup per listed tile (TsdGrid.cpp:237-274), up to EIGHT workgroups per compute unit.  Round-3 structure: the phases of a
// tile talk to each other through LDS and keep next to nothing in registers across their boundaries, so that the kernel fits 64
// VGPRs (8 waves per SIMD; the round-2 kernel needed 128 and was bound by instruction issue at 4 waves per SIMD with ~175
// instructions per visited cell, 52 % of which were not updated):
//   staging  once per workgroup: the scan's ranges (fp64) and a per-beam fp32 distance limit (beam_limit) in LDS
//   phase A  fp32 only, 4 cells per thread: beam coordinate from the estimate above, classification, and for decided cells the
//            candidate test -- one LDS read and one compare against the beam's limit.  Candidates are COMPACTED into an LDS list
//            (cell | beam << 10), one LDS atomic per wave; cells within IDX_MARGIN of a boundary go to a wave-local list
//   fix-up   (same wave, no barrier) the undecided cells, densely, one lane each: the side of the boundary direction beta_jb the
//            cell's fp64 sensor-frame vector lies on -- the sign of |l| sin(angle - beta) = bx ly - by lx, good to 1e-16 where
//            the reference's own rounding chain is good to 1e-15 -- names the reference's beam unless |sin| < 1e-11; those cells,
//            and cells at the +-pi cut, take the reference's formulation itself (fp64 atan2), a cold path
//   phase C  the exact part over the COMPACTED candidates, full waves, one cell per lane and pass: tsd / weight reads, the IEEE
//            distance, signed distance, addTsd (TsdGridPartition.h:170-212), the writes
// Lazy TsdGridPartition::init (TsdGridPartition.cpp:88-134) is folded in (a fresh tile's old value is known: non-candidates get
// the init value from phase A / the fix-up, candidates start from it in phase C); KIND_EMPTY: increaseEmptiness over the 33x33
// cells.  The workgroup leaves the tile's record and adds it to the tile's running totals (no-return atomics).
// (Measured and not kept, rounds 3-4: single-tile workgroups requesting their tile's 128 lines ahead of phase A -- 0.4 us off the kernel inside
// the SLAM loop for 5.5 MB of reads per push that nothing uses, traffic / algorithmic bytes 1.63 against 1.26; requesting only the lines that
// hold candidates: slower, eight more live registers spill.)
constexpr int UPDATE_WPS = 5;                                    // resident workgroups per SIMD the launch bounds ask for
constexpr int UPD_CAND_MAX = TILE_INTERIOR;
constexpr int UPD_CPT = TILE_INTERIOR / UPDATE_BLOCK;            // cells per thread: 4
constexpr int UPD_CB = 2;                                        // exact part: cells per lane and pass
__host__ __device__ inline size_t update_lds_bytes(int beams)
{
  const size_t bp = (size_t)((beams + 3) & ~3);
  return bp * sizeof(double) + 2 * 2 * TILE_DIM * sizeof(double) + ROT_N * sizeof(double2) + bp * sizeof(float) +
         2 * UPD_CAND_MAX * sizeof(uint32_t);
}

// wave-uniform values of phase A
struct TileA {
  float A, B, C, D, lc2, th_c, lcx, lcy;     // PushListAux
  float axx, axy, ayx, ayy;                   // PoseInv's rotation * cellSize (near tiles)
  float cs2;                                  // cellSize^2
  float phi_min, inv_res, mt, low2;
  int beams, wlo, whi;
};

// Beam classification of ONE cell, offset (dxc, dyc) cells from the tile's centroid (CellClass: decided beam / decided outside the
// field of view / undecided with its boundary); d2f = the fp32 squared sensor distance.  FAR: the sensor is further
// than three circumradii from the centroid -- the cell's angle is the centroid's plus a small delta, |delta| < 0.34 rad,
// tan(delta) = cross / dot with both products LINEAR in the cell offset (coefficients from k_push_classify), atan by a four-term
// series; near tiles use the six-term minimax arctangent.  INTERIOR (implies FAR): no end of the field of view, no cut -- only
// rounding boundaries.
template <bool FAR, bool INTERIOR>
__device__ __forceinline__ CellClass classify_cell(const TileA& t, float dxc, float dyc, float pA, float pC, float qx, float vc, float& d2f)
{
  float th_rel;          // angle relative to th_c (FAR) or the angle itself
  if constexpr (FAR) {
    const float cr = fmaf(dyc, t.B, pA);                        // l_c x l
    const float dt = fmaf(dyc, t.D, pC);                        // l_c . l  (> 0: |delta| < 0.34 rad)
    d2f = fmaf(2.0f, dt, fmaf(t.cs2 * dyc, dyc, qx));          // |l|^2 = 2 l_c.l - |l_c|^2 + cs^2 |d|^2
    const float tt = cr * __builtin_amdgcn_rcpf(dt);
    const float t2 = tt * tt;
    float r = fmaf(t2, -0.142857143f, 0.2f);
    r = fmaf(r, t2, -0.333333333f);
    r = fmaf(r, t2, 1.0f);
    th_rel = r * tt;
  } else {
    const float lxf = fmaf(t.axy, dyc, fmaf(t.axx, dxc, t.lcx)), lyf = fmaf(t.ayy, dyc, fmaf(t.ayx, dxc, t.lcy));
    d2f = fmaf(lxf, lxf, lyf * lyf);
    th_rel = atan2_estimate(lyf, lxf);
  }
  if constexpr (INTERIOR) {
    // v = (th_c + delta - phi_min) / res + 1/2: boundaries at integer v
    const float v = fmaf(th_rel, t.inv_res, vc);
    const float jf = rintf(v);
    CellClass c;
    c.uns = !(fabsf(v - jf) >= IDX_MARGIN); c.out = false;
    c.j = c.uns ? (int)jf : (int)floorf(v);
    return c;
  } else {
    return classify_angle(FAR ? t.th_c + th_rel : th_rel, t.phi_min, t.inv_res, t.beams);
  }
}

// ---- TsdGrid::propagateBorders (TsdGrid.cpp:372-427), incremental form, for ONE listed tile by ONE wave --------------------------------
// The tile's own halo from R / U / UR and the halos of L / D / DL that mirror its first column / row / cell.  Equal to the reference's
// full sweep by induction (untouched pairs are already consistent).  Every lane has up to three copy jobs (source cell -> destination
// cell, value and weight); all loads of a tile are issued before the first store, so a tile costs one memory round trip after its flags.
// Lanes 0..31: the two column copies; lanes 32..63: the two row copies; lane 0 / lane 32: the corner cells.
// Shared by k_push_halo (a kernel of its own behind k_push_update) and by k_raycast's prologue (the fused scan: the NEXT scan's ray cast
// is the kernel behind the push, and its first waves do this before they cast: raycast_kernels.hip).  The destination cells are stored
// WRITE-THROUGH at agent scope (`sc1`): in the ray cast's prologue they are handed to other waves of the same launch.
struct HaloArgs {
  uint8_t* dirty; unsigned long long* pushes; const PushArgs* a_dev;
  const uint32_t* list; const uint32_t* list_h; const uint32_t* tile_rec;
  unsigned int* cnt;              // this push's counter words (list_cnt + CNT_WORDS * parity)
  double cx, cy, slack;           // where the host laid the push's window, and how far the sensor may be from it
  int on;                         // 0: no push ahead of this launch (nothing to do)
};
template <typename T>
__device__ __forceinline__ void st_agent(T* p, T v)
{
  if constexpr (sizeof(T) == 8) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __builtin_bit_cast(unsigned int, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void halo_tile_job(const GridDev& g, uint8_t* __restrict__ dirty, const uint32_t* __restrict__ tile_rec, uint32_t entry, int lane)
{
  const int PX = g.PX;
  const bool colhalf = lane < TILE_DIM;
  const int i = lane & 31;
  const int p = (int)(entry & LIST_TILE_MASK);
  const int px = p % PX, py = p / PX;
  const bool hasR = px < PX - 1, hasU = py < PX - 1, hasL = px > 0, hasD = py > 0;
  // all nine flags, the three records and the dirty mark in flight at once: UNCONDITIONAL reads (a missing neighbour reads the
  // tile itself and the value is dropped).  As `has ? flags[q] : 0` each read was predicated, and the compiler waits for a
  // predicated read on the spot -- up to twelve memory round trips in a row per tile (round 3, ISA of rounds 1-2).
  const int qR = hasR ? p + 1 : p, qU = hasU ? p + PX : p, qUR = (hasR && hasU) ? p + PX + 1 : p;
  const int qL = hasL ? p - 1 : p, qD = hasD ? p - PX : p, qDL = (hasL && hasD) ? p - PX - 1 : p;
  const uint8_t f0 = g.flags[p], dty = dirty[p];
  const uint8_t fR_ = g.flags[qR], fU_ = g.flags[qU], fUR_ = g.flags[qUR], fL_ = g.flags[qL], fD_ = g.flags[qD], fDL_ = g.flags[qDL];
  const uint32_t rL_ = tile_rec[qL], rD_ = tile_rec[qD], rDL_ = tile_rec[qDL];
  const uint32_t r0 = tile_rec[p];
  // ... and the SOURCE cells of the three copy jobs with them (where they are depends on the geometry only; the tile storage exists for
  // every tile): the pass is a chain of memory round trips -- in the ray cast's prologue every other wave of the launch waits for its
  // end -- and this is one of them less.  job 0: own halo from the right / upper neighbour; job 1: the left / lower neighbour's halo
  // from this tile; job 2 (lanes 0 and 32 only): the corner cells
  const size_t own = (size_t)p * TILE_STRIDE;
  size_t src[3], dst[3]; bool can[3];
  if (colhalf) {
    can[0] = hasR;  src[0] = (size_t)qR * TILE_STRIDE + (size_t)i * TILE_DIM;   dst[0] = own + HALO_COL + i;
    can[1] = hasL;  src[1] = own + (size_t)i * TILE_DIM;                       dst[1] = (size_t)qL * TILE_STRIDE + HALO_COL + i;
    can[2] = lane == 0 && hasR && hasU; src[2] = (size_t)qUR * TILE_STRIDE;     dst[2] = own + HALO_ROW + TILE_DIM;
  } else {
    can[0] = hasU;  src[0] = (size_t)qU * TILE_STRIDE + i;                      dst[0] = own + HALO_ROW + i;
    can[1] = hasD;  src[1] = own + i;                                          dst[1] = (size_t)qD * TILE_STRIDE + HALO_ROW + i;
    can[2] = lane == 32 && hasL && hasD; src[2] = own;                          dst[2] = (size_t)qDL * TILE_STRIDE + HALO_ROW + TILE_DIM;
  }
  tsd_cell_t tv[3]; w_cell_t wv_[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { const size_t sk = can[k] ? src[k] : own; tv[k] = ld_pinned(&g.tsd[sk]); wv_[k] = ld_pinned(&g.weight[sk]); }
  if (lane == 0 && dty != 0) dirty[p] = 0;
  const uint8_t fR = hasR ? fR_ : (uint8_t)0, fU = hasU ? fU_ : (uint8_t)0, fUR = (hasR && hasU) ? fUR_ : (uint8_t)0;
  uint8_t fL = hasL ? fL_ : (uint8_t)0, fD = hasD ? fD_ : (uint8_t)0, fDL = (hasL && hasD) ? fDL_ : (uint8_t)0;
  // a left / lower / diagonal neighbour that is on this push's list refreshes its own halo from this tile itself
  // (its job 0 / corner job is the very same copy): skipping the mirror job halves the column gathers where the
  // listed tiles are dense.  Records outside this push's window are never "listed" (see launch_push).
  const uint32_t rL = hasL ? rL_ : 0u, rD = hasD ? rD_ : 0u, rDL = (hasL && hasD) ? rDL_ : 0u;
  if (!f0) return;
  // An UPDATE tile that held data before this push and was not touched by freeFootprint has nothing to do here: what it changed of
  // its column 0 / row 0 / corner, its own workgroup wrote into the neighbours' halos (k_push_update's mirror pass), and what its
  // right / upper neighbours changed arrived the same way or is brought by THEIR jobs below.  Left for this pass: tiles
  // materialised by this push (everything around them), increaseEmptiness tiles (all 33 x 33 cells changed, own halo included),
  // tiles freeFootprint wrote to.
  const bool plain_u = (entry >> KIND_SHIFT) == KIND_UPDATE && !(r0 & REC_NEW) && dty == 0;
  if (plain_u) return;
  // (a neighbour that is on the list and surely does its own job 0 -- materialised, emptied or halo-only this push -- needs no
  // mirror job from here; a plain UPDATE neighbour does nothing in this pass, so it does)
  auto own_job = [](uint32_t r) { return (r & REC_LISTED) != 0u && !((r & REC_UPDATE) != 0u && !(r & REC_NEW)); };
  if (own_job(rL)) fL = 0;
  if (own_job(rD)) fD = 0;
  if (own_job(rDL)) fDL = 0;
  const bool on[3] = {can[0] && (colhalf ? fR : fU) != 0, can[1] && (colhalf ? fL : fD) != 0, can[2] && (colhalf ? fUR : fDL) != 0};
#pragma unroll
  for (int k = 0; k < 3; k++) if (on[k]) { st_agent(&g.tsd[dst[k]], tv[k]); st_agent(&g.weight[dst[k]], wv_[k]); }
}
// the push's bookkeeping that rides with the halo pass (one lane of the launch): the push counter and the check of the window's assumption
__device__ __forceinline__ void halo_bookkeeping(unsigned long long* __restrict__ pushes, const PushArgs* __restrict__ a_dev, double cx, double cy, double slack)
{
  if (a_dev->enabled != 0) {
    pushes[0] += 1ull;
    // the window was laid around (cx, cy) +- slack by the host: a sensor outside of that is a host-side bug
    const double sx = a_dev->trx, sy = a_dev->try_;
    if (!(fabs(sx - cx) <= slack && fabs(sy - cy) <= slack)) pushes[1] += 1ull;
  }
}

// per-tile values the exact part (phase C) and the tile's record need; two tiles are in flight per workgroup
struct TileC {
  tsd_cell_t* T; w_cell_t* W;
  double pw, iw;
  int p;
  bool fresh;
};

}  // namespace tsd
