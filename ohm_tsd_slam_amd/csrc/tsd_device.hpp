// tsd_device.hpp -- device-side view of the TSD grid and the small fp64 helpers every kernel shares.
//
// Data layout in HBM (DESIGN.md "Data layout"): tile-major structure-of-arrays.  Tile p = py*PX + px
// (row-major like TsdGrid::_partitions[0][p], TsdGrid.cpp:234) owns TILE_STRIDE cells in `tsd` and in
// `weight`, laid out so that everything a push streams is whole 128-byte lines:
//     [0, 1024)      the 32 x 32 interior, row-major (cell (ix, iy) at iy * 32 + ix): a row is 256 contiguous
//                    bytes (fp64) on a 128-byte boundary, a tile starts on a 128-byte boundary
//     [1024, 1056)   halo column  (ix == 32, iy = 0..31)   } the duplicated 1-cell halo of
//     [1056, 1089)   halo row     (iy == 32, ix = 0..32)   } TsdGridPartition.cpp:97, kept with its stale-halo semantics
//     [1089, 1120)   padding
// (round 1 kept the reference's 33-cell row pitch: every 256-byte interior row then straddled five 64-byte
// segments instead of four and the update kernel moved 1.4-1.5 x its algorithmic bytes.)
// `flags[p]` is TsdGridPartition::_initialized, `init_weight[p]` is _initWeight.  All arithmetic is fp64 in the
// reference's operation order; translation units are compiled with -ffp-contract=off.
//
// Cell storage is a compile-time choice:
//   default            fp64 like the reference (obfloat == double): cells bit-identical to the oracle's
//   -DTSD_STORAGE_Q32  32-bit fixed point, 8 bytes per cell instead of 16: tsd as signed Q1.30 (NaN = INT32_MIN),
//                      weight as unsigned Q6.26 (0 .. 32).  Arithmetic stays fp64; a store rounds to the nearest
//                      grid value, so after n pushes a cell is within n * 2^-31 (tsd) / n * 2^-27 (weight) of the
//                      fp64 result whatever the data -- north_star's 1e-5 holds for > 1000 pushes by construction.
//                      (fp32 cells do not give that: a weight near 10 has an ulp of 9.5e-7 and the rounding errors of
//                      1000 additions random-walk to ~1e-5; tests/test_cpu_oracle_properties.py shows it.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsd {

constexpr int TILE_DIM    = 32;
constexpr int TILE_PITCH  = 33;            // of the CANONICAL 33 x 33 dump (C ABI, oracle); not the device layout
constexpr int TILE_CELLS  = 33 * 33;
constexpr int TILE_INTERIOR = TILE_DIM * TILE_DIM;
constexpr int HALO_COL    = TILE_INTERIOR;              // + iy
constexpr int HALO_ROW    = TILE_INTERIOR + TILE_DIM;   // + ix (0..32)
constexpr int TILE_STRIDE = 1120;          // 1089 rounded up to whole 128-byte lines for 4-byte and 8-byte cells
constexpr double MAX_WEIGHT = 32.0;        // TSDGRIDMAXWEIGHT, reconstruct_defs.h:4

// device offset of cell (ix, iy), 0..32 each, inside its tile
__host__ __device__ __forceinline__ int cell_off(int ix, int iy)
{
  return iy < TILE_DIM ? (ix < TILE_DIM ? iy * TILE_DIM + ix : HALO_COL + iy) : HALO_ROW + ix;
}
// device offset -> canonical index iy * 33 + ix
__host__ __device__ __forceinline__ int canonical_of_off(int off)
{
  if (off < TILE_INTERIOR) return (off >> 5) * TILE_PITCH + (off & 31);
  if (off < HALO_ROW) return (off - HALO_COL) * TILE_PITCH + TILE_DIM;
  return TILE_DIM * TILE_PITCH + (off - HALO_ROW);
}

#ifdef TSD_STORAGE_Q32
using tsd_cell_t = int32_t;
using w_cell_t   = uint32_t;
constexpr bool STORAGE_EXACT = false;
constexpr int32_t Q_NAN = INT32_MIN;
__device__ __forceinline__ double ld_tsd(const tsd_cell_t* p) { const int32_t q = *p; return q == Q_NAN ? __builtin_nan("") : ldexp((double)q, -30); }
__device__ __forceinline__ void st_tsd(tsd_cell_t* p, double v) { *p = isnan(v) ? Q_NAN : (int32_t)rint(ldexp(v, 30)); }
__device__ __forceinline__ double ld_w(const w_cell_t* p) { return ldexp((double)*p, -26); }
__device__ __forceinline__ void st_w(w_cell_t* p, double v) { *p = (uint32_t)rint(ldexp(v, 26)); }
#else
using tsd_cell_t = double;
using w_cell_t   = double;
constexpr bool STORAGE_EXACT = true;
__device__ __forceinline__ double ld_tsd(const tsd_cell_t* p) { return *p; }
__device__ __forceinline__ void st_tsd(tsd_cell_t* p, double v) { *p = v; }
__device__ __forceinline__ double ld_w(const w_cell_t* p) { return *p; }
__device__ __forceinline__ void st_w(w_cell_t* p, double v) { *p = v; }
#endif

// store one cell (device offset `off` inside the tile).  (Round 3 tried a packed copy of interior column 0 per tile, kept up to date
// here, so that the halo refresh reads 256 contiguous bytes instead of 32 cache lines per column: k_push_halo 24.4 -> 13.5 us at
// cfg3 / comb, but the extra predicate + stores per candidate cost the VALU-bound k_push_update 9 us there and 0.5 us at cfg2, where
// the halo kernel is a latency chain and gained nothing.  Net zero / negative: taken out again.  profiles/r3_push_update_structure.txt.)
__device__ __forceinline__ void st_cell(tsd_cell_t* T, w_cell_t* W, int off, double t, double w)
{
  st_tsd(T + off, t); st_w(W + off, w);
}

struct GridDev {
  int N;            // cells per side
  int PX;           // tiles per side
  int tiles;
  double cs;        // _cellSize
  double inv_cs;    // _invCellSize
  double max_trunc; // _maxTruncation
  double min_x, max_x, min_y, max_y;
  uint8_t* flags;
  double*  init_weight;
  tsd_cell_t* tsd;
  w_cell_t*   weight;
  unsigned long long* negmask;   // [tiles] bit gy*8+gx: the 4x4-cell group (gx, gy) of the tile has (ever) held a negative tsd
};

// `negmask` bookkeeping (k_raycast skips the steps that cannot see a sign change): which 4x4-cell groups of a tile
// have ever held a negative value.  Sticky and conservative: never cleared by a push.  A halo cell is a copy of
// the owning neighbour's interior cell, so the owner's bit covers it.
__device__ __forceinline__ unsigned long long neg_bit(unsigned ix, unsigned iy)     // interior cell (ix, iy), 0..31
{
  return 1ull << ((iy >> 2) * 8u + (ix >> 2));
}
// groups gx0..gx1 x gy0..gy1 (0..7) of a tile as a mask
__device__ __forceinline__ unsigned long long neg_rect(int gx0, int gx1, int gy0, int gy1)
{
  const unsigned long long cols = (unsigned long long)(((1u << (gx1 + 1)) - (1u << gx0)) & 0xFFu) * 0x0101010101010101ull;
  const unsigned long long rows = (~0ull >> (8 * (7 - gy1))) & (~0ull << (8 * gy0));
  return cols & rows;
}

enum : int { INTERP_SUCCESS = 0, INTERP_INVALIDINDEX = 1, INTERP_EMPTYPARTITION = 2, INTERP_ISNAN = 3 };

// SensorPolar2D::backProject, batched form (SensorPolar2D.cpp:117-135): PoseInv * (x,y,1)^T through
// dgemm(NoTrans,Trans) = ((0 + a*x) + b*y) + c*1, atan2, bound checks, C round().
// Pi = first two rows of the inverse pose.
__device__ __forceinline__ int backproject(const double* __restrict__ Pi, double x, double y,
                                           double phi_min, double ang_res_inv, double phi_lower,
                                           double phi_upper)
{
  double lx = 0.0, ly = 0.0;
  lx += Pi[0] * x; lx += Pi[1] * y; lx += Pi[2] * 1.0;
  ly += Pi[3] * x; ly += Pi[4] * y; ly += Pi[5] * 1.0;
  const double phi = atan2(ly, lx);
  if (phi <= phi_lower) return -2;
  if (phi >= phi_upper) return -1;
  return (int)round((phi - phi_min) * ang_res_inv);
}

// TsdGrid::coord2Cell (TsdGrid.h:306-340)
__device__ __forceinline__ bool coord2cell(const GridDev& g, double x, double y, int& p, int& lx,
                                           int& ly, double& dx, double& dy)
{
  const double dcx = x * g.inv_cs, dcy = y * g.inv_cs;
  int xi = (int)floor(dcx), yi = (int)floor(dcy);
  dx = ((double)xi + 0.5) * g.cs;
  dy = ((double)yi + 0.5) * g.cs;
  if (x < dx) { xi--; dx -= g.cs; }
  if (y < dy) { yi--; dy -= g.cs; }
  if (xi >= g.N || xi < 0 || yi >= g.N || yi < 0) return false;
  p  = (yi >> 5) * g.PX + (xi >> 5);
  lx = xi & 31;
  ly = yi & 31;
  return true;
}

// TsdGrid::interpolateBilinear (TsdGrid.h:284-304) + TsdGridPartition::interpolateBilinear
// (TsdGridPartition.h:214-221).  Reads the halo at lx/ly == 31.
__device__ __forceinline__ int interpolate_bilinear(const GridDev& g, double x, double y, double& tsd)
{
  int p, lx, ly; double dx, dy;
  if (!coord2cell(g, x, y, p, lx, ly, dx, dy)) return INTERP_INVALIDINDEX;
  if (!g.flags[p]) return INTERP_EMPTYPARTITION;
  const double wx = fabs((x - dx) * g.inv_cs);
  const double wy = fabs((y - dy) * g.inv_cs);
  const tsd_cell_t* t = g.tsd + (size_t)p * TILE_STRIDE;
  const double t00 = ld_tsd(t + cell_off(lx, ly)), t01 = ld_tsd(t + cell_off(lx + 1, ly));
  const double t10 = ld_tsd(t + cell_off(lx, ly + 1)), t11 = ld_tsd(t + cell_off(lx + 1, ly + 1));
  tsd = t00 * (1. - wy) * (1. - wx)
      + t10 * wy * (1. - wx)
      + t01 * (1. - wy) * wx
      + t11 * wy * wx;
  if (isnan(tsd)) return INTERP_ISNAN;
  return INTERP_SUCCESS;
}

// A read the optimiser must leave where it is written: LLVM's sink pass moves a plain read of __restrict__ const memory into the
// conditional block of its only use -- where it is then issued late and waited for alone (a memory round trip of its own in kernels
// that are chains of round trips).  A relaxed single-thread-scope atomic read is an ordinary global_load in the ISA, and stays put.
template <typename T>
__device__ __forceinline__ T ld_pinned(const T* p)
{
#ifdef TSD_NO_PINNED      // diagnostic build (tools/tu_ab.sh): plain reads, for the A/B
  return *p;
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SINGLETHREAD);
#endif
}
__device__ __forceinline__ double ld_tsd_pinned(const tsd_cell_t* p)
{
#ifdef TSD_STORAGE_Q32
  const int32_t q = ld_pinned(p); return q == Q_NAN ? __builtin_nan("") : ldexp((double)q, -30);
#else
  return ld_pinned(p);
#endif
}
// the four cells a bilinear look-up at anchor (lx, ly) touches, reads issued together (the halo strip when lx / ly == 31)
struct Quad { double t00, t01, t10, t11; };
__device__ __forceinline__ Quad load_quad(const tsd_cell_t* __restrict__ tile, int lx, int ly)
{
  // (pinned: the callers issue these next to the tile's flag read; as plain reads they are sunk behind the test of the flag)
  Quad q;
  q.t00 = ld_tsd_pinned(tile + cell_off(lx, ly));     q.t01 = ld_tsd_pinned(tile + cell_off(lx + 1, ly));
  q.t10 = ld_tsd_pinned(tile + cell_off(lx, ly + 1)); q.t11 = ld_tsd_pinned(tile + cell_off(lx + 1, ly + 1));
  return q;
}

// 64-lane sum (all lanes receive lane 0's total is NOT guaranteed: result valid in lane 0)
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

}  // namespace tsd
