// tsd_device.hpp -- device-side view of the TSD grid and the small fp64 helpers every kernel shares.
//
// Data layout in HBM (DESIGN.md "Data layout"): tile-major structure-of-arrays.  Tile p = py*PX + px
// (row-major like TsdGrid::_partitions[0][p], TsdGrid.cpp:234) owns TILE_STRIDE doubles in `tsd` and
// in `weight`; the first 33*33 of them are the row-major 33x33 cells (32x32 interior + the duplicated
// 1-cell halo of TsdGridPartition.cpp:97), the rest is padding to a 64-byte line.  `flags[p]` is
// TsdGridPartition::_initialized, `init_weight[p]` is _initWeight.  All arithmetic is fp64 in the
// reference's operation order; translation units are compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsd {

constexpr int TILE_DIM    = 32;
constexpr int TILE_PITCH  = 33;
constexpr int TILE_CELLS  = 33 * 33;
constexpr int TILE_STRIDE = 1096;          // 1089 rounded up to a multiple of 8 doubles (64 B)
constexpr double MAX_WEIGHT = 32.0;        // TSDGRIDMAXWEIGHT, reconstruct_defs.h:4

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
  double*  tsd;
  double*  weight;
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
  const double* t = g.tsd + (size_t)p * TILE_STRIDE + ly * TILE_PITCH + lx;
  const double t00 = t[0], t01 = t[1], t10 = t[TILE_PITCH], t11 = t[TILE_PITCH + 1];
  tsd = t00 * (1. - wy) * (1. - wx)
      + t10 * wy * (1. - wx)
      + t01 * (1. - wy) * wx
      + t11 * wy * wx;
  if (isnan(tsd)) return INTERP_ISNAN;
  return INTERP_SUCCESS;
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
