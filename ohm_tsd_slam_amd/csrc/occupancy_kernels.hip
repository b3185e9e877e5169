// occupancy_kernels.hip -- the occupancy map of ThreadGrid::eventLoop (ThreadGrid.cpp:72-118):
// RayCastAxisAligned2D::calcCoords (RayCastAxisAligned2D.cpp:13-105) writes -1 / 0 per cell of every
// non-border tile into a PERSISTENT char map (_occGridContent, initialised to -1, ThreadGrid.cpp:27)
// and collects the sign changes along rows and columns; ThreadGrid then copies the map and marks the
// rounded sign-change cells with 100 (+ optional inflation).
//
// The reference walks tiles serially (y outer, x inner) and each initialised tile also writes its
// 1-cell halo, i.e. into the first row/column of its up/right neighbours, so for a cell at a tile's
// first row/column the LAST writer in that serial order wins: own tile > left > down > diagonal.
// k_occ_cells resolves that priority per cell (a gather, no write races); k_occ_mark does the
// idempotent "100" marks.  Streams the whole grid once: HBM-bound (cells*8 B read, cells*2 B written).
#include "tsd_ctx.hpp"

namespace tsd {

__device__ __forceinline__ bool tile_processed(int X, int Y, int PX)
{
  return X >= 1 && X <= PX - 2 && Y >= 1 && Y <= PX - 2;   // loops 1 .. partitions-2 (:25-27)
}

// value an initialised tile writes for its local cell (ly,lx), lx/ly in 0..32 (:41-47)
__device__ __forceinline__ int8_t occ_from_tsd(const GridDev& g, int p, int ly, int lx)
{
  const double t = ld_tsd(g.tsd + (size_t)p * TILE_STRIDE + cell_off(lx, ly));
  return (t > 0.0) ? 0 : -1;
}

// One workgroup per tile, one thread per 4 consecutive cells of a row: the map is read and written 4 bytes per
// lane (16-byte aligned rows), the tile's cells 32 bytes per lane.  A tile nobody writes (most of the grid) only
// forwards the persistent map to the output.
// work list of k_occ_mark: the tiles that hold cells (processed and initialised), in OCC_SHARDS segments of the list with a counter
// each on its own 128-byte line (tile p goes to shard p % OCC_SHARDS, which has room for exactly tiles / OCC_SHARDS entries) -- one
// counter for all tiles would hand out ~88 slots per microsecond (MI355X_MICROARCH.md "dequeue"), 45 us for a cfg 2 map
constexpr int OCC_SHARDS = 32, OCC_HEAD_STRIDE = 32;
__global__ void __launch_bounds__(256)
k_occ_cells(GridDev g, int8_t* __restrict__ content, int8_t* __restrict__ out, unsigned int* __restrict__ heads, uint32_t* __restrict__ list,
            unsigned int* __restrict__ heads_next, int* __restrict__ count)
{
  // (this extraction's mark counter and the NEXT extraction's list heads are cleared from here -- the heads come in two sets used in
  // turn -- instead of by two memset launches ahead of every extraction)
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) *count = 0;
    if (threadIdx.x < OCC_SHARDS) heads_next[threadIdx.x * OCC_HEAD_STRIDE] = 0u;
  }
  const int p = blockIdx.x;
  const int PX = g.PX;
  const int X = p % PX, Y = p / PX;
  const bool own_proc = tile_processed(X, Y, PX);
  const bool own_init = g.flags[p] != 0;
  const bool own_empty = !own_init && g.init_weight[p] > 0.0;   // isEmpty(), TsdGridPartition.h:72
  const bool left_w = X >= 1 && tile_processed(X - 1, Y, PX) && g.flags[p - 1];
  const bool down_w = Y >= 1 && tile_processed(X, Y - 1, PX) && g.flags[p - PX];
  const bool diag_w = X >= 1 && Y >= 1 && tile_processed(X - 1, Y - 1, PX) && g.flags[p - PX - 1];
  const int lx0 = (threadIdx.x & 7) * 4, ly = threadIdx.x >> 3;
  const size_t gi = (size_t)(Y * TILE_DIM + ly) * g.N + (size_t)(X * TILE_DIM + lx0);
  uint32_t* c4 = reinterpret_cast<uint32_t*>(content + gi);
  uint32_t* o4 = reinterpret_cast<uint32_t*>(out + gi);
  if (own_proc && own_init) {
    if (threadIdx.x == 0) {
      const unsigned sh = (unsigned)p % OCC_SHARDS, cap = ((unsigned)g.tiles + OCC_SHARDS - 1) / OCC_SHARDS;
      list[sh * cap + atomicAdd(&heads[sh * OCC_HEAD_STRIDE], 1u)] = (uint32_t)p;
    }
    const tsd_cell_t* t = g.tsd + (size_t)p * TILE_STRIDE + ly * TILE_DIM + lx0;     // interior row, 4 cells
    const double t0 = ld_tsd(t), t1 = ld_tsd(t + 1), t2 = ld_tsd(t + 2), t3 = ld_tsd(t + 3);
    const uint32_t v = (t0 > 0.0 ? 0u : 0xFFu) | (t1 > 0.0 ? 0u : 0xFF00u) | (t2 > 0.0 ? 0u : 0xFF0000u) | (t3 > 0.0 ? 0u : 0xFF000000u);
    *c4 = v; *o4 = v;
    return;
  }
  if (own_proc && own_empty) { *c4 = 0u; *o4 = 0u; return; }
  uint32_t v = *c4;
  // a neighbour's halo lands in this tile's first column / row / corner cell (the last writer of the reference's
  // serial order wins: left > down > diagonal)
  bool changed = false;
  if (lx0 == 0 && left_w) { v = (v & ~0xFFu) | (uint8_t)occ_from_tsd(g, p - 1, ly, TILE_DIM); changed = true; }
  else if (lx0 == 0 && ly == 0 && down_w) { }       // (handled with the rest of row 0 below)
  if (ly == 0 && down_w) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (lx0 + k == 0 && left_w) continue;                                        // left neighbour wins the corner cell
      v = (v & ~(0xFFu << (8 * k))) | ((uint32_t)(uint8_t)occ_from_tsd(g, p - PX, TILE_DIM, lx0 + k) << (8 * k));
    }
    changed = true;
  }
  if (lx0 == 0 && ly == 0 && !left_w && !down_w && diag_w) { v = (v & ~0xFFu) | (uint8_t)occ_from_tsd(g, p - PX - 1, TILE_DIM, TILE_DIM); changed = true; }
  if (changed) *c4 = v;
  *o4 = v;
}

__device__ __forceinline__ void occ_mark(const GridDev& g, int8_t* out, double x, double y, int inflate,
                                         int factor)
{
  // ThreadGrid.cpp:96-117
  const double ru = round(x / g.cs), rv = round(y / g.cs);
  if (!(ru > 0.0 && ru < (double)g.N && rv > 0.0 && rv < (double)g.N)) return;
  const unsigned u = (unsigned)ru, v = (unsigned)rv, N = (unsigned)g.N;
  out[(size_t)v * N + u] = 100;
  if (inflate) {
    for (unsigned i = v - (unsigned)factor; i < v + (unsigned)factor; i++)
      for (unsigned j = u - (unsigned)factor; j < u + (unsigned)factor; j++)
        if ((size_t)i * N + j < (size_t)N * N) out[(size_t)i * N + j] = 100;
  }
}

// One 256-thread workgroup per LISTED tile (k_occ_cells' work list), a fixed grid looping over the list's shards.  (Round 2: one
// workgroup per tile of the grid -- 16 384 launches at cfg 2, of which three quarters found their tile uninitialised and left.)
__global__ void __launch_bounds__(256)
k_occ_mark(GridDev g, int8_t* __restrict__ out, int* __restrict__ count, int inflate, int factor,
           const unsigned int* __restrict__ heads, const uint32_t* __restrict__ list)
{
  const int PX = g.PX;
  const unsigned sh = blockIdx.x % OCC_SHARDS, cap = ((unsigned)g.tiles + OCC_SHARDS - 1) / OCC_SHARDS;
  const unsigned n_sh = heads[sh * OCC_HEAD_STRIDE];
  // the tile (33 x 33 doubles) through LDS: every cell is looked at by up to four scan positions
  __shared__ double T[TILE_CELLS];
  // The scan positions that hold a sign change (a few dozen of a surface tile's 2 112) are LISTED, and the interpolation, the two
  // divisions and the rounding of a mark run over the list, one lane per change: inside the scan loop the ~250 instructions of a mark
  // were executed by the whole wave for every loop round in which ANY lane had a change -- 12 of this kernel's 21 us at cfg 2
  // (seven variants of the stores themselves had changed nothing).
  __shared__ unsigned short s_ev[2 * TILE_PITCH * TILE_DIM];
  __shared__ int s_nev;
  if (threadIdx.x == 0) s_nev = 0;
  const double cs = g.cs;
  int n = 0;
  // A tile's 1 089 cells are five reads per thread, requested TOGETHER (as a plain loop the compiler had read -> wait -> LDS write five
  // times in a row), and the NEXT tile's five are requested before this tile's scans, whose time covers their round trip.
  constexpr int OCC_RD = (TILE_CELLS + 255) / 256;
  const unsigned k_first = blockIdx.x / OCC_SHARDS, k_step = gridDim.x / OCC_SHARDS;
  auto request = [&](int p, double (&v)[OCC_RD]) {
    const tsd_cell_t* Tg = g.tsd + (size_t)p * TILE_STRIDE;
#pragma unroll
    for (int j = 0; j < OCC_RD; j++) { const int i = (int)threadIdx.x + 256 * j; v[j] = ld_tsd_pinned(Tg + (i < TILE_CELLS ? i : 0)); }
  };
  double cur[OCC_RD];
  int p = 0;
  if (k_first < n_sh) { p = (int)list[sh * cap + k_first]; request(p, cur); }
  for (unsigned k = k_first; k < n_sh; k += k_step) {
    const int X = p % PX, Y = p / PX;
#pragma unroll
    for (int j = 0; j < OCC_RD; j++) { const int i = (int)threadIdx.x + 256 * j; if (i < TILE_CELLS) T[canonical_of_off(i)] = cur[j]; }   // LDS copy in the 33 x 33 form
    __syncthreads();
    const bool has_next = k + k_step < n_sh;
    int p_next = p;
    if (has_next) { p_next = (int)list[sh * cap + k + k_step]; request(p_next, cur); }
    // row scans: py in 0..32, px in 1..32 (:38-60); column scans: px in 0..32, py in 1..32 (:62-80)
    for (int c = threadIdx.x; c < 2 * TILE_PITCH * TILE_DIM; c += 256) {
      const bool col = c >= TILE_PITCH * TILE_DIM;
      const int cc = col ? c - TILE_PITCH * TILE_DIM : c;
      const int a = cc / TILE_DIM;          // fixed index 0..32
      const int b = cc % TILE_DIM + 1;      // running index 1..32
      const int py = col ? b : a, px = col ? a : b;
      const double prev = col ? T[(py - 1) * TILE_PITCH + px] : T[py * TILE_PITCH + px - 1];
      const double cur = T[py * TILE_PITCH + px];
      if ((prev > 0 && cur < 0) || (prev < 0 && cur > 0)) s_ev[atomicAdd(&s_nev, 1)] = (unsigned short)c;
    }
    __syncthreads();
    const int nev = s_nev;
    for (int e = threadIdx.x; e < nev; e += 256) {
      const int c = (int)s_ev[e];
      const bool col = c >= TILE_PITCH * TILE_DIM;
      const int cc = col ? c - TILE_PITCH * TILE_DIM : c;
      const int a = cc / TILE_DIM, b = cc % TILE_DIM + 1;
      const int py = col ? b : a, px = col ? a : b;
      const double prev = col ? T[(py - 1) * TILE_PITCH + px] : T[py * TILE_PITCH + px - 1];
      const double cur = T[py * TILE_PITCH + px];
      const double interp = prev / (prev - cur);
      double x, y;
      if (!col) {
        x = px * cs + cs * (interp - 1.0) + (X * TILE_DIM) * cs;
        y = py * cs + (Y * TILE_DIM) * cs;
      } else {
        x = px * cs + (X * TILE_DIM) * cs;
        y = py * cs + cs * (interp - 1.0) + (Y * TILE_DIM) * cs;
      }
      occ_mark(g, out, x, y, inflate, factor);
      n++;
    }
    __syncthreads();            // (the LDS copy and the list are rewritten by the next tile)
    if (threadIdx.x == 0) s_nev = 0;
    p = p_next;
  }
  n = wave_sum_i(n);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(count, n);
}

// TsdGrid::grid2ColorImage (TsdGrid.cpp:429-488), what ThreadGrid publishes next to the occupancy map
// (ThreadGrid.cpp:125).  The reference walks the image with px += stepW / py += stepH (repeated fp64 addition):
// the two coordinate tables are built that way on the host and a pixel is then one coord2Cell + one cell read.
// One thread per pixel, 3 bytes out; streams the touched cells once.
__global__ void __launch_bounds__(256)
k_color_image(GridDev g, const double* __restrict__ pxs, const double* __restrict__ pys, unsigned width,
              unsigned height, uint8_t* __restrict__ image)
{
  const unsigned w = blockIdx.x * 256u + threadIdx.x, h = blockIdx.y;
  if (w >= width || h >= height) return;
  int p, lx, ly; double dx, dy;
  double t = __builtin_nan("");
  bool is_empty = false;
  if (coord2cell(g, pxs[w], pys[h], p, lx, ly, dx, dy)) {
    const bool init = g.flags[p] != 0;
    if (init) t = ld_tsd(g.tsd + (size_t)p * TILE_STRIDE + cell_off(lx, ly));
    is_empty = !init && g.init_weight[p] > 0.0;                 // isEmpty(), TsdGridPartition.h:72
  }
  uint8_t r, gch, b;
  if (t > 0.0) { r = (uint8_t)(t * 255.0); gch = 255; b = (uint8_t)(t * 255.0); }
  else if (t < 0.0) { r = (uint8_t)((1.0 + t) * 255.0); gch = 0; b = 0; }
  else if (is_empty) { r = 255; gch = 255; b = 255; }
  else { r = 0; gch = 0; b = 0; }
  uint8_t* o = image + 3 * ((size_t)h * width + w);
  o[0] = r; o[1] = gch; o[2] = b;
}

int launch_color_image(tsd_ctx* ctx, const double* d_px, const double* d_py, unsigned width, unsigned height, uint8_t* d_image)
{
  hipLaunchKernelGGL(k_color_image, dim3((width + 255) / 256, height), dim3(256), 0, ctx->stream, ctx->grid, d_px, d_py,
                     width, height, d_image);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

size_t occ_heads_bytes() { return 2 * OCC_SHARDS * OCC_HEAD_STRIDE * sizeof(unsigned int); }      // two sets, used in turn (zeroed at creation)

int launch_occupancy(tsd_ctx* ctx, int8_t* d_out, int inflate, int inflate_factor)
{
  ScopedKernelTimer t(ctx, "occupancy", true);
  unsigned int* heads = ctx->d_occ_heads + (size_t)ctx->occ_parity * OCC_SHARDS * OCC_HEAD_STRIDE;
  unsigned int* heads_next = ctx->d_occ_heads + (size_t)(ctx->occ_parity ^ 1) * OCC_SHARDS * OCC_HEAD_STRIDE;
  ctx->occ_parity ^= 1;
  hipLaunchKernelGGL(k_occ_cells, dim3(ctx->grid.tiles), dim3(256), 0, ctx->stream, ctx->grid,
                     ctx->d_occ, d_out, heads, ctx->d_occ_list, heads_next, ctx->d_occ_count);
  // (measured at cfg 2, maps of 40 / 200 scans: 2 048 workgroups 9.3 / 12.3 us, 1 024: 10.5 / 12.6, 512: 14.1 / 17.7, 256: 21.9 / 28.2 --
  //  a tile is a ~5 us chain of dependent round trips, so as many of them side by side as there are)
  constexpr int OCC_MARK_GROUPS = 2048;
  const int mark_groups = ctx->grid.tiles < OCC_MARK_GROUPS ? ((ctx->grid.tiles + OCC_SHARDS - 1) / OCC_SHARDS) * OCC_SHARDS : OCC_MARK_GROUPS;   // a multiple of the shards
  hipLaunchKernelGGL(k_occ_mark, dim3(mark_groups), dim3(256), 0, ctx->stream, ctx->grid, d_out,
                     ctx->d_occ_count, inflate, inflate_factor, heads, ctx->d_occ_list);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

}  // namespace tsd
