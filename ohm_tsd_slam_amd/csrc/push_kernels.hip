// push_kernels.hip -- TsdGrid::push (TsdGrid.cpp:217-284) as three gfx950 kernels on one stream:
//
//   k_push_classify  one lane per tile: TsdGridComponent::isInRange (TsdGridComponent.cpp:43-124).
//                    The two beam-range scans (any beam sees into the tile / every beam sees past
//                    it) are done wave-cooperatively: the wave walks the surviving lanes' [lo,hi]
//                    beam ranges 64 beams at a time with ballots, so one tile that spans the whole
//                    scan costs 17 coalesced reads instead of a 1081-step serial loop in one lane.
//                    Emits the work list (UPDATE tiles, EMPTIED initialised tiles) with a
//                    wave-aggregated atomic and bumps _initWeight of empty uninitialised tiles.
//   k_push_update    persistent 256-thread blocks walk the list; the scan (ranges + mask) is staged
//                    in LDS once per block; 4 cells per thread, row-major => coalesced 8-byte RMW.
//                    Lazy TsdGridPartition::init (TsdGridPartition.cpp:88-134) is folded in: a fresh
//                    tile's old value is known (NaN/0 or 1/_initWeight) so it is written once.
//                    EMPTIED entries run TsdGridPartition::increaseEmptiness over all 33x33 cells.
//   k_push_halo      TsdGrid::propagateBorders (TsdGrid.cpp:372-427) restricted to what can have
//                    changed: for every listed tile refresh its own halo from R/U/UR and the halos of
//                    L/D/DL that mirror its first column/row/cell.  Equal to the reference's full
//                    sweep by induction (untouched pairs are already consistent).
//
// HBM-bound integer/fp64 work: no MFMA.  Roofline accounting in DESIGN.md.
#include "tsd_ctx.hpp"

namespace tsd {

constexpr uint32_t LIST_EMPTIED = 0x80000000u;
constexpr int UPDATE_BLOCK = 256;

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

// RMQ = true: sparse-table range queries (beams <= RMQ_MAX_BEAMS, the tables fit in LDS);
// RMQ = false: wave-cooperative scan of every tile's beam range (any beam count up to TSD_MAX_BEAMS)
constexpr int RMQ_MAX_BEAMS = 2048;

template <bool RMQ>
__global__ void __launch_bounds__(256)
k_push_classify(GridDev g, PushArgs a_val, const PushArgs* __restrict__ a_dev,
                const double* __restrict__ ranges,
                const uint8_t* __restrict__ mask, PushCounters* __restrict__ ctr,
                PushCounters* __restrict__ ctr_next, uint32_t* __restrict__ list,
                int* __restrict__ block_stats)
{
  const PushArgs a = a_dev ? *a_dev : a_val;
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) {   // clear the other epoch's counter set (consumed before this push was enqueued)
    ctr_next->cells_updated = 0; ctr_next->cells_visited = 0; ctr_next->list_count = 0;
    ctr_next->tiles_range_pass = 0; ctr_next->tiles_update = 0; ctr_next->tiles_new = 0;
    ctr_next->tiles_new_from_empty = 0; ctr_next->tiles_emptied_init = 0;
    ctr_next->tiles_emptied_uninit = 0;
  }
  // a push gated off on the device (fused scan): no tile is classified, the work list stays empty
  const bool valid = p < g.tiles && a.enabled;

  bool range_pass = false, need_scan = false, all_vis = true;
  int lo = 0, hi = -1;
  double distance = 0.0, closest = 0.0, farthest = 0.0;
  if (valid) {
    double e[4][2], cx, cy, rad;
    tile_geometry(g, p, e, cx, cy, rad);
    // euklideanDistance<obfloat>(pos, _centroid, 2) (mathbase.h:369-378)
    double sqr = 0.0;
    { const double t0 = a.trx - cx; sqr += t0 * t0; const double t1 = a.try_ - cy; sqr += t1 * t1; }
    distance = sqrt(sqr);
    closest = distance - rad - g.max_trunc;
    farthest = distance + rad + g.max_trunc;
    range_pass = !(closest > a.max_range) && !(farthest < a.min_range);
    if (range_pass) {
      int idx[4];
      bool any_vis = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        idx[k] = backproject(a.Pi, e[k][0], e[k][1], a.phi_min, a.ang_res_inv, a.phi_lower, a.phi_upper);
        if (idx[k] == -1) { idx[k] = a.beams - 1; all_vis = false; }
        else if (idx[k] == -2) { idx[k] = 0; all_vis = false; }
        else any_vis = true;
      }
      // minmaxArray<int> (mathbase.h:55-64)
      lo = idx[0]; hi = idx[0];
#pragma unroll
      for (int k = 1; k < 4; k++) { if (lo > idx[k]) lo = idx[k]; else if (hi < idx[k]) hi = idx[k]; }
      need_scan = any_vis;
    }
  }

  // The two beam-range tests of isInRange over [lo, hi] (TsdGridComponent.cpp:86-119),
  //   visible := any j: data[j] > closest && mask[j]
  //   empty   := all j: isinf(data[j]) ? distance < lowReflectivityRange : (data[j] > farthest && mask[j])
  // are range-maximum / range-minimum queries:  visible <=> max A > closest with A[j] = mask ? data : -inf,
  // and (over the finite beams) empty <=> min B > farthest with B[j] = isinf ? +inf : (mask ? data : -inf),
  // plus "is there an infinite beam in the range".  Each block that has a tile to test builds two sparse
  // tables of ARG-max / ARG-min indices in LDS (11 levels x beams x 2 B each, the values stay fp64), after
  // which a tile costs four LDS look-ups however many beams it spans (a tile near the sensor, or one
  // that straddles the +-pi cut of a 360 degree scanner, spans hundreds).
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int B = a.beams;
  const int Bp = (B + 1) & ~1;
  double* s_A = reinterpret_cast<double*>(smem);
  double* s_B = s_A + Bp;
  unsigned short* s_inf = reinterpret_cast<unsigned short*>(s_B + Bp);        // [B + 1] prefix count of infinite beams
  unsigned short* s_tmax = s_inf + ((B + 2 + 7) & ~7);                       // [levels][B]
  int levels = 1;
  while ((1 << levels) <= B) levels++;                                       // 2^(levels-1) <= B
  unsigned short* s_tmin = s_tmax + (size_t)levels * Bp;

  bool visible = false, empty = false;
  if (!RMQ) {
    double* s_ranges = reinterpret_cast<double*>(smem);
    uint8_t* s_mask = reinterpret_cast<uint8_t*>(smem + (size_t)Bp * sizeof(double));
    if (__syncthreads_or(need_scan ? 1 : 0)) {
      for (int i = threadIdx.x; i < B; i += blockDim.x) { s_ranges[i] = ranges[i]; s_mask[i] = mask[i]; }
      __syncthreads();
    }
    // the wave walks the surviving lanes' [lo, hi] beam ranges 64 beams at a time with ballots
    unsigned long long todo = __ballot(need_scan);
    while (todo) {
      const int s = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int lo_s = __shfl(lo, s, 64), hi_s = __shfl(hi, s, 64);
      const double closest_s = __shfl(closest, s, 64), farthest_s = __shfl(farthest, s, 64);
      const double distance_s = __shfl(distance, s, 64);
      bool vis = false, fail = false;
      for (int j = lo_s + lane; j <= hi_s; j += 64) {
        const double d = s_ranges[j];
        const bool mk = s_mask[j] != 0;
        vis = vis || ((d > closest_s) && mk);
        if (isinf(d)) fail = fail || !(distance_s < a.low_refl);
        else fail = fail || !((d > farthest_s) && mk);
      }
      const bool any_vis_beam = __any(vis);
      const bool any_fail = __any(fail);
      if (lane == s) { visible = any_vis_beam; empty = !any_fail; }
    }
  } else if (__syncthreads_or(need_scan ? 1 : 0)) {
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
      const double d = ranges[i];
      const bool mk = mask[i] != 0;
      s_A[i] = mk ? d : -__builtin_inf();
      s_B[i] = isinf(d) ? __builtin_inf() : (mk ? d : -__builtin_inf());
      s_tmax[i] = (unsigned short)i; s_tmin[i] = (unsigned short)i;
    }
    if (threadIdx.x < 64) {
      // prefix count of infinite readings by one wave (B <= 4096: 64 lanes x 64 beams)
      const int per = (B + 63) / 64;
      const int j0 = threadIdx.x * per;
      int c = 0;
      for (int j = j0; j < j0 + per && j < B; j++) c += isinf(ranges[j]) ? 1 : 0;
      int incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if ((int)threadIdx.x >= off) incl += t; }
      int run = incl - c;
      if (threadIdx.x == 0) s_inf[0] = 0;
      for (int j = j0; j < j0 + per && j < B; j++) { run += isinf(ranges[j]) ? 1 : 0; s_inf[j + 1] = (unsigned short)run; }
    }
    __syncthreads();
    for (int k = 1; k < levels; k++) {
      const int half = 1 << (k - 1), span = 1 << k;
      const unsigned short* pmax = s_tmax + (size_t)(k - 1) * Bp; unsigned short* cmax = s_tmax + (size_t)k * Bp;
      const unsigned short* pmin = s_tmin + (size_t)(k - 1) * Bp; unsigned short* cmin = s_tmin + (size_t)k * Bp;
      for (int j = threadIdx.x; j + span <= B; j += blockDim.x) {
        const unsigned short a0 = pmax[j], a1 = pmax[j + half];
        cmax[j] = s_A[a1] > s_A[a0] ? a1 : a0;
        const unsigned short b0 = pmin[j], b1 = pmin[j + half];
        cmin[j] = s_B[b1] < s_B[b0] ? b1 : b0;
      }
      __syncthreads();
    }
    if (need_scan) {
      const int len = hi - lo + 1;
      const int k = 31 - __clz(len);                                         // floor(log2(len))
      const unsigned short* tm = s_tmax + (size_t)k * Bp;
      const unsigned short* tn = s_tmin + (size_t)k * Bp;
      const int j2 = hi - (1 << k) + 1;
      const double amax = fmax(s_A[tm[lo]], s_A[tm[j2]]);
      const double bmin = fmin(s_B[tn[lo]], s_B[tn[j2]]);
      const bool has_inf = s_inf[hi + 1] != s_inf[lo];
      visible = amax > closest;
      empty = (bmin > farthest) && (!has_inf || distance < a.low_refl);
    }
  }

  // actions
  bool do_update = false, do_empty_init = false, do_empty_uninit = false, is_new = false, new_from_empty = false;
  if (need_scan && visible) {
    if (all_vis && empty) {
      if (g.flags[p]) do_empty_init = true;
      else {
        // TsdGridPartition::increaseEmptiness, uninitialised branch (TsdGridPartition.cpp:159-163)
        double iw = g.init_weight[p];
        iw += 1.0;
        iw = fmin(iw, MAX_WEIGHT);
        g.init_weight[p] = iw;
        do_empty_uninit = true;
      }
    } else {
      do_update = true;
      if (!g.flags[p]) { is_new = true; new_from_empty = g.init_weight[p] > 0.0; }
    }
  }

  // list append + counters.  Same-address global atomics serialise at ~12 ns each (MI355X_MICROARCH
  // "fanin"), so the four waves of the block are combined in LDS first: ONE global atomic per block
  // (the list base) and plain stores of the per-block statistics, reduced later by k_push_halo.
  __shared__ int s_cnt[4][8];
  __shared__ int s_base;
  const int wave = threadIdx.x >> 6;
  const bool listed = do_update || do_empty_init;
  const unsigned long long lm = __ballot(listed);
  const int c1 = __popcll(__ballot(range_pass)), c2 = __popcll(__ballot(do_update)),
            c3 = __popcll(__ballot(is_new)), c4 = __popcll(__ballot(new_from_empty)),
            c5 = __popcll(__ballot(do_empty_init)), c6 = __popcll(__ballot(do_empty_uninit));
  if (lane == 0) {
    s_cnt[wave][0] = __popcll(lm);
    s_cnt[wave][1] = c1; s_cnt[wave][2] = c2; s_cnt[wave][3] = c3;
    s_cnt[wave][4] = c4; s_cnt[wave][5] = c5; s_cnt[wave][6] = c6;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot[7];
#pragma unroll
    for (int k = 0; k < 7; k++) tot[k] = s_cnt[0][k] + s_cnt[1][k] + s_cnt[2][k] + s_cnt[3][k];
    s_base = tot[0] ? atomicAdd(&ctr->list_count, tot[0]) : 0;
    int* bs = block_stats + (size_t)blockIdx.x * 8;
#pragma unroll
    for (int k = 1; k < 7; k++) bs[k] = tot[k];
  }
  __syncthreads();
  if (listed) {
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_cnt[w][0];
    off += __popcll(lm & ((1ull << lane) - 1ull));
    list[off] = (uint32_t)p | (do_empty_init ? LIST_EMPTIED : 0u);
  }
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
      tsd = (tsd * weight + v * w) / (weight + w);
      weight = fmin(weight + w, MAX_WEIGHT);
    }
    return true;
  }
  return false;
}

__global__ void __launch_bounds__(UPDATE_BLOCK)
k_push_update(GridDev g, PushArgs a_val, const PushArgs* __restrict__ a_dev,
              const double* __restrict__ ranges,
              const uint8_t* __restrict__ mask, PushCounters* __restrict__ ctr,
              const uint32_t* __restrict__ list, uint32_t* __restrict__ entry_upd)
{
  const PushArgs a = a_dev ? *a_dev : a_val;
  // all LDS in the dynamic region (16-byte aligned carve): [0,16) block counter, ranges, mask
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long& s_upd = *reinterpret_cast<unsigned long long*>(smem);
  double* s_ranges = reinterpret_cast<double*>(smem + 16);
  uint8_t* s_mask = reinterpret_cast<uint8_t*>(smem + 16 + (size_t)((a.beams + 1) & ~1) * sizeof(double));

  const int count = ctr->list_count;
  if ((int)blockIdx.x >= count) return;

  const int tid = threadIdx.x;
  for (int i = tid; i < a.beams; i += UPDATE_BLOCK) { s_ranges[i] = ranges[i]; s_mask[i] = mask[i]; }
  if (tid == 0) s_upd = 0ull;
  __syncthreads();

  const double max_trunc = g.max_trunc;
  const double inv_max_trunc = 1.0 / max_trunc;
  const double eps = -g.cs / 2.0;
  for (int li = blockIdx.x; li < count; li += gridDim.x) {
    unsigned int n_upd = 0;
    if (tid == 0) s_upd = 0ull;
    __syncthreads();
    const uint32_t entry = list[li];
    const int p = (int)(entry & ~LIST_EMPTIED);
    double* __restrict__ T = g.tsd + (size_t)p * TILE_STRIDE;
    double* __restrict__ W = g.weight + (size_t)p * TILE_STRIDE;

    if (entry & LIST_EMPTIED) {
      // TsdGridPartition::increaseEmptiness, initialised branch (TsdGridPartition.cpp:138-157):
      // all 33x33 cells, halo included; the average uses the NEW weight.
      for (int i = tid; i < TILE_CELLS; i += UPDATE_BLOCK) {
        double t = T[i], w = W[i];
        if (isnan(t)) {
          w += 1.0;
          t = 1.0;
        } else {
          w = fmin(w + 1, MAX_WEIGHT);
          t = (t * (w - 1.0) + 1.0) / w;
        }
        T[i] = t; W[i] = w;
      }
      if (tid == 0) entry_upd[li] = 0u;
      __syncthreads();
      continue;
    }

    const bool fresh = g.flags[p] == 0;
    const double iw = g.init_weight[p];
    // TsdGridPartition::init values (TsdGridPartition.cpp:98-120)
    const double t_init = (iw > 0.0) ? 1.0 : __builtin_nan("");
    const double w_init = iw;

    // partition weight (TsdGrid.cpp:239-243)
    double e[4][2], cx, cy, rad;
    tile_geometry(g, p, e, cx, cy, rad);
    double dist_c = sqrt((cx - a.trx) * (cx - a.trx) + (cy - a.try_) * (cy - a.try_));
    if (dist_c > a.max_range) dist_c = a.max_range;
    double pw = (a.max_range - dist_c) / a.max_range;
    pw *= pw;

    const unsigned x0 = (unsigned)(p % g.PX) * TILE_DIM, y0 = (unsigned)(p / g.PX) * TILE_DIM;
#pragma unroll
    for (int k = 0; k < (TILE_DIM * TILE_DIM) / UPDATE_BLOCK; k++) {
      const int c = tid + UPDATE_BLOCK * k;
      const unsigned ix = (unsigned)c & 31u, iy = (unsigned)c >> 5;
      const double ccx = ((double)(x0 + ix) + 0.5) * g.cs;   // TsdGridPartition.cpp:127-128
      const double ccy = ((double)(y0 + iy) + 0.5) * g.cs;
      const int ci = (int)(iy * TILE_PITCH + ix);
      const int index = backproject(a.Pi, ccx, ccy, a.phi_min, a.ang_res_inv, a.phi_lower, a.phi_upper);
      bool touched = false;
      double t = t_init, w = w_init;
      if (index >= 0 && s_mask[index]) {
        const double r = s_ranges[index];
        const double dist = sqrt((ccx - a.trx) * (ccx - a.trx) + (ccy - a.try_) * (ccy - a.try_));
        // the cell is only read when addTsd will update it (sd >= -maxTruncation): cells behind the
        // surface cost no HBM traffic
        double sd = 0.0; bool cand = false;
        if (!isinf(r)) { sd = r - dist; cand = true; }
        else if (dist < a.low_refl) { sd = max_trunc; cand = true; }
        if (cand && sd >= -max_trunc) {
          if (!fresh) { t = T[ci]; w = W[ci]; }
          touched = add_tsd(t, w, sd, pw, max_trunc, inv_max_trunc, eps);
        }
      }
      if (touched) n_upd++;
      if (touched || fresh) { T[ci] = t; W[ci] = w; }
    }
    if (fresh) {
      // halo cells of a freshly materialised tile keep the init value until k_push_halo
      if (tid < TILE_DIM) {
        T[tid * TILE_PITCH + TILE_DIM] = t_init; W[tid * TILE_PITCH + TILE_DIM] = w_init;   // column 32
      } else if (tid < 2 * TILE_DIM + 1) {
        const int i = tid - TILE_DIM;                                                          // row 32, 0..32
        T[TILE_DIM * TILE_PITCH + i] = t_init; W[TILE_DIM * TILE_PITCH + i] = w_init;
      }
    }
    // cells updated in this tile -> plain store (summed by k_push_halo; no same-address atomics)
    const unsigned wu = (unsigned)wave_sum_i((int)n_upd);
    if ((tid & 63) == 0 && wu) atomicAdd(&s_upd, (unsigned long long)wu);   // LDS
    __syncthreads();
    if (tid == 0) {
      entry_upd[li] = (uint32_t)s_upd;
      // publish the tile only after every wave of the block has read `fresh` (barrier above)
      if (fresh) g.flags[p] = 1;
    }
  }
}

// TsdGrid::propagateBorders (TsdGrid.cpp:372-427), incremental form.  One wave per listed tile.
__device__ __forceinline__ void copy_col(const GridDev& g, int dst, int src, int lane)
{
  if (lane < TILE_DIM) {
    const size_t d = (size_t)dst * TILE_STRIDE + lane * TILE_PITCH + TILE_DIM;
    const size_t s = (size_t)src * TILE_STRIDE + lane * TILE_PITCH;
    g.tsd[d] = g.tsd[s]; g.weight[d] = g.weight[s];
  }
}
__device__ __forceinline__ void copy_row(const GridDev& g, int dst, int src, int lane)
{
  if (lane >= TILE_DIM) {
    const int i = lane - TILE_DIM;
    const size_t d = (size_t)dst * TILE_STRIDE + TILE_DIM * TILE_PITCH + i;
    const size_t s = (size_t)src * TILE_STRIDE + i;
    g.tsd[d] = g.tsd[s]; g.weight[d] = g.weight[s];
  }
}
__device__ __forceinline__ void copy_corner(const GridDev& g, int dst, int src, int lane)
{
  if (lane == 0) {
    const size_t d = (size_t)dst * TILE_STRIDE + TILE_DIM * TILE_PITCH + TILE_DIM;
    const size_t s = (size_t)src * TILE_STRIDE;
    g.tsd[d] = g.tsd[s]; g.weight[d] = g.weight[s];
  }
}

__global__ void __launch_bounds__(256)
k_push_halo(GridDev g, const uint32_t* __restrict__ list, const int* __restrict__ count_ptr,
            PushCounters* __restrict__ ctr, const int* __restrict__ block_stats, int n_stat_blocks,
            const uint32_t* __restrict__ entry_upd, PushCounters* __restrict__ total,
            const PushArgs* __restrict__ a_dev)
{
  const int count = *count_ptr;
  if (ctr != nullptr && blockIdx.x == gridDim.x - 1) {
    // statistics of this push (the last block does it so that it overlaps the halo work of the others)
    __shared__ unsigned long long s_tot[8];
    if (threadIdx.x < 8) s_tot[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < count; i += blockDim.x) v[0] += entry_upd[i];
    for (int b = threadIdx.x; b < n_stat_blocks; b += blockDim.x)
#pragma unroll
      for (int k = 1; k < 7; k++) v[k] += (unsigned long long)block_stats[(size_t)b * 8 + k];
#pragma unroll
    for (int k = 0; k < 7; k++) {
      unsigned long long x = v[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if ((threadIdx.x & 63) == 0 && x) atomicAdd(&s_tot[k], x);   // LDS
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      ctr->cells_updated = s_tot[0];
      ctr->tiles_range_pass = (int)s_tot[1]; ctr->tiles_update = (int)s_tot[2]; ctr->tiles_new = (int)s_tot[3];
      ctr->tiles_new_from_empty = (int)s_tot[4]; ctr->tiles_emptied_init = (int)s_tot[5];
      ctr->tiles_emptied_uninit = (int)s_tot[6];
      ctr->cells_visited = 1024ull * s_tot[2];     // every UPDATE tile back-projects its 32x32 cells
      if (total != nullptr && (a_dev == nullptr || a_dev->enabled)) {
        // running totals of every push on this grid (read back on demand: tsd_push_stats_total)
        total[0].cells_updated += s_tot[0];
        total[0].cells_visited += 1024ull * s_tot[2];
        total[0].tiles_range_pass += (int)s_tot[1]; total[0].tiles_update += (int)s_tot[2];
        total[0].tiles_new += (int)s_tot[3]; total[0].tiles_new_from_empty += (int)s_tot[4];
        total[0].tiles_emptied_init += (int)s_tot[5]; total[0].tiles_emptied_uninit += (int)s_tot[6];
        total[1].list_count += 1;                  // pushes
      }
    }
  }
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int PX = g.PX;
  for (int li = wave; li < count; li += nwaves) {
    const int p = (int)(list[li] & ~LIST_EMPTIED);
    if (!g.flags[p]) continue;     // freeFootprint dirty list may race nothing: all listed tiles are initialised
    const int px = p % PX, py = p / PX;
    const bool hasR = px < PX - 1, hasU = py < PX - 1, hasL = px > 0, hasD = py > 0;
    // (a) own halo from right / up / up-right
    if (hasR && g.flags[p + 1]) copy_col(g, p, p + 1, lane);
    if (hasU && g.flags[p + PX]) copy_row(g, p, p + PX, lane);
    if (hasR && hasU && g.flags[p + PX + 1]) copy_corner(g, p, p + PX + 1, lane);
    // (b)-(d) neighbours whose halo mirrors this tile
    if (hasL && g.flags[p - 1]) copy_col(g, p - 1, p, lane);
    if (hasD && g.flags[p - PX]) copy_row(g, p - PX, p, lane);
    if (hasL && hasD && g.flags[p - PX - 1]) copy_corner(g, p - PX - 1, p, lane);
  }
}

// TsdGrid::freeFootprint (TsdGrid.cpp:609-638): lazily initialise touched tiles, set tsd = 1.0
// (weight untouched).  One 256-thread block per tile of the rectangle's tile range.
__global__ void __launch_bounds__(256)
k_free_footprint(GridDev g, unsigned minX, unsigned maxX, unsigned minY, unsigned maxY,
                 unsigned tx0, unsigned ty0, unsigned ntx)
{
  const unsigned tx = tx0 + blockIdx.x % ntx, ty = ty0 + blockIdx.x / ntx;
  const int p = (int)(ty * (unsigned)g.PX + tx);
  const int tid = threadIdx.x;
  double* T = g.tsd + (size_t)p * TILE_STRIDE;
  double* W = g.weight + (size_t)p * TILE_STRIDE;
  const bool fresh = g.flags[p] == 0;
  const double iw = g.init_weight[p];
  const double t_init = (iw > 0.0) ? 1.0 : __builtin_nan("");
  for (int i = tid; i < TILE_CELLS; i += 256) {
    const unsigned lx = (unsigned)(i % TILE_PITCH), ly = (unsigned)(i / TILE_PITCH);
    const unsigned col = tx * TILE_DIM + lx, row = ty * TILE_DIM + ly;
    const bool inside = lx < TILE_DIM && ly < TILE_DIM && col >= minX && col < maxX && row >= minY && row < maxY;
    if (inside) { T[i] = 1.0; if (fresh) W[i] = iw; }
    else if (fresh) { T[i] = t_init; W[i] = iw; }
  }
  __syncthreads();            // every thread has read `fresh`
  if (fresh && tid == 0) g.flags[p] = 1;
}

// PMC calibration (MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are only calibrated for 16 B/lane
// streams): a read-modify-write of n doubles with the push kernel's access shape, 8 B per lane, two
// arrays, so that the counters of a known byte count (16 n read, 16 n written) can be compared with
// what rocprofv3 reports for k_push_update.
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
int launch_free_footprint(tsd_ctx* ctx, unsigned minX, unsigned maxX, unsigned minY, unsigned maxY)
{
  if (maxX <= minX || maxY <= minY) return TSD_OK;
  const unsigned tx0 = minX / TILE_DIM, tx1 = (maxX - 1) / TILE_DIM;
  const unsigned ty0 = minY / TILE_DIM, ty1 = (maxY - 1) / TILE_DIM;
  const unsigned ntx = tx1 - tx0 + 1, nty = ty1 - ty0 + 1;
  hipLaunchKernelGGL(k_free_footprint, dim3(ntx * nty), dim3(256), 0, ctx->stream, ctx->grid, minX,
                     maxX, minY, maxY, tx0, ty0, ntx);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  // remember the touched tiles for the halo refresh of the next push (init-time call: synchronous)
  int n = ctx->n_dirty;
  for (unsigned ty = ty0; ty <= ty1; ty++)
    for (unsigned tx = tx0; tx <= tx1; tx++) {
      if (n >= ctx->dirty_cap) return set_error(ctx, TSD_E_CAPACITY, "freeFootprint dirty list", hipSuccess);
      ctx->h_dirty[n++] = ty * (unsigned)ctx->grid.PX + tx;
    }
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_dirty, ctx->h_dirty, sizeof(uint32_t) * (size_t)n,
                                    hipMemcpyHostToDevice, ctx->stream));
  ctx->h_dirty[ctx->dirty_cap] = (uint32_t)n;
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_dirty_count, &ctx->h_dirty[ctx->dirty_cap], sizeof(int),
                                    hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->n_dirty = n;
  return TSD_OK;
}

int launch_push(tsd_ctx* ctx, const PushArgs& a, const PushArgs* a_dev, const double* d_ranges, const uint8_t* d_mask)
{
  if (!d_ranges) d_ranges = ctx->d_ranges;
  if (!d_mask) d_mask = ctx->d_mask;
  const GridDev& g = ctx->grid;
  PushCounters* ctr = ctx->d_counters + (ctx->epoch & 1u);
  PushCounters* ctr_next = ctx->d_counters + ((ctx->epoch + 1u) & 1u);

  {
    ScopedKernelTimer t(ctx, "push_classify");
    const int blocks = (g.tiles + 255) / 256;
    int levels = 1;
    while ((1 << levels) <= a.beams) levels++;
    const size_t bp = (size_t)((a.beams + 1) & ~1);
    const bool rmq = a.beams <= RMQ_MAX_BEAMS;
    const size_t lds = rmq ? 2 * bp * sizeof(double) + (size_t)((a.beams + 2 + 7) & ~7) * 2 + 2 * (size_t)levels * bp * 2 + 64
                           : bp * sizeof(double) + (size_t)((a.beams + 15) & ~15) + 64;
    static size_t configured = 0;
    if (rmq && lds > configured) {
      TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_push_classify<true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured = lds;
    }
    if (rmq)
      hipLaunchKernelGGL(k_push_classify<true>, dim3(blocks), dim3(256), lds, ctx->stream, g, a, a_dev, d_ranges,
                         d_mask, ctr, ctr_next, ctx->d_list, ctx->d_block_stats);
    else
      hipLaunchKernelGGL(k_push_classify<false>, dim3(blocks), dim3(256), lds, ctx->stream, g, a, a_dev, d_ranges,
                         d_mask, ctr, ctr_next, ctx->d_list, ctx->d_block_stats);
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  {
    ScopedKernelTimer t(ctx, "push_update");
    const size_t lds = 16 + (size_t)((a.beams + 1) & ~1) * sizeof(double) + (size_t)((a.beams + 15) & ~15);
    const int blocks = g.tiles < 2048 ? g.tiles : 2048;
    hipLaunchKernelGGL(k_push_update, dim3(blocks), dim3(UPDATE_BLOCK), lds, ctx->stream, g, a, a_dev,
                       d_ranges, d_mask, ctr, ctx->d_list, ctx->d_entry_upd);
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  {
    ScopedKernelTimer t(ctx, "push_halo");
    const int blocks = g.tiles / 4 < 512 ? (g.tiles + 3) / 4 : 512;
    hipLaunchKernelGGL(k_push_halo, dim3(blocks), dim3(256), 0, ctx->stream, g, ctx->d_list,
                       &ctr->list_count, ctr, ctx->d_block_stats, (g.tiles + 255) / 256, ctx->d_entry_upd,
                       ctx->d_stat_total, a_dev);
    if (ctx->n_dirty > 0) {
      // tiles written by freeFootprint since the previous push: same refresh over the list that
      // launch_free_footprint left in d_dirty / d_dirty_count
      hipLaunchKernelGGL(k_push_halo, dim3((ctx->n_dirty + 3) / 4), dim3(256), 0, ctx->stream, g,
                         ctx->d_dirty, ctx->d_dirty_count, (PushCounters*)nullptr, (const int*)nullptr, 0,
                         (const uint32_t*)nullptr, (PushCounters*)nullptr, (const PushArgs*)nullptr);
      hipMemsetAsync(ctx->d_dirty_count, 0, sizeof(int), ctx->stream);
      ctx->n_dirty = 0;
    }
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  ctx->epoch++;
  return TSD_OK;
}

}  // namespace tsd
