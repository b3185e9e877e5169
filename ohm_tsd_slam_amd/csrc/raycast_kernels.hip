// raycast_kernels.hip -- RayCastPolar2D::calcCoordsFromCurrentViewMask (RayCastPolar2D.cpp:113-192)
// and rayCastFromCurrentView (:194-281) for gfx950.
//
// One 64-lane wavefront per beam.  The reference marches a beam one cell at a time with
// `position += ray` (REPEATED fp64 addition, :246-247) and looks for the first sign change of the
// bilinearly interpolated TSD.  The event at step k only depends on the samples of steps k-1 and k, so
// a round samples 64 x RC_S consecutive steps at once (each lane RC_S of them, all their loads in
// flight together) and finds the first event with a ballot.
//
// The positions must carry the reference's accumulated rounding bit for bit.  A serial chain of 1200
// dependent additions per beam would cost more than everything else, so it is evaluated in closed
// form: while a coordinate p stays inside one binade [2^e, 2^(e+1)) every representable value is a
// multiple of u = ulp, and fl(p + r) = p + rhat with rhat = r rounded to a multiple of u (the rounding
// of each addition is the same constant unless r/u ends in exactly .5, where ties-to-even alternates).
// A beam crosses at most a handful of binades, so per coordinate a short table of segments
// (first step, anchor value, rhat) is built once per beam -- the crossing step itself is a genuine fp64
// addition -- and step k is fma(k - n0, rhat, anchor), exact because the result is representable.  The
// loop counter `i += 1.0` of the reference (it decides how many steps run) goes through the same
// machinery.  Beams that hit the .5 case or need more than RC_MAXSEG segments (coordinates within a few
// ulps of zero) take the serial chain instead (slow path of the kernel, same results).
//
// The coarse 32-cell skip loop (:225-236) samples one position per lane; the four normal look-ups of
// TsdGrid::interpolateNormal (TsdGrid.cpp:517-546) run on 4 lanes.
//
// Latency-bound gather (L2 / Infinity-Cache resident tiles): reported as time, not as a roofline.
#include "push_device.hpp"
#include <cstring>

namespace tsd {

constexpr int RC_S = 4;            // steps per lane and round (256 steps per round)
constexpr int RC_MAXSEG = 16;
constexpr int RC_BLK = 15;         // steps per block of the fine march (one group of 16 lanes; < one tile)
constexpr int RC_HALO_WAVES = 128; // k_raycast_halo: waves ahead of the beams' that do the preceding push's halo pass (one listed tile at a time each)

struct RcSeg { double anchor, rhat; int n0, pad; };

// exponent field of a positive normal double, -1 otherwise
__device__ __forceinline__ int binade_of(double p)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(p);
  const int ef = (int)((b >> 52) & 0x7ffu);
  if ((b >> 63) != 0ull || ef == 0 || ef == 0x7ff) return -1;
  return ef;
}

__device__ __forceinline__ double pow2_from_field(int ef)      // 2^(ef - 1023) for a normal exponent field
{
  return __longlong_as_double((long long)ef << 52);
}

// Segment table of the chain v_0 = p0, v_{n+1} = fl(v_n + r) for n < N.  Wave-uniform; every lane
// computes it, lane 0 stores it.  Returns false if the closed form is not applicable (more than RC_MAXSEG segments).
__device__ bool build_segments(double p0, double r, int N, RcSeg* seg, int& nseg, int lane)
{
  nseg = 0;
  int n = 0;
  double p = p0;
  int even_ef = -2;                        // binade in which p / ulp is known to be even (tie case below)
  while (n <= N) {
    if (nseg >= RC_MAXSEG) return false;
    const int ef = binade_of(p);
    double rhat = 0.0;
    long long M = 0;                       // steps n+1 .. n+M stay in the binade and follow the closed form
    if (ef > 53 && ef < 0x7fe) {
      const double lo = pow2_from_field(ef), hi = pow2_from_field(ef + 1);
      const double u = pow2_from_field(ef - 52), uinv = pow2_from_field(2046 - (ef - 52));   // ulp and 1/ulp
      const double q = r * uinv;                         // r / ulp, exact (power-of-two scaling)
      const double rq = rint(q);                         // (half-way cases go to the even integer)
      if (fabs(q - rq) == 0.5 && even_ef != ef) {
        // r / ulp ends in exactly .5 (one beam coordinate in ~2000): every addition is a tie and goes to the EVEN
        // multiple of ulp.  From an even p / ulp the chain therefore adds the even one of floor(q), floor(q) + 1,
        // which is rint(q), every time and stays even; from an odd one the first sum lands one ulp off that
        // pattern and is even afterwards.  So: one genuine addition, then the closed form with rint(q) * ulp.
        if (lane == 0) { seg[nseg].anchor = p; seg[nseg].rhat = 0.0; seg[nseg].n0 = n; seg[nseg].pad = 0; }
        nseg++;
        p = p + r;
        n += 1;
        if (binade_of(p) == ef) even_ef = ef;
        continue;
      }
      rhat = rq * u;
      // common case: the rest of the beam stays inside this binade (fma is exact for in-binade values)
      const double pend = fma((double)(N - n), rhat, p);
      if ((rhat >= 0.0 && pend < hi) || (rhat < 0.0 && pend > lo)) {
        if (lane == 0) { seg[nseg].anchor = p; seg[nseg].rhat = rhat; seg[nseg].n0 = n; seg[nseg].pad = 0; }
        nseg++;
        return true;
      }
      if (rhat > 0.0) {
        // values strictly below 2^(e+1) were rounded on this binade's grid.  The estimate uses the
        // hardware reciprocal; the two loops make the count exact whatever its error.
        double t = floor((hi - p) * __builtin_amdgcn_rcp(rhat));
        if (!(t < 4.0e9)) t = 4.0e9;
        if (!(t >= 0.0)) t = 0.0;
        M = (long long)t;
        while (M > 0 && !(fma((double)M, rhat, p) < hi)) M--;
        while (M < 4000000000ll && fma((double)(M + 1), rhat, p) < hi) M++;
      } else if (rhat < 0.0) {
        // going down, a value that lands exactly on 2^e may have been rounded on the finer grid below:
        // stay strictly above it
        double t = floor((p - lo) * __builtin_amdgcn_rcp(-rhat));
        if (!(t < 4.0e9)) t = 4.0e9;
        if (!(t >= 0.0)) t = 0.0;
        M = (long long)t;
        while (M > 0 && !(fma((double)M, rhat, p) > lo)) M--;
        while (M < 4000000000ll && fma((double)(M + 1), rhat, p) > lo) M++;
      } else {
        M = 4000000000ll;
      }
    }
    if (lane == 0) { seg[nseg].anchor = p; seg[nseg].rhat = rhat; seg[nseg].n0 = n; seg[nseg].pad = 0; }
    nseg++;
    if ((long long)n + M >= (long long)N) break;
    const double pm = fma((double)M, rhat, p);           // last value of the segment (exact)
    p = pm + r;                                          // the crossing step is a genuine addition
    n += (int)M + 1;
  }
  return true;
}

// A loop counter i_0 = p0, i += step with step a small power of two (1.0 or 32.0) and 0 <= i < 2^40: the
// step is a multiple of every ulp in range, so inside a binade i_n = anchor + m * step exactly and only the
// crossings round.  One pass over the (at most ~14) binades, in registers: returns the number of
// iterations n = 0, 1, ... with i_n <= limit (or < limit when `strict`) and the value i_k of iteration k.
__device__ __forceinline__ int counter_chain(double p0, double step, double inv_step, double limit, bool strict,
                                             int k, double& value_k, bool& ok)
{
  int count = 0, n = 0;
  double p = p0;
  value_k = p0;
  ok = true;
  for (int it = 0; it < 64; it++) {
    const int ef = binade_of(p);
    double M = 0.0;                                        // iterations n+1 .. n+M stay in the binade
    if (ef > 0 && ef < 1023 + 40) {
      if (pow2_from_field(ef > 52 ? ef - 52 : 1) > step) { ok = false; return 0; }   // (no grid is that large)
      const double hi = pow2_from_field(ef + 1);
      M = fmax(ceil((hi - p) * inv_step) - 1.0, 0.0);      // p + m*step < hi, all exact
    }
    const int Mi = (int)fmin(M, 1.0e9);
    if (k >= n && k <= n + Mi) value_k = fma((double)(k - n), step, p);
    const bool in = strict ? (p < limit) : (p <= limit);
    if (!in) break;                                        // count == n: every earlier iteration passed
    // last m in [0, M] with p + m*step inside the limit: estimate, then make it exact (fma is exact here)
    double mm = fmin(floor((limit - p) * inv_step), M);
    if (!(mm >= 0.0)) mm = 0.0;
    if (strict) {
      while (mm > 0.0 && !(fma(mm, step, p) < limit)) mm -= 1.0;
      while (mm < M && fma(mm + 1.0, step, p) < limit) mm += 1.0;
    } else {
      while (mm > 0.0 && !(fma(mm, step, p) <= limit)) mm -= 1.0;
      while (mm < M && fma(mm + 1.0, step, p) <= limit) mm += 1.0;
    }
    if (mm < M) { count = n + (int)mm + 1; break; }        // the limit falls inside this binade
    count = n + Mi + 1;
    p = fma(M, step, p) + step;                            // the crossing step is a genuine addition
    n += Mi + 1;
  }
  return count;
}

__device__ __forceinline__ double seg_value(const RcSeg* seg, int nseg, int k)
{
  int s = 0;
  for (int j = 1; j < nseg; j++) if (seg[j].n0 <= k) s = j;
  return fma((double)(k - seg[s].n0), seg[s].rhat, seg[s].anchor);
}

// the four cells of a bilinear look-up, read at agent scope (`sc1`: past this compute unit's L1): the halo cells among them may have been
// written by another wave of the same launch (HALO below)
__device__ __forceinline__ double ld_tsd_agent(const tsd_cell_t* p)
{
#ifdef TSD_STORAGE_Q32
  const int32_t q = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return q == Q_NAN ? __builtin_nan("") : ldexp((double)q, -30);
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
template <bool HALO>
__device__ __forceinline__ Quad load_quad_rc(const tsd_cell_t* __restrict__ tile, int lx, int ly)
{
  if constexpr (!HALO) return load_quad(tile, lx, ly);
#ifdef TSD_RC_HALO_PLAIN      // (A/B only: the cells by plain reads behind the wait)
  return load_quad(tile, lx, ly);
#endif
  Quad q;
  q.t00 = ld_tsd_agent(tile + cell_off(lx, ly));     q.t01 = ld_tsd_agent(tile + cell_off(lx + 1, ly));
  q.t10 = ld_tsd_agent(tile + cell_off(lx, ly + 1)); q.t11 = ld_tsd_agent(tile + cell_off(lx + 1, ly + 1));
  return q;
}

// interpolate_bilinear (tsd_device.hpp) with the cells read the launch's way (the serial chain's look-ups)
template <bool HALO>
__device__ __forceinline__ int interpolate_bilinear_rc(const GridDev& g, double x, double y, double& tsd)
{
  if constexpr (!HALO) return interpolate_bilinear(g, x, y, tsd);
  int p, lx, ly; double dx, dy;
  if (!coord2cell(g, x, y, p, lx, ly, dx, dy)) return INTERP_INVALIDINDEX;
  if (!g.flags[p]) return INTERP_EMPTYPARTITION;
  const double wx = fabs((x - dx) * g.inv_cs), wy = fabs((y - dy) * g.inv_cs);
  const Quad q = load_quad_rc<true>(g.tsd + (size_t)p * TILE_STRIDE, lx, ly);
  tsd = q.t00 * (1. - wy) * (1. - wx) + q.t10 * wy * (1. - wx) + q.t01 * (1. - wy) * wx + q.t11 * wy * wx;
  if (isnan(tsd)) return INTERP_ISNAN;
  return INTERP_SUCCESS;
}

// one wave = one beam; shared by k_raycast (one sensor per launch) and k_raycast_batch (block row y = sensor y of a batch).
// HALO (the fused scan's ray cast, launched right behind a push that left its halo pass to it: launch_push(.., defer_halo)): the first
// waves of the launch do TsdGrid::propagateBorders for the push's listed tiles before they cast -- one wave per tile, destination cells
// stored write-through, `s_waitcnt vmcnt(0)`, one agent-scope add of the entries done to the push's counter -- and EVERY wave waits for
// that counter to reach the list's length before its first read of a cell (the clipping, the coarse traversal over the tiles' flags
// and the position tables, ~3 us, come first: by then the pass is done as a rule).  No wave of the launch has read a cell before the
// pass is complete, so no stale line of a halo can sit in any L1 or L2; the cells are read `sc1` all the same (the hand-off's form in
// MI355X_MICROARCH.md: sc1 stores, drained, a counter add per storing wave, an sc1 poll, sc1 loads).  What it saves: k_push_halo's
// launch (4.3 us: a chain of three memory round trips for a few dozen tiles) and the kernel boundary in front of it.
template <bool HALO>
__device__ __forceinline__ void
raycast_beam(const GridDev& g, const RaycastArgs& a_val, const RaycastArgs* __restrict__ a_dev, const double* __restrict__ rays,
             double* __restrict__ coords, double* __restrict__ normals, uint8_t* __restrict__ mask, double* dbg, const HaloArgs* hp = nullptr)
{
  unsigned int halo_n = 0u;
  if constexpr (HALO) {
    const HaloArgs& h = *hp;
    const unsigned int n_u = h.cnt[CNT_H];
    halo_n = n_u + h.cnt[CNT_O];
    if (blockIdx.x < (unsigned)RC_HALO_WAVES) {
      // one of the launch's RC_HALO_WAVES extra waves, at the FRONT of the grid (dispatched first): the pass, no beam
      // (the first half of these waves takes the UPDATE tiles of the halo list, the second half the emptied / halo-only tiles at the back of
      // the work list: either way a wave's first entry is at an index it knows without the list lengths -- requested with them)
      const unsigned int wv = blockIdx.x, lane_h = threadIdx.x;
      constexpr unsigned int HALF = (unsigned)RC_HALO_WAVES / 2u;
      const bool back = wv >= HALF;
      const unsigned int w2 = back ? wv - HALF : wv, n_mine = back ? halo_n - n_u : n_u;
      const unsigned int w2c = w2 < (unsigned)g.tiles ? w2 : 0u;                              // (a grid of fewer tiles than waves)
      const uint32_t first = back ? h.list[(unsigned)g.tiles - 1u - w2c] : h.list_h[w2c];    // speculative: arrives with the list lengths
      if (wv == 0 && lane_h == 0) halo_bookkeeping(h.pushes, h.a_dev, h.cx, h.cy, h.slack);
      unsigned int mine = 0u;
      for (unsigned int k = w2; k < n_mine; k += HALF) {
        const uint32_t entry = k == w2 ? first : (back ? h.list[(unsigned)g.tiles - 1u - k] : h.list_h[k]);
        halo_tile_job(g, h.dirty, h.tile_rec, entry, (int)lane_h);
        mine++;
      }
      if (mine) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's halo cells have left for memory ...
        if (lane_h == 0) __hip_atomic_fetch_add(h.cnt + CNT_HDONE, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ... before it says so
      }
      return;
    }
  }
#ifdef TSD_RC_STAMPS   // diagnostic build: cycles per phase summed over beams into dbg[0..7], max beam total in dbg[8]
  long long st_t = clock64(); const long long st_begin = st_t;
#define RSTAMP(i) do { const long long now_ = clock64(); if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 128) dbg[(blockIdx.x >> 3) * 8 + i] = (double)(now_ - st_t); st_t = now_; } while (0)
#else
#define RSTAMP(i) do {} while (0)
#endif
  const RaycastArgs a = a_dev ? *a_dev : a_val;
  const int beam = (int)blockIdx.x - (HALO ? RC_HALO_WAVES : 0);
  const int lane = threadIdx.x;
  if (beam >= a.beams) return;
  __shared__ RcSeg s_segx[RC_MAXSEG], s_segy[RC_MAXSEG];
  __shared__ int s_blk[64];                                  // blocks of the current chunk that can hold an event
  const double rx = rays[beam], ry = rays[a.beams + beam];
  const double trx = a.trx, try_ = a.try_;
  const double cs = g.cs;
  const int xDim = g.N, yDim = g.N;

  // slab clipping (RayCastPolar2D.cpp:204-222)
  double xmin = a.gxmin, ymin = a.gymin;
  if (fabs(rx) > 10e-6) xmin = ((double)(rx > 0.0 ? 0 : (xDim - 1) * cs) - trx) / rx;
  if (fabs(ry) > 10e-6) ymin = ((double)(ry > 0.0 ? 0 : (yDim - 1) * cs) - try_) / ry;
  double idxMin = fmax(xmin, ymin);
  idxMin = fmax(idxMin, 0.0);
  double xmax = a.gxmax, ymax = a.gymax;
  if (fabs(rx) > 10e-6) xmax = ((double)(rx > 0.0 ? (xDim - 1) * cs : 0) - trx) / rx;
  if (fabs(ry) > 10e-6) ymax = ((double)(ry > 0.0 ? (yDim - 1) * cs : 0) - try_) / ry;
  double idxMax = fmin(xmax, ymax);
  idxMin = fmax(idxMin, a.idx_min);
  idxMax = fmin(idxMax, a.idx_max);
  if (idxMin >= idxMax) { if (lane == 0) mask[beam] = 0; return; }

  // coarse traversal: for(i = idxMin; i < idxMax; i += 32) { if tile usable: break; else idxMin = i; }
  {
    double i_run = idxMin;          // wave-uniform loop variable of the reference (serial fall-back only)
    bool cclosed = true;
    bool done = false;
    {
      // the usual case: the first sample (i = idxMin, no counter arithmetic needed) already lies in a tile with
      // data -- the sensor stands in mapped space -- and the loop breaks at once, idxMin unchanged
      int p, lx, ly; double dx, dy;
      if (coord2cell(g, trx + idxMin * rx, try_ + idxMin * ry, p, lx, ly, dx, dy) && g.flags[p] != 0) done = true;
    }
    for (int cbase = 0; !done; cbase += 64) {
      double my_i = 0.0; bool my_act = false;
      if (cclosed) {
        bool okc;
        const int cnt = counter_chain(idxMin, 32.0, 1.0 / 32.0, idxMax, true, cbase + lane, my_i, okc);
        my_act = cbase + lane < cnt;
        cclosed = __ballot(!okc) == 0ull;
      }
      if (!cclosed) {
#pragma unroll 8
        for (int s = 0; s < 64; s++) {
          const bool act = i_run < idxMax;
          if (lane == s) { my_i = i_run; my_act = act; }
          i_run += 32.0;
        }
      }
      bool ok = false;
      if (my_act) {
        // interpolateBilinear's EMPTYPARTITION / INVALIDINDEX verdicts only need the cell index and the tile flag
        int p, lx, ly; double dx, dy;
        ok = coord2cell(g, trx + my_i * rx, try_ + my_i * ry, p, lx, ly, dx, dy) && g.flags[p] != 0;
      }
      const unsigned long long m_act = __ballot(my_act);
      const unsigned long long m_ok = __ballot(ok);
      if (m_ok) {
        const int f = __ffsll((long long)m_ok) - 1;       // first usable sample: loop breaks there
        if (f > 0) idxMin = __shfl(my_i, f - 1, 64);        // last failing i of this round
        done = true;
      } else {
        const int n_act = __popcll(m_act);
        if (n_act > 0) idxMin = __shfl(my_i, n_act - 1, 64);
        if (n_act < 64) done = true;                        // loop ran out (i >= idxMax)
      }
    }
  }

  RSTAMP(0);
  // fine march: position_0 = tr + idxMin * ray; for (i = idxMin; i <= idxMax; i += 1.0) { position += ray; ... }
  const double px0 = trx + idxMin * rx, py0 = try_ + idxMin * ry;
  const int cap = (int)fmin(fmax(idxMax - idxMin, 0.0), 1.0e6) + 2;     // more than the loop can run
  int nsx = 0, nsy = 0;
  // iterations of `for (i = idxMin; i <= idxMax; i += 1.0)`: the accumulated rounding of i is below 1e-9,
  // so unless idxMax - idxMin is that close to an integer the count is floor(idxMax - idxMin) + 1; only
  // then the exact chain decides
  int nsteps_i; bool oki = true;
  {
    const double dspan = idxMax - idxMin, fl_ = floor(dspan), fr = dspan - fl_;
    if (fr > 1e-9 && fr < 1.0 - 1e-9 && dspan < 1.0e9) nsteps_i = (int)fl_ + 1;
    else { double dummy; nsteps_i = counter_chain(idxMin, 1.0, 1.0, idxMax, false, -1, dummy, oki); }
  }
  bool closed = oki;
  {
    // the x table by lanes 0..31, the y table by lanes 32..63, at the same time
    const bool yh = lane >= 32;
    int ns = 0;
    const bool okh = build_segments(yh ? py0 : px0, yh ? ry : rx, cap, yh ? s_segy : s_segx, ns, lane & 31);
    nsx = __shfl(ns, 0, 64); nsy = __shfl(ns, 32, 64);
    closed = closed && (__ballot(!okh) == 0ull);
  }
  __syncthreads();
  RSTAMP(1);
  // HALO: no cell has been read so far (tile flags only), and none is before this has returned: from then on the push's halos stand.
  // Called right ahead of the wave's first cell read -- behind the position tables and the first chunk's mask culling (negmask words,
  // not cells), which is where the pass's own chain (list -> flags -> cells -> stores drained -> counter) has usually ended.
  // (`seen`: a reading of the counter the caller requested earlier, together with reads it had to make anyway -- the wait is then no
  // memory round trip of its own when the pass has ended, which is the rule)
  bool halo_waited = !HALO || halo_n == 0u;
  auto halo_peek = [&]() -> unsigned int {
    if constexpr (HALO) { if (!halo_waited) return __hip_atomic_load(hp->cnt + CNT_HDONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return 0u;
  };
  auto halo_wait = [&](unsigned int seen) {
    if constexpr (HALO) {
      if (!halo_waited) {
        while (seen < halo_n) { __builtin_amdgcn_s_sleep(4); seen = __hip_atomic_load(hp->cnt + CNT_HDONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        halo_waited = true;
      }
    }
  };
  const int nsteps = closed ? nsteps_i : 0;


  bool found = false, ended = false;
  double hit_x = 0.0, hit_y = 0.0;


  if (closed) {
    // the first two segments of each coordinate in registers (nearly every beam has at most two)
    const RcSeg x0 = s_segx[0], x1 = s_segx[nsx > 1 ? 1 : 0], y0 = s_segy[0], y1 = s_segy[nsy > 1 ? 1 : 0];
    const int x1n = nsx > 1 ? x1.n0 : 0x7fffffff, y1n = nsy > 1 ? y1.n0 : 0x7fffffff;
    auto posx = [&](int k) {
      if (nsx > 2) return seg_value(s_segx, nsx, k);
      return k >= x1n ? fma((double)(k - x1n), x1.rhat, x1.anchor) : fma((double)k, x0.rhat, x0.anchor);
    };
    auto posy = [&](int k) {
      if (nsy > 2) return seg_value(s_segy, nsy, k);
      return k >= y1n ? fma((double)(k - y1n), y1.rhat, y1.anchor) : fma((double)k, y0.rhat, y0.anchor);
    };
    // The reference samples every step; an event at step k (hit: prev > 0 > cur, miss: prev < 0 < cur) needs a
    // negative sample, i.e. a negative cell among the four a sample interpolates.  GridDev::negmask says for
    // every 4x4-cell group of every tile whether it has ever held a negative value (a halo cell is a copy of the
    // owning neighbour's cell, so the owner's bit stands for it).  The march looks at the path in blocks of
    // RC_BLK steps: the cells the samples of a block can touch lie in the box spanned by the block's first and
    // last sample (the path is monotone) plus one cell; when no group under that box has its bit set the block
    // cannot hold an event and is not sampled.  The others are sampled in order and the first event wins --
    // exactly what the full march finds.  Free space costs a few mask reads per block.
    // A group of 16 lanes takes one block: its first lane samples the step before the block ("prev" of the
    // block's first step), the others the block's RC_BLK steps.
    auto block_may_hold_event = [&](int k_first, int k_last) {
      int p0, lx0, ly0, p1, lx1, ly1; double dx, dy;
      if (!coord2cell(g, posx(k_first), posy(k_first), p0, lx0, ly0, dx, dy)) return true;
      if (!coord2cell(g, posx(k_last), posy(k_last), p1, lx1, ly1, dx, dy)) return true;
      const int PX = g.PX;
      const int xs = (p0 % PX) * TILE_DIM + lx0, ys = (p0 / PX) * TILE_DIM + ly0;
      const int xe = (p1 % PX) * TILE_DIM + lx1, ye = (p1 / PX) * TILE_DIM + ly1;
      const int xa = xs < xe ? xs : xe, ya = ys < ye ? ys : ye;
      int xb = (xs < xe ? xe : xs) + 1, yb = (ys < ye ? ye : ys) + 1;      // + the second column / row of the footprint
      if (xb > g.N - 1) xb = g.N - 1;
      if (yb > g.N - 1) yb = g.N - 1;
      const int txa = xa >> 5, txb = xb >> 5, tya = ya >> 5, tyb = yb >> 5;   // at most 2 x 2 tiles (RC_BLK + 2 <= 32)
      // the (up to) four tiles' masks are requested TOGETHER, whether the box reaches into a second column / row of tiles or not (then
      // the same word again): read one at a time behind its condition, a box over 2 x 2 tiles was four memory round trips in a row --
      // and the slowest beam is the kernel's duration
      unsigned long long nm[2][2];
#pragma unroll
      for (int iy = 0; iy < 2; iy++)
#pragma unroll
        for (int ix = 0; ix < 2; ix++) nm[iy][ix] = ld_pinned(&g.negmask[(iy ? tyb : tya) * PX + (ix ? txb : txa)]);
      bool any = false;
#pragma unroll
      for (int iy = 0; iy < 2; iy++) {
#pragma unroll
        for (int ix = 0; ix < 2; ix++) {
          const int tx = ix ? txb : txa, ty = iy ? tyb : tya;
          if ((ix && txb == txa) || (iy && tyb == tya)) continue;
          const int cx0 = (xa > tx * TILE_DIM ? xa : tx * TILE_DIM) - tx * TILE_DIM;
          const int cx1 = (xb < tx * TILE_DIM + 31 ? xb : tx * TILE_DIM + 31) - tx * TILE_DIM;
          const int cy0 = (ya > ty * TILE_DIM ? ya : ty * TILE_DIM) - ty * TILE_DIM;
          const int cy1 = (yb < ty * TILE_DIM + 31 ? yb : ty * TILE_DIM + 31) - ty * TILE_DIM;
          any |= (nm[iy][ix] & neg_rect(cx0 >> 2, cx1 >> 2, cy0 >> 2, cy1 >> 2)) != 0ull;
        }
      }
      return any;
    };
    const int nblk = (nsteps + RC_BLK - 1) / RC_BLK;
#ifdef TSD_RC_STAMPS
    int dbg_cand = 0, dbg_blocks = 0;
#endif
    for (int chunk = 0; chunk * 64 < nblk && !found && !ended; chunk++) {
      // which of this chunk's 64 blocks can hold an event
      const int bq = chunk * 64 + lane;
      const unsigned int halo_seen = halo_peek();              // (requested with the culling's mask reads below)
      bool cand = false;
      if (bq < nblk) {
        const int k0 = bq * RC_BLK, k1 = k0 + RC_BLK < nsteps ? k0 + RC_BLK : nsteps;
        cand = block_may_hold_event(k0, k1);
      }
      const unsigned long long m_cand = __ballot(cand);
      const int n_cand = __popcll(m_cand);
#ifdef TSD_RC_STAMPS
      dbg_cand += n_cand; dbg_blocks += (nblk - chunk * 64 < 64 ? nblk - chunk * 64 : 64);
#endif
      if (cand) s_blk[__popcll(m_cand & ((1ull << lane) - 1ull))] = bq;
      __syncthreads();
      if (n_cand) halo_wait(halo_seen);
      for (int r0 = 0; r0 < n_cand && !found && !ended; r0 += 4 * RC_S) {
        double qx[RC_S], qy[RC_S], v[RC_S];
        bool act[RC_S];
        int st[RC_S]; int lxy[RC_S]; int tp[RC_S]; double wx[RC_S], wy[RC_S];
#pragma unroll
        for (int j = 0; j < RC_S; j++) {
          act[j] = false; st[j] = INTERP_INVALIDINDEX; tp[j] = 0; lxy[j] = 0; qx[j] = 0.0; qy[j] = 0.0; wx[j] = 0.0; wy[j] = 0.0;
          if (r0 + 4 * j >= n_cand) continue;                   // (wave-uniform) no block left for this sub-round
          const int li = r0 + 4 * j + (lane >> 4);
          const int blk = li < n_cand ? s_blk[li] : -1;
          const int k = blk * RC_BLK + (lane & 15);             // lane 0 of the group: the step before the block
          qx[j] = posx(k < 0 ? 0 : k); qy[j] = posy(k < 0 ? 0 : k);
          act[j] = blk >= 0 && k <= nsteps;
          int p = 0, lx = 0, ly = 0; double dx = 0.0, dy = 0.0;
          const bool inside = act[j] && coord2cell(g, qx[j], qy[j], p, lx, ly, dx, dy);
          st[j] = inside ? INTERP_SUCCESS : INTERP_INVALIDINDEX;
          tp[j] = inside ? p : 0;
          lxy[j] = inside ? (lx | (ly << 8)) : 0;
          wx[j] = fabs((qx[j] - dx) * g.inv_cs);
          wy[j] = fabs((qy[j] - dy) * g.inv_cs);
        }
        // the tile storage exists for every tile (only `flags` says whether it holds data), so the cell
        // reads need not wait for the flag
        uint8_t fl[RC_S]; Quad qv[RC_S];
#pragma unroll
        for (int j = 0; j < RC_S; j++) {
          fl[j] = 0; qv[j].t00 = qv[j].t01 = qv[j].t10 = qv[j].t11 = 0.0;
          if (r0 + 4 * j >= n_cand) continue;                   // (wave-uniform)
          fl[j] = ld_pinned(&g.flags[tp[j]]);                   // (pinned, like the quad's reads: one memory round trip for the five)
          qv[j] = load_quad_rc<HALO>(g.tsd + (size_t)tp[j] * TILE_STRIDE, lxy[j] & 0xFF, lxy[j] >> 8);
        }
#pragma unroll
        for (int j = 0; j < RC_S; j++) {
          double r = __builtin_nan("");
          if (r0 + 4 * j < n_cand && st[j] == INTERP_SUCCESS && fl[j]) {
            r = qv[j].t00 * (1. - wy[j]) * (1. - wx[j]) + qv[j].t10 * wy[j] * (1. - wx[j])
              + qv[j].t01 * (1. - wy[j]) * wx[j] + qv[j].t11 * wy[j] * wx[j];   // NaN stays NaN = "not SUCCESS"
          }
          v[j] = r;
        }
        // events in step order: groups hold ascending blocks, lanes ascending steps
#pragma unroll
        for (int j = 0; j < RC_S; j++) {
          if (found || ended || r0 + 4 * j >= n_cand) break;
          const double cur = v[j];
          const double prev = __shfl_up(cur, 1, 64);
          const bool is_step = act[j] && (lane & 15) != 0;      // (the group's first lane only supplies `prev`)
          const bool hit = is_step && (prev > 0 && cur < 0);
          const bool miss = is_step && !hit && (prev < 0 && cur > 0);
          const unsigned long long m_hit = __ballot(hit), m_miss = __ballot(miss);
          const unsigned long long m_ev = m_hit | m_miss;
          if (m_ev) {
            const int f = __ffsll((long long)m_ev) - 1;
            if ((m_hit >> f) & 1ull) {
              // interp = tsd_prev / (tsd_prev - tsd); c = position + ray * (interp - 1)
              const double interp = prev / (prev - cur);
              const double cx = qx[j] + rx * (interp - 1.0);
              const double cy = qy[j] + ry * (interp - 1.0);
              hit_x = __shfl(cx, f, 64);
              hit_y = __shfl(cy, f, 64);
              found = true;
            }
            ended = true;
          }
        }
      }
      __syncthreads();                                         // the block list is rewritten by the next chunk
    }
#ifdef TSD_RC_STAMPS
    if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 128) { dbg[(blockIdx.x >> 3) * 8 + 2] = dbg_cand; dbg[(blockIdx.x >> 3) * 8 + 3] = dbg_blocks; }
#endif
  } else {
    // serial chain (rare): 64 steps per round, positions by the reference's own additions
    halo_wait(halo_peek());
    double carry;                                           // sample of the previous step (NaN = none)
    {
      double t0;
      carry = (interpolate_bilinear_rc<HALO>(g, px0, py0, t0) == INTERP_SUCCESS) ? t0 : __builtin_nan("");
    }
    double px = px0, py = py0;
    double i_run = idxMin;
    for (;;) {
      double mx = 0.0, my = 0.0; bool my_act = false;
#pragma unroll 8
      for (int s = 0; s < 64; s++) {
        const bool act = i_run <= idxMax;
        px += rx; py += ry;
        if (lane == s) { mx = px; my = py; my_act = act; }
        i_run += 1.0;
      }
      double cur = __builtin_nan("");
      if (my_act) {
        double t;
        if (interpolate_bilinear_rc<HALO>(g, mx, my, t) == INTERP_SUCCESS) cur = t;
      }
      double prev = __shfl_up(cur, 1, 64);
      if (lane == 0) prev = carry;
      const bool hit = my_act && (prev > 0 && cur < 0);
      const bool miss = my_act && !hit && (prev < 0 && cur > 0);
      const unsigned long long m_hit = __ballot(hit), m_miss = __ballot(miss);
      const unsigned long long m_ev = m_hit | m_miss;
      if (m_ev) {
        const int f = __ffsll((long long)m_ev) - 1;
        if ((m_hit >> f) & 1ull) {
          const double interp = prev / (prev - cur);
          const double cx = mx + rx * (interp - 1.0);
          const double cy = my + ry * (interp - 1.0);
          hit_x = __shfl(cx, f, 64);
          hit_y = __shfl(cy, f, 64);
          found = true;
        }
        break;
      }
      if (__ballot(my_act) != ~0ull) break;       // i > idxMax reached inside this round
      carry = __shfl(cur, 63, 64);
    }
  }
  RSTAMP(4);
#ifdef TSD_RC_STAMPS
  if (lane == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 128) dbg[(blockIdx.x >> 3) * 8 + 6] = closed ? 0.0 : 1.0;
#endif
  if (!found) { if (lane == 0) mask[beam] = 0; return; }

  // TsdGrid::interpolateNormal: lanes 0..3 sample (x+cs,y) (x-cs,y) (x,y+cs) (x,y-cs)
  halo_wait(halo_peek());                                    // (a hit was sampled: long done)
  double sx = hit_x, sy = hit_y;
  if (lane == 0) sx = hit_x + cs;
  else if (lane == 1) sx = hit_x - cs;
  else if (lane == 2) sy = hit_y + cs;
  else if (lane == 3) sy = hit_y - cs;
  double v = 0.0; bool okn = true;
  if (lane < 4) {
    // interpolateBilinear with the cell reads issued together with the flag read (the tile storage exists for
    // every tile; only `flags` says whether it holds data)
    int p, lx, ly; double dx, dy;
    okn = coord2cell(g, sx, sy, p, lx, ly, dx, dy);
    if (okn) {
      const uint8_t f = ld_pinned(&g.flags[p]);
      const Quad q = load_quad_rc<HALO>(g.tsd + (size_t)p * TILE_STRIDE, lx, ly);
      const double wx = fabs((sx - dx) * g.inv_cs), wy = fabs((sy - dy) * g.inv_cs);
      v = q.t00 * (1. - wy) * (1. - wx) + q.t10 * wy * (1. - wx) + q.t01 * (1. - wy) * wx + q.t11 * wy * wx;
      okn = f != 0 && !isnan(v);
    }
  }
  const bool all_ok = __ballot(!okn) == 0ull;
  const double v0 = __shfl(v, 0, 64), v1 = __shfl(v, 1, 64), v2 = __shfl(v, 2, 64), v3 = __shfl(v, 3, 64);
  if (lane == 0) {
    if (!all_ok) { mask[beam] = 0; return; }
    double nx = v0 - v1, ny = v2 - v3;
    // norm2<obfloat> (mathbase.h:211-218)
    const double len = sqrt(nx * nx + ny * ny);
    if (!(fabs(len) <= 10e-6)) { nx /= len; ny /= len; }
    // M = T * [c;1], N = T * [n;0] with T = pose^-1 (RayCastPolar2D.cpp:167-176)
    double m0 = 0.0, m1 = 0.0, n0 = 0.0, n1 = 0.0;
    m0 += a.Pi[0] * hit_x; m0 += a.Pi[1] * hit_y; m0 += a.Pi[2] * 1.0;
    m1 += a.Pi[3] * hit_x; m1 += a.Pi[4] * hit_y; m1 += a.Pi[5] * 1.0;
    n0 += a.Pi[0] * nx; n0 += a.Pi[1] * ny; n0 += a.Pi[2] * 0.0;
    n1 += a.Pi[3] * nx; n1 += a.Pi[4] * ny; n1 += a.Pi[5] * 0.0;
    coords[2 * beam] = m0; coords[2 * beam + 1] = m1;
    normals[2 * beam] = n0; normals[2 * beam + 1] = n1;
    mask[beam] = 1;
  }
  RSTAMP(5);
#ifdef TSD_RC_STAMPS
  if (lane == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 128) dbg[(blockIdx.x >> 3) * 8 + 7] = (double)(clock64() - st_begin);
#endif
}

__global__ void __launch_bounds__(64)
k_raycast(GridDev g, RaycastArgs a_val, const RaycastArgs* __restrict__ a_dev, const double* __restrict__ rays,
          double* __restrict__ coords, double* __restrict__ normals, uint8_t* __restrict__ mask, double* dbg)
{
  raycast_beam<false>(g, a_val, a_dev, rays, coords, normals, mask, dbg);
}
// the ray cast right behind a push of the fused scan: carries that push's halo pass (see raycast_beam<true>)
__global__ void __launch_bounds__(64)
k_raycast_halo(GridDev g, RaycastArgs a_val, const RaycastArgs* __restrict__ a_dev, const double* __restrict__ rays,
               double* __restrict__ coords, double* __restrict__ normals, uint8_t* __restrict__ mask, double* dbg, HaloArgs halo)
{
  raycast_beam<true>(g, a_val, a_dev, rays, coords, normals, mask, dbg, &halo);
}

// the ray casts of a batch of sensors on one grid in ONE launch (tsd_batch_begin): blockIdx.y picks the sensor
__global__ void __launch_bounds__(64)
k_raycast_batch(GridDev g, const RaycastBatchEntry* __restrict__ entries, double* dbg)
{
  const RaycastBatchEntry e = entries[blockIdx.y];
  RaycastArgs none;
  none.beams = 0;
  raycast_beam<false>(g, none, e.a_dev, e.rays, e.coords, e.normals, e.mask, dbg);
}

// the same with the entries as kernel arguments: the launch does not depend on the batch's copy of its argument tables
__global__ void __launch_bounds__(64)
k_raycast_batch_args(GridDev g, RaycastBatchArgs args, double* dbg)
{
  const RaycastBatchEntry e = args.e[blockIdx.y];
  RaycastArgs none;
  none.beams = 0;
  raycast_beam<false>(g, none, e.a_dev, e.rays, e.coords, e.normals, e.mask, dbg);
}

int launch_raycast_batch_byval(tsd_ctx* ctx, hipStream_t stream, const RaycastBatchEntry* h_entries, int n, int max_beams)
{
  RaycastBatchArgs args;
  std::memset(&args, 0, sizeof(args));
  for (int i = 0; i < n && i < RC_BATCH_BYVAL; i++) args.e[i] = h_entries[i];
  ScopedKernelTimer t(ctx, "raycast");
  hipExtLaunchKernelGGL(k_raycast_batch_args, dim3(max_beams, n), dim3(64), 0, stream, t.a, t.b, 0, ctx->grid, args, ctx->d_icp_trace);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_raycast_batch(tsd_ctx* ctx, hipStream_t stream, const RaycastBatchEntry* d_entries, int n, int max_beams)
{
  ScopedKernelTimer t(ctx, "raycast");
  hipExtLaunchKernelGGL(k_raycast_batch, dim3(max_beams, n), dim3(64), 0, stream, t.a, t.b, 0, ctx->grid, d_entries, ctx->d_icp_trace);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_raycast(tsd_ctx* ctx, const RaycastArgs& a, const RaycastArgs* a_dev, const double* d_rays, const HaloArgs* halo)
{
  ScopedKernelTimer t(ctx, "raycast");
  const LaunchTarget* tg = launch_target();       // concurrent multi-robot path: the sensor's own stream and output buffers
  hipEvent_t stop = t.b;
  if (tg && tg->rc_done && !t.b) { stop = tg->rc_done; const_cast<LaunchTarget*>(tg)->rc_done_used = true; }
  if (halo)
    hipExtLaunchKernelGGL(k_raycast_halo, dim3(a.beams + RC_HALO_WAVES), dim3(64), 0, launch_stream(ctx), t.a, stop, 0, ctx->grid, a, a_dev, d_rays ? d_rays : ctx->d_rays,
                       tg && tg->coords ? tg->coords : ctx->d_coords, tg && tg->normals ? tg->normals : ctx->d_normals,
                       tg && tg->mask_m ? tg->mask_m : ctx->d_mask_m, ctx->d_icp_trace, *halo);
  else
  hipExtLaunchKernelGGL(k_raycast, dim3(a.beams), dim3(64), 0, launch_stream(ctx), t.a, stop, 0, ctx->grid, a, a_dev, d_rays ? d_rays : ctx->d_rays,
                     tg && tg->coords ? tg->coords : ctx->d_coords, tg && tg->normals ? tg->normals : ctx->d_normals,
                     tg && tg->mask_m ? tg->mask_m : ctx->d_mask_m, ctx->d_icp_trace);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

}  // namespace tsd
