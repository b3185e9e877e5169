// push_multi.hip -- the pushes of a BATCH of robots that share one grid (tsd_batch_push; the reference's multi-robot mode: N
// ThreadLocalize workers queue their sensors into one ThreadMapping, which pushes them one after the other, ThreadMapping.cpp:43-76)
// in ONE pass per tile instead of one pass per robot.
//
// Why this is the same grid.  TsdGrid::push (TsdGrid.cpp:217-284) reads and writes only the INTERIOR cells of a tile (the halo is
// written by propagateBorders and by increaseEmptiness, read by nobody on this path), every cell's updates depend on that cell alone,
// and whether a robot's scan updates / empties / ignores a tile (TsdGridComponent::isInRange) depends on the tile's geometry, the pose
// and the scan -- not on the tile's content.  So the robots' pushes commute ACROSS tiles; inside a tile they are applied here in the
// batch's robot order, which is the order the serial pushes run in, including the tile's own state machine (lazy init from
// _initWeight, increaseEmptiness of a materialised tile against the +1 on _initWeight of one that is not: TsdGridPartition.cpp:88-164).
// propagateBorders runs as a full sweep at the end of every reference push; its result after the LAST push of the batch is a function
// of the final interiors for every pair of tiles that are both initialised then, and leaves every other halo alone -- so one halo
// pass over the tiles the batch touched, after all of them, gives the halos the serial pushes leave.  tests/test_gpu_batch.py and
// tools/fuzz_batch.py compare the grids cell for cell (halo included) with the oracle's serial pushes.
//
//   k_mp_classify   block y = robot: isInRange for every tile of the batch's window (the union of the robots' windows), the same
//                   arithmetic as k_push_classify; per (tile, robot) the decision and the 128-byte record of the update's beam
//                   estimate, per tile a 64-bit mask of who updates / who empties; first arrival lists the tile
//   k_mp_update     one workgroup per listed tile: the tile's 33 x 33 cells into LDS once, then robot after robot -- stage the robot's
//                   scan window, phase A (fp32 beam estimate, candidates compacted), fix-up of the undecided cells, the exact part on
//                   the LDS cells -- and back to memory once.  A tile that eight robots see is read and written once, not eight times,
//                   and the eight latency chains of record -> scan -> cells -> stores become one.
//   k_mp_halo       propagateBorders for the listed tiles (gather form): own halo from the right / upper / diagonal neighbours, the
//                   left / lower / diagonal neighbours' halos from the tile's own first column / row / cell
#include "push_device.hpp"
#include <cstring>

namespace tsd {

constexpr int MP_MAX_ROBOTS = 16;                       // (arguments by value; bits of the per-tile mask)
constexpr unsigned long long MP_DIRTY = 1ull << 32;     // mask bit: freeFootprint wrote to the tile since the last push
constexpr int MP_EMPTY_SHIFT = 16;                      // mask bits 0-15: robot r updates, 16-31: robot r's scan sees past the tile

struct MultiPushRobot {
  const PushArgs* args; const double* ranges; const uint8_t* mask; const char* rmq;
  double cx, cy, slack;                                 // where the host assumed the sensor (window check)
};
struct MultiPushArgs { MultiPushRobot r[MP_MAX_ROBOTS]; int n, tx0, ty0, ntx, nty; };

// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_mp_classify(GridDev g, MultiPushArgs mp, uint32_t* __restrict__ tile_rec, const uint8_t* __restrict__ dirty, uint32_t* __restrict__ tile_totals,
              unsigned long long* __restrict__ tmask, PushListAux* __restrict__ rec, uint32_t* __restrict__ list, unsigned int* __restrict__ cnt, int parity)
{
  const int r = blockIdx.y;
  const MultiPushRobot& rb = mp.r[r];
  const PushArgs a = *rb.args;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int corner = lane & 3;
  const bool owner = corner == 0;
  const int t = blockIdx.x * 64 + ((int)threadIdx.x >> 2);
  const int ntiles = mp.ntx * mp.nty;
  const bool in_window = t < ntiles;
  const int p = in_window ? (mp.ty0 + t / mp.ntx) * g.PX + mp.tx0 + t % mp.ntx : 0;
  if (r == 0) {
    // (robot 0's blocks run whether its push is enabled or not) the window's per-push records are not kept by this path; the next
    // batch's list counter
    if (in_window && owner) tile_rec[p] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cnt[parity ^ 1] = 0u; cnt[2 + (parity ^ 1)] = 0u; }
  }
  const uint8_t t_dirty = ld_pinned(&dirty[p]);
  uint32_t kind = 0u, far_flag = 0u;
  double pw = 0.0, tcx = 0.0, tcy = 0.0;
  uint32_t win = (uint32_t)(a.beams - 1) << 16;
  double2 bd0 = make_double2(1.0, 0.0);
  if (in_window && a.enabled) {
    // TsdGridComponent::isInRange (TsdGridComponent.cpp:43-124): range cull, four corner back-projections (one lane each), the two
    // beam-range tests as table look-ups -- k_push_classify's statements
    double e[4][2], cx, cy, rad;
    tile_geometry(g, p, e, cx, cy, rad);
    double sqr = 0.0;
    { const double t0 = a.trx - cx; sqr += t0 * t0; const double t1 = a.try_ - cy; sqr += t1 * t1; }
    const double distance = sqrt(sqr);
    const double closest = distance - rad - g.max_trunc;
    const double farthest = distance + rad + g.max_trunc;
    if (!(closest > a.max_range || farthest < a.min_range)) {
      bool all_vis = true, any_vis = false;
      int lo = 0, hi = 0;
      {
        const int k = corner;
        const double ex = (k & 1) ? e[1][0] : e[0][0], ey = (k & 2) ? e[2][1] : e[0][1];
        int ik = backproject(a.Pi, ex, ey, a.phi_min, a.ang_res_inv, a.phi_lower, a.phi_upper);
        if (ik == -1) { ik = a.beams - 1; all_vis = false; }
        else if (ik == -2) { ik = 0; all_vis = false; }
        else any_vis = true;
        lo = ik; hi = ik;
      }
#pragma unroll
      for (int m = 1; m <= 2; m <<= 1) {
        const int lo2 = __shfl_xor(lo, m, 64), hi2 = __shfl_xor(hi, m, 64);
        const int av2 = __shfl_xor((int)all_vis, m, 64), an2 = __shfl_xor((int)any_vis, m, 64);
        lo = lo2 < lo ? lo2 : lo; hi = hi2 > hi ? hi2 : hi;
        all_vis = all_vis && av2 != 0; any_vis = any_vis || an2 != 0;
      }
      const RmqView rv = rmq_view(const_cast<char*>(rb.rmq), a.beams);
      bd0 = rv.bdir[(distance > 3.0 * rad && lo > 1) ? lo - 1 : 0];
      int action = 0;
      if (any_vis) {
        const int len = hi - lo + 1;
        const int k = 31 - __clz(len);
        const unsigned short* tm = rv.tmax + (size_t)k * rv.Bp;
        const unsigned short* tn = rv.tmin + (size_t)k * rv.Bp;
        const int j2 = hi - (1 << k) + 1;
        const unsigned short n0 = ld_pinned(&rv.inf[lo]), n1 = ld_pinned(&rv.inf[hi + 1]);
        const unsigned short i0 = tm[lo], i1 = tm[j2], i2 = tn[lo], i3 = tn[j2];
        const double amax = fmax(rv.A[i0], rv.A[i1]);
        const double bmin = fmin(rv.Bv[i2], rv.Bv[i3]);
        const bool has_inf = n1 != n0;
        const bool visible = amax > closest;
        const bool empty = (bmin > farthest) && (!has_inf || distance < a.low_refl);
        if (visible) action = (all_vis && empty) ? 1 : 2;
      }
      if (distance > 3.0 * rad) {
        win = (uint32_t)lo | ((uint32_t)hi << 16); far_flag = LIST_FAR;
        if (all_vis && lo >= 1 && hi <= a.beams - 2 && (double)(hi - lo) <= 0.7 * a.ang_res_inv + 2.0) far_flag |= LIST_INTERIOR;
      }
      if (action == 2) {
        kind = KIND_UPDATE;
        double dc = distance;
        if (dc > a.max_range) dc = a.max_range;
        pw = (a.max_range - dc) / a.max_range;
        pw *= pw;
        tcx = cx; tcy = cy;
      } else if (action == 1) kind = KIND_EMPTY;
      if (owner) atomicAdd(&tile_totals[(size_t)p * TOT_FIELDS + 1], 1u);
    }
  }
  bool first = false;
  if (owner && in_window) {
    unsigned long long bits = 0ull;
    if (kind == KIND_UPDATE) {
      PushListAuxBody x;
      x.entry = (uint32_t)p | far_flag | (kind << KIND_SHIFT); x.win = win; x.pw = 0.01 * pw;
      const double lcx = a.Pi[0] * tcx + a.Pi[1] * tcy + a.Pi[2], lcy = a.Pi[3] * tcx + a.Pi[4] * tcy + a.Pi[5];
      const double axx = a.Pi[0] * g.cs, axy = a.Pi[1] * g.cs, ayx = a.Pi[3] * g.cs, ayy = a.Pi[4] * g.cs;
      x.A = (float)(lcx * ayx - lcy * axx); x.B = (float)(lcx * ayy - lcy * axy);
      x.C = (float)(lcx * axx + lcy * ayx); x.D = (float)(lcx * axy + lcy * ayy);
      x.lc2 = (float)(lcx * lcx + lcy * lcy);
      x.lcx = (float)lcx; x.lcy = (float)lcy;
      x.th_c = atan2_estimate(x.lcy, x.lcx);
      x.iw = 0.0; x.flag = 0u;                 // (the tile's state is the update kernel's business here)
      x.jb0 = (win & 0xFFFFu) > 0u ? (win & 0xFFFFu) - 1u : 0u;
      x.bd = bd0;
      static_cast<PushListAuxBody&>(rec[(size_t)t * MP_MAX_ROBOTS + r]) = x;
      bits = 1ull << r;
    } else if (kind == KIND_EMPTY) bits = 1ull << (MP_EMPTY_SHIFT + r);
    if (r == 0 && t_dirty != 0) bits |= MP_DIRTY;
    if (bits) first = atomicOr(&tmask[t], bits) == 0ull;
  }
  // the tile list: the workgroup's first arrivals drawn with one atomic
  const unsigned long long fb = __ballot(first);
  __shared__ unsigned int s_w[4], s_base;
  if (lane == 0) s_w[wave] = (unsigned int)__popcll(fb);
  lds_barrier();
  if (threadIdx.x == 0) {
    const unsigned int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    s_base = tot ? atomicAdd(&cnt[parity], tot) : 0u;
  }
  lds_barrier();
  if (first) {
    unsigned int base = s_base;
    for (int w = 0; w < wave; w++) base += s_w[w];
    list[base + (unsigned)__popcll(fb & ((1ull << lane) - 1ull))] = (uint32_t)t;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
constexpr int MP_BLOCK = 256;
constexpr int MP_ARG_DOUBLES = (sizeof(PushArgs) + 7) / 8;
__host__ __device__ inline size_t mp_update_lds_bytes(int beams)
{
  const size_t bp = (size_t)((beams + 3) & ~3);
  return 2 * TILE_STRIDE * sizeof(double) + bp * sizeof(double) + 2 * TILE_DIM * sizeof(double) + ROT_N * sizeof(double2) +
         MP_MAX_ROBOTS * MP_ARG_DOUBLES * sizeof(double) + MP_MAX_ROBOTS * 32 * sizeof(uint32_t) + bp * sizeof(float) + UPD_CAND_MAX * sizeof(uint32_t);
}

// One workgroup per listed tile (tiles come off a ticket counter: a tile costs what the number of robots that see it costs).  Everything
// the robots' updates of a tile read from memory is requested in TWO round trips for all of them -- (1) the tile's mask, state and cells
// and every robot's 128-byte record, (2) the robots' scan windows, packed into one LDS pool (a far tile's window is a few dozen beams) --
// and then robot after robot runs out of LDS: phase A, the fix-up, the exact part on the LDS cells.  (First form: record -> scan window
// -> rotation table as three dependent round trips PER ROBOT: 9.5 us per robot and tile.)  The robots' arguments are staged once per
// workgroup; the rotation table (cos, sin)(k * res) of the fix-up is robot 0's, shared by every robot with the same angular resolution
// (another resolution reads its own table from memory).
__global__ void __launch_bounds__(MP_BLOCK)
k_mp_update(GridDev g, MultiPushArgs mp, uint32_t* __restrict__ tile_totals, unsigned long long* __restrict__ tmask,
            const PushListAux* __restrict__ rec, const uint32_t* __restrict__ list, unsigned int* __restrict__ cnt, int parity, int max_beams)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Bp = (max_beams + 3) & ~3;
  double* s_t = reinterpret_cast<double*>(smem);                           // [TILE_STRIDE] the tile's values ...
  double* s_w = s_t + TILE_STRIDE;                                         // [TILE_STRIDE] ... and weights (device offsets)
  double* s_ranges = s_w + TILE_STRIDE;                                    // [Bp] pool: the robots' scan windows, one after the other
  double* s_d2 = s_ranges + Bp;                                            // [2][32] (ccx - trx)^2 per column, (ccy - try)^2 per row
  double2* s_rot = reinterpret_cast<double2*>(s_d2 + 2 * TILE_DIM);        // [ROT_N] (cos, sin)(k * res) of robot 0's resolution
  double* s_args = reinterpret_cast<double*>(s_rot + ROT_N);               // [MP_MAX_ROBOTS][MP_ARG_DOUBLES] the robots' PushArgs
  uint32_t* s_rec = reinterpret_cast<uint32_t*>(s_args + MP_MAX_ROBOTS * MP_ARG_DOUBLES);      // [MP_MAX_ROBOTS][32] the tile's records
  float* s_lim = reinterpret_cast<float*>(s_rec + MP_MAX_ROBOTS * 32);     // [Bp] pool: beam_limit of every staged beam
  uint32_t* s_cand = reinterpret_cast<uint32_t*>(s_lim + Bp);              // [1024] candidates: cell | beam << 10
  __shared__ unsigned long long s_cu;                                      // candidates listed (low word) | undecided cells listed (high word)
  __shared__ unsigned int s_upd;                                           // cells updated by the current robot
  __shared__ unsigned long long s_neg;                                     // groups that received a negative value
  __shared__ int s_off[MP_MAX_ROBOTS + 1];                                 // pool offset of every robot's window (-1: staged on its turn)
  __shared__ unsigned int s_ticket;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned int n_list = cnt[parity];
  const double max_trunc = g.max_trunc, inv_max_trunc = 1.0 / max_trunc;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned ix = (unsigned)tid & 31u, iy0 = (unsigned)tid >> 5;
  const int c0 = (int)(iy0 * 32u + ix);
  const float dxc = (float)ix - 16.0f;
  const float mtf = (float)max_trunc;
  // once per workgroup: every robot's arguments, robot 0's rotation table
  for (int i = tid; i < mp.n * MP_ARG_DOUBLES; i += MP_BLOCK) {
    const int r = i / MP_ARG_DOUBLES, k = i % MP_ARG_DOUBLES;
    s_args[r * MP_ARG_DOUBLES + k] = reinterpret_cast<const double*>(mp.r[r].args)[k];
  }
  const int beams0 = mp.r[0].args->beams;
  if (tid < ROT_N) s_rot[tid] = rmq_view(const_cast<char*>(mp.r[0].rmq), beams0).rot[tid];
  __syncthreads();
  const double res_inv0 = reinterpret_cast<const PushArgs*>(s_args)->ang_res_inv;

  unsigned int li = blockIdx.x;
  while (li < n_list) {
    const int t = (int)list[li];
    const int p = (mp.ty0 + t / mp.ntx) * g.PX + mp.tx0 + t % mp.ntx;
    // ---- round trip 1: mask, tile state, cells (unconditionally: an uninitialised tile's storage exists), every robot's record
    const unsigned long long m = tmask[t];
    const uint8_t flag0 = g.flags[p];
    double iw = g.init_weight[p];
    tsd_cell_t* const T = g.tsd + (size_t)p * TILE_STRIDE;
    w_cell_t* const W = g.weight + (size_t)p * TILE_STRIDE;
    {
      const uint32_t* rw = reinterpret_cast<const uint32_t*>(rec + (size_t)t * MP_MAX_ROBOTS);
      const uint32_t w0 = rw[tid], w1 = rw[tid + MP_BLOCK];              // 16 records x 32 words = 512 words
      for (int i = tid; i < TILE_CELLS; i += MP_BLOCK) { s_t[i] = ld_tsd(T + i); s_w[i] = ld_w(W + i); }
      s_rec[tid] = w0; s_rec[tid + MP_BLOCK] = w1;
    }
    bool flag = flag0 != 0;
    const unsigned x0 = (unsigned)(p % g.PX) * TILE_DIM, y0 = (unsigned)(p / g.PX) * TILE_DIM;
    if (tid == 0) { s_cu = 0ull; s_upd = 0u; s_neg = 0ull; }
    __syncthreads();                       // (every thread has read the mask: it is given back for the next batch)
    if (tid == 0) {
      tmask[t] = 0ull;
      // the pool: the updating robots' windows one after the other while they fit
      int off = 0;
      for (int r = 0; r < mp.n; r++) {
        s_off[r] = -1;
        if (!(m & (1ull << r))) continue;
        const uint32_t win = s_rec[r * 32 + 1];
        const int beams = reinterpret_cast<const PushArgs*>(s_args + r * MP_ARG_DOUBLES)->beams;
        int wlo = (int)(win & 0xFFFFu) - 1, whi = (int)(win >> 16) + 1;
        if (wlo < 0) wlo = 0;
        if (whi > beams - 1) whi = beams - 1;
        const int len = whi - wlo + 1;
        if (off + len <= Bp) { s_off[r] = off; off += len; }
      }
      s_off[MP_MAX_ROBOTS] = off;
    }
    __syncthreads();
    // ---- round trip 2: the pooled scan windows, every read issued before the first LDS write
    {
      const int total = s_off[MP_MAX_ROBOTS];
      for (int e0 = 0; e0 < total; e0 += 4 * MP_BLOCK) {
        double rr[4]; unsigned mm[4]; int at[4]; float lo2[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int e = e0 + tid + i * MP_BLOCK;
          at[i] = -1; rr[i] = 0.0; mm[i] = 0u; lo2[i] = 0.f;
          if (e < total) {
            int r = 0;
            for (int q = 0; q < mp.n; q++) if (s_off[q] >= 0 && s_off[q] <= e) r = q;       // (offsets ascend with the robot)
            const uint32_t win = s_rec[r * 32 + 1];
            int wlo = (int)(win & 0xFFFFu) - 1;
            if (wlo < 0) wlo = 0;
            const int j = wlo + (e - s_off[r]);
            rr[i] = mp.r[r].ranges[j]; mm[i] = (unsigned)mp.r[r].mask[j]; at[i] = e;
            const double lr = reinterpret_cast<const PushArgs*>(s_args + r * MP_ARG_DOUBLES)->low_refl;
            lo2[i] = (float)(lr * lr) * 1.00001f;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) if (at[i] >= 0) { s_ranges[at[i]] = rr[i]; s_lim[at[i]] = beam_limit(rr[i], mm[i], mtf, lo2[i]); }
      }
    }
    bool changed = false, iw_changed = false;
    const bool was_init = flag;
    unsigned st_cells = 0u, st_upd = 0u, st_new = 0u, st_new_e = 0u, st_emp_i = 0u, st_emp_u = 0u;     // (thread 0's counts)
    for (int r = 0; r < mp.n; r++) {
      if (m & (1ull << (MP_EMPTY_SHIFT + r))) {
        // TsdGridPartition::increaseEmptiness (TsdGridPartition.cpp:136-164)
        if (flag) {
          __syncthreads();
          for (int i = tid; i < TILE_CELLS; i += MP_BLOCK) {
            double tv = s_t[i], wv = s_w[i];
            if (isnan(tv)) { wv += 1.0; tv = 1.0; }
            else { wv = fmin(wv + 1, MAX_WEIGHT); tv = (tv * (wv - 1.0) + 1.0) / wv; }
            s_t[i] = tv; s_w[i] = wv;
          }
          changed = true; st_emp_i++;
          __syncthreads();
        } else { iw = fmin(iw + 1.0, MAX_WEIGHT); iw_changed = true; st_emp_u++; }
        continue;
      }
      if (!(m & (1ull << r))) continue;
      // ---- robot r updates this tile (TsdGrid.cpp:237-274)
      const MultiPushRobot& rb = mp.r[r];
      const PushArgs& a = *reinterpret_cast<const PushArgs*>(s_args + r * MP_ARG_DOUBLES);
      const PushListAux& x = *reinterpret_cast<const PushListAux*>(s_rec + r * 32);
      if (!flag) {
        // lazy TsdGridPartition::init (TsdGridPartition.cpp:88-134): every cell, halo included, starts from the init value
        const double t_init = (iw > 0.0) ? 1.0 : __builtin_nan("");
        for (int i = tid; i < TILE_CELLS; i += MP_BLOCK) { s_t[i] = t_init; s_w[i] = iw; }
        flag = true; st_new++; if (iw > 0.0) st_new_e++;
      }
      int wlo = (int)(x.win & 0xFFFFu) - 1, whi = (int)(x.win >> 16) + 1;
      if (wlo < 0) wlo = 0;
      if (whi > a.beams - 1) whi = a.beams - 1;
      const float low2f = (float)(a.low_refl * a.low_refl) * 1.00001f;
      int pool = s_off[r];                 // where this robot's window sits in the pool
      if (pool < 0) {
        // (did not fit beside the others -- a robot that stands on the tile sees it with its whole scan: staged on its turn, at the
        // pool's start, which the robots before it are through with)
        __syncthreads();
        for (int j0 = wlo; j0 <= whi; j0 += 4 * MP_BLOCK) {
          double rr[4]; unsigned mm[4];
#pragma unroll
          for (int i = 0; i < 4; i++) { const int j = j0 + tid + i * MP_BLOCK, jc = j <= whi ? j : wlo; rr[i] = rb.ranges[jc]; mm[i] = (unsigned)rb.mask[jc]; }
#pragma unroll
          for (int i = 0; i < 4; i++) { const int j = j0 + tid + i * MP_BLOCK; if (j <= whi) { s_ranges[j - wlo] = rr[i]; s_lim[j - wlo] = beam_limit(rr[i], mm[i], mtf, low2f); } }
        }
        pool = 0;
        // (pooled windows of LATER robots that sat there are gone: they are staged on their turn too)
        if (tid == 0) for (int q = r + 1; q < mp.n; q++) s_off[q] = -1;
      }
      const int pb = pool - wlo;           // pool index of beam j: pb + j
      if (tid < 2 * TILE_DIM) {
        const bool col = tid < TILE_DIM;
        const unsigned i = (unsigned)tid & 31u;
        const double cc = ((double)((col ? x0 : y0) + i) + 0.5) * g.cs;       // TsdGridPartition.cpp:127-128
        const double dw = cc - (col ? a.trx : a.try_);
        s_d2[tid] = dw * dw;
      }
      // ---- phase A (fp32): the beam estimate of every cell, candidates compacted (k_push_update's, see push_device.hpp)
      TileA ta;
      ta.phi_min = (float)a.phi_min; ta.inv_res = (float)a.ang_res_inv; ta.beams = a.beams;
      ta.mt = mtf; ta.low2 = low2f;
      ta.axx = (float)(a.Pi[0] * g.cs); ta.axy = (float)(a.Pi[1] * g.cs); ta.ayx = (float)(a.Pi[3] * g.cs); ta.ayy = (float)(a.Pi[4] * g.cs);
      ta.cs2 = (float)(g.cs * g.cs);
      ta.wlo = wlo; ta.whi = whi;
      ta.A = x.A; ta.B = x.B; ta.C = x.C; ta.D = x.D; ta.lc2 = x.lc2; ta.th_c = x.th_c; ta.lcx = x.lcx; ta.lcy = x.lcy;
      const bool far = (x.entry & LIST_FAR) != 0u, interior = (x.entry & LIST_INTERIOR) != 0u;
      const float pA = dxc * ta.A, pC = fmaf(dxc, ta.C, ta.lc2), qx = fmaf(ta.cs2 * dxc, dxc, -ta.lc2);
      const float vc = fmaf(ta.th_c - ta.phi_min, ta.inv_res, 0.5f);
      int idx[UPD_CPT]; float d2f[UPD_CPT];
      bool uns[UPD_CPT], in[UPD_CPT];
#pragma unroll
      for (int k = 0; k < UPD_CPT; k++) {
        const float dyc = (float)(iy0 + 8u * (unsigned)k) - 16.0f;
        CellClass cc;
        if (interior) cc = classify_cell<true, true>(ta, dxc, dyc, pA, pC, qx, vc, d2f[k]);
        else if (far) cc = classify_cell<true, false>(ta, dxc, dyc, pA, pC, qx, vc, d2f[k]);
        else          cc = classify_cell<false, false>(ta, dxc, dyc, pA, pC, qx, vc, d2f[k]);
        idx[k] = cc.j; uns[k] = cc.uns; in[k] = !cc.uns && !cc.out;
      }
      lds_barrier();                       // the windows, the distance tables and (a fresh tile) the init values are in LDS
      float lim[UPD_CPT];
#pragma unroll
      for (int k = 0; k < UPD_CPT; k++) {
        const int il = min(max(idx[k], wlo), whi);
        lim[k] = s_lim[pb + il];
        if (in[k] && il != idx[k]) { in[k] = false; uns[k] = true; }
      }
      bool cand[UPD_CPT];
#pragma unroll
      for (int k = 0; k < UPD_CPT; k++) cand[k] = in[k] && !(d2f[k] > lim[k]);
      unsigned long long bc[UPD_CPT], bu[UPD_CPT];
      unsigned nc = 0u, nu = 0u;
#pragma unroll
      for (int k = 0; k < UPD_CPT; k++) { bc[k] = __ballot(cand[k]); bu[k] = __ballot(uns[k]); nc += (unsigned)__popcll(bc[k]); nu += (unsigned)__popcll(bu[k]); }
      unsigned base = 0u, ub = 0u;
      if (nc | nu) {
        unsigned long long got = 0ull;
        if (lane == 0) got = atomicAdd(&s_cu, (unsigned long long)nc | ((unsigned long long)nu << 32));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)got);
        ub = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(got >> 32));
      }
#pragma unroll
      for (int k = 0; k < UPD_CPT; k++) {
        const uint32_t e = (uint32_t)(c0 + MP_BLOCK * k) | ((uint32_t)idx[k] << 10);
        if (cand[k]) s_cand[base + (unsigned)__popcll(bc[k] & lt)] = e;
        if (bu[k] && uns[k]) s_cand[(unsigned)(UPD_CAND_MAX - 1) - (ub + (unsigned)__popcll(bu[k] & lt))] = e;
        base += (unsigned)__popcll(bc[k]); ub += (unsigned)__popcll(bu[k]);
      }
      lds_barrier();
      const unsigned long long cu = s_cu;
      const unsigned n_cand = (unsigned)cu, n_uns = (unsigned)(cu >> 32);
      const unsigned n_tot = n_cand + n_uns;
      const bool own_rot = a.ang_res_inv != res_inv0;                       // (another scanner than robot 0's: its own table, from memory)
      // ---- fix-up of the undecided cells, one lane each, in place (entry n_cand + u of the exact part lives at list[1023 - u])
      for (unsigned u = ((unsigned)tid - n_cand) & (unsigned)(MP_BLOCK - 1); u < n_uns; u += MP_BLOCK) {
        const uint32_t e = s_cand[(unsigned)(UPD_CAND_MAX - 1) - u];
        const int c = (int)(e & 1023u);
        const int jbq = (int)(e >> 10);
        const double ccx = ((double)(x0 + ((unsigned)c & 31u)) + 0.5) * g.cs;
        const double ccy = ((double)(y0 + ((unsigned)c >> 5)) + 0.5) * g.cs;
        double lx = 0.0, ly = 0.0;
        lx += a.Pi[0] * ccx; lx += a.Pi[1] * ccy; lx += a.Pi[2] * 1.0;
        ly += a.Pi[3] * ccx; ly += a.Pi[4] * ccy; ly += a.Pi[5] * 1.0;
        int index = -1;
        bool hard = jbq == IDX_CUT;
        const double l2 = lx * lx + ly * ly;
        if (!hard) {
          const int kr = jbq - (int)x.jb0;
          const double2 bdc = x.bd;
          double2 bd;
          if (__builtin_expect(kr >= 0 && kr < ROT_N, 1)) {
            const double2 rc = own_rot ? rmq_view(const_cast<char*>(rb.rmq), a.beams).rot[kr] : s_rot[kr];
            bd.x = bdc.x * rc.x - bdc.y * rc.y; bd.y = bdc.y * rc.x + bdc.x * rc.y;
          } else bd = rmq_view(const_cast<char*>(rb.rmq), a.beams).bdir[jbq];
          const double cr = bd.x * ly - bd.y * lx;                  // |l| sin(angle - beta_jb)
          if (cr * cr > 1e-22 * l2) index = cr > 0.0 ? (jbq < a.beams ? jbq : -1) : jbq - 1;
          else hard = true;
        }
        if (__builtin_expect(__any(hard), 0)) {
          if (hard) index = backproject_cold(rb.args, ccx, ccy);
        }
        bool cd = false;
        if (index >= 0) {
          const int il = min(max(index, wlo), whi);
          float lm = s_lim[pb + il];
          if (__builtin_expect(il != index, 0)) lm = beam_limit(rb.ranges[index], (unsigned)rb.mask[index], mtf, low2f);
          cd = !((float)l2 > lm * 1.00001f);
        }
        s_cand[(unsigned)(UPD_CAND_MAX - 1) - u] = cd ? ((uint32_t)c | ((uint32_t)index << 10)) : 0xFFFFFFFFu;
      }
      // ---- the exact part on the tile's cells in LDS (a cell belongs to one lane of one pass: no two lanes meet)
      unsigned long long wrote_neg = 0ull;
      unsigned n_upd = 0u;
      const double w_meas = x.pw;
      for (unsigned q = (unsigned)tid; q < n_tot; q += MP_BLOCK) {
        const uint32_t ce = s_cand[q < n_cand ? q : ((unsigned)(UPD_CAND_MAX - 1) + n_cand) - q];
        const bool on = ce != 0xFFFFFFFFu;
        const int c = (int)(ce & 1023u);
        const int index = on ? (int)(ce >> 10) : wlo;
        const int il = min(max(index, wlo), whi);
        double rg = s_ranges[pb + il];
        if (__builtin_expect(il != index, 0)) rg = rb.ranges[index];
        const double dist = sqrt_normal(s_d2[c & 31] + s_d2[TILE_DIM + (c >> 5)]);
        double sd = 0.0; bool ok = false;
        if (!isinf(rg)) { sd = rg - dist; ok = true; }
        else if (dist < a.low_refl) { sd = max_trunc; ok = true; }
        double tv = s_t[c], wv = s_w[c];
        bool touched = false;
        if (on && ok && sd >= -max_trunc) touched = add_tsd(tv, wv, sd, w_meas, max_trunc, inv_max_trunc);
        n_upd += (unsigned)__popcll(__ballot(touched));
        if (touched) { s_t[c] = tv; s_w[c] = wv; if (tv < 0.0) wrote_neg |= neg_bit((unsigned)c & 31u, (unsigned)c >> 5); }
      }
      if (wrote_neg) atomicOr(&s_neg, wrote_neg);
      if (lane == 0 && n_upd) atomicAdd(&s_upd, n_upd);
      __syncthreads();                     // robot r is through with the tile and its lists
      if (tid == 0) { st_cells += s_upd; st_upd++; s_upd = 0u; s_cu = 0ull; }
      changed = true;
    }
    // ---- the tile goes back to memory once; the next tile's ticket is drawn meanwhile
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(&cnt[2 + parity], 1u);
    if (changed) for (int i = tid; i < TILE_CELLS; i += MP_BLOCK) st_cell(T, W, i, s_t[i], s_w[i]);
    if (tid == 0) {
      if (changed && !was_init) g.flags[p] = 1;
      if (iw_changed) g.init_weight[p] = iw;
      const unsigned long long nm = s_neg;
      if (nm) atomicOr(&g.negmask[p], nm);
      uint32_t* tot = tile_totals + (size_t)p * TOT_FIELDS;
      if (st_cells) atomicAdd(&tot[0], st_cells);
      if (st_upd) atomicAdd(&tot[2], st_upd);
      if (st_new) atomicAdd(&tot[3], st_new);
      if (st_new_e) atomicAdd(&tot[4], st_new_e);
      if (st_emp_i) atomicAdd(&tot[5], st_emp_i);
      if (st_emp_u) atomicAdd(&tot[6], st_emp_u);
    }
    __syncthreads();                       // (the LDS tile and the counters are reused by the workgroup's next tile)
    li = gridDim.x + s_ticket;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// TsdGrid::propagateBorders (TsdGrid.cpp:372-427) for the tiles the batch listed, one wave per tile: the tile's own halo from the right
// / upper / diagonal neighbour, the left / lower / diagonal neighbours' halos from the tile's first column / row / cell -- wherever
// both tiles hold data now.  (Two listed neighbours write the same values into the same cells: harmless.)
__global__ void __launch_bounds__(256)
k_mp_halo(GridDev g, MultiPushArgs mp, uint8_t* __restrict__ dirty, unsigned long long* __restrict__ pushes, const uint32_t* __restrict__ list,
          const unsigned int* __restrict__ cnt, int parity)
{
  const int lane = threadIdx.x & 63;
  const unsigned int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv == 0 && lane < mp.n) {
    // (one lane per robot) the pushes that ran, and the check that every sensor was inside the square its window was laid around
    const PushArgs* a = mp.r[lane].args;
    if (a->enabled != 0) {
      atomicAdd(&pushes[0], 1ull);
      if (!(fabs(a->trx - mp.r[lane].cx) <= mp.r[lane].slack && fabs(a->try_ - mp.r[lane].cy) <= mp.r[lane].slack)) atomicAdd(&pushes[1], 1ull);
    }
  }
  const unsigned int n_list = cnt[parity];
  const int PX = g.PX;
  const bool colhalf = lane < TILE_DIM;
  const int i = lane & 31;
  for (unsigned int li = wv; li < n_list; li += gridDim.x * 4) {
    const int t = (int)list[li];
    const int p = (mp.ty0 + t / mp.ntx) * PX + mp.tx0 + t % mp.ntx;
    const int px = p % PX, py = p / PX;
    const bool hasR = px < PX - 1, hasU = py < PX - 1, hasL = px > 0, hasD = py > 0;
    const int qR = hasR ? p + 1 : p, qU = hasU ? p + PX : p, qUR = (hasR && hasU) ? p + PX + 1 : p;
    const int qL = hasL ? p - 1 : p, qD = hasD ? p - PX : p, qDL = (hasL && hasD) ? p - PX - 1 : p;
    const uint8_t f0 = g.flags[p], dty = dirty[p];
    const uint8_t fR_ = g.flags[qR], fU_ = g.flags[qU], fUR_ = g.flags[qUR], fL_ = g.flags[qL], fD_ = g.flags[qD], fDL_ = g.flags[qDL];
    if (lane == 0 && dty != 0) dirty[p] = 0;
    if (!f0) continue;
    const uint8_t fR = hasR ? fR_ : (uint8_t)0, fU = hasU ? fU_ : (uint8_t)0, fUR = (hasR && hasU) ? fUR_ : (uint8_t)0;
    const uint8_t fL = hasL ? fL_ : (uint8_t)0, fD = hasD ? fD_ : (uint8_t)0, fDL = (hasL && hasD) ? fDL_ : (uint8_t)0;
    const size_t own = (size_t)p * TILE_STRIDE;
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
    for (int k = 0; k < 3; k++) { const size_t sk = on[k] ? src[k] : own; tv[k] = g.tsd[sk]; wv_[k] = g.weight[sk]; }
#pragma unroll
    for (int k = 0; k < 3; k++) if (on[k]) { g.tsd[dst[k]] = tv[k]; g.weight[dst[k]] = wv_[k]; }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// n robots' pushes in one pass (n <= MP_MAX_ROBOTS); every robot's range-query tables must be in its rmq buffer (ordered before this).
int push_multi_max_robots() { return MP_MAX_ROBOTS; }
int launch_push_multi(tsd_ctx* ctx, hipStream_t stream, int n, const PushArgs* const* a_dev, const double* const* d_ranges, const uint8_t* const* d_mask,
                      const char* const* d_rmq, const double* cx, const double* cy, const double* slack, const int* beams, const double* max_range)
{
  if (n < 1 || n > MP_MAX_ROBOTS) return set_error(ctx, TSD_E_ARG, "launch_push_multi: robots", hipSuccess);
  const GridDev& g = ctx->grid;
  MultiPushArgs mp;
  std::memset(&mp, 0, sizeof(mp));
  mp.n = n;
  TileBox box;
  int max_beams = 1;
  for (int i = 0; i < n; i++) {
    mp.r[i] = MultiPushRobot{a_dev[i], d_ranges[i], d_mask[i], d_rmq[i], cx[i], cy[i], slack[i] + g.cs};
    if (beams[i] > max_beams) max_beams = beams[i];
    // the robot's own window (launch_push's rule): a tile passes the range cull only within max_range + radius + max_trunc of the sensor
    const double tile = TILE_DIM * g.cs;
    const double reach = max_range[i] + 0.75 * tile + g.max_trunc + slack[i] + g.cs;
    const double last = (double)(g.PX - 1);
    const double fx0 = floor((cx[i] - reach) / tile) - 1.0, fy0 = floor((cy[i] - reach) / tile) - 1.0;
    const double fx1 = floor((cx[i] + reach) / tile) + 1.0, fy1 = floor((cy[i] + reach) / tile) + 1.0;
    TileBox b;
    if (!(reach < 1e300) || !(fx0 == fx0)) { b.x0 = 0; b.y0 = 0; b.x1 = g.PX - 1; b.y1 = g.PX - 1; }
    else {
      b.x0 = (int)fmax(0.0, fmin(last, fx0)); b.y0 = (int)fmax(0.0, fmin(last, fy0));
      b.x1 = (int)fmax(0.0, fmin(last, fx1)); b.y1 = (int)fmax(0.0, fmin(last, fy1));
    }
    box.add(b);
  }
  const TileBox cur = box;
  box.add(ctx->box_prev);
  box.add(ctx->box_dirty);
  ctx->box_prev = cur; ctx->box_dirty = TileBox{};
  mp.tx0 = box.x0; mp.ty0 = box.y0; mp.ntx = box.x1 - box.x0 + 1; mp.nty = box.y1 - box.y0 + 1;
  const size_t n_window = (size_t)mp.ntx * (size_t)mp.nty;
  // per-window-tile state of this path: the masks (zero between batches: every listed tile's workgroup gives its word back), the
  // (tile, robot) records, the tile list and its two counters
  if (n_window > ctx->mp_tiles) {
    TSD_HIP_CHECK(ctx, hipStreamSynchronize(stream));
    if (ctx->d_mp_mask) hipFree(ctx->d_mp_mask);
    if (ctx->d_mp_rec) hipFree(ctx->d_mp_rec);
    if (ctx->d_mp_list) hipFree(ctx->d_mp_list);
    ctx->d_mp_mask = nullptr; ctx->d_mp_rec = nullptr; ctx->d_mp_list = nullptr; ctx->mp_tiles = 0;
    const size_t cap = n_window + n_window / 4 + 256;
    TSD_HIP_CHECK(ctx, hipMalloc(&ctx->d_mp_mask, cap * sizeof(unsigned long long)));
    TSD_HIP_CHECK(ctx, hipMalloc(&ctx->d_mp_rec, cap * MP_MAX_ROBOTS * sizeof(PushListAux)));
    TSD_HIP_CHECK(ctx, hipMalloc(&ctx->d_mp_list, (cap + 4) * sizeof(uint32_t)));
    TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_mp_mask, 0, cap * sizeof(unsigned long long), stream));
    TSD_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_mp_list, 0, (cap + 4) * sizeof(uint32_t), stream));
    ctx->mp_tiles = cap;
    ctx->mp_parity = 0;
  }
  unsigned long long* tmask = reinterpret_cast<unsigned long long*>(ctx->d_mp_mask);
  PushListAux* rec = reinterpret_cast<PushListAux*>(ctx->d_mp_rec);
  unsigned int* cnt = reinterpret_cast<unsigned int*>(ctx->d_mp_list);            // [2] list lengths by parity, [2] ticket counters, then the list
  uint32_t* list = reinterpret_cast<uint32_t*>(ctx->d_mp_list) + 4;
  const int parity = (int)(ctx->mp_parity & 1u);
  ctx->mp_parity++;
  {
    ScopedKernelTimer t(ctx, "push_classify");
    hipExtLaunchKernelGGL(k_mp_classify, dim3((unsigned)((n_window + 63) / 64), (unsigned)n), dim3(256), 0, stream, t.a, t.b, 0, g, mp, ctx->d_tile_rec,
                          ctx->d_dirty, ctx->d_tile_totals, tmask, rec, list, cnt, parity);
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  const size_t lds = mp_update_lds_bytes(max_beams);
  {
    std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
    size_t& configured = ctx->lds_configured[reinterpret_cast<const void*>(k_mp_update)];
    if (lds > configured) {
      TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mp_update), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured = lds;
    }
  }
  int per_cu = (int)((160u * 1024u) / (lds + 256));
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  const size_t resident = (size_t)ctx->n_cus * (size_t)per_cu;
  const unsigned groups = (unsigned)(n_window < resident ? n_window : resident);
  {
    ScopedKernelTimer t(ctx, "push_update");
    hipExtLaunchKernelGGL(k_mp_update, dim3(groups), dim3(MP_BLOCK), lds, stream, t.a, t.b, 0, g, mp, ctx->d_tile_totals, tmask, rec, list, cnt, parity,
                          max_beams);
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  {
    ScopedKernelTimer t(ctx, "push_halo");
    const unsigned waves = (unsigned)(n_window < 4096 ? n_window : 4096);
    hipExtLaunchKernelGGL(k_mp_halo, dim3((waves + 3) / 4), dim3(256), 0, stream, t.a, t.b, 0, g, mp, ctx->d_dirty, ctx->d_pushes, list, cnt, parity);
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

}  // namespace tsd
