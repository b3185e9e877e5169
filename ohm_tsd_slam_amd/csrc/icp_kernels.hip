// icp_kernels.hip -- the registration step of ThreadLocalize::doRegistration, registration_mode 0
// (ThreadLocalize.cpp:571-581): Icp::iterate (Icp.cpp:464-512) over Icp::step (:410-462) with
//   OutOfBoundsFilter2D (OutOfBoundsFilter2D.cpp:27-37)  -> pre-filter
//   exact 1-NN (FlannPairAssignment.cpp:64-92)            -> uniform-grid search in LDS (below)
//   DistanceFilter (DistanceFilter.cpp:32-64)             -> d2 <= thr, thr = max(thr*m, min^2)
//   ReciprocalFilter (ReciprocalFilter.cpp:32-78)         -> best scene point per model point
//   ClosedFormEstimator2D (ClosedFormEstimator2D.cpp:36-109)
// as ONE persistent single-workgroup kernel: all icp_iterations steps run inside one launch with
// model, scene and the search structure resident in LDS, wave __shfl reductions for the centroid /
// MSE / nominator / denominator sums, and no host round trip between steps.  In the fused mode the
// kernel first does dataToCartesianVectorMask (Sensor.cpp:168-190) and the maskMatrix compaction
// (ThreadLocalize.cpp:738-755) from the ray-cast outputs.
//
// Nearest neighbour without a kd-tree, with identical filtered output: the reference finds the
// exact NN and then DROPS the pair unless d2 <= thr <= dist_filter_max^2.  So only neighbours within
// dist_filter_max matter.  Model points are counting-sorted into a uniform grid whose cell edge is
// >= dist_filter_max; the NN of a scene point, if it is going to survive the distance filter, lies in
// the 3x3 cells around it, which are three contiguous runs of the sorted arrays.  Pairs whose true NN
// is farther than dist_filter_max are dropped here exactly as the filter would drop them.
// Ties (equal d2) go to the lower original model index, like a first-minimum linear scan.
//
// No dense contraction anywhere => no MFMA; fp64 VALU + LDS.  Latency-bound: reported as ms/iterate.
#include "tsd_ctx.hpp"
#include <climits>

namespace tsd {

constexpr int ICP_THREADS = 1024;
constexpr int ICP_WAVES = ICP_THREADS / 64;
constexpr int GDIM = 64;
constexpr int GCELLS = GDIM * GDIM;
constexpr int PTS_PER_THREAD = TSD_MAX_ICP_POINTS / ICP_THREADS;   // 2

struct IcpLds {
  double* msx; double* msy;        // model, grid-sorted
  double* sx;  double* sy;         // scene (current estimate)
  double* ux;  double* uy;         // model staging (aliases nn region)
  unsigned long long* best_bits;   // per sorted model slot: min d2 (bit pattern) among its pairs
  int* morig;                      // original model index of a sorted slot
  int* best_i;                     // winning scene index per sorted model slot
  int* key;                        // cell of unsorted model point (build only)
  int* cell_end;                   // after the scatter: end offset of each cell
  double* red;                     // [ICP_WAVES][8] partials + [16] totals/broadcast
  int* ired;                       // [ICP_WAVES*2 + 8]
};

__host__ __device__ inline size_t icp_lds_layout(int cap, size_t off[12])
{
  size_t o = 0;
  off[0] = o; o += sizeof(double) * cap;          // msx
  off[1] = o; o += sizeof(double) * cap;          // msy
  off[2] = o; o += sizeof(double) * cap;          // sx
  off[3] = o; o += sizeof(double) * cap;          // sy
  off[4] = o; o += sizeof(double) * cap;          // ux  (later unused)
  off[5] = o; o += sizeof(double) * cap;          // uy / best_bits
  off[6] = o; o += sizeof(int) * cap;             // morig
  off[7] = o; o += sizeof(int) * cap;             // best_i
  off[8] = o; o += sizeof(int) * cap;             // key
  off[9] = o; o += sizeof(int) * (GCELLS + 16);   // cell_end
  off[10] = o; o += sizeof(double) * (ICP_WAVES * 8 + 16);   // red
  off[11] = o; o += sizeof(int) * (ICP_WAVES * 2 + 16);      // ired
  return (o + 15) & ~(size_t)15;
}

// sum of `nv` doubles held per thread -> totals in red[ICP_WAVES*8 .. +nv) (all threads may read them
// after the function returns).  Deterministic tree: lane shuffles, then 16 wave partials by wave 0.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red, int tid)
{
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    const double s = wave_sum(v[k]);
    if (lane == 0) red[wave * 8 + k] = s;
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) {
      double x = (lane < ICP_WAVES) ? red[lane * 8 + k] : 0.0;
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0) red[ICP_WAVES * 8 + k] = x;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = red[ICP_WAVES * 8 + k];
  __syncthreads();   // red may be reused right away
}

__global__ void __launch_bounds__(ICP_THREADS)
k_icp(IcpArgs a, int cap, const double* __restrict__ g_model, const double* __restrict__ g_scene,
      const double* __restrict__ g_coords, const uint8_t* __restrict__ g_mask_m,
      const double* __restrict__ g_rays_local, const double* __restrict__ g_ranges,
      const uint8_t* __restrict__ g_mask, IcpResultDev* __restrict__ out,
      double* __restrict__ trace /* [TSD_ICP_TRACE_MAX][4] = pairs, rms, thr_before, state */)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  size_t off[12];
  icp_lds_layout(cap, off);
  IcpLds L;
  L.msx = reinterpret_cast<double*>(smem + off[0]);
  L.msy = reinterpret_cast<double*>(smem + off[1]);
  L.sx = reinterpret_cast<double*>(smem + off[2]);
  L.sy = reinterpret_cast<double*>(smem + off[3]);
  L.ux = reinterpret_cast<double*>(smem + off[4]);
  L.uy = reinterpret_cast<double*>(smem + off[5]);
  L.best_bits = reinterpret_cast<unsigned long long*>(smem + off[5]);
  L.morig = reinterpret_cast<int*>(smem + off[6]);
  L.best_i = reinterpret_cast<int*>(smem + off[7]);
  L.key = reinterpret_cast<int*>(smem + off[8]);
  L.cell_end = reinterpret_cast<int*>(smem + off[9]);
  L.red = reinterpret_cast<double*>(smem + off[10]);
  L.ired = reinterpret_cast<int*>(smem + off[11]);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int nM = 0, nS = 0;

  // ---------------------------------------------------------------- inputs
  if (a.beams > 0) {
    // fused mode: maskMatrix compaction of the ray-cast model and of the scan's cartesian points
    int baseM = 0, baseS = 0;
    for (int b0 = 0; b0 < a.beams; b0 += ICP_THREADS) {
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
      for (int w = 0; w < ICP_WAVES; w++) {
        const int cm = L.ired[w * 2], cs_ = L.ired[w * 2 + 1];
        if (w < wave) { offM += cm; offS += cs_; }
        totM += cm; totS += cs_;
      }
      offM += __popcll(bm & lt); offS += __popcll(bs & lt);
      if (fm && offM < cap) { L.ux[offM] = g_coords[2 * b]; L.uy[offM] = g_coords[2 * b + 1]; }
      if (fs && offS < cap) {
        // coords = raysLocal(j,i) * data[i] (Sensor.cpp:176-179)
        L.sx[offS] = g_rays_local[b] * r; L.sy[offS] = g_rays_local[a.beams + b] * r;
      }
      baseM += totM; baseS += totS;
      __syncthreads();
    }
    nM = baseM; nS = baseS;
  } else {
    nM = a.n_model; nS = a.n_scene;
    for (int j = tid; j < nM; j += ICP_THREADS) { L.ux[j] = g_model[2 * j]; L.uy[j] = g_model[2 * j + 1]; }
    for (int i = tid; i < nS; i += ICP_THREADS) { L.sx[i] = g_scene[2 * i]; L.sy[i] = g_scene[2 * i + 1]; }
    __syncthreads();
  }

  double Tf[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // _Tfinal4x4 (thread 0 is the owner)
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

  // ---------------------------------------------------------------- search grid over the model
  double gminx, gminy, h;
  {
    double mn[2] = {__builtin_inf(), __builtin_inf()}, mxv[2] = {-__builtin_inf(), -__builtin_inf()};
    for (int j = tid; j < nM; j += ICP_THREADS) {
      mn[0] = fmin(mn[0], L.ux[j]); mn[1] = fmin(mn[1], L.uy[j]);
      mxv[0] = fmax(mxv[0], L.ux[j]); mxv[1] = fmax(mxv[1], L.uy[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[0] = fmin(mn[0], __shfl_down(mn[0], o, 64)); mn[1] = fmin(mn[1], __shfl_down(mn[1], o, 64));
      mxv[0] = fmax(mxv[0], __shfl_down(mxv[0], o, 64)); mxv[1] = fmax(mxv[1], __shfl_down(mxv[1], o, 64));
    }
    if (lane == 0) { L.red[wave * 8 + 0] = mn[0]; L.red[wave * 8 + 1] = mn[1]; L.red[wave * 8 + 2] = mxv[0]; L.red[wave * 8 + 3] = mxv[1]; }
    __syncthreads();
    gminx = L.red[0]; gminy = L.red[1];
    double gmaxx = L.red[2], gmaxy = L.red[3];
    for (int w = 1; w < ICP_WAVES; w++) {
      gminx = fmin(gminx, L.red[w * 8 + 0]); gminy = fmin(gminy, L.red[w * 8 + 1]);
      gmaxx = fmax(gmaxx, L.red[w * 8 + 2]); gmaxy = fmax(gmaxy, L.red[w * 8 + 3]);
    }
    const double ext = fmax(gmaxx - gminx, gmaxy - gminy);
    h = fmax(sqrt(a.thr0) * (1.0 + 1e-9), ext / (double)GDIM * (1.0 + 1e-9));
    if (!(h > 0.0)) h = 1.0;
    __syncthreads();
  }
  const double inv_h = 1.0 / h;
  for (int c = tid; c < GCELLS + 1; c += ICP_THREADS) L.cell_end[c] = 0;
  __syncthreads();
  for (int j = tid; j < nM; j += ICP_THREADS) {
    int cx = (int)fmin(fmax(floor((L.ux[j] - gminx) * inv_h), 0.0), (double)(GDIM - 1));
    int cy = (int)fmin(fmax(floor((L.uy[j] - gminy) * inv_h), 0.0), (double)(GDIM - 1));
    const int c = cy * GDIM + cx;
    L.key[j] = c;
    atomicAdd(&L.cell_end[c], 1);
  }
  __syncthreads();
  {
    // exclusive prefix sum over GCELLS counts, 4 per thread
    const int c0 = tid * (GCELLS / ICP_THREADS);
    int loc[GCELLS / ICP_THREADS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < GCELLS / ICP_THREADS; k++) { loc[k] = L.cell_end[c0 + k]; s += loc[k]; }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) L.ired[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; w++) wbase += L.ired[w];
    int run = wbase + incl - s;
#pragma unroll
    for (int k = 0; k < GCELLS / ICP_THREADS; k++) { L.cell_end[c0 + k] = run; run += loc[k]; }
    __syncthreads();
  }
  // scatter: cell_end[c] starts as the cell's begin offset and ends as its end offset.  The slot a
  // point lands in depends on atomic arrival order, so every order-sensitive loop below walks the
  // ORIGINAL model index j and goes through pos_of[j]; results are then reproducible run to run.
  for (int j = tid; j < nM; j += ICP_THREADS) {
    const int pos = atomicAdd(&L.cell_end[L.key[j]], 1);
    L.msx[pos] = L.ux[j]; L.msy[pos] = L.uy[j]; L.morig[pos] = j;
    L.key[j] = pos;                                  // key[] becomes pos_of[]
  }
  __syncthreads();
  const int* __restrict__ pos_of = L.key;

  // ---------------------------------------------------------------- iterate
  double thr = a.thr0;                       // DistanceFilter::_distSqr after reset()
  double rms_prev = 10e12;
  unsigned int conv_cnt = 0;
  const unsigned int max_it = (unsigned)a.iterations, conv_need = (unsigned)a.iterations;

  while (state == TSD_ICP_PROCESSING) {
    const double thr_before = thr;
    // -- phase A: pre-filter + NN + distance filter (per scene point), reset reciprocal slots
    for (int k = tid; k < nM; k += ICP_THREADS) { L.best_bits[k] = ~0ull; L.best_i[k] = INT_MAX; }
    int my_k[PTS_PER_THREAD]; double my_d[PTS_PER_THREAD]; bool my_keep[PTS_PER_THREAD];
#pragma unroll
    for (int q = 0; q < PTS_PER_THREAD; q++) {
      const int i = tid + q * ICP_THREADS;
      my_keep[q] = false; my_k[q] = -1; my_d[q] = __builtin_inf();
      if (i < nS) {
        const double x = L.sx[i], y = L.sy[i];
        // S.transform(P): (0 + x*R00) + y*R01, then + t (gsl/Matrix.cpp:403-432)
        double wx = 0.0, wy = 0.0;
        wx += x * a.P[0]; wx += y * a.P[1];
        wy += x * a.P[3]; wy += y * a.P[4];
        wx += a.P[2]; wy += a.P[5];
        const bool pre = !(wx < a.min_x || wx > a.max_x || wy < a.min_y || wy > a.max_y);
        if (pre) {
          const double fcx = fmin(fmax(floor((x - gminx) * inv_h), -2.0), (double)(GDIM + 1));
          const double fcy = fmin(fmax(floor((y - gminy) * inv_h), -2.0), (double)(GDIM + 1));
          const int cx = (int)fcx, cy = (int)fcy;
          const int x0 = max(cx - 1, 0), x1 = min(cx + 1, GDIM - 1);
          const int y0 = max(cy - 1, 0), y1 = min(cy + 1, GDIM - 1);
          double bd = __builtin_inf(); int bk = -1;
          if (x0 <= x1) {
            for (int r = y0; r <= y1; r++) {
              const int cb = r * GDIM + x0, ce = r * GDIM + x1;
              const int kb = (cb == 0) ? 0 : L.cell_end[cb - 1], ke = L.cell_end[ce];
              for (int k = kb; k < ke; k++) {
                const double dx = x - L.msx[k], dy = y - L.msy[k];
                const double d = dx * dx + dy * dy;
                if (d < bd) { bd = d; bk = k; }
                else if (d == bd && bk >= 0 && L.morig[k] < L.morig[bk]) { bk = k; }
              }
            }
          }
          my_k[q] = bk; my_d[q] = bd;
          my_keep[q] = (bk >= 0) && (bd <= thr);      // DistanceFilter::filter
        }
      }
    }
    // threshold schedule (DistanceFilter.cpp:62-63)
    thr *= a.multiplier;
    if (thr < a.min_sqr) thr = a.min_sqr;
    __syncthreads();
    // -- phase B/C: ReciprocalFilter = per model point keep the pair with the smallest d2
#pragma unroll
    for (int q = 0; q < PTS_PER_THREAD; q++)
      if (my_keep[q]) atomicMin(&L.best_bits[my_k[q]], (unsigned long long)__double_as_longlong(my_d[q]));
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PTS_PER_THREAD; q++)
      if (my_keep[q] && L.best_bits[my_k[q]] == (unsigned long long)__double_as_longlong(my_d[q]))
        atomicMin(&L.best_i[my_k[q]], tid + q * ICP_THREADS);
    __syncthreads();

    // -- phase D: ClosedFormEstimator2D::setPairs: centroids, "rms" (mean squared distance), count
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int j = tid; j < nM; j += ICP_THREADS) {
      const int k = pos_of[j];
      const int i = L.best_i[k];
      if (i != INT_MAX) {
        const double mxk = L.msx[k], myk = L.msy[k], sxi = L.sx[i], syi = L.sy[i];
        v[0] += mxk; v[1] += myk; v[2] += sxi; v[3] += syi;
        const double dx = sxi - mxk, dy = syi - myk;
        v[4] += dx * dx + dy * dy;
        v[5] += 1.0;
      }
    }
    block_sum<6>(v, L.red, tid);
    pairs = (int)v[5];

    if (pairs > 2) {
      const double size_inv = 1.0 / (double)pairs;
      rms = v[4] * size_inv;
      const double cmx = v[0] * size_inv, cmy = v[1] * size_inv, csx = v[2] * size_inv, csy = v[3] * size_inv;
      // -- phase F: estimateTransformation: nominator / denominator over centred pairs
      double nd[2] = {0, 0};
      for (int j = tid; j < nM; j += ICP_THREADS) {
        const int k = pos_of[j];
        const int i = L.best_i[k];
        if (i != INT_MAX) {
          const double xF = L.msx[k] - cmx, yF = L.msy[k] - cmy;
          const double xS = L.sx[i] - csx, yS = L.sy[i] - csy;
          nd[0] += yF * xS - xF * yS;
          nd[1] += xF * xS + yF * yS;
        }
      }
      // wave partials -> wave 0 finishes the sum and evaluates the closed form once for the block
      {
        const double s0 = wave_sum(nd[0]), s1 = wave_sum(nd[1]);
        if (lane == 0) { L.red[wave * 8 + 0] = s0; L.red[wave * 8 + 1] = s1; }
        __syncthreads();
        if (wave == 0) {
          double x0 = (lane < ICP_WAVES) ? L.red[lane * 8 + 0] : 0.0;
          double x1 = (lane < ICP_WAVES) ? L.red[lane * 8 + 1] : 0.0;
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { x0 += __shfl_down(x0, o, 64); x1 += __shfl_down(x1, o, 64); }
          const double nom = __shfl(x0, 0, 64), den = __shfl(x1, 0, 64);
          const double th_ = atan2(nom, den);
          const double co_ = cos(th_), si_ = sin(th_);
          if (lane == 0) {
            L.red[ICP_WAVES * 8 + 0] = co_; L.red[ICP_WAVES * 8 + 1] = si_;
            L.red[ICP_WAVES * 8 + 2] = (cmx - (co_ * csx - si_ * csy));
            L.red[ICP_WAVES * 8 + 3] = (cmy - (co_ * csy + si_ * csx));
          }
        }
        __syncthreads();
      }
      const double co = L.red[ICP_WAVES * 8 + 0], si = L.red[ICP_WAVES * 8 + 1];
      const double dX = L.red[ICP_WAVES * 8 + 2], dY = L.red[ICP_WAVES * 8 + 3];
      // applyTransformation(sceneTmp): data * R^T (dgemm NoTrans,Trans), then + t (Icp.cpp:371-408)
      for (int i = tid; i < nS; i += ICP_THREADS) {
        const double x = L.sx[i], y = L.sy[i];
        double nx = 0.0, ny = 0.0;
        nx += x * co; nx += y * (-si);
        ny += x * si; ny += y * co;
        L.sx[i] = nx + dX; L.sy[i] = ny + dY;
      }
      if (tid == 0) {
        // Tfinal = Tlast * Tfinal (Icp.cpp:452)
        const double Tl[16] = {co, -si, 0, dX, si, co, 0, dY, 0, 0, 1, 0, 0, 0, 0, 1};
        double R[16];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int c = 0; c < 4; c++) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) t += Tl[4 * r + k] * Tf[4 * k + c];
            R[4 * r + c] = t;
          }
#pragma unroll
        for (int q = 0; q < 16; q++) Tf[q] = R[q];
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
    if (tid == 0 && iter <= TSD_ICP_TRACE_MAX) {
      double* tr = trace + 4 * (iter - 1);
      tr[0] = (double)pairs; tr[1] = rms; tr[2] = thr_before; tr[3] = (double)state;
    }
    __syncthreads();
  }

  if (tid == 0) {
    // Icp::getFinalTransformation (Icp.cpp:528-546)
    out->T[0] = Tf[0]; out->T[1] = Tf[1]; out->T[2] = Tf[3];
    out->T[3] = Tf[4]; out->T[4] = Tf[5]; out->T[5] = Tf[7];
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

int launch_icp(tsd_ctx* ctx, const IcpArgs& a)
{
  const int n = a.beams > 0 ? a.beams : (a.n_model > a.n_scene ? a.n_model : a.n_scene);
  if (n > TSD_MAX_ICP_POINTS) return set_error(ctx, TSD_E_CAPACITY, "icp points > TSD_MAX_ICP_POINTS", hipSuccess);
  const int cap = icp_cap_for(n);
  size_t off[12];
  const size_t lds = icp_lds_layout(cap, off);
  static size_t configured = 0;
  if (lds > configured) {
    TSD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_icp),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = lds;
  }
  ScopedKernelTimer t(ctx, "icp");
  hipLaunchKernelGGL(k_icp, dim3(1), dim3(ICP_THREADS), lds, ctx->stream, a, cap, ctx->d_model,
                     ctx->d_scene, ctx->d_coords, ctx->d_mask_m, ctx->d_rays_local, ctx->d_ranges,
                     ctx->d_mask, ctx->d_icp_res, ctx->d_icp_trace);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

size_t icp_lds_bytes()
{
  size_t off[12];
  return icp_lds_layout(TSD_MAX_ICP_POINTS, off);
}

}  // namespace tsd
