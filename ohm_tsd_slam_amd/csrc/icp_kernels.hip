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
// exact NN and then DROPS the pair unless d2 <= thr <= dist_filter_max^2, so only neighbours within
// sqrt(thr) matter.  The model is sorted once per scan (bitonic sort in LDS) by (strip, x): strips are
// horizontal bands of height h >= dist_filter_max, inside a strip points ascend in x.  A scene point
// then only has to look at the window |dx| <= sqrt(limit) of its own strip and of the two neighbouring
// strips, where limit = min(thr, best d2 so far).  From the second iteration on, the search starts at
// the previous iteration's neighbour, which already bounds the window to a few centimetres, so a step
// evaluates a handful of candidates per point instead of hundreds.  Everything within sqrt(thr) is
// always visited, hence the result equals exact-NN + DistanceFilter.  Ties (equal d2) go to the lower
// original model index, like a first-minimum linear scan.
//
// No dense contraction anywhere => no MFMA; fp64 VALU + LDS.  Latency-bound: reported as ms/iterate.
#include "tsd_ctx.hpp"
#include <climits>

namespace tsd {

constexpr int ICP_THREADS = 1024;
constexpr int ICP_WAVES = ICP_THREADS / 64;
constexpr int MAX_STRIPS = 4096;
constexpr int PTS_PER_THREAD = TSD_MAX_ICP_POINTS / ICP_THREADS;   // 2
constexpr double QSCALE = 1048576.0;            // x quantisation of the sort key: 2^-20 m
constexpr double QMARGIN = 4.0 / 1048576.0;     // window slack covering the quantisation disorder

struct IcpLds {
  double* msx; double* msy;        // model, sorted by (strip, x)
  double* sx;  double* sy;         // scene (current estimate)
  double* ux;  double* uy;         // model staging (uy is reused as best_bits)
  unsigned long long* keys;        // sort keys: strip << 52 | qx << 12 | original index
  unsigned long long* best_bits;   // per sorted model slot: min d2 (bit pattern) among its pairs
  int* morig;                      // original model index of a sorted slot
  int* best_i;                     // winning scene index per sorted model slot
  int* pos_of;                     // sorted slot of original model index
  int* strip_start;                // [MAX_STRIPS + 2] first sorted slot of every strip
  double* red;                     // [ICP_WAVES][8] partials + [16] totals/broadcast
  int* ired;                       // [ICP_WAVES*2 + 16]
};

__host__ __device__ inline int icp_pow2(int n) { int p = 64; while (p < n) p <<= 1; return p; }

__host__ __device__ inline size_t icp_lds_bytes_for(int cap)
{
  return sizeof(double) * 6 * (size_t)cap + sizeof(unsigned long long) * (size_t)icp_pow2(cap) +
         sizeof(int) * 3 * (size_t)cap + sizeof(int) * (MAX_STRIPS + 16) +
         sizeof(double) * (ICP_WAVES * 8 + 16) + sizeof(int) * (ICP_WAVES * 2 + 16) + 64;
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
  IcpLds L;
  {
    char* p = smem;
    const size_t cd = sizeof(double) * (size_t)cap, ci = sizeof(int) * (size_t)cap;
    L.msx = reinterpret_cast<double*>(p); p += cd;
    L.msy = reinterpret_cast<double*>(p); p += cd;
    L.sx = reinterpret_cast<double*>(p); p += cd;
    L.sy = reinterpret_cast<double*>(p); p += cd;
    L.ux = reinterpret_cast<double*>(p); p += cd;
    L.uy = reinterpret_cast<double*>(p); L.best_bits = reinterpret_cast<unsigned long long*>(p); p += cd;
    L.keys = reinterpret_cast<unsigned long long*>(p); p += sizeof(unsigned long long) * (size_t)icp_pow2(cap);
    L.red = reinterpret_cast<double*>(p); p += sizeof(double) * (ICP_WAVES * 8 + 16);
    L.morig = reinterpret_cast<int*>(p); p += ci;
    L.best_i = reinterpret_cast<int*>(p); p += ci;
    L.pos_of = reinterpret_cast<int*>(p); p += ci;
    L.strip_start = reinterpret_cast<int*>(p); p += sizeof(int) * (MAX_STRIPS + 16);
    L.ired = reinterpret_cast<int*>(p);
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int nM = 0, nS = 0;
#ifdef TSD_ICP_STAMPS   // diagnostic build: cycles per phase (thread 0), written behind the trace records
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long st_t = clock64();
#define STAMP(i) do { const long long now_ = clock64(); st_acc[i] += now_ - st_t; st_t = now_; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

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

  // ---------------------------------------------------------------- sort the model by (strip, x)
  double gminx, gminy, h;
  int nstrips;
  {
    double mn[2] = {__builtin_inf(), __builtin_inf()}, mxv = -__builtin_inf();
    for (int j = tid; j < nM; j += ICP_THREADS) {
      mn[0] = fmin(mn[0], L.ux[j]); mn[1] = fmin(mn[1], L.uy[j]);
      mxv = fmax(mxv, L.uy[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[0] = fmin(mn[0], __shfl_down(mn[0], o, 64)); mn[1] = fmin(mn[1], __shfl_down(mn[1], o, 64));
      mxv = fmax(mxv, __shfl_down(mxv, o, 64));
    }
    if (lane == 0) { L.red[wave * 8 + 0] = mn[0]; L.red[wave * 8 + 1] = mn[1]; L.red[wave * 8 + 2] = mxv; }
    __syncthreads();
    gminx = L.red[0]; gminy = L.red[1];
    double gmaxy = L.red[2];
    for (int w = 1; w < ICP_WAVES; w++) {
      gminx = fmin(gminx, L.red[w * 8 + 0]); gminy = fmin(gminy, L.red[w * 8 + 1]);
      gmaxy = fmax(gmaxy, L.red[w * 8 + 2]);
    }
    // strip height >= sqrt(thr0) so that everything within the distance threshold of a point lies in
    // its own strip or a direct neighbour; at most MAX_STRIPS - 2 strips
    h = fmax(sqrt(a.thr0) * (1.0 + 1e-9), (gmaxy - gminy) / (double)(MAX_STRIPS - 2) * (1.0 + 1e-9));
    if (!(h > 0.0)) h = 1.0;
    nstrips = (int)fmin(floor((gmaxy - gminy) / h) + 1.0, (double)(MAX_STRIPS - 1));
    __syncthreads();
  }
  const double inv_h = 1.0 / h;
  const int n2 = icp_pow2(nM);
  for (int j = tid; j < n2; j += ICP_THREADS) {
    unsigned long long key = ~0ull;
    if (j < nM) {
      const unsigned long long st = (unsigned long long)fmin(fmax(floor((L.uy[j] - gminy) * inv_h), 0.0), (double)(nstrips - 1));
      const unsigned long long qx = (unsigned long long)fmin(fmax(floor((L.ux[j] - gminx) * QSCALE), 0.0), 1099511627775.0);
      key = (st << 52) | (qx << 12) | (unsigned long long)j;
    }
    L.keys[j] = key;
  }
  __syncthreads();
  // bitonic sort of the keys (unique: the original index is part of the key => deterministic order)
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (n2 >> 1); t += ICP_THREADS) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const unsigned long long ka = L.keys[i], kb = L.keys[l];
        const bool asc = (i & k) == 0;
        if ((ka > kb) == asc) { L.keys[i] = kb; L.keys[l] = ka; }
      }
      __syncthreads();
    }
  }
  for (int k = tid; k < nM; k += ICP_THREADS) {
    const int j = (int)(L.keys[k] & 0xFFFull);
    L.msx[k] = L.ux[j]; L.msy[k] = L.uy[j]; L.morig[k] = j;
    L.pos_of[j] = k;
  }
  // first slot of every strip: lower bound of (strip << 52) in the sorted keys
  for (int st = tid; st <= nstrips; st += ICP_THREADS) {
    const unsigned long long target = (unsigned long long)st << 52;
    int lo = 0, hi = nM;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.keys[mid] < target) lo = mid + 1; else hi = mid; }
    L.strip_start[st] = lo;
  }
  __syncthreads();
  STAMP(0);
  const int* __restrict__ pos_of = L.pos_of;
  int hint[PTS_PER_THREAD];
#pragma unroll
  for (int q = 0; q < PTS_PER_THREAD; q++) hint[q] = -1;

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
          double bd = __builtin_inf(); int bk = -1;
          auto eval = [&](int k) {
            const double dx = x - L.msx[k], dy = y - L.msy[k];
            const double d = dx * dx + dy * dy;
            if (d < bd) { bd = d; bk = k; }
            else if (d == bd && bk >= 0 && L.morig[k] < L.morig[bk]) { bk = k; }
          };
          // window walk inside one strip [sb, se), starting at slot `pos` (first slot at/after x)
          auto walk = [&](int sb, int se, int pos) {
            for (int k = pos; k < se; k++) {
              const double m = (L.msx[k] - x) - QMARGIN;
              if (m > 0.0 && m * m > fmin(bd, thr)) break;
              eval(k);
            }
            for (int k = pos - 1; k >= sb; k--) {
              const double m = (x - L.msx[k]) - QMARGIN;
              if (m > 0.0 && m * m > fmin(bd, thr)) break;
              eval(k);
            }
          };
          const double fy = floor((y - gminy) * inv_h);
          const int cy = (int)fmin(fmax(fy, -2.0), (double)(nstrips + 1));
          // own strip first: start from last iteration's neighbour when it lives here, which already
          // bounds the window to the current pair distance
          const int hk = hint[q];
          bool own_done = false;
          if (hk >= 0) {
            eval(hk);
            if (cy >= 0 && cy < nstrips && hk >= L.strip_start[cy] && hk < L.strip_start[cy + 1] && bd <= thr) {
              const int sb = L.strip_start[cy], se = L.strip_start[cy + 1];
              for (int k = hk + 1; k < se; k++) {
                const double m = (L.msx[k] - x) - QMARGIN;
                if (m > 0.0 && m * m > fmin(bd, thr)) break;
                eval(k);
              }
              for (int k = hk - 1; k >= sb; k--) {
                const double m = (x - L.msx[k]) - QMARGIN;
                if (m > 0.0 && m * m > fmin(bd, thr)) break;
                eval(k);
              }
              own_done = true;
            }
          }
#pragma unroll
          for (int ds = 0; ds < 3; ds++) {
            const int st = (ds == 0) ? cy : (ds == 1 ? cy - 1 : cy + 1);
            if (st < 0 || st >= nstrips || (ds == 0 && own_done)) continue;
            // distance from the point to the strip's band in y; skip bands out of reach
            const double ylo = gminy + (double)st * h, yhi = gminy + (double)(st + 1) * h;
            const double gap = fmax(fmax(ylo - y, y - yhi), 0.0) - 1e-9 * h;
            if (gap > 0.0 && gap * gap > fmin(bd, thr)) continue;
            const int sb = L.strip_start[st], se = L.strip_start[st + 1];
            int lo = sb, hi = se;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.msx[mid] < x) lo = mid + 1; else hi = mid; }
            walk(sb, se, lo);
          }
          my_k[q] = bk; my_d[q] = bd;
          my_keep[q] = (bk >= 0) && (bd <= thr);      // DistanceFilter::filter
          hint[q] = bk;
        }
      }
    }
    // threshold schedule (DistanceFilter.cpp:62-63)
    thr *= a.multiplier;
    if (thr < a.min_sqr) thr = a.min_sqr;
    __syncthreads();
    STAMP(1);
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

    STAMP(2);
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
    STAMP(3);

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
      STAMP(4);
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
    STAMP(5);
    if (tid == 0 && iter <= TSD_ICP_TRACE_MAX) {
      double* tr = trace + 4 * (iter - 1);
      tr[0] = (double)pairs; tr[1] = rms; tr[2] = thr_before; tr[3] = (double)state;
    }
    __syncthreads();
  }

#ifdef TSD_ICP_STAMPS
  if (tid == 0) for (int i = 0; i < 8; i++) trace[4 * TSD_ICP_TRACE_MAX - 8 + i] = (double)st_acc[i];
#endif
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
  const size_t lds = icp_lds_bytes_for(cap);
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

size_t icp_lds_bytes() { return icp_lds_bytes_for(TSD_MAX_ICP_POINTS); }

}  // namespace tsd
