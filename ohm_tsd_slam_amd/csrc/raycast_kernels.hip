// raycast_kernels.hip -- RayCastPolar2D::calcCoordsFromCurrentViewMask (RayCastPolar2D.cpp:113-192)
// and rayCastFromCurrentView (:194-281) for gfx950.
//
// One 64-lane wavefront per beam.  The reference marches a beam one cell at a time with
// `position += ray` (REPEATED fp64 addition, :246-247) and looks for the first sign change of the
// bilinearly interpolated TSD.  The event at step k only depends on the samples of steps k-1 and k,
// so 64 consecutive steps are sampled by the 64 lanes at once and the first event is found with a
// ballot.  To stay bit-identical with the reference's accumulated rounding, the positions themselves
// are produced by the same chain of additions, run redundantly (wave-uniform) by all lanes; each
// lane keeps the value of its own step.  The coarse 32-cell skip loop (:225-236) is handled the same
// way.  The four normal look-ups of TsdGrid::interpolateNormal (TsdGrid.cpp:517-546) run on 4 lanes.
//
// Latency-bound gather (L2 / Infinity-Cache resident tiles): reported as time, not as a roofline.
#include "tsd_ctx.hpp"

namespace tsd {

__global__ void __launch_bounds__(64)
k_raycast(GridDev g, RaycastArgs a_val, const RaycastArgs* __restrict__ a_dev, const double* __restrict__ rays,
          double* __restrict__ coords, double* __restrict__ normals, uint8_t* __restrict__ mask)
{
  const RaycastArgs a = a_dev ? *a_dev : a_val;
  const int beam = blockIdx.x;
  const int lane = threadIdx.x;
  if (beam >= a.beams) return;
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
    double i_run = idxMin;          // wave-uniform loop variable of the reference
    bool done = false;
    while (!done) {
      double my_i = 0.0; bool my_act = false;
#pragma unroll 8
      for (int s = 0; s < 64; s++) {
        const bool act = i_run < idxMax;
        if (lane == s) { my_i = i_run; my_act = act; }
        i_run += 32.0;
      }
      bool ok = false;
      if (my_act) {
        double tmp;
        const int rv = interpolate_bilinear(g, trx + my_i * rx, try_ + my_i * ry, tmp);
        ok = (rv != INTERP_EMPTYPARTITION && rv != INTERP_INVALIDINDEX);
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

  // fine march
  double px = trx + idxMin * rx, py = try_ + idxMin * ry;   // wave-uniform running position
  double carry;                                             // sample of the previous step (NaN = none)
  {
    double t0;
    carry = (interpolate_bilinear(g, px, py, t0) == INTERP_SUCCESS) ? t0 : __builtin_nan("");
  }
  double i_run = idxMin;
  bool found = false;
  double hit_x = 0.0, hit_y = 0.0;
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
      if (interpolate_bilinear(g, mx, my, t) == INTERP_SUCCESS) cur = t;
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
        // interp = tsd_prev / (tsd_prev - tsd); c = position + ray * (interp - 1)
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
  if (!found) { if (lane == 0) mask[beam] = 0; return; }

  // TsdGrid::interpolateNormal: lanes 0..3 sample (x+cs,y) (x-cs,y) (x,y+cs) (x,y-cs)
  double sx = hit_x, sy = hit_y;
  if (lane == 0) sx = hit_x + cs;
  else if (lane == 1) sx = hit_x - cs;
  else if (lane == 2) sy = hit_y + cs;
  else if (lane == 3) sy = hit_y - cs;
  double v = 0.0; bool okn = true;
  if (lane < 4) okn = interpolate_bilinear(g, sx, sy, v) == INTERP_SUCCESS;
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
}

int launch_raycast(tsd_ctx* ctx, const RaycastArgs& a, const RaycastArgs* a_dev, const double* d_rays)
{
  ScopedKernelTimer t(ctx, "raycast");
  hipLaunchKernelGGL(k_raycast, dim3(a.beams), dim3(64), 0, ctx->stream, ctx->grid, a, a_dev, d_rays ? d_rays : ctx->d_rays,
                     ctx->d_coords, ctx->d_normals, ctx->d_mask_m);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

}  // namespace tsd
