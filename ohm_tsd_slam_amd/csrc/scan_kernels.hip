// scan_kernels.hip -- the host steps of ThreadLocalize::eventLoop that sit BETWEEN the device kernels,
// moved onto the device so that one scan is one H2D copy, one chain of kernels and one D2H copy with
// no host round trip in the middle (fused scan path, tsd_scan):
//
//   k_scan_post     after k_icp: isRegistrationError (ThreadLocalize.cpp:593-600), Sensor::transform
//                   (Sensor.cpp:50-60: rays = R * rays, pose = pose * T), isPoseChangeSignificant
//                   (:728-736) against the last pushed pose, and the pose-dependent arguments of this
//                   scan's push and of the next scan's ray cast / registration (pose^-1 as
//                   gsl_linalg_LU_decomp + LU_invert would give it, gsl/Matrix.cpp:168-179).
//   k_scan_prepare  the same argument derivation for a pose set from the host (tsd_sensor_set_pose).
//
// The arithmetic is the host facade's (csrc/host/obvision/obvious.cpp, ThreadLocalize.cpp) statement
// for statement, fp64, no contraction, so pose and rays are bit-identical to the host path; only the
// libm calls of the two gates (asin / acos / sin) are the device's.
#include "tsd_ctx.hpp"

namespace tsd {

// obvious::Matrix::operator* -> gsl_blas_dgemm(NoTrans, NoTrans): k ascending from 0.0
__device__ __forceinline__ void d_mat3_mul(const double A[9], const double B[9], double C[9])
{
  double R[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) t += A[3 * i + k] * B[3 * k + j];
      R[3 * i + j] = t;
    }
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = R[i];
}

// same elimination order as tsd::mat3_inv on the host (capi.hip)
__device__ void d_mat3_inv(const double A[9], double Ainv[9])
{
  double lu[9];
  int perm[3] = {0, 1, 2};
  for (int i = 0; i < 9; i++) lu[i] = A[i];
  for (int j = 0; j < 3; j++) {
    int piv = j;
    double best = fabs(lu[3 * j + j]);
    for (int i = j + 1; i < 3; i++)
      if (fabs(lu[3 * i + j]) > best) { best = fabs(lu[3 * i + j]); piv = i; }
    if (piv != j) {
      for (int k = 0; k < 3; k++) { const double t = lu[3 * j + k]; lu[3 * j + k] = lu[3 * piv + k]; lu[3 * piv + k] = t; }
      const int t = perm[j]; perm[j] = perm[piv]; perm[piv] = t;
    }
    for (int i = j + 1; i < 3; i++) {
      lu[3 * i + j] = lu[3 * i + j] / lu[3 * j + j];
      for (int k = j + 1; k < 3; k++) lu[3 * i + k] -= lu[3 * i + j] * lu[3 * j + k];
    }
  }
  for (int c = 0; c < 3; c++) {
    double x[3];
    for (int i = 0; i < 3; i++) x[i] = (perm[i] == c) ? 1.0 : 0.0;
    for (int i = 1; i < 3; i++)
      for (int k = 0; k < i; k++) x[i] -= lu[3 * i + k] * x[k];
    for (int i = 2; i >= 0; i--) {
      for (int k = i + 1; k < 3; k++) x[i] -= lu[3 * i + k] * x[k];
      x[i] = x[i] / lu[3 * i + i];
    }
    for (int i = 0; i < 3; i++) Ainv[3 * i + c] = x[i];
  }
}

// ThreadLocalize::calcAngle (ThreadLocalize.cpp:715-726)
__device__ __forceinline__ double d_calc_angle(const double T[9])
{
  double angle = 0.0;
  const double ARCSIN = asin(T[3]);
  const double ARCSINEG = asin(T[1]);
  const double ARCOS = acos(T[0]);
  if ((ARCSIN > 0.0) && (ARCSINEG < 0.0)) angle = ARCOS;
  else if ((ARCSIN < 0.0) && (ARCSINEG > 0.0)) angle = 2.0 * M_PI - ARCOS;
  return angle;
}

// pose dependent kernel arguments: pose^-1 for back projection / ray cast output, sensor position,
// RayCastPolar2D's "sensor inside the grid" defaults (RayCastPolar2D.cpp:128-146)
__device__ void d_derive_args(SensorDev* st, const GridDev& g)
{
  double Pi[9];
  d_mat3_inv(st->pose, Pi);
  for (int i = 0; i < 6; i++) { st->rc.Pi[i] = Pi[i]; st->push.Pi[i] = Pi[i]; st->icpP[i] = st->pose[i]; }
  const double trx = st->pose[2], try_ = st->pose[5];
  st->rc.trx = trx; st->rc.try_ = try_;
  st->push.trx = trx; st->push.try_ = try_;
  if (trx > g.min_x && trx < g.max_x && try_ > g.min_y && try_ < g.max_y) {
    st->rc.gxmin = -10e9; st->rc.gymin = -10e9; st->rc.gxmax = 10e9; st->rc.gymax = 10e9;
  } else {
    st->rc.gxmin = 10e9; st->rc.gymin = 10e9; st->rc.gxmax = -10e9; st->rc.gymax = -10e9;
  }
}

__global__ void k_scan_prepare(GridDev g, SensorDev* st)
{
  if (threadIdx.x == 0 && blockIdx.x == 0) d_derive_args(st, g);
}

__global__ void __launch_bounds__(256)
k_scan_post(GridDev g, SensorDev* st, double* __restrict__ rays, int beams, GateArgs gates,
            const IcpResultDev* __restrict__ icp, ScanResultDev* out, unsigned long long seq)
{
  double T[9];
#pragma unroll
  for (int i = 0; i < 9; i++) T[i] = icp->T[i];
  const bool no_model = icp->n_model == 0;   // "Raycasting found no coordinates" (ThreadLocalize.cpp:354-358)
  // isRegistrationError (every thread: the ray update below depends on it)
  bool reg_error = false;
  if (!no_model) {
    const double dX = T[2], dY = T[5];
    const double trns = sqrt(dX * dX + dY * dY);
    const double dphi = d_calc_angle(T);
    reg_error = (trns > gates.reg_trs_max) || (fabs(sin(dphi)) > gates.reg_sin_rot_max);
  }
  const bool moved = !no_model && !reg_error;
  if (moved) {
    // Sensor::transform: (*_rays) = R * (*_rays)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < beams; i += gridDim.x * blockDim.x) {
      const double x = rays[i], y = rays[beams + i];
      double nx = 0.0, ny = 0.0;
      nx += T[0] * x; nx += T[1] * y;
      ny += T[3] * x; ny += T[4] * y;
      rays[i] = nx; rays[beams + i] = ny;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // first scan after init: _lastPose = pose before the registration (ThreadLocalize.cpp:342-350)
    if (!st->have_last_pose) {
      for (int i = 0; i < 9; i++) st->last_pose[i] = st->pose[i];
      st->have_last_pose = 1;
    }
    int pushed = 0;
    if (moved) {
      double cur[9];
      d_mat3_mul(st->pose, T, cur);                 // _T = _T * T
      for (int i = 0; i < 9; i++) st->pose[i] = cur[i];
      // isPoseChangeSignificant(_lastPose, curPose)
      const double dX = cur[2] - st->last_pose[2], dY = cur[5] - st->last_pose[5];
      double dphi = d_calc_angle(cur) - d_calc_angle(st->last_pose);
      dphi = fabs(sin(dphi));
      const double trns = sqrt(dX * dX + dY * dY);
      if (dphi > gates.rot_min || trns > gates.trs_min) {
        pushed = 1;
        for (int i = 0; i < 9; i++) st->last_pose[i] = cur[i];
      }
      d_derive_args(st, g);
    }
    st->push.enabled = pushed;
    out->icp = *icp;
    for (int i = 0; i < 9; i++) out->pose[i] = st->pose[i];
    out->reg_error = reg_error ? 1 : 0; out->pushed = pushed; out->no_model = no_model ? 1 : 0; out->reserved = 0;
    // `out` is coherent host memory: publish the record, then the sequence number the host polls
    __threadfence_system();
    __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int launch_scan_prepare(tsd_ctx* ctx, SensorDev* st)
{
  hipLaunchKernelGGL(k_scan_prepare, dim3(1), dim3(64), 0, ctx->stream, ctx->grid, st);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_scan_post(tsd_ctx* ctx, SensorDev* st, double* d_rays, int beams, const GateArgs& gates,
                     ScanResultDev* d_out, unsigned long long seq)
{
  ScopedKernelTimer t(ctx, "scan_post");
  // one block: thread 0 owns the pose bookkeeping, all threads turn the rays
  hipLaunchKernelGGL(k_scan_post, dim3(1), dim3(256), 0, ctx->stream, ctx->grid, st, d_rays, beams, gates,
                     ctx->d_icp_res, d_out, seq);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

}  // namespace tsd
