// scan_device.hpp -- device functions of the fused scan path shared by k_icp's epilogue and k_scan_prepare:
// the host steps of ThreadLocalize::eventLoop that sit between the device kernels (isRegistrationError,
// Sensor::transform, isPoseChangeSignificant, pose^-1), statement for statement the host facade's
// arithmetic (csrc/host/obvision/obvious.cpp, ThreadLocalize.cpp), fp64, no contraction.
#pragma once
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
__device__ inline void d_mat3_inv(const double A[9], double Ainv[9])
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

// obvious::Matrix::solve (gsl/Matrix.cpp:343-355): gsl_linalg_LU_decomp (partial pivoting, first maximum) then
// gsl_linalg_LU_solve (x = P b, unit-lower forward substitution, upper back substitution); 3 x 3
__device__ inline void d_lu3_solve(const double A[9], const double b[3], double x[3])
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
  for (int i = 0; i < 3; i++) x[i] = b[perm[i]];
  for (int i = 1; i < 3; i++)
    for (int k = 0; k < i; k++) x[i] -= lu[3 * i + k] * x[k];
  for (int i = 2; i >= 0; i--) {
    for (int k = i + 1; k < 3; k++) x[i] -= lu[3 * i + k] * x[k];
    x[i] = x[i] / lu[3 * i + i];
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
__device__ inline void d_derive_args(SensorDev* st, double gmin_x, double gmax_x, double gmin_y, double gmax_y)
{
  double Pi[9];
  d_mat3_inv(st->pose, Pi);
  for (int i = 0; i < 6; i++) { st->rc.Pi[i] = Pi[i]; st->push.Pi[i] = Pi[i]; st->icpP[i] = st->pose[i]; }
  const double trx = st->pose[2], try_ = st->pose[5];
  st->rc.trx = trx; st->rc.try_ = try_;
  st->push.trx = trx; st->push.try_ = try_;
  if (trx > gmin_x && trx < gmax_x && try_ > gmin_y && try_ < gmax_y) {
    st->rc.gxmin = -10e9; st->rc.gymin = -10e9; st->rc.gxmax = 10e9; st->rc.gymax = 10e9;
  } else {
    st->rc.gxmin = 10e9; st->rc.gymin = 10e9; st->rc.gxmax = -10e9; st->rc.gymax = -10e9;
  }
}

// The steps of ThreadLocalize::eventLoop between the registration and the push, run by the whole
// workgroup that produced T (the epilogue of k_icp in the fused scan path): thread 0 owns the pose
// bookkeeping, all threads turn the rays.  T = Icp::getFinalTransformation(), n_model = ray-cast hits.
__device__ inline void scan_post_body(const ScanPostArgs& sp, const double T[9], const IcpResultDev& icp,
                                      double gmin_x, double gmax_x, double gmin_y, double gmax_y)
{
  SensorDev* st = sp.st;
  const bool no_model = icp.n_model == 0;   // "Raycasting found no coordinates" (ThreadLocalize.cpp:354-358)
  // isRegistrationError (every thread: the ray update below depends on it)
  bool reg_error = false;
  if (!no_model) {
    const double dX = T[2], dY = T[5];
    const double trns = sqrt(dX * dX + dY * dY);
    const double dphi = d_calc_angle(T);
    reg_error = (trns > sp.gates.reg_trs_max) || (fabs(sin(dphi)) > sp.gates.reg_sin_rot_max);
  }
  const bool moved = !no_model && !reg_error;
  if (moved) {
    // Sensor::transform: (*_rays) = R * (*_rays)
    double* rays = sp.rays;
    const int beams = sp.beams;
    for (int i = threadIdx.x; i < beams; i += blockDim.x) {
      const double x = rays[i], y = rays[beams + i];
      double nx = 0.0, ny = 0.0;
      nx += T[0] * x; nx += T[1] * y;
      ny += T[3] * x; ny += T[4] * y;
      rays[i] = nx; rays[beams + i] = ny;
    }
  }
  // every wave has turned its rays before thread 0 publishes the sequence numbers: a ray cast of this sensor on ANOTHER stream
  // (the batched path) is ordered behind this scan only through them
  __syncthreads();
  if (threadIdx.x == 0) {
    double pose[9], last[9];
    for (int i = 0; i < 9; i++) { pose[i] = st->pose[i]; last[i] = st->last_pose[i]; }
    // first scan after init: _lastPose = pose before the registration (ThreadLocalize.cpp:342-350)
    if (!st->have_last_pose) {
      for (int i = 0; i < 9; i++) { last[i] = pose[i]; st->last_pose[i] = pose[i]; }
      st->have_last_pose = 1;
    }
    int pushed = 0;
    if (moved) {
      double cur[9];
      d_mat3_mul(pose, T, cur);                     // _T = _T * T
      for (int i = 0; i < 9; i++) { pose[i] = cur[i]; st->pose[i] = cur[i]; }
      // isPoseChangeSignificant(_lastPose, curPose)
      const double dX = cur[2] - last[2], dY = cur[5] - last[5];
      double dphi = d_calc_angle(cur) - d_calc_angle(last);
      dphi = fabs(sin(dphi));
      const double trns = sqrt(dX * dX + dY * dY);
      if (dphi > sp.gates.rot_min || trns > sp.gates.trs_min) {
        pushed = 1;
        for (int i = 0; i < 9; i++) st->last_pose[i] = cur[i];
      }
      d_derive_args(st, gmin_x, gmax_x, gmin_y, gmax_y);
    }
    st->push.enabled = pushed;
    ScanResultDev* out = sp.out;
    out->icp = icp;
    for (int i = 0; i < 9; i++) out->pose[i] = pose[i];
    out->reg_error = reg_error ? 1 : 0; out->pushed = pushed; out->no_model = no_model ? 1 : 0; out->reserved = 0;
    // `out` is coherent host memory: publish the record, then the sequence number the host polls
    __threadfence_system();
    __hip_atomic_store(&out->seq, sp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // the same for a gate kernel on another stream (a batched robot's push starts when ITS registration is done)
    __hip_atomic_store(&st->done_seq, sp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// A batched registration that did not run (k_icp_batch: its ray casts never reported done, or the host abandoned the batch):
// no pose change, the push of this scan switched off, and a result record that says why -- tsd_batch_results turns a non-zero
// `reserved` into an error for the caller (the reference's contract: "failures are logged and the scan is skipped",
// ThreadLocalize.cpp:354-358, :381-387).  One thread.
__device__ inline void scan_post_failed(const ScanPostArgs& sp, int why)
{
  SensorDev* st = sp.st;
  st->push.enabled = 0;
  ScanResultDev* out = sp.out;
  IcpResultDev r;
  for (int i = 0; i < 9; i++) r.T[i] = (i % 4 == 0) ? 1.0 : 0.0;
  r.rms = 0.0; r.pairs = 0; r.iterations = 0; r.state = TSD_ICP_NOTMATCHABLE; r.n_model = 0; r.n_scene = 0; r.reserved = why;
  out->icp = r;
  for (int i = 0; i < 9; i++) out->pose[i] = st->pose[i];
  out->reg_error = 1; out->pushed = 0; out->no_model = 0; out->reserved = why;
  __threadfence_system();
  __hip_atomic_store(&out->seq, sp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&st->done_seq, sp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace tsd
