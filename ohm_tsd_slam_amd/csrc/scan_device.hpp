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

// 3 x 3 LU with partial pivoting (first maximum), rows in REGISTERS: the textbook loops index the rows and the permutation with the
// pivot found at run time, which the compiler can only do through scratch memory (private_segment 96 B per lane in every k_icp
// instantiation of round 2, a chain of dependent scratch round trips at the end of each registration).  Here the two possible row
// exchanges are conditional swaps of named values; the arithmetic and its order are those of gsl_linalg_LU_decomp.
struct Lu3 {
  double r0[3], r1[3], r2[3];       // rows of L\U after elimination (L below the diagonal)
  int p0, p1, p2;                   // row i of the factorisation is row p_i of A
};
__device__ __forceinline__ void lu3_swap(double (&a)[3], double (&b)[3], int& pa, int& pb, bool doit)
{
#pragma unroll
  for (int k = 0; k < 3; k++) { const double x = a[k], y = b[k]; a[k] = doit ? y : x; b[k] = doit ? x : y; }
  const int t = pa; pa = doit ? pb : pa; pb = doit ? t : pb;
}
__device__ __forceinline__ Lu3 lu3_decomp(const double A[9])
{
  Lu3 f;
#pragma unroll
  for (int k = 0; k < 3; k++) { f.r0[k] = A[k]; f.r1[k] = A[3 + k]; f.r2[k] = A[6 + k]; }
  f.p0 = 0; f.p1 = 1; f.p2 = 2;
  {  // column 0: pivot = first row of maximal |.|
    int piv = 0; double best = fabs(f.r0[0]);
    if (fabs(f.r1[0]) > best) { best = fabs(f.r1[0]); piv = 1; }
    if (fabs(f.r2[0]) > best) { piv = 2; }
    lu3_swap(f.r0, f.r1, f.p0, f.p1, piv == 1);
    lu3_swap(f.r0, f.r2, f.p0, f.p2, piv == 2);
    f.r1[0] = f.r1[0] / f.r0[0]; f.r1[1] -= f.r1[0] * f.r0[1]; f.r1[2] -= f.r1[0] * f.r0[2];
    f.r2[0] = f.r2[0] / f.r0[0]; f.r2[1] -= f.r2[0] * f.r0[1]; f.r2[2] -= f.r2[0] * f.r0[2];
  }
  {  // column 1
    lu3_swap(f.r1, f.r2, f.p1, f.p2, fabs(f.r2[1]) > fabs(f.r1[1]));
    f.r2[1] = f.r2[1] / f.r1[1]; f.r2[2] -= f.r2[1] * f.r1[2];
  }
  return f;
}
// x = A^-1 b for b already permuted (b0 = b[p0] ...): unit-lower forward substitution, upper back substitution
__device__ __forceinline__ void lu3_solve_permuted(const Lu3& f, double b0, double b1, double b2, double& x0, double& x1, double& x2)
{
  x0 = b0; x1 = b1; x2 = b2;
  x1 -= f.r1[0] * x0;
  x2 -= f.r2[0] * x0; x2 -= f.r2[1] * x1;
  x2 = x2 / f.r2[2];
  x1 -= f.r1[2] * x2; x1 = x1 / f.r1[1];
  x0 -= f.r0[1] * x1; x0 -= f.r0[2] * x2; x0 = x0 / f.r0[0];
}

// same elimination order as tsd::mat3_inv on the host (capi.hip): obvious::Matrix::invert (gsl/Matrix.cpp:168-179)
__device__ inline void d_mat3_inv(const double A[9], double Ainv[9])
{
  const Lu3 f = lu3_decomp(A);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    double x0, x1, x2;
    lu3_solve_permuted(f, f.p0 == c ? 1.0 : 0.0, f.p1 == c ? 1.0 : 0.0, f.p2 == c ? 1.0 : 0.0, x0, x1, x2);
    Ainv[c] = x0; Ainv[3 + c] = x1; Ainv[6 + c] = x2;
  }
}

// obvious::Matrix::solve (gsl/Matrix.cpp:343-355): gsl_linalg_LU_decomp (partial pivoting, first maximum) then
// gsl_linalg_LU_solve (x = P b, unit-lower forward substitution, upper back substitution); 3 x 3
__device__ inline void d_lu3_solve(const double A[9], const double b[3], double x[3])
{
  const Lu3 f = lu3_decomp(A);
  const double b0 = f.p0 == 0 ? b[0] : (f.p0 == 1 ? b[1] : b[2]);
  const double b1 = f.p1 == 0 ? b[0] : (f.p1 == 1 ? b[1] : b[2]);
  const double b2 = f.p2 == 0 ? b[0] : (f.p2 == 1 ? b[1] : b[2]);
  lu3_solve_permuted(f, b0, b1, b2, x[0], x[1], x[2]);
}

// ThreadLocalize::calcAngle (ThreadLocalize.cpp:715-726)
__device__ __forceinline__ double d_calc_angle(const double T[9])
{
  // The reference takes asin(T[3]) and asin(T[1]) and looks at their SIGNS only: asin(x) > 0 <=> 0 < x <= 1, asin(x) < 0 <=> -1 <= x < 0
  // (NaN beyond +-1 compares false either way).  The two libm calls were two thirds of the pose bookkeeping's chain in the epilogue.
  double angle = 0.0;
  const double t3 = T[3], t1 = T[1];
  const bool sin_pos = t3 > 0.0 && t3 <= 1.0, sin_neg = t3 < 0.0 && t3 >= -1.0;
  const bool neg_pos = t1 > 0.0 && t1 <= 1.0, neg_neg = t1 < 0.0 && t1 >= -1.0;
  if (sin_pos && neg_neg) angle = acos(T[0]);
  else if (sin_neg && neg_pos) angle = 2.0 * M_PI - acos(T[0]);
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

// What the epilogue needs of the sensor's state, read at the START of the registration kernel (its loads then travel with the
// kernel's input loads) and parked in LDS: the epilogue itself begins without a trip to memory.
struct ScanPostPre {
  double pose[9], last[9];
  double last_angle;          // calcAngle(_lastPose), left by the scan that set _lastPose
  int have_last, pad;
};
__device__ inline void scan_post_preload(const ScanPostArgs& sp, ScanPostPre* pre /* LDS */)
{
  const SensorDev* st = sp.st;
  for (int i = 0; i < 9; i++) { pre->pose[i] = st->pose[i]; pre->last[i] = st->last_pose[i]; }
  pre->last_angle = st->last_angle; pre->have_last = st->have_last_pose;
}

// The steps of ThreadLocalize::eventLoop between the registration and the push, run by the whole
// workgroup that produced T (the epilogue of k_icp in the fused scan path).  T = Icp::getFinalTransformation(), n_model = ray-cast
// hits.  Three things run SIDE BY SIDE once every thread knows whether the pose moves (isRegistrationError): the rays are turned by
// all waves but the last; lane 0 of the last wave does the pose bookkeeping and the push gate (isPoseChangeSignificant: libm asin /
// acos / sin chains); lane 0 of the last-but-one wave inverts the new pose for the next ray cast and this scan's push (an LU with
// three dependent divisions).  Round 2 ran these one after the other on thread 0, behind a read of the sensor state from memory.
__device__ inline void scan_post_body(const ScanPostArgs& sp, const ScanPostPre& pre, const double T[9], const IcpResultDev& icp,
                                      double gmin_x, double gmax_x, double gmin_y, double gmax_y, long long* tk = nullptr /* timeline build: stamps */)
{
  SensorDev* st = sp.st;
#ifdef TSD_ICP_TIMELINE
#define EPI_STAMP(i) do { if (tk && threadIdx.x == 0) tk[i] = clock64(); } while (0)
#else
#define EPI_STAMP(i) do {} while (0)
#endif
  EPI_STAMP(0);
  const int W = (int)(blockDim.x >> 6), wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const bool no_model = icp.n_model == 0;   // "Raycasting found no coordinates" (ThreadLocalize.cpp:354-358)
  // isRegistrationError (every thread: the ray update below depends on it)
  bool reg_error = false;
  if (!no_model) {
    const double dX = T[2], dY = T[5];
    const double trns = sqrt(dX * dX + dY * dY);
    // |sin(calcAngle(T))| is 0 or |sin(acos(T0))| = sqrt((1 - T0)(1 + T0)) (calcAngle returns 0, acos(T0) or 2 pi - acos(T0),
    // ThreadLocalize.cpp:715-726).  The libm chain (asin, asin, acos, sin: ~2 000 cycles in front of everything the epilogue does) is
    // only needed to DECIDE when that closed form lies within 1e-9 of the gate; otherwise both sides of the comparison agree.
    const double t0 = T[0], t3 = T[3], t1 = T[1];
    const bool arg_ok = fabs(t0) <= 1.0 && fabs(t3) <= 1.0 && fabs(t1) <= 1.0;
    const bool turned = ((t3 > 0.0) && (t1 < 0.0)) || ((t3 < 0.0) && (t1 > 0.0));      // (asin keeps the sign)
    const double s_fast = turned ? sqrt((1.0 - t0) * (1.0 + t0)) : 0.0;
    const double g = sp.gates.reg_sin_rot_max;
    bool rot_error;
    if (arg_ok && fabs(s_fast - g) > 1e-9 * fmax(fabs(g), s_fast)) rot_error = s_fast > g;
    else rot_error = fabs(sin(d_calc_angle(T))) > g;
    reg_error = (trns > sp.gates.reg_trs_max) || rot_error;
  }
  const bool moved = !no_model && !reg_error;
  EPI_STAMP(1);                              // gate decided
  // workgroup-shared hand-over of the two lone lanes' results (64-byte aligned scratch in the model's LDS would do as well;
  // static: a few words)
  __shared__ int s_pushed;
  __shared__ double s_pose[9];
  const bool lone_gate = wave == W - 1 && lane == 0;
  const bool lone_inv = W >= 2 ? (wave == W - 2 && lane == 0) : lone_gate;
  if (moved && !(W >= 3 && wave >= W - 2)) {
    // Sensor::transform: (*_rays) = R * (*_rays)   (the last two waves sit this out when there are others)
    double* rays = sp.rays;
    const int beams = sp.beams;
    const int nthr = W >= 3 ? (W - 2) * 64 : (int)blockDim.x;
    // (three beams per thread and round with all six reads in flight: one trip to memory where the one-beam loop took three)
    constexpr int U = 3;
    for (int i0 = threadIdx.x; i0 < beams; i0 += U * nthr) {
      double x[U], y[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int i = i0 + u * nthr;
        x[u] = i < beams ? rays[i] : 0.0; y[u] = i < beams ? rays[beams + i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int i = i0 + u * nthr;
        if (i >= beams) continue;
        double nx = 0.0, ny = 0.0;
        nx += T[0] * x[u]; nx += T[1] * y[u];
        ny += T[3] * x[u]; ny += T[4] * y[u];
        rays[i] = nx; rays[beams + i] = ny;
      }
    }
  }
  double cur[9];
  for (int i = 0; i < 9; i++) cur[i] = pre.pose[i];
  if (moved && (lone_gate || lone_inv)) d_mat3_mul(pre.pose, T, cur);          // _T = _T * T
  if (lone_inv) {
    // pose dependent kernel arguments (d_derive_args): pose^-1 for back projection / ray cast output, sensor position,
    // RayCastPolar2D's "sensor inside the grid" defaults (RayCastPolar2D.cpp:128-146)
    if (moved) {
      double Pi[9];
      d_mat3_inv(cur, Pi);
      for (int i = 0; i < 6; i++) { st->rc.Pi[i] = Pi[i]; st->push.Pi[i] = Pi[i]; st->icpP[i] = cur[i]; }
      const double trx = cur[2], try_ = cur[5];
      st->rc.trx = trx; st->rc.try_ = try_;
      st->push.trx = trx; st->push.try_ = try_;
      if (trx > gmin_x && trx < gmax_x && try_ > gmin_y && try_ < gmax_y) {
        st->rc.gxmin = -10e9; st->rc.gymin = -10e9; st->rc.gxmax = 10e9; st->rc.gymax = 10e9;
      } else {
        st->rc.gxmin = 10e9; st->rc.gymin = 10e9; st->rc.gxmax = -10e9; st->rc.gymax = -10e9;
      }
    }
  }
  if (lone_gate) {
    // first scan after init: _lastPose = pose before the registration (ThreadLocalize.cpp:342-350)
    double last_angle = pre.last_angle;
    double lastX = pre.last[2], lastY = pre.last[5];
    if (!pre.have_last) {
      for (int i = 0; i < 9; i++) st->last_pose[i] = pre.pose[i];
      st->have_last_pose = 1;
      last_angle = d_calc_angle(pre.pose); lastX = pre.pose[2]; lastY = pre.pose[5];
      st->last_angle = last_angle;
    }
    int pushed = 0;
    if (moved) {
      for (int i = 0; i < 9; i++) st->pose[i] = cur[i];
      // isPoseChangeSignificant(_lastPose, curPose)
      const double dX = cur[2] - lastX, dY = cur[5] - lastY;
      const double cur_angle = d_calc_angle(cur);
      double dphi = cur_angle - last_angle;
      dphi = fabs(sin(dphi));
      const double trns = sqrt(dX * dX + dY * dY);
      if (dphi > sp.gates.rot_min || trns > sp.gates.trs_min) {
        pushed = 1;
        for (int i = 0; i < 9; i++) st->last_pose[i] = cur[i];
        st->last_angle = cur_angle;
      }
    }
    st->push.enabled = pushed;
    s_pushed = pushed;
    for (int i = 0; i < 9; i++) s_pose[i] = cur[i];
  }
  // every wave has turned its rays and both lone lanes have written the sensor's state before the sequence numbers go out: a ray
  // cast of this sensor on ANOTHER stream (the batched path) is ordered behind this scan only through them
  // The result record leaves as TAGGED words: every 8-byte word = {low half of the scan's sequence number, four bytes of the record},
  // all of them ONE store instruction of relaxed system-scope atomics (write-through stores into the coherent pinned buffer).  The host
  // has the record when every word carries the tag -- no order among the stores is needed, so nothing is drained in between (the
  // record-then-sequence-number form waited one trip over the host link between the two: ~1 700 cycles at the end of every
  // registration; round 4's fence + release form ~3 us).  Its parts are written into LDS by the lanes that hold them, ahead of the
  // barrier: the registration's result by thread 0, pose and flags by the gate lane.
  __shared__ unsigned int s_record[SCAN_RESULT_WORDS];
  if (threadIdx.x == 0) {
    ScanResultDev r;
    r.icp = icp;
    const unsigned int* w = reinterpret_cast<const unsigned int*>(&r);
    for (int i = 0; i < (int)(offsetof(ScanResultDev, pose) / 4); i++) s_record[i] = w[i];
  }
  if (lone_gate) {
    ScanResultDev r;
    for (int i = 0; i < 9; i++) r.pose[i] = s_pose[i];       // (this lane's own writes)
    r.reg_error = reg_error ? 1 : 0; r.pushed = s_pushed; r.no_model = no_model ? 1 : 0; r.reserved = 0;
    r.seq = sp.seq;
    const unsigned int* w = reinterpret_cast<const unsigned int*>(&r);
    for (int i = (int)(offsetof(ScanResultDev, pose) / 4); i < SCAN_RESULT_WORDS; i++) s_record[i] = w[i];
  }
  EPI_STAMP(2);                              // wave 0's rays turned (stores issued)
  // every wave has turned its rays and both lone lanes have written the sensor's state before the sequence numbers go out: a ray
  // cast of this sensor on ANOTHER stream (the batched path) is ordered behind this scan only through them
  __syncthreads();
  EPI_STAMP(3);                              // everybody's, and the lone lanes' bookkeeping
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0 && sp.push_copy) *sp.push_copy = st->push;      // (asynchronous mapping: this scan's push reads its own copy; a later kernel)
    if ((int)threadIdx.x < SCAN_RESULT_WORDS)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(sp.out) + threadIdx.x,
                         ((unsigned long long)(unsigned int)sp.seq << 32) | (unsigned long long)s_record[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    EPI_STAMP(4);                            // record on its way
    // the same number for a gate kernel on another stream (a batched robot's push starts when ITS registration is done): that
    // reader takes the sensor's state, written with plain stores above -- a release, where there is such a reader
    if (threadIdx.x == 0 && sp.publish_done) __hip_atomic_store(&st->done_seq, sp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    EPI_STAMP(5);
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
  if (sp.push_copy) *sp.push_copy = st->push;
  ScanResultDev rec;
  IcpResultDev r;
  for (int i = 0; i < 9; i++) r.T[i] = (i % 4 == 0) ? 1.0 : 0.0;
  r.rms = 0.0; r.pairs = 0; r.iterations = 0; r.state = TSD_ICP_NOTMATCHABLE; r.n_model = 0; r.n_scene = 0; r.reserved = why;
  rec.icp = r;
  for (int i = 0; i < 9; i++) rec.pose[i] = st->pose[i];
  rec.reg_error = 1; rec.pushed = 0; rec.no_model = 0; rec.reserved = why;
  rec.seq = sp.seq;
  const unsigned int* w = reinterpret_cast<const unsigned int*>(&rec);
  for (int i = 0; i < SCAN_RESULT_WORDS; i++)           // (tagged words, see scan_post_body)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(sp.out) + i, ((unsigned long long)(unsigned int)sp.seq << 32) | (unsigned long long)w[i],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&st->done_seq, sp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace tsd
