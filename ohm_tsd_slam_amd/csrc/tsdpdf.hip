// tsdpdf.hip -- SURVEY 8(f) row N3: the TSD_PDF pre-registration ThreadLocalize runs before the ICP in
// registration_mode 3 (ThreadLocalize.cpp:557-567; config/single-laser.yaml:28 ships this mode):
// obvious::TSD_PDFMatching::match (registration/ransacMatching/TSD_PDFMatching.cpp:31-294) on top of
// obvious::RandomMatching (RandomMatching.cpp:41-189).
//
// What it computes.  For `trials` randomly picked model points m_idx (ray-cast hits with a PCA normal) and every
// scene point s_i within +-span beams of idx that has a normal: the rigid motion T(idx, i) that turns s_i's normal
// onto m_idx's and moves s_i onto m_idx; a control set of <= sizeControlSet scene points is carried through
// TSensor * T into the map and scored with the product of 1 - (1 - zrand) |tsd| over the control points (zrand where
// the bilinear look-up fails); the candidate with the highest product wins.
//
// Where the work goes.  Normals (a 2 x 2 PCA over <= 10 neighbours per point), masks, the control set and the
// candidate list are O(beams) host work, serial in the reference too.  The scoring -- candidates x control points
// bilinear look-ups into the TSD grid -- is the data-parallel part and runs here on the device: one lane per
// candidate, the control set in LDS, the look-ups of a candidate issued in batches and multiplied IN THE REFERENCE'S
// ORDER (s = 0 .. C-1), because the winner is an arg-max over floating-point products; then one workgroup reduces
// to the first candidate (in the reference's serial trial / i order) that reaches the maximum, which is what the
// strict `>` of TSD_PDFMatching.cpp:264 gives a serial run.
//
// Randomness.  The reference draws from rand() in RandomMatching::subsampleMask, RandomMatching::pickControlSet and
// for the trial pick (after srand(time(NULL)), inside an OpenMP loop): not reproducible.  The C ABI takes the three
// streams of raw rand() values as inputs; the C++ facade (csrc/host/obvision) draws them with rand() where the
// reference does.  With given draws the result is a function of the inputs and is parity-tested against the oracle's
// restatement (PARITY UNPINNED: the reference's translation unit needs GSL and cannot be compiled in this image).
#include "tsd_ctx.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "tsdpdf_device.hpp"

namespace tsd {

constexpr int PDF_MAX_CONTROL = 1024;      // control points held in LDS (16 KB)
constexpr int PDF_BATCH = 4;               // look-ups per LANE in flight together (64 x 4 control points per round)

// One WAVE per candidate: probability of the control set under T(idx, i).  Lane l takes the control points l, l + 64, ...: their
// bilinear look-ups (tile flag and the four cells issued together: the tile storage exists for every tile) all run side by side,
// the factors meet in LDS and are multiplied IN THE REFERENCE'S ORDER (s = 0 .. C-1: the winner is an arg-max over floating-point
// products) by every lane alike.  (Rounds 2-3 ran one LANE per candidate: 1 300 candidates = 20 waves, each a chain of C / 8 memory
// round trips -- 164 us for 180 k look-ups.)
constexpr int PDF_WAVES = 4;               // candidates per workgroup and round
constexpr int PDF_SCORE_GRID = 1024;       // workgroups of the fused scan's scoring launch (4 096 waves stride over the candidates)
__device__ __forceinline__ void
pdf_score_body(const GridDev& g, const double* __restrict__ pose /* first two rows suffice */, const double* __restrict__ M, const double* __restrict__ S,
               const double2* __restrict__ control, int n_control, const PdfCandidate* __restrict__ cand, int n_cand,
               double zrand, double* __restrict__ prob_out, const PdfHeader* __restrict__ hdr /* counts from the device, or nullptr */,
               int max_cand_alloc, int control_alloc /* entries the two arrays are allocated to (speculative reads) */)
{
  __shared__ double s_f[PDF_WAVES][PDF_MAX_CONTROL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // The header in one go; a fixed grid whose waves stride over the candidates (the fused scan does not know their number on the host:
  // launching one wave for each of the max_cand possible ones -- 24 000 for 1 300 real ones -- cost more than the scoring); then the
  // candidate and the first batch of control points together; the scene / model points follow the candidate, the look-ups the control
  // points.  As written first -- header field by field, then the candidate, then per batch control point -> wait -> look-ups -- the
  // kernel was a chain of ten memory round trips for 140 control points.
  // (round 6: the wave's FIRST candidate and the first batch of control points are requested ahead of the header -- which candidate that
  // is does not depend on the counts, and both arrays are allocated beyond them -- so the header's round trip is theirs too)
  const int c_first = blockIdx.x * PDF_WAVES + wave;
  int idx_rd, ti_rd; double phi_rd;
  double cpx[PDF_BATCH], cpy[PDF_BATCH];
  auto request = [&](int c) {
    const int c_rd = c < max_cand_alloc ? c : 0;
    idx_rd = ld_pinned(&cand[c_rd].idx); ti_rd = ld_pinned(&cand[c_rd].ti); phi_rd = ld_pinned(&cand[c_rd].phi);
#pragma unroll
    for (int b = 0; b < PDF_BATCH; b++) {
      const int s = 64 * b + lane, sc = s < control_alloc ? s : 0;
      cpx[b] = ld_pinned(&control[sc].x); cpy[b] = ld_pinned(&control[sc].y);
    }
  };
  request(c_first);
  if (hdr) {
    const int4 h0 = reinterpret_cast<const int4*>(hdr)[0];        // n_cand, n_control, n_model_valid, n_scene_valid
    const int ident = hdr->identity;
    n_cand = ident ? 0 : h0.x; n_control = h0.y;
  }
  for (int c = c_first; c < n_cand; c += (int)gridDim.x * PDF_WAVES) {       // (whole waves: no barrier below)
  if (c != c_first) request(c);
  PdfCandidate cd; cd.idx = idx_rd; cd.ti = ti_rd; cd.phi = phi_rd;
  // T = MatrixFactory::TransformationMatrix33(phi, 0, 0) + translation (TSD_PDFMatching.cpp:217-223)
  const double co = cos(cd.phi), si = sin(cd.phi);
  const int ci = cd.ti & PDF_I_MASK;
  const double sx = S[2 * ci], sy = S[2 * ci + 1];
  const double T02 = M[2 * cd.idx] - (co * sx + (-si) * sy);
  const double T12 = M[2 * cd.idx + 1] - (si * sx + co * sy);
  // TMap = TSensor * T (3 x 3 dgemm: k ascending from 0.0)
  const double T[9] = {co, -si, T02, si, co, T12, 0.0, 0.0, 1.0};
  double TM[6];
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int q = 0; q < 3; q++) {
      double t = 0.0;
      t += pose[3 * r] * T[q]; t += pose[3 * r + 1] * T[3 + q]; t += pose[3 * r + 2] * T[6 + q];
      TM[3 * r + q] = t;
    }
  for (int s0 = 0; s0 < n_control; s0 += 64 * PDF_BATCH) {
    // up to PDF_BATCH look-ups per lane, all their reads in flight together
    uint8_t fl[PDF_BATCH]; Quad qv[PDF_BATCH]; double wx[PDF_BATCH], wy[PDF_BATCH]; bool inside[PDF_BATCH];
    if (s0 > 0) {                                             // (more than 64 * PDF_BATCH control points: the next batch, again ahead of its look-ups)
#pragma unroll
      for (int b = 0; b < PDF_BATCH; b++) {
        const int s = s0 + 64 * b + lane, sc = s < n_control ? s : 0;
        cpx[b] = ld_pinned(&control[sc].x); cpy[b] = ld_pinned(&control[sc].y);
      }
    }
#pragma unroll
    for (int b = 0; b < PDF_BATCH; b++) {
      fl[b] = 0; inside[b] = false; wx[b] = 0.0; wy[b] = 0.0; qv[b].t00 = qv[b].t01 = qv[b].t10 = qv[b].t11 = 0.0;
      if (s0 + 64 * b >= n_control) continue;                // (wave-uniform)
      const double2 cp = make_double2(cpx[b], cpy[b]);
      // STemp = TMap * Control, Control column = (x, y, 1)
      double cx = 0.0, cy = 0.0;
      cx += TM[0] * cp.x; cx += TM[1] * cp.y; cx += TM[2] * 1.0;
      cy += TM[3] * cp.x; cy += TM[4] * cp.y; cy += TM[5] * 1.0;
      int p = 0, lx = 0, ly = 0; double dx = 0.0, dy = 0.0;
      inside[b] = coord2cell(g, cx, cy, p, lx, ly, dx, dy);
      if (!inside[b]) { p = 0; lx = 0; ly = 0; }
      fl[b] = ld_pinned(&g.flags[p]);
      qv[b] = load_quad(g.tsd + (size_t)p * TILE_STRIDE, lx, ly);
      wx[b] = fabs((cx - dx) * g.inv_cs); wy[b] = fabs((cy - dy) * g.inv_cs);
    }
#pragma unroll
    for (int b = 0; b < PDF_BATCH; b++) {
      const int s = s0 + 64 * b + lane;
      if (s >= n_control) continue;
      // TsdGrid::interpolateBilinear (TsdGrid.h:284-304): !interpolateBilinear(...) <=> INTERPOLATE_SUCCESS (TsdGrid.h:28): clipped
      // probability, else zrand (TSD_PDFMatching.cpp:244-254)
      const double tsd = qv[b].t00 * (1. - wy[b]) * (1. - wx[b]) + qv[b].t10 * wy[b] * (1. - wx[b])
                       + qv[b].t01 * (1. - wy[b]) * wx[b] + qv[b].t11 * wy[b] * wx[b];
      const bool ok = inside[b] && fl[b] != 0 && !isnan(tsd);
      s_f[wave][s] = ok ? (1.0 - (1.0 - zrand) * fabs(tsd)) : zrand;
    }
  }
  // (the wave's own LDS writes are visible to its own reads: LDS executes a wave's accesses in order)
  double prob = 1.0;
  for (int s = 0; s < n_control; s++) prob *= s_f[wave][s];        // the reference's order
  if (lane == 0) prob_out[c] = prob;
  }
}

__global__ void __launch_bounds__(64 * PDF_WAVES)
k_pdf_score(GridDev g, const double* __restrict__ pose, const double* __restrict__ M, const double* __restrict__ S,
            const double2* __restrict__ control, int n_control, const PdfCandidate* __restrict__ cand, int n_cand,
            double zrand, double* __restrict__ prob_out, const PdfHeader* __restrict__ hdr, int max_cand_alloc, int control_alloc)
{
  pdf_score_body(g, pose, M, S, control, n_control, cand, n_cand, zrand, prob_out, hdr, max_cand_alloc, control_alloc);
}

// The pre-registrations of the armed robots of a BATCH (tsd_batch_begin, registration_mode 3 with several robots on one grid): the same
// four kernels, one launch each for all robots -- block z (normals, scoring) / block x (list building, arg-max) = robot, its arguments
// entry z / x of a by-value array.  Round 4 launched the four kernels per robot: 32 small dependent launches per batch of eight on the
// grid's stream, 11.5 k scans/s where registration_mode 0 has 29 k.
constexpr int PDF_BATCH_BYVAL = 16;        // robots per batched launch (the entries travel as kernel arguments: <= 2 KB)
struct PdfScoreEntry {
  const double* pose; const double* M; const double* S; const double2* control; const PdfCandidate* cand; double* prob;
  const PdfHeader* hdr; double zrand; int max_cand, control_alloc;
};
struct PdfScoreBatch { PdfScoreEntry e[PDF_BATCH_BYVAL]; };
__global__ void __launch_bounds__(64 * PDF_WAVES)
k_pdf_score_batch(GridDev g, PdfScoreBatch b)
{
  const PdfScoreEntry& e = b.e[blockIdx.z];
  pdf_score_body(g, e.pose, e.M, e.S, e.control, 0, e.cand, 0, e.zrand, e.prob, e.hdr, e.max_cand, e.control_alloc);
}

__global__ void __launch_bounds__(1024)
k_pdf_argmax(const double* __restrict__ prob, const PdfCandidate* __restrict__ cand, int n_cand, const double* __restrict__ M,
             const double* __restrict__ S, PdfResult* __restrict__ out, const PdfHeader* __restrict__ hdr,
             PdfHeader* __restrict__ host_hdr, PdfResult* __restrict__ host_res)
{
  pdf_argmax_body(prob, cand, n_cand, M, S, out, hdr, host_hdr, host_res);
}
struct PdfArgmaxBatch { PdfArgmaxEntry e[PDF_BATCH_BYVAL]; };
__global__ void __launch_bounds__(1024)
k_pdf_argmax_batch(PdfArgmaxBatch b)
{
  const PdfArgmaxEntry& e = b.e[blockIdx.x];
  pdf_argmax_body(e.prob, e.cand, e.max_cand, e.M, e.S, e.out, e.hdr, e.host_hdr, e.host_res);
}

// ---- RandomMatching::calcNormals + calcPhi on the device (second half of round 3) ---------------------------------------------
// One thread per point: Matrix::pcaAnalysis over its <= 10 masked-in neighbours, the axis-ratio test, the normal's sign, its angle.
// The host restatement below (kept: it is what the oracle mirrors statement by statement) spends ~140 us per 1 081-point set, serial;
// this is the same arithmetic in the same order with two exceptions, both below 1e-15 relative: gsl_stats_mean's running mean in
// `long double` (64-bit significand, no device type) is carried in double-double (106 bits: at least as accurate, rounds to the same
// double unless the x87 chain's own rounding error crosses a rounding boundary, ~0.5 % of the means, 1 ulp then), and atan2 / cos / sin
// are the device library's instead of glibc's.  Discrete outcomes (masks) can differ only for a point whose axis ratio sits within
// ~1e-15 of its threshold.
#define PDF_LIKELY(x) __builtin_expect(!!(x), 1)
#define PDF_UNLIKELY(x) __builtin_expect(!!(x), 0)
struct DD { double hi, lo; };
__device__ __forceinline__ DD dd_quick_two_sum(double a, double b) { const double s = a + b; return DD{s, b - (s - a)}; }
__device__ __forceinline__ DD dd_two_sum(double a, double b) { const double s = a + b, bb = s - a; return DD{s, (a - (s - bb)) + (b - bb)}; }
__device__ __forceinline__ DD dd_two_prod(double a, double b) { const double p = a * b; return DD{p, __builtin_fma(a, b, -p)}; }
__device__ __forceinline__ DD dd_add(DD a, DD b)
{
  DD s = dd_two_sum(a.hi, b.hi);
  const DD t = dd_two_sum(a.lo, b.lo);
  s.lo += t.hi; s = dd_quick_two_sum(s.hi, s.lo);
  s.lo += t.lo; return dd_quick_two_sum(s.hi, s.lo);
}
// a / b for a small integer b, rb = 1 / b rounded: the two partial quotients only have to be good to an ulp or two (the remainder is
// formed exactly), so they are products with rb -- an IEEE division costs ~35 instructions, and the running mean needs 40 of them
__device__ __forceinline__ DD dd_div_d(DD a, double b, double rb)
{
  const double q1 = a.hi * rb;
  const DD p = dd_two_prod(q1, b);
  DD r = dd_two_sum(a.hi, -p.hi);
  r.lo += a.lo; r.lo -= p.lo;
  const double q2 = (r.hi + r.lo) * rb;
  return dd_quick_two_sum(q1, q2);
}
struct PdfNormalsSet { const double* xy; const uint8_t* mask_in; const uint8_t* mask_io_init; uint8_t* mask_io; double* phi; };

// TWO lanes per point (round 6): the centroid's two running means -- the longest chain of this kernel, ~45 dependent double-double
// operations per neighbour and axis -- are independent, so lane 2 i takes the x axis and lane 2 i + 1 the y axis and they swap the
// results (one DPP quad permute per word); everything else is done by both lanes alike and written by the even one.
__device__ __forceinline__ void pdf_normals_body(const PdfNormalsSet& st, int points, int sr)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 1;
  const bool odd = (t & 1) != 0;
  if (i >= points) return;                                    // (points in pairs of lanes: a pair is in or out together)
  const double NO_PHI = -1e6;                                 // calcPhi's value for a masked-out point (RandomMatching.cpp:155-174)
  // Everything the thread reads is requested HERE, unconditionally, on clamped indices (pinned reads: the optimiser cannot sink them
  // into the conditions below): its two masks, the masks and coordinates of the ten neighbour slots -- ONE memory round trip.  As
  // written first (a read per condition, coordinates only behind a set mask) the kernel was a chain of thirteen: 13 us for 2 x 1 081
  // points.
  constexpr int NB = 10;
  const uint8_t m_in = ld_pinned(&st.mask_in[i]), m_io = ld_pinned(&st.mask_io_init[i]);
  uint8_t mq[NB]; double ax_[NB], ay_[NB]; bool v_[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) {
    int q = i + j - NB / 2;
    q = q < 0 ? 0 : (q >= points ? points - 1 : q);
    mq[j] = ld_pinned(&st.mask_in[q]);
    ax_[j] = ld_pinned(&st.xy[2 * q]); ay_[j] = ld_pinned(&st.xy[2 * q + 1]);
  }
  if (i < sr || i >= points - sr) { if (!odd) { st.mask_io[i] = 0; st.phi[i] = NO_PHI; } return; }
  if (!m_in || !m_io) { if (!odd) { st.mask_io[i] = 0; st.phi[i] = NO_PHI; } return; }   // (mask_io <= mask_in on entry)
  if (!odd) st.mask_io[i] = 1;
  // the <= 10 neighbours stay in their slots j = -5 .. 4 (registers, every loop unrolled over the ten slots and skipping the
  // masked-out ones in order): the same sequence of operations as over the compacted list, without an indexed private array
  // (sr <= i < points - sr here, so no slot inside the search radius was clamped)
  const double own_x = ax_[NB / 2], own_y = ay_[NB / 2];
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < NB; j++) {
    v_[j] = (j - NB / 2 >= -sr) && (j - NB / 2 < sr) && mq[j] != 0;
    if (!v_[j]) { ax_[j] = 0.0; ay_[j] = 0.0; }
    cnt += v_[j] ? 1 : 0;
  }
  if (cnt <= 3) { if (!odd) { st.mask_io[i] = 0; st.phi[i] = NO_PHI; } return; }
  // Matrix::pcaAnalysis (gsl/Matrix.cpp:227-327), see pca2_axes below
  double cent[2];
  {
    // this lane's axis (both lanes of a pair are here: every exit above depends on the point only)
    DD mean = DD{0.0, 0.0};
    int k = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      if (PDF_UNLIKELY(!v_[j])) continue;        // (most neighbours are there: the region sits in line, without a skip branch -- a branch is 14-30 cycles)
      k++;
      DD d = dd_two_sum(odd ? ay_[j] : ax_[j], -mean.hi);                  // x - mean
      d.lo -= mean.lo; d = dd_quick_two_sum(d.hi, d.lo);
      mean = dd_add(mean, dd_div_d(d, (double)k, __builtin_amdgcn_rcp((double)k)));      // (v_rcp_f64: good to an ulp, which is all dd_div_d asks)
    }
    // the partner's mean: quad_perm [1, 0, 3, 2]
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(mean.hi), 0xB1, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(mean.hi), 0xB1, 0xf, 0xf, false);
    const double other = __hiloint2double(hi, lo);
    cent[0] = odd ? other : mean.hi; cent[1] = odd ? mean.hi : other;
  }
#pragma unroll
  for (int j = 0; j < NB; j++) { ax_[j] = ax_[j] + (-cent[0]); ay_[j] = ay_[j] + (-cent[1]); }      // mc (slots of masked-out neighbours: unused)
  double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
  for (int j = 0; j < NB; j++) if (PDF_LIKELY(v_[j])) { a += ax_[j] * ax_[j]; b += ax_[j] * ay_[j]; c += ay_[j] * ay_[j]; }
  const double th = 0.5 * atan2(2.0 * b, a - c);
  const double V[2][2] = {{cos(th), -sin(th)}, {sin(th), cos(th)}};
  double mx[2], mn[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    mx[q] = -__builtin_inf(); mn[q] = __builtin_inf();
#pragma unroll
    for (int j = 0; j < NB; j++) {
      if (PDF_UNLIKELY(!v_[j])) continue;
      double pr = 0.0;
      pr += V[0][q] * ax_[j]; pr += V[1][q] * ay_[j];
      mx[q] = fmax(mx[q], pr); mn[q] = fmin(mn[q], pr);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const double ext = mx[q] - mn[q];
    const double align = ext > 1e-6 ? (mx[q] + mn[q]) / 2.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 2; j++) cent[j] += V[j][q] * align;
  }
  double ax[2][4];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const double ext = mx[q] - mn[q];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const double e = V[j][q] * ext / 2.0;
      ax[q][2 * j] = cent[j] - e; ax[q][2 * j + 1] = cent[j] + e;
    }
  }
  // RandomMatching::calcNormals (RandomMatching.cpp:82-153)
  const double xLong = ax[0][1] - ax[0][0], yLong = ax[0][3] - ax[0][2];
  const double xShort = ax[1][1] - ax[1][0], yShort = ax[1][3] - ax[1][2];
  const double lenLongSqr = xLong * xLong + yLong * yLong, lenShortSqr = xShort * xShort + yShort * yShort;
  if (odd) return;
  if (lenShortSqr > 1e-6 && (lenLongSqr / lenShortSqr) < 4.0) { st.mask_io[i] = 0; st.phi[i] = NO_PHI; return; }
  const double len = sqrt(lenShortSqr);
  double nx, ny;
  if ((own_x * xShort + own_y * yShort) < 0.0) { nx = xShort / len; ny = yShort / len; }
  else { nx = -xShort / len; ny = -yShort / len; }
  st.phi[i] = atan2(ny, nx);
}
__global__ void __launch_bounds__(256)
k_pdf_normals(PdfNormalsSet set0, PdfNormalsSet set1, int points, int sr)
{
  pdf_normals_body(blockIdx.y == 0 ? set0 : set1, points, sr);
}
struct PdfNormalsEntry { PdfNormalsSet set0, set1; int points, sr; };
struct PdfNormalsBatch { PdfNormalsEntry e[PDF_BATCH_BYVAL]; };
__global__ void __launch_bounds__(256)
k_pdf_normals_batch(PdfNormalsBatch b)
{
  const PdfNormalsEntry& e = b.e[blockIdx.z];
  pdf_normals_body(blockIdx.y == 0 ? e.set0 : e.set1, e.points, e.sr);
}

// ---- the list building of TSD_PDFMatching::match on the device (fused scan: nothing returns to the host between the ray cast and the
// registration).  ONE workgroup: extractSamples of both sets (index order), pickControlSet and the trial picks -- the reference erases
// the picked element from a vector, i.e. picks the r-th REMAINING element in index order: a Lehmer code, decoded for all picks at once
// (see the body) -- then the candidates of every trial in one pass over the window's part of the sampled scene list, each
// carrying its place in the reference's serial order (trial-major, scene index ascending) as the key the arg-max breaks ties on.
struct PdfPrepareArgs {
  const uint8_t* mask_m; const uint8_t* mask_s;       // after the normals (mMp, mSp)
  const double* phi_m; const double* phi_s;
  const double* S;                                    // scene points, beam-indexed
  const int* draws_control; const int* draws_trials;
  double2* control; PdfCandidate* cand; PdfHeader* hdr;
  int n, sr, span, trials_cfg, size_control_set, max_cand;
  double phi_max;
};
constexpr int PDF_MAX_TRIALS = 512;        // (k_pdf_prepare keeps every per-trial array in LDS)
__device__ __forceinline__ void pdf_prepare_body(const PdfPrepareArgs& p)
{
  // everything the picks and the candidate passes read lives in LDS: the two sample lists, the scene's masks and angles
  __shared__ unsigned short s_idx[2][TSD_MAX_BEAMS];          // [0] scene, [1] model: valid indices, ascending
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double* s_phi_s = s_dyn;                                    // [n]
  double* s_phi_m = s_dyn + p.n;                              // [n] (the picked model points' angles are looked up here, not in global memory behind the picks)
  double* s_pm = s_dyn + 2 * p.n;                             // [trials] the picked model points' angles
  __shared__ unsigned short s_rank[TSD_MAX_BEAMS];            // position in s_idx[0] of the first sampled scene point with index >= i
  __shared__ int s_wcnt[2][(TSD_MAX_BEAMS + 1023) / 1024][16], s_trial[PDF_MAX_TRIALS];
  __shared__ int s_draw_c[PDF_MAX_CONTROL], s_draw_t[PDF_MAX_TRIALS];      // the draws (a global read per pick would be a round trip per pick)
  __shared__ unsigned short s_ctrl[PDF_MAX_CONTROL];                        // picked control points (scene indices)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
#ifdef TSD_PDF_STAMPS    // diagnostic build: where the kernel's time goes (thread 0, 100 MHz clock, printed)
  long long st_[8]; int sn_ = 0; long long sx_[6] = {0, 0, 0, 0, 0, 0}; int sxn_ = 0;
#define QSTAMP() do { if (tid == 0 && sn_ < 8) st_[sn_++] = wall_clock64(); } while (0)
#define XSTAMP() do { if (tid == 0 && sxn_ < 6) sx_[sxn_++] = wall_clock64(); } while (0)
#else
#define QSTAMP() do {} while (0)
#define XSTAMP() do {} while (0)
#endif
  QSTAMP();
  // Everything the kernel reads from global memory ahead of the picks is requested HERE, at once: both masks of every 1024-beam round
  // (unconditional reads of a clamped index), both angle arrays, the draws -- one memory round trip where the rounds of the list
  // building used to pay one each.
  constexpr int ROUNDS = (TSD_MAX_BEAMS + 1023) / 1024;
  uint8_t ms_[ROUNDS], mm_[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const int i = r * 1024 + tid, ic = i < p.n ? i : 0;
    ms_[r] = ld_pinned(&p.mask_s[ic]); mm_[r] = ld_pinned(&p.mask_m[ic]);
  }
  for (int i = tid; i < p.n; i += 1024) { s_phi_s[i] = p.phi_s[i]; s_phi_m[i] = p.phi_m[i]; }
  for (int i = tid; i < p.size_control_set && i < PDF_MAX_CONTROL; i += 1024) s_draw_c[i] = p.draws_control[i];
  for (int i = tid; i < p.trials_cfg && i < PDF_MAX_TRIALS; i += 1024) s_draw_t[i] = p.draws_trials[i];
  // extractSamples (RandomMatching.cpp:41-50) of both sets at once: i = sr .. n - sr - 1 with the mask set, in index order.
  // The waves' counts of every round meet in LDS behind ONE barrier; every thread then sums the rounds and waves ahead of it.
  int nS = 0, nM = 0;
  unsigned long long bs_[ROUNDS], bm_[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const int i = r * 1024 + tid;
    const bool in = i >= p.sr && i < p.n - p.sr;
    bs_[r] = __ballot(in && ms_[r] != 0); bm_[r] = __ballot(in && mm_[r] != 0);
    if (lane == 0) { s_wcnt[0][r][wave] = __popcll(bs_[r]); s_wcnt[1][r][wave] = __popcll(bm_[r]); }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    if (r * 1024 >= p.n) break;
    const int i = r * 1024 + tid;
    int offS = nS, offM = nM, totS = 0, totM = 0;
    for (int w = 0; w < 16; w++) {
      const int cs = s_wcnt[0][r][w], cm = s_wcnt[1][r][w];
      if (w < wave) { offS += cs; offM += cm; }
      totS += cs; totM += cm;
    }
    const bool fs = ((bs_[r] >> lane) & 1ull) != 0ull, fm = ((bm_[r] >> lane) & 1ull) != 0ull;
    if (i < p.n) s_rank[i] = (unsigned short)(offS + __popcll(bs_[r] & lt));      // sampled scene points with an index below i
    if (fs) s_idx[0][offS + __popcll(bs_[r] & lt)] = (unsigned short)i;
    if (fm) s_idx[1][offM + __popcll(bm_[r] & lt)] = (unsigned short)i;
    nS += totS; nM += totM;
  }
  __syncthreads();
  QSTAMP();
  const int nC = p.size_control_set < nS ? p.size_control_set : nS;
  const bool identity = nS < 3 || nM < 3;                                   // "Too less valid points" (:129-139)
  int trials = p.trials_cfg < nM ? p.trials_cfg : nM;
  if (trials > PDF_MAX_TRIALS) trials = PDF_MAX_TRIALS;                     // (the host refuses more)
  if (identity) trials = 0;
  // pickControlSet (RandomMatching.cpp:52-80) and the trial picks (TSD_PDFMatching.cpp:185-199).  The reference erases the picked element
  // from a vector, i.e. pick k takes the r_k-th REMAINING element in index order: the picks' positions p_k are the decoding of a Lehmer
  // code, p_k = r_k + #{j < k : p_j <= p_k}.  Rounds 3-5 decoded it pick by pick (one wave per set, ~140 cycles per pick: 11 us of this
  // kernel's 20); it decodes by halving instead: a block of picks whose positions are known RELATIVE TO THE ELEMENTS LEFT WHEN THE BLOCK
  // STARTS (a single pick: its rank) is kept sorted by position; two neighbouring blocks L, R merge into one by moving R's positions
  // back to L's start -- y -> y + c(y), c(y) = #{i : q_i - i <= y} over L's sorted positions q_0 < q_1 < ... (q_i - i elements lie below
  // q_i once the i picks below it are gone) -- and y + c(y)'s place in the merged order is its own index + c(y), q_i's is i + #{y < q_i - i}.
  // One search per pick and level over the sibling block, every pick of both sets at once: ceil(log2(picks)) levels.  A wave holds
  // 64 consecutive picks of ONE set, so the first six levels (blocks of up to 64 picks) stay inside the wave -- LDS executes a wave's
  // accesses in order: no barrier -- and only the levels above meet at workgroup barriers (two for the node's 140 / 100 picks).  Both
  // kinds of search are ONE loop of a fixed number of steps (the block size's bits): "how many of the sibling's entries satisfy a
  // monotone predicate", L and R lanes side by side (tests/test_cpu_oracle_properties.py: the same decoding against the erase loop).
  __shared__ unsigned int s_lh[2][PDF_MAX_CONTROL + PDF_MAX_TRIALS];      // ping-pong: position << 16 | pick; the control set at [0, nC), the trials at [PDF_MAX_CONTROL, ..)
  const int cw = (nC + 63) >> 6, tw = (trials + 63) >> 6;                 // waves' worth of picks per set
  // one level for pick k of a set of K picks in blocks of B: reads `a`, writes `o` (both the set's own part of a buffer)
  auto merge_level = [&](const unsigned int* a, unsigned int* o, int k, int K, int B) {
    if (k >= K) return;
    const int base = k & ~(2 * B - 1), i = k - base;
    const int nL = K - base < B ? K - base : B, nR = K - base - B < B ? K - base - B : B;      // (nR <= 0: a lone left block, carried over)
    const unsigned int v = a[k];
    if (nR <= 0) { o[k] = v; return; }
    const int val = (int)(v >> 16);
    const bool right = i >= B;
    // L: the picks of R whose position (relative to R's start) is below val - i.  R: the picks of L that lie at or below it once it is
    // moved back, i.e. those with q_m - m <= val.  Either way a prefix of the sibling's sorted entries.
    const unsigned int* sib = a + base + (right ? 0 : B);
    const int n = right ? nL : nR, thresh = right ? val + 1 : val - i;
    // the count by descent over the steps B, B / 2, .., 1 ("is entry pos + step - 1 still in the prefix?"), TWO steps per LDS round trip:
    // the second step's probe is one of two positions, both requested with the first step's
    auto key_at = [&](unsigned int w, int m) { return (int)(w >> 16) - (right ? m : 0); };
    auto at = [&](int m) { return sib[m < n ? m : n - 1]; };
    int pos = 0, step = B;
    while (step >= 2) {
      const int h = step >> 1;
      const int m1 = pos + step - 1, m2a = pos + h - 1, m2b = pos + step + h - 1;
      const unsigned int w1 = at(m1), w2a = at(m2a), w2b = at(m2b);
      const bool p1 = m1 < n && key_at(w1, m1 < n ? m1 : n - 1) < thresh;
      const int m2 = p1 ? m2b : m2a;
      const bool p2 = m2 < n && key_at(p1 ? w2b : w2a, m2 < n ? m2 : n - 1) < thresh;
      pos += (p1 ? step : 0) + (p2 ? h : 0);
      step >>= 2;
    }
    if (step == 1) {
      const int m = pos;
      if (m < n && key_at(at(m), m) < thresh) pos += 1;
    }
    if (right) o[base + (i - B) + pos] = ((unsigned)(val + pos) << 16) | (v & 0xFFFFu);
    else o[base + i + pos] = v;
  };
  for (int vw = wave; vw < cw + tw; vw += 16) {
    const bool tr = vw >= cw;
    const int k = ((tr ? vw - cw : vw) << 6) + lane, K = tr ? trials : nC, off = tr ? PDF_MAX_CONTROL : 0;
    // (the ranks: r_k = draw_k mod (remaining before pick k))
    if (k < K) s_lh[0][off + k] = (((unsigned)(tr ? s_draw_t[k] : s_draw_c[k]) % (unsigned)((tr ? nM : nS) - k)) << 16) | (unsigned)k;
    int lvl = 0;
#pragma unroll
    for (int B = 1; B < 64; B <<= 1, lvl++) {       // (always six levels: both sets end in buffer 0)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      merge_level(s_lh[lvl & 1] + off, s_lh[(lvl & 1) ^ 1] + off, k, K, B);
    }
  }
  __syncthreads();
  XSTAMP();
  int lvl = 0;
  for (int B = 64; B < nC || B < trials; B <<= 1, lvl++) {
    for (int vw = wave; vw < cw + tw; vw += 16) {
      const bool tr = vw >= cw;
      const int off = tr ? PDF_MAX_CONTROL : 0;
      merge_level(s_lh[lvl & 1] + off, s_lh[(lvl & 1) ^ 1] + off, ((tr ? vw - cw : vw) << 6) + lane, tr ? trials : nC, B);
    }
    __syncthreads();
  }
  XSTAMP();
  {
    // POSITIONS in the sample lists (translated below), by pick
    const unsigned int* fin = s_lh[lvl & 1];
    unsigned short* out_t = reinterpret_cast<unsigned short*>(s_trial);
    for (int vw = wave; vw < cw + tw; vw += 16) {
      const bool tr = vw >= cw;
      const int k = ((tr ? vw - cw : vw) << 6) + lane;
      if (k < (tr ? trials : nC)) {
        const unsigned int v = fin[(tr ? PDF_MAX_CONTROL : 0) + k];
        (tr ? out_t : s_ctrl)[v & 0xFFFFu] = (unsigned short)(v >> 16);
      }
    }
  }
  XSTAMP();
  __syncthreads();
  XSTAMP();
  // positions -> scene / model indices, all at once (s_trial's picks were parked as 16-bit positions in its own storage: read all, then write)
  {
    int idx_t[(PDF_MAX_TRIALS + 1023) / 1024];
#pragma unroll
    for (int j = 0; j < (PDF_MAX_TRIALS + 1023) / 1024; j++) { const int k = tid + 1024 * j; idx_t[j] = k < trials ? (int)s_idx[1][reinterpret_cast<unsigned short*>(s_trial)[k]] : 0; }
    for (int k = tid; k < nC; k += 1024) s_ctrl[k] = s_idx[0][s_ctrl[k]];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < (PDF_MAX_TRIALS + 1023) / 1024; j++) { const int k = tid + 1024 * j; if (k < trials) s_trial[k] = idx_t[j]; }
  }
  __syncthreads();
  QSTAMP();
  // the control points: requested now, stored behind the candidate pass (nothing in this kernel waits for them)
  static_assert(PDF_MAX_CONTROL <= 1024, "one control point per thread");
  double2 ctrl_xy = make_double2(0.0, 0.0);
  if (tid < nC) { const int idx = s_ctrl[tid]; ctrl_xy = make_double2(ld_pinned(&p.S[2 * idx]), ld_pinned(&p.S[2 * idx + 1])); }
  for (int t = tid; t < trials; t += 1024) s_pm[t] = s_phi_m[s_trial[t]];
  __syncthreads();
  // ---- candidates (:200-215) in ONE pass: wave w takes the trials w, w + 16, ...; the matches of a 64-beam round draw their list places
  // from one LDS counter (one atomic per wave and round).  The list order is whatever order the waves arrive in; the reference's serial
  // order lives in the key `ti` (PdfCandidate), which is all the arg-max needs.  (Round 3's first form counted per trial, scanned the
  // counts and wrote in a second pass: 14 us of this kernel's 31.)
  const double PI_D = 3.14159265358979323846;
  __shared__ int s_ncand;
  if (tid == 0) s_ncand = 0;
  __syncthreads();
  QSTAMP();
  // (a trial's window into the sampled scene list, for every trial at once: the pass below used to start each trial with three dependent
  // LDS round trips -- the trial's model index, the two list positions)
  __shared__ unsigned short s_klo[PDF_MAX_TRIALS], s_khi[PDF_MAX_TRIALS];
  for (int t = tid; t < trials; t += 1024) {
    const int idx = s_trial[t];
    const int iMin = idx - p.span > p.sr ? idx - p.span : p.sr, iMax = idx + p.span < p.n - p.sr ? idx + p.span : p.n - p.sr;
    s_klo[t] = iMin < iMax ? s_rank[iMin] : (unsigned short)0;
    s_khi[t] = iMin < iMax ? s_rank[iMax] : (unsigned short)0;
  }
  __syncthreads();
  // wave w takes the trials w, w + 16, ...: TWO of them per turn, their reads in flight together (the turn is a chain of four LDS round
  // trips: the trials' data -> list entries -> angles -> list places; the matches of both draw their places from one LDS counter at once)
  for (int t = wave; t < trials; t += 32) {
    const int t2 = t + 16;
    const bool has2 = t2 < trials;
    const int t2c = has2 ? t2 : t;
    // everything that depends on the trials alone: one round trip
    const int lo1 = (int)s_klo[t], hi1 = (int)s_khi[t], lo2 = (int)s_klo[t2c], hi2 = has2 ? (int)s_khi[t2c] : 0;
    const int idx1 = s_trial[t], idx2 = s_trial[t2c];
    const double pm1 = s_pm[t], pm2 = s_pm[t2c];
    const int len = (hi1 - lo1) > (hi2 - lo2) ? (hi1 - lo1) : (hi2 - lo2);
    for (int o = 0; o < len; o += 64) {
      const int k1 = lo1 + o + lane, k2 = lo2 + o + lane;
      const bool in1 = k1 < hi1, in2 = k2 < hi2;
      const int i1 = in1 ? (int)s_idx[0][k1] : 0, i2 = in2 ? (int)s_idx[0][k2] : 0;                 // second round trip
      const double ps1 = s_phi_s[i1], ps2 = s_phi_s[i2];                                               // third
      auto angle = [&](double pm, double ps, bool in, bool& ok) {
        double phi = pm - ps;
        if (phi > PI_D) phi -= 2.0 * PI_D;
        else if (phi < -PI_D) phi += 2.0 * PI_D;
        ok = in && fabs(phi) < p.phi_max;
        return phi;
      };
      bool ok1, ok2;
      const double ph1 = angle(pm1, ps1, in1, ok1), ph2 = angle(pm2, ps2, in2, ok2);
      const unsigned long long b1 = __ballot(ok1), b2 = __ballot(ok2);
      if (b1 | b2) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_ncand, __popcll(b1) + __popcll(b2));                       // fourth: the places of both trials' matches
        base = __builtin_amdgcn_readfirstlane(base);
        const int at1 = base + __popcll(b1 & lt), at2 = base + __popcll(b1) + __popcll(b2 & lt);
        if (ok1 && at1 < p.max_cand) p.cand[at1] = PdfCandidate{idx1, (t << PDF_I_BITS) | i1, ph1};
        if (ok2 && at2 < p.max_cand) p.cand[at2] = PdfCandidate{idx2, (t2 << PDF_I_BITS) | i2, ph2};
      }
    }
  }
  if (tid < nC) p.control[tid] = ctrl_xy;
  __syncthreads();
  QSTAMP(); QSTAMP(); QSTAMP();
#ifdef TSD_PDF_STAMPS
  if (tid == 0) printf("   picks in detail (x10 ns): ranks + barrier %lld | pass 1 %lld | pass 2 %lld | wait for the other wave %lld\n", sx_[0] - st_[1], sx_[1] - sx_[0], sx_[2] - sx_[1], sx_[3] - sx_[2]);
  if (tid == 0) printf("k_pdf_prepare (x10 ns): lists %lld | picks %lld | control + angles %lld | count pass %lld | scan %lld | write pass %lld  (nS %d nM %d nC %d trials %d)\n",
                       st_[1] - st_[0], st_[2] - st_[1], st_[3] - st_[2], st_[4] - st_[3], st_[5] - st_[4], st_[6] - st_[5], nS, nM, nC, trials);
#endif
  if (tid == 0) {
    PdfHeader h;
    h.n_cand = s_ncand < p.max_cand ? s_ncand : p.max_cand; h.n_control = nC; h.n_model_valid = nM; h.n_scene_valid = nS;
    h.identity = identity ? 1 : 0; h.pad[0] = h.pad[1] = h.pad[2] = 0;
    *p.hdr = h;
  }
}
__global__ void __launch_bounds__(1024)
k_pdf_prepare(PdfPrepareArgs p)
{
  pdf_prepare_body(p);
}
struct PdfPrepareBatch { PdfPrepareArgs e[PDF_BATCH_BYVAL]; };
__global__ void __launch_bounds__(1024)
k_pdf_prepare_batch(PdfPrepareBatch b)
{
  pdf_prepare_body(b.e[blockIdx.x]);
}

// ---- host side: RandomMatching's O(beams) preparation ------------------------------------------------------

// Matrix::pcaAnalysis for n x 2 points (obcore/math/linalg/gsl/Matrix.cpp:227-327): centroid (gsl_stats_mean: running
// mean in long double), M^T M, its eigenvectors (gsl_linalg_SV_decomp_jacobi of a symmetric 2 x 2 matrix = its
// eigen-decomposition; column signs are free and nothing below depends on them), extents along both axes.
constexpr int PCA_MAX_POINTS = 16;          // >= 2 * searchRadius (= _pcaSearchRange = 10)
static void pca2_axes(const double* pts, int n, double axes[2][4])
{
  double cent[2];
  for (int j = 0; j < 2; j++) {
    long double mean = 0.0L;
    for (int i = 0; i < n; i++) mean += ((long double)pts[2 * i + j] - mean) / (long double)(i + 1);
    cent[j] = (double)mean;
  }
  double mc[2 * PCA_MAX_POINTS];                              // (n <= 2 * searchRadius: no heap in a per-point routine)
  for (int i = 0; i < n; i++) { mc[2 * i] = pts[2 * i] + (-cent[0]); mc[2 * i + 1] = pts[2 * i + 1] + (-cent[1]); }
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = 0; i < n; i++) { a += mc[2 * i] * mc[2 * i]; b += mc[2 * i] * mc[2 * i + 1]; c += mc[2 * i + 1] * mc[2 * i + 1]; }
  const double th = 0.5 * std::atan2(2.0 * b, a - c);
  const double V[2][2] = {{std::cos(th), -std::sin(th)}, {std::sin(th), std::cos(th)}};
  double mx[2], mn[2];
  for (int i = 0; i < 2; i++) {
    mx[i] = -INFINITY; mn[i] = INFINITY;
    for (int r = 0; r < n; r++) {
      double pr = 0.0;
      pr += V[0][i] * mc[2 * r]; pr += V[1][i] * mc[2 * r + 1];
      mx[i] = std::max(mx[i], pr); mn[i] = std::min(mn[i], pr);
    }
  }
  for (int i = 0; i < 2; i++) {
    const double ext = mx[i] - mn[i];
    const double align = ext > 1e-6 ? (mx[i] + mn[i]) / 2.0 : 0.0;
    for (int j = 0; j < 2; j++) cent[j] += V[j][i] * align;
  }
  for (int i = 0; i < 2; i++) {
    const double ext = mx[i] - mn[i];
    for (int j = 0; j < 2; j++) {
      const double e = V[j][i] * ext / 2.0;
      axes[i][2 * j] = cent[j] - e; axes[i][2 * j + 1] = cent[j] + e;
    }
  }
}

// RandomMatching::calcNormals (RandomMatching.cpp:82-153)
static void calc_normals(const double* M, int points, std::vector<double>& N, const uint8_t* mask_in, std::vector<uint8_t>& mask_out, int sr)
{
  for (int i = 0; i < sr && i < points; i++) mask_out[i] = 0;
  for (int i = std::max(points - sr, 0); i < points; i++) mask_out[i] = 0;
  double A[2 * PCA_MAX_POINTS];
  for (int i = sr; i < points - sr; i++) {
    if (!mask_in[i]) continue;
    // A point that is masked out already (the scene's random subsampling runs BEFORE this, RandomMatching.cpp:176-189 /
    // TSD_PDFMatching.cpp:81-102) gets a normal in the reference too, but nothing reads it: its mask stays false, calcPhi and
    // extractSamples skip it.  Not computing it changes no output and saves ~80 % of the scene's PCAs.
    if (!mask_out[i]) continue;
    unsigned cnt = 0;
    for (int j = -sr; j < sr; j++) if (mask_in[i + j]) cnt++;
    if (cnt > 3) {
      cnt = 0;
      for (int j = -sr; j < sr; j++) if (mask_in[i + j]) { A[2 * cnt] = M[2 * (i + j)]; A[2 * cnt + 1] = M[2 * (i + j) + 1]; cnt++; }
      double ax[2][4];
      pca2_axes(A, (int)cnt, ax);
      const double xLong = ax[0][1] - ax[0][0], yLong = ax[0][3] - ax[0][2];
      const double xShort = ax[1][1] - ax[1][0], yShort = ax[1][3] - ax[1][2];
      const double lenLongSqr = xLong * xLong + yLong * yLong, lenShortSqr = xShort * xShort + yShort * yShort;
      // main axis needs to be twice as long as the second one
      if (lenShortSqr > 1e-6 && (lenLongSqr / lenShortSqr) < 4.0) { mask_out[i] = 0; continue; }
      const double len = std::sqrt(lenShortSqr);
      if ((M[2 * i] * xShort + M[2 * i + 1] * yShort) < 0.0) { N[2 * i] = xShort / len; N[2 * i + 1] = yShort / len; }
      else { N[2 * i] = -xShort / len; N[2 * i + 1] = -yShort / len; }
    } else mask_out[i] = 0;
  }
}

}  // namespace tsd

using namespace tsd;

extern "C" int tsd_tsdpdf_match(tsd_ctx* ctx, const double pose33[9], const double* model_xy_2B, const uint8_t* mask_m,
                                const double* scene_xy_2B, const uint8_t* mask_s, int beams, const tsd_tsdpdf_params* prm,
                                const int* draws_subsample, const int* draws_control, const int* draws_trials,
                                tsd_tsdpdf_result* result)
{
  if (!ctx || !pose33 || !model_xy_2B || !mask_m || !scene_xy_2B || !mask_s || !prm || !draws_subsample || !draws_control ||
      !draws_trials || !result)
    return TSD_E_ARG;
  if (beams < 1 || beams > TSD_MAX_BEAMS || prm->size_control_set < 0 || prm->size_control_set > PDF_MAX_CONTROL || prm->trials < 0)
    return set_error(ctx, TSD_E_CAPACITY, "tsd_tsdpdf_match: beams / control set out of range", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;     // (the scoring reads the grid: behind a push still on the push stream)
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // TSD_MODE3_TIMING=1: the host phases of this call (printed every 100 calls)
  static const bool timing = std::getenv("TSD_MODE3_TIMING") != nullptr;
  static double t_acc[6]; static int t_calls;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](int i) { if (!timing) return; const auto now = std::chrono::steady_clock::now(); t_acc[i] += std::chrono::duration<double, std::micro>(now - t_last).count(); t_last = now; };
  const int n = beams;
  const int SR = 10 / 2;                                   // _pcaSearchRange / 2 (TSD_PDFMatching.cpp:18)
  for (int i = 0; i < 9; i++) result->T[i] = (i % 4 == 0) ? 1.0 : 0.0;     // TBest.setIdentity()
  result->probability = 0.0; result->idx_model = -1; result->idx_scene = -1; result->candidates = 0;
  result->valid_model = 0; result->valid_scene = 0; result->control_points = 0; result->reserved = 0;
  if (n < 3) return TSD_OK;                                // "Model and scene contain too less points" (:53-57)
  const double* M = model_xy_2B; const double* S = scene_xy_2B;

  // ---- masks, then the normals of both sets (:63-102)
  std::vector<double> phiM((size_t)n), phiS((size_t)n);
  std::vector<uint8_t> mMp(mask_m, mask_m + n), mSp(mask_s, mask_s + n);
  unsigned valid = 0;
  for (int i = 0; i < n; i++) if (mSp[i]) valid++;
  double probability = 180.0 / (double)valid;
  if (probability < 0.99) {                                // subsampleMask (RandomMatching.cpp:176-189)
    if (probability > 1.0) probability = 1.0;
    if (probability < 0.0) probability = 0.0;
    const int thresh = (int)(1000.0 - probability * 1000.0 + 0.5);
    for (int i = 0; i < n; i++) if ((draws_subsample[i] % 1000) < thresh) mSp[i] = 0;
  }
  const bool res_ok = prm->ang_res > 1e-6;                 // (reported where the reference does, behind the point-count exits: :171-175)
  const double phi_max = std::min(prm->phi_max, M_PI * 0.5);
  int span = res_ok ? (int)std::floor(phi_max / prm->ang_res) : n;
  if (span > n) span = n;
  // device buffer: [M | S | mask_in M, S | mask_io M, S | phi M, S | control | candidates | pose | prob | result]; the candidate
  // list is sized for its upper bound (trials x (2 span) scene points) because the normals come back before it is known
  const size_t max_cand = (size_t)std::max(prm->trials, 0) * (size_t)std::min(2 * span + 1, n) + 1;
  const size_t bM = (size_t)n * 16, bMask = ((size_t)n + 15) & ~(size_t)15, bPhi = (size_t)n * 8;
  const size_t bC = (size_t)std::max(prm->size_control_set, 1) * 16, bK = max_cand * sizeof(PdfCandidate), bP = 80;
  const size_t off_S = bM, off_mi = 2 * bM, off_mo = off_mi + 2 * bMask, off_phi = off_mo + 2 * bMask;
  const size_t off_C = off_phi + 2 * bPhi, off_K = off_C + bC, off_P = off_K + ((bK + 15) & ~(size_t)15);
  const size_t off_prob = off_P + bP, off_res = off_prob + max_cand * sizeof(double);
  const size_t total = off_res + sizeof(PdfResult);
  if (total > ctx->pdf_bytes) {
    TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_pdf) hipFree(ctx->d_pdf);
    if (ctx->h_pdf) hipHostFree(ctx->h_pdf);
    ctx->d_pdf = nullptr; ctx->h_pdf = nullptr; ctx->pdf_bytes = 0;
    const size_t want = total + total / 4;
    TSD_HIP_CHECK(ctx, hipMalloc(&ctx->d_pdf, want));
    TSD_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_pdf, want, hipHostMallocDefault));
    ctx->pdf_bytes = want;
  }
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));          // (the pinned buffer of a previous call is free)
  char* h = ctx->h_pdf; char* d = ctx->d_pdf;
  static const bool host_normals = std::getenv("TSD_PDF_HOST_NORMALS") != nullptr;     // A/B and cross-check: the host restatement
  std::memcpy(h, M, bM); std::memcpy(h + off_S, S, bM);
  if (host_normals) {
    std::vector<double> NM(2 * (size_t)n, 0.0), NS(2 * (size_t)n, 0.0);
    calc_normals(M, n, NM, mask_m, mMp, SR);
    for (int i = 0; i < n; i++) phiM[i] = mMp[i] ? std::atan2(NM[2 * i + 1], NM[2 * i]) : -1e6;     // calcPhi (:155-174)
    lap(0);
    calc_normals(S, n, NS, mask_s, mSp, SR);
    for (int i = 0; i < n; i++) phiS[i] = mSp[i] ? std::atan2(NS[2 * i + 1], NS[2 * i]) : -1e6;
    TSD_HIP_CHECK(ctx, hipMemcpyAsync(d, h, 2 * bM, hipMemcpyHostToDevice, ctx->stream));
  } else {
    std::memcpy(h + off_mi, mask_m, (size_t)n); std::memcpy(h + off_mi + bMask, mask_s, (size_t)n);
    std::memcpy(h + off_mo, mMp.data(), (size_t)n); std::memcpy(h + off_mo + bMask, mSp.data(), (size_t)n);
    TSD_HIP_CHECK(ctx, hipMemcpyAsync(d, h, off_phi, hipMemcpyHostToDevice, ctx->stream));
    PdfNormalsSet sm{reinterpret_cast<const double*>(d), reinterpret_cast<const uint8_t*>(d + off_mi), reinterpret_cast<const uint8_t*>(d + off_mo),
                     reinterpret_cast<uint8_t*>(d + off_mo), reinterpret_cast<double*>(d + off_phi)};
    PdfNormalsSet ss{reinterpret_cast<const double*>(d + off_S), reinterpret_cast<const uint8_t*>(d + off_mi + bMask),
                     reinterpret_cast<const uint8_t*>(d + off_mo + bMask), reinterpret_cast<uint8_t*>(d + off_mo + bMask),
                     reinterpret_cast<double*>(d + off_phi + bPhi)};
    {
      ScopedKernelTimer t(ctx, "tsdpdf", true);
      hipLaunchKernelGGL(k_pdf_normals, dim3((2 * n + 255) / 256, 2), dim3(256), 0, ctx->stream, sm, ss, n, SR);
    }
    TSD_HIP_CHECK(ctx, hipGetLastError());
    TSD_HIP_CHECK(ctx, hipMemcpyAsync(h + off_mo, d + off_mo, off_C - off_mo, hipMemcpyDeviceToHost, ctx->stream));   // masks + angles
    TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(mMp.data(), h + off_mo, (size_t)n); std::memcpy(mSp.data(), h + off_mo + bMask, (size_t)n);
    std::memcpy(phiM.data(), h + off_phi, bPhi); std::memcpy(phiS.data(), h + off_phi + bPhi, bPhi);
    lap(0);
  }
  std::vector<int> idxM, idxS;
  for (int i = SR; i < n - SR; i++) if (mMp[i]) idxM.push_back(i);                                // extractSamples (:41-50)
  for (int i = SR; i < n - SR; i++) if (mSp[i]) idxS.push_back(i);
  // ---- control set (:106-118, RandomMatching::pickControlSet :52-80)
  int nC = prm->size_control_set;
  if ((int)idxS.size() < nC) nC = (int)idxS.size();
  std::vector<double> control(2 * (size_t)std::max(nC, 1));
  {
    std::vector<int> tmp = idxS;
    for (int k = 0; k < nC; k++) {
      const unsigned r = (unsigned)draws_control[k] % (unsigned)tmp.size();
      const int idx = tmp[r];
      tmp.erase(tmp.begin() + r);
      control[2 * k] = S[2 * idx]; control[2 * k + 1] = S[2 * idx + 1];
    }
  }
  lap(1);
  result->valid_model = (int)idxM.size(); result->valid_scene = (int)idxS.size(); result->control_points = nC;
  if (idxS.size() < 3 || idxM.size() < 3) return TSD_OK;   // "Too less valid points" (:129-139): identity
  int trials = prm->trials;
  if ((int)idxM.size() < trials) trials = (int)idxM.size();
  if (!res_ok) return set_error(ctx, TSD_E_ARG, "tsd_tsdpdf_match: resolution not properly set", hipSuccess);   // :171-175
  // ---- candidates in the reference's serial order (:185-215)
  std::vector<PdfCandidate> cand;
  {
    std::vector<int> tmp = idxM;
    for (int trial = 0; trial < trials; trial++) {
      const int r = (int)((unsigned)draws_trials[trial] % (unsigned)tmp.size());
      const int idx = tmp[r];
      tmp.erase(tmp.begin() + r);
      const int iMin = std::max(idx - span, SR), iMax = std::min(idx + span, n - SR);
      for (int i = iMin; i < iMax; i++) {
        if (!mSp[i]) continue;
        double phi = phiM[idx] - phiS[i];
        if (phi > M_PI) phi -= 2.0 * M_PI;
        else if (phi < -M_PI) phi += 2.0 * M_PI;
        if (std::fabs(phi) < phi_max) cand.push_back(PdfCandidate{idx, (int)(((unsigned)trial << PDF_I_BITS) | (unsigned)i), phi});
      }
    }
  }
  result->candidates = (int)cand.size();
  if (cand.empty()) return TSD_OK;
  if (cand.size() > max_cand) return set_error(ctx, TSD_E_CAPACITY, "tsd_tsdpdf_match: candidate bound", hipSuccess);
  lap(2);

  // ---- device: score + arg-max
  const size_t bKu = cand.size() * sizeof(PdfCandidate);
  std::memcpy(h + off_C, control.data(), (size_t)nC * 16);
  std::memcpy(h + off_K, cand.data(), bKu); std::memcpy(h + off_P, pose33, 9 * sizeof(double));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(d + off_C, h + off_C, bC + bKu, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(d + off_P, h + off_P, bP, hipMemcpyHostToDevice, ctx->stream));
  lap(3);
  {
    ScopedKernelTimer t(ctx, "tsdpdf", true);
    const int nc = (int)cand.size();
    hipLaunchKernelGGL(k_pdf_score, dim3((nc + PDF_WAVES - 1) / PDF_WAVES), dim3(64 * PDF_WAVES), 0, ctx->stream, ctx->grid, reinterpret_cast<const double*>(d + off_P),
                       reinterpret_cast<const double*>(d), reinterpret_cast<const double*>(d + off_S),
                       reinterpret_cast<const double2*>(d + off_C), nC, reinterpret_cast<const PdfCandidate*>(d + off_K), nc,
                       prm->zrand, reinterpret_cast<double*>(d + off_prob), nullptr, nc, nC > 0 ? nC : 1);
    hipLaunchKernelGGL(k_pdf_argmax, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<const double*>(d + off_prob),
                       reinterpret_cast<const PdfCandidate*>(d + off_K), nc, reinterpret_cast<const double*>(d),
                       reinterpret_cast<const double*>(d + off_S), reinterpret_cast<PdfResult*>(d + off_res), nullptr, nullptr, nullptr);
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(h + off_res, d + off_res, sizeof(PdfResult), hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  lap(4);
  if (timing && ++t_calls % 100 == 0)
    std::fprintf(stderr, "tsd_tsdpdf_match, us per call: normals of both sets (%s) %.1f | lists + control set %.1f | candidates (%d) %.1f | staging + H2D issue %.1f | kernels + D2H %.1f\n",
                 host_normals ? "host" : "device", t_acc[0] / t_calls, t_acc[1] / t_calls, (int)cand.size(), t_acc[2] / t_calls, t_acc[3] / t_calls, t_acc[4] / t_calls);
  const PdfResult* r = reinterpret_cast<const PdfResult*>(h + off_res);
  std::memcpy(result->T, r->T, sizeof(r->T));
  result->probability = r->prob; result->idx_model = r->idx; result->idx_scene = r->i;
  return TSD_OK;
}

// ---- registration_mode 3 inside the fused scan -----------------------------------------------------------------------------
extern "C" int tsd_scan_preregister(tsd_sensor* s, const tsd_tsdpdf_params* prm, const double* scene_xy_2B, const uint8_t* mask_s,
                                    const int* draws_subsample, const int* draws_control, const int* draws_trials)
{
  if (!s || !s->ctx || !prm || !scene_xy_2B || !mask_s || !draws_subsample || !draws_control || !draws_trials) return TSD_E_ARG;
  tsd_ctx* ctx = s->ctx;
  const int n = s->beams;
  if (n < 1 || n > TSD_MAX_BEAMS || prm->size_control_set < 0 || prm->size_control_set > PDF_MAX_CONTROL || prm->trials < 0 ||
      prm->trials > PDF_MAX_TRIALS)
    return set_error(ctx, TSD_E_CAPACITY, "tsd_scan_preregister: beams / control set / trials out of range", hipSuccess);
  if (!(prm->ang_res > 1e-6)) return set_error(ctx, TSD_E_ARG, "tsd_scan_preregister: resolution not properly set", hipSuccess);
  // A scan of this sensor may be IN FLIGHT (submitted, not collected): the pre-registration of the scan after it is then armed ahead --
  // its inputs are copied behind that scan's own pre-registration kernels (which read the same device buffer), beside its registration.
  const bool inflight = s->submitted;
  if (s->inflight) return set_error(ctx, TSD_E_ARG, "tsd_scan_preregister: the sensor has a batched / split scan in flight (arm it between two scans)", hipSuccess);
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // The last side-stream copy out of the pinned buffer has to have left it before the host rewrites it (and, BAR mode, before the host
  // stores into d_pre itself).  That copy belongs to the in-flight scan -- or to a pre-registration that was armed ahead while a scan
  // was in flight and is now being RE-armed without having been submitted: nothing the host has seen is ordered behind that one.
  if (s->ev_pre && s->pre_copied && !s->pre_direct) TSD_HIP_CHECK(ctx, hipEventSynchronize(s->ev_pre));
  tsd_sensor::PreLayout L{};
  L.n = n; L.trials = prm->trials; L.size_control_set = prm->size_control_set; L.zrand = prm->zrand;
  L.phi_max = std::min(prm->phi_max, M_PI * 0.5);
  L.span = (int)std::floor(L.phi_max / prm->ang_res);
  if (L.span > n) L.span = n;
  L.max_cand = (int)std::min<size_t>((size_t)std::max(prm->trials, 0) * (size_t)std::min(2 * L.span + 1, n) + 1, (size_t)1 << 22);
  const size_t bM = (size_t)n * 16, bMask = ((size_t)n + 15) & ~(size_t)15, bPhi = (size_t)n * 8;
  auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
  L.off_S = 0; L.off_ms = bM; L.off_msp = L.off_ms + bMask; L.off_dc = L.off_msp + bMask;
  L.off_dt = L.off_dc + al((size_t)std::max(prm->size_control_set, 1) * 4);
  L.in_bytes = L.off_dt + al((size_t)std::max(prm->trials, 1) * 4);
  L.off_mo_m = L.in_bytes; L.off_mo_s = L.off_mo_m + bMask; L.off_phi_m = L.off_mo_s + bMask; L.off_phi_s = L.off_phi_m + bPhi;
  L.off_C = L.off_phi_s + bPhi; L.off_K = L.off_C + al((size_t)std::max(prm->size_control_set, 1) * 16);
  L.off_prob = L.off_K + al((size_t)L.max_cand * sizeof(PdfCandidate)); L.off_hdr = L.off_prob + al((size_t)L.max_cand * 8);
  L.off_res = L.off_hdr + al(sizeof(PdfHeader));
  const size_t total = L.off_res + al(sizeof(PdfResult));
  // The in-flight scan's arg-max writes its header and result into the PINNED buffer at the offsets of ITS layout (pre_res_off_*),
  // and tsd_scan_preregistration_result reads them there after the collect: a layout that needs new buffers, or whose inputs would
  // reach into those records, cannot be armed ahead -- the caller arms it after tsd_scan_collect.
  if (inflight && s->pre_ran && (total > s->pre_bytes || L.in_bytes > s->pre_res_off_hdr))
    return set_error(ctx, TSD_E_ARG, "tsd_scan_preregister: a larger layout than the in-flight scan's cannot be armed ahead of its collect", hipSuccess);
  if (total > s->pre_bytes) {
    TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream2));      // (an armed, not yet submitted pre-registration's copy may still read the old buffer)
    if (s->d_pre) hipFree(s->d_pre);
    if (s->h_pre) hipHostFree(s->h_pre);
    s->d_pre = nullptr; s->h_pre = nullptr; s->pre_bytes = 0;
    s->pre_bar = false;
    if (s->scan_bar && hipExtMallocWithFlags((void**)&s->d_pre, total, hipDeviceMallocFinegrained) == hipSuccess) s->pre_bar = true;
    else { (void)hipGetLastError(); s->d_pre = nullptr; }
    if (!s->pre_bar) TSD_HIP_CHECK(ctx, hipMalloc(&s->d_pre, total));
    TSD_HIP_CHECK(ctx, hipHostMalloc(&s->h_pre, total, hipHostMallocDefault));
    TSD_HIP_CHECK(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&s->h_pre_dev), s->h_pre, 0));
    s->pre_bytes = total;
  }
  // (the pinned buffer is free: the previous scan was collected, i.e. its copy from here and its read-back have completed)
  char* h = s->h_pre;
  std::memcpy(h + L.off_S, scene_xy_2B, bM);
  std::memcpy(h + L.off_ms, mask_s, (size_t)n);
  uint8_t* mSp = reinterpret_cast<uint8_t*>(h + L.off_msp);
  std::memcpy(mSp, mask_s, (size_t)n);
  unsigned valid = 0;
  for (int i = 0; i < n; i++) if (mSp[i]) valid++;
  double probability = 180.0 / (double)valid;
  if (probability < 0.99) {                                // subsampleMask (RandomMatching.cpp:176-189)
    if (probability > 1.0) probability = 1.0;
    if (probability < 0.0) probability = 0.0;
    const int thresh = (int)(1000.0 - probability * 1000.0 + 0.5);
    for (int i = 0; i < n; i++) if ((draws_subsample[i] % 1000) < thresh) mSp[i] = 0;
  }
  std::memcpy(h + L.off_dc, draws_control, (size_t)prm->size_control_set * 4);
  std::memcpy(h + L.off_dt, draws_trials, (size_t)prm->trials * 4);
  s->pre = L;
  // The inputs go to the device NOW, on the side stream: the caller is between two scans -- the previous scan's push and this scan's ray
  // cast are still running on the main stream -- so the copy (6 us on a DMA engine + ~9 us until a kernel behind it on the same stream
  // starts) is off the scan's chain.  The previous scan's kernels that read d_pre ran ahead of its registration, which has ended (the
  // scan was collected).  TSD_PRE_COPY_MAIN=1: the copy at tsd_scan_submit, on the main stream (round 3's first form; A/B).
  static const bool copy_main = [] { const char* e = std::getenv("TSD_PRE_COPY_MAIN"); return e && *e == '1'; }();
  s->pre_copied = false; s->pre_direct = false;
  if (s->pre_bar && !inflight && !copy_main) {
    // the host stores the inputs into the device buffer itself (the previous scan was collected: nothing on the device reads it)
    std::memcpy(s->d_pre, s->h_pre, L.in_bytes);
#if defined(__x86_64__)
    _mm_sfence();
#endif
    s->pre_copied = true; s->pre_direct = true;
    s->pre_armed = true;
    return TSD_OK;
  }
  if (copy_main && inflight) return set_error(ctx, TSD_E_ARG, "tsd_scan_preregister ahead of the collect needs the side-stream copy (TSD_PRE_COPY_MAIN is set)", hipSuccess);
  if (!copy_main) {
    if (inflight && s->pre_done_valid) TSD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream2, s->ev_pre_done, 0));
    if (!s->ev_pre) TSD_HIP_CHECK(ctx, hipEventCreateWithFlags(&s->ev_pre, hipEventDisableTiming));
    TSD_HIP_CHECK(ctx, hipMemcpyAsync(s->d_pre, s->h_pre, L.in_bytes, hipMemcpyHostToDevice, ctx->stream2));
    TSD_HIP_CHECK(ctx, hipEventRecord(s->ev_pre, ctx->stream2));
    (void)hipStreamQuery(ctx->stream2);
    s->pre_copied = true;
  }
  s->pre_armed = true;
  return TSD_OK;
}

namespace tsd {
// the four kernels' arguments of one sensor's armed pre-registration (its layout in s->d_pre, the ray cast's outputs, the pose)
struct PreLaunch {
  PdfNormalsEntry normals; PdfPrepareArgs prepare; PdfScoreEntry score; PdfArgmaxEntry argmax;
  size_t prep_lds; int score_blocks;
};
static PreLaunch pre_launch_args(const tsd_sensor* s, const double* d_coords, const uint8_t* d_mask_m, const double* d_pose6)
{
  const tsd_sensor::PreLayout& L = s->pre;
  char* d = s->d_pre;
  char* h_dev = s->h_pre_dev;               // the pinned buffer as the device sees it (header + result are written there by k_pdf_argmax)
  const int n = L.n, SR = 10 / 2;
  PreLaunch pl;
  pl.normals.set0 = PdfNormalsSet{d_coords, d_mask_m, d_mask_m, reinterpret_cast<uint8_t*>(d + L.off_mo_m), reinterpret_cast<double*>(d + L.off_phi_m)};
  pl.normals.set1 = PdfNormalsSet{reinterpret_cast<const double*>(d + L.off_S), reinterpret_cast<const uint8_t*>(d + L.off_ms),
                                  reinterpret_cast<const uint8_t*>(d + L.off_msp), reinterpret_cast<uint8_t*>(d + L.off_mo_s),
                                  reinterpret_cast<double*>(d + L.off_phi_s)};
  pl.normals.points = n; pl.normals.sr = SR;
  PdfPrepareArgs& pa = pl.prepare;
  pa.mask_m = reinterpret_cast<const uint8_t*>(d + L.off_mo_m); pa.mask_s = reinterpret_cast<const uint8_t*>(d + L.off_mo_s);
  pa.phi_m = reinterpret_cast<const double*>(d + L.off_phi_m); pa.phi_s = reinterpret_cast<const double*>(d + L.off_phi_s);
  pa.S = reinterpret_cast<const double*>(d + L.off_S);
  pa.draws_control = reinterpret_cast<const int*>(d + L.off_dc); pa.draws_trials = reinterpret_cast<const int*>(d + L.off_dt);
  pa.control = reinterpret_cast<double2*>(d + L.off_C); pa.cand = reinterpret_cast<PdfCandidate*>(d + L.off_K);
  pa.hdr = reinterpret_cast<PdfHeader*>(d + L.off_hdr);
  pa.n = n; pa.sr = SR; pa.span = L.span; pa.trials_cfg = L.trials; pa.size_control_set = L.size_control_set; pa.max_cand = L.max_cand;
  pa.phi_max = L.phi_max;
  pl.prep_lds = (2 * (size_t)n + (size_t)PDF_MAX_TRIALS) * sizeof(double);
  pl.score = PdfScoreEntry{d_pose6, d_coords, reinterpret_cast<const double*>(d + L.off_S), reinterpret_cast<const double2*>(d + L.off_C),
                           reinterpret_cast<const PdfCandidate*>(d + L.off_K), reinterpret_cast<double*>(d + L.off_prob),
                           reinterpret_cast<const PdfHeader*>(d + L.off_hdr), L.zrand, L.max_cand, std::max(L.size_control_set, 1)};
  pl.score_blocks = std::min((L.max_cand + PDF_WAVES - 1) / PDF_WAVES, PDF_SCORE_GRID);
  pl.argmax = PdfArgmaxEntry{reinterpret_cast<const double*>(d + L.off_prob), reinterpret_cast<const PdfCandidate*>(d + L.off_K), d_coords,
                             reinterpret_cast<const double*>(d + L.off_S), reinterpret_cast<PdfResult*>(d + L.off_res),
                             reinterpret_cast<const PdfHeader*>(d + L.off_hdr), reinterpret_cast<PdfHeader*>(h_dev + L.off_hdr),
                             reinterpret_cast<PdfResult*>(h_dev + L.off_res), L.max_cand};
  return pl;
}
// the inputs of an armed pre-registration must be on the device before its kernels: copied at arm time on the side stream (waited for
// by event) or, where that was switched off, here
static int pre_inputs_ready(tsd_ctx* ctx, tsd_sensor* s, hipStream_t stream)
{
  if (s->pre_direct) return TSD_OK;
  if (!s->pre_copied) TSD_HIP_CHECK(ctx, hipMemcpyAsync(s->d_pre, s->h_pre, s->pre.in_bytes, hipMemcpyHostToDevice, stream));
  else if (hipEventQuery(s->ev_pre) != hipSuccess) {
    (void)hipGetLastError();                 // (hipErrorNotReady is sticky as "last error")
    TSD_HIP_CHECK(ctx, hipStreamWaitEvent(stream, s->ev_pre, 0));
  }
  return TSD_OK;
}
static int pre_configure_lds(tsd_ctx* ctx, const void* kernel, size_t prep_lds)
{
  // (k_pdf_prepare's static LDS is ~33 KB; beyond ~3 000 beams the dynamic part needs the attribute)
  std::lock_guard<std::mutex> lk_misc(ctx->misc_mutex);
  size_t& configured = ctx->lds_configured[kernel];
  if (prep_lds > configured) {
    TSD_HIP_CHECK(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prep_lds));
    configured = prep_lds;
  }
  return TSD_OK;
}

int launch_preregistration(tsd_ctx* ctx, tsd_sensor* s, hipStream_t stream, const double* d_coords, const uint8_t* d_mask_m,
                           const double* d_pose6, const double** tinit_dev, hipEvent_t before_score, IcpPreLaunch* fold)
{
  const tsd_sensor::PreLayout& L = s->pre;
  if (int rc = pre_inputs_ready(ctx, s, stream)) return rc;
  const PreLaunch pl = pre_launch_args(s, d_coords, d_mask_m, d_pose6);
  if (int rc = pre_configure_lds(ctx, reinterpret_cast<const void*>(k_pdf_prepare), pl.prep_lds)) return rc;
  {
    ScopedKernelTimer t(ctx, "tsdpdf", true);
    hipLaunchKernelGGL(k_pdf_normals, dim3((2 * L.n + 255) / 256, 2), dim3(256), 0, stream, pl.normals.set0, pl.normals.set1, L.n, pl.normals.sr);
    hipLaunchKernelGGL(k_pdf_prepare, dim3(1), dim3(1024), pl.prep_lds, stream, pl.prepare);
    // (asynchronous mapping: the scoring is the first kernel of the chain that reads the grid)
    if (before_score) TSD_HIP_CHECK(ctx, hipStreamWaitEvent(stream, before_score, 0));
    hipLaunchKernelGGL(k_pdf_score, dim3(pl.score_blocks), dim3(64 * PDF_WAVES), 0, stream, ctx->grid, pl.score.pose, pl.score.M, pl.score.S,
                       pl.score.control, 0, pl.score.cand, 0, pl.score.zrand, pl.score.prob, pl.score.hdr, pl.score.max_cand, pl.score.control_alloc);
    // (the arg-max's own completion is the event a pre-registration armed AHEAD waits for before it overwrites the inputs: no marker
    // between this kernel and the registration)
    if (!s->ev_pre_done) TSD_HIP_CHECK(ctx, hipEventCreateWithFlags(&s->ev_pre_done, hipEventDisableTiming | hipEventDisableSystemFence));
    if (fold) {
      // the arg-max rides as the first workgroup of the registration's launch (k_icp_pre); its outcome is announced through a flag word
      if (!s->d_pre_flag) {
        TSD_HIP_CHECK(ctx, hipMalloc(&s->d_pre_flag, 128));
        TSD_HIP_CHECK(ctx, hipMemsetAsync(s->d_pre_flag, 0, 128, stream));
      }
      if (++s->pre_seq == 0u) ++s->pre_seq;
      fold->dev.am = pl.argmax; fold->dev.flag = s->d_pre_flag; fold->dev.seq = s->pre_seq; fold->done = s->ev_pre_done;
    } else
    hipExtLaunchKernelGGL(k_pdf_argmax, dim3(1), dim3(1024), 0, stream, nullptr, s->ev_pre_done, 0, pl.argmax.prob, pl.argmax.cand, pl.argmax.max_cand, pl.argmax.M,
                       pl.argmax.S, pl.argmax.out, pl.argmax.hdr, pl.argmax.host_hdr, pl.argmax.host_res);
    s->pre_done_valid = true;
    s->pre_res_off_hdr = L.off_hdr; s->pre_res_off_res = L.off_res;
  }
  TSD_HIP_CHECK(ctx, hipGetLastError());
  *tinit_dev = reinterpret_cast<const double*>(s->d_pre + L.off_res);
  return TSD_OK;
}

// The armed pre-registrations of a batch's robots (tsd_batch_begin) in four launches instead of four per robot; each robot's result
// lands where its own launch_preregistration would have put it (s->d_pre + off_res: the Tinit of its registration).
int launch_preregistration_batch(tsd_ctx* ctx, hipStream_t stream, tsd_sensor* const* sensors, int n)
{
  for (int i0 = 0; i0 < n; i0 += PDF_BATCH_BYVAL) {
    const int m = std::min(n - i0, PDF_BATCH_BYVAL);
    PdfNormalsBatch bn{}; PdfPrepareBatch bp{}; PdfScoreBatch bs{}; PdfArgmaxBatch ba{};
    int max_points = 0, max_blocks = 1;
    size_t max_lds = 0;
    for (int i = 0; i < m; i++) {
      tsd_sensor* s = sensors[i0 + i];
      if (int rc = pre_inputs_ready(ctx, s, stream)) return rc;
      const PreLaunch pl = pre_launch_args(s, s->d_coords, s->d_mask_m, s->d_state->icpP);
      bn.e[i] = pl.normals; bp.e[i] = pl.prepare; bs.e[i] = pl.score; ba.e[i] = pl.argmax;
      max_points = std::max(max_points, pl.normals.points); max_blocks = std::max(max_blocks, pl.score_blocks); max_lds = std::max(max_lds, pl.prep_lds);
      s->pre_done_valid = false;              // (no event of this sensor's own behind the batched arg-max; batches are armed between scans)
      s->pre_res_off_hdr = s->pre.off_hdr; s->pre_res_off_res = s->pre.off_res;
    }
    if (int rc = pre_configure_lds(ctx, reinterpret_cast<const void*>(k_pdf_prepare_batch), max_lds)) return rc;
    // (the scoring strides a fixed grid over each robot's candidates: a share of the single launch's grid per robot keeps the
    // batch's workgroups at about that launch's number)
    const int blocks = std::max(64, std::min(max_blocks, PDF_SCORE_GRID / std::max(1, m / 2)));
    ScopedKernelTimer t(ctx, "tsdpdf", true);
    hipLaunchKernelGGL(k_pdf_normals_batch, dim3((2 * max_points + 255) / 256, 2, m), dim3(256), 0, stream, bn);
    hipLaunchKernelGGL(k_pdf_prepare_batch, dim3(m), dim3(1024), max_lds, stream, bp);
    hipLaunchKernelGGL(k_pdf_score_batch, dim3(blocks, 1, m), dim3(64 * PDF_WAVES), 0, stream, ctx->grid, bs);
    hipLaunchKernelGGL(k_pdf_argmax_batch, dim3(m), dim3(1024), 0, stream, ba);
    TSD_HIP_CHECK(ctx, hipGetLastError());
  }
  return TSD_OK;
}
}  // namespace tsd

extern "C" int tsd_scan_preregistration_result(tsd_sensor* s, tsd_tsdpdf_result* result)
{
  if (!s || !s->ctx || !result) return TSD_E_ARG;
  if (!s->pre_ran || s->submitted) return set_error(s->ctx, TSD_E_ARG, "tsd_scan_preregistration_result: no collected scan with a pre-registration", hipSuccess);
  const PdfHeader* hd = reinterpret_cast<const PdfHeader*>(s->h_pre + s->pre_res_off_hdr);
  const PdfResult* r = reinterpret_cast<const PdfResult*>(s->h_pre + s->pre_res_off_res);
  std::memcpy(result->T, r->T, sizeof(r->T));
  result->probability = r->prob; result->idx_model = r->idx; result->idx_scene = r->i;
  result->candidates = hd->identity ? 0 : hd->n_cand;
  result->valid_model = hd->n_model_valid; result->valid_scene = hd->n_scene_valid; result->control_points = hd->n_control;
  result->reserved = 0;
  return TSD_OK;
}
