// tsdpdf_device.hpp -- what the TSD_PDF pre-registration's kernels (tsdpdf.hip) and the registration kernel (icp_kernels.hip) share: the
// candidate / header / result records and the arg-max over the scored candidates -- a kernel of its own (k_pdf_argmax) in the unfused
// and batched paths, the FIRST WORKGROUP of the registration's launch in the fused scan (k_icp_pre: the registration needs Tinit only
// ~2.5 us into its set-up, which is about what the arg-max takes).
#pragma once
#include "tsd_ctx.hpp"
#include "tsd_device.hpp"

namespace tsd {

// a candidate (idx, i) of trial t: `ti` = t << 12 | i -- its place in the reference's serial order (trial-major, scene index ascending),
// which is what the arg-max breaks ties on; the LIST order is free (k_pdf_prepare writes it in one pass, in whatever order its waves arrive)
struct PdfCandidate { int idx, ti; double phi; };
constexpr int PDF_I_BITS = 12, PDF_I_MASK = (1 << PDF_I_BITS) - 1;
static_assert(TSD_MAX_BEAMS <= (1 << PDF_I_BITS), "scene index field of PdfCandidate::ti");
struct PdfResult { double T[9]; double prob; int idx, i, candidates, pad; };
// what k_pdf_prepare (the fused scan's device-side list building) leaves for the scoring and arg-max kernels, which the host then
// launches without knowing the counts
struct PdfHeader { int n_cand, n_control, n_model_valid, n_scene_valid, identity, pad[3]; };


// arguments of the arg-max (by value in k_pdf_argmax_batch / k_icp_pre)
struct PdfArgmaxEntry { const double* prob; const PdfCandidate* cand; const double* M; const double* S; PdfResult* out; const PdfHeader* hdr; PdfHeader* host_hdr; PdfResult* host_res; int max_cand; };
// ... and, inside the registration's launch, where its outcome is announced: twelve tagged granules {seq, half a double} of TBest's two
// rows at `flag` (128 bytes)
struct IcpPreArgs { PdfArgmaxEntry am; unsigned int* flag; unsigned int seq; };
struct IcpPreLaunch { IcpPreArgs dev; hipEvent_t done; };        // (host side: + the event to record at the launch's completion)

// wave reductions for the arg-max: DPP row shifts inside the 16-lane rows, the four row results through scalar registers
__device__ __forceinline__ double pdf_wave_max_nonneg(double v)            // v >= 0 in every lane (lanes without a source read 0)
{
#define PDF_SHR_F64(CTRL) { const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false), \
                                      hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false); v = fmax(v, __hiloint2double(hi_, lo_)); }
  PDF_SHR_F64(0x111) PDF_SHR_F64(0x112) PDF_SHR_F64(0x114) PDF_SHR_F64(0x118)
#undef PDF_SHR_F64
  auto rl = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
  return fmax(fmax(rl(15), rl(31)), fmax(rl(47), rl(63)));
}
__device__ __forceinline__ int pdf_wave_min_int(int v)
{
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xf, false));
  return min(min(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)), min(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}

// first candidate in the reference's serial trial / i order (the key `ti`) that reaches the maximum; bestProb starts at
// 0.0 and is replaced on `>` only (TSD_PDFMatching.cpp:188,264)
// SPEC: candidates per thread requested ahead of the header's counts (1: the kernel of its own, 1 024 threads; 3: inside the
// registration's launch, 512 threads -- 1 536 of the usual ~1 300 candidates, so that the header's round trip is theirs too)
template <int SPEC = 1>
__device__ __forceinline__ void
pdf_argmax_body(const double* __restrict__ prob, const PdfCandidate* __restrict__ cand, int n_cand, const double* __restrict__ M,
                const double* __restrict__ S, PdfResult* __restrict__ out, const PdfHeader* __restrict__ hdr,
                PdfHeader* __restrict__ host_hdr, PdfResult* __restrict__ host_res /* fused scan: pinned host memory, or nullptr */,
                unsigned int* __restrict__ publish_flag = nullptr, unsigned int publish_seq = 0u /* inside the registration's launch */)
{
  // A chain of dependent memory round trips by nature (header -> probabilities and candidates -> the winner's points); round 6 took one
  // of them out: a thread keeps its best candidate's model index and angle in registers (the winner used to read its candidate again),
  // and the reduction is one shuffle tree per wave + one barrier (it was ten barriers).
  // (with a header, n_cand arrives as the ALLOCATION's size: the thread's first candidate is requested ahead of the header's counts)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = (int)blockDim.x, W = T >> 6;                 // (1 024 threads as a kernel of its own, the registration's 512 inside its launch)
  const int n_alloc = hdr ? n_cand : 0;
  double p_sp[SPEC], phi_sp[SPEC]; int idx_sp[SPEC], ti_sp[SPEC];
#pragma unroll
  for (int j = 0; j < SPEC; j++) {
    const int c = tid + j * T, cc = c < n_alloc ? c : 0;
    p_sp[j] = 0.0; phi_sp[j] = 0.0; idx_sp[j] = 0; ti_sp[j] = 0;
    if (n_alloc > 0) { p_sp[j] = ld_pinned(&prob[cc]); idx_sp[j] = ld_pinned(&cand[cc].idx); ti_sp[j] = ld_pinned(&cand[cc].ti); phi_sp[j] = ld_pinned(&cand[cc].phi); }
  }
  if (hdr) n_cand = hdr->identity ? 0 : hdr->n_cand;
  __shared__ double s_p[16];
  __shared__ int s_o[16], s_t[16];                          // per wave: serial-order key and thread of its best
  double bp = 0.0, bphi = 0.0; int bk = -1, bo = 0x7fffffff, bidx = 0;
#pragma unroll
  for (int j = 0; j < SPEC; j++) {
    const int c = tid + j * T;
    if (c < n_cand && c < n_alloc) {
      const double p = p_sp[j];
      if (p > bp || (p == bp && p > 0.0 && ti_sp[j] < bo)) { bp = p; bk = c; bo = ti_sp[j]; bidx = idx_sp[j]; bphi = phi_sp[j]; }
    }
  }
  for (int c = tid + (n_alloc > 0 ? SPEC * T : 0); c < n_cand; c += T) {
    const double p = prob[c];
    const PdfCandidate cd = cand[c];
    if (p > bp || (p == bp && p > 0.0 && cd.ti < bo)) { bp = p; bk = c; bo = cd.ti; bidx = cd.idx; bphi = cd.phi; }
  }
  const int bci = bk >= 0 ? (bo & PDF_I_MASK) : 0;
  // the wave's best: the largest probability (DPP row shifts, no LDS), then the earliest serial-order key among the lanes that hold it
  // (a shuffle tree over the triple cost six dependent LDS-crossbar round trips: 1.1 us of this 5 us kernel)
  const double wp = pdf_wave_max_nonneg(bp);
  const bool cont = bk >= 0 && bp == wp;
  const int wo = pdf_wave_min_int(cont ? bo : 0x7fffffff);
  const unsigned long long wb = __ballot(cont && bo == wo);
  const int wt = wb ? wave * 64 + (__ffsll((long long)wb) - 1) : -1;      // (keys are unique: one lane)
  if (lane == 0) { s_p[wave] = wp; s_o[wave] = wo; s_t[wave] = wt; }
  __syncthreads();
  double gp = s_p[0]; int go = s_o[0], gt = s_t[0];
  for (int w = 1; w < W; w++) {
    const double p2 = s_p[w]; const int o2 = s_o[w], t2 = s_t[w];
    if (t2 >= 0 && (p2 > gp || (p2 == gp && (gt < 0 || o2 < go)))) { gp = p2; go = o2; gt = t2; }
  }
  const bool found = gt >= 0 && gp > 0.0;
  if (tid == (found ? gt : 0)) {
    PdfResult r;
    for (int i = 0; i < 9; i++) r.T[i] = (i % 4 == 0) ? 1.0 : 0.0;
    r.prob = 0.0; r.idx = -1; r.i = -1; r.candidates = n_cand; r.pad = 0;
    if (found) {
      // (the winner's two points: requested by the winner alone, ahead of the sine and cosine.  Requested by every thread for its own
      // best ahead of the reduction -- 4 000 scattered 8-byte reads -- the kernel took 6.2 us instead of 5.2)
      const double msx = ld_pinned(&M[2 * bidx]), msy = ld_pinned(&M[2 * bidx + 1]), ssx = ld_pinned(&S[2 * bci]), ssy = ld_pinned(&S[2 * bci + 1]);
      const double co = cos(bphi), si = sin(bphi);
      r.T[0] = co; r.T[1] = -si; r.T[3] = si; r.T[4] = co;
      r.T[2] = msx - (co * ssx + (-si) * ssy);
      r.T[5] = msy - (si * ssx + co * ssy);
      r.prob = gp; r.idx = bidx; r.i = bci;
    }
    *out = r;
    if (publish_flag) {
      // inside the registration's launch: TBest's two rows are what the other workgroups of the launch wait for.  They travel as twelve
      // tagged 8-byte GRANULES {launch number, half a double}, each ONE relaxed agent-scope store (write-through, carries its own tag:
      // no flag, no drain between rows and flag, no second read behind the flag -- the hand-off form of the registration's helpers)
      unsigned long long* gr = reinterpret_cast<unsigned long long*>(publish_flag);
      const unsigned long long tag = (unsigned long long)publish_seq << 32;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(r.T[i]);
        __hip_atomic_store(gr + 2 * i, tag | (b & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gr + 2 * i + 1, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // fused scan: header and result go to the host from here (stores into pinned memory, complete when the kernel ends) -- a copy
    // behind this kernel would sit between it and the registration (a blit kernel: ~8 us with its hand-offs).  Inside the
    // registration's launch the host reads them when that launch's RESULT RECORD has arrived, i.e. before the kernel has ended:
    // system-scope (write-through) stores then, issued ~100 us ahead of that record's.
    if (host_res) {
      if (publish_flag) {
        static_assert(sizeof(PdfResult) % 8 == 0 && sizeof(PdfHeader) % 8 == 0, "records as 8-byte words");
        const PdfHeader hh = *hdr;
        const unsigned long long* wr = reinterpret_cast<const unsigned long long*>(&r);
        const unsigned long long* wh = reinterpret_cast<const unsigned long long*>(&hh);
        for (int i = 0; i < (int)(sizeof(PdfResult) / 8); i++) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_res) + i, wr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int i = 0; i < (int)(sizeof(PdfHeader) / 8); i++) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_hdr) + i, wh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else { *host_res = r; *host_hdr = *hdr; }
    }
  }
}


}  // namespace tsd
