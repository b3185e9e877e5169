// experiment: what the control constructs of k_icp's step cost a wave -- an exec-masked region (s_and_saveexec / s_or), a never-taken
// and a taken branch on a vector compare, a scalar branch on v_readfirstlane, an LDS add by one lane inside a region -- beside the
// plain fp64 instruction (tools/exp/valu.hip), for a wave alone on its SIMD and for the older / younger of two waves on one.
// Each pattern sits between two independent v_fma_f64 of eight chains; the figure is cycles per PATTERN (fma included) per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N 1024
template <int MODE>
__global__ void k(long long* out, double seed, int* sink)
{
  __shared__ int lds[64];
  const int tid = threadIdx.x, wave = tid >> 6;
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed * (i + 1) + tid * 1e-3;
  double m = 1.0000001, c = 1e-9, big = 1e300;
  asm volatile("" : "+v"(m), "+v"(c), "+v"(big));
  asm volatile("s_mov_b64 s[22:23], exec\n s_mov_b32 s24, 0" ::: "s22", "s23", "s24");
  int acc = tid, zero = 0;
  asm volatile("" : "+v"(zero));
  if (tid < 64) lds[tid] = 0;
  __syncthreads();
  const long long c0 = clock64();
#pragma unroll 2
  for (int it = 0; it < N / 8; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      a[i] = __builtin_fma(a[i], m, c);
      if (MODE == 1) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, s[20:21]" : "+v"(acc) : "v"(a[i]), "v"(big) : "vcc", "s20", "s21");
      if (MODE == 2) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n s_cbranch_vccnz 1f\n v_add_u32 %0, %0, 1\n 1:" : "+v"(acc) : "v"(a[i]), "v"(big) : "vcc");          // never taken (falls through)
      if (MODE == 3) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n s_cbranch_vccnz 1f\n v_add_u32 %0, %0, 1\n 1:" : "+v"(acc) : "v"(a[i]), "v"(big) : "vcc");          // always taken (jumps over one instruction)
      if (MODE == 4) asm volatile("v_readfirstlane_b32 s20, %1\n s_cmp_eq_u32 s20, 0x7fffffff\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, 1\n 1:" : "+v"(acc) : "v"(acc) : "s20", "scc");
      if (MODE == 5) asm volatile("v_cmp_eq_u32 vcc, 0, %1\n s_and_saveexec_b64 s[20:21], vcc\n ds_add_u32 %2, %0\n s_or_b64 exec, exec, s[20:21]" : : "v"(acc), "v"(tid & 63), "v"(zero) : "vcc", "s20", "s21", "memory");
      if (MODE == 6) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, 1\n 1: s_or_b64 exec, exec, s[20:21]" : "+v"(acc) : "v"(a[i]), "v"(big) : "vcc", "s20", "s21");
      if (MODE == 7) asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1" : "+v"(acc));     // three dependent 32-bit adds
      if (MODE == 8) asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1" ::: "s20", "scc");
      if (MODE == 10) asm volatile("s_and_saveexec_b64 s[20:21], s[22:23]\n v_nop\n s_or_b64 exec, exec, s[20:21]" ::: "s20", "s21");                   // region on a mask that has long been there
      if (MODE == 11) asm volatile("s_cmp_eq_u32 s24, 0x7fffffff\n s_cbranch_scc1 1f\n v_nop\n 1:" ::: "scc");                                             // scalar compare + branch, never taken
      if (MODE == 12) asm volatile("s_cmp_lg_u32 s24, 0x7fffffff\n s_cbranch_scc1 1f\n v_nop\n 1:" ::: "scc");                                             // ... always taken
      if (MODE == 13) asm volatile("s_and_saveexec_b64 s[20:21], s[22:23]\n s_cbranch_execz 1f\n v_nop\n 1: s_or_b64 exec, exec, s[20:21]" ::: "s20", "s21");  // region + skip branch, mask ready
      if (MODE == 14) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_nop\n s_or_b64 exec, exec, s[20:21]" : : "v"(a[i]), "v"(big) : "vcc", "s20", "s21");
      if (MODE == 15) asm volatile("v_cmp_gt_f64 vcc, %0, %1\n s_cbranch_vccnz 1f\n v_nop\n 1:" : : "v"(a[(i + 1) & 7]), "v"(big) : "vcc");    // the compare on a value that is four instructions old
      if (MODE == 16) asm volatile("v_cmp_gt_f64 s[20:21], %0, %1\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n s_and_b64 vcc, exec, s[20:21]\n s_cbranch_vccnz 1f\n v_nop\n 1:" : : "v"(a[i]), "v"(big) : "vcc", "s20", "s21", "scc");    // six instructions between compare and branch
      if (MODE == 9) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(acc) : "v"(a[i]), "v"(big), "v"(tid) : "vcc");
    }
  }
  const long long c1 = clock64();
  double s = 0; for (int i = 0; i < 8; i++) s += a[i];
  if ((tid & 63) == 0) { out[2 * wave] = c1 - c0; out[2 * wave + 1] = (long long)s + acc; sink[wave] = lds[0]; }
}
template <int MODE> void run(long long* d, int* sink, const char* name)
{
  long long h[32];
  for (int threads : {64, 512}) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, 1.0, sink); (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); }
    printf("%-58s threads %3d  cycles per pattern, by wave:", name, threads);
    for (int w = 0; w < threads / 64; w++) printf(" %5.1f", (double)h[2 * w] / N);
    printf("\n");
  }
}
int main(int argc, char** argv)
{
  setvbuf(stdout, NULL, _IONBF, 0);
  long long* d; (void)hipMalloc(&d, 256);
  int* sink; (void)hipMalloc(&sink, 256);
  const int only = argc > 1 ? atoi(argv[1]) : -1;      // (one pattern per process: gpurun -- 'for m in 0 1 ..; do timeout 20 tools/exp/ctl $m; done')
  if (only < 0 || only == 0) run<0>(d, sink, "v_fma_f64 alone");
  if (only < 0 || only == 1) run<1>(d, sink, "+ v_cmp, s_and_saveexec, v_add, s_or exec");
  if (only < 0 || only == 6) run<6>(d, sink, "+ v_cmp, s_and_saveexec, s_cbranch_execz, v_add, s_or exec");
  if (only < 0 || only == 2) run<2>(d, sink, "+ v_cmp, s_cbranch_vccnz (never taken), v_add");
  if (only < 0 || only == 3) run<3>(d, sink, "+ v_cmp, s_cbranch_vccnz (always taken, skips 1)");
  if (only < 0 || only == 4) run<4>(d, sink, "+ v_readfirstlane, s_cmp, s_cbranch_scc1 (never), v_add");
  if (only < 0 || only == 5) run<5>(d, sink, "+ v_cmp, s_and_saveexec, ds_add_u32 (lane 0), s_or exec");
  if (only < 0 || only == 7) run<7>(d, sink, "+ 3 dependent v_add_u32");
  if (only < 0 || only == 8) run<8>(d, sink, "+ 3 dependent s_add_u32");
  if (only < 0 || only == 9) run<9>(d, sink, "+ v_cmp_f64, v_cndmask_b32");
  if (only == 10) run<10>(d, sink, "+ s_and_saveexec (mask ready), v_nop, s_or exec");
  if (only == 11) run<11>(d, sink, "+ s_cmp, s_cbranch_scc1 (never), v_nop");
  if (only == 12) run<12>(d, sink, "+ s_cmp, s_cbranch_scc1 (always, skips 1)");
  if (only == 13) run<13>(d, sink, "+ s_and_saveexec (mask ready), s_cbranch_execz (never), v_nop, s_or");
  if (only == 14) run<14>(d, sink, "+ v_cmp, s_and_saveexec vcc, v_nop, s_or exec");
  if (only == 15) run<15>(d, sink, "+ v_cmp (old operand), s_cbranch_vccnz (never), v_nop");
  if (only == 16) run<16>(d, sink, "+ v_cmp, 6 v_nop, s_and, s_cbranch_vccnz (never), v_nop");
  return 0;
}
