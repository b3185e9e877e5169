// experiment: issue cost of the instruction kinds the ICP step is made of, for a wave that has its SIMD to itself and for two waves
// that share one (waves i and i + 4 of a workgroup, tools/exp/hwid.hip).  Eight independent chains per kind, so that latency is
// covered and the figure is the issue interval; cycles per instruction per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
template <int MODE>
__global__ void k(long long* out, double seed)
{
  const int tid = threadIdx.x, wave = tid >> 6;
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed * (i + 1) + tid * 1e-3;
  double m = 1.0000001, c = 1e-9;
  asm volatile("" : "+v"(m), "+v"(c));
  int acc = 0;
  __syncthreads();
  const long long c0 = clock64();
#pragma unroll 4
  for (int it = 0; it < N / 8; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) a[i] = __builtin_fma(a[i], m, c);                       // v_fma_f64
      if (MODE == 1) a[i] = a[i] + c;                                        // v_add_f64
      if (MODE == 2) a[i] = a[i] * m;                                        // v_mul_f64
      if (MODE == 3) { asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(acc) : "v"(a[i]), "v"(m), "v"(tid) : "vcc"); }   // cmp + cndmask (2 instr)
      if (MODE == 4) { float f = (float)a[i]; f = __builtin_fmaf(f, 1.0000001f, 1e-9f); a[i] = f; }   // cvt + fma32 + cvt
      if (MODE == 5) a[i] = __builtin_amdgcn_rcp(a[i]);                      // v_rcp_f64
      if (MODE == 6) a[i] = __builtin_amdgcn_rsq(a[i]);                      // v_rsq_f64
      if (MODE == 7) a[i] = fmax(a[i], c);                                   // v_max_f64
      if (MODE == 8) { int lo = __double2loint(a[i]); asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(lo)); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
      if (MODE == 9) { asm volatile("s_and_b64 vcc, vcc, exec\n s_or_b64 vcc, vcc, exec" ::: "vcc"); }   // 2 SALU
    }
  }
  const long long c1 = clock64();
  double s = 0; for (int i = 0; i < 8; i++) s += a[i];
  if ((tid & 63) == 0) { out[2 * wave] = c1 - c0; out[2 * wave + 1] = (long long)s + acc; }
}
template <int MODE> void run(long long* d, const char* name, int per)
{
  long long h[32];
  for (int threads : {64, 320, 384, 512}) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, 1.0); (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); }
    printf("%-28s threads %3d  cycles per instruction, by wave:", name, threads);
    for (int w = 0; w < threads / 64; w++) printf(" %5.2f", (double)h[2 * w] / (N * per));
    printf("\n");
  }
}
int main()
{
  setvbuf(stdout, NULL, _IONBF, 0);
  long long* d; (void)hipMalloc(&d, 256);
  run<0>(d, "v_fma_f64", 1); run<1>(d, "v_add_f64", 1); run<2>(d, "v_mul_f64", 1); run<3>(d, "v_cmp_f64 + v_cndmask_b32", 2);
  run<4>(d, "cvt + v_fma_f32 + cvt", 3); run<5>(d, "v_rcp_f64", 1); run<6>(d, "v_rsq_f64", 1); run<7>(d, "v_max_f64", 1);
  run<8>(d, "v_mov_b32_dpp", 1); run<9>(d, "s_and_b64 + s_or_b64", 2);
  return 0;
}
