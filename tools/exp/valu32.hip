// experiment: issue cost of the 32-bit instruction kinds a single-precision candidate filter would be made of (tools/exp/valu.hip has the
// fp64 ones): eight independent chains per kind, cycles per instruction per wave, for one wave per SIMD and for two.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(long long* out, float seed)
{
  const int tid = threadIdx.x, wave = tid >> 6;
  float a[8]; f2 p[8];
  for (int i = 0; i < 8; i++) { a[i] = seed * (i + 1) + tid * 1e-3f; p[i] = f2{a[i], a[i] * 0.5f}; }
  float m = 1.0000001f, c = 1e-9f; f2 pm = {m, c};
  asm volatile("" : "+v"(m), "+v"(c), "+v"(pm));
  __syncthreads();
  const long long c0 = clock64();
#pragma unroll 4
  for (int it = 0; it < N / 8; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);                                           // v_fma_f32
      if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));              // v_pk_add_f32
      if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));              // v_pk_mul_f32
      if (MODE == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));     // v_med3_f32
      if (MODE == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));                  // v_min_f32
      if (MODE == 5) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));   // v_and_or_b32
      if (MODE == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));                  // v_add_f32
      if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pm));          // v_pk_fma_f32
      if (MODE == 8) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));     // v_min3_f32
    }
  }
  const long long c1 = clock64();
  float s = 0; for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  if ((tid & 63) == 0) { out[2 * wave] = c1 - c0; out[2 * wave + 1] = (long long)s; }
}
template <int MODE> void run(long long* d, const char* name)
{
  long long h[32];
  for (int threads : {64, 512}) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, 1.0f); (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); }
    printf("%-16s threads %3d  cycles per instruction, by wave:", name, threads);
    for (int w = 0; w < threads / 64; w++) printf(" %5.2f", (double)h[2 * w] / N);
    printf("\n");
  }
}
int main()
{
  setvbuf(stdout, NULL, _IONBF, 0);
  long long* d; (void)hipMalloc(&d, 256);
  run<0>(d, "v_fma_f32"); run<1>(d, "v_pk_add_f32"); run<2>(d, "v_pk_mul_f32"); run<3>(d, "v_med3_f32"); run<4>(d, "v_min_f32");
  run<5>(d, "v_and_or_b32"); run<6>(d, "v_add_f32"); run<7>(d, "v_pk_fma_f32"); run<8>(d, "v_min3_f32");
  return 0;
}
