// experiment: how much straight-line code stays warm?  A body of S KB (independent fp64 fma chains, 8 bytes per instruction) is run
// six times in a row by one workgroup; cycles per instruction of each pass.  If the body fits the instruction cache, passes 2.. run at
// the issue rate (4.9 cycles per fp64 instruction for a wave that has its SIMD to itself); beyond it every pass fetches again.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Unroll {
  __device__ static __forceinline__ void run(double (&a)[8], double m, double c) {
    Unroll<N - 1>::run(a, m, c);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = __builtin_fma(a[i], m, c);
  }
};
template <> struct Unroll<0> { __device__ static __forceinline__ void run(double (&)[8], double, double) {} };
template <int KB>
__global__ void k(long long* out, double seed)
{
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed * (i + 1) + threadIdx.x * 1e-3;
  double m = 1.0000001, c = 1e-9;
  asm volatile("" : "+v"(m), "+v"(c));
  long long t[7];
  t[0] = clock64();
#pragma unroll 1
  for (int rep = 0; rep < 6; rep++) {
    Unroll<KB * 1024 / 64>::run(a, m, c);          // 8 fma of 8 bytes per level
    asm volatile("" : "+v"(a[0]));
    t[rep + 1] = clock64();
  }
  double s = 0; for (int i = 0; i < 8; i++) s += a[i];
  if (threadIdx.x == 0) { for (int i = 0; i < 6; i++) out[i] = t[i + 1] - t[i]; out[6] = (long long)s; }
}
template <int KB> void run(long long* d)
{
  long long h[7];
  for (int threads : {64, 512}) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<KB>, dim3(1), dim3(threads), 0, 0, d, 1.0); (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); }
    printf("body %3d KB, %3d threads: cycles per instruction of passes 1..6:", KB, threads);
    for (int i = 0; i < 6; i++) printf(" %5.2f", (double)h[i] / (KB * 1024 / 8));
    printf("\n");
  }
}
int main()
{
  setvbuf(stdout, NULL, _IONBF, 0);
  long long* d; (void)hipMalloc(&d, 256);
  run<8>(d); run<16>(d); run<24>(d); run<32>(d); run<40>(d); run<48>(d); run<56>(d); run<64>(d); run<80>(d); run<96>(d); run<128>(d);
  return 0;
}
