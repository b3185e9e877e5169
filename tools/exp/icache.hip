// experiment: cost of executing straight-line code for the first time (instruction fetch from L2/HBM) vs again
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Unroll { __device__ static __forceinline__ void run(double& x, double& y, double& z) { Unroll<N - 1>::run(x, y, z); x = fma(x, y, z); y = fma(y, z, x); z = fma(z, x, y); } };
template <> struct Unroll<0> { __device__ static __forceinline__ void run(double&, double&, double&) {} };
__global__ void k(long long* out, double seed)
{
  double x = seed + threadIdx.x, y = seed * 0.5, z = seed * 1e-9;
  long long t[5];
  t[0] = clock64();
  for (int rep = 0; rep < 4; rep++) {
    Unroll<700>::run(x, y, z);        // 2100 fp64 fma, register operands: ~17 KB of code
    t[rep + 1] = clock64();
  }
  if (threadIdx.x == 0) { for (int i = 0; i < 4; i++) out[i] = t[i + 1] - t[i]; out[4] = (long long)(x + y + z); }
}
__global__ void filler(double* p, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = p[i] * 0.5 + 1.0; }
int main()
{
  long long* d; (void)hipMalloc(&d, 64); long long h[5];
  double* big; size_t n = 64ull << 20; (void)hipMalloc(&big, n * 8);
  for (int trial = 0; trial < 6; trial++) {
    if (trial >= 2) hipLaunchKernelGGL(filler, dim3(4096), dim3(256), 0, 0, big, n);    // evict L2 / I-cache between launches
    hipLaunchKernelGGL(k, dim3(1), dim3(384), 0, 0, d, 1.0);
    (void)hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    printf("trial %d (%s): pass cycles %lld %lld %lld %lld  (2100 dependent fma, ~17 KB)\n", trial, trial >= 2 ? "after 512 MiB filler" : "back to back", h[0], h[1], h[2], h[3]);
  }
  return 0;
}
