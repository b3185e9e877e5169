// experiment: shader clock seen by one busy workgroup vs a busy chip (clock64 = s_memtime, wall_clock64 = 100 MHz)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long* out, int iters, double seed)
{
  const long long c0 = clock64(), w0 = wall_clock64();
  double x = seed + threadIdx.x;
  for (int i = 0; i < iters; i++) x = fma(x, 1.0000001, 1e-9);
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main()
{
  long long* d; hipMalloc(&d, 64); long long h[3];
  const int blocks[] = {1, 1, 256, 4096, 1, 1};
  for (int rep = 0; rep < 6; rep++) {
    for (int iters : {20000, 200000, 2000000}) {
      hipLaunchKernelGGL(spin, dim3(blocks[rep]), dim3(256), 0, 0, d, iters, 1.0);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("blocks %5d iters %8d: clock64 %10lld wall(100MHz) %8lld -> clock64 rate %.0f MHz, %.2f clock64 ticks per fma\n", blocks[rep], iters, h[0], h[1],
             h[0] / (h[1] / 100.0), (double)h[0] / iters);
    }
  }
  return 0;
}
