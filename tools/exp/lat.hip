// experiment: latencies that bound the single-workgroup ICP kernel (384 threads, one CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void k(long long* out, double seed, int mode)
{
  __shared__ double sd[1024];
  __shared__ int si[1024];
  __shared__ unsigned long long su[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += blockDim.x) { sd[i] = seed * i; si[i] = (i * 17 + 1) & 1023; su[i] = ~0ull; }
  __syncthreads();
  double x = seed + tid, y = seed * 2 + tid, z = seed * 3, w = seed * 4;
  int p = tid & 1023;
  unsigned long long acc = 0;
  const long long c0 = clock64();
  if (mode == 0) {
#pragma unroll 64
    for (int i = 0; i < N; i++) x = fma(x, 1.0000001, 1e-9);
  } else if (mode == 1) {
#pragma unroll 16
    for (int i = 0; i < N / 4; i++) { x = fma(x, 1.0000001, 1e-9); y = fma(y, 1.0000001, 1e-9); z = fma(z, 1.0000001, 1e-9); w = fma(w, 1.0000001, 1e-9); }
  } else if (mode == 2) {
#pragma unroll 64
    for (int i = 0; i < N; i++) p = si[p];
  } else if (mode == 3) {
#pragma unroll 16
    for (int i = 0; i < N; i++) { __syncthreads(); }
  } else if (mode == 4) {
#pragma unroll 16
    for (int i = 0; i < N; i++) { acc += atomicMin(&su[(p + i) & 1023], (unsigned long long)(i + tid)); }
  } else if (mode == 5) {
    float f = (float)seed + tid;
#pragma unroll 64
    for (int i = 0; i < N; i++) f = fmaf(f, 1.0000001f, 1e-9f);
    x = f;
  } else if (mode == 6) {   // LDS write then read of another thread's value with a barrier (one exchange)
#pragma unroll 8
    for (int i = 0; i < N; i++) { sd[tid] = x; __syncthreads(); x += sd[(tid + 64) % blockDim.x]; __syncthreads(); }
  } else if (mode == 7) {   // DPP wave reduction style: readlane chain
#pragma unroll 16
    for (int i = 0; i < N; i++) { int lo = __builtin_amdgcn_readlane(__double2loint(x), 15); x += (double)lo; }
  }
  const long long c1 = clock64();
  if (tid == 0) { out[0] = c1 - c0; out[1] = (long long)(x + y + z + w) + p + (long long)acc; }
}
int main()
{
  long long* d; (void)hipMalloc(&d, 64); long long h[2];
  const char* names[] = {"dependent fp64 fma", "4 independent fp64 fma chains (per fma)", "LDS pointer chase (ds_read_b32)", "__syncthreads", "LDS atomicMin u64 with return",
                         "dependent fp32 fma", "LDS exchange (write,bar,read,bar)", "readlane + cvt + add chain"};
  for (int threads : {64, 256, 384, 512})
    for (int mode = 0; mode < 8; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, 1.0, mode);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      }
      printf("threads %3d  %-42s %7.1f cycles each\n", threads, names[mode], (double)h[0] / N);
    }
  return 0;
}
