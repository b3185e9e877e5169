// experiment: which SIMD do the waves of one workgroup land on? (HW_ID: SIMD_ID bits 5:4, CU_ID 11:8, SE 14:13 ... on gfx9)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main()
{
  unsigned* d; (void)hipMalloc(&d, 4096); unsigned h[64];
  for (int threads : {256, 384, 512, 320}) {
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(k, dim3(2), dim3(threads), 0, 0, d);
      (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
      printf("threads %d rep %d:", threads, rep);
      for (int b = 0; b < 2; b++) { printf("  block %d simd:", b); for (int w = 0; w < threads / 64; w++) printf(" %u", (h[b * 16 + w] >> 4) & 3); printf(" (cu %u)", (h[b * 16] >> 8) & 15); }
      printf("\n");
    }
  }
  return 0;
}
