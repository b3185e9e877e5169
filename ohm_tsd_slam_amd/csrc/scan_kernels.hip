// scan_kernels.hip -- fused scan path (tsd_scan): the host steps of ThreadLocalize::eventLoop that sit BETWEEN
// the device kernels run on the device, so that one scan is one H2D copy and one chain of kernels with no
// host round trip in the middle.  The steps after the registration (isRegistrationError, Sensor::transform,
// isPoseChangeSignificant, arguments of the push and of the next ray cast) are the epilogue of k_icp
// (scan_device.hpp: scan_post_body); this file holds the argument derivation for a pose set from the host.
#include "scan_device.hpp"

namespace tsd {

__global__ void k_scan_prepare(GridDev g, SensorDev* st)
{
  if (threadIdx.x == 0 && blockIdx.x == 0) d_derive_args(st, g.min_x, g.max_x, g.min_y, g.max_y);
}

int launch_scan_prepare(tsd_ctx* ctx, SensorDev* st)
{
  hipLaunchKernelGGL(k_scan_prepare, dim3(1), dim3(64), 0, ctx->stream, ctx->grid, st);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

// Gate ahead of a batched robot's push: ONE wave polls the sensor's sequence number (written by its registration's epilogue,
// which runs on the batch's own stream) and ends when it has arrived; the push kernels behind it on the stream then start with
// fresh caches.  Replaces a cross-queue event wait for the whole batch's kernel by a wait for this robot's own workgroup.
// Bounded (~2 s): a registration that never ran ends in a push of stale arguments gated off by `enabled` at worst, not a hang.
__global__ void __launch_bounds__(64) k_wait_seq(const unsigned long long* seq, unsigned long long value)
{
  if (threadIdx.x == 0) {
    unsigned int polls = 0u;
    while (__hip_atomic_load(seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != value && ++polls < (1u << 21))
      __builtin_amdgcn_s_sleep(32);
  }
}

// The other direction: the batched registrations are launched AHEAD of their ray casts (which run on the grid's stream, between
// the pushes) and wait on the device for this word, set by a one-wave kernel right behind the ray casts -- a kernel boundary, so
// the ray casts' outputs are visible device-wide when it runs.  No event on the grid's stream, no cross-queue hand-off.
__global__ void __launch_bounds__(64) k_set_flag(unsigned int* flag, unsigned int value)
{
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

int launch_set_flag(tsd_ctx* ctx, unsigned int* flag, unsigned int value)
{
  hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(64), 0, ctx->stream, flag, value);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_wait_seq(tsd_ctx* ctx, const unsigned long long* seq, unsigned long long value)
{
  hipLaunchKernelGGL(k_wait_seq, dim3(1), dim3(64), 0, ctx->stream, seq, value);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

}  // namespace tsd
