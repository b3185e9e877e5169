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

}  // namespace tsd
