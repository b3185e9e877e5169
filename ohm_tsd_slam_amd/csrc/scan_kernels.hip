// scan_kernels.hip -- fused scan path (tsd_scan): the host steps of ThreadLocalize::eventLoop that sit BETWEEN
// the device kernels run on the device, so that one scan is one H2D copy and one chain of kernels with no
// host round trip in the middle.  The steps after the registration (isRegistrationError, Sensor::transform,
// isPoseChangeSignificant, arguments of the push and of the next ray cast) are the epilogue of k_icp
// (scan_device.hpp: scan_post_body); this file holds the argument derivation for a pose set from the host.
#include "scan_device.hpp"
#include <cstring>

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
// Bounded (~2 s).  A registration that never reported done must not let the push through with the PREVIOUS scan's arguments:
// the gate then switches the push off (the three push kernels read `enabled` from this very record) and raises the slot's error
// word in coherent host memory, which the next tsd_batch_* call on the slot turns into TSD_E_HIP.
__global__ void __launch_bounds__(64) k_wait_seq(const unsigned long long* seq, unsigned long long value, PushArgs* push,
                                                 unsigned int* err_host, unsigned int poll_bound)
{
  if (threadIdx.x == 0) {
    unsigned int polls = 0u;
    bool arrived;
    while (!(arrived = __hip_atomic_load(seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == value) && ++polls < poll_bound)
      __builtin_amdgcn_s_sleep(32);
    if (!arrived) {
      push->enabled = 0;
      __hip_atomic_store(err_host, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The other direction: the batched registrations are launched AHEAD of their ray casts (which run on the grid's stream, between
// the pushes) and wait on the device for this word, set by a one-wave kernel right behind the ray casts -- a kernel boundary, so
// the ray casts' outputs are visible device-wide when it runs.  No event on the grid's stream, no cross-queue hand-off.
__global__ void __launch_bounds__(64) k_set_flag(unsigned int* flag, unsigned int value)
{
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// probe: wait (bounded) for *flag == value; out = 1 seen, 2 gave up
__global__ void __launch_bounds__(64) k_probe_wait(const unsigned int* flag, unsigned int value, unsigned int poll_bound, unsigned int* out)
{
  if (threadIdx.x == 0) {
    unsigned int polls = 0u;
    bool seen;
    while (!(seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == value) && ++polls < poll_bound)
      __builtin_amdgcn_s_sleep(32);
    *out = seen ? 1u : 2u;
  }
}

// tests only (tsd_debug_stall_push_stream): one wave that holds its stream for `us` microseconds of the 100 MHz wall clock (bounded)
__global__ void __launch_bounds__(64) k_stall(unsigned int us)
{
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64(), ticks = 100ull * us;
    unsigned int polls = 0u;
    while (wall_clock64() - t0 < ticks && ++polls < 4000000u) __builtin_amdgcn_s_sleep(32);
  }
}

int launch_stall(tsd_ctx* ctx, hipStream_t stream, unsigned int us)
{
  hipLaunchKernelGGL(k_stall, dim3(1), dim3(64), 0, stream, us);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_set_flag(tsd_ctx* ctx, hipStream_t stream, unsigned int* flag, unsigned int value)
{
  hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(64), 0, stream, flag, value);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

// the same gate for ALL robots of a batch at once (their pushes run as one pass: push_multi.hip): lane i waits for robot i
struct WaitSeqMulti { const unsigned long long* seq[16]; unsigned long long value[16]; PushArgs* push[16]; int n; };
__global__ void __launch_bounds__(64) k_wait_seq_multi(WaitSeqMulti w, unsigned int* err_host, unsigned int poll_bound)
{
  const int i = threadIdx.x;
  if (i < w.n) {
    unsigned int polls = 0u;
    bool arrived;
    while (!(arrived = __hip_atomic_load(w.seq[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == w.value[i]) && ++polls < poll_bound)
      __builtin_amdgcn_s_sleep(32);
    if (!arrived) {
      w.push[i]->enabled = 0;
      __hip_atomic_store(err_host, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
int launch_wait_seq_multi(tsd_ctx* ctx, int n, const unsigned long long* const* seq, const unsigned long long* value, PushArgs* const* push,
                          unsigned int* err_host, unsigned int poll_bound)
{
  if (n < 1 || n > 16) return set_error(ctx, TSD_E_ARG, "launch_wait_seq_multi", hipSuccess);
  WaitSeqMulti w;
  std::memset(&w, 0, sizeof(w));
  w.n = n;
  for (int i = 0; i < n; i++) { w.seq[i] = seq[i]; w.value[i] = value[i]; w.push[i] = push[i]; }
  hipLaunchKernelGGL(k_wait_seq_multi, dim3(1), dim3(64), 0, ctx->stream, w, err_host, poll_bound);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

int launch_wait_seq(tsd_ctx* ctx, const unsigned long long* seq, unsigned long long value, PushArgs* push, unsigned int* err_host,
                    unsigned int poll_bound)
{
  hipLaunchKernelGGL(k_wait_seq, dim3(1), dim3(64), 0, ctx->stream, seq, value, push, err_host, poll_bound);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  return TSD_OK;
}

// d_words: [0] flag, [1] result of the waiter.  The waiter goes out FIRST, on `a`; the setter behind it in host order, on `b`.
// Side by side the waiter sees the flag after a few microseconds; on a shared in-order queue (or with dispatches serialised by a
// profiler / blocking launches) it can only leave through its bound (~5 ms here), and says so.
int probe_cross_stream_wait(tsd_ctx* ctx, hipStream_t a, hipStream_t b, unsigned int* d_words, bool* ok)
{
  *ok = false;
  TSD_HIP_CHECK(ctx, hipMemsetAsync(d_words, 0, 2 * sizeof(unsigned int), a));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(a));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(b));
  hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(64), 0, a, d_words, 1u, 5000u, d_words + 1);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(64), 0, b, d_words, 1u);
  TSD_HIP_CHECK(ctx, hipGetLastError());
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(a));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(b));
  unsigned int res = 0u;
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(&res, d_words + 1, sizeof(res), hipMemcpyDeviceToHost, a));       // (not the NULL stream: see tsd_create)
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(a));
  *ok = res == 1u;
  return TSD_OK;
}

}  // namespace tsd
