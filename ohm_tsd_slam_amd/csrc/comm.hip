// comm.hip -- lib/libtsd_comm.so: the occupancy-map merge of the multi-grid case on RCCL (include/tsd_comm.h).
// Uses only the public C ABI of libtsd_hip.so (tsd_stream, tsd_occupancy_dev_async) plus RCCL.  The extraction kernels
// run on the context's stream (ordered behind the pushes already enqueued); ncclAllReduce(int8, max) runs on the
// communicator's own stream behind an event, so the scans that follow on the context's stream overlap the collective
// (xGMI ring: ~0.2 ms for 16 MiB on 8 GPUs); the next extraction waits for the previous collective through a second
// event.  No hipStreamSynchronize anywhere but in tsd_comm_occupancy_wait.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/tsd_comm.h"

struct tsd_comm {
  tsd_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
  size_t cells2 = 0;
  int8_t* d_map = nullptr;
  hipStream_t cstream = nullptr;       // the collective's stream
  hipEvent_t ev_extracted = nullptr;   // map written by the extraction kernels (context stream)
  hipEvent_t ev_reduced = nullptr;     // all-reduce finished with the map (collective stream)
  bool reduced_pending = false;
  // tsd_comm_profile: HIP events around the extraction (context stream) and the collective (communicator stream) of every merge
  bool profile = false;
  struct Timed { hipEvent_t t0, t1, t2; };   // before the extraction, after it, after the all-reduce
  std::vector<Timed> pending;
  std::vector<hipEvent_t> pool;
  double extract_ms = 0.0, allreduce_ms = 0.0;
  int merges_timed = 0;
  std::string err;
};

static hipEvent_t comm_event(tsd_comm* c)
{
  if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

static int fail(tsd_comm* c, const char* what, const char* detail)
{
  if (c) c->err = std::string(what) + ": " + (detail ? detail : "");
  return TSD_E_HIP;
}

extern "C" {

int tsd_comm_unique_id(char id_out[TSD_COMM_ID_BYTES])
{
  if (!id_out) return TSD_E_ARG;
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) { std::fprintf(stderr, "tsd_comm_unique_id: %s\n", ncclGetErrorString(r)); return TSD_E_HIP; }
  static_assert(sizeof(id.internal) == TSD_COMM_ID_BYTES, "NCCL_UNIQUE_ID_BYTES");
  std::memcpy(id_out, id.internal, TSD_COMM_ID_BYTES);
  return TSD_OK;
}

// what tsd_comm_create has built so far, given back on any of its failure paths (and by tsd_comm_destroy)
static void comm_release(tsd_comm* c)
{
  if (c->ev_extracted) hipEventDestroy(c->ev_extracted);
  if (c->ev_reduced) hipEventDestroy(c->ev_reduced);
  if (c->cstream) hipStreamDestroy(c->cstream);
  if (c->d_map) hipFree(c->d_map);
  delete c;
}

tsd_comm* tsd_comm_create(tsd_ctx* ctx, int world_size, int rank, const char id_in[TSD_COMM_ID_BYTES])
{
  if (!ctx || !id_in || world_size < 1 || rank < 0 || rank >= world_size) return nullptr;
  tsd_comm* c = new (std::nothrow) tsd_comm();
  if (!c) return nullptr;
  c->ctx = ctx; c->world = world_size; c->rank = rank; c->device = tsd_device(ctx);
  c->cells2 = (size_t)tsd_cells(ctx) * (size_t)tsd_cells(ctx);
  auto fail = [&](const char* what) -> tsd_comm* {
    std::fprintf(stderr, "tsd_comm_create: %s\n", what);
    comm_release(c);
    return nullptr;
  };
  if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&c->d_map, c->cells2) != hipSuccess) return fail("no device memory for the map");
  // (-1 = unknown; filled on the communicator's own stream, created first: a plain hipMemset would bring the NULL stream alive,
  // which takes one of the few hardware queues the scan's streams are mapped onto -- DESIGN 5)
  if (hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_extracted, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_reduced, hipEventDisableTiming) != hipSuccess)
    return fail("stream / events");
  if (hipMemsetAsync(c->d_map, 0xFF, c->cells2, c->cstream) != hipSuccess || hipStreamSynchronize(c->cstream) != hipSuccess)
    return fail("clearing the map");
  ncclUniqueId id;
  std::memcpy(id.internal, id_in, TSD_COMM_ID_BYTES);
  (void)hipGetLastError();      // RCCL reads the thread's sticky last error: an earlier hipErrorNotReady is not its business
  const ncclResult_t r = ncclCommInitRank(&c->comm, world_size, id, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    std::fprintf(stderr, "tsd_comm_create: ncclCommInitRank: %s\n", ncclGetErrorString(r));
    comm_release(c);               // (round 4 leaked the stream and the two events here)
    return nullptr;
  }
  return c;
}

void tsd_comm_destroy(tsd_comm* c)
{
  if (!c) return;
  hipSetDevice(c->device);
  if (c->ctx) tsd_sync(c->ctx);
  if (c->cstream) hipStreamSynchronize(c->cstream);
  if (c->comm) ncclCommDestroy(c->comm);
  for (auto& t : c->pending) { hipEventDestroy(t.t0); hipEventDestroy(t.t1); hipEventDestroy(t.t2); }
  for (hipEvent_t e : c->pool) hipEventDestroy(e);
  comm_release(c);
}

int tsd_comm_world_size(const tsd_comm* c) { return c ? c->world : 0; }
int tsd_comm_rank(const tsd_comm* c) { return c ? c->rank : -1; }
const char* tsd_comm_last_error(const tsd_comm* c) { return c ? c->err.c_str() : "null comm"; }
void* tsd_comm_map_dev(tsd_comm* c) { return c ? c->d_map : nullptr; }

static int allreduce_map_impl(tsd_comm* c, hipEvent_t t0)
{
  hipStream_t stream = static_cast<hipStream_t>(tsd_stream(c->ctx));
  hipEvent_t t1 = nullptr, t2 = nullptr;
  if (c->profile) {
    if (!t0) { t0 = comm_event(c); if (t0) hipEventRecord(t0, stream); }
    t1 = comm_event(c); t2 = comm_event(c);
    if (t1) hipEventRecord(t1, stream);                 // the map is written: the extraction's share ends here
  }
  // the map is complete once everything enqueued on the context's stream so far has run
  if (hipEventRecord(c->ev_extracted, stream) != hipSuccess || hipStreamWaitEvent(c->cstream, c->ev_extracted, 0) != hipSuccess)
    return fail(c, "event hand-over to the collective stream", "");
  (void)hipGetLastError();
  // -1 unknown < 0 free < 100 occupied: the signed maximum is "occupied wins over free wins over unknown"
  const ncclResult_t r = ncclAllReduce(c->d_map, c->d_map, c->cells2, ncclInt8, ncclMax, c->comm, c->cstream);
  if (r != ncclSuccess) return fail(c, "ncclAllReduce", ncclGetErrorString(r));
  if (hipEventRecord(c->ev_reduced, c->cstream) != hipSuccess) return fail(c, "hipEventRecord", "");
  if (t0 && t1 && t2) { hipEventRecord(t2, c->cstream); c->pending.push_back({t0, t1, t2}); }
  c->reduced_pending = true;
  return TSD_OK;
}

int tsd_comm_allreduce_map(tsd_comm* c)
{
  if (!c) return TSD_E_ARG;
  if (hipSetDevice(c->device) != hipSuccess) return fail(c, "hipSetDevice", "");
  return allreduce_map_impl(c, nullptr);
}

int tsd_comm_profile(tsd_comm* c, int on)
{
  if (!c) return TSD_E_ARG;
  c->profile = on != 0;
  return TSD_OK;
}

int tsd_comm_merge_times(tsd_comm* c, double* extract_ms_total, double* allreduce_ms_total, int* merges)
{
  if (!c) return TSD_E_ARG;
  if (hipSetDevice(c->device) != hipSuccess) return fail(c, "hipSetDevice", "");
  for (auto& t : c->pending) {
    float a = 0.f, b = 0.f;
    if (hipEventSynchronize(t.t2) == hipSuccess && hipEventElapsedTime(&a, t.t0, t.t1) == hipSuccess &&
        hipEventElapsedTime(&b, t.t1, t.t2) == hipSuccess) {
      c->extract_ms += (double)a; c->allreduce_ms += (double)b; c->merges_timed++;
    }
    c->pool.push_back(t.t0); c->pool.push_back(t.t1); c->pool.push_back(t.t2);
  }
  c->pending.clear();
  if (extract_ms_total) *extract_ms_total = c->extract_ms;
  if (allreduce_ms_total) *allreduce_ms_total = c->allreduce_ms;
  if (merges) *merges = c->merges_timed;
  return TSD_OK;
}

int tsd_comm_occupancy_allreduce(tsd_comm* c, int inflate, int inflate_factor)
{
  if (!c) return TSD_E_ARG;
  if (hipSetDevice(c->device) != hipSuccess) return fail(c, "hipSetDevice", "");
  if (c->reduced_pending) {       // the previous collective still owns the map: the extraction waits for it on the device
    if (hipStreamWaitEvent(static_cast<hipStream_t>(tsd_stream(c->ctx)), c->ev_reduced, 0) != hipSuccess) return fail(c, "hipStreamWaitEvent", "");
    c->reduced_pending = false;
  }
  hipEvent_t t0 = nullptr;
  if (c->profile) { t0 = comm_event(c); if (t0) hipEventRecord(t0, static_cast<hipStream_t>(tsd_stream(c->ctx))); }
  const int rc = tsd_occupancy_dev_async(c->ctx, c->d_map, inflate, inflate_factor);     // extraction kernels, stream order
  if (rc != TSD_OK) { c->err = tsd_last_error(c->ctx); if (t0) c->pool.push_back(t0); return rc; }
  return allreduce_map_impl(c, t0);
}

int tsd_comm_occupancy_wait(tsd_comm* c, int8_t* merged_host)
{
  if (!c) return TSD_E_ARG;
  if (hipSetDevice(c->device) != hipSuccess) return fail(c, "hipSetDevice", "");
  hipStream_t stream = c->cstream;
  if (merged_host) {
    const hipError_t e = hipMemcpyAsync(merged_host, c->d_map, c->cells2, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return fail(c, "hipMemcpyAsync", hipGetErrorString(e));
  }
  const hipError_t e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return fail(c, "hipStreamSynchronize", hipGetErrorString(e));
  return TSD_OK;
}

}  // extern "C"
