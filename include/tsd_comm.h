/*
 * tsd_comm.h -- the ONE exchange step of the multi-grid case behind a C ABI: merging the per-GPU occupancy maps.
 *
 * one robot => one tsd_ctx => one GPU; localise and push never communicate.  Every
 * occ_grid_time_interval each GPU extracts its int8 occupancy map (-1 unknown / 0 free / 100 occupied;
 * RayCastAxisAligned2D::calcCoords + ThreadGrid marking, ThreadGrid.cpp:72-118) and the maps are merged with
 *     ncclAllReduce(map, map, cells * cells, ncclInt8, ncclMax, comm, stream)
 * over RCCL / xGMI: occupied wins over free wins over unknown, which is what the reference's own multi-robot mode --
 * N ThreadLocalize threads writing ONE shared TsdGrid in one process (SlamNode.cpp:77-86, :101-122) -- converges to.
 * The reference has no merge (and no collective anywhere), so the semantics are defined here and tested against a CPU
 * element-wise maximum.
 *
 * Library: lib/libtsd_comm.so (links librccl and libtsd_hip; kept out of libtsd_hip.so so that a single-GPU host does
 * not map RCCL).  The extraction kernels are enqueued on the context's own stream, behind the pushes already there;
 * the all-reduce runs on the communicator's stream behind an event, so the scans that follow overlap it; nothing
 * synchronises until tsd_comm_occupancy_wait.  One communicator per context; world_size ranks = world_size contexts (one process per GPU,
 * or several contexts of one process).
 */
#ifndef TSD_COMM_H
#define TSD_COMM_H

#include <stdint.h>
#include "tsd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TSD_COMM_ID_BYTES 128   /* NCCL_UNIQUE_ID_BYTES */

typedef struct tsd_comm tsd_comm;

/* ncclGetUniqueId: call on one rank, hand the bytes to the others (MPI, a ROS parameter, a file, torch.distributed ...) */
int tsd_comm_unique_id(char id_out[TSD_COMM_ID_BYTES]);
/* ncclCommInitRank on the context's device + the int8 map buffer (cells * cells bytes in HBM).  NULL on failure. */
tsd_comm* tsd_comm_create(tsd_ctx* ctx, int world_size, int rank, const char id[TSD_COMM_ID_BYTES]);
void tsd_comm_destroy(tsd_comm* comm);
int tsd_comm_world_size(const tsd_comm* comm);
int tsd_comm_rank(const tsd_comm* comm);
const char* tsd_comm_last_error(const tsd_comm* comm);

/* Occupancy extraction of this rank's grid (tsd_occupancy_dev_async) followed by the max all-reduce of the int8 maps,
 * both on the context's stream, no host synchronisation.  Returns as soon as the work is enqueued. */
int tsd_comm_occupancy_allreduce(tsd_comm* comm, int inflate, int inflate_factor);
/* Same collective on a map the caller wrote into tsd_comm_map_dev() (tests; hosts with their own extraction). */
int tsd_comm_allreduce_map(tsd_comm* comm);
/* Wait for the merge; merged_host (cells * cells bytes, row = y like nav_msgs/OccupancyGrid.data) may be NULL. */
int tsd_comm_occupancy_wait(tsd_comm* comm, int8_t* merged_host);
/* What a merge costs, measured with HIP events when switched on: the extraction kernels (context stream) and the collective
 * (from "map written" to the end of ncclAllReduce on the communicator's stream, i.e. including the wait for the slowest rank).
 * tsd_comm_merge_times waits for the merges issued so far and returns the running totals since tsd_comm_create. */
int tsd_comm_profile(tsd_comm* comm, int on);
int tsd_comm_merge_times(tsd_comm* comm, double* extract_ms_total, double* allreduce_ms_total, int* merges);
/* device address of the (merged) map */
void* tsd_comm_map_dev(tsd_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* TSD_COMM_H */
