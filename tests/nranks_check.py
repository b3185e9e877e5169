"""Rank body of tests/test_gpu_multigpu_nranks.py, started as
    python -m torch.distributed.run --nproc-per-node N tests/nranks_check.py
by a parent that has not touched the GPU: BASELINE configs 4 / 5 in small -- one robot + one TSD grid per rank (GPU
`LOCAL_RANK`), robot r starting `multigpu.robot_offset_x(r)` from the grid centre of ONE common room
(launch/multi_slam.launch:40; the reference's robots share one TsdGrid in one process, SlamNode.cpp:101-122), a few
pushes through the C ABI, then the NATIVE merge of include/tsd_comm.h (`tsd_comm_occupancy_allreduce`: extraction kernels
+ ncclAllReduce(int8, max) over RCCL) -- checked on EVERY rank against the element-wise maximum of the ranks' own
`tsd_occupancy` maps, which travel over the gloo control plane.  The merge semantics are this repository's (the reference
has no merge), so the checker is the definition itself: merged == max over ranks, occupied > free > unknown.

Also checked: two robots that start 0.7 m apart mark the room's walls in the SAME cells (tests/nranks_common.py) -- between
ranks 0 and 1 when there are two, and, on every world size, between this rank's grid and a second context on the same GPU
that plays the next robot (so a one-GPU box exercises that property of the HIP path too).

Rank 0 prints one JSON line {"nranks_check": "ok", "world": N, ...}; any failed assertion ends the rank non-zero."""
import ctypes as C
import json
import os
import sys

# (read when HSA initialises: before anything of this process touches HIP -- see the note at the top of bench.py)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import torch
    import torch.distributed as dist
    assert torch.cuda.device_count() > local_rank, f"rank {rank}: no GPU {local_rank}"
    dist.init_process_group("gloo", rank=rank, world_size=world_size)      # control plane only (bench.py does the same)

    from ohm_tsd_slam_amd import capi, facade, multigpu
    from tests import helpers as H, nranks_common as NC
    gc, geo, world = NC.setup()
    host = facade.load_library()

    def ingest(ranges_f32):
        """the product's own Sensor::setRealMeasurementData + setStandardMask (csrc/host/obvision)"""
        data = np.zeros(geo.beams); mask = np.zeros(geo.beams, dtype=np.uint8)
        r = np.ascontiguousarray(ranges_f32, dtype=np.float32)
        host.tsd_host_sensor_ingest_f32(r.ctypes.data_as(C.POINTER(C.c_float)), geo.beams, geo.angle_increment, geo.angle_min, H.MAX_RANGE,
                                        data.ctypes.data_as(C.POINTER(C.c_double)), mask.ctypes.data_as(C.POINTER(C.c_uint8)), 0)
        return data, mask

    def push_scans(grid, robot, k0, k1):
        for k in range(k0, k1):
            pose, (x, y, yaw) = NC.robot_pose(world, robot, k)
            data, mask = ingest(world.scan(x, y, yaw, geo))
            grid.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL, want_stats=False)

    def all_maps(own):
        t = torch.from_numpy(np.ascontiguousarray(own.reshape(-1)))
        got = [torch.empty_like(t) for _ in range(world_size)]
        dist.all_gather(got, t)
        return [g.numpy().reshape(gc.cells, gc.cells) for g in got]

    grid = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc, device=local_rank)
    push_scans(grid, rank, 0, 4)
    ids = [multigpu.NativeOccupancyMerger.new_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    merger = multigpu.NativeOccupancyMerger(grid, world_size, rank, ids[0])
    assert merger.world_size() == world_size, f"rank {rank}: RCCL says {merger.world_size()} ranks, launcher {world_size}"

    summary = {}
    for rnd in range(2):
        merger.merge_async()
        if rnd == 1:
            merger.merge_async()              # back to back: the second extraction waits for the first collective on the device
        merged = merger.merged()
        own, n_occ = grid.occupancy(False, 2)
        maps = all_maps(own)
        want = np.maximum.reduce(maps)
        assert set(np.unique(want).tolist()) <= {-1, 0, 100}
        assert np.array_equal(merged, want), (f"rank {rank} round {rnd}: native RCCL merge != element-wise max of the ranks' maps "
                                              f"({int((merged != want).sum())} cells differ)")
        assert (own == 100).sum() > 200, "this rank's robot has not marked the room's walls"
        if world_size > 1:
            assert any((maps[0] != m).any() for m in maps[1:]), "all ranks hold the same map: the merge is not exercised"
            agree = NC.assert_common_frame(maps[0], maps[1], gc, 0, 1)
            # occupied wins over free wins over unknown
            assert np.array_equal(want == 100, np.logical_or.reduce([m == 100 for m in maps]))
            assert np.array_equal(want == -1, np.logical_and.reduce([m == -1 for m in maps]))
        else:
            agree = None
        summary[f"round{rnd}"] = {"occupied_own": int((own == 100).sum()), "occupied_merged": int((merged == 100).sum()),
                                  "known_merged": int((merged >= 0).sum()), "agreement_r0_r1": agree}
        push_scans(grid, rank, 4 + 2 * rnd, 6 + 2 * rnd)      # the maps change between the rounds

    # the next robot on a second context of THIS GPU: same walls, same cells (runs on a one-GPU box too)
    other = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc, device=local_rank)
    push_scans(other, rank + 1, 0, 8)
    own, _ = grid.occupancy(False, 2)
    theirs, _ = other.occupancy(False, 2)
    local = NC.assert_common_frame(own, theirs, gc, rank, rank + 1)
    other.close()
    merger.close()
    grid.close()

    oks = [None] * world_size
    dist.all_gather_object(oks, {"rank": rank, "pid": os.getpid(), "device": local_rank})
    if rank == 0:
        assert sorted(o["rank"] for o in oks) == list(range(world_size)) and len({o["pid"] for o in oks}) == world_size
        print(json.dumps({"nranks_check": "ok", "world": world_size, "rccl_world": world_size, "devices": [o["device"] for o in oks],
                          "merge_checked": True, "common_frame_local": local, **summary}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
