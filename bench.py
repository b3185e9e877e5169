#!/usr/bin/env python3
"""bench.py -- scans/s of the per-scan hot path (ray cast -> ICP -> TSD push) on MI355X.

One "step" = one synthetic 1081-beam scan through ThreadLocalize's event-loop body and
ThreadMapping's push on a 4096x4096-cell TSD grid (BASELINE.json configs[1]): scan
ingest on the host, tsd_localize (ray-cast kernel + persistent ICP kernel) and tsd_push (classify /
update / halo kernels) on the GPU, pose read-back included.  The grid stays resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg1] [--scene ...]

N > 1 (launched by torch.distributed.run, one rank per GPU): one robot + one grid per GPU (the
multi-robot case, BASELINE configs[3]/[4]); every 50 scans the ranks merge their int8 occupancy maps
with an RCCL max all-reduce.  Weak scaling: `value` = scans of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # CPU baseline: no active spinning
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
MERGE_EVERY = 50             # occupancy merge period in scans (cfg 4/5)
CALIB_DOUBLES = 8 << 20      # k_calib_rmw: 2 arrays x 8 Mi doubles -> 128 MiB read + 128 MiB written per launch


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/*_pmc.json, newest
    tag; collected with tools/profile_bench.sh on this same command in separate --pmc passes).  FETCH_SIZE /
    WRITE_SIZE are KB; they are scaled by the factors the calibration kernel of the same run gives for this
    8 B/lane access shape (known bytes / reported bytes).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=os.path.getmtime)
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    k = d["kernels"].get("tsd::" + kernel)
    if not k or k.get("FETCH_SIZE_KB") is None or k.get("WRITE_SIZE_KB") is None:
        return None, None
    fr = fw = 1.0
    cal = d["kernels"].get("tsd::k_calib_rmw")
    if cal and cal.get("FETCH_SIZE_KB") and cal.get("WRITE_SIZE_KB"):
        known = 16.0 * CALIB_DOUBLES
        fr = known / (cal["FETCH_SIZE_KB"] * 1024.0)
        fw = known / (cal["WRITE_SIZE_KB"] * 1024.0)
    return (k["FETCH_SIZE_KB"] * fr + k["WRITE_SIZE_KB"] * fw) * 1024.0, os.path.basename(files[-1])


def algorithmic_bytes(st: dict, pushes: int, beams: int) -> float:
    """Bytes one push MUST move, fp64 SoA storage (DESIGN.md "Roofline"): per updated cell tsd+weight
    read+write = 32 B; an emptied initialised tile RMWs 1089 cells; a tile materialised from
    _initWeight > 0 writes 1024 cells; the scan (8 B range + 1 B mask per beam) is read once; every tile
    that passes the range cull reads/writes 16 B of tile state.  k_push_update (one workgroup per listed tile)
    moves all of these bytes but the tile state of the tiles k_push_classify rejects."""
    return (32.0 * st["cells_updated"] + 32.0 * 1089 * st["tiles_emptied_init"]
            + 16.0 * 1024 * st["tiles_new_from_empty"] + 16.0 * st["tiles_range_pass"] + 9.0 * beams * pushes)


def cpu_baseline(cfg_name: str, scene: str, n_scans: int):
    """The oracle (CPU restatement, OpenMP over tiles / beams like the reference, kd-tree NN like FLANN)
    timed on this box's host cores on the same synthetic workload."""
    from oracle import pyoracle as O
    from ohm_tsd_slam_amd import synth
    from tests.slam_driver import slam_kwargs
    gc, geo, _ = synth.CONFIGS[cfg_name]
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, n_scans + 1)
    scans = synth.scans_for(world, geo, poses)
    cores = len(os.sched_getaffinity(0))
    slam = O.Slam(**slam_kwargs(gc, geo, nn_mode=1, threads=cores))
    slam.process_scan(scans[0])
    t_rc = t_icp = t_push = 0.0
    t0 = time.perf_counter()
    for k in range(1, n_scans + 1):
        r = slam.process_scan(scans[k])
        t_rc += r.t_raycast; t_icp += r.t_icp; t_push += r.t_push
    dt = time.perf_counter() - t0
    return {
        "value": n_scans / dt, "unit": "scans/s", "cores": cores, "kind": "port",
        "sample": f"{n_scans} scans of {cfg_name}/{scene} after the init push; oracle/tsd_oracle.c, "
                  f"OpenMP {cores} threads, kd-tree NN, OMP_WAIT_POLICY=passive",
        "ms_raycast": 1e3 * t_rc / n_scans, "ms_icp": 1e3 * t_icp / n_scans, "ms_push": 1e3 * t_push / n_scans,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3"])
    ap.add_argument("--scene", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scans", type=int, default=30)
    ap.add_argument("--estimator", type=int, default=0, choices=[0, 1],
                    help="0: ClosedFormEstimator2D, what the node constructs (the bench line); 1: PointToLine2DEstimator")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the occupancy merge even with one rank (checks the "
                         "multi-GPU plumbing on a single GPU; tests/test_gpu_multigpu_plumbing.py)")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch the PMC calibration kernel (known byte count; used by tools/profile_bench.sh)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    dist = None
    torch = None
    use_dist = world_size > 1 or args.force_dist
    if use_dist:
        import torch  # plumbing only: device tensors for the collective + torch.distributed (RCCL)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))

    from ohm_tsd_slam_amd import facade, multigpu, synth
    gc, geo, default_scene = synth.CONFIGS[args.config]
    scene = args.scene or default_scene

    # robot r starts 0.7 m further along -x (launch/multi_slam.launch:40); every rank owns one grid
    off_x = multigpu.robot_offset_x(rank)
    world = synth.World(scene, gc, start_xy=[0.5 * gc.width + off_x, 0.5 * gc.width - 0.21])
    K, W = args.steps, args.warmup
    poses = synth.trajectory(world, 1 + W + K)
    scans = synth.scans_for(world, geo, poses)

    params = facade.node_params(gc, geo)
    params["tsd_slam/local_offset_x"] = off_x
    if args.estimator:
        params["icp_estimator"] = args.estimator
    node = facade.SlamNode(params, device=local_rank if use_dist else 0, synchronous=True)
    grid = node.grid()
    merger = None
    if use_dist:
        merger = multigpu.OccupancyMerger(gc.cells, device=f"cuda:{local_rank}")

    def step(k):
        node.laser(scans[k], geo.angle_min, geo.angle_increment)
        if merger is not None and k % MERGE_EVERY == 0:
            merger.fill_from_grid(grid)     # occupancy extraction kernels on the ctx stream
            merger.merge_async(force=args.force_dist)   # RCCL max all-reduce over xGMI, overlaps the next scans

    node.laser(scans[0], geo.angle_min, geo.angle_increment)          # init: freeFootprint + initPush
    for k in range(1, 1 + W):
        step(k)
    grid.sync()
    grid.push_stats_total(reset=True)
    grid.profile(True, kernels="push_update/4")      # HIP events on the ctx stream around every 4th launch
    grid.profile_reset()
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(1 + W, 1 + W + K):
        step(k)
    grid.sync()
    if dist is not None:
        merger.wait()
        torch.cuda.synchronize()
        dist.barrier()
    elapsed = time.perf_counter() - t0
    upd_ms, upd_launches = grid.profile_get("push_update")
    st, pushes = grid.push_stats_total()
    final = node.report()
    track_err = math.hypot(final["pose"][0, 2] - poses[-1, 0], final["pose"][1, 2] - poses[-1, 1])

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-stage breakdown: a short extra pass with every kernel timed (not part of `value`)
    grid.profile(True, kernels="all")
    grid.profile_reset()
    extra = min(20, K)
    more = synth.scans_for(world, geo, synth.trajectory(world, 1 + W + K + extra)[-extra:])
    for s in more:
        node.laser(s, geo.angle_min, geo.angle_increment)
    grid.sync()
    stages = {}
    for name in ("raycast", "icp", "push_classify", "push_update", "push_halo"):
        ms, n = grid.profile_get(name)
        stages[name] = ms / n if n else None
    grid.profile(False)
    if args.calibrate:
        grid.calibrate_rmw(CALIB_DOUBLES, 3)

    if rank == 0:
        bytes_total = algorithmic_bytes(st, pushes, geo.beams)
        bytes_per_launch = bytes_total / max(pushes, 1)
        upd_avg_ms = upd_ms / max(upd_launches, 1)
        achieved = bytes_per_launch / (upd_avg_ms * 1e-3) / 1e9 if upd_avg_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic("k_push_update") if args.config == "cfg2" and scene == default_scene else (None, None)
        out = {
            "metric": "scans/sec + ms/ICP-iterate, 4096^2 TSD grid, 1081-beam scan" if args.config == "cfg2"
                      else f"scans/sec + ms/ICP-iterate, {gc.cells}^2 TSD grid, {geo.beams}-beam scan",
            "value": world_size * K / elapsed, "unit": "scans/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {gc.cells}x{gc.cells} cells @ {gc.cell_size} m, {geo.beams} beams, "
                                   f"scene '{scene}', icp_iterations 30, one robot + one grid per GPU"
                                   + (", point-to-line estimator" if args.estimator else ""),
                       "robots": world_size, "occupancy_merge_every": MERGE_EVERY if world_size > 1 else None},
            "ms_icp_iterate": stages["icp"], "ms_icp_per_iteration": (stages["icp"] / 30.0) if stages["icp"] else None,
            "ms_raycast": stages["raycast"],
            "ms_push_kernels": sum(v for k, v in stages.items() if k.startswith("push") and v is not None),
            "pushes_in_timed_region": pushes, "cells_updated_per_push": st["cells_updated"] / max(pushes, 1),
            "tiles_updated_per_push": st["tiles_update"] / max(pushes, 1),
            "tracking_error_m": track_err,
            "roofline": {"kernel": "k_push_update", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": upd_avg_ms,
                         "launches": upd_launches},
        }
        if not args.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, scene, min(args.cpu_scans, K))
        print(json.dumps(out), flush=True)
    node.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
