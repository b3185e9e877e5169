#!/usr/bin/env python3
"""bench.py -- scans/s of the per-scan hot path (ray cast -> ICP -> TSD push) on MI355X.

One "step" = one synthetic 1081-beam scan through ThreadLocalize's event-loop body and ThreadMapping's push
on a 4096x4096-cell TSD grid (BASELINE.json configs[1]): scan ingest on the host, the fused
tsd_scan (ray-cast kernel + persistent ICP kernel + gates + classify / update / halo push kernels) on the
GPU, pose read-back included.  The grid stays resident in HBM.  `value` is a SINGLE-STREAM LATENCY CHAIN:
one robot's scans are strictly sequential (the next ray cast needs this push), so it is 1 / (time of one scan).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg1|cfg2|cfg3] [--scene room|pillars|comb]
                  [--mode slam|push] [--storage f64|q32] [--robots R]

  The next scan is announced to the localiser (a replay knows it): its ingest, copy and tables run while the current scan is being
  registered (tsd_scan_stage); --no-lookahead turns that off.

  --mode push   (default for --scene comb, whose ranges do not depend on the pose, so there is nothing to localise
                against): a step = one TsdGrid::push from the ground-truth pose (tables + classify + update + halo).
                This is the HBM-bound leg SURVEY 8(d) prices against the roofline (cfg3 / comb = bandwidth stress).
  --storage q32 (push mode) the 8-byte-per-cell build, lib/libtsd_hip_q32.so.
  --robots R    R robots on ONE grid in one process (the reference's own multi-robot mode, SlamNode.cpp:101-122),
                scans replayed by one native publisher thread per robot (tsd_node_play; --python-feeders: Python
                threads); the facade's dispatcher batches the robots' scans (tsd_batch_*); value = all robots' scans / wall time.

N > 1: one rank per GPU, one robot + one grid per rank (BASELINE configs[3]/[4]); every 50 scans the ranks merge their int8
occupancy maps with an RCCL max all-reduce.  Weak scaling: `value` = scans of all ranks / max-over-ranks time.  The ranks come
from `python -m torch.distributed.run ... bench.py --gpus N` (the driver's form), or -- when --gpus N > 1 is given WITHOUT a
launcher -- bench.py starts that very command itself as a child process before anything touches the GPU and relays rank 0's
line.  It never prints `n_gpus: N` from fewer than N ranks: a launcher world that is not --gpus, a machine with fewer than N
GPUs, an RCCL communicator whose size is not N on every rank, or a child that fails all end in a non-zero exit code.

Prints ONE JSON line on rank 0.  Stage times (`stages_ms`, `ms_icp_iterate`, ...) and the roofline kernel's
duration come from HIP events on every n-th dispatch of each kernel INSIDE the timed region.
"""
from __future__ import annotations

import argparse
import gc as pygc
import json
import math
import os
import sys
import threading
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # CPU baseline: no active spinning
# RCCL between the ranks of one node shares device buffers across processes (hipIpcGetMemHandle).  This pool's host driver only
# supports dmabuf handles; with the legacy mode the runtime defaults to, the call fails with "invalid argument" and the
# communicator never comes up.  The variable is read when HSA initialises, i.e. at the first HIP call of the process: set HERE, before
# anything can have made one, so that ranks started by somebody else's `python -m torch.distributed.run ... bench.py` (the driver's
# form) get it exactly like the ranks of bench.py's own self-launch.  An explicit setting in the environment wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
SPREAD_STEPS = 200           # scans of the pass that times EVERY registration (ms_icp_iterate_spread: p50 / p99 / max)
MERGE_EVERY = 50             # occupancy merge period in scans (cfg 4/5)
CALIB_DOUBLES = 8 << 20      # k_calib_rmw: 2 arrays x 8 Mi doubles -> 128 MiB read + 128 MiB written per launch
PROFILE_TAG = "r6"              # profiles/<tag>_<workload>_*: the committed rocprofv3 summaries the line reads its traffic from
STREAM_DOUBLES = 48 << 20    # tsd_measure_stream: 2 arrays x 48 Mi doubles = 768 MiB footprint (3x the 256 MiB Infinity Cache)
CPU_REPEATS = 3              # cpu_baseline: passes per thread count (the median is reported)
STAGES = ("raycast", "icp", "push_classify", "push_update", "push_halo")


def workload_key(cfg: str, scene: str, mode: str, storage: str) -> str:
    return f"{cfg}_{scene}" + ("_push" if mode == "push" else "") + ("_q32" if storage == "q32" else "")


def pmc_traffic(kernel: str, key: str):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC summary committed for THIS workload under
    profiles/<PROFILE_TAG>_<workload>_pmc.json (tools/profile_bench.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes
    of this same command).  FETCH_SIZE / WRITE_SIZE are KB; they are scaled by the factors the calibration kernel of
    the same run gives for this access shape (known bytes / reported bytes; MI355X_MICROARCH.md HBM section: FETCH_SIZE
    tallies 128-byte requests at 64 bytes).  None when no summary is committed for the workload."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{key}_pmc.json")
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    k = d["kernels"].get("tsd::" + kernel)
    if not k or k.get("FETCH_SIZE_KB") is None or k.get("WRITE_SIZE_KB") is None:
        return None, None
    fr = fw = 1.0
    cal = d["kernels"].get("tsd::k_calib_rmw")
    if cal and cal.get("FETCH_SIZE_KB") and cal.get("WRITE_SIZE_KB"):
        known = 16.0 * CALIB_DOUBLES
        fr = known / (cal["FETCH_SIZE_KB"] * 1024.0)
        fw = known / (cal["WRITE_SIZE_KB"] * 1024.0)
    return (k["FETCH_SIZE_KB"] * fr + k["WRITE_SIZE_KB"] * fw) * 1024.0, os.path.basename(path)


def profile_algorithmic_bytes(key: str):
    """algorithmic bytes per launch of the bench line that ran UNDER rocprofv3 for this workload (the run the PMC counters belong to)"""
    bj = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{key}_bench_under_rocprof.json")
    try:
        lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
        return float(json.loads(lines[-1])["roofline"]["algorithmic_bytes_per_launch"])
    except Exception:
        return None


def profile_fraction(kernel: str, key: str):
    """roofline fraction recomputed from the files committed under profiles/ for this workload: the algorithmic bytes per
    launch of the bench line that ran under rocprofv3 (<tag>_<key>_bench_under_rocprof.json) over the average duration of
    `kernel` in rocprofv3's own --kernel-trace --stats table (<tag>_<key>_kernel_stats.csv).  None when not committed."""
    import csv
    bj = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{key}_bench_under_rocprof.json")
    ks = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{key}_kernel_stats.csv")
    if not (os.path.exists(bj) and os.path.exists(ks)):
        return None
    try:
        lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
        b = json.loads(lines[-1])["roofline"]["algorithmic_bytes_per_launch"]
        for r in csv.DictReader(open(ks)):
            if r["Name"].split("(")[0].strip().endswith("tsd::" + kernel):
                avg_ns = float(r["AverageNs"])
                gbs = b / avg_ns
                return {"frac": gbs / HBM_PEAK_GBS, "achieved": gbs, "avg_launch_us": avg_ns / 1e3, "calls": int(r["Calls"]),
                        "source": os.path.basename(ks)}
    except Exception:      # a malformed summary is not the bench's problem
        return None
    return None


def algorithmic_bytes(st: dict, pushes: int, beams: int, cell_bytes: int = 16) -> float:
    """Bytes one push MUST move (DESIGN.md "Roofline"; SURVEY 8(d)'s formula with `cell_bytes` = bytes of one cell's
    (tsd, weight): 16 for the reference's fp64 cells, 8 for the Q32 build): per updated cell tsd+weight read+write; an
    emptied initialised tile RMWs 1089 cells; a tile materialised from _initWeight > 0 writes 1024 cells; the scan
    (8 B range + 1 B mask per beam) is read once; every tile that passes the range cull reads/writes 16 B of tile
    state.  k_push_update (one workgroup per listed tile) moves all of these bytes but the tile state of the tiles
    k_push_classify rejects."""
    return (2.0 * cell_bytes * st["cells_updated"] + 2.0 * cell_bytes * 1089 * st["tiles_emptied_init"]
            + 1.0 * cell_bytes * 1024 * st["tiles_new_from_empty"] + 16.0 * st["tiles_range_pass"] + 9.0 * beams * pushes)


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = len(os.sched_getaffinity(0))
    physical = min(len(phys), logical) if phys else logical
    return model, max(physical, 1), logical


def cpu_baseline(cfg_name: str, scene: str, mode: str, n_multi: int):
    """The oracle (oracle/tsd_oracle.c: CPU restatement, OpenMP over tiles / beams like the reference, kd-tree NN like
    FLANN) timed on this box's host cores on the same synthetic workload: a thread sweep {1, 16, 64, physical cores},
    OMP_WAIT_POLICY=passive.  `value` = the best of the sweep, `cores` = its thread count; the 1-thread figure has its
    own key (SURVEY 8(d) "CPU baseline")."""
    from oracle import pyoracle as O
    from ohm_tsd_slam_amd import synth
    from tests.slam_driver import slam_kwargs
    from tests import helpers as H
    gc, geo, _ = synth.CONFIGS[cfg_name]
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, n_multi + 1)
    scans = synth.scans_for(world, geo, poses)
    model, physical, logical = cpu_info()
    sweep = sorted({t for t in (1, 16, 64, physical) if t <= logical})
    results = {}

    def one_pass(thr):
        n_scans = n_multi if thr > 1 else max(n_multi // 4, min(n_multi, 10))      # (one thread: 6x slower per scan and steadier)
        if mode == "push":
            og = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
            t_total = 0.0
            for k in range(n_scans + 1):
                pose = synth.pose_matrix(*poses[k])
                data, mask = O.ingest_f32(scans[k], 30.0, geo.angle_increment)
                t0 = time.perf_counter()
                og.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0, threads=thr)
                if k > 0:
                    t_total += time.perf_counter() - t0
            og.close()
            return dict(scans_per_s=n_scans / t_total, ms_push=1e3 * t_total / n_scans)
        slam = O.Slam(**slam_kwargs(gc, geo, nn_mode=1, threads=thr))
        slam.process_scan(scans[0])
        t_rc = t_icp = t_push = 0.0
        t0 = time.perf_counter()
        for k in range(1, n_scans + 1):
            r = slam.process_scan(scans[k])
            t_rc += r.t_raycast; t_icp += r.t_icp; t_push += r.t_push
        dt = time.perf_counter() - t0
        slam.close()
        return dict(scans_per_s=n_scans / dt, ms_raycast=1e3 * t_rc / n_scans, ms_icp=1e3 * t_icp / n_scans, ms_push=1e3 * t_push / n_scans)

    # every thread count: CPU_REPEATS passes over the same scans, the MEDIAN pass is the figure (a 20-scan sample of the 64-thread leg
    # is 80 ms of work and moved by +-10 % between runs of the same box: round 5's 235-285 scans/s)
    for thr in sweep:
        passes = sorted((one_pass(thr) for _ in range(CPU_REPEATS)), key=lambda r: r["scans_per_s"])
        results[thr] = dict(passes[len(passes) // 2], passes_scans_per_s=[p["scans_per_s"] for p in passes])
    best = max(results, key=lambda t: results[t]["scans_per_s"])
    out = {"value": results[best]["scans_per_s"], "unit": "scans/s", "cores": best, "kind": "port",
           "sample": f"median of {CPU_REPEATS} passes of {n_multi} {'pushes' if mode == 'push' else 'scans'} of {cfg_name}/{scene} after the init "
                     f"push per thread count ({max(n_multi // 4, min(n_multi, 10))} on one thread); oracle/tsd_oracle.c, OpenMP, kd-tree NN, OMP_WAIT_POLICY=passive",
           "cpu_model": model, "physical_cores": physical, "logical_cpus": logical,
           "value_1thread": results[1]["scans_per_s"] if 1 in results else None,
           "threads_sweep": {str(t): results[t] for t in sweep}}
    return out


# ------------------------------------------------------------------------------------------------ timing helpers
def stage_table(grid, steps):
    """Mean dispatch duration of every stage kernel from the events sampled in the timed region, and its share of a
    step (launches per step x mean duration)."""
    out = {}
    for name in STAGES:
        ms, n = grid.profile_get(name)
        out[name] = (ms / n) if n else None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3"])
    ap.add_argument("--scene", default=None)
    ap.add_argument("--mode", default=None, choices=["slam", "push"])
    ap.add_argument("--storage", default="f64", choices=["f64", "q32"])
    ap.add_argument("--robots", type=int, default=1)
    ap.add_argument("--pg-backend", choices=["gloo", "nccl"], default="gloo",
                    help="torch.distributed backend of the control plane (rendezvous, barrier, max over ranks); the occupancy merge is RCCL either way")
    ap.add_argument("--lookahead", action="store_true", help="replay mode as the line's own: announce scan k+1 to the localiser during registration k "
                    "(staged on the device ahead).  Default: no announcement -- what a sensor_msgs/LaserScan subscriber sees; the replay "
                    "rate is reported beside it as value_lookahead")
    ap.add_argument("--no-lookahead", action="store_true", help="(the default since round 5; accepted for old command lines)")
    ap.add_argument("--python-feeders", action="store_true", help="--robots: Python feeder threads instead of the native replay tsd_node_play")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scans", type=int, default=0, help="scans per pass of the CPU baseline's multi-thread legs (the 1-thread leg takes a quarter); "
                                                             "default 100 (cfg 1 / 2: ~10 s of CPU work over the sweep) or 12 (cfg 3)")
    ap.add_argument("--sample-every", type=int, default=0, help="time every n-th dispatch of each kernel (0 = auto)")
    ap.add_argument("--estimator", type=int, default=0, choices=[0, 1],
                    help="0: ClosedFormEstimator2D, what the node constructs (the bench line); 1: PointToLine2DEstimator")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the occupancy merge even with one rank (checks the "
                         "multi-GPU plumbing on a single GPU; tests/test_gpu_multigpu_plumbing.py)")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch the PMC calibration kernel (known byte count; used by tools/profile_bench.sh)")
    ap.add_argument("--occupancy", type=int, default=0, help="also run the occupancy extraction (row N1) this many times "
                                                           "after the timed region (profiling)")
    ap.add_argument("--occupancy-every-ms", type=float, default=0.0,
                    help="a map thread calls tsd_occupancy (extraction kernels + the map's copy to the host) every so many milliseconds "
                         "DURING the timed region, beside the localisers -- what ThreadGrid does every occ_grid_time_interval "
                         "(ThreadGrid.cpp:72-133); the line reports occupancy_calls_in_timed_region")
    ap.add_argument("--registration-mode", type=int, default=0, choices=[0, 3],
                    help="0: ICP only (the bench line, SURVEY 8(d)); 3: TSD_PDF pre-registration ahead of the ICP (config/single-laser.yaml:28), "
                         "fixed tsdpdf_seed; stages_ms.tsdpdf = the scoring kernels")
    ap.add_argument("--no-stream", action="store_true", help="skip the stream-bandwidth measurement (roofline.peak_measured): profile passes, whose "
                                                            "calibration counts k_calib_rmw launches of ONE known size")
    ap.add_argument("--no-second-pass", action="store_true", help="skip the comparison passes (value_repeat, value_lookahead, value_async_mapping, the spread pass)")
    ap.add_argument("--comparison-passes", type=int, default=3, help="passes behind value_repeat / value_lookahead / value_async_mapping (median reported)")
    ap.add_argument("--async-mapping", action="store_true",
                    help="`value` with the facade's async_mapping = 1 (the push beside the next registration, the next ray cast one "
                         "push behind: the reference's own ThreadMapping is asynchronous); default: strict order, and the "
                         "asynchronous rate as value_async_mapping")
    ap.add_argument("--launch-check", action="store_true",
                    help="multi-rank plumbing only (no GPU work): the ranks rendezvous over gloo, sum their ranks, rank 0 prints "
                         "{launch_check, world_size}.  With --gpus N and no launcher this goes through the self-launch path.")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if not launched and (args.gpus > 1 or args.force_dist or args.launch_check):
        # no launcher around us: start the ranks ourselves, as a CHILD process, before anything here has touched the GPU
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if launched and args.gpus != world_size:
        # one line of output must never claim more (or other) GPUs than ranks that ran
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_size} ranks; refusing to run "
                  f"(start it as: python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)", file=sys.stderr)
        sys.exit(2)
    n_gpus = world_size
    dist = None
    torch = None
    use_dist = launched
    if args.launch_check:
        sys.exit(launch_check(rank, world_size))
    if use_dist:
        # torch.distributed is the launcher's control plane only: rendezvous, the 128-byte RCCL id, barrier and max-over-ranks
        # of the contract -- on the CPU (gloo).  The data-path collective is RCCL behind the C ABI (include/tsd_comm.h).  A "nccl"
        # process group here would bring torch's own communicator and streams into the process, and HIP maps streams onto a few
        # hardware queues: measured, the scan's side stream then shares a queue with its main stream (12 % slower scans, and the
        # staging of the next scan lands behind the ray cast instead of beside the registration).  --pg-backend nccl restores it.
        import torch
        import torch.distributed as dist
        if torch.cuda.device_count() <= local_rank:
            print(f"bench.py: rank {rank} has no GPU (local rank {local_rank}, {torch.cuda.device_count()} visible)", file=sys.stderr)
            sys.exit(3)
        torch.cuda.set_device(local_rank)
        # (one node by contract: RCCL's bootstrap over the loopback interface too, the container's hostname may not resolve)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if args.pg_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))
        else:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo", rank=rank, world_size=world_size)

    from ohm_tsd_slam_amd import capi, facade, multigpu, synth
    gc, geo, default_scene = synth.CONFIGS[args.config]
    scene = args.scene or default_scene
    mode = args.mode or ("push" if scene == "comb" else "slam")
    if args.storage == "q32" and mode != "push":
        ap.error("--storage q32 is measured in --mode push (the C++ facade links the fp64 library)")
    if mode == "push" and world_size > 1:
        ap.error("--mode push is a one-GPU workload")
    K, W = args.steps, args.warmup
    # (several robots: many more dispatches per step, and every sampled one costs the chain of dependent launches ~10 us)
    every = args.sample_every or (32 if args.robots > 1 else 8)
    # the registration (the metric's second half, and its spread) keeps every second dispatch in a short run: ten samples in the driver's
    # 20 steps, at ~3 us of the chain per sampled dispatch
    every_icp = args.sample_every or (every if (K >= 80 or args.robots > 1) else (2 if K < 50 else 4))
    # the roofline kernel: every SECOND dispatch in a short run (the driver's 20 steps would otherwise leave five samples; every
    # dispatch was measured to cost the 20-step `value` 2.5 %: 4 800 against 4 920, 5 040 with a sixteenth of the dispatches sampled)
    every_upd = args.sample_every or (2 if (K < 50 and args.robots == 1) else every)
    device = local_rank if use_dist else 0
    cell_bytes = 8 if args.storage == "q32" else 16

    if mode == "push":
        out = run_push(args, gc, geo, scene, K, W, every, every_upd, device, capi, synth)
    else:
        out = run_slam(args, gc, geo, scene, K, W, every, every_upd, every_icp, device, rank, local_rank, world_size, use_dist, dist, torch,
                       facade, multigpu, synth)
    rc = 0
    if rank == 0:
        st, pushes, upd_ms, upd_launches = out.pop("_stats")
        stream_best, stream_mean = out.pop("_stream")
        ranks = out.pop("_ranks", None)
        bytes_per_launch = algorithmic_bytes(st, pushes, geo.beams, cell_bytes) / max(pushes, 1)
        upd_avg_ms = upd_ms / max(upd_launches, 1)
        achieved = bytes_per_launch / (upd_avg_ms * 1e-3) / 1e9 if upd_avg_ms > 0 else 0.0
        key = workload_key(args.config, scene, mode, args.storage)
        traffic, traffic_src = pmc_traffic("k_push_update", key)
        line = {
            "metric": "scans/sec + ms/ICP-iterate, 4096^2 TSD grid, 1081-beam scan" if args.config == "cfg2"
                      else f"scans/sec + ms/ICP-iterate, {gc.cells}^2 TSD grid, {geo.beams}-beam scan",
            "value": out.pop("value"), "unit": "scans/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
            "ms_per_step": out.pop("ms_per_step"), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
        }
        line.update(out)
        if ranks is not None:
            # what the ranks themselves report: the size of the RCCL communicator each one is in, its own rate, its merge cost
            worlds = sorted({r["rccl_world"] for r in ranks})
            rates = [r["scans_per_s"] for r in ranks]
            line["rccl_world"] = worlds[0] if len(worlds) == 1 else worlds
            line["ranks_reporting"] = len(ranks)
            line["per_rank_scans_per_s"] = {"min": min(rates), "max": max(rates)}
            nm = sum(r["merges_timed"] for r in ranks)
            line["ms_occupancy_merge"] = {
                "extract": max(r["ms_merge_extract"] for r in ranks), "allreduce": max(r["ms_merge_allreduce"] for r in ranks),
                "total": max(r["ms_merge_extract"] + r["ms_merge_allreduce"] for r in ranks), "merges_timed_per_rank": nm // max(len(ranks), 1),
                "of": "mean per merge on the slowest rank; HIP events: extraction kernels on the grid's stream, then map-written -> "
                      "end of ncclAllReduce(int8, max) on the communicator's stream (includes the wait for the other ranks)",
                "bytes_per_rank_ring": multigpu.merge_bytes_per_rank(gc.cells, world_size)}
            if line.get("merge_checked") is False:
                print("bench.py: the RCCL occupancy merge is NOT the element-wise maximum of the ranks' maps: not a valid measurement", file=sys.stderr)
                rc = 4
            if len(ranks) != world_size or worlds != [world_size]:
                print(f"bench.py: {len(ranks)} ranks reported RCCL world sizes {worlds}, expected {world_size} of {world_size}: not a valid "
                      f"{world_size}-GPU measurement", file=sys.stderr)
                rc = 4
        line["roofline"] = {"kernel": "k_push_update", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                            # (counter bytes over the algorithmic bytes of the SAME run -- the one under rocprofv3 -- not of this one,
                            # which may have a different number of steps)
                            "traffic_ratio": (traffic / profile_algorithmic_bytes(key)) if (traffic and profile_algorithmic_bytes(key)) else None,
                            "peak_measured": stream_best, "peak_measured_mean": stream_mean,
                            "frac_of_peak_measured": (achieved / stream_best) if stream_best else None,
                            "peak_measured_how": f"k_calib_rmw (read-modify-write stream, 8 B per lane like the push) over a "
                                                 f"{16 * STREAM_DOUBLES >> 20} MiB footprint, best / mean of 5 event-timed launches, this box, after the timed region",
                            "frac_from_profile": profile_fraction("k_push_update", key),
                            "algorithmic_bytes_per_launch": bytes_per_launch, "cell_bytes": cell_bytes,
                            "avg_launch_ms": upd_avg_ms, "launches": upd_launches,
                            "timing": f"HIP events on every {'' if every_upd == 1 else str(every_upd) + 'th '}dispatch of k_push_update, inside the timed region"}
        # SURVEY 8(d) prices B against the time of the push's kernels together; the object above is the dominant one.  Since round 3
        # k_push_update also writes the halo cells its changes belong to (propagateBorders' work, taken off k_push_halo), so the pair
        # of figures to watch is this one.
        stg = line.get("stages_ms") or {}
        if all(stg.get(k) for k in ("push_classify", "push_update")):
            # (the fused scan has no k_push_halo: the pass runs in the prologue of the ray cast that follows the push, inside stages_ms.raycast)
            halo = stg.get("push_halo")
            t_push = stg["push_classify"] + stg["push_update"] + (halo or 0.0)
            line["roofline"]["push_kernels"] = {
                "kernels": "k_push_classify + k_push_update" + (" + k_push_halo" if halo else " (halo pass: first waves of the following k_raycast)"),
                "sum_avg_launch_ms": t_push,
                "achieved": bytes_per_launch / (t_push * 1e-3) / 1e9, "frac": bytes_per_launch / (t_push * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "of": "the same algorithmic bytes over the summed average durations of the push's kernels (sampled dispatches)"}
        if not args.no_cpu_baseline and args.robots == 1 and not args.registration_mode:
            # (rank 0 only, whatever N: one robot's workload on this box's host cores; the other ranks wait at the closing barrier)
            # (its own sample size: the bound is CPU seconds -- ~10 s over the whole sweep at cfg 2 -- not the GPU leg's K)
            line["cpu_baseline"] = cpu_baseline(args.config, scene, mode, args.cpu_scans or (12 if args.config == "cfg3" else 100))
        if rc == 0:
            print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


def self_launch(args) -> int:
    """`bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py <same args>`
    as a child process (this process has not touched the GPU and never will), relay rank 0's JSON line, return the child's
    exit code.  Refuses (non-zero, no JSON line) when the machine has fewer than N GPUs."""
    import socket
    import subprocess
    n = max(args.gpus, 1)
    if not args.launch_check:
        try:
            import torch
            have = torch.cuda.device_count()            # (counts devices without initialising the GPU)
        except Exception as e:                           # noqa: BLE001
            print(f"bench.py: cannot count GPUs ({e})", file=sys.stderr)
            return 3
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} GPUs, this machine shows {have}: not starting (no line is printed for a run "
                  f"that did not happen)", file=sys.stderr)
            return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)          # (carries HSA_ENABLE_IPC_MODE_LEGACY: set at the top of this file; the ranks set it themselves too)
    env.pop("WORLD_SIZE", None)
    try:
        child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    except OSError as e:
        print(f"bench.py: could not start the ranks: {e}", file=sys.stderr)
        return 3
    lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
    for l in child.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if child.returncode != 0:
        print(f"bench.py: the {n}-rank child exited with {child.returncode}", file=sys.stderr)
        return child.returncode
    if len(lines) != 1:
        print(f"bench.py: expected ONE line from rank 0 of the {n}-rank child, got {len(lines)}", file=sys.stderr)
        return 5
    d = json.loads(lines[0])
    if not args.launch_check and d.get("n_gpus") != n:
        print(f"bench.py: the child reported n_gpus={d.get('n_gpus')}, asked for {n}", file=sys.stderr)
        return 5
    print(lines[0], flush=True)
    return 0


def launch_check(rank: int, world_size: int) -> int:
    """--launch-check: what the launcher path needs and nothing else -- every rank joins a gloo group on 127.0.0.1, the ranks
    are summed; rank 0 prints one line.  Runs without a GPU (the CPU test of the self-launch path)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t)
    got = [None] * world_size
    dist.all_gather_object(got, {"rank": rank, "pid": os.getpid(), "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")})
    ok = int(t.item()) == world_size * (world_size - 1) // 2 and sorted(g["rank"] for g in got) == list(range(world_size)) \
        and len({g["pid"] for g in got}) == world_size
    if rank == 0:
        print(json.dumps({"launch_check": bool(ok), "world_size": world_size, "ranks_seen": len(got), "processes": len({g["pid"] for g in got}),
                          "HSA_ENABLE_IPC_MODE_LEGACY": [g["ipc_legacy"] for g in sorted(got, key=lambda g: g["rank"])]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 6


def run_push(args, gc, geo, scene, K, W, every, every_upd, device, capi, synth):
    """Push-only steps: TsdGrid::push from the ground-truth pose, host scan in, nothing read back."""
    from ohm_tsd_slam_amd import facade
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, 1 + W + K)
    scans = synth.scans_for(world, geo, poses)
    grid = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc, device=device, storage=args.storage)
    lib = facade.load_library()
    import ctypes as C
    items = []
    for k in range(len(scans)):          # the product's own host ingest (Sensor::setRealMeasurementData + setStandardMask)
        data = np.zeros(geo.beams); mask = np.zeros(geo.beams, dtype=np.uint8)
        r = np.ascontiguousarray(scans[k], dtype=np.float32)
        lib.tsd_host_sensor_ingest_f32(r.ctypes.data_as(C.POINTER(C.c_float)), geo.beams, geo.angle_increment, geo.angle_min, 30.0,
                                       data.ctypes.data_as(C.POINTER(C.c_double)), mask.ctypes.data_as(C.POINTER(C.c_uint8)), 0)
        items.append((synth.pose_matrix(*poses[k]), data, mask))

    def step(k):
        pose, data, mask = items[k]
        grid.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0, want_stats=False)

    for k in range(0, 1 + W):
        step(k)
    grid.sync()
    grid.push_stats_total(reset=True)
    grid.profile(True, kernels=f"push_classify,push_update:{every_upd},push_halo/{every}")
    grid.profile_reset()
    t0 = time.perf_counter()
    for k in range(1 + W, 1 + W + K):
        step(k)
    grid.sync()
    elapsed = time.perf_counter() - t0
    upd_ms, upd_launches = grid.profile_get("push_update")
    stages = stage_table(grid, K)
    st, pushes = grid.push_stats_total()
    grid.profile(False)
    stream = grid.measure_stream(STREAM_DOUBLES, 5) if not args.no_stream else (None, None)
    if args.calibrate:
        grid.calibrate_rmw(CALIB_DOUBLES, 3)
    for _ in range(args.occupancy):
        grid.occupancy(False, 2)
    out = {
        "value": K / elapsed, "ms_per_step": 1e3 * elapsed / K,
        "config": {"workload": f"{args.config}: {gc.cells}x{gc.cells} cells @ {gc.cell_size} m, {geo.beams} beams, scene "
                               f"'{scene}', PUSH ONLY (TsdGrid::push from the ground-truth pose; no localisation), "
                               f"cell storage {args.storage}", "robots": 1, "mode": "push", "storage": args.storage},
        "stages_ms": {k: v for k, v in stages.items() if k.startswith("push")},
        "ms_push_kernels": sum(v for k, v in stages.items() if k.startswith("push") and v is not None),
        "pushes_in_timed_region": pushes, "cells_updated_per_push": st["cells_updated"] / max(pushes, 1),
        "cells_visited_per_push": st["cells_visited"] / max(pushes, 1),
        "tiles_updated_per_push": st["tiles_update"] / max(pushes, 1),
        "_stats": (st, pushes, upd_ms, upd_launches), "_stream": stream,
    }
    grid.close()
    return out


def run_slam(args, gc, geo, scene, K, W, every, every_upd, every_icp, device, rank, local_rank, world_size, use_dist, dist, torch, facade, multigpu, synth):
    R = args.robots
    # robot r starts 0.7 m further along -x (launch/multi_slam.launch:40).  Ranks own one grid each (--gpus N);
    # --robots R puts R robots on this rank's ONE grid (the reference's own multi-robot mode)
    # ONE world for all robots of a grid; robots beyond the first start on pillar-free lanes next to it
    world = synth.World(scene, gc, start_xy=[0.5 * gc.width + multigpu.robot_offset_x(rank), 0.5 * gc.width - 0.21])
    # (several robots: 3 m legs back and forth, so that enough pillar-free lanes exist among the 200 pillars)
    leg = 50 if R > 1 else None
    lanes = synth.free_lanes(world, R, 0.06 * leg, clearance=0.6) if R > 1 else [(float(world.start[0]), float(world.start[1]))]
    poses, scans = [], []
    for r in range(R):
        p = synth.trajectory(world, 1 + W + max(K, SPREAD_STEPS if R == 1 else K), leg=leg)
        p[:, 1] += lanes[r][1] - world.start[1]
        poses.append(p); scans.append(synth.scans_for(world, geo, p))
    scans32 = [np.ascontiguousarray(np.stack(sc), dtype=np.float32) for sc in scans]
    params = facade.node_params(gc, geo)
    if R == 1:
        params["tsd_slam/local_offset_x"] = multigpu.robot_offset_x(rank)
    else:
        params["robot_nbr"] = R
        for r in range(R):
            params[f"robot_{r}/name"] = f"robot{r}"
            params[f"tsd_slam/robot{r}/local_offset_x"] = lanes[r][0] - 0.5 * gc.width
            params[f"tsd_slam/robot{r}/local_offset_y"] = lanes[r][1] - 0.5 * gc.width
            params[f"tsd_slam/robot{r}/local_offset_yaw"] = 0.1
            # the ICP keys are per robot in multi-robot mode (ThreadLocalize.cpp:86-88: _robotName + "dist_filter_max" ...)
            params.update({f"robot{r}/dist_filter_max": 0.4, f"robot{r}/dist_filter_min": 0.02, f"robot{r}/icp_iterations": 30,
                           f"robot{r}/registration_mode": args.registration_mode})
            if args.registration_mode:
                params[f"robot{r}/tsdpdf_seed"] = 20261003 + 1000 * r
    if args.estimator:
        params["icp_estimator"] = args.estimator
    if args.registration_mode:
        # registration_mode 3 = TSD_PDF pre-registration ahead of the ICP (what config/single-laser.yaml:28 ships); a fixed seed
        # makes the reference's rand() draws reproducible
        params["registration_mode"] = args.registration_mode
        params["tsdpdf_seed"] = 20261003

    def one_pass(lookahead: bool, full: bool, async_mapping: bool = False, K=K, every_icp_dispatch: bool = False):
        """init + W warm-up scans + K timed scans on a fresh node; `full`: with the occupancy merge (N > 1), the stage table and
        everything else the line reports; otherwise just the rate (the comparison passes).  every_icp_dispatch: HIP events around EVERY
        registration of the region, returned as samples (the spread pass: its rate is not reported, the events cost stream time)."""
        node = facade.SlamNode(dict(params, async_mapping=1) if async_mapping else params, device=device, synchronous=True)
        grid = node.grid()
        merger = None
        if use_dist and full:
            # the merge itself is the C ABI of include/tsd_comm.h (extraction kernels + ncclAllReduce(int8, max) on the grid's
            # stream); torch.distributed only carries the communicator's unique id to the ranks
            ids = [multigpu.NativeOccupancyMerger.new_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            merger = multigpu.NativeOccupancyMerger(grid, world_size, rank, ids[0])
            merger.profile(True)

        merges = [0]

        def step(k, r=0):
            # (a replay knows the next scan: announced, the localiser stages it on the device during this registration)
            nxt = scans[r][k + 1] if (R == 1 and lookahead and k + 1 < len(scans[r])) else None
            node.laser(scans[r][k], geo.angle_min, geo.angle_increment, robot=r, ahead=nxt)
            # the occupancy merge: every MERGE_EVERY scans of the timed region, placed in the middle of each period (a region
            # shorter than one period still holds ONE merge, in its middle, so that the collective is always part of what is
            # timed); a few more at the end of the warm-up (RCCL sets its channels up on the first calls)
            if merger is not None and r == 0:
                s_t = k - (1 + W)
                period = min(MERGE_EVERY, max(K, 1))
                if (s_t >= 0 and s_t % period == period // 2) or (k <= W and k > W - 3):      # (warm-up: up to three, RCCL's first calls are slow)
                    merger.merge_async()        # extraction kernels + RCCL max all-reduce over xGMI, in stream order, no wait
                    merges[0] += 1 if s_t >= 0 else 0

        def run_range(k0, k1):
            if R == 1:
                for k in range(k0, k1):
                    step(k)
                return
            # one publisher thread per robot, like `rosbag play` feeding the reference node: laserCallBack runs the event-loop
            # body on the publisher's thread (synchronous facade), so the robots' scans overlap on the device like the reference's
            # N ThreadLocalize workers do.  Native threads (tsd_node_play) unless --python-feeders.
            if not args.python_feeders:
                node.play(scans32, k0, k1 - k0, geo.angle_min, geo.angle_increment)
                return
            ts = [threading.Thread(target=lambda rr=r: [step(k, rr) for k in range(k0, k1)]) for r in range(R)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()

        for r in range(R):
            node.laser(scans[r][0], geo.angle_min, geo.angle_increment, robot=r)          # init: freeFootprint + initPush
        run_range(1, 1 + W)
        grid.sync()
        merge_checked = None
        if merger is not None:
            merger.wait()
            # ONCE, before anything is timed: is the native merge (extraction kernels + ncclAllReduce(int8, max) over RCCL) the
            # element-wise maximum of the ranks' OWN occupancy maps?  Every rank extracts its own map (tsd_occupancy), the maps'
            # maximum is formed over the control plane (gloo, CPU), and every rank compares.  The reference has no merge
            # (SlamNode.cpp:101-122 shares one grid): the checker is the definition.
            merger.merge_async()
            merged = merger.merged().reshape(-1)
            own, _ = grid.occupancy(False, 2)
            t = torch.from_numpy(np.ascontiguousarray(own.reshape(-1)))
            if args.pg_backend == "nccl":
                t = t.to(f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            want = t.cpu().numpy()
            mine_ok = bool(np.array_equal(merged, want)) and bool((own == 100).any())
            ok = torch.tensor([1 if mine_ok else 0], dtype=torch.int32, device=t.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            merge_checked = bool(int(ok.item()) == 1)
            if not mine_ok:
                print(f"bench.py: rank {rank}: the RCCL occupancy merge differs from the element-wise maximum of the ranks' maps in "
                      f"{int((merged != want).sum())} cells", file=sys.stderr)
            warm_merge = merger.merge_times()    # (the warm-up merges -- RCCL's channel set-up -- are not what a merge costs)
        grid.push_stats_total(reset=True)
        if every_icp_dispatch:
            grid.profile(True, kernels="icp:1")
        else:
            grid.profile(True, kernels=f"push_update:{every_upd},icp:{every_icp},all/{every}")      # HIP events on every n-th dispatch of each kernel
        grid.profile_reset()
        if dist is not None and full:
            torch.cuda.synchronize()
            dist.barrier()
        b0 = node.batch_stats()
        # (the interpreter's cyclic collector off for the timed loop: with torch imported -- the multi-GPU launcher -- one
        # generation-2 pass is a 40 ms pause in the middle of the region, the same step in every run)
        pygc.collect()
        pygc.disable()
        occ_calls = [0]
        occ_stop = threading.Event()
        occ_thread = None
        if args.occupancy_every_ms > 0 and full:
            def map_thread():
                while not occ_stop.wait(args.occupancy_every_ms * 1e-3):
                    grid.occupancy(False, 2)
                    occ_calls[0] += 1
            grid.occupancy(False, 2)                   # (its buffers and its copy stream exist before anything is timed)
            occ_thread = threading.Thread(target=map_thread)
            occ_thread.start()
        t0 = time.perf_counter()
        run_range(1 + W, 1 + W + K)
        if occ_thread is not None:
            occ_stop.set(); occ_thread.join()
        if merger is not None:
            torch.cuda.synchronize()      # the whole device: the grid's streams, the communicator's stream and the merge on it
        else:
            grid.sync()
        # this rank's K steps are done (device idle, merge complete).  The closing barrier follows; the job's time is the MAX over the
        # ranks of these local times (all ranks left the opening barrier together), which the all-reduce below takes -- so the
        # barrier's own latency (0.2-0.4 ms over gloo, a tenth of a 20-step region) is not part of anybody's K steps.
        elapsed = time.perf_counter() - t0
        if merger is not None:
            grid.sync(); merger.wait()    # (nothing left to wait for: the library's own bookkeeping of the merge's events)
        if dist is not None and full:
            dist.barrier()
        pygc.enable()
        if not full:
            samples = grid.profile_samples("icp") if every_icp_dispatch else None
            grid.profile(False)
            node.close()
            return {"value": R * K / elapsed, "icp_samples_ms": samples}
        upd_ms, upd_launches = grid.profile_get("push_update")
        stages = stage_table(grid, K)
        icp_min, icp_max, icp_std = grid.profile_spread("icp")
        _, icp_n = grid.profile_get("icp")
        b1 = node.batch_stats()
        bstats = (b1[0] - b0[0], b1[1] - b0[1])
        st, pushes = grid.push_stats_total()
        grid.profile(False)
        errs = []
        for r in range(R):
            final = node.report(robot=r)
            errs.append(math.hypot(final["pose"][0, 2] - poses[r][W + K, 0], final["pose"][1, 2] - poses[r][W + K, 1]))    # (the last scan processed)

        ranks = None
        local_elapsed = elapsed
        if dist is not None:
            ex, ar, nm = merger.merge_times()
            ex -= warm_merge[0]; ar -= warm_merge[1]; nm -= warm_merge[2]
            mine = {"rank": rank, "rccl_world": merger.world_size(), "scans_per_s": R * K / local_elapsed, "merges_timed": nm,
                    "ms_merge_extract": ex / max(nm, 1), "ms_merge_allreduce": ar / max(nm, 1)}
            ranks = [None] * world_size
            dist.all_gather_object(ranks, mine)
            t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if args.pg_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        stream = grid.measure_stream(STREAM_DOUBLES, 5) if (rank == 0 and not args.no_stream) else (None, None)
        if args.calibrate:
            grid.calibrate_rmw(CALIB_DOUBLES, 3)
        pdf_ms, pdf_n = grid.profile_get("tsdpdf")
        occ_ms, occ_n = 0.0, 0
        if args.occupancy:
            # row N1 (occupancy extraction: k_occ_cells + k_occ_mark, events around both) on the map the timed region built
            occ_map, occ_changes = grid.occupancy(False, 2)
            occ_marked = int((occ_map == 100).sum())
            grid.profile(True, kernels="occupancy"); grid.profile_reset()
            for _ in range(args.occupancy):
                grid.occupancy(False, 2)
            occ_ms, occ_n = grid.profile_get("occupancy")
            grid.profile(False)

        pushes_per_step = pushes / max(K * R, 1)
        stage_sum = sum(v * (pushes_per_step if k.startswith("push") else 1.0) for k, v in stages.items() if v is not None)
        if pdf_n:
            stages = dict(stages, tsdpdf=pdf_ms / pdf_n)
            stage_sum += pdf_ms / pdf_n
        out = {
            "value": world_size * R * K / elapsed, "ms_per_step": 1e3 * elapsed / K,
            "config": {"workload": f"{args.config}: {gc.cells}x{gc.cells} cells @ {gc.cell_size} m, {geo.beams} beams, "
                                   f"scene '{scene}', icp_iterations 30, registration_mode {args.registration_mode}, "
                                   + (f"{R} robots on one grid per GPU" if R > 1 else "one robot + one grid per GPU")
                                   + (", point-to-line estimator" if args.estimator else ""),
                       "robots": world_size * R, "robots_per_grid": R, "mode": "slam", "storage": "f64",
                       "lookahead": bool(lookahead and R == 1),
                       "occupancy_merge_every": MERGE_EVERY if use_dist else None,
                       "occupancy_merges_in_timed_region": merges[0] if merger is not None else None,
                       "occupancy_merge_checked": "merged map == element-wise max of every rank's own tsd_occupancy map, on every rank, once in the warm-up"
                                                  if merge_checked else None,
                       "note": "single-stream latency chain per robot: a scan's ray cast needs the previous scan's push"
                               if R == 1 else "robots' scans batched by the facade's dispatcher (tsd_batch_*), two batch slots in turn",
                       "scans_per_batch": (bstats[1] / max(bstats[0], 1)) if R > 1 else None,
                       "stage_note": ("the batched registration kernel is launched ahead of its ray casts and waits for them on the device: "
                                      "its dispatch time (stages_ms.icp, ms_icp_iterate) includes that wait") if R > 1 else None},
            "ms_icp_iterate": stages["icp"], "ms_icp_per_iteration": (stages["icp"] / 30.0) if stages["icp"] else None,
            "ms_icp_iterate_sampled": {"min": icp_min, "max": icp_max, "std": icp_std, "samples": icp_n,
                                       "max_over_mean": (icp_max / stages["icp"]) if stages["icp"] else None,
                                       "of": "the dispatches sampled inside the timed region"},
            "ms_raycast": stages["raycast"],
            "ms_push_kernels": sum(v for k, v in stages.items() if k.startswith("push") and v is not None),
            "stages_ms": stages, "stages_sum_ms_per_scan": stage_sum,
            # what of a step is NOT inside a kernel of the chain: the host's ingest / launches / result poll where they are not hidden
            # behind a kernel, and the gaps between the kernels (one robot: the chain is strictly serial, so this is a plain difference)
            "ms_host_and_gaps_per_scan": (1e3 * elapsed / K - stage_sum) if R == 1 else None,
            "stage_timing": f"HIP events on every {every}th dispatch of each kernel (k_icp: every {every_icp}th; k_push_update: every "
                            f"{'one' if every_upd == 1 else ('second' if every_upd == 2 else str(every_upd) + 'th')}), inside the timed region",
            "pushes_in_timed_region": pushes, "cells_updated_per_push": st["cells_updated"] / max(pushes, 1),
            "cells_visited_per_push": st["cells_visited"] / max(pushes, 1),
            "tiles_updated_per_push": st["tiles_update"] / max(pushes, 1),
            "tracking_error_m": max(errs),
            "_stats": (st, pushes, upd_ms, upd_launches), "_stream": stream, "_ranks": ranks,
        }
        if args.occupancy_every_ms > 0:
            out["occupancy_calls_in_timed_region"] = occ_calls[0]
        if merger is not None:
            out["merge_checked"] = merge_checked
        if args.occupancy and occ_n:
            out["ms_occupancy_extract"] = occ_ms / occ_n
            out["occupancy_sign_changes"] = int(occ_changes)      # what RayCastAxisAligned2D::calcCoords returns: marks written
            out["occupancy_cells_marked"] = occ_marked
        if merger is not None:
            merger.close()        # (the communicator refers to the grid context: it goes first)
        node.close()
        return out

    out = one_pass(args.lookahead, True, args.async_mapping and R == 1)
    out["config"]["mapping"] = ("asynchronous: the push runs beside the next registration, the next ray cast is one push behind"
                                if (args.async_mapping and R == 1) else "strict: the next ray cast sees this scan's push")
    if R == 1 and not use_dist and not args.no_second_pass:
        # `value` is the rate a live scanner's subscriber gets: scan k+1 is handed over when scan k's result has been seen (it never has the
        # next LaserScan queued; the reference's localiser takes the newest scan, ThreadLocalize.cpp:319-332).  Beside it, each the
        # median of three passes on fresh nodes with its spread (one pass of 20 scans on a fresh node is not a number -- VERDICT r3):
        #   value_repeat          the line's own configuration again
        #   value_lookahead       a replay: scan k+1 announced during registration k and staged on the device ahead
        #   value_async_mapping   the mapper asynchronous like the reference's (ThreadMapping.cpp:51-76): a different, equally legitimate
        #                         order of the same work (tests/test_gpu_async_mapping.py) -- never reported as `value`
        def med3(*a):
            v = sorted(one_pass(*a)["value"] for _ in range(args.comparison_passes))
            return v[len(v) // 2], {"min": v[0], "max": v[-1], "passes": len(v)}
        out["value_repeat"], out["value_repeat_spread"] = med3(args.lookahead, False, args.async_mapping)
        if not args.lookahead:
            out["value_lookahead"], out["value_lookahead_spread"] = med3(True, False, args.async_mapping)
        else:
            out["value_no_lookahead"], out["value_no_lookahead_spread"] = med3(False, False, args.async_mapping)
        if not args.async_mapping:
            out["value_async_mapping"], out["value_async_mapping_spread"] = med3(args.lookahead, False, True)
        # the registration's tail: EVERY dispatch of k_icp over SPREAD_STEPS scans (HIP events around each one; a pass of its own,
        # because two event records per launch cost stream time)
        sp = one_pass(args.lookahead, False, args.async_mapping, K=max(K, SPREAD_STEPS), every_icp_dispatch=True)["icp_samples_ms"]
        if sp is not None and len(sp):
            sp = np.sort(np.asarray(sp, dtype=np.float64))
            out["ms_icp_iterate_spread"] = {
                "p50": float(sp[len(sp) // 2]), "p90": float(sp[int(0.9 * (len(sp) - 1))]), "p99": float(sp[int(0.99 * (len(sp) - 1))]),
                "max": float(sp[-1]), "min": float(sp[0]), "mean": float(sp.mean()), "samples": int(len(sp)),
                "max_over_mean": float(sp[-1] / sp.mean()),
                "of": f"every dispatch of k_icp over {max(K, SPREAD_STEPS)} scans of the same trajectory (a pass of its own)"}
    return out


if __name__ == "__main__":
    main()
