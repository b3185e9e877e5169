"""The multi-GPU path of bench.py on ONE GPU: `torch.distributed.run` with one rank, process group on RCCL
("nccl"), occupancy extraction kernels into a torch device buffer, the max all-reduce issued for real
(`--force-dist`).  The 2/4/8-GPU runs are the driver's; this checks that nothing in the plumbing (imports,
device selection, stream hand-over, collective on an int8 device tensor, barriers) is broken before it gets there.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_dist_line(out):
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["scaling"] == "weak"
    assert d["value"] > 100.0 and d["tracking_error_m"] < 0.5
    # what the rank itself says: the RCCL communicator's size, its own rate, what a merge cost
    assert d["rccl_world"] == 1 and d["ranks_reporting"] == 1
    assert d["per_rank_scans_per_s"]["min"] > 100.0
    m = d["ms_occupancy_merge"]
    assert m["merges_timed_per_rank"] >= 1 and 0.0 < m["extract"] < 5.0 and 0.0 < m["allreduce"] < 50.0
    assert d["config"]["occupancy_merges_in_timed_region"] >= 1
    assert 1000.0 < d["roofline"]["peak_measured"] < 8000.0
    return d


def test_bench_distributed_plumbing_one_rank():
    """the driver's form: torch.distributed.run around bench.py"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "5", "--no-cpu-baseline",
           "--force-dist"]
    _check_dist_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))


def test_bench_self_launch_one_rank():
    """no launcher: bench.py starts its rank(s) itself -- the path `bench.py --gpus N` takes on an N-GPU node"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "5", "--no-cpu-baseline", "--force-dist"]
    _check_dist_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_native_rccl_merge_one_rank():
    """include/tsd_comm.h on one GPU, in a FRESH interpreter (tests/comm_check.py): a process that has both this image's
    ROCm runtime and the HIP / RCCL copies bundled with the torch wheel mapped -- as a pytest process that collected the
    gloo tests has -- mixes RCCL and HIP builds, which is no property of the library under test (a C++ host has no torch)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "tests.comm_check"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "comm_check ok" in out.stdout
