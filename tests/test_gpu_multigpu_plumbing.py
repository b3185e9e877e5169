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


def test_bench_distributed_plumbing_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "5", "--no-cpu-baseline",
           "--force-dist"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["scaling"] == "weak"
    assert d["value"] > 100.0 and d["tracking_error_m"] < 0.5


def test_native_rccl_merge_one_rank():
    """include/tsd_comm.h on one GPU, in a FRESH interpreter (tests/comm_check.py): a process that has both this image's
    ROCm runtime and the HIP / RCCL copies bundled with the torch wheel mapped -- as a pytest process that collected the
    gloo tests has -- mixes RCCL and HIP builds, which is no property of the library under test (a C++ host has no torch)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "tests.comm_check"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "comm_check ok" in out.stdout
