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
