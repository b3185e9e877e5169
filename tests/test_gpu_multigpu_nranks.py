"""BASELINE configs 4 / 5 (2 and 8 robots, one grid per GPU, RCCL occupancy merge) the moment more than one GPU is visible.

The ranks are CHILD processes started before this process has touched the GPU (`torch.cuda.device_count()` does not
initialise it): `python -m torch.distributed.run --nproc-per-node N tests/nranks_check.py`, N = min(devices, 8).  Every rank
checks the native merge (include/tsd_comm.h: extraction kernels + ncclAllReduce(int8, max)) against the element-wise maximum
of all ranks' own `tsd_occupancy` maps, and that robots 0.7 m apart mark the same walls in the same cells
(/root/reference: SlamNode.cpp:101-122, launch/multi_slam.launch:40 are the model).  On a one-GPU box the N-rank test
skips; the one-rank run of the same body still runs there (RCCL world of one, local common-frame check)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices() -> int:
    import torch
    return torch.cuda.device_count()          # (counts devices without initialising the GPU)


def _run_ranks(n: int):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "nranks_check.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["nranks_check"] == "ok" and d["world"] == n and d["merge_checked"] is True
    assert sorted(d["devices"]) == list(range(n))
    return d


def test_native_merge_one_rank_and_local_common_frame():
    d = _run_ranks(1)
    assert d["round0"]["occupied_merged"] == d["round0"]["occupied_own"] > 200
    assert min(d["common_frame_local"]) >= 0.99


@pytest.mark.skipif(_devices() < 2, reason="needs two or more GPUs (BASELINE configs 4 / 5); the driver's multi-GPU node runs it")
def test_native_merge_n_ranks_equals_elementwise_max():
    n = min(_devices(), 8)
    d = _run_ranks(n)
    for rnd in ("round0", "round1"):
        assert d[rnd]["occupied_merged"] >= d[rnd]["occupied_own"] > 200
        assert min(d[rnd]["agreement_r0_r1"]) >= 0.99
