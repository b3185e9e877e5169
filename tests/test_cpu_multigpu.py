"""The N > 1 path on CPU: two ranks (gloo, 127.0.0.1), one robot + one grid per rank.

* The merge SEMANTICS (element-wise max: occupied > free > unknown) with two ranks, through ``tests/gloo_merger.py`` --
  a test-only twin over ``torch.distributed`` tensors.  It is NOT what ``bench.py --gpus N`` runs: that is the native RCCL
  merge behind the C ABI (``include/tsd_comm.h`` -> ``multigpu.NativeOccupancyMerger``), which needs GPUs (RCCL refuses two
  ranks on one device; the one-rank run is tests/test_gpu_multigpu_plumbing.py, the N-rank run with its own check against
  the element-wise maximum is tests/test_gpu_multigpu_nranks.py, skipped below two GPUs).  The per-rank maps come from the
  oracle here.
* The COMMON MAP FRAME: robots that start `local_offset_x` apart in one room mark the room's walls in the same cells of their
  own maps (tests/nranks_common.py holds the check the GPU test applies to the HIP path's maps).
* The LAUNCH path of ``bench.py``: ``--gpus 2`` without a launcher starts two ranks itself (``--launch-check``: rendezvous only,
  no GPU), refuses to print a line when the machine does not have the GPUs, and refuses a launcher world that is not ``--gpus``."""
import json
import subprocess
import sys
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from ohm_tsd_slam_amd import multigpu, synth  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import gloo_merger, nranks_common  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_map(rank, n_scans=3):
    """Occupancy map of robot `rank` after a few pushes (oracle = checker-side generator of test data)."""
    from oracle import pyoracle as O
    gc, geo, world = nranks_common.setup()
    grid = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    for k in range(n_scans):
        pose, (x, y, yaw) = nranks_common.robot_pose(world, rank, k)
        data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), H.MAX_RANGE, geo.angle_increment)
        grid.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
    content = np.full(gc.cells * gc.cells, -1, dtype=np.int8)
    occ, _ = grid.occupancy(content)
    return gc, occ.reshape(-1)


def _worker(rank, world_size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        gc, mine = _rank_map(rank)
        merger = gloo_merger.OccupancyMerger(gc.cells)
        assert merger.active
        merger.fill_from_host(mine)
        merger.merge_async()                       # would overlap the next scans on a GPU
        merged = merger.merged().numpy().reshape(-1).copy()
        # second round: maps only grow, merging again is idempotent
        merger.fill_from_host(merged)
        merger.merge_async()
        again = merger.merged().numpy().reshape(-1).copy()
        q.put((rank, mine, merged, bool(np.array_equal(again, merged))))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_occupancy_merge_two_ranks_gloo():
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(ws)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    maps = [r[1] for r in res]
    want = np.maximum(maps[0], maps[1])
    assert set(np.unique(want)).issubset({-1, 0, 100})
    assert (maps[0] != maps[1]).any(), "the two robots must see different maps for the test to mean anything"
    for r in res:
        assert np.array_equal(r[2], want), f"rank {r[0]}: merged map != element-wise max"
        assert r[3], "merge is not idempotent"
    # occupied wins over free wins over unknown
    both = (maps[0] == 100) | (maps[1] == 100)
    assert np.array_equal(want == 100, both)
    assert np.array_equal(want == -1, (maps[0] == -1) & (maps[1] == -1))
    # the two robots start 0.7 m apart in ONE room: the walls land in the same cells of both maps (common map frame)
    gc = nranks_common.setup()[0]
    nranks_common.assert_common_frame(maps[0].reshape(gc.cells, gc.cells), maps[1].reshape(gc.cells, gc.cells), gc, 0, 1)


def test_common_frame_check_detects_a_wrong_frame():
    """the check itself: a map expressed in the robot's OWN start frame (shifted by its local_offset_x) must fail it"""
    gc = nranks_common.setup()[0]
    a = _rank_map(0)[1].reshape(gc.cells, gc.cells)
    b = _rank_map(1)[1].reshape(gc.cells, gc.cells)
    nranks_common.assert_common_frame(a, b, gc, 0, 1)
    shift = int(round((multigpu.robot_offset_x(0) - multigpu.robot_offset_x(1)) / gc.cell_size))
    with pytest.raises(AssertionError):
        nranks_common.assert_common_frame(a, np.roll(b, shift, axis=1), gc, 0, 1)


def test_single_rank_is_a_no_op():
    merger = gloo_merger.OccupancyMerger(8)
    assert not merger.active
    merger.fill_from_host(np.arange(64, dtype=np.int8) % 3 - 1)
    assert merger.merge_async() is None
    assert merger.merged().shape == (8, 8)
    assert multigpu.merge_bytes_per_rank(4096, 1) == 0.0
    assert multigpu.merge_bytes_per_rank(4096, 8) == 2 * 7 / 8 * 4096 * 4096
    assert multigpu.robot_offset_x(0) == 0.37 and abs(multigpu.robot_offset_x(1) + 0.33) < 1e-12


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)


def test_bench_gpus2_without_launcher_spawns_two_ranks():
    """`bench.py --gpus 2` with no launcher around it: two rank PROCESSES come up (torch.distributed.run as a child) and meet
    over gloo on 127.0.0.1; rank 0's one line is relayed."""
    out = _bench("--gpus", "2", "--launch-check")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"launch_check": True, "world_size": 2, "ranks_seen": 2, "processes": 2, "HSA_ENABLE_IPC_MODE_LEGACY": ["0", "0"]}


def test_launcher_started_ranks_have_the_ipc_mode_set():
    """The driver's form -- `python -m torch.distributed.run ... bench.py --gpus 2` with NOTHING about IPC in the environment: every
    rank must still come up with HSA_ENABLE_IPC_MODE_LEGACY=0 (RCCL's cross-process buffer sharing needs dmabuf handles on this pool;
    bench.py sets it in-process before anything can initialise HIP).  An explicit setting of the caller's wins."""
    import socket
    for given, want in ((None, "0"), ("1", "1")):
        e = dict(os.environ)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"):
            e.pop(k, None)
        if given is not None:
            e["HSA_ENABLE_IPC_MODE_LEGACY"] = given
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                             cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert d["launch_check"] is True and d["HSA_ENABLE_IPC_MODE_LEGACY"] == [want, want]


def test_bench_never_claims_gpus_it_does_not_have():
    """No GPUs for the ranks => non-zero exit and NO result line (round 2 printed n_gpus: 8 from one process)."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the GPUs")
    out = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--no-cpu-baseline")
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout[-500:]
    assert "GPUs" in out.stderr


def test_bench_refuses_a_launcher_world_that_is_not_gpus():
    out = _bench("--gpus", "8", "--steps", "5", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "WORLD_SIZE=2" in out.stderr
