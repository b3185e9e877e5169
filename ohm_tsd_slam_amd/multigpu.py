"""One robot + one TSD grid per GPU, and the ONE exchange step the multi-robot case has: merging the
ranks' occupancy maps.

The reference's multi-robot mode (SlamNode.cpp:101-122, config/double-laser.yaml, launch/multi_slam)
runs N ``ThreadLocalize`` workers against one shared ``TsdGrid`` in one process; nothing is exchanged
between robots except through that grid, and the only consumer of the combined map is
``ThreadGrid``'s ``nav_msgs/OccupancyGrid`` (-1 unknown / 0 free / 100 occupied, ThreadGrid.cpp:72-118).
Here every rank (one process per GPU, ``torch.distributed``; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests) owns its robot's grid; localise and push never communicate.  The
shared occupancy map is the element-wise maximum of the per-rank int8 maps (all robots use the same
grid geometry, their start poses differ by ``local_offset_*``): occupied (100) wins over free (0) wins
over unknown (-1), which is what writing all robots' scans into one grid converges to.

The collective itself lives behind the C ABI (``include/tsd_comm.h``, ``lib/libtsd_comm.so``): extraction kernels and
``ncclAllReduce(int8, max)`` enqueued on the grid context's own stream, so a C++ host can merge without Python or
torch.  :class:`NativeOccupancyMerger` is the thin ctypes caller ``bench.py`` uses; ``torch.distributed`` is only the
launcher's process group (rank / world size, the 128-byte unique id travels through it, barrier and max-over-ranks of
the bench contract).  The merge semantics are also checked with two ranks on CPU (gloo, ``tests/test_cpu_multigpu.py``,
through the test-only ``tests/gloo_merger.py``) where RCCL cannot run; with two or more GPUs visible
``tests/test_gpu_multigpu_nranks.py`` and ``bench.py --gpus N`` compare the native merge with the element-wise maximum
of the ranks' own maps.
"""
from __future__ import annotations

import os

import ctypes as C

import numpy as np

from . import capi

UNKNOWN, FREE, OCCUPIED = -1, 0, 100
COMM_LIB_PATH = os.path.join(capi.LIB_DIR, "libtsd_comm.so")
COMM_ID_BYTES = 128

COMM_ABI = {
    "tsd_comm_unique_id": (C.c_int, [C.c_char_p]),
    "tsd_comm_create": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "tsd_comm_destroy": (None, [C.c_void_p]),
    "tsd_comm_world_size": (C.c_int, [C.c_void_p]),
    "tsd_comm_rank": (C.c_int, [C.c_void_p]),
    "tsd_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "tsd_comm_occupancy_allreduce": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "tsd_comm_allreduce_map": (C.c_int, [C.c_void_p]),
    "tsd_comm_occupancy_wait": (C.c_int, [C.c_void_p, C.POINTER(C.c_int8)]),
    "tsd_comm_map_dev": (C.c_void_p, [C.c_void_p]),
    "tsd_comm_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "tsd_comm_merge_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}
_comm_lib = None


def load_comm_library():
    """``lib/libtsd_comm.so`` (links librccl + libtsd_hip); every symbol of include/tsd_comm.h is bound."""
    global _comm_lib
    if _comm_lib is None:
        capi.load_library()
        if not os.path.exists(COMM_LIB_PATH):
            raise capi.TsdError(f"{COMM_LIB_PATH} not found: run __graft_entry__.build()")
        lib = C.CDLL(COMM_LIB_PATH)
        for name, (res, args) in COMM_ABI.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _comm_lib = lib
    return _comm_lib


class NativeOccupancyMerger:
    """The RCCL merge through the C ABI: one communicator per grid context, ``merge_async`` enqueues the extraction
    kernels and ``ncclAllReduce(int8, max)`` on the context's stream (nothing waits), ``wait`` synchronises.
    ``unique_id`` comes from rank 0 (``NativeOccupancyMerger.new_id()``) and is handed to every rank by the launcher."""

    @staticmethod
    def new_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        if load_comm_library().tsd_comm_unique_id(buf) != 0:
            raise capi.TsdError("tsd_comm_unique_id failed")
        return buf.raw

    def __init__(self, grid, world_size: int, rank: int, unique_id: bytes):
        self.lib = load_comm_library()
        self.grid = grid
        self.cells = grid.cells
        assert len(unique_id) == COMM_ID_BYTES
        self.h = self.lib.tsd_comm_create(grid.h, world_size, rank, unique_id)
        if not self.h:
            raise capi.TsdError("tsd_comm_create failed (RCCL communicator)")

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsd_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise capi.TsdError(f"{what} failed ({rc}): {self.lib.tsd_comm_last_error(self.h).decode()}")

    def merge_async(self, inflate: bool = False, inflate_factor: int = 2):
        self._check(self.lib.tsd_comm_occupancy_allreduce(self.h, int(inflate), inflate_factor), "tsd_comm_occupancy_allreduce")

    def wait(self):
        self._check(self.lib.tsd_comm_occupancy_wait(self.h, None), "tsd_comm_occupancy_wait")

    def world_size(self) -> int:
        """what the RCCL communicator itself says (``tsd_comm_world_size``), not what the launcher's environment claims"""
        return int(self.lib.tsd_comm_world_size(self.h))

    def profile(self, on: bool = True):
        self._check(self.lib.tsd_comm_profile(self.h, int(on)), "tsd_comm_profile")

    def merge_times(self):
        """(extraction ms, all-reduce ms, merges timed): totals over the merges issued while ``profile`` was on"""
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        self._check(self.lib.tsd_comm_merge_times(self.h, C.byref(a), C.byref(b), C.byref(n)), "tsd_comm_merge_times")
        return a.value, b.value, n.value

    def merged(self) -> np.ndarray:
        out = np.empty(self.cells * self.cells, dtype=np.int8)
        self._check(self.lib.tsd_comm_occupancy_wait(self.h, out.ctypes.data_as(C.POINTER(C.c_int8))), "tsd_comm_occupancy_wait")
        return out.reshape(self.cells, self.cells)


def env_rank():
    """(rank, local_rank, world_size) as ``torch.distributed.run`` exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def robot_offset_x(rank: int, base: float = 0.37, spacing: float = 0.7) -> float:
    """``local_offset_x`` of robot ``rank``: robots start ``spacing`` metres apart along -x
    (launch/multi_slam.launch:40 uses -0.7 for the second robot)."""
    return base - spacing * rank


def merge_bytes_per_rank(cells: int, world_size: int) -> float:
    """Bytes each rank sends in a ring all-reduce of the int8 map: 2 (W-1)/W of the map (DESIGN.md
    "Multi-GPU")."""
    if world_size <= 1:
        return 0.0
    return 2.0 * (world_size - 1) / world_size * cells * cells
