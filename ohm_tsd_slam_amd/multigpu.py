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

``torch`` is plumbing here (device buffer + process group), not the compute path: the per-rank map is
produced by the HIP kernels behind ``tsd_occupancy_dev``.
"""
from __future__ import annotations

import os

UNKNOWN, FREE, OCCUPIED = -1, 0, 100


def env_rank():
    """(rank, local_rank, world_size) as ``torch.distributed.run`` exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def robot_offset_x(rank: int, base: float = 0.37, spacing: float = 0.7) -> float:
    """``local_offset_x`` of robot ``rank``: robots start ``spacing`` metres apart along -x
    (launch/multi_slam.launch:40 uses -0.7 for the second robot)."""
    return base - spacing * rank


class OccupancyMerger:
    """Max all-reduce of the int8 occupancy map across the process group.

    ``buffer`` is a flat int8 tensor of ``cells * cells`` elements on the rank's device (CPU tensors
    with gloo).  ``merge_async`` starts the collective and returns immediately so that it overlaps
    the next scans' ray-cast / ICP / push kernels, which run on the grid context's own HIP stream;
    ``wait`` blocks until the merged map is in ``buffer``."""

    def __init__(self, cells: int, device=None, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.group = group
        self.cells = cells
        self.buffer = torch.full((cells * cells,), UNKNOWN, dtype=torch.int8, device=device)
        self._work = None

    @property
    def active(self) -> bool:
        return self._dist.is_available() and self._dist.is_initialized() and self._dist.get_world_size(self.group) > 1

    def fill_from_grid(self, grid, inflate: bool = False, inflate_factor: int = 2):
        """Run the occupancy extraction kernels of ``grid`` (a ``capi.TsdGridDevice``) into the buffer."""
        self.wait()
        grid.occupancy_into(self.buffer.data_ptr(), inflate, inflate_factor)

    def fill_from_host(self, occ_int8):
        """CPU path of the tests: take a host map as this rank's contribution."""
        self.wait()
        t = self._torch.as_tensor(occ_int8, dtype=self._torch.int8).reshape(-1)
        self.buffer.copy_(t)

    def merge_async(self, force: bool = False):
        """``force`` issues the collective even in a one-rank group (plumbing check on a single GPU)."""
        self.wait()
        if self.active or (force and self._dist.is_initialized()):
            self._work = self._dist.all_reduce(self.buffer, op=self._dist.ReduceOp.MAX, group=self.group, async_op=True)
        return self._work

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None

    def merged(self):
        """The merged map as a (cells, cells) int8 tensor (row = y, column = x like OccupancyGrid.data)."""
        self.wait()
        return self.buffer.view(self.cells, self.cells)


def merge_bytes_per_rank(cells: int, world_size: int) -> float:
    """Bytes each rank sends in a ring all-reduce of the int8 map: 2 (W-1)/W of the map (DESIGN.md
    "Multi-GPU")."""
    if world_size <= 1:
        return 0.0
    return 2.0 * (world_size - 1) / world_size * cells * cells
