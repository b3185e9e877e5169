"""TEST-ONLY twin of the native RCCL occupancy merge (include/tsd_comm.h): the same element-wise maximum over
``torch.distributed`` tensors, so that the merge semantics can be checked with two ranks on CPU (gloo) where RCCL cannot
run.  Nothing in the product, ``bench.py`` or the facade uses it."""
from __future__ import annotations

UNKNOWN, FREE, OCCUPIED = -1, 0, 100


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
