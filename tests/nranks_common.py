"""What the N-rank occupancy-merge checks share (CPU: tests/test_cpu_multigpu.py on the oracle's maps; GPU:
tests/nranks_check.py on the HIP path's maps): ONE room for all robots, robot r starting ``multigpu.robot_offset_x(r)``
from the grid centre (launch/multi_slam.launch:40: the second robot 0.7 m behind the first), and the property that makes
the per-rank maps mergeable at all -- they are expressed in a COMMON map frame (every rank uses the same grid geometry;
the per-robot start offsets only move the sensor inside it, ThreadLocalize.cpp:466-468), so the same wall is marked in the
same cells whoever saw it."""
import numpy as np

from ohm_tsd_slam_amd import multigpu, synth


def setup(map_size_log2: int = 9):
    # (512^2 cells @ 0.05 m = 25.6 m: the 16 x 12 m room lies inside the tiles the extraction visits -- it skips the outer ring)
    gc = synth.GridConfig(map_size_log2, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)          # the room is centred on the grid centre: the same world for every rank
    return gc, geo, world


def robot_pose(world, rank: int, k: int, step_x=0.06, step_yaw=0.01, yaw0=0.1):
    """ground-truth pose of robot `rank` at its k-th scan: (3x3 matrix, (x, y, yaw))"""
    x = world.cx + multigpu.robot_offset_x(rank) + step_x * k
    y = world.cy - 0.21
    yaw = yaw0 + step_yaw * k
    return synth.pose_matrix(x, y, yaw), (x, y, yaw)


def _dilate1(m):
    d = m.copy()
    d[1:, :] |= m[:-1, :]; d[:-1, :] |= m[1:, :]; d[:, 1:] |= m[:, :-1]; d[:, :-1] |= m[:, 1:]
    return d


def wall_agreement(a, b):
    """fraction of a's occupied cells that have an occupied cell of b within one cell (the zero crossing of a wall that
    sits exactly on a cell boundary lands in either of the two cells next to it, depending on the view point)"""
    oa, ob = (a == 100), (b == 100)
    return float((oa & _dilate1(ob)).sum()) / max(int(oa.sum()), 1)


def assert_common_frame(a, b, gc, rank_a: int, rank_b: int):
    """a, b: (cells, cells) int8 maps of two robots in the room of `setup`.  Both see all four walls (the room is convex):
    >= 99 % of each map's wall cells have a wall cell of the other within one cell, and the same maps compared in the
    robots' OWN start frames (b shifted by the difference of the offsets) do NOT agree -- the check can fail."""
    assert (a == 100).sum() > 200 and (b == 100).sum() > 200, "the robots must have seen the room's walls"
    ab, ba = wall_agreement(a, b), wall_agreement(b, a)
    assert ab >= 0.99 and ba >= 0.99, f"ranks {rank_a}/{rank_b}: the same walls are not in the same cells ({ab:.3f}, {ba:.3f})"
    shift = int(round((multigpu.robot_offset_x(rank_a) - multigpu.robot_offset_x(rank_b)) / gc.cell_size))
    if abs(shift) >= 3:
        # (rows well inside the room hold only the two walls that run along y, which an offset along x moves)
        hy = min(6.0, 0.3 * gc.width)
        r0, r1 = int((0.5 * gc.width - hy + 0.5) / gc.cell_size), int((0.5 * gc.width + hy - 0.5) / gc.cell_size)
        wrong = wall_agreement(a[r0:r1], np.roll(b, shift, axis=1)[r0:r1])
        assert wall_agreement(a[r0:r1], b[r0:r1]) >= 0.99
        assert wrong < 0.2, f"a map shifted by {shift} cells still agrees ({wrong:.3f}): the check is blind"
    return ab, ba
