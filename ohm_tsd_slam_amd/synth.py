"""Deterministic synthetic worlds and laser scans.

Closed-form ray intersections in fp64, ranges rounded to float32 exactly as a ``sensor_msgs/LaserScan``
would carry them.  Used identically by the parity tests, ``bench.py`` and ``smoke()``; nothing here is
read from the reference at run time.

Scenes
  room     axis-aligned box centred on the grid centre, half extents min(8, 0.4 W) x min(6, 0.3 W)
  pillars  box of half extent 20 m (clamped to 0.45 W) + 200 circular pillars, radii U(0.15, 0.5),
           none within 1.5 m of the start, numpy PCG64 seeded with 20261002
  comb     ranges alternate 5 m / 25 m every 8 beams (bandwidth stress; pose independent)
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

SEED = 20261002


@dataclass
class ScanGeometry:
    beams: int
    angle_min: float
    angle_increment: float

    @staticmethod
    def full_circle_360():
        # cfg 1: 360 beams, 1 degree over 360 degrees
        return ScanGeometry(360, -math.pi, 2.0 * math.pi / 360.0)

    @staticmethod
    def utm30lx():
        # cfgs 2-5: 1081 beams, 0.25 degree over 270 degrees (Hokuyo UTM-30LX shape)
        return ScanGeometry(1081, math.radians(-135.0), math.radians(270.0) / 1080.0)


@dataclass
class GridConfig:
    map_size_log2: int
    cell_size: float
    truncation_radius: int = 3

    @property
    def cells(self):
        return 1 << self.map_size_log2

    @property
    def width(self):
        return self.cells * self.cell_size

    @property
    def max_trunc(self):
        return self.truncation_radius * self.cell_size


# the BASELINE.json configurations
CONFIGS = {
    "cfg1": (GridConfig(9, 0.05), ScanGeometry.full_circle_360(), "room"),
    "cfg2": (GridConfig(12, 0.025), ScanGeometry.utm30lx(), "pillars"),
    "cfg3": (GridConfig(14, 0.01), ScanGeometry.utm30lx(), "pillars"),
}


class World:
    def __init__(self, scene: str, grid: GridConfig, start_xy=None):
        self.scene = scene
        W = grid.width
        self.cx = self.cy = 0.5 * W
        self.start = np.array(start_xy if start_xy is not None else [self.cx + 0.37, self.cy - 0.21])
        if scene == "room":
            self.hx, self.hy = min(8.0, 0.4 * W), min(6.0, 0.3 * W)
            self.circles = np.zeros((0, 3))
        elif scene == "pillars":
            self.hx = self.hy = min(20.0, 0.45 * W)
            rng = np.random.Generator(np.random.PCG64(SEED))
            cs = []
            while len(cs) < 200:
                x = rng.uniform(self.cx - self.hx, self.cx + self.hx)
                y = rng.uniform(self.cy - self.hy, self.cy + self.hy)
                r = rng.uniform(0.15, 0.5)
                if math.hypot(x - self.start[0], y - self.start[1]) < 1.5 + r:
                    continue
                cs.append((x, y, r))
            self.circles = np.array(cs)
        elif scene == "comb":
            self.hx = self.hy = 0.0
            self.circles = np.zeros((0, 3))
        else:
            raise ValueError(scene)

    def scan(self, x: float, y: float, yaw: float, geo: ScanGeometry) -> np.ndarray:
        """float32 ranges seen from (x, y, yaw)."""
        i = np.arange(geo.beams)
        if self.scene == "comb":
            return np.where((i // 8) % 2 == 1, 25.0, 5.0).astype(np.float32)
        phi = yaw + geo.angle_min + i * geo.angle_increment
        dx, dy = np.cos(phi), np.sin(phi)
        with np.errstate(divide="ignore", invalid="ignore"):
            tx = np.where(dx > 0, (self.cx + self.hx - x) / dx, (self.cx - self.hx - x) / dx)
            ty = np.where(dy > 0, (self.cy + self.hy - y) / dy, (self.cy - self.hy - y) / dy)
        tx = np.where(np.abs(dx) < 1e-300, np.inf, tx)
        ty = np.where(np.abs(dy) < 1e-300, np.inf, ty)
        t = np.minimum(tx, ty)
        if len(self.circles):
            ox = x - self.circles[:, 0][None, :]
            oy = y - self.circles[:, 1][None, :]
            b = ox * dx[:, None] + oy * dy[:, None]
            c = ox * ox + oy * oy - (self.circles[:, 2] ** 2)[None, :]
            disc = b * b - c
            with np.errstate(invalid="ignore"):
                tc = -b - np.sqrt(disc)
            tc = np.where((disc > 0) & (tc > 0), tc, np.inf)
            t = np.minimum(t, tc.min(axis=1))
        return t.astype(np.float32)


def trajectory(world: World, n: int, step_x=0.06, step_yaw=0.01, yaw0=0.1, leg=None):
    """Ground-truth poses: start at grid centre + (0.37, -0.21), yaw 0.1; 0.06 m along x and +0.01 rad
    per scan, so every scan exceeds the reference's 0.05 m push gate (ThreadLocalize.h:63-64).  After `leg`
    scans -- 5 m short of the scene's wall: 250 scans in the 20 m scenes, 50 in the 8 m room -- the robot
    drives back the same way, and so on, so that a run of any length stays inside the scene (the first `leg`
    poses are the plain straight line).  Closer to the wall the reference algorithm itself loses track on this
    scene (oracle and HIP alike, tools/long_run_check.py), which would turn a long benchmark into a no-op."""
    if leg is None:
        leg = max(50, int(round((world.hx - 5.0) / step_x))) if world.hx > 0 else 250
    k = np.arange(n)
    tri = leg - np.abs((k % (2 * leg)) - leg)          # 0, 1, .., leg, leg-1, .., 0, 1, ..
    return np.stack([world.start[0] + step_x * tri, world.start[1] + 0.0 * k, yaw0 + step_yaw * k], axis=1)


def free_lanes(world: World, n: int, length: float, spacing: float = 0.9, clearance: float = 0.9):
    """Start points of `n` robots in ONE world (the reference's multi-robot mode: N localisers, one grid): robot 0 starts
    at the world's own start, the others on lanes parallel to x at multiples of `spacing` in y, nearest first, keeping
    the lanes whose corridor [x0 - 1, x0 + length + 1] x [y - clearance, y + clearance] holds no pillar."""
    x0, y0 = float(world.start[0]), float(world.start[1])
    out = [(x0, y0)]
    k = 1
    while len(out) < n and k < 200:
        for sgn in (-1.0, 1.0):
            y = y0 + sgn * spacing * k
            if abs(y - world.cy) > world.hy - 2.0 and world.hy > 0:
                continue
            ok = True
            for (cx, cy, r) in world.circles:
                if (x0 - 1.0 - r) <= cx <= (x0 + length + 1.0 + r) and abs(cy - y) <= clearance + r:
                    ok = False
                    break
            if ok and len(out) < n:
                out.append((x0, y))
        k += 1
    if len(out) < n:
        raise ValueError("not enough free lanes in this world")
    return out


def scans_for(world: World, geo: ScanGeometry, poses: np.ndarray) -> np.ndarray:
    return np.stack([world.scan(p[0], p[1], p[2], geo) for p in poses])


def pose_matrix(x, y, yaw):
    c, s = math.cos(yaw), math.sin(yaw)
    return np.array([[c, -s, x], [s, c, y], [0.0, 0.0, 1.0]])
