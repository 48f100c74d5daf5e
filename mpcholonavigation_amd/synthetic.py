"""Synthetic inputs for the benchmark configurations.

A seeded random costmap with Nav2-style inflation, a straight +x plan, a pose
at the map-centre row and a warm-started control sequence — identical for the
GPU path and the CPU oracle.  Nothing here is on the hot path.

Costmap recipe: SplitMix64(seed) places W*H/4000 lethal discs (radius 2-6
cells); a corridor of half-width 0.4 m around the plan and a 1 m disc at the
start are cleared of lethal cells; then every cell gets the InflationLayer cost
of its distance d to the nearest lethal cell: 254 at d=0, 253 for d <= r_in,
(uint8)(252*exp(-k*(d-r_in))) for d <= R, else 0 (nav2_costmap_2d
InflationLayer::computeCost, the formula ObstaclesCritic::distanceToObstacle
inverts, reference src/critics/obstacles_critic.cpp:99-112).
"""
from dataclasses import dataclass

import numpy as np

from .tick import Tick

RESOLUTION = 0.05
INSCRIBED_RADIUS = 0.1
COST_SCALING_FACTOR = 10.0
INFLATION_RADIUS = 0.55


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n


@dataclass
class Scenario:
    cells: np.ndarray          # uint8 [H, W]
    origin_x: float
    origin_y: float
    resolution: float
    tick: Tick
    u0: np.ndarray             # float32 [3, T] warm start
    inscribed_radius: float = INSCRIBED_RADIUS
    cost_scaling_factor: float = COST_SCALING_FACTOR
    inflation_radius: float = INFLATION_RADIUS


def _inflate(lethal, res, r_in, k, R):
    from scipy import ndimage
    d = ndimage.distance_transform_edt(~lethal) * res
    cost = np.zeros(lethal.shape, np.uint8)
    band = (d > r_in) & (d <= R)
    cost[band] = (252.0 * np.exp(-k * (d[band] - r_in))).astype(np.uint8)
    cost[(d <= r_in)] = 253
    cost[lethal] = 254
    return cost


def make_costmap(width, height, plan_y, start_x, seed=42, res=RESOLUTION, all_lethal=False):
    """uint8 [height, width] costmap following the recipe in the module docstring."""
    if all_lethal:
        return np.full((height, width), 254, np.uint8)
    rng = SplitMix64(seed)
    lethal = np.zeros((height, width), bool)
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(max(1, width * height // 4000)):
        cx, cy, r = rng.below(width), rng.below(height), 2 + rng.below(5)
        x0, x1 = max(cx - r, 0), min(cx + r + 1, width)
        y0, y1 = max(cy - r, 0), min(cy + r + 1, height)
        sub = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= r * r
        lethal[y0:y1, x0:x1] |= sub
    # keep the corridor around the plan and a disc at the start free of lethal cells
    wy = (yy + 0.5) * res
    wx = (xx + 0.5) * res
    corridor = (np.abs(wy - plan_y) <= 0.4) & (wx >= start_x - 0.4)
    disc = (wx - start_x) ** 2 + (wy - plan_y) ** 2 <= 1.0
    lethal &= ~(corridor | disc)
    return _inflate(lethal, res, INSCRIBED_RADIUS, COST_SCALING_FACTOR, INFLATION_RADIUS)


def make_scenario(time_steps, map_size=200, seed=42, near_goal=False, all_lethal=False,
                  speed=(0.3, 0.0, 0.0), warm_vx=0.3, path_points=60) -> Scenario:
    """Cruise (default) or near-goal scenario on a map_size x map_size costmap."""
    res = RESOLUTION
    W = H = map_size
    pose_x = W * res / 4.0
    pose_y = H * res / 2.0
    cells = make_costmap(W, H, pose_y, pose_x, seed=seed, res=res, all_lethal=all_lethal)
    if near_goal:
        P = 9          # goal 0.4 m ahead: GoalAngle live, the others gated off
    else:
        P = int(min(path_points, (W * res - pose_x) / res - 1))
    path_x = (pose_x + res * np.arange(P)).astype(np.float32)
    path_y = np.full(P, pose_y, np.float32)
    path_yaw = np.zeros(P, np.float32)
    if near_goal:
        path_yaw[-1] = 0.7
    tick = Tick(pose_x=pose_x, pose_y=pose_y, pose_yaw=0.0, speed=tuple(speed),
                path_x=path_x, path_y=path_y, path_yaw=path_yaw,
                goal_x=float(path_x[-1]), goal_y=float(path_y[-1]))
    u0 = np.zeros((3, time_steps), np.float32)
    u0[0, :] = warm_vx
    return Scenario(cells=cells, origin_x=0.0, origin_y=0.0, resolution=res, tick=tick, u0=u0)


def make_noise(batch, time_steps, std=(0.2, 0.2, 0.4), seed=1234):
    """Host noise tensors for parity runs: draw order vx, wz, vy
    (reference src/noise_generator.cpp:107-122); returns (nvx, nvy, nwz)."""
    g = np.random.Generator(np.random.PCG64(seed))
    nvx = g.standard_normal((batch, time_steps), dtype=np.float32) * np.float32(std[0])
    nwz = g.standard_normal((batch, time_steps), dtype=np.float32) * np.float32(std[2])
    nvy = g.standard_normal((batch, time_steps), dtype=np.float32) * np.float32(std[1])
    return nvx, nvy, nwz
