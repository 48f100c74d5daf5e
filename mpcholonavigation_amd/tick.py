"""Per-tick inputs of the hot path: the Python face of smpc_tick_in.

Mirrors what Optimizer::prepare() copies into the optimizer each
computeVelocityCommands() tick (reference src/optimizer.cpp:185-204): robot
pose and speed, the pruned plan as three float tensors
(utils::toTensor, tools/utils.hpp:180-192) and the goal.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _abi as A


@dataclass
class Tick:
    pose_x: float
    pose_y: float
    pose_yaw: float
    speed: tuple            # (vx, vy, wz) of state.speed
    path_x: np.ndarray      # [P] float32
    path_y: np.ndarray
    path_yaw: np.ndarray
    goal_x: float
    goal_y: float
    path_pts_valid: Optional[np.ndarray] = None   # [P-1] uint8 or None (derive from costmap)
    fail_flag_in: bool = False
    goal_checker_xy_tolerance: float = -1.0       # GoalChecker xy tolerance; < 0 = no goal checker
    _c: Optional[A.SmpcTickIn] = field(default=None, repr=False, compare=False)

    def __post_init__(self):
        self.path_x = np.ascontiguousarray(self.path_x, dtype=np.float32)
        self.path_y = np.ascontiguousarray(self.path_y, dtype=np.float32)
        self.path_yaw = np.ascontiguousarray(self.path_yaw, dtype=np.float32)
        if not (self.path_x.shape == self.path_y.shape == self.path_yaw.shape) or \
                self.path_x.ndim != 1:
            raise ValueError("path_x, path_y, path_yaw must be 1-D and of equal length")
        if self.path_pts_valid is not None:
            self.path_pts_valid = np.ascontiguousarray(self.path_pts_valid, dtype=np.uint8)
            if self.path_pts_valid.shape != (max(len(self.path_x) - 1, 0),):
                raise ValueError("path_pts_valid must have P-1 entries")

    def __setattr__(self, name, value):
        # any field assignment invalidates the cached C struct (arrays edited in place do not
        # need to: the struct points at them)
        object.__setattr__(self, name, value)
        if name != "_c":
            object.__setattr__(self, "_c", None)

    @property
    def c(self) -> A.SmpcTickIn:
        """The C struct (built once per change of a field: it is on the tick's critical path);
        arrays stay owned (and kept alive) by this object."""
        if self._c is not None:
            return self._c
        t = A.SmpcTickIn()
        t.pose_x, t.pose_y, t.pose_yaw = self.pose_x, self.pose_y, self.pose_yaw
        t.speed_vx, t.speed_vy, t.speed_wz = self.speed
        f32p = C.POINTER(C.c_float)
        t.path_x = self.path_x.ctypes.data_as(f32p)
        t.path_y = self.path_y.ctypes.data_as(f32p)
        t.path_yaw = self.path_yaw.ctypes.data_as(f32p)
        t.path_len = len(self.path_x)
        t.goal_x, t.goal_y = self.goal_x, self.goal_y
        if self.path_pts_valid is not None:
            t.path_pts_valid = self.path_pts_valid.ctypes.data_as(C.POINTER(C.c_uint8))
        t.fail_flag_in = int(self.fail_flag_in)
        t.goal_checker_xy_tolerance = float(self.goal_checker_xy_tolerance)
        self._c = t
        return t


def default_config(**kw) -> A.SmpcConfig:
    """Optimizer::getParams() defaults (src/optimizer.cpp:69-82), Omni model."""
    c = A.SmpcConfig()
    c.batch_size, c.time_steps, c.iteration_count = 1000, 56, 1
    c.motion_model = A.SMPC_MODEL_OMNI
    c.model_dt, c.temperature, c.gamma = 0.05, 0.3, 0.015
    c.vx_max, c.vx_min, c.vy_max, c.wz_max = 0.5, -0.35, 0.5, 1.9
    c.vx_std, c.vy_std, c.wz_std = 0.2, 0.2, 0.4
    c.device, c.flags = -1, 0
    c.ackermann_min_turning_r = 0.2   # include/.../motion_models.hpp:94
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def default_critics() -> A.SmpcCriticParams:
    """Every critic's initialize() defaults; the north star's five enabled."""
    p = A.SmpcCriticParams()
    o = p.obstacles    # src/critics/obstacles_critic.cpp:21-31
    o.enabled, o.consider_footprint, o.cost_power = 1, 0, 1
    o.repulsion_weight, o.critical_weight, o.collision_cost = 1.5, 20.0, 10000.0
    o.collision_margin_distance, o.near_goal_distance = 0.10, 0.5
    a = p.path_align   # src/critics/path_align_critic.cpp:26-38
    a.enabled, a.use_path_orientations, a.cost_power, a.cost_weight = 1, 0, 1, 10.0
    a.max_path_occupancy_ratio, a.offset_from_furthest = 0.07, 20
    a.trajectory_point_step, a.threshold_to_consider = 4, 0.5
    f = p.path_follow  # src/critics/path_follow_critic.cpp:23-33
    f.enabled, f.cost_power, f.cost_weight = 1, 1, 5.0
    f.threshold_to_consider, f.offset_from_furthest = 1.4, 6
    g = p.goal_angle   # src/critics/goal_angle_critic.cpp:20-27
    g.enabled, g.cost_power, g.cost_weight, g.threshold_to_consider = 1, 1, 3.0, 0.5
    w = p.prefer_forward  # src/critics/prefer_forward_critic.cpp:20-27
    w.enabled, w.cost_power, w.cost_weight, w.threshold_to_consider = 1, 1, 5.0, 0.5
    # the other registered critics: their initialize() defaults, not in the list (enabled 0)
    c = p.cost         # src/critics/cost_critic.cpp:25-31
    c.enabled, c.consider_footprint, c.cost_power, c.cost_weight = 0, 0, 1, 3.81
    c.critical_cost, c.collision_cost, c.near_goal_distance = 300.0, 1000000.0, 0.5
    gl = p.goal        # src/critics/goal_critic.cpp:26-28
    gl.enabled, gl.cost_power, gl.cost_weight, gl.threshold_to_consider = 0, 1, 5.0, 1.4
    k = p.constraint   # src/critics/constraint_critic.cpp:27-35 (parent vx_max, vy_max, vx_min)
    k.enabled, k.cost_power, k.cost_weight = 0, 1, 4.0
    k.vx_max, k.vy_max, k.vx_min = 0.5, 0.5, -0.35
    tw = p.twirling    # src/critics/twirling_critic.cpp:24-25
    tw.enabled, tw.cost_power, tw.cost_weight = 0, 1, 10.0
    pa = p.path_angle  # src/critics/path_angle_critic.cpp:24-45
    pa.enabled, pa.cost_power, pa.cost_weight, pa.offset_from_furthest = 0, 1, 2.0, 4
    pa.threshold_to_consider, pa.max_angle_to_furthest = 0.5, 1.2
    pa.forward_preference, pa.vx_min = 1, -0.35
    vd = p.velocity_deadband  # src/critics/velocity_deadband_critic.cpp:24-33
    vd.enabled, vd.cost_power, vd.cost_weight = 0, 1, 35.0
    lg = p.path_align_legacy  # src/critics/path_align_legacy_critic.cpp:26-37
    lg.enabled, lg.use_path_orientations, lg.cost_power, lg.cost_weight = 0, 0, 1, 10.0
    lg.max_path_occupancy_ratio, lg.offset_from_furthest = 0.07, 20
    lg.trajectory_point_step, lg.threshold_to_consider = 4, 0.5
    return p
