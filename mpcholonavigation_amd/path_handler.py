"""Python face of sortham::PathHandler and sortham::TrajectoryVisualizer for plain types
(libsortham_host.so, include/smpc_host.h; reference src/path_handler.cpp,
src/trajectory_visualizer.cpp).  Poses are [n, 3] float64 arrays {x, y, yaw}; what tf2
supplies in the reference is passed in: the robot pose in the plan's frame and the rigid
transform (tx, ty, yaw) from the plan's frame to the costmap's (None = identity).  Where the
reference throws std::runtime_error this raises RuntimeError with the same message."""
import ctypes as C

import numpy as np

from . import host_optimizer as H

SORTHAM_ERR_THROWN = -10


class PathHandlerConfig(C.Structure):
    _fields_ = [
        ("costmap_size_x", C.c_uint32), ("costmap_size_y", C.c_uint32),
        ("costmap_resolution", C.c_double), ("costmap_origin_x", C.c_double), ("costmap_origin_y", C.c_double),
        ("max_robot_pose_search_dist", C.c_double), ("prune_distance", C.c_double),
        ("enforce_path_inversion", C.c_int32),
        ("inversion_xy_tolerance", C.c_float), ("inversion_yaw_tolerance", C.c_float),
    ]


_h = C.c_void_p
_dp = C.POINTER(C.c_double)
PROTOTYPES = {
    "sortham_path_handler_config_default": (None, [C.POINTER(PathHandlerConfig)]),
    "sortham_path_handler_create": (C.c_int, [C.POINTER(PathHandlerConfig), C.POINTER(_h)]),
    "sortham_path_handler_destroy": (None, [_h]),
    "sortham_path_handler_last_error": (C.c_char_p, [_h]),
    "sortham_path_handler_set_path": (C.c_int, [_h, C.c_void_p, C.c_uint32]),
    "sortham_path_handler_get_path": (C.c_uint32, [_h, C.c_int, C.c_void_p, C.c_uint32]),
    "sortham_path_handler_transform_path": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                      C.POINTER(C.c_uint32)]),
    "sortham_path_handler_plan_in_bounds": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sortham_path_handler_prune": (C.c_int, [_h, C.c_int, C.c_uint32]),
    "sortham_path_handler_transformed_goal": (C.c_int, [_h, C.c_void_p, C.c_void_p]),
    "sortham_path_handler_within_inversion_tolerances": (C.c_int, [_h, C.c_void_p]),
    "sortham_path_handler_max_costmap_dist": (C.c_double, [_h]),
    "sortham_utils_find_first_path_inversion": (C.c_uint32, [C.c_void_p, C.c_uint32]),
    "sortham_utils_remove_poses_after_first_inversion": (C.c_uint32, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "sortham_visualizer_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(_h)]),
    "sortham_visualizer_destroy": (None, [_h]),
    "sortham_visualizer_add_trajectory": (C.c_int, [_h, C.c_void_p, C.c_uint32, C.c_uint32]),
    "sortham_visualizer_add_candidates": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "sortham_visualizer_visualize": (C.c_uint32, [_h, C.c_void_p, C.c_uint32]),
    "sortham_visualizer_frame": (C.c_char_p, [_h]),
}
_lib = None


def load_library():
    global _lib
    if _lib is None:
        from . import _abi as A
        H.load_library()
        _lib = A.bind(C.CDLL(H.LIB_PATH), PROTOTYPES)
    return _lib


def _poses(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 3)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("poses must be [n, 3] {x, y, yaw}")
    return a


def _vec3(v):
    return None if v is None else np.ascontiguousarray(v, dtype=np.float64).reshape(3)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def find_first_path_inversion(poses):
    """utils::findFirstPathInversion (tools/utils.hpp:612-639)."""
    a = _poses(poses)
    return int(load_library().sortham_utils_find_first_path_inversion(_p(a), len(a)))


def remove_poses_after_first_inversion(poses):
    """utils::removePosesAfterFirstInversion (tools/utils.hpp:646-658) -> (inversion index, cropped poses)."""
    a = _poses(poses).copy()
    n = C.c_uint32(len(a))
    r = load_library().sortham_utils_remove_poses_after_first_inversion(_p(a), C.byref(n))
    return int(r), a[:n.value].copy()


class PathHandler:
    def __init__(self, costmap_size=(100, 100), resolution=0.05, origin=(0.0, 0.0), **params):
        self.lib = load_library()
        cfg = PathHandlerConfig()
        self.lib.sortham_path_handler_config_default(C.byref(cfg))
        cfg.costmap_size_x, cfg.costmap_size_y = costmap_size
        cfg.costmap_resolution = resolution
        cfg.costmap_origin_x, cfg.costmap_origin_y = origin
        for k, v in params.items():
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
        h = _h()
        if self.lib.sortham_path_handler_create(C.byref(cfg), C.byref(h)) != 0:
            raise RuntimeError("sortham_path_handler_create failed")
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.sortham_path_handler_destroy(self.h)
            self.h = None

    def _ck(self, rc):
        if rc == SORTHAM_ERR_THROWN:
            raise RuntimeError(self.lib.sortham_path_handler_last_error(self.h).decode())
        if rc != 0:
            raise ValueError(f"path handler error {rc}")

    def set_path(self, poses):
        a = _poses(poses)
        self._ck(self.lib.sortham_path_handler_set_path(self.h, _p(a), len(a)))

    def get_path(self, up_to_inversion=False):
        n = self.lib.sortham_path_handler_get_path(self.h, int(up_to_inversion), None, 0)
        out = np.zeros((n, 3))
        self.lib.sortham_path_handler_get_path(self.h, int(up_to_inversion), _p(out), n)
        return out

    def max_costmap_dist(self):
        return float(self.lib.sortham_path_handler_max_costmap_dist(self.h))

    def plan_in_bounds(self, robot_pose, plan_to_costmap=None):
        """getGlobalPlanConsideringBoundsInCostmapFrame -> (plan in the costmap frame, closest index)."""
        r, t = _vec3(robot_pose), _vec3(plan_to_costmap)
        cap = max(1, self.lib.sortham_path_handler_get_path(self.h, 1, None, 0))
        out = np.zeros((cap, 3))
        n, closest = C.c_uint32(0), C.c_uint32(0)
        self._ck(self.lib.sortham_path_handler_plan_in_bounds(self.h, _p(r), _p(t), _p(out), cap, C.byref(n),
                                                              C.byref(closest)))
        return out[:n.value].copy(), int(closest.value)

    def prune(self, end, up_to_inversion=False):
        self._ck(self.lib.sortham_path_handler_prune(self.h, int(up_to_inversion), int(end)))

    def transform_path(self, robot_pose, plan_to_costmap=None):
        r, t = _vec3(robot_pose), _vec3(plan_to_costmap)
        cap = max(1, self.lib.sortham_path_handler_get_path(self.h, 1, None, 0))
        out = np.zeros((cap, 3))
        n = C.c_uint32(0)
        self._ck(self.lib.sortham_path_handler_transform_path(self.h, _p(r), _p(t), _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def transformed_goal(self, plan_to_costmap=None):
        out = np.zeros(3)
        self._ck(self.lib.sortham_path_handler_transformed_goal(self.h, _p(_vec3(plan_to_costmap)), _p(out)))
        return out

    def within_inversion_tolerances(self, robot_pose):
        return bool(self.lib.sortham_path_handler_within_inversion_tolerances(self.h, _p(_vec3(robot_pose))))


class TrajectoryVisualizer:
    """Markers come back as dicts: id, position (x, y, z), scale (x, y, z), color (r, g, b, a)."""

    def __init__(self, frame_id="map", trajectory_step=5, time_step=3):
        self.lib = load_library()
        h = _h()
        if self.lib.sortham_visualizer_create(frame_id.encode(), trajectory_step, time_step, C.byref(h)) != 0:
            raise ValueError("bad visualizer parameters")
        self.h = h
        self._pending = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.sortham_visualizer_destroy(self.h)
            self.h = None

    @property
    def frame_id(self):
        return self.lib.sortham_visualizer_frame(self.h).decode()

    def add_trajectory(self, xy):
        a = np.ascontiguousarray(xy, dtype=np.float32)
        if a.size == 0:
            return
        self.lib.sortham_visualizer_add_trajectory(self.h, _p(a), a.shape[0], a.shape[1])
        self._pending += a.shape[0]

    def add_candidates(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        self.lib.sortham_visualizer_add_candidates(self.h, _p(x), _p(y), x.shape[0], x.shape[1])
        self._pending += x.size

    def visualize(self):
        cap = max(1, self._pending)
        buf = np.zeros((cap, 10))
        n = self.lib.sortham_visualizer_visualize(self.h, _p(buf), cap)
        self._pending = 0
        return [dict(id=int(m[0]), position=(m[1], m[2], m[3]), scale=(m[4], m[5], m[6]),
                     color=(0.0, m[7], m[8], m[9]), frame_id=self.frame_id) for m in buf[:n]]
