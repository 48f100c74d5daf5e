"""Python face of the C++ host optimizer (libsortham_host.so, include/smpc_host.h):
sortham::Optimizer of the reference with the [batch, time] work on the GPU.

Method names follow the reference (src/optimizer.cpp): eval_control = evalControl,
set_speed_limit = setSpeedLimit, reset = reset, get_optimized_trajectory =
getOptimizedTrajectory.  Where the reference throws std::runtime_error this
raises RuntimeError with the same message."""
import ctypes as C
import os

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsortham_host.so")
SORTHAM_ERR_THROWN = -10
_lib = None


class SorthamOptimizerConfig(C.Structure):
    _fields_ = [
        ("base", A.SmpcConfig),
        ("controller_frequency", C.c_double),
        ("retry_attempt_limit", C.c_uint32),
        ("regenerate_noises", C.c_int32),
        ("visualize", C.c_int32),
        ("noise_seed", C.c_uint64),
        ("critics", C.c_char_p * 16),
        ("n_critics", C.c_uint32),
        ("cost_scaling_factor", C.c_float),
        ("inflation_radius", C.c_float),
        ("motion_model", C.c_char_p),
    ]


_ctx = C.c_void_p
PROTOTYPES = {
    "sortham_optimizer_create": (C.c_int, [C.POINTER(SorthamOptimizerConfig),
                                           C.POINTER(A.SmpcCriticParams), C.POINTER(_ctx)]),
    "sortham_optimizer_initialize": (C.c_int, [_ctx, C.POINTER(SorthamOptimizerConfig),
                                               C.POINTER(A.SmpcCriticParams)]),
    "sortham_optimizer_destroy": (None, [_ctx]),
    "sortham_optimizer_last_error": (C.c_char_p, [_ctx]),
    "sortham_optimizer_set_costmap": (C.c_int, [_ctx, C.c_void_p, C.c_uint32, C.c_uint32,
                                                C.c_double, C.c_double, C.c_double, C.c_int,
                                                C.c_float, C.c_int]),
    "sortham_optimizer_set_noise": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sortham_optimizer_eval_control": (C.c_int, [_ctx, C.POINTER(A.SmpcTickIn), C.c_void_p,
                                                 C.POINTER(A.SmpcTickOut)]),
    "sortham_optimizer_set_speed_limit": (C.c_int, [_ctx, C.c_double, C.c_int]),
    "sortham_optimizer_reset": (C.c_int, [_ctx]),
    "sortham_optimizer_get_control_sequence": (C.c_int, [_ctx, C.c_void_p]),
    "sortham_optimizer_set_control_sequence": (C.c_int, [_ctx, C.c_void_p]),
    "sortham_optimizer_get_constraints": (C.c_int, [_ctx, C.c_void_p, C.POINTER(C.c_int32)]),
    "sortham_optimizer_get_optimized_trajectory": (C.c_int, [_ctx, C.c_void_p]),
    "sortham_utils_savitsky_golay": (None, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]),
    "sortham_run_ticks": (C.c_int, [_ctx, C.POINTER(A.SmpcTickIn), C.c_void_p, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.POINTER(A.SmpcTickOut), C.POINTER(C.c_uint32)]),
}

TICKS_SHIFT, TICKS_REDRAW_ASYNC, TICKS_SHARD, TICKS_SHARD_SPECULATE = 1, 2, 4, 8

DEFAULT_CRITICS = ["ObstaclesCritic", "PathAlignCritic", "PathFollowCritic", "GoalAngleCritic",
                   "PreferForwardCritic"]


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: run `make -C mpcholonavigation_amd/csrc`")
        _lib = A.bind(C.CDLL(LIB_PATH), PROTOTYPES)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def run_ticks(smpc, tick, u, n, flags=TICKS_SHIFT):
    """sortham_run_ticks: n closed-loop ticks of the low-level context `smpc` (optimizer.Smpc) issued
    by the compiled loop of host/tick_loop.cpp -> (u after the last tick [and shift], [SmpcTickOut] * n).
    Raises SmpcError with the ticks completed in the message."""
    lib = load_library()
    u = np.array(u, dtype=np.float32, order="C")
    if u.shape != (3, smpc.T):
        raise ValueError(f"u must be [3, {smpc.T}]")
    outs = (A.SmpcTickOut * n)()
    done = C.c_uint32(0)
    rc = lib.sortham_run_ticks(smpc.h, C.byref(tick.c), _ptr(u), smpc.T, n, flags, outs, C.byref(done))
    if rc != 0:
        from .optimizer import SmpcError
        raise SmpcError(rc, f"after {done.value} of {n} ticks: " + (smpc.lib.smpc_last_error(smpc.h) or b"").decode())
    return u, outs


class Optimizer:
    def __init__(self, cfg: A.SmpcConfig, critic_params: A.SmpcCriticParams,
                 controller_frequency, critics=None, motion_model="Omni", retry_attempt_limit=1,
                 regenerate_noises=False, visualize=False, noise_seed=0,
                 cost_scaling_factor=10.0, inflation_radius=0.55):
        self.lib = load_library()
        c = self._config(cfg, controller_frequency, critics, motion_model, retry_attempt_limit,
                         regenerate_noises, visualize, noise_seed, cost_scaling_factor, inflation_radius)
        h = _ctx()
        rc = self.lib.sortham_optimizer_create(C.byref(c), C.byref(critic_params), C.byref(h))
        if rc != 0:
            raise RuntimeError(self.lib.sortham_optimizer_last_error(None).decode() or f"error {rc}")
        self.h = h

    def initialize(self, cfg: A.SmpcConfig, critic_params: A.SmpcCriticParams, controller_frequency,
                   critics=None, motion_model="Omni", retry_attempt_limit=1, regenerate_noises=False,
                   visualize=False, noise_seed=0, cost_scaling_factor=10.0, inflation_radius=0.55):
        """Optimizer::initialize() again on the live object (the plugin's reset()): an unchanged
        configuration keeps the device context, anything else rebuilds it."""
        c = self._config(cfg, controller_frequency, critics, motion_model, retry_attempt_limit,
                         regenerate_noises, visualize, noise_seed, cost_scaling_factor, inflation_radius)
        self._ck(self.lib.sortham_optimizer_initialize(self.h, C.byref(c), C.byref(critic_params)))

    def _config(self, cfg, controller_frequency, critics, motion_model, retry_attempt_limit,
                regenerate_noises, visualize, noise_seed, cost_scaling_factor, inflation_radius):
        self.T, self.B = cfg.time_steps, cfg.batch_size
        c = SorthamOptimizerConfig()
        c.base = cfg
        c.controller_frequency = controller_frequency
        c.retry_attempt_limit = retry_attempt_limit
        c.regenerate_noises = int(regenerate_noises)
        c.visualize = int(visualize)
        c.noise_seed = noise_seed
        names = DEFAULT_CRITICS if critics is None else critics
        for i, n in enumerate(names):
            c.critics[i] = n.encode()
        c.n_critics = len(names)
        c.cost_scaling_factor, c.inflation_radius = cost_scaling_factor, inflation_radius
        c.motion_model = motion_model.encode()
        self._keep = c       # (the name strings stay alive with the struct)
        return c

    def close(self):
        if getattr(self, "h", None):
            self.lib.sortham_optimizer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.sortham_optimizer_last_error(self.h).decode())

    def set_costmap(self, cells, origin_x, origin_y, resolution, track_unknown=False,
                    inscribed_radius=0.1, has_inflation_layer=True):
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        h, w = cells.shape
        self._ck(self.lib.sortham_optimizer_set_costmap(
            self.h, _ptr(cells), w, h, origin_x, origin_y, resolution, int(track_unknown),
            inscribed_radius, int(has_inflation_layer)))

    def set_noise(self, nvx, nvy, nwz):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (nvx, nvy, nwz)]
        self._ck(self.lib.sortham_optimizer_set_noise(self.h, _ptr(a[0]), _ptr(a[1]), _ptr(a[2])))

    def eval_control(self, tick):
        """-> (twist [vx, vy, wz] float64, SmpcTickOut); raises where the reference throws."""
        tw = np.zeros(3, np.float64)
        out = A.SmpcTickOut()
        self._ck(self.lib.sortham_optimizer_eval_control(self.h, C.byref(tick.c), _ptr(tw),
                                                         C.byref(out)))
        return tw, out

    def set_speed_limit(self, speed_limit, percentage):
        self._ck(self.lib.sortham_optimizer_set_speed_limit(self.h, speed_limit, int(percentage)))

    def reset(self):
        self._ck(self.lib.sortham_optimizer_reset(self.h))

    def get_control_sequence(self):
        u = np.zeros((3, self.T), np.float32)
        self._ck(self.lib.sortham_optimizer_get_control_sequence(self.h, _ptr(u)))
        return u

    def set_control_sequence(self, u):
        u = np.ascontiguousarray(u, np.float32)
        self._ck(self.lib.sortham_optimizer_set_control_sequence(self.h, _ptr(u)))

    def get_constraints(self):
        c = np.zeros(4, np.float32)
        s = C.c_int32(0)
        self._ck(self.lib.sortham_optimizer_get_constraints(self.h, _ptr(c), C.byref(s)))
        return c, bool(s.value)

    def get_optimized_trajectory(self):
        t = np.zeros((self.T, 3), np.float32)
        self._ck(self.lib.sortham_optimizer_get_optimized_trajectory(self.h, _ptr(t)))
        return t
