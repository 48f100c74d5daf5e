"""ctypes mirror of include/smpc.h (structs, constants, prototypes).

Only plain C types: this module knows nothing about HIP or torch.  The same
struct classes are handed to the product library (libsmpc.so) and, in tests,
to the CPU oracle, so one set of inputs feeds both.
"""
import ctypes as C

SMPC_ABI_VERSION = 2

SMPC_OK = 0
SMPC_ERR_INVALID = -1
SMPC_ERR_UNSUPPORTED = -2
SMPC_ERR_DEVICE = -3
SMPC_ERR_STATE = -4
SMPC_ERR_NOMEM = -5

SMPC_COST_NO_INFORMATION = 255
SMPC_COST_LETHAL = 254
SMPC_COST_INSCRIBED = 253
SMPC_COST_FREE = 0

SMPC_MODEL_OMNI = 0
SMPC_MODEL_DIFF_DRIVE = 1
SMPC_MODEL_ACKERMANN = 2

SMPC_FLAG_STORE_TRAJECTORIES = 0x1
SMPC_FLAG_NO_SPECULATION = 0x2
SMPC_FLAG_PROFILE = 0x4
SMPC_FLAG_WAVE_PER_ROLLOUT = 0x8
SMPC_FLAG_LANE_PER_ROLLOUT = 0x10

SMPC_TUPLE_HEADER = 4


class SmpcConfig(C.Structure):
    """smpc_config — reference models/optimizer_settings.hpp:28-41."""
    _fields_ = [
        ("batch_size", C.c_uint32),
        ("time_steps", C.c_uint32),
        ("iteration_count", C.c_uint32),
        ("motion_model", C.c_uint32),
        ("model_dt", C.c_float),
        ("temperature", C.c_float),
        ("gamma", C.c_float),
        ("vx_max", C.c_float),
        ("vx_min", C.c_float),
        ("vy_max", C.c_float),
        ("wz_max", C.c_float),
        ("vx_std", C.c_float),
        ("vy_std", C.c_float),
        ("wz_std", C.c_float),
        ("device", C.c_int32),
        ("flags", C.c_uint32),
        ("shard_offset", C.c_uint64),
        ("global_batch_size", C.c_uint64),
        ("ackermann_min_turning_r", C.c_float),
        ("reserved0", C.c_uint32),
    ]


class SmpcObstaclesParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("consider_footprint", C.c_int32),
        ("cost_power", C.c_uint32),
        ("repulsion_weight", C.c_float),
        ("critical_weight", C.c_float),
        ("collision_cost", C.c_float),
        ("collision_margin_distance", C.c_float),
        ("near_goal_distance", C.c_float),
    ]


class SmpcPathAlignParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("use_path_orientations", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
        ("max_path_occupancy_ratio", C.c_float),
        ("offset_from_furthest", C.c_uint32),
        ("trajectory_point_step", C.c_uint32),
        ("threshold_to_consider", C.c_float),
    ]


class SmpcPathFollowParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
        ("threshold_to_consider", C.c_float),
        ("offset_from_furthest", C.c_uint32),
    ]


class SmpcGoalAngleParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
        ("threshold_to_consider", C.c_float),
    ]


class SmpcPreferForwardParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
        ("threshold_to_consider", C.c_float),
    ]


class SmpcCostParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("consider_footprint", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
        ("critical_cost", C.c_float),
        ("collision_cost", C.c_float),
        ("near_goal_distance", C.c_float),
    ]


class SmpcGoalParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
        ("threshold_to_consider", C.c_float),
    ]


class SmpcConstraintParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
        ("vx_max", C.c_float),
        ("vy_max", C.c_float),
        ("vx_min", C.c_float),
    ]


class SmpcTwirlingParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
    ]


class SmpcPathAngleParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
        ("offset_from_furthest", C.c_uint32),
        ("threshold_to_consider", C.c_float),
        ("max_angle_to_furthest", C.c_float),
        ("forward_preference", C.c_int32),
        ("vx_min", C.c_float),
    ]


class SmpcVelocityDeadbandParams(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32),
        ("cost_power", C.c_uint32),
        ("cost_weight", C.c_float),
        ("deadband_velocities", C.c_float * 3),
    ]


class SmpcCriticParams(C.Structure):
    _fields_ = [
        ("obstacles", SmpcObstaclesParams),
        ("path_align", SmpcPathAlignParams),
        ("path_follow", SmpcPathFollowParams),
        ("goal_angle", SmpcGoalAngleParams),
        ("prefer_forward", SmpcPreferForwardParams),
        ("cost", SmpcCostParams),
        ("goal", SmpcGoalParams),
        ("constraint", SmpcConstraintParams),
        ("twirling", SmpcTwirlingParams),
        ("path_angle", SmpcPathAngleParams),
        ("velocity_deadband", SmpcVelocityDeadbandParams),
        ("path_align_legacy", SmpcPathAlignParams),
    ]


_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


class SmpcTickIn(C.Structure):
    """smpc_tick_in — what Optimizer::prepare() copies (optimizer.cpp:185-204)."""
    _fields_ = [
        ("pose_x", C.c_double),
        ("pose_y", C.c_double),
        ("pose_yaw", C.c_float),
        ("speed_vx", C.c_double),
        ("speed_vy", C.c_double),
        ("speed_wz", C.c_double),
        ("path_x", _f32p),
        ("path_y", _f32p),
        ("path_yaw", _f32p),
        ("path_len", C.c_uint32),
        ("goal_x", C.c_double),
        ("goal_y", C.c_double),
        ("path_pts_valid", _u8p),
        ("fail_flag_in", C.c_int32),
        ("goal_checker_xy_tolerance", C.c_float),
    ]


class SmpcTickOut(C.Structure):
    _fields_ = [
        ("fail_flag", C.c_int32),
        ("furthest_valid", C.c_int32),
        ("furthest_reached_path_point", C.c_uint32),
        ("non_colliding", C.c_uint32),
        ("min_cost", C.c_float),
        ("sum_w", C.c_float),
        ("passes", C.c_uint32),
        ("device_ms", C.c_float),
        ("score_pass_ms", C.c_float),
        ("pass_kind", C.c_uint32),
    ]


# name -> (restype, argtypes) for every symbol include/smpc.h declares.
_ctx = C.c_void_p
PROTOTYPES = {
    "smpc_config_default": (None, [C.POINTER(SmpcConfig)]),
    "smpc_critic_params_default": (None, [C.POINTER(SmpcCriticParams)]),
    "smpc_create": (C.c_int, [C.POINTER(SmpcConfig), C.POINTER(_ctx)]),
    "smpc_destroy": (None, [_ctx]),
    "smpc_last_error": (C.c_char_p, [_ctx]),
    "smpc_abi_version": (C.c_int, []),
    "smpc_build_info": (C.c_char_p, []),
    "smpc_reset": (C.c_int, [_ctx]),
    "smpc_set_constraints": (C.c_int, [_ctx, C.c_float, C.c_float, C.c_float, C.c_float]),
    "smpc_set_critics": (C.c_int, [_ctx, C.POINTER(SmpcCriticParams)]),
    "smpc_set_costmap": (C.c_int, [_ctx, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double,
                                   C.c_double, C.c_double, C.c_int, C.c_float, C.c_float,
                                   C.c_float]),
    "smpc_update_costmap_region": (C.c_int, [_ctx, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_uint32]),
    "smpc_costmap_upload_bytes": (C.c_int, [_ctx, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "smpc_set_footprint": (C.c_int, [_ctx, C.c_void_p, C.c_uint32, C.c_double, C.c_double]),
    "smpc_set_noise": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smpc_seed": (C.c_int, [_ctx, C.c_uint64]),
    "smpc_redraw_noise": (C.c_int, [_ctx]),
    "smpc_redraw_noise_async": (C.c_int, [_ctx]),
    "smpc_get_noise": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smpc_optimize": (C.c_int, [_ctx, C.POINTER(SmpcTickIn), C.c_void_p,
                                C.POINTER(SmpcTickOut)]),
    "smpc_get_trajectories": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smpc_get_costs": (C.c_int, [_ctx, C.c_void_p]),
    "smpc_selftest_sincos": (C.c_int, [_ctx, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "smpc_selftest_lane_reduce": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smpc_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "smpc_set_profile": (C.c_int, [_ctx, C.c_int]),
    "smpc_tuple_len": (C.c_uint32, [_ctx]),
    "smpc_shard_begin": (C.c_int, [_ctx, C.POINTER(SmpcTickIn), C.c_void_p]),
    "smpc_shard_furthest": (C.c_int, [_ctx, C.c_void_p]),
    "smpc_shard_predicted_furthest": (C.c_int, [_ctx, C.POINTER(C.c_uint32)]),
    "smpc_shard_score": (C.c_int, [_ctx, C.c_void_p, C.c_uint32, C.c_void_p]),
    "smpc_shard_rescore_failed": (C.c_int, [_ctx, C.c_void_p]),
    "smpc_shard_combine": (C.c_int, [_ctx, C.c_void_p, C.c_uint32, C.c_void_p,
                                     C.POINTER(SmpcTickOut)]),
    "smpc_group_create": (C.c_int, [C.POINTER(_ctx), C.c_uint32, C.POINTER(_ctx)]),
    "smpc_group_destroy": (None, [_ctx]),
    "smpc_group_optimize": (C.c_int, [_ctx, C.POINTER(SmpcTickIn), C.POINTER(C.c_void_p),
                                      C.POINTER(SmpcTickOut)]),
    "smpc_shard_comm_id": (C.c_int, [C.c_void_p, C.c_uint32]),
    "smpc_shard_comm_init": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "smpc_shard_p2p_handle": (C.c_int, [_ctx, C.c_void_p, C.c_uint32]),
    "smpc_shard_p2p_init": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "smpc_shard_p2p_set_timeout": (C.c_int, [_ctx, C.c_uint32]),
    "smpc_shard_tick": (C.c_int, [_ctx, C.POINTER(SmpcTickIn), C.c_void_p, C.POINTER(SmpcTickOut),
                                  C.c_int]),
}
SMPC_COMM_ID_BYTES = 128


def bind(lib, prototypes=None):
    """Attach restype/argtypes to every prototype; raises AttributeError on a
    missing symbol (the "exports every symbol smpc.h declares" check)."""
    for name, (res, args) in (prototypes or PROTOTYPES).items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
