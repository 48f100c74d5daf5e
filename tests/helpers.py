"""Shared builders for the parity tests: one set of inputs, two libraries."""
import numpy as np

from mpcholonavigation_amd import _abi as A
from mpcholonavigation_amd.synthetic import make_noise, make_scenario
from mpcholonavigation_amd.tick import default_config, default_critics

TWIST_RTOL = 1e-4   # north star: emitted Twist within 1e-4 relative of the CPU path


def configure(obj, scn, critics=None, noise=None, track_unknown=False):
    """Apply the same costmap / critics / noise to a Smpc or an Oracle."""
    obj.set_critics(critics if critics is not None else default_critics())
    obj.set_costmap(scn.cells, scn.origin_x, scn.origin_y, scn.resolution,
                    track_unknown=track_unknown, inscribed_radius=scn.inscribed_radius,
                    cost_scaling_factor=scn.cost_scaling_factor,
                    inflation_radius=scn.inflation_radius)
    if noise is not None:
        obj.set_noise(*noise)


def make_case(B, T, map_size=200, seed=42, noise_seed=1234, **scn_kw):
    cfg = default_config(batch_size=B, time_steps=T)
    scn = make_scenario(T, map_size=map_size, seed=seed, **scn_kw)
    noise = make_noise(B, T, std=(cfg.vx_std, cfg.vy_std, cfg.wz_std), seed=noise_seed)
    return cfg, scn, noise


def reference_smoke_fixture(model="Omni"):
    """The reference's only Obstacles-bearing fixture, re-encoded as data
    (test/optimizer_smoke_test.cpp:45-116, test/utils/models.hpp:37-60,
    test/utils/utils.hpp:135-170, test/utils/factory.hpp:101-131,235-247): a 40x40 costmap at
    0.1 m, origin (0, 0), blank except an 8x8 block of cost 250 around the centre; the robot at
    the centre (on the block: cost 250 is not a collision), zero velocity, yaw 0; a 50-point
    plan running diagonally from the robot in steps of one cell in x and y, goal = its last
    point; batch 400, horizon 15, iteration_count 1, no goal checker; the critic list of the
    motion model's case, every parameter at its default (setUpOptimizerParams passes none, so
    consider_footprint stays false).  The reference asserts only that evalControl does not
    throw, i.e. that the tick does not end with every rollout colliding.
    Returns (cfg, cells, resolution, tick, u0, critics)."""
    from mpcholonavigation_amd.tick import Tick
    models = {"Omni": (A.SMPC_MODEL_OMNI, ("goal", "goal_angle", "obstacles", "path_align", "twirling",
                                            "path_follow", "prefer_forward")),
              "DiffDrive": (A.SMPC_MODEL_DIFF_DRIVE, ("goal", "goal_angle", "cost", "path_angle", "path_follow",
                                                      "prefer_forward")),
              "Ackermann": (A.SMPC_MODEL_ACKERMANN, ("goal", "goal_angle", "obstacles", "path_angle",
                                                     "path_follow", "prefer_forward"))}
    mm, names = models[model]
    cfg = default_config(batch_size=400, time_steps=15, iteration_count=1, motion_model=mm)
    cells = np.zeros((40, 40), np.uint8)
    cells[16:24, 16:24] = 250          # addObstacle(costmap, {20 - 4, 20 - 4, 8, 250})
    P = 50
    px = (2.0 + 0.1 * np.arange(P)).astype(np.float32)
    tick = Tick(pose_x=2.0, pose_y=2.0, pose_yaw=0.0, speed=(0.0, 0.0, 0.0), path_x=px, path_y=px.copy(),
                path_yaw=np.zeros(P, np.float32), goal_x=float(2.0 + 0.1 * (P - 1)),
                goal_y=float(2.0 + 0.1 * (P - 1)))
    cr = default_critics()
    for n in ("obstacles", "path_align", "path_follow", "goal_angle", "prefer_forward", "cost", "goal",
              "constraint", "twirling", "path_angle", "velocity_deadband"):
        getattr(cr, n).enabled = 1 if n in names else 0
    u0 = np.zeros((3, 15), np.float32)
    return cfg, cells, 0.1, tick, u0, cr


def twist(u, offset=1):
    """Optimizer::getControlFromSequenceAsTwist with shift_control_sequence on
    (reference src/optimizer.cpp:396-410): (vx, vy, wz) at index 1."""
    offset = min(offset, u.shape[1] - 1)
    return np.array([u[0, offset], u[1, offset], u[2, offset]], np.float64)


def rel_err(a, b):
    """max |a - b| / max |b| (vector-relative, the Twist is one vector)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-12))


def cost_flips(c_gpu, c_ref, big=100.0):
    """Number of rollouts whose cost differs by more than `big`: a costmap cell
    (or collision) classified differently because of a last-ulp difference in
    the rollout position (SURVEY.md §7 'discontinuous costs')."""
    return int(np.sum(np.abs(c_gpu.astype(np.float64) - c_ref.astype(np.float64)) > big))


TWIST_ATOL = 1e-6   # absolute floor for a near-zero Twist component: 1e-6 m/s (rad/s) is nil


def twist_component_errors(u_gpu, u_ref):
    """Per component (vx, vy, wz) of the emitted Twist: (|delta|, |delta| / |ref|)."""
    tg, tr = twist(u_gpu), twist(u_ref)
    d = np.abs(tg - tr)
    return d, d / np.maximum(np.abs(tr), 1e-30)


def assert_parity(u_gpu, out_gpu, u_ref, out_ref, c_gpu=None, c_ref=None, max_flips=0,
                  rtol=TWIST_RTOL, label="", max_soft=None, report=True):
    """The north star's bar, per component: every component of the emitted Twist within
    rtol (1e-4) relative of the oracle's, with an absolute floor TWIST_ATOL for components that
    are themselves ~0 (|delta_i| <= rtol |ref_i| + atol) — whenever both sides scored the same
    cells; the integer outputs exact.  With the per-rollout costs also: counted "hard" flips (a
    collision classified differently: one lookup a last ulp across a cell edge) and
    "soft" flips (a neighbouring cell's cost); with a flip counted the bound is rtol of the
    largest Twist component for every component.  Prints what it measured (pytest -s / the
    failure report shows it)."""
    assert out_gpu.fail_flag == out_ref.fail_flag, label
    if out_ref.furthest_valid:
        assert out_gpu.furthest_valid, label
        assert out_gpu.furthest_reached_path_point == out_ref.furthest_reached_path_point, label
    hard = soft = None
    if c_gpu is not None:
        d = np.abs(c_gpu.astype(np.float64) - c_ref.astype(np.float64))
        tight = d <= 2e-4 * np.maximum(np.abs(c_ref), 1.0)
        hard = int(np.sum(d > 100.0))
        soft = int(np.sum(~tight)) - hard
        if max_soft is None:
            max_soft = max(1, int(2e-5 * c_ref.size * u_ref.shape[1]))
        assert hard <= max_flips, f"{label}: {hard} collision flips"
        assert soft <= max_soft, f"{label}: {soft} soft cell flips (max {max_soft})"
        assert float(d[d <= 100.0].max()) < 0.5, f"{label}: cost mismatch {float(d.max())}"
    d_t, r_t = twist_component_errors(u_gpu, u_ref)
    e_t = rel_err(twist(u_gpu), twist(u_ref))
    e_u = rel_err(u_gpu, u_ref)
    if report:
        print(f"[parity] {label}: twist |d| vx {d_t[0]:.2e} vy {d_t[1]:.2e} wz {d_t[2]:.2e}; "
              f"rel vx {r_t[0]:.2e} vy {r_t[1]:.2e} wz {r_t[2]:.2e}; vector-rel {e_t:.2e}; "
              f"sequence {e_u:.2e}; flips hard {hard} soft {soft}")
    tr = np.abs(twist(u_ref))
    if hard or soft:
        # A counted flip means the two sides did NOT score the same discretised input: one lookup
        # of one rollout read the neighbouring cell (its position differs in the last ulp, and
        # neither libm's sinf nor the reference's xsimd/-ffast-math build is "the" rounding).
        # Its weight moves every component by an amount set by the Twist's scale, not by the
        # component's own size, so the bound is rtol of the largest component.
        bound = np.full(3, rtol * float(tr.max()))
    else:
        bound = rtol * tr + TWIST_ATOL
    assert np.all(d_t <= bound), (f"{label}: twist component error {d_t} exceeds {bound} "
                                  f"(rtol {rtol}, flips hard {hard} soft {soft}; relative {r_t})")
    assert e_t <= rtol, f"{label}: twist rel err {e_t:.3e}"
    # the whole sequence is T values, each exposed to the same cell-flip noise as the
    # Twist entry: its maximum gets a wider band
    assert e_u <= 5 * rtol, f"{label}: control sequence rel err {e_u:.3e}"
    return e_t, e_u
