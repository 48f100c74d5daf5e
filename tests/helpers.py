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


def assert_parity(u_gpu, out_gpu, u_ref, out_ref, c_gpu=None, c_ref=None, max_flips=0,
                  rtol=TWIST_RTOL, label="", max_soft=None):
    assert out_gpu.fail_flag == out_ref.fail_flag, label
    if out_ref.furthest_valid:
        assert out_gpu.furthest_valid, label
        assert out_gpu.furthest_reached_path_point == out_ref.furthest_reached_path_point, label
    if c_gpu is not None:
        # A last-ulp difference in a rollout position can put one lookup in the
        # neighbouring costmap cell (SURVEY.md §7 "discontinuous costs"): "hard" flips
        # change a collision, "soft" ones a few percent of one rollout's cost.
        d = np.abs(c_gpu.astype(np.float64) - c_ref.astype(np.float64))
        tight = d <= 2e-4 * np.maximum(np.abs(c_ref), 1.0)
        hard = int(np.sum(d > 100.0))
        soft = int(np.sum(~tight)) - hard
        if max_soft is None:
            max_soft = max(1, int(2e-5 * c_ref.size * u_ref.shape[1]))
        assert hard <= max_flips, f"{label}: {hard} collision flips"
        assert soft <= max_soft, f"{label}: {soft} soft cell flips (max {max_soft})"
        assert float(d[d <= 100.0].max()) < 0.5, f"{label}: cost mismatch {float(d.max())}"
    e_t = rel_err(twist(u_gpu), twist(u_ref))
    e_u = rel_err(u_gpu, u_ref)
    assert e_t <= rtol, f"{label}: twist rel err {e_t:.3e}"
    # the whole sequence is T values, each exposed to the same cell-flip noise as the
    # Twist entry: its maximum gets a wider band
    assert e_u <= 5 * rtol, f"{label}: control sequence rel err {e_u:.3e}"
    return e_t, e_u
