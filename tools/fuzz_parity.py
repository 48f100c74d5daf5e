#!/usr/bin/env python3
"""Developer tool: randomised parity sweep, GPU library against the oracle.

Every case draws a configuration (batch, horizon, motion model, iteration count, pass flags,
temperature, dt, stds), a critic list with perturbed parameters, a scenario (map size and seed, a
curved or straight plan of random length and spacing, pose off the plan with a random yaw, random
speed, a warm-started control sequence) and runs three closed-loop ticks on both sides with the
same stored noise.  The bar is tests/helpers.assert_parity (1e-4 on the Twist, integer outputs
exact, flips counted).  Failing case numbers are printed with the draw, so that one can be re-run
alone:   tools/fuzz_parity.py FIRST COUNT [only=CASE]
"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mpcholonavigation_amd import _abi as A
from mpcholonavigation_amd.optimizer import Smpc
from mpcholonavigation_amd.synthetic import make_noise, make_scenario
from mpcholonavigation_amd.tick import Tick, default_config, default_critics
from oracle.loader import Oracle
from tests.helpers import assert_parity, configure

default_config = default_config   # (re-exported for the other fuzz tools)

CRITICS = ("obstacles", "path_align", "path_follow", "goal_angle", "prefer_forward", "cost", "goal",
           "constraint", "twirling", "path_angle", "velocity_deadband", "path_align_legacy")


def draw(case):
    r = np.random.default_rng(1000003 * case + 17)
    d = {}
    d["T"] = int(r.choice([20, 24, 30, 36, 40, 48, 56, 56, 60, 64, 64, 64, 72, 100, 128, 140]))
    d["B"] = int(r.choice([1, 63, 64, 65, 400, 1000, 2000, 2049, 4096, 7000, 12288, 16384, 20000, 24000]))
    if d["B"] * d["T"] > 1.6e6:
        d["B"] = int(1.6e6 // d["T"])
    d["model"] = int(r.choice([A.SMPC_MODEL_OMNI] * 4 + [A.SMPC_MODEL_DIFF_DRIVE, A.SMPC_MODEL_ACKERMANN]))
    d["iters"] = int(r.choice([1, 1, 1, 2]))
    d["flags"] = int(r.choice([0, 0, 0, A.SMPC_FLAG_LANE_PER_ROLLOUT, A.SMPC_FLAG_WAVE_PER_ROLLOUT,
                               A.SMPC_FLAG_NO_SPECULATION]))
    d["env_pass"] = str(r.choice(["", "", "", "split", "lane"]))
    d["dt"] = float(r.choice([0.05, 0.05, 0.1, 0.033]))
    d["temperature"] = float(r.choice([0.3, 0.3, 0.1, 1.0]))
    d["gamma"] = float(r.choice([0.015, 0.015, 0.0, 0.1]))
    d["map"] = int(r.choice([100, 200, 200, 200, 333, 500, 2000]))
    d["map_seed"] = int(r.integers(1, 1 << 30))
    d["noise_seed"] = int(r.integers(1, 1 << 30))
    kind = str(r.choice(["default5", "default5", "deployed9", "random", "random"]))
    d["critic_kind"] = kind
    if kind == "default5":
        on = CRITICS[:5]
    elif kind == "deployed9":
        on = ("constraint", "cost", "goal", "goal_angle", "path_align", "path_follow", "path_angle", "prefer_forward",
              "twirling")
    else:
        on = tuple(c for c in CRITICS if r.random() < 0.5)
    d["critics"] = on
    d["perturb"] = bool(r.random() < 0.5)
    d["power2"] = tuple(c for c in on if d["perturb"] and r.random() < 0.15)
    d["curve"] = float(r.choice([0.0, 0.0, 0.3, -0.5, 1.0]))
    d["P"] = int(r.choice([2, 5, 9, 20, 40, 60, 60, 90, 120]))
    d["spacing"] = float(r.choice([0.05, 0.05, 0.1, 0.025]))
    d["pose_dy"] = float(r.choice([0.0, 0.0, 0.05, -0.2]))
    d["pose_yaw"] = float(r.choice([0.0, 0.0, 0.4, -1.0, 3.0]))
    d["speed"] = (float(r.uniform(-0.2, 0.5)), float(r.uniform(-0.2, 0.2)), float(r.uniform(-1.0, 1.0)))
    d["warm"] = (float(r.uniform(-0.1, 0.5)), float(r.uniform(-0.2, 0.2)), float(r.uniform(-0.8, 0.8)))
    d["xy_tol"] = float(r.choice([-1.0, -1.0, 0.25]))
    d["weights"] = [float(w) for w in r.uniform(0.5, 2.0, size=len(CRITICS))]
    # (drawn last, so that the cases of the earlier sweeps keep their numbers)
    d["footprint"] = str(r.choice(["", "", "", "obstacles", "cost", "both"]))
    d["unknown"] = float(r.choice([0.0, 0.0, 0.0, 0.002, 0.02]))      # share of NO_INFORMATION cells
    d["track_unknown"] = bool(r.random() < 0.5)
    d["rng"] = bool(r.random() < 0.2)                                  # device RNG (smpc_seed) against its CPU twin
    d["edge"] = bool(r.random() < 0.1)                                 # the robot near the map's edge: rollouts leave the map
    d["limit"] = float(r.choice([0.0, 0.0, 0.0, 0.5, 0.8]))            # constraints scaled before tick 1 (setSpeedLimit)
    d["map_edit"] = bool(r.random() < 0.2)                             # a block of the costmap changes before tick 2
    d["lethal_blob"] = bool(r.random() < 0.1)                          # lethal cells right in front of the robot
    d["res"] = float(r.choice([0.05, 0.05, 0.05, 0.1, 0.025]))         # costmap resolution (the scene scales with it)
    d["origin"] = (float(r.choice([0.0, 0.0, -3.7, 12.25])), float(r.choice([0.0, 0.0, 5.5, -1.05])))
    d["crop"] = (int(r.choice([0, 0, 1, 3, 17])), int(r.choice([0, 0, 2, 5])))   # columns / rows cut off the map
    d["pa_step"] = int(r.choice([4, 4, 4, 2, 3, 5, 8]))               # PathAlign trajectory_point_step
    d["occupied_path"] = float(r.choice([0.0, 0.0, 0.0, 0.05, 0.3]))   # share of the plan's points put on lethal cells
    d["big"] = bool(r.random() < 0.004)                                # a batch the lane pass takes by itself
    d["regen"] = str(r.choice(["", "sync", "async", "async"]))         # (with rng) regenerate_noises: a new epoch behind every tick
    return d


def build(d):
    if d.get("big"):
        d["B"] = 70001 if d["T"] <= 64 else 66000
    cfg = default_config(batch_size=d["B"], time_steps=d["T"], iteration_count=d["iters"], motion_model=d["model"],
                         model_dt=d["dt"], temperature=d["temperature"], gamma=d["gamma"], flags=d["flags"])
    if d["model"] != A.SMPC_MODEL_OMNI:
        cfg.vy_max = 0.0
        cfg.vy_std = 0.0
    scn = make_scenario(d["T"], map_size=d["map"], seed=d["map_seed"], speed=d["speed"])
    t = scn.tick
    # the plan: P points from the robot's x, spacing `spacing`, bending with curvature `curve`
    s = d["spacing"] * np.arange(d["P"])
    if d["curve"] == 0.0:
        px, py, pyaw = t.pose_x + s, np.full_like(s, t.pose_y), np.zeros_like(s)
    else:
        k = d["curve"]
        px, py, pyaw = t.pose_x + np.sin(k * s) / k, t.pose_y + (1.0 - np.cos(k * s)) / k, k * s
    W = d["map"] * scn.resolution
    keep = (px > 0.1) & (px < W - 0.1) & (py > 0.1) & (py < W - 0.1)
    n = max(2, int(np.argmin(keep)) if not keep.all() else len(s))
    px, py, pyaw = px[:n], py[:n], pyaw[:n]
    tick = Tick(pose_x=t.pose_x, pose_y=t.pose_y + d["pose_dy"], pose_yaw=d["pose_yaw"], speed=d["speed"],
                path_x=px.astype(np.float32), path_y=py.astype(np.float32), path_yaw=pyaw.astype(np.float32),
                goal_x=float(np.float32(px[-1])), goal_y=float(np.float32(py[-1])),
                goal_checker_xy_tolerance=d["xy_tol"])
    u0 = np.zeros((3, d["T"]), np.float32)
    u0[0], u0[1], u0[2] = d["warm"]
    if d["model"] != A.SMPC_MODEL_OMNI:
        u0[1] = 0.0
    cr = default_critics()
    for i, name in enumerate(CRITICS):
        c = getattr(cr, name)
        c.enabled = 1 if name in d["critics"] else 0
        if d["perturb"]:
            if hasattr(c, "cost_weight"):
                c.cost_weight = float(np.float32(c.cost_weight * d["weights"][i]))
            if name in d["power2"]:
                c.cost_power = 2
    if d["perturb"]:
        cr.obstacles.repulsion_weight = float(np.float32(1.5 * d["weights"][0]))
        cr.path_align.offset_from_furthest = int(5 + 20 * d["weights"][1] / 2)
        cr.path_follow.offset_from_furthest = int(2 + 6 * d["weights"][2] / 2)
        cr.path_align.use_path_orientations = int(d["weights"][3] > 1.5)
    noise = make_noise(d["B"], d["T"], std=(cfg.vx_std, cfg.vy_std, cfg.wz_std), seed=d["noise_seed"])
    r = np.random.default_rng(d["map_seed"])
    if d["unknown"] > 0.0:
        scn.cells = scn.cells.copy()
        scn.cells[r.random(scn.cells.shape) < d["unknown"]] = 255
    if d["lethal_blob"]:
        scn.cells = scn.cells.copy()
        cy, cx = int(tick.pose_y / scn.resolution), int(tick.pose_x / scn.resolution) + 12
        scn.cells[max(cy - 3, 0):cy + 4, cx:cx + 5] = 254
    if d["edge"]:
        tick.pose_x = float(W - 0.3)
        tick.pose_y = float(min(max(tick.pose_y, 0.3), W - 0.3))
    if d["occupied_path"] > 0.0:
        scn.cells = scn.cells.copy()
        for i in np.nonzero(r.random(len(tick.path_x)) < d["occupied_path"])[0]:
            if i > 6:      # (not under the robot)
                scn.cells[min(int(tick.path_y[i] / scn.resolution), d["map"] - 1), min(int(tick.path_x[i] / scn.resolution), d["map"] - 1)] = 254
    cr.path_align.trajectory_point_step = d["pa_step"]
    cr.path_align_legacy.trajectory_point_step = d["pa_step"]
    # columns / rows cut off (odd widths: the LDS staging's byte path), then the scene scaled to the
    # resolution and moved to the origin
    cw, ch = d["crop"]
    if cw or ch:
        scn.cells = np.ascontiguousarray(scn.cells[:scn.cells.shape[0] - ch, :scn.cells.shape[1] - cw])
    k = d["res"] / scn.resolution
    ox, oy = d["origin"]
    if k != 1.0 or ox != 0.0 or oy != 0.0:
        scn.resolution = d["res"]
        scn.origin_x, scn.origin_y = ox, oy
        tick = Tick(pose_x=ox + k * tick.pose_x, pose_y=oy + k * tick.pose_y, pose_yaw=tick.pose_yaw, speed=tick.speed,
                    path_x=(ox + k * tick.path_x.astype(np.float64)).astype(np.float32),
                    path_y=(oy + k * tick.path_y.astype(np.float64)).astype(np.float32), path_yaw=tick.path_yaw,
                    goal_x=float(np.float32(ox + k * tick.goal_x)), goal_y=float(np.float32(oy + k * tick.goal_y)),
                    goal_checker_xy_tolerance=tick.goal_checker_xy_tolerance)
    if d["footprint"]:
        if d["footprint"] in ("obstacles", "both") and "obstacles" in d["critics"]:
            cr.obstacles.consider_footprint = 1
        if d["footprint"] in ("cost", "both") and "cost" in d["critics"]:
            cr.cost.consider_footprint = 1
    return cfg, scn, tick, u0, cr, noise


DETAIL = False


def kernel_name(g):
    import ctypes
    f = g.lib.smpc_debug_last_pass_kernel
    f.restype = ctypes.c_char_p
    f.argtypes = []
    return f().decode()


class Mismatch(AssertionError):
    pass


def check(case, k, d, ug, og, uo, oo, cg, co):
    """-> notes (what was tolerated); raises Mismatch on what is not.  Integer outputs exact; per-rollout
    costs within 2e-4 relative except for a FEW rollouts (one lookup a last ulp across a cell edge:
    counted, not tolerated in bulk); the Twist within 1e-4 of its largest component when both sides
    scored the same cells and some rollout survived."""
    lab = f"case {case} tick {k}"
    notes = []
    if og.fail_flag != oo.fail_flag:
        raise Mismatch(f"{lab}: fail_flag {og.fail_flag} / {oo.fail_flag}")
    if oo.furthest_valid and (not og.furthest_valid or og.furthest_reached_path_point != oo.furthest_reached_path_point):
        raise Mismatch(f"{lab}: furthest {og.furthest_reached_path_point} / {oo.furthest_reached_path_point}")
    cg, co = cg.astype(np.float64), co.astype(np.float64)
    dd = np.abs(cg - co)
    rel = dd / np.maximum(np.abs(co), 1.0)
    hard = int(np.sum((dd > 100.0) & (rel > 2e-4)))     # (costs of 1e6 — a power-2 critic far from the plan — differ by 100 in the last ulps)
    soft = int(np.sum(rel > 2e-4)) - hard
    few = max(3, int(4e-5 * co.size * ug.shape[1]))
    if d["iters"] > 1:
        # the costs are the LAST iteration's: its control sequence already carries the first
        # iteration's differences (one flipped rollout of a handful that carry the weight moves u by
        # 1e-3), and every rollout is then scored from a slightly different place
        few = max(few * 20, co.size)
    if hard > (3 if d["iters"] == 1 else max(3, co.size // 1000)):
        raise Mismatch(f"{lab}: {hard} collision flips of {co.size}")
    if soft > few:
        i = int(np.argmax(np.where(dd > 100.0, 0.0, dd)))
        raise Mismatch(f"{lab}: SYSTEMATIC: {soft} of {co.size} rollouts' costs differ by more than 2e-4 relative "
                       f"(largest {cg[i]:.6g} vs {co[i]:.6g})")
    # (a collision verdict that flips moves the cost by collision_cost x weight / T — below the 100
    # that marks a "hard" flip when the weight is small or the horizon long: count the soft ones too)
    if abs(int(og.non_colliding) - int(oo.non_colliding)) > hard + soft:
        raise Mismatch(f"{lab}: non_colliding {og.non_colliding} / {oo.non_colliding} with {hard} + {soft} flips")
    if hard or soft:
        notes.append(f"tick {k}: flips hard {hard} soft {soft}")
    tg, tr = ug[:, min(1, ug.shape[1] - 1)].astype(np.float64), uo[:, min(1, uo.shape[1] - 1)].astype(np.float64)
    err = float(np.max(np.abs(tg - tr)))
    e = err / max(float(np.max(np.abs(tr))), 1e-3)
    if oo.non_colliding == 0:
        return notes        # every rollout collided: the weights hang on the last ulp of 2e5 (and the tick is discarded)
    # float32 costs carry ~8 ulps of summation noise whatever the order; the softmax turns an ulp of
    # the minimum cost into ulp / temperature on a weight
    cond = 32.0 * float(np.spacing(np.float32(abs(oo.min_cost) + 1.0))) / d["temperature"]
    # (... and the Twist is a weighted mean of NOISE: its error scales with the sampling std, 0.2 - 0.4,
    # not with its own size)
    if err > (1e-4 + cond) * float(np.max(np.abs(tr))) + 2e-6 + 0.1 * cond:
        if d["temperature"] < 0.3 and e <= 1e-3 and not (hard or soft):
            # costs agree to a few 1e-6 relative (float sums in another order); at temperature 0.1 with
            # two or three rollouts carrying the weight that is 1e-4 on the Twist
            notes.append(f"tick {k}: sharp softmax (temperature {d['temperature']}, sum_w {oo.sum_w:.3g}): twist {e:.1e}")
            return notes
        if d["iters"] > 1 and e <= 1e-2:
            notes.append(f"tick {k}: second iteration from a control sequence that differs: twist {e:.1e}")
            return notes
        # how sharp the softmax is: weights move by exp(dc / temperature)
        mid = int(np.sum((rel > 2e-5) & (rel <= 2e-4)))
        msg = (f"{lab}: twist {e:.2e} of its largest component (flips hard {hard} soft {soft}, {mid} more rollouts between "
               f"2e-5 and 2e-4; sum_w {oo.sum_w:.3g}, temperature {d['temperature']})")
        if hard or soft or mid:
            notes.append("TWIST-BY-FLIPS " + msg)
        else:
            raise Mismatch(msg)
    return notes


def run(case):
    d = draw(case)
    cfg, scn, tick, u0, cr, noise = build(d)
    os.environ.pop("SMPC_PASS", None)
    if d["env_pass"]:
        os.environ["SMPC_PASS"] = d["env_pass"]
    try:
        g = Smpc(cfg)
    finally:
        os.environ.pop("SMPC_PASS", None)
    o = Oracle(cfg)
    kinds, notes = [], []
    try:
        fp = np.array([[0.25, 0.15], [0.25, -0.15], [-0.2, -0.15], [-0.2, 0.15]])
        cells = scn.cells
        for obj in (g, o):
            if cr.obstacles.consider_footprint or cr.cost.consider_footprint:
                obj.set_footprint(fp, 0.3)
            configure(obj, scn, critics=cr, noise=None if d["rng"] else noise, track_unknown=d["track_unknown"])
            if d["rng"]:
                obj.seed(d["noise_seed"])
        ug = uo = u0
        for k in range(3):
            if k == 1 and d["limit"] > 0.0:
                f = d["limit"]
                for obj in (g, o):
                    obj.set_constraints(cfg.vx_max * f, cfg.vx_min * f, cfg.vy_max * f, cfg.wz_max * f)
            if k == 2 and d["map_edit"]:
                cells = cells.copy()
                cy, cx = int((tick.pose_y - scn.origin_y) / scn.resolution), int((tick.pose_x - scn.origin_x) / scn.resolution)
                cy, cx = min(max(cy, 0), cells.shape[0] - 1), min(max(cx, 0), cells.shape[1] - 1)
                cells[max(cy - 20, 0):cy + 20, cx + 5:cx + 30] = np.roll(cells[max(cy - 20, 0):cy + 20, cx + 5:cx + 30], 3, axis=0)
                for obj in (g, o):
                    obj.set_costmap(cells, scn.origin_x, scn.origin_y, scn.resolution, track_unknown=d["track_unknown"],
                                    inscribed_radius=scn.inscribed_radius, cost_scaling_factor=scn.cost_scaling_factor,
                                    inflation_radius=scn.inflation_radius)
            tk = Tick(tick.pose_x + 0.02 * k, tick.pose_y, tick.pose_yaw + 0.01 * k, tick.speed, tick.path_x, tick.path_y,
                      tick.path_yaw, tick.goal_x, tick.goal_y, goal_checker_xy_tolerance=tick.goal_checker_xy_tolerance)
            ug, og = g.optimize(tk, uo)
            if d["rng"] and d["regen"] and k > 0:
                o.set_noise(*g.get_noise())      # the epoch this tick was scored with (tick 0: the oracle's own twin of the stream)
            uo, oo = o.optimize(tk, uo)
            if d["rng"] and d["regen"] == "sync":
                g.redraw_noise()
            elif d["rng"] and d["regen"] == "async":
                g.redraw_noise_async()
            kinds.append(og.pass_kind)
            if DETAIL:
                cg, co = g.get_costs().astype(np.float64), o.get_costs().astype(np.float64)
                dd = np.abs(cg - co)
                i = int(np.argmax(dd))
                rel = dd / np.maximum(np.abs(co), 1.0)
                j = int(np.argmax(rel))
                print(f"  tick {k}: kind {og.pass_kind} passes {og.passes} kernel {kernel_name(g)}\n"
                      f"    twist gpu {ug[:, 1]} ref {uo[:, 1]}\n"
                      f"    min_cost {og.min_cost} / {oo.min_cost}  sum_w {og.sum_w} / {oo.sum_w}  furthest "
                      f"{og.furthest_reached_path_point} / {oo.furthest_reached_path_point}  non_colliding {og.non_colliding} / "
                      f"{oo.non_colliding}\n"
                      f"    costs: max |d| {dd[i]:.4g} at {i} ({cg[i]:.6g} vs {co[i]:.6g}); max rel {rel[j]:.3g} at {j} "
                      f"({cg[j]:.6g} vs {co[j]:.6g}); |d| > 2e-4 rel: {int(np.sum(rel > 2e-4))}; cost range "
                      f"{co.min():.4g} .. {co.max():.4g}", flush=True)
            notes += check(case, k, d, ug, og, uo, oo, g.get_costs(), o.get_costs())
            uo = np.concatenate([uo[:, 1:], uo[:, -1:]], axis=1)
    finally:
        g.close()
        o.close()
    return d, kinds, notes


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    only = [int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("only=")]
    cases = only or range(first, first + count)
    global DETAIL
    DETAIL = bool(only)
    bad = 0
    t0 = time.time()
    for case in cases:
        try:
            d, kinds, notes = run(case)
            print(f"case {case}: ok{' (' + '; '.join(notes) + ')' if notes else ''}  B {d['B']} T {d['T']} model {d['model']} it {d['iters']} flags {d['flags']:#x} "
                  f"pass {d['env_pass'] or '-'} kinds {kinds} critics {d['critic_kind']} map {d['map']} "
                  f"{'fp:' + d['footprint'] + ' ' if d['footprint'] else ''}{'unk ' if d['unknown'] else ''}{'rng' + (':' + d['regen'] if d['regen'] else '') + ' ' if d['rng'] else ''}"
                  f"{'res %g ' % d['res'] if d['res'] != 0.05 else ''}{'big ' if d['big'] else ''}{'edge ' if d['edge'] else ''}{'limit ' if d['limit'] else ''}{'edit ' if d['map_edit'] else ''}{'blob ' if d['lethal_blob'] else ''}"
                  f"({time.time() - t0:.0f} s)", flush=True)
        except Exception as e:
            msg = str(e).splitlines()[0] if str(e) else type(e).__name__
            if "more than 63 samples per trajectory" in msg:      # a documented refusal (SMPC_ERR_UNSUPPORTED)
                continue
            bad += 1
            print(f"case {case}: FAILED  {type(e).__name__}: {msg[:300]}\n    draw: {draw(case)}", flush=True)
            if only:
                traceback.print_exc()
    print(f"{bad} of {len(list(cases))} cases failed", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
