#!/usr/bin/env python3
"""Developer tool: randomised SEQUENCES of calls on one long-lived context (what a controller does
over minutes: ticks with a moving pose and changing plans, a new costmap or a changed region, new
critic parameters, speed limits, reset, new noise, the device RNG and its epochs), mirrored on the
oracle; every tick compared (tools/fuzz_parity.py's bar).  State that outlives a tick — the
furthest-point predictor, the LDS window plan, gates and look-up tables, the tick block's size, the
costmap mirror — is what this is after.   tools/fuzz_state.py FIRST COUNT [only=CASE]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import fuzz_parity as F
from mpcholonavigation_amd import _abi as A
from mpcholonavigation_amd.optimizer import Smpc
from mpcholonavigation_amd.synthetic import make_costmap, make_noise
from mpcholonavigation_amd.tick import Tick
from oracle.loader import Oracle
from tests.helpers import configure

STEPS = 24


def run(case):
    r = np.random.default_rng(17 * case + 9)
    d = F.draw(300000 + case)
    d["iters"], d["footprint"], d["rng"], d["edge"] = 1, "", False, False
    d["res"], d["origin"], d["crop"] = 0.05, (0.0, 0.0), (0, 0)
    d["B"] = min(d["B"], 16384)
    store = bool(r.random() < 0.2)
    d["flags"] = (d["flags"] & ~A.SMPC_FLAG_NO_SPECULATION if r.random() < 0.7 else d["flags"]) | \
                 (A.SMPC_FLAG_STORE_TRAJECTORIES if store else 0)
    cfg, scn, tick, u0, cr, noise = F.build(d)
    if d["env_pass"]:
        os.environ["SMPC_PASS"] = d["env_pass"]
    try:
        g = Smpc(cfg)
    finally:
        os.environ.pop("SMPC_PASS", None)
    o = Oracle(cfg)
    log = []
    try:
        for obj in (g, o):
            configure(obj, scn, critics=cr, noise=noise, track_unknown=d["track_unknown"])
        cells = scn.cells.copy()
        res = scn.resolution
        x, y, yaw = tick.pose_x, tick.pose_y, tick.pose_yaw
        px, py, pyaw = tick.path_x, tick.path_y, tick.path_yaw
        gx, gy = tick.goal_x, tick.goal_y
        u = u0
        rng_mode = False
        ticks = 0
        for step in range(STEPS):
            op = r.random()
            if op < 0.06:                       # a new costmap: same size, or another one
                m = int(r.choice([d["map"], d["map"], 160, 257]))
                cells = make_costmap(m, m, y, x, seed=int(r.integers(1, 1 << 30)), res=res)
                for obj in (g, o):
                    obj.set_costmap(cells, 0.0, 0.0, res, track_unknown=d["track_unknown"], inscribed_radius=scn.inscribed_radius,
                                    cost_scaling_factor=scn.cost_scaling_factor, inflation_radius=scn.inflation_radius)
                log.append(f"{step}:map {m}")
            elif op < 0.14:                     # a region of the costmap changes (the layered costmap's update window)
                H, W = cells.shape
                w, h = int(r.integers(1, min(60, W))), int(r.integers(1, min(60, H)))
                x0, y0 = int(r.integers(0, W - w + 1)), int(r.integers(0, H - h + 1))
                cells = cells.copy()
                cells[y0:y0 + h, x0:x0 + w] = r.choice([0, 0, 60, 200, 253, 254], size=(h, w)).astype(np.uint8)
                g.update_costmap_region(cells, x0, y0, w, h)
                o.set_costmap(cells, 0.0, 0.0, res, track_unknown=d["track_unknown"], inscribed_radius=scn.inscribed_radius,
                              cost_scaling_factor=scn.cost_scaling_factor, inflation_radius=scn.inflation_radius)
                log.append(f"{step}:region {w}x{h}@{x0},{y0}")
            elif op < 0.22:                     # critic parameters
                name = str(r.choice(F.CRITICS))
                c = getattr(cr, name)
                what = r.random()
                if what < 0.4:
                    c.enabled = 0 if c.enabled else 1
                elif hasattr(c, "cost_weight"):
                    c.cost_weight = float(np.float32(c.cost_weight * r.uniform(0.5, 2.0)))
                if hasattr(c, "threshold_to_consider") and r.random() < 0.3:
                    c.threshold_to_consider = float(r.choice([0.2, 0.5, 1.4, 3.0]))
                for obj in (g, o):
                    obj.set_critics(cr)
                log.append(f"{step}:critics {name}")
            elif op < 0.27:                     # setSpeedLimit
                f = float(r.choice([0.4, 0.7, 1.0]))
                for obj in (g, o):
                    obj.set_constraints(cfg.vx_max * f, cfg.vx_min * f, cfg.vy_max * f, cfg.wz_max * f)
                log.append(f"{step}:limit {f}")
            elif op < 0.30:
                for obj in (g, o):
                    obj.reset()
                if rng_mode:
                    # NoiseGenerator::reset draws again (noise_generator.cpp:54-63): the library's device RNG
                    # moves to its next epoch; the oracle is handed that epoch
                    o.set_noise(*g.get_noise())
                u = np.zeros_like(u)
                log.append(f"{step}:reset")
            elif op < 0.34:                     # stored noise again
                noise = make_noise(cfg.batch_size, cfg.time_steps, std=(cfg.vx_std, cfg.vy_std, cfg.wz_std), seed=int(r.integers(1, 1 << 30)))
                for obj in (g, o):
                    obj.set_noise(*noise)
                rng_mode = False
                log.append(f"{step}:noise")
            elif op < 0.38:                     # the device RNG
                s = int(r.integers(1, 1 << 40))
                for obj in (g, o):
                    obj.seed(s)
                rng_mode = True
                log.append(f"{step}:seed")
            elif op < 0.44 and rng_mode:        # a new epoch, in the foreground or behind the next tick
                how = str(r.choice(["sync", "async"]))
                (g.redraw_noise if how == "sync" else g.redraw_noise_async)()
                if how == "sync":
                    o.set_noise(*g.get_noise())
                log.append(f"{step}:redraw {how}")
                if how == "async":
                    # the tick takes the epoch; the oracle gets it right after the library's tick below
                    tk = Tick(x, y, yaw, tick.speed, px, py, pyaw, gx, gy, goal_checker_xy_tolerance=tick.goal_checker_xy_tolerance)
                    ug, og = g.optimize(tk, u)
                    o.set_noise(*g.get_noise())
                    uo, oo = o.optimize(tk, u)
                    F.check(case, step, d, ug, og, uo, oo, g.get_costs(), o.get_costs())
                    u = np.concatenate([uo[:, 1:], uo[:, -1:]], axis=1)
                    ticks += 1
            else:                               # a tick; sometimes with a new plan, a jump of the pose
                if r.random() < 0.15:
                    P = int(r.choice([2, 7, 30, 60, 150]))
                    s = float(r.choice([0.05, 0.1])) * np.arange(P)
                    k = float(r.choice([0.0, 0.0, 0.4, -0.8]))
                    if k == 0.0:
                        qx, qy, qyaw = x + s, np.full_like(s, y), np.zeros_like(s)
                    else:
                        qx, qy, qyaw = x + np.sin(k * s) / k, y + (1.0 - np.cos(k * s)) / k, k * s
                    W = cells.shape[1] * res
                    Hh = cells.shape[0] * res
                    keep = (qx > 0.1) & (qx < W - 0.1) & (qy > 0.1) & (qy < Hh - 0.1)
                    n = max(2, int(np.argmin(keep)) if not keep.all() else len(s))
                    px, py, pyaw = qx[:n].astype(np.float32), qy[:n].astype(np.float32), qyaw[:n].astype(np.float32)
                    gx, gy = float(px[-1]), float(py[-1])
                    log.append(f"{step}:plan {n}")
                if r.random() < 0.05:
                    x += float(r.uniform(-0.5, 0.5))
                    yaw += float(r.uniform(-1.0, 1.0))
                    log.append(f"{step}:jump")
                tk = Tick(x, y, yaw, tick.speed, px, py, pyaw, gx, gy, goal_checker_xy_tolerance=tick.goal_checker_xy_tolerance)
                ug, og = g.optimize(tk, u)
                uo, oo = o.optimize(tk, u)
                F.check(case, step, d, ug, og, uo, oo, g.get_costs(), o.get_costs())
                if store and r.random() < 0.3 and not og.fail_flag:
                    tg, to = g.get_generated_trajectories(), o.get_trajectories()
                    for a, b, nm in zip(tg, to, "xyθ"):
                        if np.max(np.abs(a - b)) > 2e-5:
                            raise F.Mismatch(f"step {step}: trajectories {nm} differ by {np.max(np.abs(a - b)):.3g}")
                u = np.concatenate([uo[:, 1:], uo[:, -1:]], axis=1)
                # the robot moves along the emitted Twist for one model step
                x += float(uo[0, 0] * np.cos(yaw) - uo[1, 0] * np.sin(yaw)) * cfg.model_dt
                y += float(uo[0, 0] * np.sin(yaw) + uo[1, 0] * np.cos(yaw)) * cfg.model_dt
                yaw += float(uo[2, 0]) * cfg.model_dt
                ticks += 1
    except F.Mismatch as e:
        raise F.Mismatch(f"{e}   after {log}") from None
    finally:
        g.close()
        o.close()
    return d, ticks, log


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    only = [int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("only=")]
    cases = only or range(first, first + count)
    F.DETAIL = False
    bad = 0
    t0 = time.time()
    for case in cases:
        try:
            d, ticks, log = run(case)
            print(f"case {case}: ok  B {d['B']} T {d['T']} flags {d['flags']:#x} pass {d['env_pass'] or '-'} critics {d['critic_kind']} "
                  f"{ticks} ticks, {len(log)} other calls ({time.time() - t0:.0f} s)", flush=True)
        except Exception as e:
            msg = str(e).splitlines()[0] if str(e) else type(e).__name__
            if "more than 63 samples per trajectory" in msg or "both ObstaclesCritic and CostCritic" in msg:
                continue
            bad += 1
            print(f"case {case}: FAILED  {type(e).__name__}: {msg[:700]}", flush=True)
            if only:
                traceback.print_exc()
    print(f"{bad} of {len(list(cases))} cases failed", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
