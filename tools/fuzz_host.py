#!/usr/bin/env python3
"""Developer tool: randomised closed loops of the C++ host optimizer (host/optimizer.cpp over
libsmpc) against the oracle's restatement of the reference host logic: evalControl tick after
tick with the pose advanced by the emitted Twist, speed limits set and lifted, reset(), a costmap
that turns lethal for a few ticks (fallback, retries, the throw), controller frequencies with and
without shifting.  The draws of tools/fuzz_parity.py supply configuration, critics and scene.
    tools/fuzz_host.py FIRST COUNT [only=CASE]"""
import math
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import fuzz_parity as F
from mpcholonavigation_amd import _abi as A
from mpcholonavigation_amd.host_optimizer import Optimizer
from mpcholonavigation_amd.tick import Tick
from oracle.loader import OracleOptimizer

NAMES = {"obstacles": "ObstaclesCritic", "path_align": "PathAlignCritic", "path_follow": "PathFollowCritic",
         "goal_angle": "GoalAngleCritic", "prefer_forward": "PreferForwardCritic", "cost": "CostCritic", "goal": "GoalCritic",
         "constraint": "ConstraintCritic", "twirling": "TwirlingCritic", "path_angle": "PathAngleCritic",
         "velocity_deadband": "VelocityDeadbandCritic", "path_align_legacy": "PathAlignLegacyCritic"}
MODELS = {A.SMPC_MODEL_OMNI: "Omni", A.SMPC_MODEL_DIFF_DRIVE: "DiffDrive", A.SMPC_MODEL_ACKERMANN: "Ackermann"}


class Mismatch(AssertionError):
    pass


def run(case):
    d = F.draw(case)
    r = np.random.default_rng(7 * case + 3)
    d["B"] = min(d["B"], 4096)
    d["iters"] = 1
    d["footprint"] = ""
    d["rng"] = False
    d["temperature"] = max(d["temperature"], 0.3)
    cfg, scn, tick, u0, cr, noise = F.build(d)
    freq = float(r.choice([1.0 / cfg.model_dt, 1.0 / cfg.model_dt, 2.0 / cfg.model_dt]))
    retry = int(r.choice([1, 1, 2, 3]))
    names = [NAMES[c] for c in d["critics"]]
    h = Optimizer(cfg, cr, freq, critics=names, motion_model=MODELS[d["model"]], retry_attempt_limit=retry)
    o = OracleOptimizer(cfg, cr, freq, retry_attempt_limit=retry)
    events = []
    try:
        for x in (h, o):
            x.set_costmap(scn.cells, scn.origin_x, scn.origin_y, scn.resolution, track_unknown=d["track_unknown"])
            x.set_noise(*noise)
        lethal_from = int(r.integers(3, 9)) if r.random() < 0.25 else 99
        lethal_for = int(r.integers(1, 4))
        x_, y_, yaw_ = tick.pose_x, tick.pose_y, tick.pose_yaw
        speed = tick.speed
        for k in range(10):
            if r.random() < 0.12:
                lim, pct = (float(r.choice([30.0, 60.0, 100.0])), True) if r.random() < 0.5 else (float(r.choice([0.2, 0.4, 0.0])), False)
                for x in (h, o):
                    x.set_speed_limit(lim, pct)
                events.append(f"{k}:limit {lim}{'%' if pct else ''}")
            if r.random() < 0.05:
                for x in (h, o):
                    x.reset()
                events.append(f"{k}:reset")
            if k == lethal_from or k == lethal_from + lethal_for:
                cells = np.full_like(scn.cells, 254) if k == lethal_from else scn.cells
                for x in (h, o):
                    x.set_costmap(cells, scn.origin_x, scn.origin_y, scn.resolution, track_unknown=d["track_unknown"])
                events.append(f"{k}:{'lethal' if k == lethal_from else 'map back'}")
            tk = Tick(x_, y_, yaw_, speed, tick.path_x, tick.path_y, tick.path_yaw, tick.goal_x, tick.goal_y,
                      goal_checker_xy_tolerance=tick.goal_checker_xy_tolerance)
            res = []
            for x in (h, o):
                try:
                    res.append(x.eval_control(tk))
                except RuntimeError as e:
                    res.append(str(e))
            if isinstance(res[0], str) or isinstance(res[1], str):
                if not (isinstance(res[0], str) and isinstance(res[1], str)):
                    raise Mismatch(f"tick {k}: one side threw: {res[0] if isinstance(res[0], str) else 'ok'} / "
                                   f"{res[1] if isinstance(res[1], str) else 'ok'}")
                if ("fail to compute path" in res[0]) != ("fail to compute path" in res[1]):
                    raise Mismatch(f"tick {k}: different exceptions: {res[0]} / {res[1]}")
                events.append(f"{k}:threw")
                if np.abs(h.get_control_sequence() - o.get_control_sequence()).max() > 0:
                    raise Mismatch(f"tick {k}: control sequences differ after the throw")
                continue
            (tw_h, out_h), (tw_o, out_o) = res
            if out_h.fail_flag != out_o.fail_flag:
                raise Mismatch(f"tick {k}: fail_flag {out_h.fail_flag} / {out_o.fail_flag}")
            if out_o.furthest_valid and out_h.furthest_reached_path_point != out_o.furthest_reached_path_point:
                raise Mismatch(f"tick {k}: furthest {out_h.furthest_reached_path_point} / {out_o.furthest_reached_path_point}")
            err = float(np.max(np.abs(tw_h - tw_o)))
            scale = max(float(np.max(np.abs(tw_o))), 1e-2)
            if out_o.non_colliding and err > 5e-4 * scale:
                # (flips are not counted here: the bar is loose, a logic error is not subtle)
                if err > 2e-2 * scale:
                    raise Mismatch(f"tick {k}: twist {tw_h} / {tw_o}; non_colliding {out_h.non_colliding} / {out_o.non_colliding} "
                                   f"min_cost {out_h.min_cost} / {out_o.min_cost} sum_w {out_h.sum_w} / {out_o.sum_w} furthest "
                                   f"{out_h.furthest_reached_path_point} / {out_o.furthest_reached_path_point} passes {out_h.passes}")
                events.append(f"{k}:twist {err / scale:.1e}")
            uh, uo = h.get_control_sequence(), o.get_control_sequence()
            if out_o.non_colliding and float(np.max(np.abs(uh - uo))) > 2e-2 * max(float(np.max(np.abs(uo))), 1e-2):
                raise Mismatch(f"tick {k}: control sequence differs by {float(np.max(np.abs(uh - uo))):.3g}")
            ch, co = h.get_constraints(), o.get_constraints()
            if not np.array_equal(ch[0], co[0]) or ch[1] != co[1]:
                raise Mismatch(f"tick {k}: constraints {ch} / {co}")
            o.set_control_sequence(uh)
            # the robot follows the emitted Twist for one controller period
            dtc = 1.0 / freq
            x_ += (tw_h[0] * math.cos(yaw_) - tw_h[1] * math.sin(yaw_)) * dtc
            y_ += (tw_h[0] * math.sin(yaw_) + tw_h[1] * math.cos(yaw_)) * dtc
            yaw_ += tw_h[2] * dtc
            speed = (float(tw_h[0]), float(tw_h[1]), float(tw_h[2]))
        th, to = h.get_optimized_trajectory(), o.get_optimized_trajectory()
        if float(np.max(np.abs(th - to))) > 1e-4 * max(float(np.max(np.abs(to))), 1.0):
            raise Mismatch(f"optimized trajectory differs by {float(np.max(np.abs(th - to))):.3g}")
    finally:
        h.close()
        o.close()
    return d, freq, retry, events


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    only = [int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("only=")]
    cases = only or range(first, first + count)
    bad = skipped = 0
    t0 = time.time()
    for case in cases:
        try:
            d, freq, retry, ev = run(case)
            print(f"case {case}: ok  B {d['B']} T {d['T']} model {d['model']} freq {freq:g} retry {retry} critics {d['critic_kind']} "
                  f"events {ev} ({time.time() - t0:.0f} s)", flush=True)
        except Exception as e:
            msg = str(e).splitlines()[0] if str(e) else type(e).__name__
            if "more than 63 samples" in msg or "Controller period more then model dt" in msg:
                skipped += 1
                continue
            bad += 1
            print(f"case {case}: FAILED  {type(e).__name__}: {msg[:300]}\n    draw: {F.draw(case)}", flush=True)
            if only:
                traceback.print_exc()
    print(f"{bad} of {len(list(cases))} cases failed ({skipped} refused configurations skipped)", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
