#!/usr/bin/env python3
"""Developer tool: what a tick costs when the robot is inside the goal critics' thresholds
(GoalAngle live; PathAlign / PathFollow / PreferForward gated off) against the same batch
away from the goal."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpcholonavigation_amd.optimizer import Smpc
from mpcholonavigation_amd.synthetic import make_scenario
from mpcholonavigation_amd.tick import default_config, default_critics

for B, T in ((2000, 56), (65536, 64), (262144, 64), (2097152, 64)):
    for near in (False, True):
        cfg = default_config(batch_size=B, time_steps=T)
        scn = make_scenario(T, near_goal=near)
        g = Smpc(cfg); g.set_critics(default_critics()); g.set_costmap(scn.cells, 0.0, 0.0, 0.05); g.seed(1)
        n = 200 if B <= 262144 else 30
        for _ in range(10):
            u, out = g.optimize(scn.tick, scn.u0)
        t0 = time.perf_counter()
        for _ in range(n):
            u, out = g.optimize(scn.tick, scn.u0)
        dt = (time.perf_counter() - t0) / n
        print(f"{B}x{T} {'near the goal' if near else 'on the way   '}: {dt*1e6:8.1f} us/tick, pass_kind {out.pass_kind}, passes {out.passes}")
        g.close()
