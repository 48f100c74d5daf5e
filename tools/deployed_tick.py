#!/usr/bin/env python3
"""Developer tool: tick time of the deployed configuration (robot_bringup/config/nav2_params.yaml:
2000 rollouts x 56 steps, nine critics) against the north star's five on the same batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpcholonavigation_amd import _abi as A
from mpcholonavigation_amd.optimizer import Smpc
from mpcholonavigation_amd.synthetic import make_scenario
from mpcholonavigation_amd.tick import default_config, default_critics

DEPLOYED = ("constraint", "cost", "goal", "goal_angle", "path_align", "path_follow", "path_angle",
            "prefer_forward", "twirling")
FIVE = ("obstacles", "path_align", "path_follow", "goal_angle", "prefer_forward")
ALL = ("obstacles", "path_align", "path_follow", "goal_angle", "prefer_forward", "cost", "goal", "constraint",
       "twirling", "path_angle", "velocity_deadband")
SIZES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2000, 56), (65536, 56)]
for B, T in SIZES:
    for label, names in (("five", FIVE), ("deployed nine", DEPLOYED)):
        cr = default_critics()
        for n in ALL:
            getattr(cr, n).enabled = 1 if n in names else 0
        cfg = default_config(batch_size=B, time_steps=T, flags=A.SMPC_FLAG_PROFILE)
        scn = make_scenario(T)
        g = Smpc(cfg)
        g.set_critics(cr)
        g.set_costmap(scn.cells, scn.origin_x, scn.origin_y, scn.resolution)
        g.seed(1)
        u = scn.u0
        ps, ds = [], []
        for k in range(60):
            un, out = g.optimize(scn.tick, u)
            u = np.concatenate([un[:, 1:], un[:, -1:]], axis=1)
            if k >= 10:
                ps.append(out.score_pass_ms); ds.append(out.device_ms)
        g.set_profile(False)
        t0 = time.perf_counter()
        for k in range(200):
            un, out = g.optimize(scn.tick, u)
            u = np.concatenate([un[:, 1:], un[:, -1:]], axis=1)
        el = (time.perf_counter() - t0) / 200
        print(f"{B}x{T} {label:14s}: tick {el*1e6:7.1f} us, scoring pass {np.median(ps)*1e3:7.1f} us, device {np.median(ds)*1e3:7.1f} us, "
              f"pass_kind {out.pass_kind}, passes {out.passes}")
        g.close()
