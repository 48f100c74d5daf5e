"""Developer aid: furthest-point prediction on the grouped-contexts test scenario (alone contexts)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from mpcholonavigation_amd import _abi as A
from mpcholonavigation_amd.optimizer import Smpc
from mpcholonavigation_amd.synthetic import make_scenario, make_noise
from mpcholonavigation_amd.tick import default_config, Tick
from tests.helpers import configure
B, T = 2048, 56
for i in range(2):
    cfg = default_config(batch_size=B, time_steps=T, flags=A.SMPC_FLAG_LANE_PER_ROLLOUT)
    scn = make_scenario(T, seed=60 + i, path_points=40 + 5 * i)
    g = Smpc(cfg); configure(g, scn, noise=make_noise(B, T, seed=900 + i))
    u = scn.u0
    print("member", i, "path spacing", np.hypot(np.diff(scn.tick.path_x), np.diff(scn.tick.path_y))[:3], file=sys.stderr)
    for k in range(8):
        t = scn.tick
        tk = Tick(t.pose_x + 0.02 * k, t.pose_y, t.pose_yaw, (0.3, 0.0, 0.0), t.path_x, t.path_y, t.path_yaw, t.goal_x, t.goal_y)
        ug, og = g.optimize(tk, u)
        print(f" tick {k}: furthest {og.furthest_reached_path_point} passes {og.passes} vx0 {ug[0,0]:.3f}", file=sys.stderr)
        u = np.concatenate([ug[:, 1:], ug[:, -1:]], axis=1)
    g.close()
