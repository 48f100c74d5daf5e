#!/usr/bin/env python3
"""Developer tool: a soak of the per-tick hand-over through the PCIe BAR.  Hundreds of
thousands of ticks whose inputs change EVERY tick (the pose creeps, so the tick block — u, the
per-candidate tables, the launch constants — differs), on the passes that read the block from
device memory: the lane pass, the split pass, a group of eight.  Every tick the library checks
that the scoring pass echoed this tick's number (a stale block fails the tick with SMPC_ERR_DEVICE);
here additionally every 1000th tick is compared with a second context that takes the stream copy
(SMPC_NO_BAR_TICK=1).   python tools/soak_bar.py [seconds per case]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpcholonavigation_amd import _abi as A
from mpcholonavigation_amd.optimizer import Smpc, SmpcGroup
from mpcholonavigation_amd.synthetic import make_scenario
from mpcholonavigation_amd.tick import Tick, default_config, default_critics

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0


def make(B, T, flags=0, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    cfg = default_config(batch_size=B, time_steps=T, flags=flags)
    scn = make_scenario(T)
    g = Smpc(cfg); g.set_critics(default_critics()); g.set_costmap(scn.cells, 0.0, 0.0, 0.05); g.seed(B)
    for k in (env or {}):
        os.environ.pop(k, None)
    return g, scn


def creep(scn, k):
    t = scn.tick
    return Tick(t.pose_x + 1e-4 * (k % 2000), t.pose_y + 5e-5 * (k % 700), t.pose_yaw + 1e-4 * (k % 300), t.speed, t.path_x,
                t.path_y, t.path_yaw, t.goal_x, t.goal_y)


for name, B, T, flags in (("lane pass 65536x64", 65536, 64, 0), ("split pass 16384x64", 16384, 64, 0),
                          ("lane pass 70000x56", 70000, 56, 0)):
    g, scn = make(B, T, flags)
    ref, _ = make(B, T, flags, env={"SMPC_NO_BAR_TICK": "1"})
    u = scn.u0
    t0 = time.perf_counter()
    k = 0
    kinds = set()
    while time.perf_counter() - t0 < budget:
        tk = creep(scn, k)
        un, out = g.optimize(tk, u)
        kinds.add(int(out.pass_kind))
        if k % 1000 == 0:
            ur, outr = ref.optimize(tk, u)
            assert np.array_equal(un, ur), (name, k)
            assert out.furthest_reached_path_point == outr.furthest_reached_path_point
        u = np.concatenate([un[:, 1:], un[:, -1:]], axis=1)
        k += 1
    print(f"[soak] {name}: {k} ticks in {time.perf_counter() - t0:.1f} s, pass kinds {sorted(kinds)}, no stale block", flush=True)
    g.close(); ref.close()

# a group of eight: one hand-over for all members
members = [make(16384, 64, A.SMPC_FLAG_LANE_PER_ROLLOUT) for _ in range(8)]
grp = SmpcGroup([m for m, _ in members])
us = [s.u0 for _, s in members]
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < budget:
    ticks = [creep(s, k + 17 * i) for i, (_, s) in enumerate(members)]
    res = grp.optimize(ticks, us)
    us = [np.concatenate([u[:, 1:], u[:, -1:]], axis=1) for u, _ in res]
    k += 1
print(f"[soak] group of 8 x 16384x64: {k} rounds in {time.perf_counter() - t0:.1f} s, no stale block", flush=True)
grp.close()
