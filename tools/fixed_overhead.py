#!/usr/bin/env python3
"""Developer tool: wall time of one smpc_optimize tick at tiny batch = fixed per-tick overhead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from mpcholonavigation_amd import _abi as A
from bench import make_ctx, shift
for B, flags in ((64, 0), (64, A.SMPC_FLAG_PROFILE), (65536, 0)):
    from mpcholonavigation_amd.optimizer import Smpc
    from mpcholonavigation_amd.synthetic import make_scenario
    from mpcholonavigation_amd.tick import default_config, default_critics
    cfg = default_config(batch_size=B, time_steps=64, flags=flags)
    scn = make_scenario(64)
    g = Smpc(cfg); g.set_critics(default_critics()); g.set_costmap(scn.cells, 0.0, 0.0, 0.05); g.seed(1)
    u = scn.u0
    for _ in range(50): u2, out = g.optimize(scn.tick, u)
    t0 = time.perf_counter()
    N = 500
    for _ in range(N): u2, out = g.optimize(scn.tick, u)
    t1 = time.perf_counter()
    # raw C call without the Python wrapper's copies
    tc = scn.tick.c; uu = np.ascontiguousarray(u).copy(); o = A.SmpcTickOut()
    lib = g.lib
    t2 = time.perf_counter()
    for _ in range(N): lib.smpc_optimize(g.h, C.byref(tc), uu.ctypes.data_as(C.c_void_p), C.byref(o))
    t3 = time.perf_counter()
    print(f"B={B} flags={flags}: python wrapper {1e6*(t1-t0)/N:.1f} us/tick, raw ctypes call {1e6*(t3-t2)/N:.1f} us/tick, device {out.device_ms*1e3:.1f} us")
