#!/usr/bin/env python3
"""Developer tool: wall time per tick with the reduction inside the scoring launch (smpc_tail.h)
against the separate smpc_reduce_partials launch, same process, interleaved blocks of ticks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from mpcholonavigation_amd import _abi as A
from mpcholonavigation_amd.optimizer import Smpc
from mpcholonavigation_amd.synthetic import make_scenario
from mpcholonavigation_amd.tick import default_config, default_critics

VARIANTS = {"default": {}}
for a in sys.argv[1:]:
    k, v = a.split("=", 1)
    VARIANTS[a] = {k: v}


def make(B, T, env):
    os.environ.update(env)
    cfg = default_config(batch_size=B, time_steps=T)
    scn = make_scenario(T)
    g = Smpc(cfg); g.set_critics(default_critics()); g.set_costmap(scn.cells, 0.0, 0.0, 0.05); g.seed(1)
    for k in env:
        os.environ.pop(k, None)
    return g, scn


def block(g, scn, n):
    tc = scn.tick.c; uu = np.ascontiguousarray(scn.u0).copy(); o = A.SmpcTickOut()
    t0 = time.perf_counter()
    for _ in range(n):
        g.lib.smpc_optimize(g.h, C.byref(tc), uu.ctypes.data_as(C.c_void_p), C.byref(o))
    return (time.perf_counter() - t0) / n * 1e6, o.passes


SIZES = ((64, 64), (2000, 56), (4096, 64), (16384, 64), (32768, 64), (65536, 64), (70000, 64), (131072, 64), (262144, 64), (2097152, 64))
if os.environ.get("TAILAB_SIZES"):      # e.g. TAILAB_SIZES=16384x56,32768x56
    SIZES = tuple(tuple(int(v) for v in s.split("x")) for s in os.environ["TAILAB_SIZES"].split(","))
for B, T in SIZES:
    ctxs = {k: make(B, T, e) for k, e in VARIANTS.items()}
    n = 2000 if B <= 262144 else 300
    for k, (g, scn) in ctxs.items():
        block(g, scn, 200)
    res = {k: [] for k in ctxs}
    for rep in range(5):
        for k, (g, scn) in ctxs.items():
            res[k].append(block(g, scn, n)[0])
    print(f"{B}x{T}: " + ", ".join(f"{k} {min(v):.1f} us/tick (median {sorted(v)[2]:.1f})" for k, v in res.items()))
    for g, _ in ctxs.values():
        g.close()
