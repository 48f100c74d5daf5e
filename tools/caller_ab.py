#!/usr/bin/env python3
"""Developer tool: wall time per closed-loop tick by who issues the ticks — bench.py's Python loop
(Smpc.optimize + numpy shift), a bare ctypes loop, and the compiled loop of host/tick_loop.cpp
(sortham_run_ticks) — same context, interleaved blocks.   tools/caller_ab.py [BxT ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from mpcholonavigation_amd import _abi as A, host_optimizer as H
from mpcholonavigation_amd.optimizer import Smpc
from mpcholonavigation_amd.synthetic import make_scenario
from mpcholonavigation_amd.tick import default_config, default_critics


def shift(u):
    return np.concatenate([u[:, 1:], u[:, -1:]], axis=1)


def python_loop(g, scn, n):
    u = scn.u0
    t0 = time.perf_counter()
    for _ in range(n):
        u_new, out = g.optimize(scn.tick, u)
        u = shift(u_new)
    return (time.perf_counter() - t0) / n * 1e6


def ctypes_loop(g, scn, n):
    tc = scn.tick.c; uu = np.ascontiguousarray(scn.u0).copy(); o = A.SmpcTickOut()
    t0 = time.perf_counter()
    for _ in range(n):
        g.lib.smpc_optimize(g.h, C.byref(tc), uu.ctypes.data_as(C.c_void_p), C.byref(o))
    return (time.perf_counter() - t0) / n * 1e6


def compiled_loop(g, scn, n):
    t0 = time.perf_counter()
    H.run_ticks(g, scn.tick, scn.u0, n, H.TICKS_SHIFT)
    return (time.perf_counter() - t0) / n * 1e6


sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2000, 56), (65536, 64), (262144, 64), (2097152, 64)]
for B, T in sizes:
    cfg = default_config(batch_size=B, time_steps=T)
    scn = make_scenario(T)
    g = Smpc(cfg); g.set_critics(default_critics()); g.set_costmap(scn.cells, 0.0, 0.0, 0.05); g.seed(1)
    n = 2000 if B <= 262144 else 300
    loops = {"python (bench.py)": python_loop, "ctypes": ctypes_loop, "compiled": compiled_loop}
    for f in loops.values():
        f(g, scn, 200)
    res = {k: [] for k in loops}
    for rep in range(5):
        for k, f in loops.items():
            res[k].append(f(g, scn, n))
    print(f"{B}x{T}: " + ", ".join(f"{k} {min(v):.1f} us/tick (median {sorted(v)[2]:.1f})" for k, v in res.items()), flush=True)
    g.close()
