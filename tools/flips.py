#!/usr/bin/env python3
"""Developer tool: how far is the GPU tick from the oracle on configs[1]-like inputs?
Counts per-rollout cost mismatches (cell flips) and prints the Twist / sequence errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpcholonavigation_amd.optimizer import Smpc
from oracle.loader import Oracle
from tests.helpers import make_case, configure, rel_err, twist
for B, T, M in ((65536, 64, 200), (16384, 128, 2000)):
    for seed in (1234, 77):
        cfg, scn, noise = make_case(B, T, map_size=M, noise_seed=seed)
        g, o = Smpc(cfg), Oracle(cfg)
        for x in (g, o):
            configure(x, scn, noise=noise)
        o.set_accumulate_double(True)
        ug, og = g.optimize(scn.tick, scn.u0)
        uo, oo = o.optimize(scn.tick, scn.u0)
        cg, co = g.get_costs().astype(np.float64), o.get_costs().astype(np.float64)
        d = np.abs(cg - co)
        tight = d <= 2e-4 * np.maximum(np.abs(co), 1.0)
        print(f"B={B} T={T} map={M} seed={seed}: hard {int((d > 100).sum())} soft {int((~tight).sum() - (d > 100).sum())} "
              f"of {B*T} lookups; twist rel {rel_err(twist(ug), twist(uo)):.2e} seq rel {rel_err(ug, uo):.2e}; "
              f"median |dcost| {np.median(d):.2e}")
