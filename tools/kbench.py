#!/usr/bin/env python3
"""Developer tool: one-shot check of a kernel change — parity of the lane pass against the oracle
on small batches (costs, flips, Twist), then the scoring pass's duration (HIP events) and the
tick's wall time at the bench sizes.  Needs a GPU.   python tools/kbench.py [tag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpcholonavigation_amd import _abi as A
from bench import make_ctx, shift, algorithmic_bytes
from tests.helpers import assert_parity, configure, make_case
from mpcholonavigation_amd.optimizer import Smpc
from oracle.loader import Oracle, build

tag = sys.argv[1] if len(sys.argv) > 1 else ""
build()
PARITY = () if os.environ.get("KBENCH_NO_PARITY") else ((16384, 64, 200), (8192, 40, 200), (4096, 64, 2000), (4096, 128, 2000), (8192, 128, 200))
for (B, T, M) in PARITY:
    cfg, scn, noise = make_case(B, T, map_size=M)
    cfg.flags |= A.SMPC_FLAG_LANE_PER_ROLLOUT
    g, o = Smpc(cfg), Oracle(cfg)
    for obj in (g, o):
        configure(obj, scn, noise=noise)
    ug, og = g.optimize(scn.tick, scn.u0)
    uo, oo = o.optimize(scn.tick, scn.u0)
    assert_parity(ug, og, uo, oo, g.get_costs(), o.get_costs(), max_flips=2,
                  label=f"{tag} {B}x{T} map {M} pass_kind {og.pass_kind}")
    g.close()

import torch
SIZES = [tuple(int(v) for v in s.split("x")) for s in os.environ["KBENCH_SIZES"].split(",")] if os.environ.get("KBENCH_SIZES") else [(262144, 64, 200), (2097152, 64, 200), (262144, 128, 2000)]
for (B, T, M) in SIZES:
    g, scn, cfg = make_ctx(B, T, M)
    u = scn.u0
    for _ in range(5):
        un, out = g.optimize(scn.tick, u); u = shift(un)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 60
    for _ in range(N):
        un, out = g.optimize(scn.tick, u); u = shift(un)
    el = (time.perf_counter() - t0) / N
    g.set_profile(True)
    ps = []
    for _ in range(40):
        un, out = g.optimize(scn.tick, u); u = shift(un)
        ps.append(out.score_pass_ms)
    by = algorithmic_bytes(B, T, M, M, 60)
    p = float(np.median(ps))
    print(f"[kbench {tag}] {B}x{T} map {M}: pass {p*1e3:.1f} us ({by/p/1e6/8000:.3f} of 8 TB/s), tick {el*1e6:.1f} us "
          f"({by/el/1e9/8000:.3f}), passes/tick {out.passes}, pass_kind {out.pass_kind}")
    g.close()
