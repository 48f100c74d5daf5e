#!/usr/bin/env python3
"""Developer tool: wall time per tick over the first ticks of a fresh process — is the slow start
the device (clocks, caches) or the workload (the control sequence converging)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_ctx, shift

import torch
for mode in ("closed loop (u shifted every tick)", "the same u every tick", "the same u every tick, after 80 ms of other device work"):
    g, scn, cfg = make_ctx(2097152, 64, 200)
    u = scn.u0
    if "after" in mode:
        a = torch.ones(64 * 1024 * 1024, device="cuda")
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.08:
            for _ in range(20):
                a = a * 1.0000001 + 1e-9
            torch.cuda.synchronize()
    ts, ks = [], []
    g.set_profile(True)        # (HIP events around the scoring pass: its own duration per tick)
    for k in range(400):
        t0 = time.perf_counter()
        un, out = g.optimize(scn.tick, u)
        if mode.startswith("closed"):
            u = shift(un)
        ts.append((time.perf_counter() - t0) * 1e6)
        ks.append(out.score_pass_ms * 1e3)
    print(mode)
    print("  tick, first 10:", " ".join(f"{t:.0f}" for t in ts[:10]))
    print("  tick, then per 10:", " ".join(f"{sum(ts[a:a+10])/10:.0f}" for a in range(10, 400, 10)))
    print("  scoring pass, first 10:", " ".join(f"{t:.0f}" for t in ks[:10]))
    print("  scoring pass, then per 10:", " ".join(f"{sum(ks[a:a+10])/10:.0f}" for a in range(10, 400, 10)))
    g.close()
