#!/usr/bin/env python3
"""Developer tool: which device work in front of the first tick brings the clocks up (see
tools/clock_watch.py: sclk climbs from 2.0 to 2.34 GHz over the first ~150 ms of this load)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_ctx, shift

def ticks(g, scn, n=30):
    u = scn.u0; ts = []
    for k in range(n):
        t0 = time.perf_counter(); un, out = g.optimize(scn.tick, u); u = shift(un)
        ts.append((time.perf_counter() - t0) * 1e6)
    return ts

for mode in ("nothing", "200 ms of device-RNG redraws", "200 ms of fp32 matmuls", "400 ms of device-RNG redraws"):
    g, scn, cfg = make_ctx(2097152, 64, 200)
    time.sleep(0.5)
    t0 = time.perf_counter()
    if "redraws" in mode:
        lim = 0.4 if mode.startswith("400") else 0.2
        while time.perf_counter() - t0 < lim:
            g.redraw_noise(); torch.cuda.synchronize()
        g.seed(1234)
    elif "matmul" in mode:
        a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
        while time.perf_counter() - t0 < 0.2:
            for _ in range(4): c = a @ b
            torch.cuda.synchronize()
    ts = ticks(g, scn)
    print(f"{mode:32s}: first tick {ts[0]:.0f}, ticks 5-24 mean {sum(ts[5:25])/20:.0f} us")
    g.close()
