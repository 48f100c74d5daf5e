#!/usr/bin/env python3
"""Developer tool: ticks with regenerate_noises = true (smpc_redraw_noise_async behind every tick)
at the metric's batch: wall time per tick and scoring passes per tick.  Under
`rocprofv3 --kernel-trace --stats` it shows what the draw and the passes cost.
python tools/regen_tick.py [B] [T] [ticks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_ctx, shift

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 60
g, scn, cfg = make_ctx(B, T, 200)
u = scn.u0
for _ in range(20):
    un, out = g.optimize(scn.tick, u); g.redraw_noise_async(); u = shift(un)
torch.cuda.synchronize()
t0 = time.perf_counter()
passes = 0
for _ in range(N):
    un, out = g.optimize(scn.tick, u); g.redraw_noise_async(); u = shift(un)
    passes += out.passes
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / N
print(f"[regen] {B}x{T}: {el*1e3:.4f} ms/tick, {passes/N:.2f} scoring passes per tick")
g.close()
