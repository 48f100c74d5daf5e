#!/usr/bin/env python3
"""Developer tool: N ticks of one bench configuration (what bench.py's other_configs time), to put
behind `rocprofv3 --kernel-trace --stats`.   python tools/config_ticks.py B T MAP [ticks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_ctx, shift

B, T, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 300
g, scn, cfg = make_ctx(B, T, M)
u = scn.u0
for _ in range(N):
    un, out = g.optimize(scn.tick, u)
    u = shift(un)
print(f"[config_ticks] {B}x{T} map {M}: {N} ticks, passes last tick {out.passes}, pass kind {out.pass_kind}")
g.close()
