#!/usr/bin/env python3
"""Developer tool: the device RNG's draw of one epoch (three tensors) in microseconds, and the
RNG self-checks that guard a change of the fill kernel (same stream as the CPU twin).
SMPC_LIB selects the library (tools/kbench_all.py style A/B).   python tools/fill_time.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_ctx

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g, scn, cfg = make_ctx(B, T, 200)
for _ in range(5):
    g.redraw_noise()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(20):
        g.redraw_noise()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 20)
print(f"[fill {os.environ.get('SMPC_LIB', 'product')[-24:]}] {B}x{T}: {min(ts) * 1e6:.1f} us per epoch of three tensors (median {sorted(ts)[2] * 1e6:.1f})")
g.close()
