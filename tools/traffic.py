#!/usr/bin/env python3
"""Developer tool for the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: runs a few ticks
of the bench workload in the exact two-pass mode, so that one process launches both the
furthest-only pass (reads the three noise tensors and nothing else of size: the known-byte
calibration kernel for this access pattern) and the scoring pass."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpcholonavigation_amd import _abi as A
from bench import make_ctx, shift

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
MAP = int(sys.argv[3]) if len(sys.argv) > 3 else 200
g, scn, cfg = make_ctx(B, T, MAP, flags=A.SMPC_FLAG_NO_SPECULATION)
u = scn.u0
for _ in range(6):
    u_new, out = g.optimize(scn.tick, u)
    u = shift(u_new)
print("noise bytes per pass:", 12 * B * T, "passes", out.passes)
