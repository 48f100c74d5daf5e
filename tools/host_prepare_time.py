"""Developer tool: host time of the per-tick preparation (smpc_shard_begin = prepare_tick + the
asynchronous upload, no kernels) — the part of a tick during which the GPU waits for the host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_ctx

g, scn, cfg = make_ctx(262144, 64, 200)
g.optimize(scn.tick, scn.u0)
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    g.shard_begin(scn.tick, scn.u0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"smpc_shard_begin (prepare_tick + upload enqueue) through ctypes: {1e6 * (t1 - t0) / n:.2f} us per call")
