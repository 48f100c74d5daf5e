"""Developer tool: where the reduction inside the scoring launch (smpc_tail.h) spends its time.
s_memrealtime stamps (10 ns) of thread 0 of every reducing block, last launch of a few ticks."""
import ctypes as C
import os
import sys

os.environ["SMPC_LANE_TIMELINE"] = "1"
os.environ["SMPC_FUSED_REDUCE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpcholonavigation_amd.optimizer import Smpc
from mpcholonavigation_amd.synthetic import make_scenario
from mpcholonavigation_amd.tick import default_config, default_critics

NAMES = ["partial stores complete + barrier", "ticket (atomic add)", "wait for the grid's last block",
         "headers + rows loaded, min/max known", "weights, rows accumulated (LDS)", "columns finished, host stores issued",
         "system-scope fence + barrier", "completion word stored"]
for B, T in ((64, 64), (2000, 56), (65536, 64), (262144, 64)):
    cfg = default_config(batch_size=B, time_steps=T)
    scn = make_scenario(T)
    g = Smpc(cfg); g.set_critics(default_critics()); g.set_costmap(scn.cells, 0.0, 0.0, 0.05); g.seed(1)
    for _ in range(20):
        g.optimize(scn.tick, scn.u0)
    fn = g.lib.smpc_debug_tail_timeline
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    buf = (C.c_ulonglong * 128)()
    assert fn(g.h, buf) == 0
    print(f"{B}x{T}: microseconds per stage, one column per reducing block")
    reds = [r for r in range(8) if buf[16 * r] != 0]
    t_last = max(buf[16 * r + 3] for r in reds)     # the moment the last block was counted in
    for k, n in enumerate(NAMES):
        print(f"  {n:42s} " + " ".join(f"{(buf[16 * r + k + 1] - buf[16 * r + k]) / 100.0:7.2f}" for r in reds))
    print(f"  {'completion word after the last ticket':42s} " + " ".join(f"{(buf[16 * r + 8] - t_last) / 100.0:7.2f}" for r in reds))
    g.close()
