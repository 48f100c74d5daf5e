#!/usr/bin/env python3
"""Developer tool.  Two modes:
  run B T [n]      — n ticks of smpc_optimize (the command to put behind rocprofv3 --kernel-trace
                     --memory-copy-trace --output-format csv)
  report DIR       — read the kernel / memory-copy traces under DIR and print, per steady-state
                     tick, where the time goes: copy, gap, scoring pass, gap, reduction, and the
                     host's turn-around until the next tick's first device activity."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(B, T, n):
    import numpy as np, ctypes as C
    from mpcholonavigation_amd import _abi as A
    from mpcholonavigation_amd.optimizer import Smpc
    from mpcholonavigation_amd.synthetic import make_scenario
    from mpcholonavigation_amd.tick import default_config, default_critics
    cfg = default_config(batch_size=B, time_steps=T)
    scn = make_scenario(T)
    g = Smpc(cfg); g.set_critics(default_critics()); g.set_costmap(scn.cells, 0.0, 0.0, 0.05); g.seed(1)
    tc = scn.tick.c; uu = np.ascontiguousarray(scn.u0).copy(); o = A.SmpcTickOut()
    for _ in range(n):
        g.lib.smpc_optimize(g.h, C.byref(tc), uu.ctypes.data_as(C.c_void_p), C.byref(o))
    g.close()


def report(d):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
    ev.sort()
    ev = ev[len(ev) // 2:]          # steady state
    # a tick starts at each scoring pass
    import collections
    rows = collections.defaultdict(list)
    prev_end = None
    for s, e, name in ev:
        rows[name + " duration"].append((e - s) / 1e3)
        if prev_end is not None:
            rows["gap before " + name].append((s - prev_end) / 1e3)
        prev_end = e
    for k in sorted(rows):
        v = sorted(rows[k])
        print(f"  {k:78s} n={len(v):5d} median {v[len(v)//2]:8.2f} us   p10 {v[len(v)//10]:8.2f}   p90 {v[9*len(v)//10]:8.2f}")


if sys.argv[1] == "run":
    run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 400)
else:
    report(sys.argv[2])
