#!/usr/bin/env python3
"""Developer tool: per-critic VALU budget of the lane-per-rollout pass from one rocprofv3 PMC pass
over tools/ablate.py (which scores the same batch with subsets of the critics enabled):

    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES \
              SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d OUT -o t \
              -- python3 tools/ablate.py 262144 64
    python3 tools/valu_budget.py OUT/t_counter_collection.csv 262144 64
"""
import collections, csv, sys
path, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
disp = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    if "smpc_pass_lane" not in r["Kernel_Name"]:
        continue
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
ds = list(disp.values())
sets = ["none", "obstacles", "path_align", "path_follow", "prefer_forward", "all"]
per = len(ds) // len(sets)
print(f"{len(ds)} launches of the lane pass, {per} per critic set; {B} rollouts x {T} steps; wave-instructions per wave-step")
print(f"{'critics enabled':16s} {'kernel':44s} {'VALU':>6s} {'SALU':>6s} | per wave: {'VALU busy':>9s} {'issuing':>8s} {'waiting':>8s}")
base = None
for i, s in enumerate(sets):
    chunk = ds[i * per + 5:(i + 1) * per]
    m = lambda k: sum(d[k] for d in chunk) / len(chunk)
    valu, salu, wc = m("SQ_INSTS_VALU") * 64 / (B * T), m("SQ_INSTS_SALU") * 64 / (B * T), m("SQ_WAVE_CYCLES")
    if base is None:
        base = valu
    print(f"{s:16s} {chunk[0]['name'][5:49]:44s} {valu:6.1f} {salu:6.1f} |           {m('SQ_ACTIVE_INST_VALU') / wc:9.3f} "
          f"{m('SQ_ACTIVE_INST_ANY') / wc:8.3f} {m('SQ_WAIT_ANY') / wc:8.3f}   (+{valu - base:.1f} over none)")
