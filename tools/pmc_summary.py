#!/usr/bin/env python3
"""Developer tool: mean of each PMC counter per kernel from a rocprofv3 --pmc output dir; with B and T
also the derived figures of DESIGN.md §4.2 (wave-instructions per wave-step, share of a wave's life spent
issuing VALU).   tools/pmc_summary.py DIR [B T]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
B, T = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 0)
for k, d in acc.items():
    if "pass" not in k: continue
    print(k)
    m = {}
    for c, v in sorted(d.items()):
        m[c] = sum(v) / len(v)
        print(f"   {c:28s} n={len(v):3d} mean={m[c]:.4g}")
    if B and "SQ_INSTS_VALU" in m:
        print(f"   -> VALU wave-instructions per 64 rollout-steps ({B} x {T}): {m['SQ_INSTS_VALU'] * 64 / (B * T):.1f}"
              + (f", SALU {m['SQ_INSTS_SALU'] * 64 / (B * T):.1f}" if "SQ_INSTS_SALU" in m else ""))
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        for c, label in (("SQ_ACTIVE_INST_VALU", "issuing VALU"), ("SQ_ACTIVE_INST_ANY", "issuing anything"), ("SQ_WAIT_ANY", "waiting")):
            if c in m:
                print(f"   -> share of a wave's life {label}: {m[c] / m['SQ_WAVE_CYCLES']:.3f}")
