#!/usr/bin/env python3
"""Developer tool: profiles/<round>/traffic.json from the two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) over tools/traffic.py.

The furthest-only pass (smpc_pass<R,1,..>) reads the three noise tensors and nothing else of
size, so its FETCH_SIZE calibrates the counter's unit on this access pattern (gfx950 counts
128-B requests as 64 B: MI355X_MICROARCH.md, HBM section); the scoring pass's traffic is
FETCH_SIZE x that factor + WRITE_SIZE, in bytes per launch."""
import csv, glob, json, sys

fetch_dir, write_dir, B, T, costmap, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]

def means(d, counter):
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            acc.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

fetch, write = means(fetch_dir, "FETCH_SIZE"), means(write_dir, "WRITE_SIZE")
cal = [k for k in fetch if "smpc_pass<" in k and ", 1," in k][0]
score = [k for k in fetch if ("smpc_pass_lane" in k) or ("smpc_pass<" in k and ", 1," not in k)][0]
known = 12 * B * T
factor = known / (fetch[cal] * 1024.0)
traffic = fetch[score] * 1024.0 * factor + write[score] * 1024.0
json.dump({
    "workload": {"rollouts": B, "horizon": T, "costmap": costmap},
    "kernel": score.replace("void ", ""),
    "FETCH_SIZE_KB_mean": fetch[score], "WRITE_SIZE_KB_mean": write[score],
    "calibration": {"kernel": cal.replace("void ", "") + " (furthest-only pass: reads the three noise tensors)",
                    "known_bytes": known, "FETCH_SIZE_KB_mean": fetch[cal],
                    "fetch_correction_factor": factor},
    "traffic_bytes_per_launch": traffic,
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) -- python3 tools/traffic.py",
}, open(out, "w"), indent=1)
print(open(out).read())
