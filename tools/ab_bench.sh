#!/bin/bash
# Developer aid: bench.py twice inside ONE gpurun call (boxes differ by up to 7 %), the second time
# with the environment assignment given as $1 (e.g. SMPC_NO_FUSED_REDUCE=1); prints both summaries.
set -e
out=${2:-gpurun_out/ab}
mkdir -p "$out"
python bench.py --no-cpu-baseline > "$out/a.json" 2> "$out/a.err"
env "$1" python bench.py --no-cpu-baseline > "$out/b.json" 2> "$out/b.err"
python - "$out" "$1" <<PY
import json, sys
out, what = sys.argv[1], sys.argv[2]
for n, label in (("a", "default"), ("b", what)):
    d = json.load(open(f"{out}/{n}.json"))
    r = d["roofline"]
    print(f"{label}: {d['ms_per_step']*1e3:.1f} us/tick, pass {r['avg_launch_ms']*1e3:.1f} us, frac {r['frac']:.3f}, frac_tick {r['frac_tick']:.3f}, moving {d['moving_pose']['ms_per_step']*1e3:.1f} us")
    for k, v in d["other_configs"].items():
        print(f"    {k}: {v['rollouts_per_s']:.4g} rollouts/s" + (f", {v['ms_per_step']*1e3:.1f} us/tick" if 'ms_per_step' in v else ""))
PY
