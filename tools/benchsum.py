import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g  tick %.4f ms  pass %.4f ms" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
for k,v in d.get("other_configs",{}).items(): print(k, "tick %.4f pass %.4f" % (v["ms_per_tick"], v["score_pass_ms"])) if "ms_per_tick" in v else print(k, "round %.4f ms" % v["ms_per_round"])
