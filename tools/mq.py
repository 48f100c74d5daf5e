import sys; sys.path.insert(0,'.')
import bench
for n, B in ((1,16384),(8,16384),(16,16384),(64,2048),(8,32768),(32,8192)):
    r = bench.time_multi_query(n, B, 64, 200, 200, 20)
    print(n, B, "%.3g rollouts/s  %.1f us per round  passes %.2f" % (r["rollouts_per_s"], 1e3*r["ms_per_round"], r["passes_per_query"]))
