#!/usr/bin/env python3
"""Developer tool: sample the GPU's clocks and power (rocm-smi) while ticks of the headline
workload run from a cold start — is the slow start of tools/ramp.py the power management?"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5)
        return r.stdout.strip().replace("\n", " | ")[:400]
    except Exception as e:
        return f"rocm-smi failed: {e}"


print("idle:", smi())
from bench import make_ctx, shift
g, scn, cfg = make_ctx(2097152, 64, 200)
samples = []
stop = False


def watch():
    while not stop:
        samples.append((time.perf_counter(), smi()))
        time.sleep(0.02)


th = threading.Thread(target=watch)
t_start = time.perf_counter()
th.start()
u = scn.u0
marks = []
for k in range(600):
    t0 = time.perf_counter()
    un, out = g.optimize(scn.tick, u)
    marks.append((t0 - t_start, (time.perf_counter() - t0) * 1e6))
stop = True
th.join()
for t, s in samples[:14]:
    k = sum(1 for m in marks if m[0] <= t - t_start)
    print(f"t = {1e3 * (t - t_start):7.1f} ms (tick {k:3d}): {s}")
print("tick time per 50:", " ".join(f"{sum(m[1] for m in marks[a:a+50]) / 50:.0f}" for a in range(0, 600, 50)))
