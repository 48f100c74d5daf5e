"""Developer tool: where one launch of the lane-per-rollout pass spends its time
(SMPC_LANE_TIMELINE=1; smpc_debug_lane_timeline is a debug export, not part of include/smpc.h).
Stamps are shader clocks of wave 0 of every block."""
import ctypes as C
import os
import sys

os.environ["SMPC_LANE_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_ctx, shift

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
GHZ = float(sys.argv[2]) if len(sys.argv) > 2 else 2.2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
M = int(sys.argv[4]) if len(sys.argv) > 4 else 200
g, scn, cfg = make_ctx(B, T, M)
u = scn.u0
for _ in range(5):
    u_new, out = g.optimize(scn.tick, u)
    u = shift(u_new)
fn = g.lib.smpc_debug_lane_timeline
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
buf = (C.c_double * 29)()
nb = C.c_uint32(0)
assert fn(g.h, buf, C.byref(nb)) == 0
names = ["entry (spread over blocks)", "LDS staged (barrier)", "constants", "group 1", "group 2",
         "all waves at the final barrier", "partial written"]
print(f"B={B}, {nb.value} blocks; microseconds at {GHZ} GHz (min / median / max over blocks)")
tot = 0.0
for k, n in enumerate(names):
    mn, md, mx = (buf[3 * k + j] / (GHZ * 1e3) for j in range(3))
    tot += md
    print(f"  {n:32s} {mn:7.2f} {md:7.2f} {mx:7.2f}")
print(f"  sum of medians {tot:.2f} us")
print("  per wave, end of its groups after the constants (median): " +
      " ".join(f"w{w}={buf[21 + w] / (GHZ * 1e3):.1f}" for w in range(8)))
