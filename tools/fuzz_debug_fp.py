#!/usr/bin/env python3
"""Developer tool: for one tools/fuzz_parity.py case with consider_footprint, the rollouts whose
collision verdict differs between the library and the oracle, with the first colliding step as a
plain-Python restatement of costAtPose / footprintCost sees it.   tools/fuzz_debug_fp.py CASE"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import fuzz_parity as F
from mpcholonavigation_amd.optimizer import Smpc
from oracle.loader import Oracle
from tests.helpers import configure

case = int(sys.argv[1])
d = F.draw(case)
cfg, scn, tick, u0, cr, noise = F.build(d)
g, o = Smpc(cfg), Oracle(cfg)
fp = np.array([[0.25, 0.15], [0.25, -0.15], [-0.2, -0.15], [-0.2, 0.15]])
for obj in (g, o):
    obj.set_footprint(fp, 0.3)
    configure(obj, scn, critics=cr, noise=noise, track_unknown=d["track_unknown"])
ug, og = g.optimize(tick, u0)
uo, oo = o.optimize(tick, u0)
cg, co = g.get_costs(), o.get_costs()
bad = np.nonzero(np.abs(cg.astype(np.float64) - co) > 100)[0]
print(f"non_colliding {og.non_colliding} / {oo.non_colliding}; fail {og.fail_flag} / {oo.fail_flag}; {len(bad)} differ: {bad[:10]}")
tx, ty, tyaw = o.get_trajectories()
cells, res = scn.cells, scn.resolution
H, W = cells.shape
pic = 252.0 * math.exp(-10.0 * (0.3 - scn.inscribed_radius))
print("possibly inscribed cost", int(pic), "track_unknown", d["track_unknown"])


def w2m(wx, wy):
    if wx < 0 or wy < 0:
        return None
    mx, my = int(wx / res), int(wy / res)
    return (mx, my) if mx < W and my < H else None


def line(x0, y0, x1, y1):
    out = []
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    x, y = x0, y0
    xi1 = xi2 = 1 if x1 >= x0 else -1
    yi1 = yi2 = 1 if y1 >= y0 else -1
    if dx >= dy:
        xi1 = 0; yi2 = 0; den = dx; num = dx // 2; add = dy; n = dx
    else:
        xi2 = 0; yi1 = 0; den = dy; num = dy // 2; add = dx; n = dy
    for _ in range(n + 1):
        out.append((x, y))
        num += add
        if num >= den:
            num -= den; x += xi1; y += yi1
        x += xi2; y += yi2
    return out


def fp_cost(x, y, th):
    c, s = math.cos(th), math.sin(th)
    vs = []
    for fx, fy in fp:
        m = w2m(x + fx * c - fy * s, y + fx * s + fy * c)
        if m is None:
            return 254, "vertex off the map"
        vs.append(m)
    best = 0
    for i in range(len(vs)):
        a, b = vs[i], vs[(i + 1) % len(vs)]
        for (cx, cy) in line(a[0], a[1], b[0], b[1]):
            if not (0 <= cx < W and 0 <= cy < H):
                return -1, f"line cell ({cx},{cy}) outside"
            v = int(cells[cy, cx])
            if v == 254:
                return 254, f"lethal cell ({cx},{cy})"
            best = max(best, v)
    return best, "max over the outline"


for b in bad[:6]:
    for t in range(cfg.time_steps):
        m = w2m(float(tx[b, t]), float(ty[b, t]))
        if m is None:
            print(f"rollout {b} step {t}: centre off the map -> NO_INFORMATION ({'no ' if d['track_unknown'] else ''}collision)")
            if not d["track_unknown"]:
                break
            continue
        c = int(cells[m[1], m[0]])
        if c >= pic:
            f, why = fp_cost(float(tx[b, t]), float(ty[b, t]), float(tyaw[b, t]))
            coll = f == 254 or (f == 255 and not d["track_unknown"])
            if coll or t < 2:
                print(f"rollout {b} step {t}: centre ({tx[b,t]:.4f},{ty[b,t]:.4f}) cell {m} cost {c} -> footprint {f} ({why}) collide {coll}")
            if coll:
                break
        elif c == 254 or (c == 255 and not d["track_unknown"]):
            print(f"rollout {b} step {t}: centre cost {c}: collision without a footprint check")
            break
    else:
        print(f"rollout {b}: no collision found by the restatement (gpu cost {cg[b]}, oracle {co[b]})")
