#!/usr/bin/env python3
"""Developer tool: randomised multi-query groups (smpc_group_optimize: BASELINE configs[4]).
2-8 members that share only the horizon — own batch size, critics, scene, noise — ticked three
times in one launch per tick; every member against a twin context ticked alone (same bits) and
against the oracle (tools/fuzz_parity.py's bar).   tools/fuzz_group.py FIRST COUNT [only=CASE]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import fuzz_parity as F
from mpcholonavigation_amd.optimizer import Smpc, SmpcGroup
from mpcholonavigation_amd.tick import Tick
from oracle.loader import Oracle
from tests.helpers import configure


def run(case):
    r = np.random.default_rng(11 * case + 5)
    n = int(r.integers(2, 9))
    T = int(r.choice([30, 40, 56, 56, 64, 64, 100]))
    members, twins, oracles, draws, builds = [], [], [], [], []
    lane = bool(r.random() < 0.3)
    if lane:
        os.environ["SMPC_PASS"] = "lane"
    try:
        for i in range(n):
            d = F.draw(100000 + 37 * case + i)
            d["T"], d["iters"], d["footprint"], d["rng"], d["flags"] = T, int(r.choice([1, 1, 2])), "", False, 0
            d["B"] = int(r.choice([64, 500, 2048, 4096, 16384]))
            d["pa_step"] = 4 if T > 64 else d["pa_step"]
            cfg, scn, tick, u0, cr, noise = F.build(d)
            draws.append(d)
            builds.append((cfg, scn, tick, u0, cr, noise))
            for lst in (members, twins):
                g = Smpc(cfg)
                configure(g, scn, critics=cr, noise=noise, track_unknown=d["track_unknown"])
                lst.append(g)
            o = Oracle(cfg)
            configure(o, scn, critics=cr, noise=noise, track_unknown=d["track_unknown"])
            oracles.append(o)
    finally:
        os.environ.pop("SMPC_PASS", None)
    grp = SmpcGroup(members)
    notes = []
    try:
        us = [b[3] for b in builds]
        for k in range(3):
            ticks = []
            for (cfg, scn, tick, u0, cr, noise) in builds:
                ticks.append(Tick(tick.pose_x + 0.02 * k * (scn.resolution / 0.05), tick.pose_y, tick.pose_yaw + 0.01 * k, tick.speed,
                                  tick.path_x, tick.path_y, tick.path_yaw, tick.goal_x, tick.goal_y,
                                  goal_checker_xy_tolerance=tick.goal_checker_xy_tolerance))
            res = grp.optimize(ticks, us)
            nxt = []
            for i in range(n):
                ug, og = res[i]
                ut, ot = twins[i].optimize(ticks[i], us[i])
                uo, oo = oracles[i].optimize(ticks[i], us[i])
                # (a member may be scored by another kernel inside the group than alone — the split pass is
                # not used in groups, the batched launch partitions the grid differently — so sums associate
                # differently: last-ulp differences are allowed, nothing more)
                cm, ct = members[i].get_costs().astype(np.float64), twins[i].get_costs().astype(np.float64)
                if F.DETAIL and np.abs(ug - ut).max() > 5e-6:
                    print(f"  member {i} tick {k}: u grouped - alone {np.abs(ug - ut).max():.3g} at "
                          f"{np.unravel_index(np.abs(ug - ut).argmax(), ug.shape)}; grouped {ug[:, :3]} alone {ut[:, :3]} oracle {uo[:, :3]}\n"
                          f"    grouped: min {og.min_cost} sum_w {og.sum_w} nc {og.non_colliding} passes {og.passes} kind {og.pass_kind} fail {og.fail_flag}\n"
                          f"    alone:   min {ot.min_cost} sum_w {ot.sum_w} nc {ot.non_colliding} passes {ot.passes} kind {ot.pass_kind} fail {ot.fail_flag}\n"
                          f"    oracle:  min {oo.min_cost} sum_w {oo.sum_w} nc {oo.non_colliding} fail {oo.fail_flag}\n    draw {draws[i]}", flush=True)
                relc = np.abs(cm - ct) / np.maximum(np.abs(ct), 1.0)
                # (... and a handful of rollouts may read a neighbouring cell: two kernels, two last ulps)
                if np.abs(ug - ut).max() > 1e-4 or int(np.sum(relc > 2e-5)) > max(3, len(ct) // 2000):
                    raise F.Mismatch(f"tick {k} member {i}: grouped and alone differ (u {np.abs(ug - ut).max():.3g}, costs "
                                     f"{np.abs(members[i].get_costs() - twins[i].get_costs()).max():.3g}); B {draws[i]['B']} "
                                     f"critics {draws[i]['critics']} kernel {F.kernel_name(members[i])}")
                for f in ("fail_flag", "furthest_reached_path_point", "non_colliding"):
                    if getattr(og, f) != getattr(ot, f):
                        raise F.Mismatch(f"tick {k} member {i}: {f} {getattr(og, f)} grouped, {getattr(ot, f)} alone")
                try:
                    notes += F.check(f"{case}.{i}", k, draws[i], ug, og, uo, oo, members[i].get_costs(), oracles[i].get_costs())
                except F.Mismatch:
                    if F.DETAIL:
                        dd = draws[i]
                        print(f"  member {i} tick {k}: twist grouped {ug[:, 1]} alone {ut[:, 1]} oracle {uo[:, 1]}\n"
                              f"    grouped: min {og.min_cost} sum_w {og.sum_w} nc {og.non_colliding} passes {og.passes} kind {og.pass_kind}\n"
                              f"    alone:   min {ot.min_cost} sum_w {ot.sum_w} nc {ot.non_colliding} passes {ot.passes} kind {ot.pass_kind}\n"
                              f"    oracle:  min {oo.min_cost} sum_w {oo.sum_w} nc {oo.non_colliding}\n"
                              f"    max |u grouped - oracle| {np.abs(ug - uo).max():.3g} at {np.unravel_index(np.abs(ug - uo).argmax(), ug.shape)}; "
                              f"alone - oracle {np.abs(ut - uo).max():.3g}\n    draw {dd}", flush=True)
                    raise
                nxt.append(np.concatenate([uo[:, 1:], uo[:, -1:]], axis=1))
            us = nxt
    finally:
        grp.close()
        for x in members + twins + oracles:
            x.close()
    return n, T, lane, [d["B"] for d in draws], notes


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    only = [int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("only=")]
    cases = only or range(first, first + count)
    F.DETAIL = bool(only)
    bad = 0
    t0 = time.time()
    for case in cases:
        try:
            n, T, lane, Bs, notes = run(case)
            print(f"case {case}: ok  {n} members T {T}{' lane' if lane else ''} B {Bs} {notes if notes else ''} ({time.time() - t0:.0f} s)", flush=True)
        except Exception as e:
            msg = str(e).splitlines()[0] if str(e) else type(e).__name__
            if "more than 63 samples per trajectory" in msg:
                continue
            bad += 1
            print(f"case {case}: FAILED  {type(e).__name__}: {msg[:400]}", flush=True)
            if only:
                traceback.print_exc()
    print(f"{bad} of {len(list(cases))} cases failed", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
