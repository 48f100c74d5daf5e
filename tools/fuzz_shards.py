#!/usr/bin/env python3
"""Developer tool: randomised batch sharding on one GPU (the data path of the N-GPU tick, SURVEY
8(e)): the batch of a tools/fuzz_parity.py case cut into 2-5 shards of unequal size, each a context
with its slice of the noise; per tick the phases of sharded.ShardedOptimizer by hand — begin on
every shard, (two-pass: local furthest, MAX) | (speculating: the predicted index, a re-score on a
miss), score into the shard's slot, combine on EVERY shard (all must agree bit for bit), the
all-collide re-score — against the oracle on the whole batch.   tools/fuzz_shards.py FIRST COUNT [only=CASE]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import fuzz_parity as F
from mpcholonavigation_amd.optimizer import Smpc
from mpcholonavigation_amd.tick import Tick
from oracle.loader import Oracle
from tests.helpers import configure


def run(case):
    r = np.random.default_rng(13 * case + 1)
    d = F.draw(200000 + case)
    d["iters"], d["footprint"], d["rng"] = 1, "", False
    d["flags"] = 0
    d["B"] = max(d["B"], 8)
    cfg, scn, tick, u0, cr, noise = F.build(d)
    B = cfg.batch_size
    G = int(r.integers(2, 6))
    cuts = sorted(set(int(x) for x in r.integers(1, B, size=G - 1)))
    bounds = [0] + cuts + [B]
    speculate = bool(r.random() < 0.6)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    shards = []
    pass_env = d["env_pass"] if d["env_pass"] in ("lane",) else ""
    if pass_env:
        os.environ["SMPC_PASS"] = pass_env
    try:
        for a, b in zip(bounds[:-1], bounds[1:]):
            c = F.default_config(batch_size=b - a, time_steps=cfg.time_steps, iteration_count=1, motion_model=cfg.motion_model,
                                 model_dt=cfg.model_dt, temperature=cfg.temperature, gamma=cfg.gamma, flags=0,
                                 vy_max=cfg.vy_max, vy_std=cfg.vy_std, shard_offset=a, global_batch_size=B)
            g = Smpc(c)
            configure(g, scn, critics=cr, noise=[n[a:b] for n in noise], track_unknown=d["track_unknown"])
            g.set_stream(stream)
            shards.append(g)
    finally:
        os.environ.pop("SMPC_PASS", None)
    o = Oracle(cfg)
    configure(o, scn, critics=cr, noise=noise, track_unknown=d["track_unknown"])
    L = shards[0].tuple_len
    t_all = torch.zeros(len(shards) * L, dtype=torch.float32, device=dev)
    t_f = torch.zeros(len(shards), dtype=torch.float32, device=dev)
    notes = []
    try:
        u = u0
        hint = None
        for k in range(3):
            tk = Tick(tick.pose_x + 0.02 * k * (scn.resolution / 0.05), tick.pose_y, tick.pose_yaw + 0.01 * k, tick.speed,
                      tick.path_x, tick.path_y, tick.path_yaw, tick.goal_x, tick.goal_y,
                      goal_checker_xy_tolerance=tick.goal_checker_xy_tolerance)
            for g in shards:
                g.shard_begin(tk, u)

            def score(dF, S):
                for i, g in enumerate(shards):
                    g.shard_score(dF, S, t_all[i * L:].data_ptr())

            def combine():
                res = [g.shard_combine(t_all.data_ptr(), len(shards)) for g in shards]
                for i, (ui, oi) in enumerate(res[1:], 1):
                    if not np.array_equal(ui, res[0][0]) or oi.furthest_reached_path_point != res[0][1].furthest_reached_path_point \
                            or oi.non_colliding != res[0][1].non_colliding or oi.fail_flag != res[0][1].fail_flag:
                        raise F.Mismatch(f"tick {k}: shard {i} combined something else than shard 0")
                return res[0]

            if speculate and hint is not None:
                preds = [g.shard_predicted_furthest() for g in shards]
                if any(p != preds[0] for p in preds):
                    raise F.Mismatch(f"tick {k}: the shards predict different furthest points {preds}")
                S = preds[0] if preds[0] is not None else hint
                score(0, S)
                ug, og = combine()
                if og.furthest_valid and og.furthest_reached_path_point != S:
                    notes.append(f"tick {k}: miss")
                    score(0, int(og.furthest_reached_path_point))
                    ug, og = combine()
            else:
                for i, g in enumerate(shards):
                    g.shard_furthest(t_f[i:].data_ptr())
                fmax = t_f.max().reshape(1).clone()
                score(fmax.data_ptr(), 0)
                ug, og = combine()
            if og.furthest_valid:
                hint = int(og.furthest_reached_path_point)
            if og.fail_flag and not tk.fail_flag_in:
                for i, g in enumerate(shards):
                    g.shard_rescore_failed(t_all[i * L:].data_ptr())
                ug, og2 = combine()
                og2.fail_flag = 1
                og = og2
            uo, oo = o.optimize(tk, u)
            cg = np.concatenate([g.get_costs() for g in shards])
            notes += F.check(case, k, d, ug, og, uo, oo, cg, o.get_costs())
            u = np.concatenate([uo[:, 1:], uo[:, -1:]], axis=1)
    finally:
        torch.cuda.synchronize()
        for g in shards:
            g.close()
        o.close()
    return d, bounds, speculate, notes


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    only = [int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("only=")]
    cases = only or range(first, first + count)
    F.DETAIL = False
    bad = 0
    t0 = time.time()
    for case in cases:
        try:
            d, bounds, spec, notes = run(case)
            print(f"case {case}: ok  B {d['B']} T {d['T']} shards {np.diff(bounds).tolist()} {'speculating' if spec else 'two-pass'} "
                  f"critics {d['critic_kind']} {notes if notes else ''} ({time.time() - t0:.0f} s)", flush=True)
        except Exception as e:
            msg = str(e).splitlines()[0] if str(e) else type(e).__name__
            if "more than 63 samples per trajectory" in msg:
                continue
            bad += 1
            print(f"case {case}: FAILED  {type(e).__name__}: {msg[:400]}\n    draw: {F.draw(200000 + case)}", flush=True)
            if only:
                traceback.print_exc()
    print(f"{bad} of {len(list(cases))} cases failed", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
