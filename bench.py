#!/usr/bin/env python3
"""bench.py — rollouts/sec of one Optimizer::optimize() tick on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one computeVelocityCommands() tick's optimize() call through the
C-ABI (libsmpc.so): per-tick upload (control sequence, plan, tables), the
kernels, the exchange(s) when N > 1, and the read-back of the new control
sequence.  The noise tensors and the costmap are resident in HBM before the
timed region (the reference draws noise once per reset and reuses it,
src/noise_generator.cpp:26-42).

Workload (weak scaling): every GPU owns 262 144 rollouts x 64 steps on the
200x200 synthetic costmap — BASELINE.json configs[3] (2 097 152 x 64 on 8 GPUs)
is exactly the N=8 run; N=1 is its per-GPU shard.  configs[1] (65 536 x 64) and
configs[2] (262 144 x 128, 2000x2000 map) are timed too at N=1 and reported
under "other_configs".

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SHARD_ROLLOUTS = 262144
HORIZON = 64


def algorithmic_bytes(B, T, W, H, P):
    """noise read once, cost written+read once, compulsory
    costmap bytes, path, control sequence in/out."""
    return B * (12 * T + 8) + min(W * H, B * T) + 12 * P + 24 * T


def measured_traffic(B, T, costmap):
    """HBM bytes per launch of the scoring pass from the committed rocprofv3 PMC passes
    (profiles/<round>/traffic.json: FETCH_SIZE with the gfx950 x2 correction calibrated on the
    furthest-only pass + WRITE_SIZE) when one exists for exactly this workload, else None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            w = t["workload"]
            if (w["rollouts"], w["horizon"], w["costmap"]) == (B, T, costmap):
                return float(t["traffic_bytes_per_launch"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def shift(u):
    """Optimizer::shiftControlSequence (src/optimizer.cpp:206-225)."""
    return np.concatenate([u[:, 1:], u[:, -1:]], axis=1)


def make_ctx(B, T, map_size, shard_offset=0, global_batch=0, seed=1234, flags=0):
    from mpcholonavigation_amd import _abi as A
    from mpcholonavigation_amd.optimizer import Smpc
    from mpcholonavigation_amd.synthetic import make_scenario
    from mpcholonavigation_amd.tick import default_config, default_critics
    cfg = default_config(batch_size=B, time_steps=T, flags=flags,
                         shard_offset=shard_offset, global_batch_size=global_batch)
    scn = make_scenario(T, map_size=map_size)
    g = Smpc(cfg)
    g.set_critics(default_critics())
    g.set_costmap(scn.cells, scn.origin_x, scn.origin_y, scn.resolution,
                  inscribed_radius=scn.inscribed_radius,
                  cost_scaling_factor=scn.cost_scaling_factor,
                  inflation_radius=scn.inflation_radius)
    g.seed(seed)     # device RNG fills the stored noise tensors once (HBM resident)
    return g, scn, cfg


def run_ticks(step_fn, scn, steps, warmup, sync, barrier):
    u = scn.u0
    for _ in range(warmup):
        u_new, out = step_fn(scn.tick, u)
        u = shift(u_new)
    barrier()
    sync()
    t0 = time.perf_counter()
    pass_ms = dev_ms = 0.0
    passes = 0
    for _ in range(steps):
        u_new, out = step_fn(scn.tick, u)
        u = shift(u_new)
        pass_ms += out.score_pass_ms
        dev_ms += out.device_ms
        passes += out.passes
    sync()
    barrier()
    t1 = time.perf_counter()
    return (t1 - t0), pass_ms / steps, dev_ms / steps, passes / steps, out


def time_config(B, T, map_size, steps, warmup):
    import torch
    g, scn, cfg = make_ctx(B, T, map_size)
    el, _, _, passes, out = run_ticks(g.optimize, scn, steps, warmup,
                                      torch.cuda.synchronize, lambda: None)
    g.set_profile(True)
    _, pass_ms, dev_ms, _, _ = run_ticks(g.optimize, scn, steps, 2, torch.cuda.synchronize,
                                         lambda: None)
    P = len(scn.tick.path_x)
    by = algorithmic_bytes(B, T, map_size, map_size, P)
    r = {
        "rollouts_per_s": B * steps / el,
        "ms_per_tick": 1e3 * el / steps,
        "score_pass_ms": pass_ms,
        "device_ms": dev_ms,
        "algorithmic_bytes": by,
        "roofline_frac_score_pass": (by / (pass_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if pass_ms else None,
        "furthest": int(out.furthest_reached_path_point),
        "passes_per_tick": passes,
    }
    g.close()
    return r


def time_multi_query(n_ctx, B, T, map_size, steps, warmup):
    """BASELINE configs[4] per GPU: n_ctx independent planning instances (replicas only: no
    collective) ticked together by smpc_group_optimize — one upload, one scoring launch with the
    instance as second grid dimension, one reduction launch."""
    from mpcholonavigation_amd import _abi as A
    from mpcholonavigation_amd.optimizer import SmpcGroup
    ctxs = [make_ctx(B, T, map_size, seed=1234 + i, flags=A.SMPC_FLAG_LANE_PER_ROLLOUT)
            for i in range(n_ctx)]
    grp = SmpcGroup([g for g, _, _ in ctxs])
    ticks = [scn.tick for _, scn, _ in ctxs]
    us = [scn.u0 for _, scn, _ in ctxs]
    for _ in range(warmup):
        us = [shift(u) for u, _ in grp.optimize(ticks, us)]
    t0 = time.perf_counter()
    for _ in range(steps):
        res = grp.optimize(ticks, us)
        us = [shift(u) for u, _ in res]
    el = time.perf_counter() - t0
    passes = sum(o.passes for _, o in res) / n_ctx
    grp.close()
    for g, _, _ in ctxs:
        g.close()
    return {"queries": n_ctx, "rollouts_per_query": B, "rollouts_per_s": n_ctx * B * steps / el,
            "ms_per_round": 1e3 * el / steps, "passes_per_query": passes,
            "note": "replicas only: independent contexts, no exchange; one batched launch per round "
                    "(smpc_group_optimize)"}


def cpu_baseline_all_cores(T, map_size, threads, budget_s=8.0, B=65536):
    """The same CPU restatement on `threads` host threads: the batch sharded over them exactly as
    over GPUs (furthest point -> max, per-shard tuples -> combine; the oracle's shard entry
    points, ctypes releases the GIL).  The reference itself is single-threaded; this is what an
    OpenMP build of it could at best reach on this host."""
    from concurrent.futures import ThreadPoolExecutor
    from mpcholonavigation_amd.synthetic import make_noise, make_scenario
    from mpcholonavigation_amd.tick import default_config, default_critics
    from oracle.loader import Oracle
    scn = make_scenario(T, map_size=map_size)
    noise = make_noise(B, T)
    per = B // threads
    shards = []
    for k in range(threads):
        cfg = default_config(batch_size=per, time_steps=T, shard_offset=k * per, global_batch_size=per * threads)
        o = Oracle(cfg, fast=True)
        o.set_critics(default_critics())
        o.set_costmap(scn.cells, scn.origin_x, scn.origin_y, scn.resolution,
                      inscribed_radius=scn.inscribed_radius, cost_scaling_factor=scn.cost_scaling_factor,
                      inflation_radius=scn.inflation_radius)
        o.set_noise(*[n[k * per:(k + 1) * per] for n in noise])
        shards.append(o)
    pool = ThreadPoolExecutor(threads)

    hint = [None]

    def tick(u):
        # the protocol of ShardedOptimizer: speculate on the previous tick's furthest point,
        # re-score on a miss (one rollout per tick in the steady state)
        if hint[0] is None:
            hint[0] = int(max(pool.map(lambda o: o.shard_furthest(scn.tick, u), shards)))
        tuples = np.stack(list(pool.map(lambda o: o.shard_score(scn.tick, u, hint[0]), shards)))
        u_new, out = shards[0].shard_combine(tuples)
        if out.furthest_valid and int(out.furthest_reached_path_point) != hint[0]:
            hint[0] = int(out.furthest_reached_path_point)
            tuples = np.stack(list(pool.map(lambda o: o.shard_score(scn.tick, u, hint[0]), shards)))
            u_new, out = shards[0].shard_combine(tuples)
        return u_new, out

    u = scn.u0
    u_new, _ = tick(u)
    u = shift(u_new)
    n, t0 = 0, time.perf_counter()
    while True:
        u_new, _ = tick(u)
        u = shift(u_new)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 400:
            break
    el = time.perf_counter() - t0
    pool.shutdown()
    return {"value": per * threads * n / el, "unit": "rollouts/s", "cores": threads,
            "sample": f"{n} ticks of {per * threads} rollouts x {T} steps sharded over {threads} threads "
                      "(furthest point speculated, re-scored on a miss)"}


def cpu_baseline(T, map_size, budget_s=12.0, B=65536, max_ticks=200):
    """The CPU restatement built with the reference's flags, one thread
    (the reference is single-threaded: CMakeLists.txt:7-8), bounded sample."""
    from mpcholonavigation_amd.synthetic import make_noise, make_scenario
    from mpcholonavigation_amd.tick import default_config, default_critics
    from oracle.loader import Oracle, build
    build()
    cfg = default_config(batch_size=B, time_steps=T)
    scn = make_scenario(T, map_size=map_size)
    o = Oracle(cfg, fast=True)
    o.set_critics(default_critics())
    o.set_costmap(scn.cells, scn.origin_x, scn.origin_y, scn.resolution,
                  inscribed_radius=scn.inscribed_radius,
                  cost_scaling_factor=scn.cost_scaling_factor,
                  inflation_radius=scn.inflation_radius)
    o.set_noise(*make_noise(B, T))
    u = scn.u0
    u_new, _ = o.optimize(scn.tick, u)     # warm-up tick
    u = shift(u_new)
    n = 0
    t0 = time.perf_counter()
    while True:
        u_new, _ = o.optimize(scn.tick, u)
        u = shift(u_new)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= max_ticks:
            break
    el = time.perf_counter() - t0
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": B * n / el, "unit": "rollouts/s", "cores": 1, "kind": "port",
        "sample": f"{n} ticks of {B} rollouts x {T} steps, {map_size}x{map_size} costmap, "
                  f"oracle built -O3 -mavx2 -mfma -ffast-math; host {cpu}, "
                  f"{os.cpu_count()} logical cores present",
    }


def _exchange_agrees(g, scn, rank, HipShard, ShardedOptimizer, dist, torch):
    """One tick through smpc_shard_tick (whichever exchange the ctx is set up for) and one through
    the torch.distributed driver from the same state must give the same control sequence on every
    rank; the verdict is shared, so that every rank keeps or drops the implementation together."""
    agree = 0
    try:
        u_nat, o_nat = g.shard_tick(scn.tick, scn.u0, False)
        u_ref, o_ref = ShardedOptimizer(HipShard(g), speculate=False).optimize(scn.tick, scn.u0)
        agree = int(np.allclose(u_nat, u_ref, rtol=0, atol=1e-6) and
                    o_nat.furthest_reached_path_point == o_ref.furthest_reached_path_point)
    except Exception as e:
        print(f"[bench] rank {rank}: exchange self-check raised: {e}", file=sys.stderr, flush=True)
    g.set_stream(-1)      # SMPC_STREAM_OWN: back to the ctx's own stream
    t = torch.tensor([agree], dtype=torch.int32, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rollouts-per-gpu", type=int, default=SHARD_ROLLOUTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-speculate", action="store_true",
                    help="N > 1: always exchange the furthest point first (two collectives per tick) "
                         "instead of speculating on the previous tick's value and re-scoring on a miss")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    # SMPC_BENCH_SHARE_GPU=1 (with SMPC_BENCH_BACKEND=gloo): every rank on device 0 — a rehearsal
    # of the N > 1 code path with several processes on a one-GPU box (small --rollouts-per-gpu)
    if os.environ.get("SMPC_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # SMPC_BENCH_FORCE_DIST=1 drives the sharded path (RCCL collectives included) with a
    # single rank: a rehearsal of the N > 1 code path on a one-GPU box
    force_dist = os.environ.get("SMPC_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("SMPC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    B, T, MAP = args.rollouts_per_gpu, HORIZON, 200
    g, scn, cfg = make_ctx(B, T, MAP, shard_offset=rank * B, global_batch=world * B)
    P = len(scn.tick.path_x)

    exchange_impl = "none"
    if world > 1 or force_dist:
        from mpcholonavigation_amd.sharded import (HipShard, MailboxShardedOptimizer, NativeShardedOptimizer,
                                                   ShardedOptimizer)
        so = None
        # SMPC_BENCH_EXCHANGE = mailbox | rccl | torch picks the first implementation tried; each
        # falls through to the next if it cannot be set up on every rank or fails the cross-check
        first = os.environ.get("SMPC_BENCH_EXCHANGE", "torch" if os.environ.get("SMPC_BENCH_TORCH_EXCHANGE") == "1"
                               else "mailbox")
        candidates = {"mailbox": ["mailbox", "rccl"], "rccl": ["rccl"], "torch": []}[first]
        for kind in candidates:
            try:
                if kind == "mailbox":
                    # no collective: tuples written into the peers' mailboxes over xGMI (smpc_shard_p2p_*)
                    so = MailboxShardedOptimizer(g, speculate=not args.no_speculate)
                    exchange_impl = "mailboxes over IPC/xGMI, no collective (smpc_shard_p2p_*, smpc_shard_tick)"
                else:
                    # exchanges inside libsmpc: ncclAllGather on the ctx's stream between the kernels
                    so = NativeShardedOptimizer(g, speculate=not args.no_speculate)
                    exchange_impl = "RCCL called from libsmpc (smpc_shard_tick)"
            except Exception as e:
                print(f"[bench] {kind} exchange unavailable ({e})", file=sys.stderr, flush=True)
                so = None
                continue
            if _exchange_agrees(g, scn, rank, HipShard, ShardedOptimizer, dist, torch):
                break
            print(f"[bench] {kind} exchange disagrees with the torch.distributed driver", file=sys.stderr,
                  flush=True)
            so = None
        if so is None:
            so = ShardedOptimizer(HipShard(g), speculate=not args.no_speculate)
            exchange_impl = "RCCL through torch.distributed (ShardedOptimizer)"
        step_fn = so.optimize

        def barrier():
            dist.barrier()
    else:
        step_fn = g.optimize

        def barrier():
            pass

    # the timed region: exactly K ticks, no event records in the stream
    el, _, _, passes, out = run_ticks(step_fn, scn, args.steps, args.warmup,
                                      torch.cuda.synchronize, barrier)
    # kernel duration for the roofline: the same K ticks again with HIP events around every
    # scoring-pass launch, on the stream it is launched on (SMPC_FLAG_PROFILE; the event
    # records cost ~15 us of queue time per tick, which is why they are not in the region above)
    g.set_profile(True)
    el_prof, pass_ms, dev_ms, _, _ = run_ticks(step_fn, scn, args.steps, 2,
                                               torch.cuda.synchronize, barrier)
    if world > 1 or force_dist:
        t = torch.tensor([el, pass_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, pass_ms = float(t[0]), float(t[1])

    if rank == 0:
        by = algorithmic_bytes(B, T, MAP, MAP, P)
        achieved = by / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(B, T, f"{MAP}x{MAP}")
        line = {
            "metric": "rollouts/sec per computeVelocityCommands() tick",
            "value": world * B * cfg.iteration_count * args.steps / el,
            "unit": "rollouts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[3] weak-scaled: {B} rollouts x {T} steps per GPU "
                            f"({world * B} total), full critic stack, 200x200 costmap in LDS, "
                            "stored noise (HBM resident), iteration_count 1",
                "rollouts_per_gpu": B, "horizon": T, "costmap": "200x200", "path_points": P,
                "critics": ["Obstacles", "PathAlign", "PathFollow", "GoalAngle", "PreferForward"],
                "exchange": ("none" if world == 1 else
                             ("all_reduce(max furthest) + all_gather(tuple)" if args.no_speculate else
                              "all_gather(tuple); furthest point speculated, re-scored on a miss")),
                "exchange_impl": exchange_impl,
                "furthest_reached_path_point": int(out.furthest_reached_path_point),
                "scoring_passes_per_tick": passes,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "smpc_pass_lane<true,true>" if out.pass_kind == 1 else "smpc_pass<1,0,true>",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": by,
                "avg_launch_ms": pass_ms,
                "device_ms_per_tick": dev_ms,
                "timing": f"HIP events around each smpc_pass launch over {args.steps} further ticks "
                          f"({1e3 * el_prof / args.steps:.4f} ms/tick with the event records in the stream)",
            },
        }
        if world == 1 and not args.no_other_configs:
            line["other_configs"] = {
                "configs[1] 65536x64 200x200": time_config(65536, 64, 200, args.steps, args.warmup),
                "configs[2] 262144x128 2000x2000": time_config(262144, 128, 2000,
                                                                max(20, args.steps // 4),
                                                                max(5, args.warmup // 4)),
                "configs[4] 8 queries x 16384x64 per GPU": time_multi_query(8, 16384, 64, MAP,
                                                                            args.steps, args.warmup),
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(T, MAP)
            line["cpu_baseline"]["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
            nthr = max(1, min(32, (os.cpu_count() or 2) // 2))
            line["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(T, MAP, nthr)
            # configs[0], the reference's own CPU-runnable case (1000 x 30), on both sides
            c0 = cpu_baseline(30, MAP, budget_s=3.0, B=1000, max_ticks=2000)
            g0 = time_config(1000, 30, MAP, 400, 40)
            line["cpu_baseline"]["configs[0] 1000x30"] = {
                "cpu_rollouts_per_s": c0["value"], "cpu_ms_per_tick": 1e3 * 1000 / c0["value"],
                "gpu_rollouts_per_s": g0["rollouts_per_s"], "gpu_ms_per_tick": g0["ms_per_tick"]}
        print(json.dumps(line), flush=True)
    g.close()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
