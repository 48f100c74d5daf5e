#!/usr/bin/env python3
"""bench.py — rollouts/sec of one Optimizer::optimize() tick on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak]

A "step" is one computeVelocityCommands() tick's optimize() call through the
C-ABI (libsmpc.so): per-tick upload (control sequence, plan, tables), the
kernels, the exchange(s) when N > 1, and the read-back of the new control
sequence.  The noise tensors and the costmap are resident in HBM before the
timed region (the reference draws noise once per reset and reuses it,
src/noise_generator.cpp:26-42).

Workload (default, strong scaling): BASELINE.json configs[3] — 2 097 152 rollouts
x 64 steps in total on the 200x200 synthetic costmap, the batch the metric is
quoted on; N GPUs own 2 097 152 / N rollouts each (N = 1 runs the whole batch on
one GPU).  `--scaling weak` keeps 262 144 rollouts per GPU instead (N = 8 is then
the same job).  At N = 1 the other BASELINE configs are timed too and reported
under "other_configs", together with three variants of the headline workload:
the tick without furthest-point speculation, a closed loop with a moving pose,
and regenerate_noises = true.

At N > 1 the exchange is RCCL called from inside libsmpc (smpc_shard_tick: one
ncclAllGather of the shard tuples per tick on the ctx's stream).  The
collective-free mailbox exchange (smpc_shard_p2p_*) is timed next to it (by
default in the two-rank run only; SMPC_BENCH_ALTERNATIVES=1: at every N) and
reported under "exchange_alternatives"; SMPC_BENCH_EXCHANGE=mailbox|torch
makes another implementation the headline.  Every fall-through is recorded in
the JSON line (config.exchange_fallbacks), not only on stderr.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TOTAL_ROLLOUTS = 2097152       # BASELINE.json configs[3]
SHARD_ROLLOUTS = 262144        # its per-GPU share at 8 GPUs (the --scaling weak workload)
HORIZON = 64


def algorithmic_bytes(B, T, W, H, P):
    """noise read once, cost written+read once, compulsory
    costmap bytes, path, control sequence in/out."""
    return B * (12 * T + 8) + min(W * H, B * T) + 12 * P + 24 * T


def measured_traffic(B, T, costmap):
    """HBM bytes per launch of the scoring pass from the committed rocprofv3 PMC passes
    (profiles/<round>/traffic*.json: FETCH_SIZE with the gfx950 x2 correction calibrated on the
    furthest-only pass + WRITE_SIZE) when one exists for exactly this workload, else None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic*.json")), reverse=True):
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


DEPLOYED_CRITICS = ("constraint", "cost", "goal", "goal_angle", "path_align", "path_follow", "path_angle",
                    "prefer_forward", "twirling")      # robot_bringup/config/nav2_params.yaml:222


def critic_set(names):
    from mpcholonavigation_amd.tick import default_critics
    cr = default_critics()
    for n in ("obstacles", "path_align", "path_follow", "goal_angle", "prefer_forward", "cost", "goal",
              "constraint", "twirling", "path_angle", "velocity_deadband"):
        getattr(cr, n).enabled = 1 if n in names else 0
    return cr


def make_ctx(B, T, map_size, shard_offset=0, global_batch=0, seed=1234, flags=0, critics=None):
    from mpcholonavigation_amd.optimizer import Smpc
    from mpcholonavigation_amd.synthetic import make_scenario
    from mpcholonavigation_amd.tick import default_config, default_critics
    cfg = default_config(batch_size=B, time_steps=T, flags=flags,
                         shard_offset=shard_offset, global_batch_size=global_batch)
    scn = make_scenario(T, map_size=map_size)
    g = Smpc(cfg)
    g.set_critics(default_critics() if critics is None else critic_set(critics))
    g.set_costmap(scn.cells, scn.origin_x, scn.origin_y, scn.resolution,
                  inscribed_radius=scn.inscribed_radius,
                  cost_scaling_factor=scn.cost_scaling_factor,
                  inflation_radius=scn.inflation_radius)
    g.seed(seed)     # device RNG fills the stored noise tensors once (HBM resident)
    return g, scn, cfg


CLOCK_WARMUP_MS = 200.0     # --clock-warmup-ms: every timed context gets the same treatment


def raise_clocks(g, ms, seed=1234, scn=None):
    """Bring the device's clocks up before the warm-up ticks: `ms` milliseconds of the workload's own
    ticks (scn given), then smpc_reset and the seed again — the context is bit for bit where it was:
    stored noise, no furthest-point prediction, the caller's warm-started control sequence — so the W
    warm-up ticks and the K timed ones start from the same state as without it.  (Round 2 used the
    device RNG's redraw kernel for this; it is write-bound, and a 5 + 20 run behind it still read
    3-5 % slower than its own later ticks: SMPC_BENCH_CLOCK_WARMUP=redraw keeps it, and contexts
    without a scene — the multi-query group — use it.)  From idle the shader clock takes ~150 ms of load to climb from 2.0
    to 2.34 GHz (rocm-smi samples: tools/clock_watch.py), so the first ~60 ticks of a fresh
    process run 10-20 % slower than the rest (tools/ramp.py, tools/warm_ticks.py: ticks 5-24 take
    481 us cold, 419 us behind this; steady state ~410).  The metric is steady-state throughput:
    a run of 5 + 20 ticks should measure what a run of 100 + 200 measures.  W and K are untouched."""
    if ms <= 0:
        return
    import torch
    t0 = time.perf_counter()
    if scn is not None and os.environ.get("SMPC_BENCH_CLOCK_WARMUP", "ticks") == "ticks":
        u = scn.u0
        while (time.perf_counter() - t0) * 1e3 < ms:
            un, _ = g.optimize(scn.tick, u)
            u = shift(un)
        g.reset()
    else:
        while (time.perf_counter() - t0) * 1e3 < ms:
            g.redraw_noise()
            torch.cuda.synchronize()
    g.seed(seed)
    torch.cuda.synchronize()


def run_ticks(step_fn, scn, steps, warmup, sync, barrier, before_tick=None, after_tick=None, compiled=None):
    """W untimed ticks, then exactly K timed ones between barrier + synchronize on both sides.
    before_tick / after_tick (e.g. smpc_redraw_noise_async) run inside the timed region when given.
    compiled = (Smpc, sortham_run_ticks flags): the same W + K closed-loop ticks issued by the
    compiled loop of host/tick_loop.cpp (one C call for the K timed ticks) instead of this
    interpreter loop — the reference's caller is C++ (controller.cpp:80-116)."""
    if compiled is not None:
        from mpcholonavigation_amd import host_optimizer as H
        g, flags = compiled
        u = scn.u0
        if warmup:
            u, _ = H.run_ticks(g, scn.tick, u, warmup, flags)
        barrier()
        sync()
        t0 = time.perf_counter()
        u, outs = H.run_ticks(g, scn.tick, u, steps, flags)
        sync()
        barrier()
        t1 = time.perf_counter()
        return ((t1 - t0), sum(o.score_pass_ms for o in outs) / steps, sum(o.device_ms for o in outs) / steps,
                sum(o.passes for o in outs) / steps, outs[steps - 1])
    u = scn.u0
    for _ in range(warmup):
        if before_tick:
            before_tick()
        u_new, out = step_fn(scn.tick, u)
        if after_tick:
            after_tick()
        u = shift(u_new)
    barrier()
    sync()
    t0 = time.perf_counter()
    pass_ms = dev_ms = 0.0
    passes = 0
    for _ in range(steps):
        if before_tick:
            before_tick()
        u_new, out = step_fn(scn.tick, u)
        if after_tick:
            after_tick()
        u = shift(u_new)
        pass_ms += out.score_pass_ms
        dev_ms += out.device_ms
        passes += out.passes
    sync()
    barrier()
    t1 = time.perf_counter()
    return (t1 - t0), pass_ms / steps, dev_ms / steps, passes / steps, out


def last_pass_kernel(g):
    """The scoring-pass instance launched last, spelled as rocprofv3's kernel trace spells it."""
    import ctypes
    f = g.lib.smpc_debug_last_pass_kernel
    f.restype = ctypes.c_char_p
    f.argtypes = []
    return f().decode()


class MovingScene:
    """Closed loop: the pose advances by the emitted Twist every tick (holonomic integration
    over the controller period = model_dt) and the plan slides with it — a straight +x line on
    a fixed 0.05 m grid, pruned to start at the grid point nearest behind the robot and cut
    after `P` points, as PathHandler hands it over (src/path_handler.cpp:48-143).  Only the
    optimize() calls are timed (their durations are summed): building the next tick's inputs
    is the controller's job, not the optimizer's."""

    def __init__(self, scn, dt):
        from mpcholonavigation_amd.tick import Tick
        self.Tick = Tick
        self.scn = scn
        self.dt = dt
        t = scn.tick
        self.x, self.y, self.yaw = t.pose_x, t.pose_y, t.pose_yaw
        self.line_x0, self.line_y = float(t.path_x[0]), float(t.path_y[0])
        self.P = len(t.path_x)
        self.res = scn.resolution
        self.x_wrap = self.line_x0 + 0.35 * scn.cells.shape[1] * scn.resolution   # stay on the map
        self.speed = t.speed
        self.wraps = 0

    def tick(self):
        k0 = max(0, int(math.floor((self.x - self.line_x0) / self.res)))
        px = (self.line_x0 + self.res * (k0 + np.arange(self.P))).astype(np.float32)
        py = np.full(self.P, self.line_y, np.float32)
        return self.Tick(pose_x=self.x, pose_y=self.y, pose_yaw=self.yaw, speed=self.speed,
                         path_x=px, path_y=py, path_yaw=np.zeros(self.P, np.float32),
                         goal_x=float(px[-1]), goal_y=float(py[-1]))

    def advance(self, u_new):
        vx, vy, wz = (float(u_new[i, 1]) for i in range(3))    # Twist = u[offset = 1]
        c, s = math.cos(self.yaw), math.sin(self.yaw)
        self.x += (vx * c - vy * s) * self.dt
        self.y += (vx * s + vy * c) * self.dt
        self.yaw += wz * self.dt
        self.speed = (vx, vy, wz)
        if self.x > self.x_wrap:          # a new plan from the start (counts as what it is: a jump)
            self.x, self.y, self.yaw = self.scn.tick.pose_x, self.scn.tick.pose_y, self.scn.tick.pose_yaw
            self.wraps += 1


def run_moving(step_fn, scn, dt, steps, warmup, sync, barrier):
    mv = MovingScene(scn, dt)
    u = scn.u0
    el = 0.0
    passes = 0
    for k in range(warmup + steps):
        tk = mv.tick()
        if k >= warmup:
            barrier()
            sync()
            t0 = time.perf_counter()
        u_new, out = step_fn(tk, u)
        if k >= warmup:
            sync()
            el += time.perf_counter() - t0
            passes += out.passes
        mv.advance(u_new)
        u = shift(u_new)
    return el, passes / steps, mv


def time_config(B, T, map_size, steps, warmup, flags=0, redraw=False, critics=None):
    import torch
    g, scn, cfg = make_ctx(B, T, map_size, flags=flags, critics=critics)
    raise_clocks(g, CLOCK_WARMUP_MS, scn=scn)
    # regenerate_noises = true: the next epoch is requested behind every tick and drawn in the
    # background (smpc_redraw_noise_async: the reference's noise thread); the next tick waits for
    # it on the device, so in this back-to-back loop the whole draw is inside the timed region
    from mpcholonavigation_amd import host_optimizer as H
    loop = (g, H.TICKS_SHIFT | (H.TICKS_REDRAW_ASYNC if redraw else 0))
    el, _, _, passes, out = run_ticks(g.optimize, scn, steps, warmup,
                                      torch.cuda.synchronize, lambda: None, compiled=loop)
    g.set_profile(True)
    _, pass_ms, dev_ms, _, _ = run_ticks(g.optimize, scn, max(5, steps // 2), 2, torch.cuda.synchronize,
                                         lambda: None, compiled=(g, H.TICKS_SHIFT))
    P = len(scn.tick.path_x)
    by = algorithmic_bytes(B, T, map_size, map_size, P)
    tick_s = el / steps
    r = {
        "rollouts_per_s": B * steps / el,
        "ms_per_tick": 1e3 * tick_s,
        "score_pass_ms": pass_ms,
        "device_ms": dev_ms,
        "algorithmic_bytes": by,
        "roofline_frac_score_pass": (by / (pass_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if pass_ms else None,
        "roofline_frac_tick": (by / tick_s / 1e9) / HBM_PEAK_GBS,
        "pass_kind": "lane per rollout" if out.pass_kind == 1 else "wave per rollout",
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
    raise_clocks(ctxs[0][0], CLOCK_WARMUP_MS, seed=1234)
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
        "sample": f"{n} ticks of {B} rollouts x {T} steps (a bounded sample: rollouts/s of the CPU path does not "
                  f"depend on the batch size), {map_size}x{map_size} costmap, "
                  f"oracle built -O3 -mavx2 -mfma -ffast-math; host {cpu}, "
                  f"{os.cpu_count()} logical cores present",
    }


def _all_min(dist, torch, v):
    t = torch.tensor([int(v)], dtype=torch.int32, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def _exchange_agrees(g, scn, rank, HipShard, ShardedOptimizer, dist, torch):
    """One tick through smpc_shard_tick (whichever exchange the ctx is set up for) and one through
    the torch.distributed driver from the same state must give the same control sequence on every
    rank.  The ranks stay in lockstep: the outcome of the library tick is shared (all_reduce MIN)
    BEFORE anyone enters the reference tick's collectives, and so is the verdict, so that every
    rank keeps or drops the implementation together.  Returns (ok, reason)."""
    note = ""
    try:
        u_nat, o_nat = g.shard_tick(scn.tick, scn.u0, False)
        ok = 1
    except Exception as e:
        ok, note = 0, f"rank {rank}: shard_tick raised: {e}"
        print(f"[bench] {note}", file=sys.stderr, flush=True)
    if not _all_min(dist, torch, ok):
        return False, note or "shard_tick failed on another rank"
    agree = 0
    try:
        u_ref, o_ref = ShardedOptimizer(HipShard(g), speculate=False).optimize(scn.tick, scn.u0)
        agree = int(np.allclose(u_nat, u_ref, rtol=0, atol=1e-6) and
                    o_nat.furthest_reached_path_point == o_ref.furthest_reached_path_point)
        if not agree:
            note = f"rank {rank}: control sequence differs from the torch.distributed driver's"
    except Exception as e:
        note = f"rank {rank}: reference tick raised: {e}"
        print(f"[bench] {note}", file=sys.stderr, flush=True)
    g.set_stream(-1)      # SMPC_STREAM_OWN: back to the ctx's own stream
    if not _all_min(dist, torch, agree):
        return False, note or "disagreement on another rank"
    return True, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the first ~60 ticks of a fresh process run inside the device's power-management
    # transient (tools/ramp.py); 100 + 200 ticks are 0.13 s
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): --total-rollouts in all, split over the GPUs; "
                         "weak: --rollouts-per-gpu on every GPU")
    ap.add_argument("--total-rollouts", type=int, default=TOTAL_ROLLOUTS)
    ap.add_argument("--rollouts-per-gpu", type=int, default=SHARD_ROLLOUTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--clock-warmup-ms", type=float, default=200.0,
                    help="milliseconds of device-RNG redraws in front of the warm-up ticks, so that the shader "
                         "clock has climbed (2.0 -> 2.34 GHz over ~150 ms of load) before anything is timed; "
                         "0: off.  Reported in the JSON line")
    ap.add_argument("--no-speculate", action="store_true",
                    help="N > 1: always exchange the furthest point first (two collectives per tick) "
                         "instead of speculating on the previous tick's value and re-scoring on a miss")
    args = ap.parse_args()
    global CLOCK_WARMUP_MS
    CLOCK_WARMUP_MS = args.clock_warmup_ms

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
    # of the N > 1 code path with several processes on a one-GPU box (small --total-rollouts)
    if os.environ.get("SMPC_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # SMPC_BENCH_FORCE_DIST=1 drives the sharded path (RCCL collectives included) with a
    # single rank: a rehearsal of the N > 1 code path on a one-GPU box
    force_dist = os.environ.get("SMPC_BENCH_FORCE_DIST") == "1"
    sharded = world > 1 or force_dist
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("SMPC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    T, MAP = HORIZON, 200
    if args.scaling == "strong":
        if args.total_rollouts % world:
            raise SystemExit(f"--total-rollouts {args.total_rollouts} is not divisible by {world} GPUs")
        B = args.total_rollouts // world
    else:
        B = args.rollouts_per_gpu
    g, scn, cfg = make_ctx(B, T, MAP, shard_offset=rank * B, global_batch=world * B)
    P = len(scn.tick.path_x)

    exchange_impl = "none"
    fallbacks = []          # every implementation tried and dropped, with the reason
    alternatives = {}
    if sharded:
        from mpcholonavigation_amd.sharded import (HipShard, MailboxShardedOptimizer, NativeShardedOptimizer,
                                                   ShardedOptimizer)
        labels = {
            "rccl": ("RCCL called from libsmpc (smpc_shard_tick: ncclAllGather on the ctx's stream)"
                     if not os.environ.get("SMPC_RCCL_LIB") else
                     "NOT RCCL: the nccl* entry points of smpc_shard_tick come from SMPC_RCCL_LIB="
                     + os.environ["SMPC_RCCL_LIB"] + " (a rehearsal of the control flow, not a measurement)"),
            "mailbox": "mailboxes over IPC/xGMI, no collective (smpc_shard_p2p_*, smpc_shard_tick)",
            "torch": "RCCL through torch.distributed (ShardedOptimizer)",
        }

        def set_up(kind):
            """-> optimizer or None; collective; records the reason of a failure."""
            if kind == "torch":
                g.set_stream(-1)
                return ShardedOptimizer(HipShard(g), speculate=not args.no_speculate)
            try:
                cls = MailboxShardedOptimizer if kind == "mailbox" else NativeShardedOptimizer
                so = cls(g, speculate=not args.no_speculate)
                err = ""
            except Exception as e:
                so, err = None, f"set-up failed: {e}"
            if not _all_min(dist, torch, so is not None):
                fallbacks.append({"tried": kind, "reason": err or "set-up failed on another rank"})
                return None
            ok, why = _exchange_agrees(g, scn, rank, HipShard, ShardedOptimizer, dist, torch)
            g.reset()      # every rank back to the same state (no furthest-point hint), dropped or kept
            if not ok:
                fallbacks.append({"tried": kind, "reason": "cross-check against the torch.distributed driver: " + why})
                return None
            return so

        # SMPC_BENCH_EXCHANGE = rccl (default) | mailbox | torch picks the headline implementation;
        # each falls through to the next if it cannot be set up on every rank or fails the
        # cross-check.  north_star names the RCCL collective: it is the default.
        first = os.environ.get("SMPC_BENCH_EXCHANGE", "rccl")
        order = {"rccl": ["rccl", "torch"], "mailbox": ["mailbox", "rccl", "torch"], "torch": ["torch"]}[first]
        so = None
        for kind in order:
            so = set_up(kind)
            if so is not None:
                exchange_impl = labels[kind]
                headline_kind = kind
                break
        step_fn = so.optimize

        def barrier():
            dist.barrier()
    else:
        step_fn = g.optimize
        headline_kind = "single"

        def barrier():
            pass

    # who issues the timed ticks: the compiled loop of host/tick_loop.cpp (sortham_run_ticks: the
    # reference's caller is C++, controller.cpp:80-116) around smpc_optimize / smpc_shard_tick;
    # the torch.distributed driver of the exchange exists in Python only.  SMPC_BENCH_CALLER=python
    # times the interpreter loop instead (it is reported beside the headline either way).
    from mpcholonavigation_amd import host_optimizer as H

    def compiled_for(kind):
        if kind == "torch" or os.environ.get("SMPC_BENCH_CALLER") == "python":
            return None
        if kind == "single":
            return (g, H.TICKS_SHIFT)
        return (g, H.TICKS_SHIFT | (H.TICKS_SHARD if args.no_speculate else H.TICKS_SHARD_SPECULATE))
    loop = compiled_for(headline_kind)

    # the timed region: exactly K ticks, no event records in the stream
    if sharded and args.clock_warmup_ms > 0 and os.environ.get("SMPC_BENCH_CLOCK_WARMUP", "ticks") == "ticks":
        # N > 1: the same warm-up with the SHARDED tick (a collective: every rank follows rank 0's clock, in
        # blocks of 50 ticks), then reset + seed on every rank and the Python driver's state cleared
        flag = torch.zeros(1, device="cuda")
        u_w = scn.u0
        t_w = time.perf_counter()
        while True:
            for _ in range(50):
                un_w, _ = step_fn(scn.tick, u_w)
                u_w = shift(un_w)
            flag[0] = 1.0 if (time.perf_counter() - t_w) * 1e3 >= args.clock_warmup_ms else 0.0
            dist.broadcast(flag, src=0)
            if float(flag.item()) > 0.0:
                break
        g.reset()
        g.seed(1234)
        if hasattr(so, "hint"):
            so.hint = None
        torch.cuda.synchronize()
        barrier()
    else:
        raise_clocks(g, args.clock_warmup_ms, scn=None if sharded else scn)
    el, _, _, passes, out = run_ticks(step_fn, scn, args.steps, args.warmup,
                                      torch.cuda.synchronize, barrier, compiled=loop)
    # the same K ticks issued from this interpreter (Smpc.optimize + numpy shift per tick)
    el_py = run_ticks(step_fn, scn, args.steps, 2, torch.cuda.synchronize, barrier)[0] if loop else el
    # kernel duration for the roofline: the same K ticks again with HIP events around every
    # scoring-pass launch, on the stream it is launched on (SMPC_FLAG_PROFILE; the event
    # records cost ~15 us of queue time per tick, which is why they are not in the region above)
    g.set_profile(True)
    el_prof, pass_ms, dev_ms, _, _ = run_ticks(step_fn, scn, args.steps, 2,
                                               torch.cuda.synchronize, barrier, compiled=loop)
    g.set_profile(False)
    # the same workload with a moving pose (closed loop; the furthest point changes as the robot
    # advances, so the speculation of a frozen scene does not flatter the tick)
    el_mv, passes_mv, mv = run_moving(step_fn, scn, cfg.model_dt, args.steps, min(args.warmup, 5),
                                      torch.cuda.synchronize, barrier)
    if sharded:
        t = torch.tensor([el, pass_ms, el_mv, el_py], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, pass_ms, el_mv, el_py = float(t[0]), float(t[1]), float(t[2]), float(t[3])

    # N > 1: the exchange implementations that are not the headline, timed on the same workload.
    # By default only in the two-rank run: the mailbox exchange has never crossed xGMI (one-GPU
    # rehearsals only), and if it were to fault there it would take this process — and the line
    # of this N — with it; the N = 4 and N = 8 lines do not depend on it.  SMPC_BENCH_ALTERNATIVES=1
    # forces them at every N, =0 never.
    alt_env = os.environ.get("SMPC_BENCH_ALTERNATIVES", "")
    if world > 1 and (alt_env == "1" or (alt_env == "" and world == 2)):
        for kind in ("rccl", "mailbox", "torch"):
            if kind == headline_kind:
                continue
            n_fb = len(fallbacks)
            alt = set_up(kind)
            if alt is None:
                alternatives[kind] = {"error": fallbacks[-1]["reason"] if len(fallbacks) > n_fb else "unavailable"}
                del fallbacks[n_fb:]
                continue
            ok, el_a, p_a = 1, 0.0, 0.0
            try:
                el_a, _, _, p_a, _ = run_ticks(alt.optimize, scn, args.steps, args.warmup,
                                               torch.cuda.synchronize, barrier, compiled=compiled_for(kind))
            except Exception as e:
                ok = 0
                alternatives[kind] = {"error": f"rank {rank}: {e}"}
            if not _all_min(dist, torch, ok):
                alternatives.setdefault(kind, {"error": "failed on another rank"})
                g.reset()
                continue
            t = torch.tensor([el_a], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            alternatives[kind] = {"exchange_impl": labels[kind], "ms_per_step": 1e3 * float(t[0]) / args.steps,
                                  "rollouts_per_s": world * B * args.steps / float(t[0]),
                                  "scoring_passes_per_tick": p_a}
            g.reset()
        g.set_stream(-1)

    if rank == 0:
        by = algorithmic_bytes(B, T, MAP, MAP, P)
        achieved = by / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
        tick_s = el / args.steps
        # tick level, SURVEY §8(d): the GPU's algorithmic bytes of ONE scoring pass over the whole
        # tick's wall time (uploads, every pass incl. re-scores, reduction, exchange, read-back)
        achieved_tick = by / tick_s / 1e9
        traffic, traffic_src = measured_traffic(B, T, f"{MAP}x{MAP}")
        total = world * B
        line = {
            "metric": "rollouts/sec per computeVelocityCommands() tick",
            "value": total * cfg.iteration_count * args.steps / el,
            "unit": "rollouts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * tick_s,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[3]: {total} rollouts x {T} steps in total"
                             + (f" on one GPU" if world == 1 else f", {B} per GPU on {world} GPUs")
                             + (" (strong scaling: the total is fixed)" if args.scaling == "strong"
                                else " (weak scaling: the per-GPU share is fixed)")
                             + ", full critic stack, 200x200 costmap in LDS, stored noise (HBM resident), "
                               "iteration_count 1"),
                "total_rollouts": total, "rollouts_per_gpu": B, "horizon": T, "costmap": "200x200",
                "path_points": P,
                "critics": ["Obstacles", "PathAlign", "PathFollow", "GoalAngle", "PreferForward"],
                "exchange": ("none" if world == 1 else
                             ("all_reduce(max furthest) + all_gather(tuple)" if args.no_speculate else
                              "all_gather(tuple); furthest point speculated, re-scored on a miss")),
                "exchange_impl": exchange_impl,
                "exchange_fallbacks": fallbacks,
                "furthest_reached_path_point": int(out.furthest_reached_path_point),
                "scoring_passes_per_tick": passes,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": last_pass_kernel(g),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": by,
                "avg_launch_ms": pass_ms,
                "achieved_tick": achieved_tick,
                "frac_tick": achieved_tick / HBM_PEAK_GBS,
                "frac_note": "frac: algorithmic bytes of one launch / average duration of the scoring-pass "
                             "kernel; frac_tick: the same bytes / wall time of the whole tick (SURVEY 8(d))",
                "device_ms_per_tick": dev_ms,
                "timing": f"HIP events around each scoring-pass launch over {args.steps} further ticks "
                          f"({1e3 * el_prof / args.steps:.4f} ms/tick with the event records in the stream)",
            },
            "caller": {
                "timed_loop": ("compiled: sortham_run_ticks (host/tick_loop.cpp) issues the W + K ticks — "
                               + ("smpc_optimize" if headline_kind == "single" else "smpc_shard_tick")
                               + " + shiftControlSequence per tick, as the reference's C++ controller does"
                               if loop else "this interpreter: Smpc.optimize + numpy shift per tick"),
                "interpreted_ms_per_step": 1e3 * el_py / args.steps,
                "note": "interpreted_ms_per_step: the same K ticks issued from bench.py's Python loop "
                        "(a ctypes call and two numpy allocations between ticks)",
            },
            "clock_warmup": {
                "ms": args.clock_warmup_ms,
                "what": ("ticks of this workload in front of the W warm-up ticks, then smpc_reset and the seed again: the "
                         "context is back where it started (stored noise unchanged, no furthest-point prediction)"
                         if os.environ.get("SMPC_BENCH_CLOCK_WARMUP", "ticks") == "ticks" else
                         "device-RNG redraw kernels in front of the W warm-up ticks, then the seed again (the stored "
                         "noise is unchanged)")
                        + "; --clock-warmup-ms 0 turns it off.  From idle the shader clock climbs from 2.0 to 2.34 GHz over "
                          "~150 ms of load (tools/clock_watch.py): without this, ticks 5-24 of a fresh process take 481 us, "
                          "in steady state ~400 us (tools/warm_ticks.py)",
            },
            "moving_pose": {
                "ms_per_step": 1e3 * el_mv / args.steps,
                "rollouts_per_s": total * args.steps / el_mv,
                "scoring_passes_per_tick": passes_mv,
                "frac_tick": (by / (el_mv / args.steps) / 1e9) / HBM_PEAK_GBS,
                "note": "closed loop: pose advanced by the emitted Twist every tick, plan pruned to the "
                        f"robot; optimize() calls only ({mv.wraps} plan restarts)",
            },
        }
        if alternatives:
            line["exchange_alternatives"] = alternatives
        if world == 1 and not args.no_other_configs:
            from mpcholonavigation_amd import _abi as A
            k4 = max(20, args.steps // 4)
            line["other_configs"] = {
                "configs[1] 65536x64 200x200": time_config(65536, 64, 200, 2 * args.steps, 2 * args.warmup),
                "configs[2] 262144x128 2000x2000": time_config(262144, 128, 2000, k4, 5),
                "configs[3] per-GPU share at 8 GPUs 262144x64 200x200":
                    time_config(SHARD_ROLLOUTS, 64, MAP, 2 * args.steps, 2 * args.warmup),
                "configs[4] 8 queries x 16384x64 per GPU": time_multi_query(8, 16384, 64, MAP,
                                                                            2 * args.steps, 2 * args.warmup),
                f"{B}x64 without speculation (furthest-only pass + scoring pass every tick)":
                    time_config(B, T, MAP, k4, 3, flags=A.SMPC_FLAG_NO_SPECULATION),
                f"{B}x64 regenerate_noises=true (device RNG redraw behind every tick, inside the timed region)":
                    time_config(B, T, MAP, k4, 3, redraw=True),
                "deployed configuration 2000x56, the nine critics of nav2_params.yaml:222":
                    time_config(2000, 56, MAP, 4 * args.steps, 40, critics=DEPLOYED_CRITICS),
                f"the deployed nine critics and horizon at the metric's batch, {B}x56":
                    time_config(B, 56, MAP, k4, 3, critics=DEPLOYED_CRITICS),
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
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
