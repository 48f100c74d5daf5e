"""The batch-sharded tick (mpcholonavigation_amd/sharded.py) with world_size 2 over
gloo on the CPU.  Each rank's shard arithmetic comes from the oracle's shard
phases (tests may use the oracle; the driver itself never does), so this
checks the driver: the two exchanges, the tuple combine, the speculation /
re-score protocol and the batch-wide fail flag — against the unsharded oracle.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpcholonavigation_amd import _abi as A
from mpcholonavigation_amd.synthetic import make_noise, make_scenario
from mpcholonavigation_amd.tick import Tick, default_config, default_critics

WORLD = 2


class OracleShard:
    """Shard backend over the CPU oracle (host tensors instead of device pointers)."""

    def __init__(self, oracle):
        self.o = oracle
        self.tuple_len = A.SMPC_TUPLE_HEADER + 3 * oracle.T
        self.device = torch.device("cpu")
        self.tick = self.u = None

    def begin(self, tick, u):
        self.tick, self.u = tick, np.ascontiguousarray(u, np.float32)

    def furthest(self, t_furthest):
        t_furthest[0] = self.o.shard_furthest(self.tick, self.u)

    def score(self, t_furthest, hint, t_tuple):
        S = int(t_furthest[0]) if t_furthest is not None else int(hint)
        t_tuple.copy_(torch.from_numpy(self.o.shard_score(self.tick, self.u, S)))

    def rescore_failed(self, t_tuple):
        t_tuple.copy_(torch.from_numpy(self.o.shard_rescore_failed(self.tick, self.u)))

    def combine(self, t_tuples, n):
        return self.o.shard_combine(t_tuples.numpy().reshape(n, self.tuple_len))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(B, T, all_lethal=False):
    from oracle.loader import Oracle
    scn = make_scenario(T, all_lethal=all_lethal)
    noise = make_noise(B, T)

    def mk(cfg, rows):
        o = Oracle(cfg)
        o.set_critics(default_critics())
        o.set_costmap(scn.cells, scn.origin_x, scn.origin_y, scn.resolution)
        o.set_noise(*[n[rows] for n in noise])
        return o
    return scn, mk


def _worker(rank, port, B, T, all_lethal, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from mpcholonavigation_amd.sharded import ShardedOptimizer
        scn, mk = _setup(B, T, all_lethal)
        Bs = B // WORLD
        whole = mk(default_config(batch_size=B, time_steps=T), slice(0, B))
        whole.set_accumulate_double(True)
        shard = mk(default_config(batch_size=Bs, time_steps=T, shard_offset=rank * Bs,
                                  global_batch_size=B), slice(rank * Bs, (rank + 1) * Bs))
        res = {}
        for spec in (False, True):
            so = ShardedOptimizer(OracleShard(shard), speculate=spec)
            u_s = u_w = scn.u0
            t = scn.tick
            errs, rescored = [], 0
            for k in range(4):
                if k == 2:   # change the plan spacing: the speculated furthest point misses
                    px = (t.pose_x + 0.08 * np.arange(len(t.path_x))).astype(np.float32)
                    t = Tick(t.pose_x, t.pose_y, t.pose_yaw, t.speed, px, t.path_y, t.path_yaw,
                             float(px[-1]), t.goal_y)
                u_s, out_s = so.optimize(t, u_s)
                u_w, out_w = whole.optimize(t, u_w)
                assert out_s.fail_flag == out_w.fail_flag
                assert out_s.non_colliding == out_w.non_colliding
                if out_w.furthest_valid:
                    assert out_s.furthest_reached_path_point == out_w.furthest_reached_path_point
                errs.append(float(np.max(np.abs(u_s - u_w)) / np.max(np.abs(u_w))))
                u_w = u_s.copy()
            res[spec] = (max(errs), so.rescored, out_s.fail_flag)
        # every rank ends with the same control sequence
        t_u = torch.from_numpy(u_s.copy())
        gathered = [torch.zeros_like(t_u) for _ in range(WORLD)]
        dist.all_gather(gathered, t_u)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        q.put((rank, res, same))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("all_lethal", [False, True])
def test_sharded_tick_world2_gloo(oracle_lib, all_lethal):
    B, T = 512, 56
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, B, T, all_lethal, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, same in out:
        assert same, "ranks disagree on the new control sequence"
        for spec, (err, rescored, fail) in res.items():
            assert err < 2e-5, (rank, spec, err)
            assert fail == (1 if all_lethal else 0)
            if spec and not all_lethal:
                assert rescored >= 1        # the plan change was caught and re-scored
            if not spec:
                assert rescored == 0


def test_shard_tuples_combine_to_the_unsharded_update(oracle_lib):
    """Softmax shift invariance: combining G shard tuples equals the whole-batch update,
    for uneven shard sizes too (single process)."""
    from oracle.loader import Oracle
    B, T = 600, 40
    scn, mk = _setup(B, T)
    whole = mk(default_config(batch_size=B, time_steps=T), slice(0, B))
    whole.set_accumulate_double(True)
    u_w, out_w = whole.optimize(scn.tick, scn.u0)
    cuts = [0, 100, 101, 350, 600]
    shards = [mk(default_config(batch_size=b - a, time_steps=T, shard_offset=a, global_batch_size=B),
                 slice(a, b)) for a, b in zip(cuts[:-1], cuts[1:])]
    S = max(s.shard_furthest(scn.tick, scn.u0) for s in shards)
    assert int(S) == out_w.furthest_reached_path_point
    tuples = np.stack([s.shard_score(scn.tick, scn.u0, int(S)) for s in shards])
    u_c, out_c = shards[0].shard_combine(tuples)
    assert out_c.non_colliding == out_w.non_colliding
    assert abs(out_c.min_cost - out_w.min_cost) < 1e-6
    assert np.max(np.abs(u_c - u_w)) / np.max(np.abs(u_w)) < 1e-5


def test_sharded_device_rng_is_a_slice_of_the_global_stream(oracle_lib):
    """Counter-based noise: shard g draws exactly rows [offset, offset+B_g) of the global batch."""
    from oracle.loader import Oracle
    B, T = 96, 17          # odd T: pairs of the Box-Muller block straddle rows
    whole = Oracle(default_config(batch_size=B, time_steps=T))
    whole.seed(99)
    full = whole.get_noise()
    for a, b in ((0, 31), (31, 64), (64, 96)):
        sh = Oracle(default_config(batch_size=b - a, time_steps=T, shard_offset=a,
                                   global_batch_size=B))
        sh.seed(99)
        for x, y in zip(sh.get_noise(), full):
            assert np.array_equal(x, y[a:b])
