"""Batch-sharded tick over torch.distributed.

One process per GPU; rank g owns rows [g*B/G, (g+1)*B/G) of the rollout batch.
The control sequence, path and costmap are replicated.  Per tick there are at
most two exchanges, both tiny (latency-bound, not link-bound):

  1. MAX over ranks of the local furthest reached path point (one float) —
     PathAlign / PathFollow consume the batch-wide value
     (reference tools/utils.hpp:292-319);
  2. all-gather of the shard tuples {min cost, sum w, furthest, non-colliding,
     sum w*c[3T]}; every rank then combines them with the softmax's shift
     invariance (rescale by exp(-(min_g - min)/temperature)) — the reference's
     updateControlSequence (src/optimizer.cpp:382-393) on the whole batch.

`NativeShardedOptimizer` runs the same protocol inside libsmpc with RCCL called from
C++ (one library call per tick); `ShardedOptimizer` drives it phase by phase from Python
through torch.distributed and is what the gloo CPU tests exercise.

With `speculate=True` exchange 1 is skipped: each rank scores with the furthest
point of the previous tick; the all-gathered tuples carry the true value, and on
a miss every rank re-scores with it (exact result either way).

The driver is backend-agnostic: `HipShard` feeds libsmpc device pointers; the
CPU tests plug in an oracle-backed backend with gloo.  torch is plumbing here
(tensors for the collectives, streams); the arithmetic is in the library.
"""
import numpy as np
import torch
import torch.distributed as dist


class HipShard:
    """Adapter: Smpc context <-> torch CUDA tensors (device pointers)."""

    def __init__(self, smpc):
        self.smpc = smpc
        self.tuple_len = smpc.tuple_len
        self.device = torch.device("cuda", torch.cuda.current_device())
        smpc.set_stream(torch.cuda.current_stream().cuda_stream)

    def begin(self, tick, u):
        self.smpc.shard_begin(tick, u)

    def predicted_furthest(self):
        return self.smpc.shard_predicted_furthest()

    def furthest(self, t_furthest):
        self.smpc.shard_furthest(t_furthest.data_ptr())

    def score(self, t_furthest, hint, t_tuple):
        self.smpc.shard_score(t_furthest.data_ptr() if t_furthest is not None else 0, hint,
                              t_tuple.data_ptr())

    def rescore_failed(self, t_tuple):
        self.smpc.shard_rescore_failed(t_tuple.data_ptr())

    def combine(self, t_tuples, n):
        return self.smpc.shard_combine(t_tuples.data_ptr(), n)


class NativeShardedOptimizer:
    """The same tick with the exchanges inside libsmpc (smpc_shard_tick: ncclAllGather /
    ncclAllReduce on the ctx's stream between the kernels; include/smpc.h).  torch.distributed
    only ships the RCCL unique id to the ranks once.  Raises if RCCL cannot be set up; the
    caller then falls back to ShardedOptimizer."""

    def __init__(self, smpc, group=None, speculate=False):
        self.smpc = smpc
        self.speculate = speculate
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        # every rank checks that it can load RCCL BEFORE anyone enters the collective init
        try:
            my_id, ok = smpc.shard_comm_id(), 1
        except Exception:
            my_id, ok = None, 0
        if world > 1:
            t = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            ok = int(t.item())
        if not ok:
            raise RuntimeError("RCCL is not loadable from libsmpc on every rank")
        ids = [my_id if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0, group=group)
        smpc.shard_comm_init(ids[0], rank, world)
        self.G = world

    def optimize(self, tick, u):
        return self.smpc.shard_tick(tick, u, self.speculate)


class MailboxShardedOptimizer:
    """The same tick with no collective at all (smpc_shard_p2p_*: every rank's finishing kernel
    writes its tuple into its peers' mailboxes over xGMI and waits for theirs).
    torch.distributed only ships the IPC handles of the mailboxes once.  Raises if the
    mailboxes cannot be set up on every rank; the caller then falls back."""

    def __init__(self, smpc, group=None, speculate=False):
        self.smpc = smpc
        self.speculate = speculate
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        try:
            mine, ok = smpc.shard_p2p_handle(), 1
        except Exception:
            mine, ok = b"\0" * 64, 0
        handles = [None] * world
        if world > 1:
            dist.all_gather_object(handles, (ok, mine), group=group)
        else:
            handles = [(ok, mine)]
        if not all(o for o, _ in handles):
            raise RuntimeError("the shard mailbox could not be created on every rank")
        err = None
        try:
            smpc.shard_p2p_init([h for _, h in handles], rank, world)
        except Exception as e:      # every rank must learn of it before anyone ticks
            err = str(e)
        errs = [None] * world
        if world > 1:
            dist.all_gather_object(errs, err, group=group)
        else:
            errs = [err]
        if any(errs):
            raise RuntimeError("the shard mailboxes could not be mapped: " + "; ".join(e for e in errs if e))
        self.G = world

    def optimize(self, tick, u):
        return self.smpc.shard_tick(tick, u, self.speculate)


class ShardedOptimizer:
    """Optimizer::optimize() over a batch sharded across the ranks of `group`."""

    def __init__(self, backend, group=None, speculate=False):
        self.b = backend
        self.group = group
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = backend.device
        L = backend.tuple_len
        self.t_furthest = torch.zeros(1, dtype=torch.float32, device=dev)
        self.t_tuple = torch.zeros(L, dtype=torch.float32, device=dev)
        self.t_all = torch.zeros(self.G * L, dtype=torch.float32, device=dev)
        self.speculate = speculate
        self.hint = None          # furthest point of the previous tick
        self.rescored = 0         # speculation misses so far

    def _gather(self):
        if dist.is_initialized():
            dist.all_gather_into_tensor(self.t_all, self.t_tuple, group=self.group)
        else:
            self.t_all.copy_(self.t_tuple)

    def optimize(self, tick, u):
        """One tick: returns (u_new [3,T], SmpcTickOut) — identical on every rank."""
        b = self.b
        b.begin(tick, u)
        if self.speculate and self.hint is not None:
            # the library's prediction of this tick's index (previous value carried forward by the
            # robot's motion and the plan's pruning: the same on every rank) where the backend has one
            predict = getattr(b, "predicted_furthest", None)
            p = predict() if predict is not None else None
            if p is not None:
                self.hint = p
            b.score(None, self.hint, self.t_tuple)
            self._gather()
            u_new, out = b.combine(self.t_all, self.G)
            if out.furthest_valid and out.furthest_reached_path_point != self.hint:
                # miss: the gathered tuples carry the true batch-wide furthest point
                self.rescored += 1
                self.hint = int(out.furthest_reached_path_point)
                b.score(None, self.hint, self.t_tuple)
                self._gather()
                u_new, out = b.combine(self.t_all, self.G)
        else:
            b.furthest(self.t_furthest)
            if dist.is_initialized():
                dist.all_reduce(self.t_furthest, op=dist.ReduceOp.MAX, group=self.group)
            b.score(self.t_furthest, 0, self.t_tuple)
            self._gather()
            u_new, out = b.combine(self.t_all, self.G)
            if out.furthest_valid:
                self.hint = int(out.furthest_reached_path_point)
        if out.fail_flag and not tick.fail_flag_in:
            # all rollouts of the WHOLE batch collide: the reference scored nothing
            # past Obstacles (critic_manager.cpp:70-73)
            b.rescore_failed(self.t_tuple)
            self._gather()
            u_new, out2 = b.combine(self.t_all, self.G)
            out2.fail_flag = 1
            out = out2
        return u_new, out
