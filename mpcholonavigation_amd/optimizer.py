"""Python face of libsmpc.so (include/smpc.h): the hot path of
sortham::Optimizer::optimize() (reference src/optimizer.cpp:157-164) on MI355X.

There is no fallback: if the HIP library is missing or no GPU is present, the
constructor raises.  Names follow the reference: `optimize` is
Optimizer::optimize, `get_generated_trajectories` is
Optimizer::getGeneratedTrajectories, `reset` is Optimizer::reset's
device half.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMPC_LIB: developer override (kernel experiments build variants of the library side by side)
LIB_PATH = os.environ.get("SMPC_LIB") or os.path.join(_HERE, "libsmpc.so")
_lib = None


class SmpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"smpc error {code}: {msg}")
        self.code = code


def load_library():
    """dlopen libsmpc.so and bind every prototype of include/smpc.h.
    Raises if the library was not built (no silent fallback)."""
    global _lib
    if _lib is None:
        # libsmpc and PyTorch-ROCm both link libamdhip64.so.7 and the first copy loaded serves
        # the whole process.  torch only finds its GPUs with its own bundled runtime, while
        # libsmpc runs on either, so when torch is installed it goes first.
        if "torch" not in sys.modules and not os.environ.get("SMPC_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C mpcholonavigation_amd/csrc` "
                "(or __graft_entry__.build()); the sampling-MPC path has no CPU fallback")
        _lib = A.bind(C.CDLL(LIB_PATH))
        if _lib.smpc_abi_version() != A.SMPC_ABI_VERSION:
            raise ImportError("libsmpc.so ABI version mismatch")
    return _lib


def _ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Smpc:
    """One optimizer context on one GPU (smpc_ctx)."""

    def __init__(self, cfg: A.SmpcConfig):
        self.lib = load_library()
        self.cfg = cfg
        self.B, self.T = cfg.batch_size, cfg.time_steps
        h = C.c_void_p()
        rc = self.lib.smpc_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise SmpcError(rc, self.lib.smpc_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.smpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise SmpcError(rc, self.lib.smpc_last_error(self.h).decode())

    # ---- configuration ---------------------------------------------------
    def reset(self):
        self._ck(self.lib.smpc_reset(self.h))

    def set_constraints(self, vx_max, vx_min, vy_max, wz_max):
        self._ck(self.lib.smpc_set_constraints(self.h, vx_max, vx_min, vy_max, wz_max))

    def set_critics(self, p: A.SmpcCriticParams):
        self._ck(self.lib.smpc_set_critics(self.h, C.byref(p)))

    def set_costmap(self, cells, origin_x, origin_y, resolution, track_unknown=False,
                    inscribed_radius=0.1, cost_scaling_factor=10.0, inflation_radius=0.55):
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        if cells.ndim != 2:
            raise ValueError("cells must be [height, width]")
        h, w = cells.shape
        self._ck(self.lib.smpc_set_costmap(
            self.h, _ptr(cells), w, h, origin_x, origin_y, resolution, int(track_unknown),
            inscribed_radius, cost_scaling_factor, inflation_radius))

    def update_costmap_region(self, cells, x0, y0, width, height):
        """Hand over the window [y0:y0+height, x0:x0+width] of the caller's full map `cells`."""
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        if cells.ndim != 2:
            raise ValueError("cells must be the full [height, width] map")
        first = cells[y0:, x0:]          # a view: the pointer of the window's first cell
        self._ck(self.lib.smpc_update_costmap_region(
            self.h, C.c_void_p(first.ctypes.data), cells.shape[1], x0, y0, width, height))

    def costmap_upload_bytes(self):
        """(bytes uploaded by the last set_costmap / update_costmap_region, total so far)."""
        last, total = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.lib.smpc_costmap_upload_bytes(self.h, C.byref(last), C.byref(total)))
        return last.value, total.value

    def set_footprint(self, xy, circumscribed_radius, layer_cost_scaling_factor=10.0):
        """Robot footprint [n, 2] (robot frame) for consider_footprint=true."""
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        self._ck(self.lib.smpc_set_footprint(self.h, xy.ctypes.data_as(C.c_void_p), len(xy),
                                             float(circumscribed_radius), float(layer_cost_scaling_factor)))

    def set_noise(self, nvx, nvy, nwz):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (nvx, nvy, nwz)]
        for x in a:
            if x.shape != (self.B, self.T):
                raise ValueError(f"noise must be [{self.B}, {self.T}]")
        self._ck(self.lib.smpc_set_noise(self.h, _ptr(a[0]), _ptr(a[1]), _ptr(a[2])))

    def seed(self, seed):
        self._ck(self.lib.smpc_seed(self.h, seed))

    def redraw_noise(self):
        self._ck(self.lib.smpc_redraw_noise(self.h))

    def redraw_noise_async(self):
        """The next epoch drawn in the background; the next tick takes it (smpc_redraw_noise_async)."""
        self._ck(self.lib.smpc_redraw_noise_async(self.h))

    def get_noise(self):
        out = [np.empty((self.B, self.T), np.float32) for _ in range(3)]
        self._ck(self.lib.smpc_get_noise(self.h, _ptr(out[0]), _ptr(out[1]), _ptr(out[2])))
        return out

    # ---- hot path ----------------------------------------------------------
    def optimize(self, tick, u):
        """u: float32 [3, T] (vx, vy, wz) -> (u_new, SmpcTickOut)."""
        u = np.array(u, dtype=np.float32, order="C")      # a copy: the call updates it in place
        if u.shape != (3, self.T):
            raise ValueError(f"u must be [3, {self.T}]")
        out = A.SmpcTickOut()
        rc = self.lib.smpc_optimize(self.h, C.byref(tick.c), u.ctypes.data, C.byref(out))
        if rc != 0:
            self._ck(rc)
        return u, out

    def get_generated_trajectories(self):
        out = [np.empty((self.B, self.T), np.float32) for _ in range(3)]
        self._ck(self.lib.smpc_get_trajectories(self.h, _ptr(out[0]), _ptr(out[1]),
                                                _ptr(out[2])))
        return out

    def get_costs(self):
        c = np.empty(self.B, np.float32)
        self._ck(self.lib.smpc_get_costs(self.h, _ptr(c)))
        return c

    def selftest_sincos(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        s, c = np.empty_like(x), np.empty_like(x)
        self._ck(self.lib.smpc_selftest_sincos(self.h, _ptr(x), x.size, _ptr(s), _ptr(c)))
        return s, c

    def selftest_lane_reduce(self, v, w):
        """out[t] = sum_b w[b] v[b, t] through the lane pass's in-register transpose-reduce."""
        v = np.ascontiguousarray(v, dtype=np.float32).reshape(64, 64)
        w = np.ascontiguousarray(w, dtype=np.float32).reshape(64)
        out = np.empty(64, np.float32)
        self._ck(self.lib.smpc_selftest_lane_reduce(self.h, _ptr(v), _ptr(w), _ptr(out)))
        return out

    # ---- batch-sharded phases (device pointers are plain ints) ------------------
    def selftest_row_reduce(self, v):
        """smpc_split.hip's N x N transpose-reduce per group of N lanes: v [64 lanes][N] -> [64], N = 16 or 32."""
        v = np.ascontiguousarray(v, dtype=np.float32)
        out = np.empty(64, np.float32)
        f = self.lib.smpc_selftest_row_reduce
        f.restype = C.c_int
        f.argtypes = [A._ctx, C.c_void_p, C.c_uint32, C.c_void_p]
        self._ck(f(self.h, _ptr(v), v.shape[1], _ptr(out)))
        return out

    def set_stream(self, hip_stream):
        self._ck(self.lib.smpc_set_stream(self.h, C.c_void_p(hip_stream)))

    def set_profile(self, enable):
        """SMPC_FLAG_PROFILE on/off: HIP events around the scoring pass (costs ~15 us/tick)."""
        self._ck(self.lib.smpc_set_profile(self.h, int(bool(enable))))

    @property
    def tuple_len(self):
        return self.lib.smpc_tuple_len(self.h)

    def shard_begin(self, tick, u):
        u = np.ascontiguousarray(u, dtype=np.float32)
        self._ck(self.lib.smpc_shard_begin(self.h, C.byref(tick.c), _ptr(u)))

    def shard_predicted_furthest(self):
        """The index this tick is expected to have (after shard_begin), or None."""
        h = C.c_uint32(0)
        return int(h.value) if self.lib.smpc_shard_predicted_furthest(self.h, C.byref(h)) else None

    def shard_furthest(self, d_furthest):
        self._ck(self.lib.smpc_shard_furthest(self.h, C.c_void_p(d_furthest)))

    def shard_score(self, d_furthest, furthest_hint, d_tuple):
        self._ck(self.lib.smpc_shard_score(
            self.h, C.c_void_p(d_furthest) if d_furthest else None, int(furthest_hint),
            C.c_void_p(d_tuple)))

    def shard_rescore_failed(self, d_tuple):
        self._ck(self.lib.smpc_shard_rescore_failed(self.h, C.c_void_p(d_tuple)))

    # ---- the sharded tick with the exchanges inside the library (RCCL) ------------
    def shard_comm_id(self):
        """A fresh RCCL unique id (bytes); rank 0 makes it and ships it to every rank."""
        buf = (C.c_ubyte * A.SMPC_COMM_ID_BYTES)()
        rc = self.lib.smpc_shard_comm_id(buf, A.SMPC_COMM_ID_BYTES)
        if rc != 0:
            raise SmpcError(rc, self.lib.smpc_last_error(None).decode())
        return bytes(buf)

    def shard_comm_init(self, comm_id, rank, world):
        buf = (C.c_ubyte * A.SMPC_COMM_ID_BYTES).from_buffer_copy(comm_id)
        self._ck(self.lib.smpc_shard_comm_init(self.h, buf, int(rank), int(world)))

    def shard_p2p_handle(self):
        """IPC handle (bytes) of this context's mailbox for the collective-free exchange."""
        buf = C.create_string_buffer(64)
        self._ck(self.lib.smpc_shard_p2p_handle(self.h, buf, 64))
        return buf.raw

    def shard_p2p_init(self, handles, rank, world):
        """handles: the mailbox handles of all ranks, in rank order."""
        if len(handles) != world or any(len(h) != 64 for h in handles):
            raise ValueError("one 64-byte handle per rank")
        blob = b"".join(handles)
        self._ck(self.lib.smpc_shard_p2p_init(self.h, C.c_char_p(blob), rank, world))

    def shard_p2p_set_timeout(self, milliseconds):
        """Wall-clock bound of the in-kernel wait for the peers' tuples (default 10 000 ms)."""
        self._ck(self.lib.smpc_shard_p2p_set_timeout(self.h, int(milliseconds)))

    def shard_tick(self, tick, u, speculate=True):
        """One batch-sharded tick, ncclAllGather / ncclAllReduce included (collective)."""
        u = np.ascontiguousarray(u, dtype=np.float32).copy()
        out = A.SmpcTickOut()
        self._ck(self.lib.smpc_shard_tick(self.h, C.byref(tick.c), _ptr(u), C.byref(out),
                                          1 if speculate else 0))
        return u, out

    def shard_combine(self, d_tuples, n_tuples):
        u = np.zeros((3, self.T), np.float32)
        out = A.SmpcTickOut()
        self._ck(self.lib.smpc_shard_combine(self.h, C.c_void_p(d_tuples), n_tuples, _ptr(u),
                                             C.byref(out)))
        return u, out


class SmpcGroup:
    """Several Smpc contexts ticked with one launch (include/smpc.h: smpc_group_*)."""

    def __init__(self, members):
        self.members = list(members)
        self.lib = self.members[0].lib
        n = len(self.members)
        arr = (A._ctx * n)(*[m.h for m in self.members])
        h = A._ctx()
        rc = self.lib.smpc_group_create(arr, n, C.byref(h))
        if rc != 0:
            raise SmpcError(rc, self.lib.smpc_last_error(None).decode())
        self.h = h

    def optimize(self, ticks, us):
        """ticks, us: one per member; returns [(u_new, SmpcTickOut)] in member order."""
        n = len(self.members)
        if getattr(self, "_ins", None) is None or len(ticks) != n:
            if len(ticks) != n:
                raise ValueError("one tick per member")
            self._ins = (A.SmpcTickIn * n)()
            self._src = [None] * n
            self._bufs = np.empty((n, 3, self.members[0].T), np.float32)
            self._ptrs = (C.c_void_p * n)(*[self._bufs[i].ctypes.data for i in range(n)])
            self._outs = (A.SmpcTickOut * n)()
        for i, t in enumerate(ticks):
            # Tick.c is rebuilt whenever a field of the tick was assigned: copy it again when
            # it is not the struct object this slot was filled from (an identity check per member)
            c = t.c
            if self._src[i] is not c:
                self._ins[i] = c
                self._src[i] = c
        for i, u in enumerate(us):
            self._bufs[i] = u
        rc = self.lib.smpc_group_optimize(self.h, self._ins, self._ptrs, self._outs)
        if rc != 0:
            msg = next((m.lib.smpc_last_error(m.h).decode() for m in self.members
                        if m.lib.smpc_last_error(m.h)), "")
            raise SmpcError(rc, msg)
        res = self._bufs.copy()
        return [(res[i], A.SmpcTickOut.from_buffer_copy(self._outs[i])) for i in range(n)]

    def close(self):
        if self.h:
            self.lib.smpc_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
