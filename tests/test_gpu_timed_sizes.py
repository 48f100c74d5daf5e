"""Oracle parity AT the sizes bench.py times (the timed kernels were only covered by
self-consistency properties there).  The CPU oracle needs ~2 s per 262 144-rollout tick and
~17 s for the 2 097 152-rollout one, so each case is one tick plus one shifted follow-up.

Noise: the bench's own — the device RNG's stored tensors (smpc_seed), read back and handed to
the oracle — so the kernel under test runs on exactly the tensors it is timed on.
"""
import numpy as np
import pytest

from mpcholonavigation_amd.synthetic import make_scenario
from mpcholonavigation_amd.tick import default_config
from tests.helpers import assert_parity, configure

pytestmark = pytest.mark.gpu


def _shift(u):
    return np.concatenate([u[:, 1:], u[:, -1:]], axis=1)


def _run(B, T, map_size, expect_lane, max_hard, label, double_sums=False):
    from mpcholonavigation_amd.optimizer import Smpc
    from oracle.loader import Oracle, build
    build()
    cfg = default_config(batch_size=B, time_steps=T)
    scn = make_scenario(T, map_size=map_size)
    g, o = Smpc(cfg), Oracle(cfg)
    configure(g, scn)
    g.seed(1234)                       # what bench.make_ctx does
    noise = g.get_noise()
    configure(o, scn, noise=noise)
    del noise
    u_g = u_o = scn.u0
    for k in range(2):                 # tick 0 takes the exact two-pass route, tick 1 speculates
        ug, og = g.optimize(scn.tick, u_g)
        if double_sums:
            # The reference sums the B weights and B weighted controls sequentially in float
            # (xt::sum, src/optimizer.cpp:385-391).  Over 2 million terms that sum carries its own
            # rounding noise of the order of the tolerance: show it (float oracle against the same
            # oracle accumulating in double) and hold the GPU — whose reduction is a tree of
            # wave, block and grid partials — to the sums' exact value.
            uf, _ = o.optimize(scn.tick, u_o)
            o.set_accumulate_double(True)
            uo, oo = o.optimize(scn.tick, u_o)
            o.set_accumulate_double(False)
            from tests.helpers import rel_err, twist
            print(f"[parity] {label} tick {k}: the float-summing oracle is {rel_err(twist(uf), twist(uo)):.2e} "
                  f"(vector-rel, Twist) off its own double-summing twin; the GPU {rel_err(twist(ug), twist(uo)):.2e}")
        else:
            uo, oo = o.optimize(scn.tick, u_o)
        if expect_lane is not None:
            assert og.pass_kind == (1 if expect_lane else 0), "not the kernel the bench times"
        assert og.non_colliding == oo.non_colliding or max_hard > 0
        assert_parity(ug, og, uo, oo, g.get_costs(), o.get_costs(), max_flips=max_hard,
                      label=f"{label} tick {k}")
        # both continue from the ORACLE's sequence, so a tolerated difference does not compound
        u_g = u_o = _shift(uo)
    g.close()


def test_per_gpu_share_262144x64_lane_pass():
    """configs[3]'s per-GPU share at 8 GPUs, and the size every round-1 profile was taken at:
    smpc_pass_lane<true,true> against the oracle."""
    _run(262144, 64, 200, True, 4, "262144x64 200x200")


def test_cfg3_full_size_262144x128():
    """BASELINE configs[2] at full size: 262 144 x 128 on the 2000 x 2000 map."""
    _run(262144, 128, 2000, True, 8, "262144x128 2000x2000")


def test_headline_2097152x64():
    """BASELINE configs[3] on one GPU — bench.py's headline workload — against the oracle."""
    _run(2097152, 64, 200, True, 24, "2097152x64 200x200", double_sums=True)
