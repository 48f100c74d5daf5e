"""A bounded, seeded slice of the randomised sweeps under tools/ (fuzz_parity.py, fuzz_group.py,
fuzz_shards.py, fuzz_host.py, fuzz_state.py) as part of the GPU suite: random configurations, critic lists with
perturbed parameters, scenes (resolution, origin, odd map sizes, unknown cells, plans that curve,
robots at the map's edge), three closed-loop ticks each, library against oracle.  The full sweeps
(tens of thousands of cases, minutes on one GPU) found three defects that the hand-written cases had
not: see DESIGN.md §7.  The case numbers are the tools' own, so a failure here is re-run alone with
`python tools/fuzz_parity.py 0 0 only=CASE`."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def _tool(name):
    if TOOLS not in sys.path:
        sys.path.insert(0, TOOLS)
    return __import__(name)


def _sweep(run, cases):
    bad = []
    for case in cases:
        try:
            run(case)
        except Exception as e:      # noqa: BLE001 — every failure is collected and shown
            msg = str(e).splitlines()[0] if str(e) else type(e).__name__
            if "more than 63 samples per trajectory" in msg:      # a documented refusal
                continue
            bad.append(f"case {case}: {type(e).__name__}: {msg[:300]}")
    assert not bad, "\n".join(bad)


def test_randomised_single_context_sweep():
    F = _tool("fuzz_parity")
    _sweep(F.run, list(range(0, 250)) + list(range(20000, 20250)) + list(range(40000, 40250)))


def test_randomised_groups():
    G = _tool("fuzz_group")
    _sweep(G.run, range(0, 60))


def test_randomised_shards():
    S = _tool("fuzz_shards")
    _sweep(S.run, range(0, 200))


def test_randomised_host_closed_loops():
    H = _tool("fuzz_host")
    _sweep(H.run, range(0, 120))


def test_randomised_call_sequences():
    """tools/fuzz_state.py: one long-lived context, 24 calls each — ticks with a moving pose, new
    plans, costmaps and costmap regions, critic parameters, speed limits, reset, stored noise, the
    device RNG and its epochs in the foreground and behind a tick."""
    S = _tool("fuzz_state")
    _sweep(lambda case: S.run(case), range(0, 100))
