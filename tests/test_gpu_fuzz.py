"""A seeded, time-boxed sweep of tools/fuzz_parity.py inside the -m gpu suite: random scenes (all three generators, depth
3-9, LDS-resident and not) x cameras (inside, outside, axis-aligned on cell boundaries; odd image sizes; spp on both sides of
the two-phase limit) x schedules (first frame, cost-ordered replay, a progressive split, a 2-4 rank partition, a 2-5 share
multi-device context) against the oracle, bit for bit."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_fuzz_parity_sweep():
    import fuzz_parity
    msgs = []
    n, bad = fuzz_parity.run(budget=45.0, seed=20261004, log=msgs.append)
    assert n >= 20, f"only {n} cases in 45 s"
    assert bad == 0, "\n".join(m for m in msgs if "MISMATCH" in m)
