"""The long-run tools behind the suite at a bounded budget (VERDICT r5 task 6d): rare branches
(substream overruns, hull capacities, capacity escalation, look-ahead rewinds under random
interleavings) only show up over many sweeps; tools/soak.py and tools/ss_la_stress.py are the
open-ended forms, these are about a minute of GPU together.  Every call must return without a
chain error; the interleavings must be equal bit for bit (tests/concurrency_lib.py)."""
import os
import subprocess
import sys

import pytest

from concurrency_lib import LA_MODELS, la_stress

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_soak_at_a_quarter_scale():
    """25 000 bsts rounds, 1 500 structural rounds, 750 logit and probit rounds, all chains"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "0.25"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for what in ("bsts local level", "trend + 12 seasons", "logit,", "probit,"):
        assert what in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("name", list(LA_MODELS))
def test_look_ahead_interleavings_second_seed(name):
    """another 300 random steps per model on a seed the concurrency suite does not use"""
    la_stress(name, 300, seed=20261003)
