#!/usr/bin/env python3
"""Random interleavings of ba_ss_draw_next / readers / mutators / plain sweeps / forecasts on
an engine that serves bsts's loop from look-ahead batches, against an engine that runs one
round per call: everything compared must be equal, bit for bit (the long form of
tests/test_concurrency_gpu.py::test_interleavings_behind_the_look_ahead; the code is
tests/concurrency_lib.py).
usage: ss_la_stress.py [iterations per model [seed]]   (SS_KERNEL, SS_ONLY in the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from concurrency_lib import LA_MODELS, la_stress

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kernel = int(os.environ["SS_KERNEL"]) if os.environ.get("SS_KERNEL") else None
for name in LA_MODELS:
    if os.environ.get("SS_ONLY") and name == "local level":
        continue
    la_stress(name, iters, seed, kernel=kernel, verbose=True)
