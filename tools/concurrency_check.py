#!/usr/bin/env python3
"""Every path once alone and once beside a second engine that keeps the GPU busy with
asynchronous launches of its own: the chains must come out bit for bit the same -- what
else shares the machine may change the timing inside a kernel, never a draw.  (Round 4:
the state kernels' variance draws had a race that only showed under such load.)  The long
form of tests/test_concurrency_gpu.py::test_same_draws_alone_and_beside_a_busy_engine; the
code is tests/concurrency_lib.py.
usage: concurrency_check.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from concurrency_lib import alone_vs_loaded, families

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ok = True
for name, (make, step, extra) in families().items():
    worst = max(alone_vs_loaded(make, step, extra) for _ in range(reps))
    print("%-34s %s" % (name, "same draws alone and under load" if worst == 0 else "DIFFERS (%d arrays)" % worst), flush=True)
    ok &= worst == 0
print("ALL THE SAME" if ok else "SOME PATH DIFFERS")
sys.exit(0 if ok else 1)
