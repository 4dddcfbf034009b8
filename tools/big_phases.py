#!/usr/bin/env python3
"""Diagnostic: where ssvs_big_kernel's master wavefront spends its cycles, by command
(and, printed by the kernel itself: the build's phases and the master's own sections)
(-DBA_BSTAMPS build: make -C boom_amd/csrc ../../tools/build/libboomamd_bstamps.so).
usage: big_phases.py [signals [chains]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BOOM_AMD_LIB"] = os.path.join(ROOT, "tools", "build", "libboomamd_bstamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import regression_data, spike_slab_prior
n, p = 10000, 512
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
X, y, _ = regression_data(n, p, nsig, seed=8675309)
eng = boom_amd.Engine(chains, seed=1)
eng.build_suf_from_xy(X, y)
s = eng.get_suf()
suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
prior = spike_slab_prior(suf, nsig)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
g0 = np.zeros(p, np.uint8); g0[0] = 1
eng.set_state(g0)
eng.sweep(100)
eng.reset_summaries()
nsw = 200
t0 = time.perf_counter(); eng.sweep(nsw); dt = time.perf_counter() - t0
sm = eng.get_summaries()
ph = sm["phase_cycles"]
sw = sm["sweeps"]
print("signals %d chains %d: %.1f us per sweep-round, kbar %.2f, accepts/sweep %.3f" % (nsig, chains, dt / nsw * 1e6, sm["k_sum"] / sw, sm["accepts"] / sw))
for nm, cyc, cnt in (("master (state machine, tail)", ph[0], sw), ("table-fill rounds (EVAL)", ph[1], ph[5]),
                     ("shuffle uniforms (UNIF)", ph[2], ph[6]), ("model builds (BUILD)", ph[3], ph[7])):
    print("  %-30s %10.0f cycles/sweep   %8.3f per sweep   %10.0f cycles each" % (nm, cyc / sw, cnt / sw, cyc / max(cnt, 1)))
