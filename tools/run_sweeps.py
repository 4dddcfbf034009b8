#!/usr/bin/env python3
"""Plain C2 workload driver for profilers: set-up, 200 burn-in sweeps, then
NLAUNCH launches of NSWEEP sweeps (env), nothing else."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import regression_data, spike_slab_prior
n, p, nsig, chains = 10000, 512, 16, 1024
X, y, _ = regression_data(n, p, nsig, seed=8675309)
eng = boom_amd.Engine(chains, seed=1)
eng.build_suf_from_xy(X, y)
s = eng.get_suf()
suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
prior = spike_slab_prior(suf, nsig)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
g0 = np.zeros(p, np.uint8); g0[0] = 1
eng.set_state(g0)
eng.sweep(200)
nl, ns = int(os.environ.get("NLAUNCH", "10")), int(os.environ.get("NSWEEP", "20"))
t0 = time.perf_counter()
for _ in range(nl):
    eng.sweep(ns, sync=False)
eng.sync()
dt = time.perf_counter() - t0
print("%d x %d sweeps: %.1f us per sweep-round, %.2f M sweeps/s" % (nl, ns, dt / (nl * ns) * 1e6, chains * nl * ns / dt / 1e6))
