#!/usr/bin/env python3
"""Diagnostic: share of accepted flips served from a chain's other (table, model
block) slot on the C2 workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, boom_amd
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
eng.set_state(g0); eng.sweep(200); eng.reset_summaries(); eng.sweep(1000)
sm = eng.get_summaries()
print("accepts %d, slot hits %d (%.1f %%)" % (sm["accepts"], sm["slot_hits"], 100 * sm["slot_hits"] / sm["accepts"]))
