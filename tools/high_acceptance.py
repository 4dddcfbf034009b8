#!/usr/bin/env python3
"""Diagnostic: a posterior with weak signals (variables come and go three times as
often as on the C2 workload, mean model size 24) -- adaptive walking vs
WALK=0 (ba_set_tuning walk_policy: batch mode only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, boom_amd
from cases import regression_data, spike_slab_prior
n, p, nsig, chains = 2000, 512, 24, 1024
X, y, _ = regression_data(n, p, nsig, seed=5, noise_sd=6.0)   # weak signals: many variables come and go
eng = boom_amd.Engine(chains, seed=1)
eng.set_tuning(walk_policy=int(os.environ.get("WALK", "-1")))
eng.build_suf_from_xy(X, y)
s = eng.get_suf()
suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
prior = spike_slab_prior(suf, 30)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
g0 = np.zeros(p, np.uint8); g0[0] = 1
eng.set_state(g0); eng.sweep(200); eng.reset_summaries()
t0 = time.perf_counter(); eng.sweep(400); dt = time.perf_counter() - t0
sm = eng.get_summaries()
print("high-acceptance case: %.1f us per sweep-round, kbar %.1f, accepts/sweep %.2f, slot hits %.0f %%" % (dt / 400 * 1e6, sm["k_sum"] / sm["sweeps"], sm["accepts"] / sm["sweeps"], 100 * sm["slot_hits"] / max(sm["accepts"], 1)))
