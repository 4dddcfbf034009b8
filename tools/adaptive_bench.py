#!/usr/bin/env python3
"""Diagnostic timing of AdaptiveSpikeSlabRegressionSampler (what lm.spike uses for
p > 100) on the BASELINE config-2 workload: n=1e4, p=512, 1024 chains."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import regression_data, spike_slab_prior
n, p, nsig, chains = 10000, 512, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
X, y, _ = regression_data(n, p, nsig, seed=8675309)
eng = boom_amd.Engine(chains, seed=1)
eng.build_suf_from_xy(X, y)
s = eng.get_suf()
suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
prior = spike_slab_prior(suf, nsig)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
g0 = np.zeros(p, np.uint8); g0[0] = 1
eng.set_state(g0)
eng.adaptive_sweep(300)
eng.reset_summaries()
nsw = 500
t0 = time.perf_counter(); eng.adaptive_sweep(nsw); dt = time.perf_counter() - t0
sm = eng.get_summaries()
gam, _, _ = eng.get_states()
print("adaptive sampler C2: %.1f us per sweep-round, %.3g sweeps/s, kbar %.2f, accepted moves/sweep %.2f of 100, signals in %.3f, nulls in %.5f"
      % (dt / nsw * 1e6, chains * nsw / dt, sm["k_sum"] / sm["sweeps"], sm["accepts"] / sm["sweeps"],
         gam[:, :nsig].mean(), gam[:, nsig:].mean()))
