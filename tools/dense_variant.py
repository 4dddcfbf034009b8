#!/usr/bin/env python3
"""SURVEY 8d's dense-posterior variant of config 2 (n=1e4, p=512, NSIG true
signals, 1024 chains): models sit at or above the LDS kernel's 64-variable
limit, so the HBM-resident kernel carries most chains.  Diagnostic timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
t0 = time.perf_counter(); eng.sweep(100); print("burn-in 100 sweeps %.2fs" % (time.perf_counter() - t0))
eng.reset_summaries()
nsw = int(os.environ.get("NSWEEP", "200"))
t0 = time.perf_counter(); eng.sweep(nsw); dt = time.perf_counter() - t0
gam, beta, sig = eng.get_states()
sm = eng.get_summaries()
k = gam.sum(1)
print("dense variant nsig=%d chains=%d: %.1f us per sweep-round, %.3g sweeps/s, kbar %.2f (min %d max %d), accepts/sweep %.3f, slot hits %.0f%%, signals in %.3f"
      % (nsig, chains, dt / nsw * 1e6, chains * nsw / dt, sm["k_sum"] / sm["sweeps"], k.min(), k.max(), sm["accepts"] / sm["sweeps"],
         100 * sm["slot_hits"] / max(sm["accepts"], 1), gam[:, :nsig].mean()))
