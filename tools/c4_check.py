#!/usr/bin/env python3
"""BASELINE config 4 shape on ONE GPU's shard: n=1e5 (suf built on device from
a smaller n to keep the host light), p=4096, 32 signals, 1024 chains."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import regression_data, spike_slab_prior
n, p, nsig, chains = int(os.environ.get("C4_N", "20000")), 4096, 32, 1024
t0 = time.perf_counter()
X, y, _ = regression_data(n, p, nsig, seed=8675309)
print("data %.1fs" % (time.perf_counter() - t0))
eng = boom_amd.Engine(chains, seed=1)
t0 = time.perf_counter(); eng.build_suf_from_xy(X, y); print("suf build (incl. H2D) %.2fs" % (time.perf_counter() - t0))
s = eng.get_suf()
suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
assert np.max(np.abs(s["xtx"] - X.T @ X)) < 1e-7 * n
prior = spike_slab_prior(suf, nsig)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
g0 = np.zeros(p, np.uint8); g0[0] = 1
eng.set_state(g0)
t0 = time.perf_counter(); eng.sweep(30); print("burn-in 30 sweeps %.2fs" % (time.perf_counter() - t0))
eng.reset_summaries()
t0 = time.perf_counter(); eng.sweep(20); dt = time.perf_counter() - t0
gam, beta, sig = eng.get_states()
sm = eng.get_summaries()
if "phase_cycles" in sm and sm["phase_cycles"].sum() > 0:
    names = ["shuffle uniforms", "shuffle serial", "refactor", "proposal batches", "swap", "sigma", "beta", "rest"]
    print("  accepts/sweep %.3f; per chain: mean %.0f cycles/sweep, slowest %.0f" % (sm["accepts"] / sm["sweeps"], sm["phase_cycles"].sum() / sm["sweeps"], sm["slot_hits"] / 20))
    for nm, v in zip(names, sm["phase_cycles"]):
        print("  %-18s %10.0f cycles/sweep" % (nm, v / sm["sweeps"]))
print("C4 shard: %.1f us per sweep-round, %.3g sweeps/s, kbar %.2f, signals in %.3f, nulls in %.5f, sigma %.3f"
      % (dt / 20 * 1e6, chains * 20 / dt, sm["k_sum"] / sm["sweeps"], gam[:, :nsig].mean(), gam[:, nsig:].mean(), np.sqrt(sig).mean()))
