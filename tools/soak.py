#!/usr/bin/env python3
"""Long runs of the paths whose rare branches would only show up over many sweeps
(substream overruns, hull capacities, capacity escalation): every call must return
without a chain error.  usage: soak.py [scale]  (scale 1 = about two minutes of GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import (bsts_priors, logit_data, probit_data, probit_slab, state_space_data,
                   structural_data, structural_spec)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0

t0 = time.perf_counter()
X, y, _, _ = state_space_data(2000, 100, 5, seed=1)
prior, ss, sig_up = bsts_priors(X, y, 5)
eng = boom_amd.Engine(1024, seed=11)
eng.ss_set_data(y, X, None)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"],
                       ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
eng.set_state(np.zeros(100, np.uint8))
n = int(100000 * scale)
for _ in range(10):
    eng.ss_sweep(n // 10)
print("bsts local level: %d sweeps x 1024 chains ok (%.0f s), kbar %.2f" % (n, time.perf_counter() - t0, eng.get_states()[0].sum(1).mean()), flush=True)
eng.close()

t0 = time.perf_counter()
X, y, _, obs = structural_data(1000, 20, 3, 12, seed=2, missing_frac=0.02)
prior, _, sig_up = bsts_priors(X, y, 3)
spec = structural_spec(y, 2, 12)
eng = boom_amd.Engine(1024, seed=12)
eng.ss_set_data(y, X, obs)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
eng.ss_set_structural(2, 12, spec["var_df"], spec["var_sigma_guess"], spec["var_sigma_upper_limit"],
                      spec["var_initial_sigma"], spec["initial_state_mean"], spec["initial_state_variance"])
eng.set_state(np.zeros(20, np.uint8))
n = int(6000 * scale)
eng.ss_sweep(n)
print("trend + 12 seasons: %d sweeps x 1024 chains ok (%.0f s)" % (n, time.perf_counter() - t0), flush=True)
eng.close()

for kind in ("logit", "probit"):
    t0 = time.perf_counter()
    X, y, nt, _ = (logit_data if kind == "logit" else probit_data)(20000, 256, 6, seed=3, max_trials=5)
    slab, pi = probit_slab(X, nt, 6)
    eng = boom_amd.Engine(512, seed=13)
    (eng.logit_set_data if kind == "logit" else eng.probit_set_data)(X, y, nt, 5)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
    eng.set_spike(pi)
    g0 = np.zeros(256, np.uint8); g0[0] = 1
    eng.set_state(g0)
    n = int(3000 * scale)
    (eng.logit_sweep if kind == "logit" else eng.probit_sweep)(n)
    gam = eng.get_states()[0]
    print("%s, binomial counts up to 5: %d sweeps x 512 chains ok (%.0f s), kbar %.2f, signals %s"
          % (kind, n, time.perf_counter() - t0, gam.sum(1).mean(), gam[:, :6].mean(0).round(2)), flush=True)
    eng.close()
