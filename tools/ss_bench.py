#!/usr/bin/env python3
"""Diagnostic timing of BASELINE config 3 (bsts local level + regression,
T=2000 p=100, 1024 chains)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import bsts_priors, state_space_data
T, p, nsig, chains = 2000, 100, 5, 1024
X, y, btrue, _ = state_space_data(T, p, nsig, seed=8675309)
prior, ss, sig_up = bsts_priors(X, y, 5)
eng = boom_amd.Engine(chains, seed=4, max_model_size_hint=int(os.environ.get("KCAP_HINT", "0")))
eng.ss_set_data(y, X, None)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"],
                       ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
eng.set_state(np.zeros(p, np.uint8))
eng.ss_sweep(64)   # (launches of 64 rounds each: the counters of a profile are per launch)
t0 = time.perf_counter(); n = 128
eng.ss_sweep(n)
dt = time.perf_counter() - t0
gam, beta, sig = eng.get_states()
print("C3: %.1f us per sweep-round, %.3g sweeps/s, kbar %.2f" % (dt / n * 1e6, chains * n / dt, gam.sum(1).mean()))
