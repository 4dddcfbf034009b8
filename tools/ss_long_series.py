#!/usr/bin/env python3
"""Diagnostic timing of the bsts local level + regression path on a series LONGER than the
lane-major kernel takes (T = 4000 > 2048: kalman_simsmooth_kernel, separate launches per round)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, boom_amd
from cases import bsts_priors, state_space_data
T, p, nsig, chains = 4000, 100, 5, 1024
X, y, btrue, _ = state_space_data(T, p, nsig, seed=8675309)
prior, ss, sig_up = bsts_priors(X, y, 5)
eng = boom_amd.Engine(chains, seed=4)
eng.ss_set_data(y, X, None)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"], ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
eng.set_state(np.zeros(p, np.uint8))
eng.ss_sweep(16)
t0 = time.perf_counter(); n = 64
eng.ss_sweep(n)
dt = time.perf_counter() - t0
print("T=4000: %.1f us per sweep-round" % (dt / n * 1e6))
