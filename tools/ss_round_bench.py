#!/usr/bin/env python3
"""BASELINE configs[2] (bsts local level + regression, T=2000 p=100, 1024 chains): the round
as one persistent launch per call (ss_round_kernel.hip) beside the separate launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import bsts_priors, state_space_data
T, p, nsig = 2000, 100, 5
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
X, y, btrue, _ = state_space_data(T, p, nsig, seed=8675309)
prior, ss, sig_up = bsts_priors(X, y, 5)
for kernel in (4, 5, 4, 5):
    eng = boom_amd.Engine(chains, seed=4)
    eng.ss_set_data(y, X, None)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
    eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"],
                           ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
    eng.set_state(np.zeros(p, np.uint8))
    eng.ss_set_tuning(kernel=kernel)
    eng.ss_sweep(200)
    dts = []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.ss_sweep(200)
        dts.append((time.perf_counter() - t0) / 200)
    gam, beta, sig = eng.get_states()
    print("%s: %.1f us per round (median of 5 x 200; min %.1f), %.3g sweeps/s, kbar %.2f" % (
        "round kernel" if kernel == 5 else "separate launches", np.median(dts) * 1e6, min(dts) * 1e6,
        chains / np.median(dts), gam.sum(1).mean()), flush=True)
    eng.close()
