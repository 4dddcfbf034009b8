#!/usr/bin/env python3
"""Diagnostic timing of the structural state-space path (f2): T=2000, p=100,
regression + local linear trend + seasonal state, 1024 chains (BASELINE config 3's
shape with the richer state).  Not a bench line."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import bsts_priors, structural_data, structural_spec

T, p, nsig = 2000, 100, 5
for trend, ns, chains in [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or ((1, 0, 1024), (2, 0, 1024), (2, 7, 1024), (2, 12, 1024), (2, 12, 4096)):
    X, y, btrue, _ = structural_data(T, p, nsig, ns, seed=8675309)
    prior, _, sig_up = bsts_priors(X, y, 5)
    spec = structural_spec(y, trend, ns)
    eng = boom_amd.Engine(chains, seed=4)
    eng.ss_set_data(y, X, None)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                   sigma_upper_limit=sig_up)
    eng.ss_set_structural(trend, ns, spec["var_df"], spec["var_sigma_guess"],
                          spec["var_sigma_upper_limit"], spec["var_initial_sigma"],
                          spec["initial_state_mean"], spec["initial_state_variance"])
    eng.set_state(np.zeros(p, np.uint8))
    eng.ss_sweep(20)
    n = 30
    t0 = time.perf_counter()
    eng.ss_sweep(n)
    dt = time.perf_counter() - t0
    gam, beta, sig = eng.get_states()
    print("trend %d nseasons %2d (m=%2d) chains %4d: %8.1f us per sweep-round, %.3g sweeps/s, kbar %.2f"
          % (trend, ns, trend + max(ns - 1, 0), chains, dt / n * 1e6, chains * n / dt, gam.sum(1).mean()))
