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
# arguments: [--kernel=K] trend,nseasons,chains[,ar lags]
KERNEL = 1
args = []
for a in sys.argv[1:]:
    if a.startswith("--kernel="):
        KERNEL = int(a.split("=")[1])
    else:
        args.append(a)
for spec_arg in [tuple(int(v) for v in a.split(",")) for a in args] or ((1, 0, 1024), (2, 0, 1024), (2, 7, 1024), (2, 12, 1024), (2, 12, 4096), (2, 12, 1024, 2)):
    trend, ns, chains = spec_arg[:3]
    lags = spec_arg[3] if len(spec_arg) > 3 else 0
    X, y, btrue, _ = structural_data(T, p, nsig, ns, seed=8675309, ar_coef=[1.2, -0.4] if lags == 2 else ([0.6] if lags == 1 else ([0.2] * lags if lags else None)))
    prior, _, sig_up = bsts_priors(X, y, 5)
    spec = structural_spec(y, trend, ns, ar_lags=lags)
    m0 = trend + max(ns - 1, 0)
    eng = boom_amd.Engine(chains, seed=4)
    eng.ss_set_data(y, X, None)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                   sigma_upper_limit=sig_up)
    eng.ss_set_structural(trend, ns, spec["var_df"], spec["var_sigma_guess"],
                          spec["var_sigma_upper_limit"], spec["var_initial_sigma"],
                          spec["initial_state_mean"][:m0], spec["initial_state_variance"][:m0])
    if lags:
        ar = spec["ar"]
        eng.ss_add_ar(lags, ar["df"], ar["sigma_guess"], ar["sigma_upper_limit"], ar["initial_sigma"],
                      ar["initial_phi"], spec["initial_state_mean"][m0:], spec["initial_state_variance"][m0:])
    eng.set_state(np.zeros(p, np.uint8))
    eng.ss_set_tuning(kernel=KERNEL)
    eng.set_kernel_timing(True)
    eng.ss_sweep(20)
    eng.kernel_times()
    n = 30
    t0 = time.perf_counter()
    eng.ss_sweep(n)
    dt = time.perf_counter() - t0
    gam, beta, sig = eng.get_states()
    kt = eng.kernel_times()
    print("kernel %d trend %d nseasons %2d ar %d (m=%2d) chains %4d: %8.1f us per sweep-round, %.3g sweeps/s, kbar %.2f; state kernel %s"
          % (KERNEL, trend, ns, lags, m0 + lags, chains, dt / n * 1e6, chains * n / dt, gam.sum(1).mean(),
             {k: v for k, v in kt.items() if "ssm" in k.lower() or "state" in k.lower()}))
