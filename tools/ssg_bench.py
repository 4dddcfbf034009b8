#!/usr/bin/env python3
"""Diagnostic timing of the GENERAL structural state-space path (f2): T=2000, p=100, 1024
chains, block lists the shape-specialised kernel does not cover -- bsts's daily-data model
(trend + day-of-week + a 52-season cycle of duration 7: m = 59), a weekly + 4 x 7 cycle, a
seasonal-only model.  Not a bench line.  usage: ssg_bench.py [case ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import bsts_priors, general_data, general_spec

CASES = {
    "weekly_4x7": [("trend",), ("seasonal", 7, 1), ("seasonal", 4, 7)],
    "daily52x7": [("trend",), ("seasonal", 7, 1), ("seasonal", 52, 7)],
    "seasonal_only12": [("seasonal", 12, 1)],
    "template_trend12_general_kernel": [("trend",), ("seasonal", 12, 1)],
}
T, p, nsig, chains = 2000, 100, 5, 1024
for name in (sys.argv[1:] or list(CASES)):
    desc = CASES[name]
    seas = [(b[1], b[2]) for b in desc if b[0] == "seasonal"]
    X, y, _, _ = general_data(T, p, nsig, seas[:2], seed=8675309)
    prior, _, sig_up = bsts_priors(X, y, 5)
    blocks = general_spec(y, desc)
    eng = boom_amd.Engine(chains, seed=4)
    eng.ss_set_data(y, X, None)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                   sigma_upper_limit=sig_up)
    eng.ss_set_state_models(blocks)
    if name.endswith("general_kernel"):
        eng.ss_set_tuning(use_template_kernel=False)
    eng.set_state(np.zeros(p, np.uint8))
    eng.ss_sweep(6)
    n = 10
    t0 = time.perf_counter()
    eng.ss_sweep(n)
    dt = time.perf_counter() - t0
    gam = eng.get_states()[0]
    print("%-34s m=%2d chains %4d: %9.1f us per sweep-round, %.3g sweeps/s, kbar %.2f"
          % (name, sum(b["dim"] for b in blocks), chains, dt / n * 1e6, chains * n / dt, gam.sum(1).mean()))
    eng.close()
