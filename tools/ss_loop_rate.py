#!/usr/bin/env python3
"""Diagnostic: the callers' loop on the bsts path -- one ba_ss_sweep(1) per iteration, then
what bindings/boom/DeviceStateSpacePosteriorSampler pulls after every draw (chain 0's
regression draw, level variance and state) -- against ba_ss_sweep(n) in one call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import bsts_priors, state_space_data
T, p, chains = 2000, 100, 1024
X, y, _, _ = state_space_data(T, p, 5, seed=8675309)
prior, ss, sig_up = bsts_priors(X, y, 5)
eng = boom_amd.Engine(chains, seed=4)
eng.ss_set_data(y, X, None)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"],
                       ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
eng.set_state(np.zeros(p, np.uint8))
eng.ss_sweep(100)
n = 300
t0 = time.perf_counter()
for _ in range(n):
    eng.ss_sweep(1)
    eng.sync()
dt1 = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    eng.ss_sweep(1)
    g, b, s2 = eng.get_state(0)
    st = eng.ss_get_state(0)
dt2 = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
eng.ss_sweep(n); eng.sync()
dt3 = (time.perf_counter() - t0) / n
eng.ss_set_lookahead(64)
for _ in range(64):
    eng.ss_draw_next()
t0 = time.perf_counter()
for _ in range(n):
    eng.ss_draw_next()
    g, b, s2 = eng.get_state(0)
    st = eng.ss_get_state(0, suf=False)
dt4 = (time.perf_counter() - t0) / n
print("look-ahead 64: draw_next + chain 0's draw pulled %.1f us per draw" % (dt4 * 1e6))
print("per draw: sweep(1)+sync %.1f us, + chain 0's draw pulled %.1f us, inside one call %.1f us" % (dt1 * 1e6, dt2 * 1e6, dt3 * 1e6))
for L in (64, 256):
    eng.ss_set_lookahead(L)
    for _ in range(L):
        eng.ss_draw_next()
    m = 4 * L
    t0 = time.perf_counter()
    for _ in range(m):
        eng.ss_draw_next()
    eng.get_state(0)
    dt5 = (time.perf_counter() - t0) / m
    t0 = time.perf_counter()
    for _ in range(m):
        eng.ss_draw_next()
        g, b, s2 = eng.get_state(0)
        st = eng.ss_get_state(0, suf=False)
    dt6 = (time.perf_counter() - t0) / m
    print("look-ahead %d: draw_next only %.1f us per draw; with chain 0 pulled %.1f us" % (L, dt5 * 1e6, dt6 * 1e6))
