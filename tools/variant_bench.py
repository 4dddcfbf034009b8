#!/usr/bin/env python3
"""Timing of one build of the library (BOOM_AMD_LIB=tools/build/<name>/libboomamd.so, or
the default) on the configurations that matter, one line each, for A/B comparisons of
builds in separate processes.  Diagnostic, not a bench line.
usage: variant_bench.py [c2] [c2x2048] [c3] [loop64] [c4] [structural] [logit] [probit]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import (bsts_priors, logit_data, probit_data, probit_slab, regression_data, spike_slab_prior,
                   state_space_data, structural_data, structural_spec)

what = sys.argv[1:] or ["c2", "c2x2048", "c3", "loop64"]
tag = os.environ.get("BOOM_AMD_LIB", "default")
tag = os.path.basename(os.path.dirname(tag)) if tag != "default" else tag


def c2_engine(chains):
    X, y, _ = regression_data(10000, 512, 16, seed=8675309)
    eng = boom_amd.Engine(chains, seed=8675309)
    eng.build_suf_from_xy(X, y)
    s = eng.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
    prior = spike_slab_prior(suf, 16)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(512, np.uint8); g0[0] = 1
    eng.set_state(g0)
    eng.sweep(1000)
    return eng


for w in what:
    if w == "c2" or w.startswith("c2x"):
        chains = 1024 if w == "c2" else int(w[3:])
        eng = c2_engine(chains)
        eng.sweep(1000)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); eng.sweep(1000); ts.append(time.perf_counter() - t0)
        print("[%s] %s: %.3f ms per 1000-sweep launch (min %.3f), %.2f M sweeps/s"
              % (tag, w, np.median(ts) * 1e3, min(ts) * 1e3, chains * 1000 / np.median(ts) / 1e6), flush=True)
        eng.close()
    elif w == "loop64":
        eng = c2_engine(1024)
        L = 64
        eng.set_lookahead(L)
        for _ in range(L):
            eng.draw_next()
        eng.get_state(0)
        tb = []
        for b in range(24):
            t = time.perf_counter()
            for _ in range(L):
                eng.draw_next(); eng.get_state(0)
            tb.append(time.perf_counter() - t)
        dt = float(np.median(tb)) / L
        print("[%s] loop64: %.1f us per iteration, %.2f M sweeps/s" % (tag, dt * 1e6, 1024 / dt / 1e6), flush=True)
        eng.close()
    elif w == "c3" or w.startswith("c3x"):
        c3chains = 1024 if w == "c3" else int(w[3:])
        T, p = 2000, 100
        X, y, _, _ = state_space_data(T, p, 5, seed=8675309)
        prior, ss, sig_up = bsts_priors(X, y, 5)
        eng = boom_amd.Engine(c3chains, seed=4)
        eng.ss_set_data(y, X, None)
        eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
        eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"],
                               ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
        eng.set_state(np.zeros(p, np.uint8))
        eng.ss_sweep(50)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); eng.ss_sweep(200); ts.append((time.perf_counter() - t0) / 200)
        eng.set_kernel_timing(True)
        eng.ss_sweep(100)
        kt = {k.split("_kernel")[0]: round(ms / n * 1e3, 1) for k, (ms, n) in eng.kernel_times().items()}
        print("[%s] %s: %.1f us per round (min %.1f), %.2f M sweeps/s, kernels %s"
              % (tag, w, np.median(ts) * 1e6, min(ts) * 1e6, c3chains / np.median(ts) / 1e6, kt), flush=True)
        eng.close()
    elif w == "structural":
        T, p = 2000, 100
        X, y, _, _ = structural_data(T, p, 5, 12, seed=8675309)
        prior, _, sig_up = bsts_priors(X, y, 5)
        spec = structural_spec(y, 2, 12)
        eng = boom_amd.Engine(1024, seed=4)
        eng.ss_set_data(y, X, None)
        eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
        eng.ss_set_structural(2, 12, spec["var_df"], spec["var_sigma_guess"], spec["var_sigma_upper_limit"],
                              spec["var_initial_sigma"], spec["initial_state_mean"], spec["initial_state_variance"])
        eng.set_state(np.zeros(p, np.uint8))
        eng.ss_sweep(10)
        t0 = time.perf_counter(); eng.ss_sweep(20); dt = (time.perf_counter() - t0) / 20
        print("[%s] structural m=13: %.2f ms per round, %.1f k sweeps/s" % (tag, dt * 1e3, 1024 / dt / 1e3), flush=True)
        eng.close()
    elif w in ("logit", "probit"):
        n, p, chains = 50000, 1024, 512
        X, y, nt, _ = (probit_data if w == "probit" else logit_data)(n, p, 8, seed=8675309)
        slab, pi = probit_slab(X, nt, 8)
        eng = boom_amd.Engine(chains, seed=4)
        (eng.probit_set_data if w == "probit" else eng.logit_set_data)(X, y, nt, 5)
        eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
        eng.set_spike(pi)
        g0 = np.zeros(p, np.uint8); g0[0] = 1
        eng.set_state(g0)
        sweep = eng.probit_sweep if w == "probit" else eng.logit_sweep
        sweep(10)
        t0 = time.perf_counter(); sweep(20); dt = (time.perf_counter() - t0) / 20
        print("[%s] %s: %.2f ms per round, %.1f k sweeps/s" % (tag, w, dt * 1e3, chains / dt / 1e3), flush=True)
        eng.close()
