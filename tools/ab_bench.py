#!/usr/bin/env python3
"""Interleaved A/B timing of engine variants selected by environment variables
(read at launch time), in ONE process on ONE device.
usage: ab_bench.py VAR=a,b[,c] [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import regression_data, spike_slab_prior
var, vals = sys.argv[1].split("=")
vals = vals.split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, p, nsig, chains = 10000, 512, 16, 1024
X, y, _ = regression_data(n, p, nsig, seed=8675309)
engs = {}
for v in vals:
    os.environ[var] = v
    eng = boom_amd.Engine(chains, seed=1)
    eng.build_suf_from_xy(X, y)
    s = eng.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
    prior = spike_slab_prior(suf, nsig)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(p, np.uint8); g0[0] = 1
    eng.set_state(g0)
    eng.sweep(200)
    engs[v] = eng
times = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        os.environ[var] = v
        t0 = time.perf_counter()
        for _ in range(5):
            engs[v].sweep(20, sync=False)
        engs[v].sync()
        times[v].append((time.perf_counter() - t0) / 100 * 1e6)
for v in vals:
    t = np.array(times[v])
    print("%s=%s: median %.1f us  min %.1f us per sweep-round  (%.2f M sweeps/s)" % (var, v, np.median(t), t.min(), chains / np.median(t)))
ref = engs[vals[0]].get_states()
for v in vals[1:]:
    o = engs[v].get_states()
    print("  states identical to %s: %s" % (vals[0], all(np.array_equal(a, b) for a, b in zip(ref, o))))
