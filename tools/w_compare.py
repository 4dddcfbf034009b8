#!/usr/bin/env python3
"""Debug aid: run the same chains with 1 and 2 wavefronts per chain and report
the first sweep after which their states differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import regression_data, spike_slab_prior
n, p, nsig, chains = 400, int(os.environ.get("P", "40")), 5, 8
X, y, _ = regression_data(n, p, nsig, seed=3)
engs = []
for w in ("1", "2"):
    eng = boom_amd.Engine(chains, seed=11)
    eng.set_tuning(waves_per_chain=int(w))
    eng.build_suf_from_xy(X, y)
    s = eng.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
    prior = spike_slab_prior(suf, nsig)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(p, np.uint8); g0[0] = 1
    eng.set_state(g0)
    engs.append(eng)
for it in range(30):
    out = []
    for w, eng in zip(("1", "2"), engs):
        eng.sweep(1)
        out.append(eng.get_states())
    g_same = np.array_equal(out[0][0], out[1][0])
    b_same = np.array_equal(out[0][1], out[1][1])
    s_same = np.array_equal(out[0][2], out[1][2])
    print("sweep %d: gamma %s beta %s sigsq %s" % (it, g_same, b_same, s_same))
    if not (g_same and b_same and s_same):
        c = int(np.argmax((out[0][2] != out[1][2]) | (out[0][0] != out[1][0]).any(1) | (out[0][1] != out[1][1]).any(1)))
        print(" chain", c, "sigsq", out[0][2][c], out[1][2][c])
        print(" gamma1", np.flatnonzero(out[0][0][c]), "gamma2", np.flatnonzero(out[1][0][c]))
        print(" beta1", out[0][1][c][out[0][0][c] > 0], "beta2", out[1][1][c][out[1][0][c] > 0])
        break
