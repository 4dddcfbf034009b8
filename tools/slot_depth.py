#!/usr/bin/env python3
"""Diagnostic: how many of a chain's model changes return to one of its last d
models (what d slots per chain could serve), from recorded draws of the C2
workload (sweep resolution)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, boom_amd
from cases import regression_data, spike_slab_prior
n, p, nsig, chains, nsw = 10000, 512, 16, 64, 1000
X, y, _ = regression_data(n, p, nsig, seed=8675309)
eng = boom_amd.Engine(chains, seed=1)
eng.build_suf_from_xy(X, y)
s = eng.get_suf()
suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
prior = spike_slab_prior(suf, nsig)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
g0 = np.zeros(p, np.uint8); g0[0] = 1
eng.set_state(g0); eng.sweep(200)
eng.enable_draws(nsw); eng.sweep(nsw)
changes = 0
hits = {2: 0, 3: 0, 4: 0, 8: 0}
for c in range(chains):
    g, _, _ = eng.get_draws(c, nsw)
    keys = [row.tobytes() for row in g]
    hist = {d: [keys[0]] for d in hits}
    for k in keys[1:]:
        if k == hist[2][-1]:
            continue
        changes += 1
        for d in hits:
            h = hist[d]
            if k in h[:-1][-(d - 1):]:
                hits[d] += 1
            if k in h:
                h.remove(k)
            h.append(k)
            del h[:-d]
print("model changes %d;" % changes, ", ".join("%d slots: %.1f %%" % (d, 100.0 * hits[d] / changes) for d in sorted(hits)))
