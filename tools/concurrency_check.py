#!/usr/bin/env python3
"""Every path once alone and once beside a second engine that keeps the GPU busy with
asynchronous launches of its own: the chains must come out bit for bit the same -- what
else shares the machine may change the timing inside a kernel, never a draw.  (Round 4:
the state kernels' variance draws had a race that only showed under such load.)
usage: concurrency_check.py [repeats]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import boom_amd
from cases import (bsts_priors, general_data, general_spec, logit_data, probit_data, probit_slab,
                   regression_data, spike_slab_prior, state_space_data)
from test_state_space_gpu import make_engine as level_engine
from test_structural_general_gpu import make_engine as general_engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def noise_engine():
    X, y, _ = regression_data(4000, 256, 12, seed=3)
    e = boom_amd.Engine(1024, seed=99)
    e.build_suf_from_xy(X, y)
    s = e.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
    pr = spike_slab_prior(suf, 12)
    e.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"])
    g0 = np.zeros(256, np.uint8); g0[0] = 1
    e.set_state(g0)
    e.sweep(50)
    return e


def states(e):
    return [np.array(x) for x in e.get_states()]


def check(name, make, step, extra=None):
    worst = 0
    for r in range(reps):
        a = make()
        for _ in range(6):
            step(a, True)
        ra = states(a) + (extra(a) if extra else [])
        b = make()
        nz = noise_engine()
        for _ in range(6):
            nz.sweep(30, sync=False)      # the noise runs on its own stream beside what follows
            step(b, False)
        b.sync(); nz.sync()
        rb = states(b) + (extra(b) if extra else [])
        bad = sum(0 if np.array_equal(u, v) else 1 for u, v in zip(ra, rb))
        worst = max(worst, bad)
        a.close(); b.close(); nz.close()
    print("%-34s %s" % (name, "same draws alone and under load" if worst == 0 else "DIFFERS (%d arrays)" % worst), flush=True)
    return worst == 0


ok = True
# bsts local level
X, y, _, obs = state_space_data(500, 20, 3, seed=5, missing_frac=0.02)
prior, ss, sig_up = bsts_priors(X, y, 3)
ok &= check("bsts local level", lambda: level_engine(256, 7, y, X, obs, prior, ss, sig_up, np.zeros(20, np.uint8)),
            lambda e, s: e.ss_sweep(5, sync=s), lambda e: [e.ss_get_state(3)["state"], e.ss_get_state(200)["state"]])
# structural: template and general shapes
for nm, desc in [("structural template trend+12", [("trend",), ("seasonal", 12, 1)]),
                 ("structural template +ar(2)", [("trend",), ("seasonal", 7, 1), ("ar", 2)]),
                 ("structural general 4x3 + ar", [("level",), ("seasonal", 4, 3), ("ar", 2)]),
                 ("structural general m=27", [("trend",), ("seasonal", 7, 1), ("seasonal", 20, 2)])]:
    seas = [(d[1], d[2]) for d in desc if d[0] == "seasonal"]
    Xg, yg, _, og = general_data(300, 8, 2, seas, seed=8, missing_frac=0.02, ar_coef=[0.5] if any(d[0] == "ar" for d in desc) else None)
    pg, _, su = bsts_priors(Xg, yg, 2)
    spec = general_spec(yg, desc)
    ok &= check(nm, lambda: general_engine(200, 7, yg, Xg, og, pg, spec, su, np.zeros(8, np.uint8)),
                lambda e, s: e.ss_sweep(4, sync=s), lambda e: [e.ss_get_state_draw(0), e.ss_get_state_draw(150)])
# the headline path and its siblings
Xr, yr, _ = regression_data(5000, 300, 10, seed=4)
def reg():
    e = boom_amd.Engine(512, seed=5)
    e.build_suf_from_xy(Xr, yr)
    s = e.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
    pr = spike_slab_prior(suf, 10)
    e.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"])
    g0 = np.zeros(300, np.uint8); g0[0] = 1
    e.set_state(g0)
    return e
ok &= check("BregVsSampler sweeps", reg, lambda e, s: e.sweep(40, sync=s))
ok &= check("adaptive sampler", reg, lambda e, s: e.adaptive_sweep(20, sync=s))
# GLMs
for kind, data in [("probit", probit_data), ("logit", logit_data)]:
    Xl, yl, nt, _ = data(3000, 40, 4, seed=6)
    slab, pi = probit_slab(Xl, nt, 4)
    def glm(kind=kind, Xl=Xl, yl=yl, nt=nt, slab=slab, pi=pi):
        e = boom_amd.Engine(128, seed=9)
        (e.probit_set_data if kind == "probit" else e.logit_set_data)(Xl, yl, nt, 5)
        e.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
        e.set_spike(pi)
        g0 = np.zeros(40, np.uint8); g0[0] = 1
        e.set_state(g0)
        return e
    ok &= check(kind + " spike-and-slab", glm, (lambda e, s, kind=kind: (e.probit_sweep if kind == "probit" else e.logit_sweep)(3, sync=s)))
print("ALL THE SAME" if ok else "SOME PATH DIFFERS")
sys.exit(0 if ok else 1)
