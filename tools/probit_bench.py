#!/usr/bin/env python3
"""Diagnostic timing of the probit / logit spike-and-slab paths (f3) at BASELINE
config 5's per-GPU shape: n=5e4, p=1024, Bernoulli data, 512 chains.  Not a bench
line.  usage: probit_bench.py [n p signals chains [probit|logit|pg [timed sweeps]]]
(pg: the logit sampler with the Polya-Gamma imputer)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import logit_data, probit_data, probit_slab

n, p, nsig, chains = (int(v) for v in (sys.argv[1:5] or (50000, 1024, 8, 512)))
kind = sys.argv[5] if len(sys.argv) > 5 else "probit"
nsw = int(sys.argv[6]) if len(sys.argv) > 6 else 20
X, y, nt, btrue = (probit_data if kind == "probit" else logit_data)(n, p, nsig, seed=8675309)
slab, pi = probit_slab(X, nt, nsig)
eng = boom_amd.Engine(chains, seed=4)
t0 = time.perf_counter()
(eng.probit_set_data if kind == "probit" else eng.logit_set_data)(X, y, nt, 5)
print("set_data (X'NX build incl. upload): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
if kind == "pg":
    eng.logit_set_imputer(1)
eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
eng.set_spike(pi)
g0 = np.zeros(p, np.uint8); g0[0] = 1
eng.set_state(g0)
sweep = eng.probit_sweep if kind == "probit" else eng.logit_sweep
sweep(max(2, nsw // 2))
t0 = time.perf_counter()
sweep(nsw)
dt = time.perf_counter() - t0
gam, beta, _ = eng.get_states()
print(kind + " n=%d p=%d chains=%d: %.2f ms per sweep-round, %.3g sweeps/s, kbar %.2f, signals in: %s"
      % (n, p, chains, dt / nsw * 1e3, chains * nsw / dt, gam.sum(1).mean(), gam[:, :nsig].mean(0).round(2)))
