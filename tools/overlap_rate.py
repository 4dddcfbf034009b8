"""Sweep launches that overlap (consecutive ba_sweep calls, no sync) against launches kept
apart: equality of the draws, then the rates at 1000 / 250 / 64 sweeps per launch.  Diagnostic."""
import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import boom_amd
from cases import regression_data, spike_slab_prior

def engine(chains=1024):
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

a, b = engine(), engine()
for _ in range(12):
    a.sweep(40, sync=False)
a.sync()
for _ in range(12):
    b.sweep(40, sync=True)
sa, sb = a.get_states(), b.get_states()
print("pipelined == one at a time:", all(np.array_equal(u, v) for u, v in zip(sa, sb)), flush=True)
sma, smb = a.get_summaries(), b.get_summaries()
print("summaries equal:", all(np.array_equal(np.asarray(sma[k]), np.asarray(smb[k])) for k in sma if k not in ("min_margin",)), flush=True)
for L in (1000, 250, 64):
    K = max(4, 4000 // L)
    for name, eng in (("pipelined", a), ("synced", b)):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(K):
                eng.sweep(L, sync=(name == "synced"))
            eng.sync()
            ts.append(time.perf_counter() - t0)
        dt = min(ts)
        print("%-9s %4d sweeps x %2d launches: %.2f ms per launch, %.2f M sweeps/s" % (name, L, K, dt / K * 1e3, 1024 * L * K / dt / 1e6), flush=True)
sa, sb = a.get_states(), b.get_states()
print("still equal:", all(np.array_equal(u, v) for u, v in zip(sa, sb)))
