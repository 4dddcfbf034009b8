import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import boom_amd
from cases import regression_data, spike_slab_prior, suf_from_xy
X, y, _ = regression_data(10000, 512, 16, seed=8675309)
suf = suf_from_xy(X, y); prior = spike_slab_prior(suf, 16)
L = 16
eng = boom_amd.Engine(1024, seed=1); eng.build_suf_from_xy(X, y)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
g0 = np.zeros(512, np.uint8); g0[0] = 1
eng.set_state(g0); eng.sweep(1000); eng.set_lookahead(L)
for _ in range(64): eng.draw_next()
eng.get_state(0)
reps = 40; td = np.zeros((reps, L)); tg = np.zeros((reps, L))
for r in range(reps):
    for i in range(L):
        t0 = time.perf_counter(); eng.draw_next(); t1 = time.perf_counter(); eng.get_state(0); t2 = time.perf_counter()
        td[r, i] = t1 - t0; tg[r, i] = t2 - t1
print("draw_next us by position in batch (median):", np.round(1e6 * np.median(td, 0), 1))
print("get_state us by position in batch (median):", np.round(1e6 * np.median(tg, 0), 1))
print("get_state (max):", np.round(1e6 * tg.max(0), 1))
print("per batch total us (median over reps):", np.round(1e6 * np.median((td + tg).sum(1)), 1))
# plain launches for comparison
eng2 = boom_amd.Engine(1024, seed=1); eng2.build_suf_from_xy(X, y)
eng2.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
eng2.set_state(g0); eng2.sweep(1000)
for n in (16, 64):
    eng2.sweep(n); t = time.perf_counter()
    for _ in range(20): eng2.sweep(n)
    print("plain sweep(%d): %.1f us" % (n, 1e6 * (time.perf_counter() - t) / 20))
