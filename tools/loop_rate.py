"""The reference callers' loop -- one sample_posterior() per iteration, the
model's parameters read back after each (spike_slab_wrapper.cc:233-242,
spikeslab.py:191-207) -- through ba_draw_next + ba_get_state(chain 0), for
several look-ahead lengths.  C2 workload, 1024 chains.  Diagnostic, not a bench
line."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import boom_amd  # noqa: E402
from cases import regression_data, spike_slab_prior, suf_from_xy  # noqa: E402

X, y, _ = regression_data(10000, 512, 16, seed=8675309)
suf = suf_from_xy(X, y)
prior = spike_slab_prior(suf, 16)
for L in (1, 16, 64, 256):
    eng = boom_amd.Engine(1024, seed=1)
    eng.build_suf_from_xy(X, y)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(512, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.sweep(1000)
    eng.set_lookahead(L)
    for _ in range(max(L, 64)):
        eng.draw_next()
    eng.get_state(0)
    nb = 24                       # batches timed; the median batch is reported
    tb = np.zeros(nb)
    for b in range(nb):
        t = time.perf_counter()
        for _ in range(L):
            eng.draw_next()
            eng.get_state(0)
        tb[b] = time.perf_counter() - t
    dt = float(np.median(tb)) / L
    print("lookahead %4d: %6.1f us per iteration, %6.2f M sweeps/s (1024 chains)"
          % (L, 1e6 * dt, 1024 / dt / 1e6))
