"""Random interleavings of draw_next / get_state / mutators / plain sweeps on an engine whose
look-ahead batches overlap, against an engine that sweeps one call at a time: every chain
equal wherever compared (diagnostic; the same checks at small sizes live in
tests/test_drop_in_gpu.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import boom_amd
from cases import regression_data, spike_slab_prior


def engine(chains):
    X, y, _ = regression_data(10000, 512, 16, seed=8675309)
    eng = boom_amd.Engine(chains, seed=8675309)
    eng.build_suf_from_xy(X, y)
    s = eng.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
    prior = spike_slab_prior(suf, 16)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(512, np.uint8); g0[0] = 1
    eng.set_state(g0)
    return eng


rng = np.random.Generator(np.random.PCG64(1))
for chains, L, iters in ((1024, 64, 6000), (1024, 7, 3000), (640, 256, 5000)):
    a, b = engine(chains), engine(chains)
    a.set_lookahead(L)
    t0 = time.perf_counter()
    n = checks = 0
    while n < iters:
        k = int(rng.integers(1, 3 * L))
        for _ in range(k):
            a.draw_next()
            a.get_state(0)
        n += k
        b.sweep(k)
        what = int(rng.integers(0, 4))
        if what == 0:
            mf = int(rng.integers(50, 513))
            a.set_options(max_flips=mf); b.set_options(max_flips=mf)
        elif what == 1:
            ga, gb = a.get_states(), b.get_states()
            assert all(np.array_equal(u, v) for u, v in zip(ga, gb)), (chains, L, n)
            checks += 1
        elif what == 2:
            a.sweep(3); b.sweep(3); n += 3
    ga, gb = a.get_states(), b.get_states()
    assert all(np.array_equal(u, v) for u, v in zip(ga, gb)), (chains, L, "end")
    print("chains %d L %d: %d iterations, %d full comparisons, equal; %.1f s"
          % (chains, L, n, checks, time.perf_counter() - t0), flush=True)
    a.close(); b.close()
