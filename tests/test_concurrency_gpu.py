"""Concurrency in the suite.  A race in the round-3 state kernels passed
181 call-by-call parity tests and showed only when look-ahead batches ran back to back beside
another engine's launches; overlapped launches are the product's default mode, and the bsts
rounds now run as one persistent kernel whose chains meet in tiles.  So, bounded to seconds:

* random interleavings of the bsts look-ahead (draw_next, readers, mutators, plain sweeps,
  forecasts) against a one-round-per-call engine, four state-model lists;
* every sampler family alone against the same family beside a busy second engine;
* two identical engines stepped side by side;
* the headline's mode -- 20 unsynchronised ba_sweep(1000) on BASELINE configs[1]'s shape,
  chains handed over between launches on two streams -- against the launches kept apart,
  all 1024 chains.

Everything is compared bit for bit: what shares the machine may change the timing inside a
kernel, never a draw.  (tests/concurrency_lib.py; the long forms are tools/ss_la_stress.py,
tools/concurrency_check.py.)"""
import numpy as np
import pytest

from concurrency_lib import LA_MODELS, alone_vs_loaded, families, la_stress, noise_engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(LA_MODELS))
def test_interleavings_behind_the_look_ahead(name):
    la_stress(name, 120, seed=1)


@pytest.fixture(scope="module")
def busy_engine():
    e = noise_engine()
    yield e
    e.close()


_FAMILIES = None


def _family(name):
    global _FAMILIES
    if _FAMILIES is None:
        _FAMILIES = families()
    return _FAMILIES[name]


@pytest.mark.parametrize("name", ["bsts local level", "structural template trend+12", "structural template +ar(2)",
                                  "structural general 4x3 + ar", "structural general m=27",
                                  "BregVsSampler sweeps", "BregVsSampler, 40 signals", "BregVsSampler, 70 signals",
                                  "adaptive sampler", "probit spike-and-slab", "logit spike-and-slab"])
def test_same_draws_alone_and_beside_a_busy_engine(busy_engine, name):
    make, step, extra = _family(name)
    assert alone_vs_loaded(make, step, extra, steps=4, noise=busy_engine) == 0


@pytest.mark.parametrize("name", ["bsts local level", "structural template trend+12",
                                  "structural general 4x3 + ar"])
def test_two_identical_engines_side_by_side(name):
    """asynchronous steps of two engines with the same seed interleaved on the device"""
    make, step, extra = _family(name)
    a, b = make(), make()
    for _ in range(5):
        step(a, False)
        step(b, False)
    a.sync()
    b.sync()
    for u, v in zip(a.get_states(), b.get_states()):
        assert np.array_equal(u, v)
    for u, v in zip(extra(a), extra(b)):
        assert np.array_equal(u, v)
    a.close()
    b.close()


def test_overlapped_headline_launches_equal_the_launches_kept_apart():
    """bench.py's timed region on configs[1]'s shape (n = 1e4, p = 512, 1024 chains): 20
    consecutive ba_sweep(1000) without a sync in between alternate between the engine's two
    streams and hand the chains over one by one; the same 20 launches with a sync
    after each: every chain ends in the same state, the summaries are the same."""
    import boom_amd
    from cases import regression_data, spike_slab_prior
    n, p, nsig, chains = 10000, 512, 16, 1024
    X, y, _ = regression_data(n, p, nsig, seed=8675309)

    def engine():
        e = boom_amd.Engine(chains, seed=20240)
        e.build_suf_from_xy(X, y)
        s = e.get_suf()
        suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"],
                   xsum=s["xbar"] * s["n"])
        pr = spike_slab_prior(suf, nsig)
        e.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"])
        g0 = np.zeros(p, np.uint8)
        g0[0] = 1
        e.set_state(g0)
        e.sweep(200)
        return e
    a, b = engine(), engine()
    for _ in range(20):
        a.sweep(1000, sync=False)
    a.sync()
    for _ in range(20):
        b.sweep(1000, sync=True)
    for u, v in zip(a.get_states(), b.get_states()):
        assert np.array_equal(u, v)
    sa, sb = a.get_summaries(), b.get_summaries()
    for k in sa:
        assert np.array_equal(np.asarray(sa[k]), np.asarray(sb[k])), k
    a.close()
    b.close()
