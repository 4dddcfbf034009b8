"""AdaptiveSpikeSlabRegressionSampler on the GPU (ssvs_adaptive_kernel.hip) --
what lm.spike runs for p > 100 --
against the oracle's restatement (pinned by tests/golden/adaptive_*.npz) on
the same Philox streams, through the C-ABI.

Bars: inclusion indicators bit-exact; beta, sigma^2 and the adapted birth /
death rates within 1e-8 relative.  A proposal here is a weighted draw over
cumulative rates followed by a Metropolis-Hastings test; both margins (distance
of the draw's uniform from a boundary of the cumulative sums, |log u - log
ratio|) are tracked on both sides.
"""
import numpy as np
import pytest

from cases import regression_data, spike_slab_prior, suf_from_xy
from oracle_lib import ssvs_options
from test_ssvs_gpu import make_engine, relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def _compare(oracle, eng, suf, prior, opts, seed, g0, nsweeps, chains, step, **ada):
    ora = {c: oracle.adaptive_run(suf, prior, opts, ("philox", seed, c), g0, nsweeps,
                                  ada.get("max_flips", -1), ada.get("step_size", -1.0),
                                  ada.get("target", -1.0), want_margin=True) for c in chains}
    done = 0
    while done < nsweeps:
        eng.adaptive_sweep(step)
        done += step
        gam, beta, sig = eng.get_states()
        for c in chains:
            o = ora[c]
            assert o["status"] == 0
            assert np.array_equal(gam[c], o["gamma"][done - 1]), (c, done)
            assert relerr(beta[c], o["beta"][done - 1]) < RTOL, (c, done)
            assert abs(sig[c] - o["sigsq"][done - 1]) < RTOL * sig[c], (c, done)
    for c in chains:
        b, d, it = eng.adaptive_get_rates(c)
        assert it == nsweeps
        assert relerr(b, ora[c]["birth"]) < RTOL and relerr(d, ora[c]["death"]) < RTOL
    return ora


def test_adaptive_c1_every_sweep(oracle):
    X, y, _ = regression_data(1000, 20, 6, seed=1)
    suf = suf_from_xy(X, y)
    prior = spike_slab_prior(suf, 5)
    g0 = np.zeros(20, np.uint8)
    g0[0] = 1
    eng = make_engine(8, 8675309, suf=suf, prior=prior, g0=g0)
    ora = _compare(oracle, eng, suf, prior, ssvs_options(), 8675309, g0, 60, range(8), 1)
    sm = eng.get_summaries()
    assert sm["sweeps"] == 8 * 60
    assert sm["min_margin"] > 1e-9 and sm["min_multi_margin"] > 1e-12
    assert min(o["min_margin"] for o in ora.values()) > 1e-9


def test_adaptive_p150_batched(oracle):
    """p beyond one 64-move batch per sweep is not the point here (100 moves per
    sweep): the point is long launches, rates adapting, capacity escalation from
    a cold start"""
    X, y, _ = regression_data(600, 150, 12, seed=2)
    suf = suf_from_xy(X, y)
    prior = spike_slab_prior(suf, 12)
    g0 = np.zeros(150, np.uint8)
    g0[0] = 1
    eng = make_engine(16, 7, suf=suf, prior=prior, g0=g0, tuning=dict(kcap_start=16))
    ora = _compare(oracle, eng, suf, prior, ssvs_options(), 7, g0, 100, [0, 5, 15], 25)
    assert max(o["birth"].max() for o in ora.values()) > 1.0   # rates did adapt


def test_adaptive_options_and_general_priors(oracle):
    """fewer moves per sweep, other step size / target, a model-size cap, a
    truncated sigma draw, non-zero prior means (exact-path candidates), an empty
    start (death moves impossible at first)"""
    X, y, _ = regression_data(400, 40, 4, seed=6, collinear=[1, 7, 9, 20])
    suf = suf_from_xy(X, y)
    pm = np.zeros(40)
    pm[[0, 3, 11]] = [0.5, -0.2, 0.1]
    prior = spike_slab_prior(suf, 4, prior_mean=pm, force_intercept=False)
    opts = ssvs_options(max_model_size=9, sigma_upper_limit=30.0)
    g0 = np.zeros(40, np.uint8)
    eng = make_engine(6, 11, suf=suf, prior=prior, opts=opts, g0=g0)
    eng.adaptive_set_options(max_flips=30, step_size=0.05, target=0.2)
    _compare(oracle, eng, suf, prior, opts, 11, g0, 80, range(6), 20, max_flips=30,
             step_size=0.05, target=0.2)
    gam, _, _ = eng.get_states()
    assert gam.sum(axis=1).max() <= 9


def test_adaptive_c2_shape_properties():
    """BASELINE configs[1] shape with the sampler lm.spike would use for it"""
    import boom_amd
    n, p, nsig, chains = 10000, 512, 16, 256
    X, y, _ = regression_data(n, p, nsig, seed=8675309)
    eng = boom_amd.Engine(chains, seed=1)
    eng.build_suf_from_xy(X, y)
    s = eng.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"],
               xsum=s["xbar"] * s["n"])
    prior = spike_slab_prior(suf, nsig)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.adaptive_sweep(300)
    gam, beta, sig = eng.get_states()
    assert gam[:, :nsig].mean() > 0.97
    assert gam[:, nsig:].mean() < 0.01
    assert np.all(beta[gam == 0] == 0.0)
    assert abs(np.sqrt(sig).mean() - 1.0) < 0.05


# ------------------------------------------------- models of more than 64 variables
def test_adaptive_wide_start_above_64(oracle):
    """The chains start with 80 variables included: the LDS kernel parks them at once
    and the large-model kernel runs the birth / death moves from its table
    (tests/golden/adaptive_wide_start80.npz pins the oracle on the reference)."""
    n, p, nsig = 900, 160, 90
    X, y, _ = regression_data(n, p, nsig, seed=11)
    suf = suf_from_xy(X, y)
    prior = spike_slab_prior(suf, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[:80] = 1
    eng = make_engine(6, 5, suf=suf, prior=prior, g0=g0)
    ora = _compare(oracle, eng, suf, prior, ssvs_options(), 5, g0, 30, [0, 3, 5], 10)
    assert min(o["gamma"].sum(axis=1).min() for o in ora.values()) > 64
    sm = eng.get_summaries()
    assert sm["min_margin"] > 1e-9 and sm["min_multi_margin"] > 1e-12


def test_adaptive_wide_growth_through_64(oracle):
    """a start at one variable and 100 true signals: the model grows through the LDS
    kernel's capacities (16, 32, 48, 64: sweeps aborted and replayed) into the
    large-model kernel and through its capacities (128)"""
    n, p, nsig = 900, 130, 100
    X, y, _ = regression_data(n, p, nsig, seed=12)
    suf = suf_from_xy(X, y)
    prior = spike_slab_prior(suf, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng = make_engine(5, 9, suf=suf, prior=prior, g0=g0, tuning=dict(kcap_start=16))
    eng.adaptive_set_options(max_flips=130, step_size=-1.0, target=-1.0)
    ora = _compare(oracle, eng, suf, prior, ssvs_options(), 9, g0, 40, [0, 4], 8, max_flips=130)
    assert max(o["gamma"].sum(axis=1).max() for o in ora.values()) > 90
    # sweeps one at a time from here: the same chain
    _ = eng.get_states()
    eng2 = make_engine(5, 9, suf=suf, prior=prior, g0=g0)
    eng2.adaptive_set_options(max_flips=130, step_size=-1.0, target=-1.0)
    for _ in range(40):
        eng2.adaptive_sweep(1)
    assert all(np.array_equal(a, b) for a, b in zip(eng.get_states(), eng2.get_states()))
