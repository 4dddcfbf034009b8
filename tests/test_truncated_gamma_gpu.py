"""sigma^2 draws whose truncation point lies beyond the mode of the gamma:
rtrun_gamma_mt's adaptive-rejection (a > 1) and slice (a <= 1) branches
(distributions/trun_gamma.cpp:74-100), on the device, inside the sweep kernels.
VERDICT r1 item 10.

The adaptive-rejection hull is kept across the lanes of the wave (lane i =
point i), so every kernel carries the complete sampler at no register cost; the
draws must stay the oracle's, sweep for sweep.
"""
import numpy as np
import pytest

from cases import bsts_priors, regression_data, spike_slab_prior, state_space_data
from oracle_lib import ssvs_options
from test_ssvs_gpu import make_engine, relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def _compare(oracle, eng, suf, prior, opts, seed, g0, nsw, check, step):
    ora = {c: oracle.ssvs_run(suf, prior, opts, ("philox", seed, c), g0, nsw) for c in check}
    done = 0
    while done < nsw:
        eng.sweep(step)
        done += step
        gam, beta, sig = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            assert np.array_equal(gam[c], o["gamma"][done - 1]), (c, done)
            assert relerr(beta[c], o["beta"][done - 1]) < RTOL, (c, done)
            assert abs(sig[c] - o["sigsq"][done - 1]) < RTOL * sig[c], (c, done)
    return ora


@pytest.mark.parametrize("limit,kind", [(0.9, "always"), (1.0, "sometimes"), (1e-3, "far tail")])
def test_sigma_upper_limit_beyond_the_mode(oracle, limit, kind):
    """residual sd 1: a limit of 0.9 puts every draw in the adaptive-rejection
    branch, a limit of 1.0 only some of them, 1e-3 is the far tail (what round 1
    reported as an error)."""
    X, y, _ = regression_data(400, 24, 4, seed=12)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 4)
    opts = ssvs_options(sigma_upper_limit=limit)
    g0 = np.zeros(24, np.uint8)
    g0[0] = 1
    chains, seed, nsw = 40, 9, 36
    eng = make_engine(chains, seed, suf=suf, prior=prior, opts=opts, g0=g0)
    ora = _compare(oracle, eng, suf, prior, opts, seed, g0, nsw, [0, 17, 39], step=12)
    _, _, sig = eng.get_states()
    assert np.all(sig <= limit * limit * (1 + 1e-12))
    if kind == "sometimes":
        # both regimes occurred: draws well below the limit and draws pressed against it
        s = np.concatenate([o["sigsq"] for o in ora.values()])
        assert s.min() < 0.95 * limit * limit
    # asynchronous launches, draw records and the look-ahead buffer see the same draws
    eng2 = make_engine(chains, seed, suf=suf, prior=prior, opts=opts, g0=g0)
    eng2.enable_draws(nsw)
    eng2.sweep(nsw, sync=False)    # (the record holds one launch's draws)
    eng2.sync()
    for c in (0, 39):
        gam, beta, sg = eng2.get_draws(c, nsw)
        assert np.array_equal(gam, ora[c]["gamma"])
        assert relerr(sg, ora[c]["sigsq"], 1e-12) < RTOL
    eng3 = make_engine(chains, seed, suf=suf, prior=prior, opts=opts, g0=g0)
    eng3.set_lookahead(7)
    for s in range(nsw):
        eng3.draw_next()
        gam, beta, sg = eng3.get_states()
        for c in (0, 39):
            assert np.array_equal(gam[c], ora[c]["gamma"][s]), (c, s)
            assert abs(sg[c] - ora[c]["sigsq"][s]) < RTOL * sg[c], (c, s)


def test_large_models_with_a_tight_limit(oracle):
    """the same branch in the large-model (k > 64) kernel"""
    X, y, _ = regression_data(1000, 100, 70, seed=3)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 70)
    opts = ssvs_options(sigma_upper_limit=0.95)
    g0 = np.zeros(100, np.uint8)
    g0[0] = 1
    eng = make_engine(4, 21, suf=suf, prior=prior, opts=opts, g0=g0)
    ora = _compare(oracle, eng, suf, prior, opts, 21, g0, 12, [0, 3], step=4)
    assert max(o["gamma"].sum(axis=1).max() for o in ora.values()) > 64


def test_adaptive_sampler_with_a_tight_limit(oracle):
    X, y, _ = regression_data(400, 24, 4, seed=12)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 4)
    opts = ssvs_options(sigma_upper_limit=0.9)
    g0 = np.zeros(24, np.uint8)
    g0[0] = 1
    eng = make_engine(6, 5, suf=suf, prior=prior, opts=opts, g0=g0)
    nsw = 20
    ora = {c: oracle.adaptive_run(suf, prior, opts, ("philox", 5, c), g0, nsw) for c in (0, 5)}
    eng.adaptive_sweep(nsw)
    gam, beta, sig = eng.get_states()
    for c in (0, 5):
        o = ora[c]
        assert o["status"] == 0
        assert np.array_equal(gam[c], o["gamma"][-1])
        assert relerr(beta[c], o["beta"][-1]) < RTOL
        assert abs(sig[c] - o["sigsq"][-1]) < RTOL * sig[c]


@pytest.mark.parametrize("T", [2, 3, 5])
def test_short_series_level_variance(oracle, T):
    """T - 1 state innovations and a prior df of 0.01: shape <= 1, the slice
    sampler draws the level variance (and the observation variance's limit of
    1.2 sd(y) binds too)"""
    import boom_amd
    p = 4
    X, y, _, obs = state_space_data(T, p, 2, seed=200 + T)
    prior, ss, sig_up = bsts_priors(X, y, 2)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = boom_amd.Engine(5, seed=17)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"], sigma_upper_limit=sig_up)
    eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"],
                           ss["level_sigma_upper_limit"], ss["initial_state_mean"],
                           ss["initial_state_variance"], ss["initial_level_sigma"])
    eng.set_state(g0)
    nsw = 30
    ora = {c: oracle.ss_run(y, X, obs, prior, opts, ss, ("philox", 17, c), g0, nsw)
           for c in (0, 4)}
    for s in range(nsw):
        eng.ss_sweep(1)
        gam, beta, sig = eng.get_states()
        for c in (0, 4):
            o = ora[c]
            assert o["status"] == 0
            assert np.array_equal(gam[c], o["gamma"][s]), (c, s)
            assert relerr(beta[c], o["beta"][s]) < RTOL, (c, s)
            assert abs(sig[c] - o["sigsq"][s]) < RTOL * sig[c], (c, s)
            st = eng.ss_get_state(c)
            assert abs(st["level_sigsq"] - o["level_sigsq"][s]) < RTOL * st["level_sigsq"], (c, s)
            assert np.max(np.abs(st["state"] - o["state"][s])) < 1e-8 * np.abs(o["state"][s]).max()


def test_single_observation_small_shape_gamma(oracle):
    """one observation, no upper limits, level prior df 0.5: the level variance
    is an untruncated gamma of shape 0.25 -- rgamma's rloggamma_small_alpha
    branch (Bmath/rloggamma_small_alpha.cpp:43-79)"""
    import boom_amd
    T, p = 1, 3
    X, y, _, _ = state_space_data(8, p, 2, seed=201)
    prior, ss, _ = bsts_priors(X, y, 2)
    X, y = X[:T], y[:T]
    ss = dict(ss, level_df=0.5, level_sigma_upper_limit=np.inf)
    opts = ssvs_options(sigma_upper_limit=np.inf)
    g0 = np.zeros(p, np.uint8)
    eng = boom_amd.Engine(3, seed=41)
    eng.ss_set_data(y, X, None)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"], sigma_upper_limit=np.inf)
    eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"],
                           ss["level_sigma_upper_limit"], ss["initial_state_mean"],
                           ss["initial_state_variance"], ss["initial_level_sigma"])
    eng.set_state(g0)
    nsw = 40
    ora = {c: oracle.ss_run(y, X, None, prior, opts, ss, ("philox", 41, c), g0, nsw)
           for c in (0, 2)}
    for s in range(nsw):
        eng.ss_sweep(1)
        gam, beta, sig = eng.get_states()
        for c in (0, 2):
            o = ora[c]
            assert o["status"] == 0
            assert np.array_equal(gam[c], o["gamma"][s]), (c, s)
            assert abs(sig[c] - o["sigsq"][s]) < RTOL * sig[c], (c, s)
            st = eng.ss_get_state(c)
            assert abs(st["level_sigsq"] - o["level_sigsq"][s]) < RTOL * st["level_sigsq"], (c, s)
            assert np.max(np.abs(st["state"] - o["state"][s])) < 1e-8 * np.abs(o["state"][s]).max()
