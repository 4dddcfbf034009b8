"""The pybind11 layer (boom_amd._boom): the reference's Python driver loop
(Interfaces/python/spikeslab/BayesBoom/spikeslab/spikeslab.py:157-207) written
against `boom.*` names runs unchanged with `import boom_amd._boom as boom`, and
what it records -- chain 0 through the model's classic accessors -- is the
oracle's chain (gamma bit-exact, beta / sigma within 1e-8)."""
import numpy as np
import pytest

from cases import regression_data, spike_slab_prior
from oracle_lib import ssvs_options

pytestmark = pytest.mark.gpu


def _lm_spike_loop(boom, X, y, prior, niter, seed, chains, lookahead=1):
    """the body of lm_spike.__init__, boom.* calls only"""
    model = boom.RegressionModel(X, y, False, chains=chains, seed=seed)
    slab = boom.MvnGivenScalarSigma(prior["b"], prior["ominv"])
    siginv_prior = boom.ChisqModel(prior["df"], prior["sigma_guess"])
    spike = boom.VariableSelectionPrior(prior["pi"])
    sampler = boom.BregVsSampler(model, slab, siginv_prior, spike)
    if lookahead > 1:
        sampler.set_lookahead(lookahead)
    model.set_method(sampler)
    xdim = model.xdim
    coefficient_draws = np.zeros((niter, xdim))
    residual_sd = np.zeros(niter)
    model.coef.drop_all()
    model.coef.add(0)
    for i in range(niter):
        model.sample_posterior()
        residual_sd[i] = model.sigma
        coefficient_draws[i, :] = model.coef.Beta
    return model, sampler, coefficient_draws, residual_sd


@pytest.mark.parametrize("lookahead", [1, 25])
def test_lm_spike_driver_loop_on_the_pybind_module(oracle, lookahead):
    import boom_amd._boom as boom
    n, p, nsig, niter, seed, chains = 600, 30, 5, 60, 31337, 16
    X, y, _ = regression_data(n, p, nsig, seed=21)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, nsig)
    model, sampler, draws, sd = _lm_spike_loop(boom, X, y, prior, niter, seed, chains, lookahead)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    # (the engine's X'X comes from the MFMA build: use it for the oracle too)
    o = oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", seed, 0), g0, niter)
    assert o["status"] == 0
    for i in range(niter):
        assert np.array_equal(draws[i] != 0, o["gamma"][i] != 0), i
        err = np.max(np.abs(draws[i] - o["beta"][i]) / np.maximum(np.abs(o["beta"][i]), 1e-3))
        assert err < 1e-8, (i, err)
        assert abs(sd[i] ** 2 - o["sigsq"][i]) < 1e-8 * o["sigsq"][i]
    assert np.isfinite(sampler.logpri())
    G, B, S = model.chain_states()
    assert G.shape == (chains, p) and B.shape == (chains, p) and S.shape == (chains,)
    assert np.array_equal(G[0].astype(bool), np.array(model.coef.inc))
    ol = oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", seed, chains - 1), g0, niter)
    if lookahead == 1:
        assert np.array_equal(G[-1], ol["gamma"][-1])
